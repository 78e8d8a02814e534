// Host-side (double precision) weight folding and packing into the kernel layouts.
#pragma once
#include "common.h"

#include <cmath>

namespace fc {

struct MatD {
    int rows = 0, cols = 0;
    std::vector<double> v;
    MatD() {}
    MatD(int r, int c) : rows(r), cols(c), v((size_t)r * c, 0.0) {}
    double& at(int r, int c) { return v[(size_t)r * cols + c]; }
    double at(int r, int c) const { return v[(size_t)r * cols + c]; }
};
typedef std::vector<double> VecD;

MatD mat_from(const HostTensor& t);               // 2-D (or [N,K,1,1] conv) tensor -> N x K
VecD vec_from(const HostTensor& t);
MatD matmul(const MatD& a, const MatD& b);
MatD expm_double(const MatD& w);                  // scaling & squaring Taylor, double
double slogdet_abs(const MatD& w);                // log |det w| via partial-pivot LU
MatD inverse_double(const MatD& w);               // Gauss-Jordan with partial pivoting
inline double softplus_d(double x) { return x > 30 ? x : std::log1p(std::exp(x)); }

// index maps (packed index -> source index, -1 = zero fill)
std::vector<int> map_prefix(int n_src, int n_pad);                 // [0..n_src) then zeros
std::vector<int> map_xlayout(int d1, int d1_pad, int d2, int d2_pad);   // engine x layout -> latent index
std::vector<int> map_pairs(int d2, int second_half_offset);        // [first 32 | second 32] x ceil(d2/32)
std::vector<int> map_concat(const std::vector<std::vector<int>>& parts);

// bf16 limb image of a packed fp32 weight matrix (host side): [n_alloc][K_pad/16][3][16], see PackedLinear.W3
std::vector<unsigned short> make_f16_limbs(const std::vector<float>& w, int n_alloc, int K_pad);
std::vector<unsigned short> make_bf16_limbs(const std::vector<float>& w, int n_alloc, int K_pad);

// Build a PackedLinear: W_src [N_src x K_src], bias [N_src] (may be empty), colvec [N_src] (may be empty)
PackedLinear pack_linear(DeviceArena& arena, const MatD& W, const VecD& bias, const VecD& colvec, const std::vector<int>& nmap,
                         const std::vector<int>& kmap, const std::vector<int>& seg_k);

// One reference MLP (models/nets.py) as packed GEMMs. The first layer is built by the caller (its input layout varies).
struct PackedMLP {
    PackedLinear in_layer;
    std::vector<PackedLinear> mid;
    PackedLinear out_layer;
    std::vector<int> sizes;       // hidden widths
};
// packs layers.{i} of `prefix` (plain [round32(prev)] -> [round32(next)]) and returns hidden sizes; in/out are left to the caller
void pack_mlp_mid(DeviceArena& arena, const WeightTable& wt, const std::string& prefix, PackedMLP& out);
int max_hidden_pad(const PackedMLP& m);
// Row-resident coupling MLP (mlprows.hip): gives the K <= 512 -> 512 -> ... -> 512 layers of a coupling net their second fp16 limb image in
// MFMA-fragment order (PackedLinear::Wf), built on the device from W2; a net of any other shape is left as it is.
void attach_mlp_rows_images(DeviceArena& arena, PackedMLP& m);

// Runs in_layer + hidden layers of one reference MLP (models/nets.py:19-30: act(in); even hidden layer i: keep = x, x = act(W x);
// odd: x = act(keep + W x)) over three rotating activation buffers h[0..2] of pitch ldh; returns the buffer index holding the
// last hidden activation (the caller applies out_layer with the epilogue it needs).  With `last_limbs` the LAST hidden layer writes its
// output only as an fp16 limb image (GemmEpi::C16, pitch = its N_pad) and -1 is returned.
int run_mlp_hidden_generic(const PackedMLP& m, const ASeg* in_segs, const float* rowscal, int act, float* const h[3], int ldh, int rows,
                           hipStream_t s, int rows_valid = 0, unsigned short* last_limbs = nullptr, float last_scale = 0.f, bool wide = false);   // last_scale: GemmEpi::c16_scale of that image

}  // namespace fc
