// fc_flow: the conditional normalizing flow (augmenter -> n x [pre-conditioner -> coupling -> ActNorm -> permuter]
// -> base density) as a schedule of HIP kernel launches over a caller-owned workspace.
//
// Reference being replaced: models.Flow.log_prob (models/transform.py:70-76) over the transform list that
// initialize_flow assembles (model_initialization.py:136-160).  Weight folding done once at create (double precision):
//   * attn.fn.lin (I -> attn_dim) is folded INTO the coupling / augmenter in_layer:  W_ctx (W_lin a + b_lin) = (W_ctx W_lin) a + W_ctx b_lin
//   * LayerNorm gamma/beta, the softmax scale inner^-0.5 and log2(e) are folded into the q projection
//   * ActNorm and the permuter (LinearLU: L U; FullCombiner: w; ExponentialCombiner: expm; Permuter: P) become ONE matrix
//     z = W' x + b',  W' = P diag(e^-log_scale),  b' = -W' shift; their log-dets are data independent and summed into one constant
//   * extra context (one scalar per scene) enters every in_layer as a rank-1 epilogue term instead of a concatenated column
// Activation layout in HBM: x is [rows, d1_pad + d2_pad] = [x1 | 0-pad | x2 | 0-pad] (pads kept zero by construction), every
// other activation is [rows, round_up(width, 32)]; rows are padded to 256.
#include <algorithm>
#include <cmath>
#include <memory>
#include <cstring>

#include <atomic>
#include <exception>
#include <mutex>
#include <thread>

#include "hostpack.h"
#include "spline.h"

namespace fc {

struct AttnPack {
    PackedLinear q;      // LN-folded, pre-scaled q projection  [I_pad][A_in_pad]
    MatD q_w;            // the same folded matrix / bias on the host (double), for the LayerNorm -> q fold below
    VecD q_b;
    // LayerNorm folded THROUGH the (activation-free) pre-MLP out_layer: rows [0, A_in) = mean-centred out_layer (its outputs are only
    // squared and summed per row), rows [A_in, A_in + I_pad) = q projection of the centred outputs; q = q_unnorm * rstd + q_bias
    PackedLinear lnq;
    float* q_bias = nullptr;
    bool has_lnq = false;
    MatD lin_w;          // [attn_dim][I]  (folded into the consumer's in_layer)
    VecD lin_b;
    int kv_col = 0;      // column of this layer's [K | V] block inside the kv buffer
};

// CIFblock pieces (models/cif_block.py:49-112), all expressed in the NATURAL index order of x (D) and z2 (Dc - D): the two
// Reverse permutations are folded into the packed weights' row / column maps.
struct CifPack {
    PackedMLP dist;               // shared ConditionalNormal net of augmenter and slicer: x (x layout) -> [mean | log_std] pairs
    PackedMLP aff;                // affine_cif: flip(z2) -> (s, t) for flip(x); t rows carry the x-part ActNorm
    float* post_scale = nullptr;  // g[k] = exp(-log_scale) of the x part (behind s)
    float* z2_shift = nullptr;    // ActNorm of the z2 part: v = (z2 - shift) * scale
    float* z2_scale = nullptr;
    double log_const = 0.0;       // data-independent log-det of the CIF ActNorm
};

struct BlockPack {
    bool has_attn = false;
    PackedMLP pre;       // pre_attention_mlp
    AttnPack attn;
    PackedMLP net;       // coupling MLP (in_layer has the folded context segment)
    float* expm_scal = nullptr;   // ExponentialCoupling: {scale, shift, rescale, reshift}
    bool has_cif = false;
    CifPack cif;
    bool has_lin = false;
    PackedLinear lin;    // folded ActNorm + permuter (absent after the last block)
    MatD lin_w;          // host copy (double) for the lazily built inverse
    VecD lin_b;
    bool has_lin_inv = false;
    PackedLinear lin_inv;
    double log_const = 0.0;       // data-independent log-dets of this block's ActNorm + permuter
};

struct Dims {
    int Din, D, d1, d2, d1_pad, d2_pad, ldx;
    int E, E_pad, X;
    int A_in = 0, A_in_pad = 0, I = 0, I_pad = 0;
    int H_pad = 0;       // widest hidden activation
    int ldp = 0;         // spline / expm parameter pitch
    int Dc = 0, nz = 0, nz_pad = 0;   // CIF: cif_latent_dim, Dc - D
};

}  // namespace fc

struct fc_flow {
    int* fp16_flag = nullptr;   // device word raised by the split-fp16 GEMM loop on an activation >= 65504 (common.h: Fp16Guard)
    fc_flow_config cfg;
    fc::Dims d;
    fc::DeviceArena arena;
    bool has_augment = false;
    fc::PackedMLP aug_pre, aug_net;
    fc::AttnPack aug_attn;
    std::vector<fc::BlockPack> blocks;
    fc::PackedLinear kv_all;   // ctx -> [K|V] of every attention (augmenter first)
    int n_attn = 0;
    double log_const = 0.0;
};

namespace fc {

static int pad_inner(int I) {
    if (I <= 32) return 32;
    if (I <= 64) return 64;
    if (I <= 128) return 128;
    throw Error(FC_ERR_UNSUPPORTED, "attention inner dim (cross_heads*cross_dim_head) > 128 is not supported yet");
}

// q' = c * Wq (gamma . n + beta),  c = inner^-0.5 * log2(e)   (models/perceiver.py:18-26, 96-110)
static void build_attn(fc_flow& f, const WeightTable& wt, const std::string& p, AttnPack& out, std::vector<MatD>& kv_rows, int slot) {
    Dims& d = f.d;
    const HostTensor& wq_t = wt.get(p + ".fn.attention.to_q.weight");
    if (wq_t.shape.size() != 2) throw Error(FC_ERR_SHAPE, p + ".fn.attention.to_q.weight must be 2-D");
    const int I = (int)wq_t.shape[0], A_in = (int)wq_t.shape[1];
    if (d.I == 0) { d.I = I; d.I_pad = pad_inner(I); d.A_in = A_in; d.A_in_pad = round_up(A_in, 32); }
    if (I != d.I || A_in != d.A_in) throw Error(FC_ERR_SHAPE, p + ": all attention blocks must share inner / input dims");
    MatD wq = mat_from(wq_t);
    VecD gamma = vec_from(wt.get(p + ".norm.weight", {A_in})), beta = vec_from(wt.get(p + ".norm.bias", {A_in}));
    const double c = std::pow((double)I, -0.5) * 1.4426950408889634074;
    VecD bq(I, 0.0);
    for (int i = 0; i < I; ++i)
        for (int k = 0; k < A_in; ++k) {
            bq[i] += c * wq.at(i, k) * beta[k];
            wq.at(i, k) *= c * gamma[k];
        }
    out.q = pack_linear(f.arena, wq, bq, {}, map_prefix(I, d.I_pad), map_prefix(A_in, d.A_in_pad), {d.A_in_pad});
    out.q_w = wq;
    out.q_b = bq;
    const HostTensor& wkv_t = wt.get(p + ".fn.attention.to_kv.weight", {2 * I, d.E});
    MatD wkv = mat_from(wkv_t);                 // rows [0,I) = K, [I,2I) = V  (chunk(2, dim=-1))
    MatD blk(2 * d.I_pad, d.E);
    for (int i = 0; i < I; ++i)
        for (int k = 0; k < d.E; ++k) { blk.at(i, k) = wkv.at(i, k); blk.at(d.I_pad + i, k) = wkv.at(I + i, k); }
    out.kv_col = slot * 2 * d.I_pad;            // column block of this attention in the stacked K|V projection
    kv_rows[slot] = blk;
    const HostTensor& wl = wt.get(p + ".fn.lin.weight");
    if (wl.shape.size() != 2 || wl.shape[1] != I) throw Error(FC_ERR_SHAPE, p + ".fn.lin.weight: expected [attn_dim, inner]");
    out.lin_w = mat_from(wl);
    out.lin_b = vec_from(wt.get(p + ".fn.lin.bias", {wl.shape[0]}));
}

// LayerNorm -> q fold (see AttnPack::lnq).  h = W3 a + b3 has no activation, so its centred form h_c = h - mean(h) is linear in a:
// W3c = W3 - 1 (1^T W3)/A_in, b3c = b3 - mean(b3).  LayerNorm(h) = h_c / sigma (gamma, beta live in the q projection), hence
// q = Wq' h_c / sigma + bq' = (Wq' W3c a + Wq' b3c) / sigma + bq' with sigma^2 = mean(h_c^2) + eps.  One GEMM with N = A_in + I_pad
// columns yields h_c (only squared and summed per row in the epilogue, never stored) and q_unnorm; lnq_finalize_kernel applies
// rstd and bq'.  Replaces out_layer's store, the LayerNorm pass and the q projection GEMM.
static void build_lnq(fc_flow& f, const WeightTable& wt, const std::string& out_prefix, AttnPack& at) {
    const Dims& d = f.d;
    const HostTensor& w_t = wt.get(out_prefix + ".weight");
    const MatD w3 = mat_from(w_t);
    const int A = w3.rows, K = w3.cols;
    if (A != d.A_in || d.A_in != d.A_in_pad || d.A_in % 64 != 0 || d.I_pad != 64 || K % 32 != 0) return;     // shapes the fused epilogue handles
    const VecD b3 = vec_from(wt.get(out_prefix + ".bias", {A}));
    MatD m(A + d.I_pad, K);
    VecD bias(A + d.I_pad, 0.0);
    double bmean = 0.0;
    for (int o = 0; o < A; ++o) bmean += b3[o] / A;
    for (int k = 0; k < K; ++k) {
        double cm = 0.0;
        for (int o = 0; o < A; ++o) cm += w3.at(o, k) / A;
        for (int o = 0; o < A; ++o) m.at(o, k) = w3.at(o, k) - cm;
    }
    for (int o = 0; o < A; ++o) bias[o] = b3[o] - bmean;
    for (int i = 0; i < at.q_w.rows; ++i) {
        for (int k = 0; k < K; ++k) {
            double acc = 0.0;
            for (int o = 0; o < A; ++o) acc += at.q_w.at(i, o) * m.at(o, k);
            m.at(A + i, k) = acc;
        }
        double acc = 0.0;
        for (int o = 0; o < A; ++o) acc += at.q_w.at(i, o) * bias[o];
        bias[A + i] = acc;
    }
    at.lnq = pack_linear(f.arena, m, bias, {}, map_prefix(A + d.I_pad, A + d.I_pad), map_prefix(K, K), {K});
    std::vector<float> qb(d.I_pad, 0.f);
    for (int i = 0; i < at.q_w.rows; ++i) qb[i] = (float)at.q_b[i];
    at.q_bias = f.arena.upload(qb);
    at.has_lnq = at.lnq.W2 != nullptr;
}

// in_layer over cat(first(n_first), extra(X), ctxvec(C)) -> packed [first_pad | second_pad] + rank-1 extra column.
// With attention the context segment is the folded attention output (I_pad wide); in global mode it is the E-wide embedding.
static PackedLinear build_in_layer(fc_flow& f, const WeightTable& wt, const std::string& prefix, int n_first, int first_pad,
                                   const AttnPack* attn) {
    Dims& d = f.d;
    const HostTensor& wt_in = wt.get(prefix + ".in_layer.weight");
    MatD w = mat_from(wt_in);
    const int H = w.rows;
    VecD b = vec_from(wt.get(prefix + ".in_layer.bias", {H}));
    const int C = attn ? attn->lin_w.rows : d.E;
    if (w.cols != n_first + d.X + C) throw Error(FC_ERR_SHAPE, prefix + ".in_layer.weight: expected input width " +
                                                                  std::to_string(n_first + d.X + C) + ", got " + std::to_string(w.cols));
    const int c0 = n_first + d.X;
    const int second = attn ? d.I : d.E, second_pad = attn ? d.I_pad : d.E_pad;
    MatD fw(H, n_first + second);
    VecD colvec;
    if (d.X) colvec.assign(H, 0.0);
    for (int n = 0; n < H; ++n) {
        for (int k = 0; k < n_first; ++k) fw.at(n, k) = w.at(n, k);
        if (d.X) colvec[n] = w.at(n, n_first);
        if (attn) {
            for (int j = 0; j < d.I; ++j) {
                double s = 0;
                for (int c = 0; c < C; ++c) s += w.at(n, c0 + c) * attn->lin_w.at(c, j);
                fw.at(n, n_first + j) = s;
            }
            double sb = 0;
            for (int c = 0; c < C; ++c) sb += w.at(n, c0 + c) * attn->lin_b[c];
            b[n] += sb;
        } else {
            for (int c = 0; c < C; ++c) fw.at(n, n_first + c) = w.at(n, c0 + c);
        }
    }
    std::vector<int> k2(second_pad, -1);
    for (int j = 0; j < second; ++j) k2[j] = n_first + j;
    return pack_linear(f.arena, fw, b, colvec, map_prefix(H, round_up(H, 32)), map_concat({map_prefix(n_first, first_pad), k2}),
                       {first_pad, second_pad});
}

static PackedLinear build_plain(fc_flow& f, const WeightTable& wt, const std::string& name, int k_src, int k_pad) {
    const HostTensor& w = wt.get(name + ".weight");
    if (w.shape.size() != 2 || w.shape[1] != k_src) throw Error(FC_ERR_SHAPE, name + ".weight: unexpected input width");
    const int n = (int)w.shape[0];
    return pack_linear(f.arena, mat_from(w), vec_from(wt.get(name + ".bias", {n})), {}, map_prefix(n, round_up(n, 32)),
                       map_prefix(k_src, k_pad), {k_pad});
}

// ActNorm (models/act_norm.py:37-43) followed by the permuter (models/permuters.py) as one affine map on the x layout.
static void build_lin(fc_flow& f, const WeightTable& wt, int idx_actnorm, int idx_perm, BlockPack& blk) {
    const Dims& d = f.d;
    const int D = d.D;
    VecD shift(D, 0.0), ls(D, 0.0);
    if (idx_actnorm >= 0) {
        const std::string p = "transforms." + std::to_string(idx_actnorm);
        shift = vec_from(wt.get(p + ".shift", {1, D}));
        ls = vec_from(wt.get(p + ".log_scale", {1, D}));
        for (double v : ls) blk.log_const -= v;
    }
    const std::string p = "transforms." + std::to_string(idx_perm);
    MatD Wp(D, D);
    switch (f.cfg.permuter_type) {
        case FC_PERM_LINEAR_LU: {
            const int ntri = D * (D - 1) / 2;
            const HostTensor& lo = wt.get(p + ".lower_entries", {ntri});
            const HostTensor& up = wt.get(p + ".upper_entries", {ntri});
            const HostTensor& ud = wt.get(p + ".unconstrained_upper_diag", {D});
            MatD L(D, D), U(D, D);
            int t = 0;
            for (int i = 0; i < D; ++i) { for (int j = 0; j < i; ++j) L.at(i, j) = lo.data[t++]; L.at(i, i) = 1.0; }
            t = 0;
            for (int i = 0; i < D; ++i) for (int j = i + 1; j < D; ++j) U.at(i, j) = up.data[t++];
            for (int i = 0; i < D; ++i) {
                const double dg = softplus_d(ud.data[i]) + (double)f.cfg.linear_lu_eps;
                U.at(i, i) = dg;
                blk.log_const += std::log(dg);
            }
            Wp = matmul(L, U);               // z = L (U x)   (permuters.py:164-169)
            break;
        }
        case FC_PERM_RANDOM: {
            const HostTensor& pm = wt.get(p + ".permutation", {D});
            for (int i = 0; i < D; ++i) {
                const int src = (int)std::lround(pm.data[i]);
                if (src < 0 || src >= D) throw Error(FC_ERR_INVALID, p + ".permutation out of range");
                Wp.at(i, src) = 1.0;          // y = x.index_select(-1, permutation)
            }
            break;
        }
        case FC_PERM_FULL: {
            Wp = mat_from(wt.get(p + ".w", {D, D}));
            blk.log_const += slogdet_abs(Wp);
            break;
        }
        case FC_PERM_EXPONENTIAL: {
            MatD w = mat_from(wt.get(p + ".w", {D, D}));
            const double sc = wt.get(p + ".scale", {1}).data[0], sh = wt.get(p + ".shift", {1}).data[0];
            const double rs = wt.get(p + ".rescale", {1}).data[0], rsh = wt.get(p + ".reshift", {1}).data[0];
            for (auto& e : w.v) e = rs * std::tanh(sc * e + sh) + rsh + 1e-8;
            for (int i = 0; i < D; ++i) blk.log_const += w.at(i, i);
            Wp = expm_double(w);
            break;
        }
        default: throw Error(FC_ERR_INVALID, "unknown permuter_type");
    }
    VecD b(D, 0.0);
    for (int i = 0; i < D; ++i) {
        double s = 0;
        for (int k = 0; k < D; ++k) {
            Wp.at(i, k) *= std::exp(-ls[k]);
            s += Wp.at(i, k) * shift[k];
        }
        b[i] = -s;
    }
    const std::vector<int> xl = map_xlayout(d.d1, d.d1_pad, d.d2, d.d2_pad);
    blk.lin = pack_linear(f.arena, Wp, b, {}, xl, xl, {d.ldx});
    blk.has_lin = true;
    blk.lin_w = Wp;
    blk.lin_b = b;
}

static void build_out_layer(fc_flow& f, const WeightTable& wt, const std::string& prefix, PackedMLP& net) {
    Dims& d = f.d;
    const int hl = net.sizes.back();
    const HostTensor& w = wt.get(prefix + ".out_layer.weight");
    const int n = (int)w.shape[0];
    VecD b = vec_from(wt.get(prefix + ".out_layer.bias", {n}));
    std::vector<int> nmap;
    if (f.cfg.flow_type == FC_FLOW_AFFINE) {
        if (n != 2 * d.d2) throw Error(FC_ERR_SHAPE, prefix + ".out_layer: affine coupling expects 2*(D - D/2) outputs");
        nmap = map_pairs(d.d2, d.d2);
    } else if (f.cfg.flow_type == FC_FLOW_SPLINE) {
        const int per = 3 * f.cfg.num_bins_spline + 1;
        if (n != per * d.d1) throw Error(FC_ERR_SHAPE, prefix + ".out_layer: spline coupling expects (3K+1)*(D/2) outputs");
        if (n != per * d.d2) throw Error(FC_ERR_UNSUPPORTED, "spline coupling with odd latent_dim fails in the reference too (reshape)");
        // tile-grouped dim-major output (spline.h): a 128-column GEMM tile holds all 3K+1 parameters of DPT transformed dims, so the
        // workgroup that produced the tile evaluates those splines in its epilogue (reference order is j*(3K+1) + p)
        const int K = f.cfg.num_bins_spline;
        nmap.assign(spline_ncols(d.d2, K), -1);
        for (int j = 0; j < d.d2; ++j)
            for (int pp = 0; pp < per; ++pp) nmap[spline_col(j, pp, K)] = j * per + pp;
    } else {
        if (n != d.d2 * d.d2 + d.d2) throw Error(FC_ERR_SHAPE, prefix + ".out_layer: exponential coupling expects d2^2 + d2 outputs");
        nmap = map_prefix(n, round_up(n, 32));
    }
    net.out_layer = pack_linear(f.arena, mat_from(w), b, {}, nmap, map_prefix(hl, round_up(hl, 32)), {round_up(hl, 32)});
    if (f.cfg.flow_type == FC_FLOW_SPLINE && f.cfg.num_bins_spline == 8) {
        // the one-accumulator image of the 256 x 256 fused spline kernel (spline_wide.hip): rows in that kernel's register-slot order, pre-scaled by
        // the power of two that puts max |w| into [2^14, 2^15)
        float wmax = 0.f;
        const int64_t nel = w.numel();
        for (int64_t i = 0; i < nel; ++i) wmax = std::max(wmax, std::fabs(w.data[i]));
        spline_wide_attach(f.arena, net.out_layer, wmax, nullptr);
    }
}

// pair-packed row map from explicit (first-half row, second-half row) lists
static std::vector<int> map_pairs_rows(const std::vector<int>& first, const std::vector<int>& second) {
    const int n = (int)first.size(), np = (n + 31) / 32;
    std::vector<int> m(np * 64, -1);
    for (int j = 0; j < n; ++j) { m[64 * (j / 32) + j % 32] = first[j]; m[64 * (j / 32) + 32 + j % 32] = second[j]; }
    return m;
}

// CIFblock (models/cif_block.py:49-112).  Natural-order algebra (x: D dims, z2: nz = Dc - D dims, Reverse folded away):
//   z2 = mu(x) + eps sigma(x)                                   ldj -= log N(z2)
//   (s,t) = affine_cif.nn(flip(z2)),  zx[k] = (x[k] s'[k] + t'[k] - shift[Dc-1-k]) e^{-log_scale[Dc-1-k]},  s'[k] = s[D-1-k]   ldj += sum log s
//   x2n[j] = (z2[j] - shift[nz-1-j]) e^{-log_scale[nz-1-j]}      ldj += sum(-log_scale)  (constant)
//   ldj += log N(x2n; mu(zx), sigma(zx))                         (Slice with the SAME distribution object)
//   then the attention-conditioned coupling on zx.
static void build_cif(fc_flow& f, const WeightTable& wt, const std::string& p, CifPack& c) {
    Dims& d = f.d;
    const int D = d.D, nz = d.nz, Dc = d.Dc;
    const std::string pd = p + ".augmenter.noise_dist.net";
    if (wt.has(p + ".slicer.noise_dist.net.in_layer.weight")) {          // shared object: both prefixes must hold the same values
        const HostTensor& a = wt.get(pd + ".in_layer.weight");
        const HostTensor& b = wt.get(p + ".slicer.noise_dist.net.in_layer.weight");
        if (a.shape != b.shape || memcmp(a.data, b.data, sizeof(float) * (size_t)a.numel()) != 0)
            throw Error(FC_ERR_INVALID, p + ": augmenter.noise_dist and slicer.noise_dist must be identical (one shared ConditionalNormal)");
    }
    pack_mlp_mid(f.arena, wt, pd, c.dist);
    {
        const HostTensor& w = wt.get(pd + ".in_layer.weight");
        if (w.shape.size() != 2 || w.shape[1] != D) throw Error(FC_ERR_SHAPE, pd + ".in_layer.weight: expected input width latent_dim");
        const int h = (int)w.shape[0];
        c.dist.in_layer = pack_linear(f.arena, mat_from(w), vec_from(wt.get(pd + ".in_layer.bias", {h})), {}, map_prefix(h, round_up(h, 32)),
                                      map_xlayout(d.d1, d.d1_pad, d.d2, d.d2_pad), {d.ldx});
        const int hl = c.dist.sizes.back();
        const HostTensor& wo = wt.get(pd + ".out_layer.weight", {2 * nz, hl});
        c.dist.out_layer = pack_linear(f.arena, mat_from(wo), vec_from(wt.get(pd + ".out_layer.bias", {2 * nz})), {}, map_pairs(nz, nz),
                                       map_prefix(hl, round_up(hl, 32)), {round_up(hl, 32)});
    }
    VecD shift = vec_from(wt.get(p + ".act_norm.shift", {1, Dc})), ls = vec_from(wt.get(p + ".act_norm.log_scale", {1, Dc}));
    for (double v : ls) c.log_const -= v;
    std::vector<float> g(gemm_n_alloc(round_up(D, 32) * 2), 0.f), sh2(gemm_n_alloc(round_up(nz, 32) * 2), 0.f), g2(sh2.size(), 1.f);
    for (int k = 0; k < D; ++k) g[k] = (float)std::exp(-ls[Dc - 1 - k]);
    for (int j = 0; j < nz; ++j) { sh2[j] = (float)shift[nz - 1 - j]; g2[j] = (float)std::exp(-ls[nz - 1 - j]); }
    c.post_scale = f.arena.upload(g);
    c.z2_shift = f.arena.upload(sh2);
    c.z2_scale = f.arena.upload(g2);
    const std::string pa = p + ".affine_cif.nn";
    pack_mlp_mid(f.arena, wt, pa, c.aff);
    {
        const HostTensor& w = wt.get(pa + ".in_layer.weight");
        if (w.shape.size() != 2 || w.shape[1] != nz) throw Error(FC_ERR_SHAPE, pa + ".in_layer.weight: expected input width cif_latent_dim - latent_dim");
        const int h = (int)w.shape[0];
        std::vector<int> km(d.nz_pad, -1);
        for (int j = 0; j < nz; ++j) km[j] = nz - 1 - j;                   // input arrives as z2 in natural order; the net saw flip(z2)
        c.aff.in_layer = pack_linear(f.arena, mat_from(w), vec_from(wt.get(pa + ".in_layer.bias", {h})), {}, map_prefix(h, round_up(h, 32)), km, {d.nz_pad});
        const int hl = c.aff.sizes.back();
        MatD wo = mat_from(wt.get(pa + ".out_layer.weight", {2 * D, hl}));
        VecD bo = vec_from(wt.get(pa + ".out_layer.bias", {2 * D}));
        std::vector<int> srow(D), trow(D);
        for (int k = 0; k < D; ++k) {
            const int i = D - 1 - k;                                       // position inside flip(x)
            srow[k] = i; trow[k] = D + i;
            const double gk = std::exp(-ls[Dc - 1 - k]);
            for (int c2 = 0; c2 < hl; ++c2) wo.at(D + i, c2) *= gk;          // t'' = (t - shift) g
            bo[D + i] = (bo[D + i] - shift[Dc - 1 - k]) * gk;
        }
        c.aff.out_layer = pack_linear(f.arena, wo, bo, {}, map_pairs_rows(srow, trow), map_prefix(hl, round_up(hl, 32)), {round_up(hl, 32)});
    }
}

static void build_flow(fc_flow& f, const WeightTable& wt) {
    const fc_flow_config& c = f.cfg;
    Dims& d = f.d;
    if (c.struct_size != (int)sizeof(fc_flow_config)) throw Error(FC_ERR_INVALID, "fc_flow_config.struct_size mismatch (ABI)");
    if (c.latent_dim < c.input_dim) throw Error(FC_ERR_INVALID, "Latent dim < Input dim");
    if (c.cif_latent_dim < c.latent_dim) throw Error(FC_ERR_INVALID, "Augment dim smaller than main latent!");
    const bool cif = c.cif_latent_dim > c.latent_dim;
    if (cif && c.extra_context_dim) throw Error(FC_ERR_INVALID, "Not implemented extra context with cif");
    if (cif && c.global_context) throw Error(FC_ERR_INVALID, "CIF + global embedding not implemented");
    if (c.n_flow_layers < 1 || c.latent_dim < 2) throw Error(FC_ERR_INVALID, "need n_flow_layers >= 1 and latent_dim >= 2");
    if (c.extra_context_dim < 0 || c.extra_context_dim > 1) throw Error(FC_ERR_UNSUPPORTED, "extra_context_dim must be 0 or 1");
    d.Din = c.input_dim; d.D = c.latent_dim; d.d1 = d.D / 2; d.d2 = d.D - d.d1;
    d.d1_pad = round_up(d.d1, 32); d.d2_pad = round_up(d.d2, 32); d.ldx = d.d1_pad + d.d2_pad;
    d.E = c.input_embedding_dim; d.E_pad = round_up(d.E, 32); d.X = c.extra_context_dim;
    if (d.Din > 32) throw Error(FC_ERR_UNSUPPORTED, "input_dim > 32");
    d.Dc = c.cif_latent_dim; d.nz = d.Dc - d.D; d.nz_pad = round_up(std::max(d.nz, 1), 32);
    f.has_augment = d.D > d.Din;
    const int aug_slots = f.has_augment ? 1 : 0;
    std::vector<MatD> kv_rows(aug_slots + (c.global_context ? 0 : c.n_flow_layers));
    std::mutex dims_mu;                          // d.H_pad / d.ldp maxima are the only shared writes of the per-layer builders

    // ---- transform 0: AugmentAttentionPreconditioner (models/augmenter.py:7-22) or IdentityTransform
    if (f.has_augment) {
        const std::string p = "transforms.0";
        build_attn(f, wt, p + ".attn", f.aug_attn, kv_rows, 0);
        pack_mlp_mid(f.arena, wt, p + ".pre_attn_mlp", f.aug_pre);
        f.aug_pre.in_layer = build_plain(f, wt, p + ".pre_attn_mlp.in_layer", d.Din, 32);
        f.aug_pre.out_layer = build_plain(f, wt, p + ".pre_attn_mlp.out_layer", f.aug_pre.sizes.back(), round_up(f.aug_pre.sizes.back(), 32));
        if (f.aug_pre.out_layer.N_pad != d.A_in_pad) throw Error(FC_ERR_SHAPE, "pre_attn_mlp output width != attn_input_dim");
        build_lnq(f, wt, p + ".pre_attn_mlp.out_layer", f.aug_attn);
        const std::string pn = p + ".augment.noise_dist.net";
        pack_mlp_mid(f.arena, wt, pn, f.aug_net);
        f.aug_net.in_layer = build_in_layer(f, wt, pn, d.Din, 32, &f.aug_attn);
        const int nz = d.D - d.Din, hl = f.aug_net.sizes.back();
        const HostTensor& wo = wt.get(pn + ".out_layer.weight", {2 * nz, hl});
        f.aug_net.out_layer = pack_linear(f.arena, mat_from(wo), vec_from(wt.get(pn + ".out_layer.bias", {2 * nz})), {}, map_pairs(nz, nz),
                                          map_prefix(hl, round_up(hl, 32)), {round_up(hl, 32)});
        d.H_pad = std::max({d.H_pad, max_hidden_pad(f.aug_pre), max_hidden_pad(f.aug_net), d.A_in_pad});
    }
    // ---- blocks.  Layer l's transforms are [block, ActNorm?, permuter] at indices 1 + l * stride ...; the layers are independent, so
    //      after layer 0 (which fixes the shared attention dims) they are packed by a pool of host threads: the double-precision folds
    //      and the fp32 packing of 370 M weights (C2) are the bulk of fc_flow_create's time.
    f.blocks.resize(c.n_flow_layers);
    const int stride = 2 + (c.act_norm ? 1 : 0);
    auto build_block = [&](int l) {
        BlockPack& b = f.blocks[l];
        const int idx0 = 1 + l * stride;
        std::string p = "transforms." + std::to_string(idx0);
        b.has_attn = !c.global_context;
        b.has_cif = cif;
        int h_pad = 0, ldp = 0;
        if (cif) {
            build_cif(f, wt, p, b.cif);
            h_pad = std::max({h_pad, max_hidden_pad(b.cif.dist), max_hidden_pad(b.cif.aff)});
            p += ".flow";                              // the conditioned coupling lives one level down (cif_block.py:65)
        }
        if (b.has_attn) {
            build_attn(f, wt, p + ".pre_conditioner.attn", b.attn, kv_rows, aug_slots + l);
            const std::string pp = p + ".pre_conditioner.pre_attention_mlp";
            pack_mlp_mid(f.arena, wt, pp, b.pre);
            b.pre.in_layer = build_plain(f, wt, pp + ".in_layer", d.d1, d.d1_pad);
            b.pre.out_layer = build_plain(f, wt, pp + ".out_layer", b.pre.sizes.back(), round_up(b.pre.sizes.back(), 32));
            if (b.pre.out_layer.N_pad != d.A_in_pad) throw Error(FC_ERR_SHAPE, "pre_attention_mlp output width != attn_input_dim");
            build_lnq(f, wt, pp + ".out_layer", b.attn);
            h_pad = std::max({h_pad, max_hidden_pad(b.pre), d.A_in_pad});
        }
        const std::string pn = p + ".transform.nn";
        pack_mlp_mid(f.arena, wt, pn, b.net);
        b.net.in_layer = build_in_layer(f, wt, pn, d.d1, d.d1_pad, b.has_attn ? &b.attn : nullptr);
        build_out_layer(f, wt, pn, b.net);
        attach_mlp_rows_images(f.arena, b.net);
        h_pad = std::max(h_pad, max_hidden_pad(b.net));
        if (c.flow_type != FC_FLOW_AFFINE) ldp = b.net.out_layer.N_pad;
        if (c.flow_type == FC_FLOW_EXPONENTIAL) {
            if (d.d2 > 16)       // the permanent cap of include/fcflow.h (enum fc_flow_type): refused at create, not at the first forward
                throw Error(FC_ERR_UNSUPPORTED, "ExponentialCoupling: latent_dim - latent_dim/2 > 16 is not supported (the layer emits d2^2 numbers per point)");
            const std::string pt = p + ".transform";
            std::vector<float> sc = {wt.get(pt + ".scale", {1}).data[0], wt.get(pt + ".shift", {1}).data[0],
                                     wt.get(pt + ".rescale", {1}).data[0], wt.get(pt + ".reshift", {1}).data[0]};
            b.expm_scal = f.arena.upload(sc);
        }
        if (l != c.n_flow_layers - 1) build_lin(f, wt, c.act_norm ? idx0 + 1 : -1, idx0 + stride - 1, b);
        std::lock_guard<std::mutex> lock(dims_mu);
        d.H_pad = std::max(d.H_pad, h_pad);
        d.ldp = std::max(d.ldp, ldp);
    };
    build_block(0);
    {
        int dev = 0;
        FC_HIP(hipGetDevice(&dev));
        const int n_workers = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, c.n_flow_layers - 1}));
        std::atomic<int> next{1};
        std::exception_ptr first_error;
        std::mutex err_mu;
        auto worker = [&]() {
            try {
                if (hipSetDevice(dev) != hipSuccess) throw Error(FC_ERR_HIP, "hipSetDevice failed in a packing thread");
                for (int l = next.fetch_add(1); l < c.n_flow_layers; l = next.fetch_add(1)) build_block(l);
            } catch (...) {
                std::lock_guard<std::mutex> lock(err_mu);
                if (!first_error) first_error = std::current_exception();
                next.store(c.n_flow_layers);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < n_workers; ++t) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
        if (first_error) std::rethrow_exception(first_error);
    }
    for (const BlockPack& b : f.blocks) f.log_const += b.cif.log_const + b.log_const;      // fixed order: reproducible
    // ---- one stacked K|V projection for every attention
    f.n_attn = (int)kv_rows.size();
    if (f.n_attn) {
        MatD all(f.n_attn * 2 * d.I_pad, d.E);
        for (int a = 0; a < f.n_attn; ++a) std::copy(kv_rows[a].v.begin(), kv_rows[a].v.end(), all.v.begin() + (size_t)a * 2 * d.I_pad * d.E);
        f.kv_all = pack_linear(f.arena, all, {}, {}, map_prefix(all.rows, all.rows), map_prefix(d.E, d.E_pad), {d.E_pad});
    }
}

// ---------------------------------------------------------------- workspace plan
struct FlowWs {
    float *xa, *xb, *h[3], *q, *a, *ctxp, *kv, *xin, *rowscal, *spl, *cbuf;
    void* kv16;      // K / V limb images of the layer in flight (split-fp16 attention)
    float* ldjp;     // log-det partial slots of the fused spline / pair epilogues, [ldj_slots][P_pad] (ldj_slot_count)
    int ldj_slots;
    bool kv_limbs;         // w.kv holds the K|V projections as the GEMM's fp16 limb image (GemmEpi::C16) instead of fp32
    float* lnss;           // [A_in / 64][P_pad] per-row sums of squares of the centred pre-MLP output (LayerNorm -> q fold)
    unsigned short* h16;   // fp16 limb image of the last hidden activation feeding the spline parameter GEMM (limb chain)
    int P, P_pad, Pc, Pc_pad, ldkv;
};
// Log-det partial slots (rows of FlowWs::ldjp): one per 128-column tile of the fused spline epilogue, two (one per wave column)
// per 128-column tile of a pair-packed epilogue (affine coupling, augmenter, CIF slice) on the 8-wave split-fp16 tile.  The
// epilogues ACCUMULATE into their own slots over the layers; flow_forward zeroes the buffer first and reduces it once at the end,
// in a fixed order (bit-reproducible, unlike atomics on log-prob).
static int ldj_slot_count(const fc_flow& f) {
    int n = f.cfg.flow_type == FC_FLOW_SPLINE ? f.d.ldp / 128 : 0;
    auto pair = [&](const PackedLinear& L) { if (L.N_pad > 0) n = std::max(n, 2 * (round_up(L.N_pad, 128) / 128)); };
    if (f.has_augment) pair(f.aug_net.out_layer);
    for (const BlockPack& b : f.blocks) {
        if (f.cfg.flow_type == FC_FLOW_AFFINE) pair(b.net.out_layer);
        if (b.has_cif) { pair(b.cif.dist.out_layer); pair(b.cif.aff.out_layer); }
    }
    return n;
}

static FlowWs plan_ws(const fc_flow& f, int B, int N, int M, void* ws, size_t bytes, bool dry, size_t* need) {
    const Dims& d = f.d;
    FlowWs w{};
    w.P = B * N; w.P_pad = round_up(w.P, ROW_PAD);
    w.Pc = B * M; w.Pc_pad = round_up(w.Pc, ROW_PAD);
    w.ldkv = f.n_attn * 2 * d.I_pad;
    WsCarver c(ws, bytes, dry);
    w.xa = c.floats((size_t)w.P_pad * d.ldx);
    w.xb = c.floats((size_t)w.P_pad * d.ldx);
    for (int i = 0; i < 3; ++i) w.h[i] = c.floats((size_t)w.P_pad * std::max(d.H_pad, 32));
    w.q = c.floats((size_t)w.P_pad * std::max(d.I_pad, 32));
    w.a = c.floats((size_t)w.P_pad * std::max(d.I_pad, 32));
    w.ctxp = c.floats((size_t)w.Pc_pad * d.E_pad);
    w.kv = c.floats((size_t)w.Pc_pad * std::max(w.ldkv, 32));
    w.xin = c.floats((size_t)w.P_pad * 32);
    w.rowscal = c.floats((size_t)w.P_pad);
    w.spl = c.floats(d.ldp ? (size_t)w.P_pad * d.ldp : 1);
    w.cbuf = c.floats(d.nz > 0 ? (size_t)w.P_pad * d.nz_pad : 1);
    w.ldj_slots = ldj_slot_count(f);
    w.ldjp = c.floats(std::max<size_t>((size_t)w.ldj_slots * w.P_pad, 1));
    w.h16 = (unsigned short*)c.bytes((size_t)w.P_pad * std::max(d.H_pad, 32) * 4);
    w.lnss = c.floats(f.n_attn > 0 ? (size_t)(std::max(d.A_in, 64) / 64) * w.P_pad : 1);
    w.kv16 = c.bytes(f.n_attn > 0 ? std::max<size_t>(attention_limb_ws_bytes(w.Pc_pad, d.I_pad), 16) : 16);
    if (need) *need = c.off + 256;
    return w;
}

thread_local float* t_flow_trace = nullptr;
thread_local size_t t_flow_trace_floats = 0;
void flow_set_trace(float* buf, size_t floats) { t_flow_trace = buf; t_flow_trace_floats = floats; }

int g_premlp_chain = 0;      // knob 19: limb chain through the pre-attention MLP into the LayerNorm -> q GEMM (K = 256: 8 k-tiles per
                             // output tile, the tile-boundary cost of the DMA loop outweighs its main loop: measured 1 % slower end to end)
static int run_mlp_hidden(const fc_flow& f, const PackedMLP& m, const ASeg* in_segs, const float* rowscal, FlowWs& w, int act, hipStream_t s,
                          unsigned short* last_limbs = nullptr, float last_scale = 0.f, int n_scene = 0) {
    // round 4: hidden layers of a 512-wide coupling net on the 256 x 256 one-accumulator kernel (spline_wide.hip EPI 1).  The gate is the SCENE's
    // size (target points per scene), never the batch's: a scene's log-probs must not depend on the batch it sits in, and this arithmetic
    // (one accumulator, k32 MFMAs) is not the per-layer 128 x 128 / 64 x 64 loops' or the row-resident chain's.
    const int wk = gemm_linear_wide_knob();
    if (last_limbs && f.d.H_pad == 512 && gemm_limb_chain_all_ok() && act == FC_ACT_GELU && (wk == 2 || (wk == 1 && n_scene >= 2048)) && w.P_pad % 256 == 0 && !m.mid.empty()) {
        bool ok = true;
        for (const PackedLinear& L : m.mid) ok = ok && L.W1 && !L.w1_permuted && L.N_pad % 256 == 0 && L.K_pad % 64 == 0;
        if (ok) return run_mlp_hidden_generic(m, in_segs, rowscal, act, w.h, std::max(f.d.H_pad, 32), w.P_pad, s, w.P, last_limbs, last_scale, true);
    }
    // 512-wide coupling nets inside a guard scope: in_layer + hidden layers as ONE row-resident launch (mlprows.hip); the scratch images
    // of its intermediate activations live in the h[] buffers (same 2 KB per row as a 512-wide fp32 panel)
    // (a workgroup owns 128 rows for the whole chain: with fewer workgroups than ~3/4 of the CUs -- C1's 2 x 1024 points are 16 -- the chain
    // of ONE workgroup is the launch's duration and the per-layer launches on 64 x 64 tiles are faster: 21 vs 38 ms per C1 step)
    if (last_limbs && f.d.H_pad == 512 && gemm_limb_chain_all_ok() && mlp_rows_eligible(m.in_layer, m.mid, act) && mlp_rows_fills_the_chip(w.P_pad)) {
        launch_mlp_rows(m.in_layer, m.mid, in_segs, rowscal, act, w.h, last_limbs, w.P_pad, w.P, s, last_scale);
        return -1;
    }
    return run_mlp_hidden_generic(m, in_segs, rowscal, act, w.h, std::max(f.d.H_pad, 32), w.P_pad, s, w.P, last_limbs, last_scale);
}

// pre-conditioner: pre-MLP -> LayerNorm -> q -> attention; result in w.a  (models/cif_block.py:14-20 / augmenter.py:15-16)
// true when the pre-conditioner (pre, at) reading a latent of pitch ldx runs on the row-resident kernel AND can take the previous layer's folded
// ActNorm + permuter `lu` as its pre-layer (premlp.hip): decided once per layer pair by flow_forward, which then skips that layer's GEMM launch
static bool attention_takes_lu(const fc_flow& f, const PackedMLP& pre, const AttnPack& at, const PackedLinear& lu, const FlowWs& w, int act) {
    const Dims& d = f.d;
    const int ldh = std::max(d.H_pad, 32);
    return premlp_fusable(pre.in_layer, pre.mid, pre.out_layer, at.q) && d.ldx >= pre.in_layer.K_pad &&
           premlp_rows_ok(w.P_pad, d.I_pad, w.q, w.h[0], (size_t)w.P_pad * ldh) && premlp_lu_fusable(lu, pre.in_layer, act, d.ldx);
}

static void run_attention(const fc_flow& f, const PackedMLP& pre, const AttnPack& at, const ASeg& in, FlowWs& w, int act, int B, int N, int M,
                          hipStream_t s, const PackedLinear* lu = nullptr, const float* xprev = nullptr) {
    const Dims& d = f.d;
    const int ldh = std::max(d.H_pad, 32);
    if (premlp_fusable(pre.in_layer, pre.mid, pre.out_layer, at.q) && in.lda >= pre.in_layer.K_pad &&
        (kDevVariants || premlp_rows_ok(w.P_pad, d.I_pad, w.q, w.h[0], (size_t)w.P_pad * ldh))) {
        // the whole chain x1 -> MLP -> LayerNorm -> q in one kernel: the 64-row activation tile stays in LDS (premlp.hip); with `lu` the
        // previous layer's ActNorm + LU runs in front of it and writes this layer's latent (in.ptr) from xprev
        launch_premlp(in.ptr, in.lda, pre.in_layer, pre.mid, pre.out_layer, at.q, act, w.q, d.I_pad, w.P_pad, w.P, s, w.h[0], (size_t)w.P_pad * ldh, lu, xprev);
    } else {
        if (lu) throw Error(FC_ERR_INVALID, "run_attention: a pending ActNorm + LU pre-layer needs the row-resident pre-attention kernel");
        // limb chain through the pre-attention MLP into the LayerNorm -> q GEMM (every hidden activation as a limb image, DMA loops)
        const PackedLinear& pre_last = pre.mid.empty() ? pre.in_layer : pre.mid.back();
        const bool chain = g_premlp_chain && at.has_lnq && gemm_lnq_ok() && gemm_limb_chain_all_ok() && w.h16 && !pre.mid.empty() && at.lnq.W2 != nullptr &&
                           pre_last.W2 != nullptr && pre_last.N_pad == at.lnq.K_pad && pre_last.N_pad % 128 == 0 && at.lnq.nseg == 1;
        const int cur = run_mlp_hidden(f, pre, &in, nullptr, w, act, s, chain ? w.h16 : nullptr);
        if (at.has_lnq && gemm_lnq_ok()) {
            // out_layer, LayerNorm and the q projection as ONE GEMM (AttnPack::lnq) + a 16 MB finalize pass
            GemmEpi e{};
            e.C = w.q; e.ldc = d.I_pad; e.d2 = d.A_in; e.ldj_part = w.lnss; e.ldj_pitch = (size_t)w.P_pad; e.rows_valid = w.P;
            if (chain) e.A16 = w.h16;
            ASeg a{chain ? w.h[0] : w.h[cur], ldh};
            launch_gemm(at.lnq, &a, w.P_pad, e, EPI_LNQ, s);
            if (w.kv_limbs) {
                // the attention kernel applies rstd and the bias while it loads its queries
                const AttnLnq lq{w.lnss, d.A_in / 64, (size_t)w.P_pad, 1.0f / (float)d.A_in, at.q_bias};
                launch_attention_c16(w.q, d.I_pad, reinterpret_cast<const unsigned short*>(w.kv), w.ldkv, at.kv_col, w.a, d.I_pad, B, N, N, M, M,
                                     d.I_pad, s, &lq);
            } else {
                launch_lnq_finalize(w.q, d.I_pad, w.lnss, d.A_in / 64, (size_t)w.P_pad, d.A_in, at.q_bias, w.P, s);
                launch_attention(w.q, d.I_pad, w.kv + at.kv_col, w.ldkv, w.kv + at.kv_col + d.I_pad, w.ldkv, w.a, d.I_pad, B, N, N, M, M, d.I_pad,
                                 w.kv16, s);
            }
            return;
        }
        int o = 0;
        while (o == cur) ++o;
        GemmEpi e{};
        e.act = FC_ACT_NONE; e.C = w.h[o]; e.ldc = ldh; e.rows_valid = w.P;
        ASeg a{w.h[cur], ldh};
        launch_gemm(pre.out_layer, &a, w.P_pad, e, EPI_LINEAR, s);
        launch_layernorm(w.h[o], ldh, d.A_in, w.P, s);
        GemmEpi eq{};
        eq.act = FC_ACT_NONE; eq.C = w.q; eq.ldc = d.I_pad; eq.rows_valid = w.P;
        ASeg aq{w.h[o], ldh};
        launch_gemm(at.q, &aq, w.P_pad, eq, EPI_LINEAR, s);
    }
    if (w.kv_limbs) launch_attention_c16(w.q, d.I_pad, reinterpret_cast<const unsigned short*>(w.kv), w.ldkv, at.kv_col, w.a, d.I_pad, B, N, N, M, M, d.I_pad, s);
    else launch_attention(w.q, d.I_pad, w.kv + at.kv_col, w.ldkv, w.kv + at.kv_col + d.I_pad, w.ldkv, w.a, d.I_pad, B, N, N, M, M, d.I_pad, w.kv16, s);
}

// the conditioned coupling of one block (PreConditionApplier, models/transform.py:47-58), forward or inverse, in place on xc
static void run_coupling(fc_flow& f, const BlockPack& b, FlowWs& w, float* xc, const float* rowscal, float* logprob, bool inverse, int B, int N,
                         int M, hipStream_t s, const PackedLinear* lu = nullptr, const float* xprev = nullptr, int trace_layer = -1) {
    const Dims& d = f.d;
    const fc_flow_config& c = f.cfg;
    const int ldh = std::max(d.H_pad, 32);
    ASeg segs[2];
    segs[0] = {xc, d.ldx};
    if (b.has_attn) {
        // CIFblock builds its pre_attention_mlp with GELU regardless of the configured nonlinearity (cif_block.py:61)
        run_attention(f, b.pre, b.attn, segs[0], w, b.has_cif ? (int)FC_ACT_GELU : c.nonlinearity, B, N, M, s, lu, xprev);
        segs[1] = {w.a, d.I_pad};
    } else {
        if (lu) throw Error(FC_ERR_INVALID, "run_coupling: a pending ActNorm + LU pre-layer needs an attention pre-conditioner");
        segs[1] = {w.ctxp, d.E_pad};
    }
    if (trace_layer >= 0 && t_flow_trace) {
        // diagnostic trace (fc_debug_flow_trace): the x2 half exactly as this coupling will read it (behind a fused ActNorm + LU pre-layer)
        if ((size_t)(trace_layer + 1) * w.P * d.d2 > t_flow_trace_floats) throw Error(FC_ERR_INVALID, "fc_debug_flow_trace: buffer too small for n_flow_layers x rows x d2");
        FC_HIP(hipMemcpy2DAsync(t_flow_trace + (size_t)trace_layer * w.P * d.d2, (size_t)d.d2 * 4, xc + d.d1_pad, (size_t)d.ldx * 4, (size_t)d.d2 * 4, (size_t)w.P,
                                hipMemcpyDeviceToDevice, s));
    }
    // limb chain: the spline parameter GEMM spans 30 column tiles that would each re-split the same fp32 rows into fp16 limbs; the
    // layer before it writes its output once as the limb image instead (GemmEpi::C16) and the parameter GEMM copies it (A16)
    const bool fused_spline = c.flow_type == FC_FLOW_SPLINE && !inverse && gemm_split_enabled() && b.net.out_layer.W3 != nullptr;
    const PackedLinear& last_hidden = b.net.mid.empty() ? b.net.in_layer : b.net.mid.back();
    const bool chain = fused_spline && gemm_limb_chain_ok() && b.net.out_layer.W2 != nullptr && last_hidden.W2 != nullptr &&
                       last_hidden.N_pad == b.net.out_layer.K_pad && last_hidden.N_pad > 64 && b.net.out_layer.nseg == 1;
    // the same chain into the affine coupling's (s, t) layer: forward direction, split-fp16 scope, pair-packed epilogue on the DMA tile
    const bool chain_aff = c.flow_type == FC_FLOW_AFFINE && !inverse && gemm_limb_chain_all_ok() && b.net.out_layer.W2 != nullptr &&
                           last_hidden.W2 != nullptr && last_hidden.N_pad == b.net.out_layer.K_pad && last_hidden.N_pad % 128 == 0 &&
                           b.net.out_layer.nseg == 1 && !b.net.mid.empty();
    // round 4: the chain's last activation in the one-accumulator form (common.h kOneAccActScale) for the 256 x 256 fused spline kernel (spline_wide.hip)
    const bool wide = chain && gemm_spline_wide_on() && spline_wide_eligible(b.net.out_layer, c.num_bins_spline) && w.P_pad % 256 == 0;
    const int cur = run_mlp_hidden(f, b.net, segs, rowscal, w, c.nonlinearity, s, (chain || chain_aff) ? w.h16 : nullptr, wide ? kOneAccActScale : 0.f, N);
    ASeg a{(chain || chain_aff) ? w.h[0] : w.h[cur], ldh};
    if (c.flow_type == FC_FLOW_AFFINE) {
        GemmEpi e{};
        e.xbuf = xc; e.ldx = d.ldx; e.x2_col0 = d.d1_pad; e.d2 = d.d2; e.scale_fn = c.affine_scale_fn;
        e.logprob = logprob; e.rows_valid = w.P; e.inverse = inverse;
        if (!inverse) { e.ldj_part = w.ldjp; e.ldj_pitch = (size_t)w.P_pad; }
        if (chain_aff) e.A16 = w.h16;
        launch_gemm(b.net.out_layer, &a, w.P_pad, e, EPI_AFFINE, s);
    } else if (fused_spline) {
        // forward: the parameter GEMM evaluates the splines in its epilogue; only per-tile log-det partials leave the kernel
        GemmEpi e{};
        e.xbuf = xc; e.ldx = d.ldx; e.x2_col0 = d.d1_pad; e.d2 = d.d2; e.spline_K = c.num_bins_spline; e.rows_valid = w.P;
        e.ldj_part = w.ldjp; e.ldj_pitch = (size_t)w.P_pad;
        if (chain) { e.A16 = w.h16; e.a16_scale = wide ? kOneAccActScale : 0.f; }
        launch_gemm(b.net.out_layer, &a, w.P_pad, e, EPI_SPLINE, s);       // log-dets accumulate in w.ldjp; flow_forward reduces them once
    } else {
        GemmEpi e{};
        e.C = w.spl; e.ldc = d.ldp; e.rows_valid = w.P;
        launch_gemm(b.net.out_layer, &a, w.P_pad, e, EPI_LINEAR, s);
        if (c.flow_type == FC_FLOW_SPLINE)
            launch_spline(w.spl, d.ldp, xc, d.ldx, d.d1_pad, d.d2, c.num_bins_spline, logprob, w.P, inverse, s);
        else
            launch_expm_coupling(w.spl, d.ldp, xc, d.ldx, d.d1_pad, d.d2, b.expm_scal, logprob, w.P, inverse, s);
    }
}

// CIF: net(x) -> [mean | log_std] pairs with the given pair epilogue
static void run_cif_dist(fc_flow& f, const CifPack& cp, FlowWs& w, float* xc, GemmEpi e, int epi, hipStream_t s) {
    ASeg in{xc, f.d.ldx};
    const int cur = run_mlp_hidden(f, cp.dist, &in, nullptr, w, FC_ACT_GELU, s);
    ASeg a{w.h[cur], std::max(f.d.H_pad, 32)};
    e.clamp = f.cfg.clamp_dist; e.d2 = f.d.nz; e.rows_valid = w.P;
    if (!e.inverse) { e.ldj_part = w.ldjp; e.ldj_pitch = (size_t)w.P_pad; }
    launch_gemm(cp.dist.out_layer, &a, w.P_pad, e, epi, s);
}
static void run_cif_affine(fc_flow& f, const CifPack& cp, FlowWs& w, float* xc, float* logprob, bool inverse, hipStream_t s) {
    const Dims& d = f.d;
    ASeg in{w.cbuf, d.nz_pad};
    const int cur = run_mlp_hidden(f, cp.aff, &in, nullptr, w, FC_ACT_GELU, s);
    ASeg a{w.h[cur], std::max(d.H_pad, 32)};
    GemmEpi e{};
    e.xbuf = xc; e.ldx = d.ldx; e.x2_col0 = 0; e.split = d.d1; e.split_pad = d.d1_pad; e.d2 = d.D; e.scale_fn = FC_SCALE_SIGMOID;
    e.post_scale = cp.post_scale; e.logprob = logprob; e.rows_valid = w.P; e.inverse = inverse;
    if (!inverse) { e.ldj_part = w.ldjp; e.ldj_pitch = (size_t)w.P_pad; }
    launch_gemm(cp.aff.out_layer, &a, w.P_pad, e, EPI_AFFINE, s);
}

struct Prep {
    FlowWs w;
    const float* rowscal = nullptr;
};
static Prep prepare(fc_flow& f, const float* ctx, const float* extra, int B, int N, int M, void* ws, size_t ws_bytes, hipStream_t s) {
    const Dims& d = f.d;
    const fc_flow_config& c = f.cfg;
    if (B < 1 || N < 1 || M < 1) throw Error(FC_ERR_INVALID, "B, N, M must be positive");
    if (!ctx) throw Error(FC_ERR_INVALID, "null ctx");
    if (d.X && !extra) throw Error(FC_ERR_INVALID, "this flow was built with extra context: extra must not be NULL");
    if (c.global_context && M != N) throw Error(FC_ERR_INVALID, "global context is per target point: ctx must be [B,N,E] (M == N)");
    Prep p;
    p.w = plan_ws(f, B, N, M, ws, ws_bytes, false, nullptr);
    FlowWs& w = p.w;
    launch_pack_rows(ctx, d.E, d.E, w.ctxp, d.E_pad, 0, d.E_pad, w.Pc, s);
    // pad rows of the context panel: the K|V projection stages them like any row, and stale workspace bytes there (a NaN, a value beyond fp16's
    // range) would raise the split-fp16 range flag -- a needless repeat of the whole pass on the bf16 limbs, and a scene whose log-probs then
    // depend on what ran in the workspace before (found in round 4: one scene of 200 context points behind a three-scene batch)
    if (w.Pc_pad > w.Pc) launch_fill(w.ctxp + (size_t)w.Pc * d.E_pad, 0.f, (size_t)(w.Pc_pad - w.Pc) * d.E_pad, s);
    if (d.X) { launch_repeat_extra(extra, d.X, w.rowscal, B, N, s); p.rowscal = w.rowscal; }
    if (f.n_attn) {
        // inside a guard scope the stacked K|V projection writes its output straight as the limb image the split-fp16 attention
        // stages (same bytes, same buffer): no fp32 K/V, no per-layer conversion pass
        w.kv_limbs = gemm_limb_chain_ok() && attention_fp16_enabled() && d.I_pad <= 64 && f.kv_all.W2 != nullptr && w.ldkv % 128 == 0 && w.ldkv == f.kv_all.N_pad;
        GemmEpi e{};
        e.rows_valid = w.Pc;
        if (w.kv_limbs) { e.C16 = reinterpret_cast<unsigned short*>(w.kv); e.c16_scale = kOneAccActScale; }      // (the one-accumulator image the attention kernel multiplies)
        else { e.C = w.kv; e.ldc = w.ldkv; }
        ASeg a{w.ctxp, d.E_pad};
        launch_gemm(f.kv_all, &a, w.Pc_pad, e, EPI_LINEAR, s);
    }
    if (d.nz > 0) launch_fill(w.cbuf, 0.f, (size_t)w.P_pad * d.nz_pad, s);
    return p;
}

// Diagnostic (fc_debug_flow_trace, ops_api.cpp; tests/fullsize_util.py): when the calling thread has set a buffer, flow_forward copies the
// x2 half of the latent AS THE COUPLING OF LAYER l WILL READ IT into trace[l][row][d2] -- the fp32 values the spline's inside / outside
// decision |x2| <= 3 is taken on (models/spline_coupling.py:35-48), so a test can hand the fp64 oracle the HIP run's own decisions.

static int expected_noise(const fc_flow& f) { return (f.has_augment ? 1 : 0) + (f.d.nz > 0 ? f.cfg.n_flow_layers : 0); }

static void flow_forward(fc_flow& f, const float* x, const float* ctx, const float* extra, const float* const* eps, int n_eps,
                         float* logprob, float* z_out, int B, int N, int M, void* ws, size_t ws_bytes, hipStream_t s) {
    const Dims& d = f.d;
    const fc_flow_config& c = f.cfg;
    if (!x || !logprob) throw Error(FC_ERR_INVALID, "null x / logprob");
    if (n_eps != expected_noise(f)) throw Error(FC_ERR_INVALID, "wrong number of noise tensors");
    for (int i = 0; i < n_eps; ++i) if (!eps || !eps[i]) throw Error(FC_ERR_INVALID, "null noise tensor");
    Prep pr = prepare(f, ctx, extra, B, N, M, ws, ws_bytes, s);
    FlowWs& w = pr.w;
    const int ldh = std::max(d.H_pad, 32);
    int eps_i = 0;

    launch_fill(logprob, 0.f, (size_t)w.P, s);
    const int ldj_tiles = w.ldj_slots;                                         // epilogues accumulate their log-det partials here
    if (ldj_tiles) launch_fill(w.ldjp, 0.f, (size_t)ldj_tiles * w.P_pad, s);
    float* xc = w.xa;
    float* xn = w.xb;
    launch_fill(xc, 0.f, (size_t)w.P_pad * d.ldx, s);
    if (f.has_augment) {
        launch_pack_rows(x, d.Din, d.Din, w.xin, 32, 0, 32, w.P, s);
        if (w.P_pad > w.P) launch_fill(w.xin + (size_t)w.P * 32, 0.f, (size_t)(w.P_pad - w.P) * 32, s);      // (pad rows: zeros, not stale workspace bytes)
        const int n1 = std::min(d.Din, d.d1);                                       // latent[0:Din] = x, split over the x1 | x2 regions
        launch_pack_rows(x, d.Din, n1, xc, d.ldx, 0, n1, w.P, s);
        if (d.Din > n1) launch_pack_rows(x + n1, d.Din, d.Din - n1, xc, d.ldx, d.d1_pad, d.Din - n1, w.P, s);
        ASeg in{w.xin, 32};
        run_attention(f, f.aug_pre, f.aug_attn, in, w, c.nonlinearity, B, N, M, s);
        ASeg segs[2] = {{w.xin, 32}, {w.a, d.I_pad}};
        const int cur = run_mlp_hidden(f, f.aug_net, segs, pr.rowscal, w, c.nonlinearity, s);
        GemmEpi e{};
        e.xbuf = xc; e.ldx = d.ldx; e.d2 = d.D - d.Din; e.logprob = logprob; e.eps = eps[eps_i++];
        e.d_in = d.Din; e.d1 = d.d1; e.d1_pad = d.d1_pad; e.rows_valid = w.P;
        e.ldj_part = w.ldjp; e.ldj_pitch = (size_t)w.P_pad;
        ASeg a{w.h[cur], ldh};
        launch_gemm(f.aug_net.out_layer, &a, w.P_pad, e, EPI_AUGMENT, s);
    } else {
        launch_pack_rows(x, d.D, d.d1, xc, d.ldx, 0, d.d1, w.P, s);
        launch_pack_rows(x + d.d1, d.D, d.d2, xc, d.ldx, d.d1_pad, d.d2, w.P, s);
    }
    const PackedLinear* pend = nullptr;
    for (int l = 0; l < c.n_flow_layers; ++l) {
        BlockPack& b = f.blocks[l];
        if (b.has_cif) {
            GemmEpi ea{};                                   // Augment: z2 -> cbuf (natural order), ldj -= log N(z2)
            ea.xbuf = w.cbuf; ea.ldx = d.nz_pad; ea.d_in = 0; ea.d1 = d.nz; ea.d1_pad = d.nz_pad; ea.logprob = logprob; ea.eps = eps[eps_i++];
            run_cif_dist(f, b.cif, w, xc, ea, EPI_AUGMENT, s);
            run_cif_affine(f, b.cif, w, xc, logprob, false, s);
            GemmEpi es{};                                   // Slice: ldj += log N(actnorm(z2); mu(zx), sigma(zx))
            es.val = w.cbuf; es.ldval = d.nz_pad; es.val_shift = b.cif.z2_shift; es.val_scale = b.cif.z2_scale; es.logprob = logprob;
            run_cif_dist(f, b.cif, w, xc, es, EPI_SLICE, s);
        }
        // `pend`: the previous layer's folded ActNorm + permuter, not launched yet -- it runs as the pre-layer of this layer's row-resident
        // pre-attention kernel, reading xc and writing xn, which becomes this layer's latent
        if (pend) std::swap(xc, xn);
        run_coupling(f, b, w, xc, pr.rowscal, logprob, false, B, N, M, s, pend, pend ? xn : nullptr, l);
        pend = nullptr;
        if (b.has_lin) {
            const bool next_takes_it = l + 1 < c.n_flow_layers && f.blocks[l + 1].has_attn && !f.blocks[l + 1].has_cif &&
                                       attention_takes_lu(f, f.blocks[l + 1].pre, f.blocks[l + 1].attn, b.lin, w, c.nonlinearity);
            if (next_takes_it) pend = &b.lin;
            else {
                GemmEpi e{};
                e.C = xn; e.ldc = d.ldx; e.rows_valid = w.P;
                ASeg ax{xc, d.ldx};
                launch_gemm(b.lin, &ax, w.P_pad, e, EPI_LINEAR, s);
                std::swap(xc, xn);
            }
        }
    }
    if (pend) throw Error(FC_ERR_INVALID, "flow_forward: an ActNorm + LU pre-layer was left pending");
    if (ldj_tiles) launch_ldj_reduce(w.ldjp, ldj_tiles, (size_t)w.P_pad, logprob, w.P, s);
    launch_base_density(xc, d.ldx, d.d1, d.d1_pad, d.d2, logprob, (float)f.log_const, z_out, d.D, w.P, s);
}

// Flow.sample's inverse pass (models/transform.py:79-84): transforms in reverse order, each inverted.
static void flow_inverse(fc_flow& f, const float* z, const float* ctx, const float* extra, const float* const* eps, int n_eps, float* x_out,
                         int B, int N, int M, void* ws, size_t ws_bytes, hipStream_t s) {
    const Dims& d = f.d;
    const fc_flow_config& c = f.cfg;
    if (!z || !x_out) throw Error(FC_ERR_INVALID, "null z / x_out");
    const int need_eps = d.nz > 0 ? c.n_flow_layers : 0;     // Slice.inverse draws once per CIF block (models/slice.py:46-58)
    if (n_eps != need_eps) throw Error(FC_ERR_INVALID, "wrong number of noise tensors for the inverse pass");
    for (int i = 0; i < n_eps; ++i) if (!eps || !eps[i]) throw Error(FC_ERR_INVALID, "null noise tensor");
    for (auto& b : f.blocks)                                 // lazily invert the folded ActNorm+permuter matrices (double precision)
        if (b.has_lin && !b.has_lin_inv) {
            MatD inv = inverse_double(b.lin_w);
            VecD bi(d.D, 0.0);
            for (int i = 0; i < d.D; ++i) { double t = 0; for (int k = 0; k < d.D; ++k) t += inv.at(i, k) * b.lin_b[k]; bi[i] = -t; }
            const std::vector<int> xl = map_xlayout(d.d1, d.d1_pad, d.d2, d.d2_pad);
            b.lin_inv = pack_linear(f.arena, inv, bi, {}, xl, xl, {d.ldx});
            b.has_lin_inv = true;
        }
    Prep pr = prepare(f, ctx, extra, B, N, M, ws, ws_bytes, s);
    FlowWs& w = pr.w;
    float* xc = w.xa;
    float* xn = w.xb;
    launch_fill(xc, 0.f, (size_t)w.P_pad * d.ldx, s);
    launch_pack_rows(z, d.D, d.d1, xc, d.ldx, 0, d.d1, w.P, s);
    launch_pack_rows(z + d.d1, d.D, d.d2, xc, d.ldx, d.d1_pad, d.d2, w.P, s);
    int eps_i = 0;
    for (int l = c.n_flow_layers - 1; l >= 0; --l) {
        BlockPack& b = f.blocks[l];
        if (b.has_lin) {
            GemmEpi e{};
            e.C = xn; e.ldc = d.ldx; e.rows_valid = w.P;
            ASeg ax{xc, d.ldx};
            launch_gemm(b.lin_inv, &ax, w.P_pad, e, EPI_LINEAR, s);
            std::swap(xc, xn);
        }
        run_coupling(f, b, w, xc, pr.rowscal, nullptr, true, B, N, M, s);
        if (b.has_cif) {
            GemmEpi ea{};                                   // Slice.inverse: x2n ~ N(mu(z), sigma(z)); stored as z2 = x2n / g2 + shift2
            ea.xbuf = w.cbuf; ea.ldx = d.nz_pad; ea.d_in = 0; ea.d1 = d.nz; ea.d1_pad = d.nz_pad; ea.eps = eps[eps_i++]; ea.inverse = 1;
            ea.val_shift = b.cif.z2_shift; ea.val_scale = b.cif.z2_scale;
            run_cif_dist(f, b.cif, w, xc, ea, EPI_AUGMENT, s);
            run_cif_affine(f, b.cif, w, xc, nullptr, true, s);
        }
    }
    // Augment.inverse keeps the first input_dim latent dims (models/augmenter.py:65-67)
    const int n1 = std::min(d.Din, d.d1);
    launch_pack_rows(xc, d.ldx, n1, x_out, d.Din, 0, n1, w.P, s);
    if (d.Din > n1) launch_pack_rows(xc + d.d1_pad, d.ldx, d.Din - n1, x_out, d.Din, n1, d.Din - n1, w.P, s);
}

}  // namespace fc

// ================================================================== C ABI
namespace fc { const char* get_last_error(); void prof_set(bool); void prof_reset(); std::string prof_report_json(); }


extern "C" {

int fc_abi_version(void) { return FC_ABI_VERSION; }
const char* fc_last_error(void) { return fc::get_last_error(); }

int fc_profile_enable(int32_t on) { fc::prof_set(on != 0); return FC_OK; }
int fc_profile_reset(void) { fc::prof_reset(); return FC_OK; }
int fc_profile_filter(const char* kernel_substr) { fc::prof_filter(kernel_substr); return FC_OK; }
int fc_profile_stride(int32_t n) { fc::prof_stride(n); return FC_OK; }
int fc_profile_report(char* buf, size_t cap) {
    FC_API_BEGIN
    const std::string r = fc::prof_report_json();
    if (!buf || cap < r.size() + 1) throw fc::Error(FC_ERR_INVALID, "fc_profile_report: buffer too small");
    memcpy(buf, r.c_str(), r.size() + 1);
    FC_API_END
}

int fc_flow_create(const fc_flow_config* cfg, const fc_tensor* tensors, int32_t n_tensors, fc_flow** out) {
    FC_API_BEGIN
    if (!cfg || !out) throw fc::Error(FC_ERR_INVALID, "fc_flow_create: null argument");
    *out = nullptr;
    std::unique_ptr<fc_flow> f(new fc_flow());
    f->cfg = *cfg;
    fc::WeightTable wt(tensors, n_tensors);
    fc::build_flow(*f, wt);
    f->fp16_flag = (int*)f->arena.alloc_floats(1);
    FC_HIP(hipDeviceSynchronize());
    *out = f.release();
    FC_API_END
}

void fc_flow_destroy(fc_flow* flow) { delete flow; }

int fc_flow_workspace_bytes(const fc_flow* flow, int32_t B, int32_t N, int32_t M, size_t* bytes) {
    FC_API_BEGIN
    if (!flow || !bytes || B < 1 || N < 1 || M < 1) throw fc::Error(FC_ERR_INVALID, "fc_flow_workspace_bytes: bad argument");
    fc::plan_ws(*flow, B, N, M, nullptr, 0, true, bytes);
    FC_API_END
}

int fc_flow_noise_count(const fc_flow* flow) { return flow ? fc::expected_noise(*flow) : 0; }
int fc_flow_noise_width(const fc_flow* flow, int32_t i) {
    if (!flow || i < 0 || i >= fc::expected_noise(*flow)) return 0;
    if (flow->has_augment && i == 0) return flow->d.D - flow->d.Din;
    return flow->d.nz;
}

int fc_flow_logprob_f32(fc_flow* flow, const float* x, const float* ctx, const float* extra, const float* const* eps, int32_t n_eps,
                        float* logprob, float* z_out, int32_t B, int32_t N, int32_t M, void* workspace, size_t workspace_bytes, void* stream) {
    FC_API_BEGIN
    if (!flow || !workspace) throw fc::Error(FC_ERR_INVALID, "fc_flow_logprob_f32: null flow / workspace");
    // fast split-fp16 GEMMs first; the whole pass is repeated with the bf16-limb GEMMs if an activation left fp16's range
    // (deferred range check, fc_range_check_defer: the pass may be repeated after this call has returned -- it owns its arguments)
    const std::vector<const float*> eps_own(eps, eps + (eps && n_eps > 0 ? n_eps : 0));
    fc::run_fp16_guarded(flow->fp16_flag, (hipStream_t)stream, [=] {
        fc::flow_forward(*flow, x, ctx, extra, eps_own.data(), n_eps, logprob, z_out, B, N, M, workspace, workspace_bytes, (hipStream_t)stream);
    }, true);
    FC_API_END
}

int fc_flow_inverse_f32(fc_flow* flow, const float* z, const float* ctx, const float* extra, const float* const* eps, int32_t n_eps, float* x_out,
                        int32_t B, int32_t N, int32_t M, void* workspace, size_t workspace_bytes, void* stream) {
    FC_API_BEGIN
    if (!flow || !workspace) throw fc::Error(FC_ERR_INVALID, "fc_flow_inverse_f32: null flow / workspace");
    fc::flow_inverse(*flow, z, ctx, extra, eps, n_eps, x_out, B, N, M, workspace, workspace_bytes, (hipStream_t)stream);
    FC_API_END
}

}  // extern "C"
