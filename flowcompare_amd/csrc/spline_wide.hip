// The fused spline parameter layer, round 4: a PERSISTENT 256 x 256 workgroup tile on v_mfma_f32_16x16x32_f16 with ONE accumulator per output.
//
//   params[rows, 3750] = h[rows, 512] W^T + b ;  y2, log-det = RQ-spline(x2; params)      (models/spline_coupling.py:187-210, models/nets.py:19-30)
//
// Why another kernel (profiles/micro/wide_gemm_probe.hip measured the structure first): the shipped VAR 11 loop
// (gemm.hip: 128 x 128 tile, 32x32x16 MFMAs, main + cross-product accumulator sets) sat at 48 % matrix-pipe busy at 1.94 GHz for two rounds
// whatever the loop order.  Three things change here, each measured in the probe:
//   * ONE accumulator set.  The low limb is stored UNSCALED, lo = rn16(x - hi) (fp16 subnormals are honoured by the MFMA at full rate), so the
//     three limb products hi.hi + lo.hi + hi.lo land at their true scale in one fp32 accumulator.  Operands are pre-scaled by exact powers of
//     two so that lo stays a normal fp16 number for every value that matters: activations by kOneAccActScale (their producer's epilogue
//     writes the image that way), weights per layer so that max |w| lands in [2^14, 2^15) (spline_wide_attach); the product is scaled back
//     inside the spline's own first fma.  Error against fp64 at K = 512: 6.9e-8 mean / 7.1e-7 max on unit-scale outputs, below the fp32 fmaf
//     chain (1.05e-7 / 1.24e-6) at all three weight scales of the probe (tests/test_gpu_ops.py::test_one_accumulator_limb_form...).
//   * Half the accumulator registers pay for a 64-point x 128-parameter WAVE tile and a 256 x 256 workgroup tile: half the LDS-DMA bytes and
//     0.6 of the LDS fragment reads per MFMA of VAR 11.
//   * The 16x16x32 MFMA shape: the chip holds a higher clock on it (MI355X_MICROARCH.md, DVFS give-back item 7): main loop alone 0.50 ms
//     against 0.60-0.62 ms for the same tile on 32x32x16, same box.
// Structure (the guide's 8-phase GEMM template, cdna_hip_programming.md section 5): eight waves = 4 point blocks of 64 x 2 parameter tiles of
// 128; waves w and w + 4 share a SIMD and belong to different GROUPS; a k32 step is four phases {LOAD: fragment reads + DMA issue | barrier |
// 24 MFMAs | barrier} and group 1 runs ONE barrier behind group 0, so that on every SIMD one wave multiplies while its partner loads.
// Two 64 KB LDS stages (k32 of 256 point rows + 256 weight rows), LDS-DMA pieces of 8 rows x 128 B with the XOR swizzle on the source
// address, one continuous stream across the workgroup's tiles, bias (pre-scaled) and the x2 / log-det operands of the next tile fetched
// during the last k step.
// Epilogue in registers: with the weights as the MFMA's A operand a lane of row kq = lane >> 4 holds, for the point lane & 15 of each of
// the wave's four 16-point blocks, the 32 parameters "slot s" = 4 (16-parameter block) + register.  The layer's columns are packed so that
// slots 0..24 of row kq are ALL 25 parameters of transformed dim kq of the tile and slots 25..31 a part of dim 4 (7 + 6 + 6 + 6): four
// whole-wave evaluations (every lane busy: 16 points x 4 dims) and a fifth in which row kq takes dim 4 of point block kq after a 4 x 4
// transpose of those seven registers across the rows (v_permlane16_swap + v_permlane32_swap, one swap per register).  64 points x 5 dims =
// 5 full wave evaluations (VAR 11: 6 for the same points, one half empty).
#include "common.h"
#include "activations.h"
#include "spline.h"
#include <cstdio>
#include <cstdlib>

namespace fc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct SplineWideParams {
    const unsigned short* A16;   // [rows][KT][64] fp16: the activation image, one-accumulator form (pre-scaled by kOneAccActScale, lo unscaled)
    const unsigned short* W1;    // [nbn * 256][KT][64] fp16: the weight image in this kernel's column order, pre-scaled by 2^w1_exp
    const float* bias1;          // [nbn * 256] bias in the same order, times kOneAccActScale * 2^w1_exp
    float* xbuf; int ldx; int x2_col0; int d2; int rows_valid;
    float* ldj_part; size_t ldj_pitch;
    int KT;                      // k32 steps (even)
    int nbm, nbn;                // 256-row tiles, 256-column tiles (= pairs of 128-column spline tiles)
    int ntile128;                // 128-column spline tiles that exist (the last pair may be half empty)
    int col_group;
    float out_scale;             // 1 / (kOneAccActScale * 2^w1_exp)
    int ablate;                  // diagnostic knob 14: 1 no spline evaluation, 2 main loop only (results invalid)
    // EPI 1 (a Linear layer of the coupling MLP: bias in the accumulators, residual, exact-erf GELU, output as a limb image)
    unsigned short* out16;       // [rows][N/16][hi 16 | lo 16] fp16
    const unsigned short* res16; // residual as a one-accumulator image of the same shape, or null (models/nets.py:27: odd hidden layers)
    int n16;                     // 16-column blocks per row of out16 / res16
    float s1, s2;                // output limb split: hi = rn16(v s1), lo = rn16((v s1 - hi) s2): (kOneAccActScale, 1) or (1, 2048)
    int* ovf;                    // split-fp16 range flag (common.h Fp16Guard)
    // EPI 2 (plain product, fc_debug_one_acc_gemm_f32: the one-accumulator limb form by itself, for the accuracy test)
    float* C; int ldc;
    // EPI 3 (training Linear, csrc/train.hip): the point operand is an fp32 PANEL [rows][lda] that is split into limbs after its LDS read
    // (a k32 step of a row is 128 bytes either way), scaled by a_scale or, for gradients, per row so that the row's maximum lands in
    // [2^13, 2^14) (row_absmax: max |a| of every row, written by the producing kernel); C = (acc / scale (+ addend)) (* act'(gradu))
    const float* A32; int lda;
    const float* row_absmax;
    float a_scale;
    const float* addend; const float* gradu; int ldgu; int gact;
    int n_cols;                  // columns of C that exist (a multiple of 4; the last 256-column tile may be partly empty)
    int nt_store;                // C is written with non-temporal stores
};

// column of the kernel's tile order: row kq = (c >> 2) & 3 of 16-parameter block jb = c >> 4, register r = c & 3 -> slot s = 4 jb + r
//   s < 25: parameter s of the tile's dim kq;  s >= 25: parameter base(kq) + s - 25 of dim 4 (kq = 0: 0..6, 1: 7..12, 2: 13..18, 3: 19..24)
// returns the column of the SAME tile in spline.h's order (what PackedLinear.W holds), or -1 for the three unused slots
__host__ __device__ inline int spline_wide_src_col(int c) {
    const int jb = c >> 4, kq = (c >> 2) & 3, r = c & 3, s = 4 * jb + r;
    if (s < 25) return spline_col(kq, s, 8);
    const int base = kq == 0 ? 0 : 1 + 6 * kq, cnt = kq == 0 ? 7 : 6, i = s - 25;
    return i < cnt ? spline_col(4, base + i, 8) : -1;
}

// ---------------------------------------------------------------- weight image + bias in the kernel's order (fc_flow_create)
__global__ __launch_bounds__(256) void spline_wide_image_kernel(const float* __restrict__ W, const float* __restrict__ bias, int n_src, int K_pad, float wscale,
                                                                float bscale, unsigned short* __restrict__ W1, float* __restrict__ bias1, size_t n, int permute) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const size_t row = t / K_pad;
    const int k = (int)(t - row * K_pad);
    const int tile = (int)(row >> 7), c = (int)(row & 127);
    const int sc = permute ? spline_wide_src_col(c) : c;
    const int srow = tile * 128 + sc;
    const bool live = sc >= 0 && srow < n_src;
    const float x = live ? W[(size_t)srow * K_pad + k] * wscale : 0.f;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    const size_t blk = row * (K_pad / 16) + k / 16;
    W1[blk * 32 + (k & 15)] = __builtin_bit_cast(unsigned short, h);
    W1[blk * 32 + 16 + (k & 15)] = __builtin_bit_cast(unsigned short, l);
    if (k == 0 && bias1) bias1[row] = live && bias ? bias[srow] * bscale : 0.f;
}

int g_spline_wide_dma = 0;   // developer knob 27 (--dev builds): DMA pieces per phase, 0 = {2,3,3,0} / {2,3,3,0} (shipped: -1.5 % against {1,3,3,1} / {2,3,3,0}, same box)
int g_spline_wide_colgroup = -1;   // knob 28: column-group size of the tile order in 256-column tiles (-1 = shipped: 5)

// rows of 16 lanes (a0,a1,a2,a3 | b0,b1,b2,b3):  swap32 -> a = (a0,a1,b0,b1), b = (a2,a3,b2,b3) ;  swap16 -> a = (a0,b0,a2,b2), b = (a1,b1,a3,b3)
__device__ __forceinline__ void sw_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void sw_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
// sum over the four rows of 16 lanes, result in every row (fixed order: (r0 + r1) + (r2 + r3))
__device__ __forceinline__ float sw_row_sum(float v) {
    float a = v, b = v;
    sw_swap16(a, b);
    float c = a + b, d = c;
    sw_swap32(c, d);
    return c + d;
}

// rq_spline_fwd_regs (spline.h) on parameters that arrive SCALED: logical parameter i = u(i) * os with os an exact power of two.  The
// scale rides inside the softmax's existing fma (fma(u, L2E os, -max u L2E os) is bit for bit fma(u os, L2E, -max(u os) L2E)), the two
// derivative logits are scaled after their selection: same operations and roundings as the unscaled routine on u os.
template <int K, class U>
__device__ __forceinline__ void rq_spline_fwd_regs_scaled(float x, const U& u, float os, float& y, float& lad) {
    constexpr float B = 3.0f, MINW = 1e-3f, MINH = 1e-3f, MIND = 1e-3f, L2E = 1.4426950408889634f;
    const bool inside = x >= -B && x <= B;
    const float l2s = L2E * os;
    float ew[K], eh[K], mw = u(0), mh = u(K);
#pragma unroll
    for (int i = 0; i < K; ++i) { ew[i] = u(i); eh[i] = u(K + i); mw = fmaxf(mw, ew[i]); mh = fmaxf(mh, eh[i]); }
    float sw = 0.f, sh = 0.f;
    const float ow = -mw * l2s, oh = -mh * l2s;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        ew[i] = __builtin_amdgcn_exp2f(fmaf(ew[i], l2s, ow)); sw += ew[i];
        eh[i] = __builtin_amdgcn_exp2f(fmaf(eh[i], l2s, oh)); sh += eh[i];
    }
    const float fw = (1.0f - MINW * K) * __builtin_amdgcn_rcpf(sw), fh = (1.0f - MINH * K) * __builtin_amdgcn_rcpf(sh);
    float c = 0.f, in_cw = -B, hi = INFINITY;
    float ud0r = 0.f, ud1r = u(2 * K);
    int bin = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        c += fmaf(fw, ew[i], MINW);
        const float knot = i == K - 1 ? B : fmaf(2.0f * B, c, -B);
        const bool ge = x >= (i == K - 1 ? knot + 1e-6f : knot);
        bin += ge ? 1 : 0;
        in_cw = ge ? knot : in_cw;
        hi = ge ? hi : fminf(hi, knot);
        ud0r = ge ? u(2 * K + i) : ud0r;
        ud1r = ge ? u(2 * K + i + 1) : ud1r;
    }
    const float in_w = hi - in_cw;
    float ch = 0.f, in_ch = -B, ch_hi = B;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        ch += fmaf(fh, eh[i], MINH);
        const float knot = i == K - 1 ? B : fmaf(2.0f * B, ch, -B);
        in_ch = (i + 1 == bin) ? knot : in_ch;
        ch_hi = (i == bin) ? knot : ch_hi;
    }
    const float in_h = ch_hi - in_ch;
    const float ud0 = bin == 0 ? -1e-3f : ud0r * os, ud1 = ud1r * os;      // bin 0: left pad log(exp(1 - min_derivative - 1))
    const float d0 = MIND + (ud0 > 20.f ? ud0 : fast_log(1.0f + fast_exp(ud0)));
    const float d1 = MIND + (ud1 > 20.f ? ud1 : fast_log(1.0f + fast_exp(ud1)));
    const float rw = __builtin_amdgcn_rcpf(in_w);
    const float delta = in_h * rw;
    const float th = (x - in_cw) * rw;
    const float tt = th * (1.0f - th);
    const float num = in_h * (delta * th * th + d0 * tt);
    const float den = delta + (d0 + d1 - 2.0f * delta) * tt;
    const float yy = in_ch + fast_div(num, den);
    const float omt = 1.0f - th;
    const float dnum = delta * delta * (d1 * th * th + 2.0f * delta * tt + d0 * omt * omt);
    const float ll = fast_log(dnum) - 2.0f * fast_log(den);
    y = inside ? yy : x;
    lad = inside ? ll : 0.f;
}

// eight fp32 values (k = 8 kq .. 8 kq + 7 of one point) -> the hi / lo operand fragments of the one-accumulator form, x s = hi + lo with s an
// exact power of two; amax collects max |x s| (the caller turns it into the range flag).  Four instructions per pair of values + the running maximum (measured issue
// costs on gfx950, profiles/micro/valu_rate_probe.hip: v_pk_mul_f32 6.6, v_cvt_pk_f16_f32 8.1, v_fma_mix_f32 8.4 cycles per wave; the all-mix
// form of activations.h limb_split2s is six at 8.4 - 9.3).
typedef float sw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sw_split2v(sw_f32x2 x, sw_f32x2 sv, unsigned& hi, unsigned& lo, float& amax) {
    sw_f32x2 t;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(sv));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(t[0]), "v"(t[1]));
    // (x s - hi is exact in fp32, so v_fma_mixlo / mixhi_f16 round it once, straight into the halves of lo: the bits of the two v_fma_mix_f32 +
    //  v_cvt_pk_f16_f32 this replaced, one instruction fewer per pair -- activations.h limb_split2u)
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(t[0]), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(t[1]), "v"(hi));
    asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(t[0]), "v"(t[1]));      // (volatile: pins the running maximum; left to hipcc the raw values of a k step are kept for one reduction tree and spill)
}
__device__ __forceinline__ void sw_split8(const float4& r0, const float4& r1, float sv, f16x8& hi, f16x8& lo, float& amax) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 h, l;
    unsigned a, b;
    const sw_f32x2 s2 = {sv, sv};
    sw_split2v(sw_f32x2{r0.x, r0.y}, s2, a, b, amax); h[0] = a; l[0] = b;
    sw_split2v(sw_f32x2{r0.z, r0.w}, s2, a, b, amax); h[1] = a; l[1] = b;
    sw_split2v(sw_f32x2{r1.x, r1.y}, s2, a, b, amax); h[2] = a; l[2] = b;
    sw_split2v(sw_f32x2{r1.z, r1.w}, s2, a, b, amax); h[3] = a; l[3] = b;
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}

constexpr int SW_LDS = 4 * 32768 + 2 * 1024;      // two stages of (256 point rows + 256 weight rows) x 128 B, two bias buffers of 256 floats

// DMA pieces per LOAD segment of a k step, for the group that fetches the points (waves 0-3: P*) and the weights (waves 4-7: Q*); the
// lagging group must not issue in its last segment (it waits for its pieces there)
// EPI 0: the fused spline coupling; EPI 1: a 512-wide Linear layer of the coupling MLP with GELU (same main loop, same one-accumulator
// arithmetic; the weight image in natural row order)
template <int EPI, int P0, int P1, int P2, int P3, int Q0, int Q1, int Q2, int Q3>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2)))
void spline_wide_kernel(const SplineWideParams p) {
    static_assert(P0 + P1 + P2 + P3 == 8 && Q0 + Q1 + Q2 == 8 && Q3 == 0, "eight pieces per wave and k step");
    extern __shared__ char smc[];
    typedef __attribute__((address_space(3))) char lds_char;
    typedef const __attribute__((address_space(1))) char glb_char;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, pw = wave & 3;
    const int KT = p.KT;
    constexpr bool AF32 = EPI == 3;                                      // the point operand is an fp32 panel (split into limbs in registers)
    const unsigned rowbytes = (AF32 && grp == 0) ? (unsigned)p.lda * 4u : (unsigned)KT * 128u;      // of the operand THIS wave fetches
    const int ntiles = p.nbm * p.nbn, G = gridDim.x;
    int t = blockIdx.x;
    if (t >= ntiles) return;
    float* biasbuf = reinterpret_cast<float*>(smc + 4 * 32768);          // [2][256]

    auto tile_of = [&](int b, int& bm, int& bn) {                       // XCD-aware order (gemm.hip): blocks b, b + 8, ... share an XCD
        const int xcd = b & 7, loc = b >> 3;
        if (p.col_group > 0) {
            const int rows_x = p.nbm >> 3, Gc = p.col_group;
            const int g = loc / (rows_x * Gc);
            const int rem = loc - g * rows_x * Gc;
            const int w = p.nbn - g * Gc < Gc ? p.nbn - g * Gc : Gc;
            const int r = rem / w;
            bm = xcd * rows_x + r;
            bn = g * Gc + (rem - r * w);
        } else {
            const int q = ntiles >> 3, r = ntiles & 7;
            const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
            bm = L / p.nbn;
            bn = L - bm * p.nbn;
        }
    };
    // LDS: [stage 0 points 32 KB | stage 1 points | stage 0 weights | stage 1 weights]; a row = 128 B = one k32 step of one point / weight row
    // = 8 chunks of 16 B, logical chunk c = 4 (k16 block) + 2 limb + (k half) at physical chunk c ^ ((row >> 1) & 7).
    // DMA piece i of this wave: operand rows (wave & 3) * 64 + 8 i + (lane >> 3); the swizzle term depends on the parity of i only.
    unsigned poff[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int r = pw * 64 + par * 8 + (lane >> 3);
        const int cl = (lane & 7) ^ ((r >> 1) & 7);
        poff[par] = (unsigned)r * rowbytes + cl * 16;
    }
    auto src_of = [&](int bm, int bn) -> const char* {
        return grp == 0 ? (AF32 ? reinterpret_cast<const char*>(p.A32) : reinterpret_cast<const char*>(p.A16)) + (size_t)bm * 256 * rowbytes
                        : reinterpret_cast<const char*>(p.W1) + (size_t)bn * 256 * rowbytes;
    };
    const int dst0 = grp * 65536 + pw * 8192;
#define SW_DMA(SRC_, ST_, I0_, N_)                                                                                                      \
    {                                                                                                                                     \
        _Pragma("unroll") for (int i_ = (I0_); i_ < (I0_) + (N_); ++i_)                                                                 \
            __builtin_amdgcn_global_load_lds((glb_char*)((SRC_) + (size_t)(i_ >> 1) * 16 * rowbytes + poff[i_ & 1]),                      \
                                             (lds_char*)(smc + dst0 + (ST_) * 32768 + i_ * 1024), 16, 0, 0);                            \
    }
    auto bias_dma = [&](int bn, int par) {                              // 256 floats: waves 0-3, 4 bytes per lane
        if (grp == 0 && (EPI != 3 || p.bias1))
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) float*)(p.bias1 + bn * 256 + pw * 64 + lane),
                                             (__attribute__((address_space(3))) float*)(biasbuf + par * 256 + pw * 64), 4, 0, 0);
    };
    // a k32 step is ONE k extent of the 16x16x32 MFMA: lane (l15, kq) supplies row l15 of a 16-row block, k quarter kq = chunks 0, 1, 4, 5 of
    // the hi limb / 2, 3, 6, 7 of the lo limb
    const int xsw = (l15 >> 1) & 7;
    int abase[2], bbase[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = (((kq >> 1) * 4 + q * 2 + (kq & 1)) ^ xsw) * 16;
        abase[q] = (pw * 64 + l15) * 128 + c;
        bbase[q] = 65536 + (grp * 128 + l15) * 128 + c;
    }
    // EPI 3: the point rows arrive as fp32 (a k32 step of a row = 32 floats = the same 128 bytes) and are turned into the limb image IN PLACE, one
    // k step ahead of their use: lane (l15, kq) reads the eight floats k = 8 kq .. 8 kq + 7 (chunks 2 kq, 2 kq + 1) and writes their hi / lo
    // limbs to the image's chunks; wave w converts point blocks 0, 1 of its 64 rows, wave w + 4 (the other group, same rows) blocks 2, 3
    const int cbase = (pw * 64 + grp * 32 + l15) * 128;
    const int cin0 = ((2 * kq) ^ xsw) * 16, cin1 = ((2 * kq + 1) ^ xsw) * 16;
    const int cout0 = abase[0] - (pw * 64 + l15) * 128, cout1 = abase[1] - (pw * 64 + l15) * 128;
    // EPI 3: scale of this lane's four points (point (ib, l15)) and what undoes it in the epilogue
    float csc[2] = {1.f, 1.f}, amax = 0.f;                               // scales of the two point blocks this wave converts (of the tile whose data comes next)
    bool bad_row = false;
    auto row_scale = [&](int bm, int ib0, int n, float* sc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= n) break;
            const int ib = ib0 + j;
            if (p.row_absmax) {
                const unsigned bits = __builtin_bit_cast(unsigned, p.row_absmax[bm * 256 + pw * 64 + ib * 16 + l15]);
                const int e = (int)((bits >> 23) & 0xffu);                       // biased exponent of the row's maximum: 2^(e - 127) <= max < 2^(e - 126)
                int se = e == 0 ? 127 : 267 - e;                                 // scale 2^(13 - (e - 127)): max lands in [2^13, 2^14)
                se = se > 254 ? 254 : se;
                bad_row |= e == 255;                                             // inf / NaN in the row: range flag
                sc[j] = __builtin_bit_cast(float, (unsigned)se << 23);
            } else sc[j] = p.a_scale;
        }
    };
    // converts this wave's two point blocks of LDS stage `st` in place (reads before writes: one wave's LDS operations execute in order)
    auto convert_stage = [&](int st) {
        char* base = smc + st * 32768 + cbase;
        const float4 a0 = *reinterpret_cast<const float4*>(base + cin0), a1 = *reinterpret_cast<const float4*>(base + cin1);
        const float4 b0 = *reinterpret_cast<const float4*>(base + 2048 + cin0), b1 = *reinterpret_cast<const float4*>(base + 2048 + cin1);
        f16x8 h0, l0, h1, l1;
        sw_split8(a0, a1, csc[0], h0, l0, amax);
        sw_split8(b0, b1, csc[1], h1, l1, amax);
        *reinterpret_cast<f16x8*>(base + cout0) = h0;
        *reinterpret_cast<f16x8*>(base + cout1) = l0;
        *reinterpret_cast<f16x8*>(base + 2048 + cout0) = h1;
        *reinterpret_cast<f16x8*>(base + 2048 + cout1) = l1;
    };
    // this lane's share of a tile's x2 operands: dim kq of point (ib, l15) for ib = 0..3, dim 4 of point (kq, l15), and that point's log-det slot
    auto load_x = [&](int bm, int bn, float (&x)[5], float& ldj) {
        const int t128 = 2 * bn + grp, dim0 = t128 * 5;
        const int row0 = bm * 256 + pw * 64 + l15;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const int row = row0 + 16 * ib;
            x[ib] = row < p.rows_valid && dim0 + kq < p.d2 ? p.xbuf[(size_t)row * p.ldx + p.x2_col0 + dim0 + kq] : 0.f;
        }
        const int rowq = row0 + 16 * kq;
        x[4] = rowq < p.rows_valid && dim0 + 4 < p.d2 ? p.xbuf[(size_t)rowq * p.ldx + p.x2_col0 + dim0 + 4] : 0.f;
        ldj = t128 < p.ntile128 ? p.ldj_part[(size_t)t128 * p.ldj_pitch + rowq] : 0.f;
    };

    int bm, bn;
    tile_of(t, bm, bn);
    const char* src = src_of(bm, bn);
    SW_DMA(src, 0, 0, 8)
    bias_dma(bn, 0);
    float spl_x[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, spl_ldj = 0.f;
    if constexpr (EPI == 0) load_x(bm, bn, spl_x, spl_ldj);
    float omax = 0.f;
    if constexpr (EPI == 3) row_scale(bm, 2 * grp, 2, csc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (AF32) {
        convert_stage(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();                         // the second group runs one segment behind the first

    floatx4 acc[4][8];
    int par = 0;
    for (;;) {
        const int tn = t + G;
        const bool has_next = tn < ntiles;
        int nbm = bm, nbn = bn;
        if (has_next) tile_of(tn, nbm, nbn);
        const char* nsrc = src_of(nbm, nbn);
        float nx[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, nldj = 0.f;
        {
            // accumulators start from the (pre-scaled) bias: register r of block jb is tile column 16 jb + 4 kq + r for every point block
            const float* bb = biasbuf + par * 256 + grp * 128 + 4 * kq;
            const bool has_bias = EPI != 3 || p.bias1 != nullptr;
#pragma unroll
            for (int jb = 0; jb < 8; ++jb) {
                const float4 b4 = has_bias ? *reinterpret_cast<const float4*>(bb + jb * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) { acc[ib][jb][0] = b4.x; acc[ib][jb][1] = b4.y; acc[ib][jb][2] = b4.z; acc[ib][jb][3] = b4.w; }
            }
        }
#define SW_PHASE(ST_, F_, NP_, NQ_, I0P_, I0Q_, LAST_)                                                                                   \
        {                                                                                                                                 \
            if ((F_) == 0) {                                                                                                              \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                            \
                    _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                        \
                        xf[i][q] = *reinterpret_cast<const f16x8*>(smc + abase[q] + (ST_) * 32768 + i * 2048);                            \
            }                                                                                                                             \
            _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                             \
                _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                            \
                    wf[jj][q] = *reinterpret_cast<const f16x8*>(smc + bbase[q] + (ST_) * 32768 + (2 * (F_) + jj) * 2048);                 \
            if (grp == 0) { if ((NP_) > 0) SW_DMA(dsrc, (ST_) ^ 1, I0P_, NP_) }                                                          \
            else { if ((NQ_) > 0) SW_DMA(dsrc, (ST_) ^ 1, I0Q_, NQ_) }                                                                   \
            if constexpr (AF32) {                                        /* the NEXT k step's point rows (all of them issued in phases 0, 1): fp32 -> limb image in place */ \
                if ((F_) == 3) {                                                                                                          \
                    if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     /* own rows; the other group converts behind the next barrier */ \
                    convert_stage((ST_) ^ 1);                                                                                             \
                }                                                                                                                         \
            }                                                                                                                             \
            if ((LAST_) && grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            __builtin_amdgcn_s_barrier();                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            __builtin_amdgcn_s_setprio(1);                                                                                                \
            _Pragma("unroll") for (int pr = 0; pr < 3; ++pr)                                                                             \
                _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                         \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                        \
                        acc[i][2 * (F_) + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jj][pr == 1 ? 1 : 0], xf[i][pr == 2 ? 1 : 0],  \
                                                                                       acc[i][2 * (F_) + jj], 0, 0, 0);                   \
            __builtin_amdgcn_s_setprio(0);                                                                                                \
            if ((LAST_) && grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            __builtin_amdgcn_s_barrier();                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
        }
#define SW_STEP(ST_)                                                                                                                      \
        {                                                                                                                                 \
            SW_PHASE(ST_, 0, P0, Q0, 0, 0, 0)                                                                                             \
            SW_PHASE(ST_, 1, P1, Q1, P0, Q0, 0)                                                                                           \
            SW_PHASE(ST_, 2, P2, Q2, P0 + P1, Q0 + Q1, 0)                                                                                 \
            SW_PHASE(ST_, 3, P3, Q3, P0 + P1 + P2, Q0 + Q1 + Q2, 1)                                                                       \
        }
        f16x8 xf[4][2], wf[2][2];
        for (int kt = 0; kt < KT; kt += 2) {
            const char* dsrc = src + (size_t)(kt + 1) * 128;
            SW_STEP(0)
            const bool lastk = kt + 2 >= KT;
            dsrc = lastk ? nsrc : src + (size_t)(kt + 2) * 128;
            if (lastk && has_next) {                                    // the stream runs on into the next tile: its bias and x2 / log-det operands too
                bias_dma(nbn, par ^ 1);
                if constexpr (EPI == 0) load_x(nbm, nbn, nx, nldj);
                if constexpr (EPI == 3) row_scale(nbm, 2 * grp, 2, csc);      // (this tile's remaining conversion is the next tile's first k step)
            }
            SW_STEP(1)
        }
        // ---------------------------------------------------------------- epilogue in registers
        // Tile boundary: group 0 has finished its last MFMA segment one barrier before group 1.  It waits that one segment out, so that BOTH
        // groups evaluate their splines in the same interval: two waves per SIMD issue VALU work at twice the rate of one (a lone wave issues
        // an instruction every 4 cycles), and the evaluation is ~1400 VALU instructions per wave and tile.  Group 1 then re-enters one barrier
        // behind group 0 again.  (First form of this kernel: group 0's evaluation beside group 1's last MFMAs, group 1's beside group 0's
        // first: the two evaluations stood in series, 0.13 of 0.63 ms per launch.)
        if (grp == 0) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (EPI == 3) {
            // ---- training Linear: C = (acc / scale + addend) * act'(gradu); a lane holds columns 16 jb + 4 kq + 0..3 of point (ib, l15)
            float asc[4];
            row_scale(bm, 0, 4, asc);
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const size_t row = (size_t)(bm * 256 + pw * 64 + ib * 16 + l15);
                const int col0 = bn * 256 + grp * 128 + 4 * kq;
                // (a per-row scale is an exact power of two 2^(se - 127): its inverse is the exponent 254 - se)
                const float sc = p.row_absmax ? __builtin_bit_cast(float, (254u - (__builtin_bit_cast(unsigned, asc[ib]) >> 23)) << 23) * p.out_scale : p.out_scale;
#pragma unroll
                for (int jb = 0; jb < 8; ++jb) {
                    const int col = col0 + jb * 16;
                    if (col < p.n_cols) {
                        float4 v = make_float4(acc[ib][jb][0] * sc, acc[ib][jb][1] * sc, acc[ib][jb][2] * sc, acc[ib][jb][3] * sc);
                        if (p.addend) {
                            const float4 a = *reinterpret_cast<const float4*>(p.addend + row * p.ldc + col);
                            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
                        }
                        if (p.gradu) {
                            const float4 u = *reinterpret_cast<const float4*>(p.gradu + row * p.ldgu + col);
                            v.x *= fc_gelu_grad(u.x); v.y *= fc_gelu_grad(u.y); v.z *= fc_gelu_grad(u.z); v.w *= fc_gelu_grad(u.w);      // (GELU only: launch_train_wide checks)
                        }
                        typedef float sw_f4 __attribute__((ext_vector_type(4)));
                        const sw_f4 v4 = {v.x, v.y, v.z, v.w};
                        if (p.nt_store) __builtin_nontemporal_store(v4, reinterpret_cast<sw_f4*>(p.C + row * p.ldc + col));      // (a GB-sized panel streams out past the L2)
                        else *reinterpret_cast<sw_f4*>(p.C + row * p.ldc + col) = v4;
                    }
                }
            }
        } else
        if constexpr (EPI == 2) {
            const float os = p.out_scale;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                float* cr = p.C + (size_t)(bm * 256 + pw * 64 + ib * 16 + l15) * p.ldc + bn * 256 + grp * 128 + 4 * kq;
#pragma unroll
                for (int jb = 0; jb < 8; ++jb)
                    *reinterpret_cast<float4*>(cr + jb * 16) = make_float4(acc[ib][jb][0] * os, acc[ib][jb][1] * os, acc[ib][jb][2] * os, acc[ib][jb][3] * os);
            }
        } else
        if constexpr (EPI == 1) {
            // ---- Linear + GELU: value = acc os (+ residual), y = gelu(value), stored as the limb image the next layer copies.  A lane holds features
            // 16 jb + 4 kq + 0..3 of point (ib, l15); one v_permlane16_swap per limb word pairs the rows kq = 2 h, 2 h + 1 so that an even row holds
            // the hi limbs of features 8 h .. 8 h + 7 of its 16-block and the odd row their lo limbs: ONE 16-byte store per lane and block, 64
            // contiguous bytes per point (the image's [hi 16 | lo 16] block).  The residual image is read the same way and un-swapped.
            if (p.ablate != 2) {
                const float os = p.out_scale, s1 = p.s1, s2 = p.s2, rinv = 1.0f / kOneAccActScale;
                const size_t rowb = (size_t)p.n16 * 64;
                const size_t col_off = (size_t)(bn * 16 + grp * 8) * 64 + (kq & 1) * 32 + (kq >> 1) * 16;
                // the residual words of point block ib + 1 are requested before block ib is evaluated (two register sets): left inside the block
                // loop every load was waited for where it was issued, ~1 us of exposed latency per block and 32 blocks per tile
                uint4 rr[2][4];                                   // (half a point block = four 16-column blocks per set: a whole block per set spills)
                auto res_issue = [&](int hb, uint4 (&r)[4]) {
                    const char* src = reinterpret_cast<const char*>(p.res16) + (size_t)(bm * 256 + pw * 64 + (hb >> 1) * 16 + l15) * rowb + col_off + (hb & 1) * 256;
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq) r[jq] = *reinterpret_cast<const uint4*>(src + jq * 64);
                };
                if (p.res16) res_issue(0, rr[0]);
#pragma unroll
                for (int hb = 0; hb < 8; ++hb) {
                    const int ib = hb >> 1;
                    const size_t off = (size_t)(bm * 256 + pw * 64 + ib * 16 + l15) * rowb + col_off;
                    if (p.res16 && hb + 1 < 8) res_issue(hb + 1, rr[(hb + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int jb = (hb & 1) * 4; jb < (hb & 1) * 4 + 4; ++jb) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[ib][jb][r] * os;
                        if (p.res16) {
                            // even row: {hi f0-3 (own), hi f4-7 (partner's)}, odd row: {lo f0-3 (partner's), lo f4-7 (own)} -> swap back -> own hi / lo words
                            const uint4 rw = rr[hb & 1][jb & 3];
                            float h0 = __builtin_bit_cast(float, rw.x), h1 = __builtin_bit_cast(float, rw.y);
                            float l0 = __builtin_bit_cast(float, rw.z), l1 = __builtin_bit_cast(float, rw.w);
                            sw_swap16(h0, l0);
                            sw_swap16(h1, l1);
                            const unsigned hw[2] = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
                            const unsigned lw[2] = {__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const unsigned short hb = (unsigned short)(hw[r >> 1] >> (16 * (r & 1))), lb = (unsigned short)(lw[r >> 1] >> (16 * (r & 1)));
                                v[r] += ((float)__builtin_bit_cast(_Float16, hb) + (float)__builtin_bit_cast(_Float16, lb)) * rinv;      // (hi + lo is exact in fp32)
                            }
                        }
                        unsigned hw2[2], lw2[2];
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            const float y0 = fc_gelu(v[r]) * s1, y1 = fc_gelu(v[r + 1]) * s1;
                            omax = fmaxf(omax, fmaxf(fabsf(y0), fabsf(y1)));
                            asm volatile("" : "+v"(omax));              // (pins the running maximum here: hipcc otherwise keeps all 128 values for one reduction tree at the end and spills them)
                            const _Float16 a0 = (_Float16)y0, a1 = (_Float16)y1;
                            const _Float16 b0 = (_Float16)((y0 - (float)a0) * s2), b1 = (_Float16)((y1 - (float)a1) * s2);
                            hw2[r >> 1] = (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, a1) << 16);
                            lw2[r >> 1] = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
                        }
                        float x0 = __builtin_bit_cast(float, hw2[0]), x1 = __builtin_bit_cast(float, hw2[1]);
                        float y0 = __builtin_bit_cast(float, lw2[0]), y1 = __builtin_bit_cast(float, lw2[1]);
                        sw_swap16(x0, y0);
                        sw_swap16(x1, y1);
                        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.out16) + off + jb * 64) =
                            make_uint4(__builtin_bit_cast(unsigned, x0), __builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, y0), __builtin_bit_cast(unsigned, y1));
                        if (jb & 1) __builtin_amdgcn_sched_barrier(0);            // (left alone the scheduler interleaves all 32 blocks' GELU chains and spills the accumulators)
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
            }
        } else
        if (p.ablate != 2) {
            const float os = p.out_scale;
            const int t128 = 2 * bn + grp, dim0 = t128 * 5;
            const int row0 = bm * 256 + pw * 64 + l15, rowq = row0 + 16 * kq;
            // slots 25..31 (registers 1..3 of block 6, 0..3 of block 7): the parts of dim 4 -> row d gets the parts of point block d
#pragma unroll
            for (int s = 25; s < 32; ++s) {
                float x0 = acc[0][s >> 2][s & 3], x1 = acc[1][s >> 2][s & 3], x2 = acc[2][s >> 2][s & 3], x3 = acc[3][s >> 2][s & 3];
                sw_swap16(x0, x1);
                sw_swap16(x2, x3);
                sw_swap32(x0, x2);
                sw_swap32(x1, x3);
                acc[0][s >> 2][s & 3] = x0; acc[1][s >> 2][s & 3] = x1; acc[2][s >> 2][s & 3] = x2; acc[3][s >> 2][s & 3] = x3;
            }
            float yv[5], lv[5];
            if (p.ablate == 1) {
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) { yv[ib] = spl_x[ib] + acc[ib][0][0] * os; lv[ib] = acc[ib][0][1] * os; }
                yv[4] = spl_x[4] + acc[0][6][1] * os; lv[4] = acc[1][6][1] * os;
            } else {
#pragma unroll
                for (int ib = 0; ib < 4; ++ib)
                    rq_spline_fwd_regs_scaled<8>(spl_x[ib], [&](int q) { return acc[ib][q >> 2][q & 3]; }, os, yv[ib], lv[ib]);
                // dim 4: parameters 0..6 from (transposed) block 0, 7..12 from block 1, 13..18 from block 2, 19..24 from block 3, slots 25 + i
                rq_spline_fwd_regs_scaled<8>(spl_x[4], [&](int q) {
                    const int part = q < 7 ? 0 : (q - 1) / 6, s = 25 + (q < 7 ? q : (q - 1) % 6);
                    return acc[part][s >> 2][s & 3];
                }, os, yv[4], lv[4]);
            }
            const bool dk = dim0 + kq < p.d2, d4 = dim0 + 4 < p.d2;
            float tot = 0.f;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const bool v = dk && row0 + 16 * ib < p.rows_valid;
                const float s4 = sw_row_sum(v ? lv[ib] : 0.f);             // dims 0..3 of point (ib, l15), in every row
                tot = kq == ib ? s4 : tot;
                if (v) p.xbuf[(size_t)(row0 + 16 * ib) * p.ldx + p.x2_col0 + dim0 + kq] = yv[ib];
            }
            const bool vq = rowq < p.rows_valid;
            if (vq && d4) { p.xbuf[(size_t)rowq * p.ldx + p.x2_col0 + dim0 + 4] = yv[4]; tot += lv[4]; }
            if (t128 < p.ntile128) p.ldj_part[(size_t)t128 * p.ldj_pitch + rowq] = spl_ldj + tot;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
        }
        if (grp == 1) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!has_next) break;
        t = tn; bm = nbm; bn = nbn; src = nsrc; par ^= 1;
#pragma unroll
        for (int i = 0; i < 5; ++i) spl_x[i] = nx[i];
        spl_ldj = nldj;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    if constexpr (EPI == 1) { if (!(omax < 65504.0f) && p.ovf) atomicOr(p.ovf, 1); }      // (also on a NaN)
    if constexpr (EPI == 3) { if (p.ovf && (bad_row || !(amax < 65504.0f))) atomicOr(p.ovf, 1); }      // (amax: of the scaled values; also on a NaN)
#undef SW_DMA
#undef SW_PHASE
#undef SW_STEP
}

// ---------------------------------------------------------------- host side
bool spline_wide_eligible(const PackedLinear& L, int K_bins) {
    return K_bins == 8 && L.W1 != nullptr && L.w1_permuted && L.bias1 != nullptr && L.nseg == 1 && L.K_pad % 64 == 0 && L.N_pad % 128 == 0;
}

// Attaches the kernel's weight image and bias to the packed spline parameter layer (K = 8 bins; W / bias hold spline.h's column order).
// wmax = max |w| over the layer (host side, from the checkpoint tensor).
void spline_wide_attach(DeviceArena& arena, PackedLinear& L, float wmax, hipStream_t s, bool permute) {
    if (!L.W || !L.bias || !L.W2 || L.nseg != 1 || L.K_pad % 64 != 0 || L.N_pad % (permute ? 128 : 256) != 0 || L.n_alloc < L.N_pad) return;
    if (!(wmax < 65504.0f)) return;
    int e = 0;
    if (wmax > 0.f) {
        while (ldexpf(wmax, e) >= 32768.0f) --e;
        while (ldexpf(wmax, e) < 16384.0f && e < 100) ++e;
    }
    const int rows = round_up(L.N_pad, 256);
    const size_t n = (size_t)rows * L.K_pad;
    L.W1 = (unsigned short*)arena.alloc_bytes(n * 2 * sizeof(unsigned short));
    L.bias1 = arena.alloc_floats((size_t)rows);
    L.w1_exp = e;
    L.w1_permuted = permute;
    spline_wide_image_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(L.W, L.bias, L.N_pad, L.K_pad, ldexpf(1.f, e), kOneAccActScale * ldexpf(1.f, e), L.W1, L.bias1, n, permute ? 1 : 0);
    FC_HIP(hipGetLastError());
}

extern int g_spline_ablate;

void launch_spline_wide(const PackedLinear& L, const GemmEpi& e, int rows_alloc, hipStream_t s) {
    if (!spline_wide_eligible(L, e.spline_K) || !e.A16 || e.a16_scale != kOneAccActScale || rows_alloc % 256 != 0 || !e.xbuf || !e.ldj_part ||
        e.ldj_pitch < (size_t)rows_alloc || L.N_pad != spline_ncols(e.d2, 8))
        throw Error(FC_ERR_INVALID, "launch_spline_wide: needs the one-accumulator activation image, the attached weight image and rows padded to 256");
    SplineWideParams p{};
    p.A16 = e.A16; p.W1 = L.W1; p.bias1 = L.bias1;
    p.xbuf = e.xbuf; p.ldx = e.ldx; p.x2_col0 = e.x2_col0; p.d2 = e.d2; p.rows_valid = e.rows_valid;
    p.ldj_part = e.ldj_part; p.ldj_pitch = e.ldj_pitch;
    p.KT = L.K_pad / 32;
    p.nbm = rows_alloc / 256;
    p.ntile128 = L.N_pad / 128;
    p.nbn = (p.ntile128 + 1) / 2;
    p.col_group = (p.nbm % 8 == 0 && p.nbn > 5) ? 5 : 0;
    p.out_scale = 1.0f / (kOneAccActScale * ldexpf(1.f, L.w1_exp));
    p.ablate = g_spline_ablate;
    if (g_spline_wide_colgroup >= 0) p.col_group = (p.nbm % 8 == 0 && p.nbn > g_spline_wide_colgroup) ? g_spline_wide_colgroup : 0;
    static PerDeviceOnce slots_once;
    void (*kern)(const SplineWideParams) = spline_wide_kernel<0, 2, 3, 3, 0, 2, 3, 3, 0>;
    FC_DEV(if (g_spline_wide_dma == 1) kern = spline_wide_kernel<0, 2, 2, 2, 2, 3, 3, 2, 0>;
           else if (g_spline_wide_dma == 2) kern = spline_wide_kernel<0, 0, 3, 3, 2, 3, 3, 2, 0>;
           else if (g_spline_wide_dma == 3) kern = spline_wide_kernel<0, 1, 3, 3, 1, 2, 3, 3, 0>;
           else if (g_spline_wide_dma == 4) kern = spline_wide_kernel<0, 1, 2, 3, 2, 2, 3, 3, 0>;)
    {
        static PerDeviceOnce attr_once[5];
        attr_once[g_spline_wide_dma >= 0 && g_spline_wide_dma < 5 ? g_spline_wide_dma : 0].run(
            [&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS)); return 0; });
    }
    const int slots = slots_once.run([](int dev) {
        int cus = 0;
        FC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int n = cus & ~7;                                         // one workgroup per CU, a multiple of 8 (XCD order)
        return n < 8 ? 8 : n;
    });
    int grid = p.nbm * p.nbn;
    if (grid > slots) grid = slots;
    const double flops = 2.0 * (double)(e.rows_valid > 0 ? e.rows_valid : rows_alloc) * (double)(L.n_true ? L.n_true : L.N_pad) * (double)(L.k_true ? L.k_true : L.K_pad);
    ProfScope ps("void fc::spline_wide_kernel<0, 2, 3, 3, 0, 2, 3, 3, 0>(fc::SplineWideParams)", flops, 0.0, s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SW_LDS, s, p);
    FC_HIP(hipGetLastError());
}

}  // namespace fc

namespace fc {

// ---------------------------------------------------------------- EPI 1: a GELU Linear layer of the coupling MLP on the same tile
int g_linear_wide = 0;       // knob 29: 0 = off (SHIPPED: measured, not faster -- below), 1 = hidden layers of 512-wide coupling nets on this kernel for scenes of at
                             // least 2048 target points (the gate is the scene's size, never the batch's), 2 = at any size (tests)
// Measured on C2 (16 x 4096, same box, profiles/r04o_*): 133 us per hidden-layer launch = ~75 us of main loop (459 TF-eq at N = 512:
// profiles/micro/wide_gemm_probe.hip) + ~55 us of epilogue that nothing overlaps (4471 VALU instructions per wave and tile: 128 exact-erf
// GELUs, limb splits, lane swaps; with one workgroup per CU there is no second tile's main loop to hide them behind), and the in_layer stays
// on its fp32-A loop (100 us): 30.6 + 11.5 = 42 ms per step against 40.8 ms for the row-resident chain (mlprows.hip), whose epilogue runs
// in micro-steps under the next block's MFMAs.  The arithmetic also differs from the chain's (one accumulator, k32 MFMAs), so shipping it for
// large scenes only would make a row's log-prob depend on the size of the scene it sits in (tests/test_gpu_fullsize.py compares rows of a
// 512-point run with the 16 x 4096 run bit for bit).  Kept behind the knob with its tests; not on the default path.
int gemm_linear_wide_knob() { return g_linear_wide; }

bool linear_wide_eligible(const PackedLinear& L, const GemmEpi& e, int rows_alloc) {
    return L.W1 && !L.w1_permuted && L.bias1 && L.nseg == 1 && L.K_pad % 64 == 0 && L.N_pad % 256 == 0 && rows_alloc % 256 == 0 && e.A16 &&
           e.a16_scale == kOneAccActScale && e.C16 && !e.C && !e.Cpre && !e.gradu && !e.residual && !e.rowscal && e.act == FC_ACT_GELU &&
           (!e.residual16 || (e.r16_scale == kOneAccActScale && e.ldr16 == L.N_pad)) && (e.c16_scale == 0.f || e.c16_scale == kOneAccActScale);
}

void launch_linear_wide(const PackedLinear& L, const GemmEpi& e, int rows_alloc, hipStream_t s) {
    if (!linear_wide_eligible(L, e, rows_alloc)) throw Error(FC_ERR_INVALID, "launch_linear_wide: needs one-accumulator images in and out, GELU, N % 256 == 0, K % 64 == 0, rows % 256 == 0");
    int* flag = gemm_fp16_flag();
    if (!flag) throw Error(FC_ERR_INVALID, "launch_linear_wide: needs an open split-fp16 guard scope");
    SplineWideParams p{};
    p.A16 = e.A16; p.W1 = L.W1; p.bias1 = L.bias1;
    p.KT = L.K_pad / 32;
    p.nbm = rows_alloc / 256;
    p.nbn = L.N_pad / 256;
    p.ntile128 = L.N_pad / 128;
    p.col_group = (p.nbm % 8 == 0) ? p.nbn : 0;      // (a 512-wide layer: both column tiles of a row tile side by side on one XCD)
    p.out_scale = 1.0f / (kOneAccActScale * ldexpf(1.f, L.w1_exp));
    p.ablate = g_spline_ablate == 2 ? 2 : 0;
    p.out16 = e.C16; p.res16 = e.residual16; p.n16 = L.N_pad / 16;
    p.s1 = e.c16_scale > 0.f ? e.c16_scale : 1.0f;
    p.s2 = e.c16_scale > 0.f ? 1.0f : 2048.0f;
    p.ovf = flag;
    static PerDeviceOnce attr_once, slots_once;
    auto kern = spline_wide_kernel<1, 2, 3, 3, 0, 2, 3, 3, 0>;
    attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS)); return 0; });
    const int slots = slots_once.run([](int dev) {
        int cus = 0;
        FC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int n = cus & ~7;
        return n < 8 ? 8 : n;
    });
    int grid = p.nbm * p.nbn;
    if (grid > slots) grid = slots;
    const double flops = 2.0 * (double)(e.rows_valid > 0 ? e.rows_valid : rows_alloc) * (double)(L.n_true ? L.n_true : L.N_pad) * (double)(L.k_true ? L.k_true : L.K_pad);
    ProfScope ps("void fc::spline_wide_kernel<1, 2, 3, 3, 0, 2, 3, 3, 0>(fc::SplineWideParams)", flops, 0.0, s);
    static const bool trace = getenv("FC_FLAG_TRACE") != nullptr;
    int before = 0, after = 0;
    if (trace) { FC_HIP(hipStreamSynchronize(s)); FC_HIP(hipMemcpy(&before, flag, 4, hipMemcpyDeviceToHost)); }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SW_LDS, s, p);
    FC_HIP(hipGetLastError());
    if (trace) {
        FC_HIP(hipStreamSynchronize(s)); FC_HIP(hipMemcpy(&after, flag, 4, hipMemcpyDeviceToHost));
        fprintf(stderr, "[flag trace] linear_wide rows %d N %d res %d s1 %g: flag %d -> %d\n", rows_alloc, L.N_pad, e.residual16 != nullptr, p.s1, before, after);
    }
}

}  // namespace fc

namespace fc {

// ---------------------------------------------------------------- EPI 2: the one-accumulator limb product by itself (debug ABI, accuracy tests)
// out[rows, N] = x[rows, K] W[N, K]^T + bias on the 256 x 256 main loop: x and W are split into one-accumulator images here (x by
// kOneAccActScale, W by the power of two that puts max |w| into [2^14, 2^15) -- wmax given by the caller), rows / N padded to 256.
namespace { struct SwTmp { void* p = nullptr; explicit SwTmp(size_t b) { FC_HIP(hipMalloc(&p, b ? b : 4)); } ~SwTmp() { (void)hipFree(p); } float* f() const { return (float*)p; } }; }
void one_acc_gemm_debug(const float* x, const float* W, const float* bias, float wmax, float* out, int rows, int N, int K, hipStream_t s) {
    if (!x || !W || !out || rows < 1 || N < 1 || K < 64 || K % 64 != 0 || N % 256 != 0 || !(wmax < 65504.0f)) throw Error(FC_ERR_INVALID, "one_acc_gemm_debug: K % 64 == 0, N % 256 == 0");
    const int rp = round_up(rows, 256);
    int e = 0;
    if (wmax > 0.f) {
        while (ldexpf(wmax, e) >= 32768.0f) --e;
        while (ldexpf(wmax, e) < 16384.0f && e < 100) ++e;
    }
    SwTmp xa((size_t)rp * K * 4), wi((size_t)N * K * 4), b1((size_t)N * 4), cp((size_t)rp * N * 4), xp((size_t)rp * K * 4);
    launch_fill(xp.f(), 0.f, (size_t)rp * K, s);
    FC_HIP(hipMemcpyAsync(xp.f(), x, (size_t)rows * K * 4, hipMemcpyDeviceToDevice, s));
    const size_t nx = (size_t)rp * K, nw = (size_t)N * K;
    spline_wide_image_kernel<<<dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, s>>>(xp.f(), nullptr, rp, K, kOneAccActScale, 0.f, (unsigned short*)xa.p, nullptr, nx, 0);
    spline_wide_image_kernel<<<dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s>>>(W, bias, N, K, ldexpf(1.f, e), kOneAccActScale * ldexpf(1.f, e), (unsigned short*)wi.p, b1.f(), nw, 0);
    FC_HIP(hipGetLastError());
    SplineWideParams p{};
    p.A16 = (const unsigned short*)xa.p; p.W1 = (const unsigned short*)wi.p; p.bias1 = b1.f();
    p.KT = K / 32; p.nbm = rp / 256; p.nbn = N / 256; p.ntile128 = N / 128;
    p.col_group = (p.nbm % 8 == 0 && p.nbn > 5) ? 5 : 0;
    p.out_scale = 1.0f / (kOneAccActScale * ldexpf(1.f, e));
    p.C = cp.f(); p.ldc = N;
    static PerDeviceOnce attr_once;
    auto kern = spline_wide_kernel<2, 2, 3, 3, 0, 2, 3, 3, 0>;
    attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS)); return 0; });
    int cus = 0, dev = 0;
    FC_HIP(hipGetDevice(&dev));
    FC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int grid = p.nbm * p.nbn;
    const int slots = (cus & ~7) < 8 ? 8 : (cus & ~7);
    if (grid > slots) grid = slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SW_LDS, s, p);
    FC_HIP(hipGetLastError());
    FC_HIP(hipMemcpyAsync(out, cp.f(), (size_t)rows * N * 4, hipMemcpyDeviceToDevice, s));
    FC_HIP(hipStreamSynchronize(s));
}

}  // namespace fc

namespace fc {

// ---------------------------------------------------------------- EPI 3: the wide Linear layers of a training step (train.hip)
int g_train_wide = 1;

void launch_train_wide(const TrainWideArgs& a, hipStream_t s) {
    if (!a.A || !a.W1 || !a.C || a.K_pad < 64 || a.K_pad % 64 != 0 || a.rows_pad < 256 || a.rows_pad % 256 != 0 || a.n_cols < 4 || a.n_cols % 4 != 0 ||
        a.lda < a.K_pad || a.lda % 4 != 0 || a.ldc < a.n_cols || a.ldc % 4 != 0 || (a.gradu && (a.ldgu < a.n_cols || a.ldgu % 4 != 0)) ||
        (((uintptr_t)a.A | (uintptr_t)a.C | (uintptr_t)a.addend | (uintptr_t)a.gradu) & 15) || (a.row_absmax && a.bias1) || (a.gradu && a.gact != FC_ACT_GELU))
        throw Error(FC_ERR_INVALID, "launch_train_wide: K % 64 == 0, rows % 256 == 0, 16-byte aligned panels with pitches in multiples of 4, no bias beside per-row scales, GELU as the only activation gradient");
    SplineWideParams p{};
    p.A32 = a.A; p.lda = a.lda; p.W1 = a.W1; p.bias1 = a.bias1;
    p.KT = a.K_pad / 32;
    p.nbm = a.rows_pad / 256;
    p.nbn = (a.n_cols + 255) / 256;
    p.ntile128 = (a.n_cols + 127) / 128;
    p.col_group = p.nbm % 8 == 0 ? (p.nbn > 5 ? 5 : p.nbn) : 0;
    p.row_absmax = a.row_absmax;
    p.a_scale = a.a_scale;
    p.out_scale = a.row_absmax ? ldexpf(1.f, -kTrainWideWExp) : 1.0f / (a.a_scale * ldexpf(1.f, kTrainWideWExp));
    p.C = a.C; p.ldc = a.ldc; p.addend = a.addend; p.gradu = a.gradu; p.ldgu = a.ldgu; p.gact = a.gact; p.n_cols = a.n_cols;
    p.ovf = a.ovf;
    // knob 31 = 3: a C panel of 256 MB or more streams out with non-temporal stores (measured on the C2 training step, same box: 1116 against 1092 ms
    // with ordinary stores, profiles/r04w_train_*.json -- not shipped)
    p.nt_store = g_train_wide == 3 && (size_t)a.rows_pad * a.n_cols * 4 >= ((size_t)256 << 20);
    static PerDeviceOnce attr_once, slots_once;
    auto kern = spline_wide_kernel<3, 4, 4, 0, 0, 2, 3, 3, 0>;
    attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS)); return 0; });
    const int slots = slots_once.run([](int dev) {
        int cus = 0;
        FC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int n = cus & ~7;
        return n < 8 ? 8 : n;
    });
    int grid = p.nbm * p.nbn;
    if (grid > slots) grid = slots;
    ProfScope ps("void fc::spline_wide_kernel<3, 4, 4, 0, 0, 2, 3, 3, 0>(fc::SplineWideParams)", a.flops, 0.0, s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SW_LDS, s, p);
    FC_HIP(hipGetLastError());
}

}  // namespace fc
