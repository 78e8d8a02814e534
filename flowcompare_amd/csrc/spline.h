// Rational-quadratic spline element (shared by the stand-alone spline kernels in misc.hip and the fused GEMM epilogue in gemm.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace fc {

// ---------------------------------------------------------------- rational-quadratic spline
// One element: models/spline_coupling.py:24-66 (tails) + :69-169 (spline) + :17-19 (searchsorted).
// u points at 3K+1 parameters [K widths | K heights | K+1 derivative logits] with element stride `us`.
// Quirks reproduced: derivative logits are padded left with log(exp(1-min_d-1)); knot i>=1 uses ud[i-1];
// ud[K] is never used; last knot + 1e-6 only for the bin search; outside [-3,3] identity with logabsdet 0.
// v_exp_f32 / v_log_f32 / v_rcp_f32 based helpers (about 1 ulp on the base-2 function): the spline evaluates 16 exponentials,
// 2 softplus, 2 logs and ~10 divisions per element, which made the ocml versions the kernel's dominant VALU cost.
__device__ __forceinline__ float fast_exp(float v) { return __builtin_amdgcn_exp2f(v * 1.4426950408889634f); }
__device__ __forceinline__ float fast_log(float v) { return __builtin_amdgcn_logf(v) * 0.6931471805599453f; }
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

template <int K>
__device__ __forceinline__ void rq_spline_elem(float x, const float* u, int us, bool inverse, float& y, float& lad) {
    constexpr float B = 3.0f, MINW = 1e-3f, MINH = 1e-3f, MIND = 1e-3f;
    if (!(x >= -B && x <= B)) { y = x; lad = 0.f; return; }
    float cw[K + 1], ch[K + 1];
    {
        float e[K], mx = u[0];
#pragma unroll
        for (int i = 0; i < K; ++i) { e[i] = u[i * us]; mx = fmaxf(mx, e[i]); }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) { e[i] = fast_exp(e[i] - mx); sum += e[i]; }
        float c = 0.f;
        const float rs = __builtin_amdgcn_rcpf(sum);
        cw[0] = -B;
#pragma unroll
        for (int i = 0; i < K; ++i) { c += MINW + (1.0f - MINW * K) * (e[i] * rs); cw[i + 1] = 2.0f * B * c - B; }
        cw[K] = B;
    }
    {
        float e[K], mx = u[K * us];
#pragma unroll
        for (int i = 0; i < K; ++i) { e[i] = u[(K + i) * us]; mx = fmaxf(mx, e[i]); }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) { e[i] = fast_exp(e[i] - mx); sum += e[i]; }
        float c = 0.f;
        const float rs = __builtin_amdgcn_rcpf(sum);
        ch[0] = -B;
#pragma unroll
        for (int i = 0; i < K; ++i) { c += MINH + (1.0f - MINH * K) * (e[i] * rs); ch[i + 1] = 2.0f * B * c - B; }
        ch[K] = B;
    }
    // bin = #{knots <= x} - 1 over the searched knots (last one + 1e-6)
    int bin = 0;
#pragma unroll
    for (int i = 1; i <= K; ++i) {
        const float knot = (inverse ? ch[i] : cw[i]) + (i == K ? 1e-6f : 0.f);
        bin += (x >= knot) ? 1 : 0;
    }
    float in_cw = cw[0], in_w = cw[1] - cw[0], in_ch = ch[0], in_h = ch[1] - ch[0];
    float ud0 = 0.f, ud1 = u[(2 * K) * us];
#pragma unroll
    for (int i = 1; i < K; ++i) {
        if (bin == i) {
            in_cw = cw[i]; in_w = cw[i + 1] - cw[i]; in_ch = ch[i]; in_h = ch[i + 1] - ch[i];
            ud0 = u[(2 * K + i - 1) * us]; ud1 = u[(2 * K + i) * us];
        }
    }
    const float cst = -1e-3f;                                   // log(exp(1 - min_derivative - 1))
    const float raw0 = bin == 0 ? cst : ud0;
    auto softplus = [](float v) { return v > 20.f ? v : fast_log(1.0f + fast_exp(v)); };
    const float d0 = MIND + softplus(raw0), d1 = MIND + softplus(ud1);
    const float rw = __builtin_amdgcn_rcpf(in_w);
    const float delta = in_h * rw;
    if (!inverse) {
        const float th = (x - in_cw) * rw;
        const float tt = th * (1.0f - th);
        const float num = in_h * (delta * th * th + d0 * tt);
        const float den = delta + (d0 + d1 - 2.0f * delta) * tt;
        y = in_ch + fast_div(num, den);
        const float omt = 1.0f - th;
        const float dnum = delta * delta * (d1 * th * th + 2.0f * delta * tt + d0 * omt * omt);
        lad = fast_log(dnum) - 2.0f * fast_log(den);
    } else {
        const float dy = x - in_ch;
        const float t3 = d0 + d1 - 2.0f * delta;
        const float qa = dy * t3 + in_h * (delta - d0);
        const float qb = in_h * d0 - dy * t3;
        const float qc = -delta * dy;
        const float disc = qb * qb - 4.0f * qa * qc;
        const float root = fast_div(2.0f * qc, -qb - sqrtf(disc));
        y = root * in_w + in_cw;
        const float tt = root * (1.0f - root);
        const float den = delta + t3 * tt;
        const float omr = 1.0f - root;
        const float dnum = delta * delta * (d1 * root * root + 2.0f * delta * tt + d0 * omr * omr);
        lad = -(fast_log(dnum) - 2.0f * fast_log(den));
    }
}

// Forward direction only, BRANCH-FREE (the fused GEMM epilogue evaluates 2-3 independent elements per thread as straight-line code, so
// the compiler can interleave their dependency chains; the general routine above compiles to ~14 divergent branches per element for
// its bin selection).  Same algorithm and quirks; differences are rounding-level only: exp via one fma + v_exp (log2 e folded into the
// argument), the softmax scale folded into the prefix sums, bin edges picked by running selects instead of knot arrays, the two
// derivative logits by an indexed read.  u must be readable at every index 0 .. 3K (it is: the LDS parameter tile).
template <int K>
__device__ __forceinline__ void rq_spline_fwd(float x, const float* u, int us, float& y, float& lad) {
    constexpr float B = 3.0f, MINW = 1e-3f, MINH = 1e-3f, MIND = 1e-3f, L2E = 1.4426950408889634f;
    const bool inside = x >= -B && x <= B;
    float ew[K], eh[K], mw = u[0], mh = u[K * us];
#pragma unroll
    for (int i = 0; i < K; ++i) { ew[i] = u[i * us]; eh[i] = u[(K + i) * us]; mw = fmaxf(mw, ew[i]); mh = fmaxf(mh, eh[i]); }
    float sw = 0.f, sh = 0.f;
    const float ow = -mw * L2E, oh = -mh * L2E;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        ew[i] = __builtin_amdgcn_exp2f(fmaf(ew[i], L2E, ow)); sw += ew[i];
        eh[i] = __builtin_amdgcn_exp2f(fmaf(eh[i], L2E, oh)); sh += eh[i];
    }
    const float fw = (1.0f - MINW * K) * __builtin_amdgcn_rcpf(sw), fh = (1.0f - MINH * K) * __builtin_amdgcn_rcpf(sh);
    // widths: bin = #{knots <= x} (last knot + 1e-6), in_cw = largest knot <= x, hi = smallest knot > x
    float c = 0.f, in_cw = -B, hi = INFINITY;
    int bin = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        c += fmaf(fw, ew[i], MINW);
        const float knot = i == K - 1 ? B : fmaf(2.0f * B, c, -B);
        const bool ge = x >= (i == K - 1 ? knot + 1e-6f : knot);
        bin += ge ? 1 : 0;
        in_cw = ge ? knot : in_cw;
        hi = ge ? hi : fminf(hi, knot);
    }
    const float in_w = hi - in_cw;
    // heights: knot index bin (lower edge) and bin + 1 (upper edge) of the cumulative heights
    float ch = 0.f, in_ch = -B, ch_hi = B;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        ch += fmaf(fh, eh[i], MINH);
        const float knot = i == K - 1 ? B : fmaf(2.0f * B, ch, -B);
        in_ch = (i + 1 == bin) ? knot : in_ch;
        ch_hi = (i == bin) ? knot : ch_hi;
    }
    const float in_h = ch_hi - in_ch;
    const int b0 = bin > 0 ? bin - 1 : 0;
    const float ud0 = u[(2 * K + b0) * us], ud1 = u[(2 * K + bin) * us];
    const float raw0 = bin == 0 ? -1e-3f : ud0;                  // log(exp(1 - min_derivative - 1))
    const float d0 = MIND + (raw0 > 20.f ? raw0 : fast_log(1.0f + fast_exp(raw0)));
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

// The same forward evaluation on parameters held in REGISTERS: every index is a compile-time constant after unrolling (the two
// derivative logits are picked by running selects along the bin search instead of an indexed read), so a caller can hand over
// accumulator registers.  Same operations in the same order as rq_spline_fwd: results are bit-identical.
template <int K, class U>
__device__ __forceinline__ void rq_spline_fwd_regs(float x, const U& u, float& y, float& lad) {
    constexpr float B = 3.0f, MINW = 1e-3f, MINH = 1e-3f, MIND = 1e-3f, L2E = 1.4426950408889634f;
    const bool inside = x >= -B && x <= B;
    float ew[K], eh[K], mw = u(0), mh = u(K);
#pragma unroll
    for (int i = 0; i < K; ++i) { ew[i] = u(i); eh[i] = u(K + i); mw = fmaxf(mw, ew[i]); mh = fmaxf(mh, eh[i]); }
    float sw = 0.f, sh = 0.f;
    const float ow = -mw * L2E, oh = -mh * L2E;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        ew[i] = __builtin_amdgcn_exp2f(fmaf(ew[i], L2E, ow)); sw += ew[i];
        eh[i] = __builtin_amdgcn_exp2f(fmaf(eh[i], L2E, oh)); sh += eh[i];
    }
    const float fw = (1.0f - MINW * K) * __builtin_amdgcn_rcpf(sw), fh = (1.0f - MINH * K) * __builtin_amdgcn_rcpf(sh);
    float c = 0.f, in_cw = -B, hi = INFINITY;
    float ud0 = -1e-3f, ud1 = u(2 * K);                          // bin 0: left pad log(exp(1 - min_derivative - 1)) and logit 0
    int bin = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        c += fmaf(fw, ew[i], MINW);
        const float knot = i == K - 1 ? B : fmaf(2.0f * B, c, -B);
        const bool ge = x >= (i == K - 1 ? knot + 1e-6f : knot);
        bin += ge ? 1 : 0;
        in_cw = ge ? knot : in_cw;
        hi = ge ? hi : fminf(hi, knot);
        ud0 = ge ? u(2 * K + i) : ud0;                           // knots increase: the last `ge` is i = bin - 1
        ud1 = ge ? u(2 * K + i + 1) : ud1;
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

template <int K>
__device__ __forceinline__ void rq_dispatch(float x, const float* u, int us, bool inv, float& y, float& lad) {
    rq_spline_elem<K>(x, u, us, inv, y, lad);
}

__device__ __forceinline__ void rq_any(int K, float x, const float* u, int us, bool inv, float& y, float& lad) {
    switch (K) {
        case 4: rq_dispatch<4>(x, u, us, inv, y, lad); break;
        case 8: rq_dispatch<8>(x, u, us, inv, y, lad); break;
        case 16: rq_dispatch<16>(x, u, us, inv, y, lad); break;
        default: y = x; lad = 0.f; break;     // rejected on the host
    }
}

// Column layout of the spline parameter layer's output ("tile-grouped"): the 3K+1 parameters of DPT = 128 / (3K+1) transformed dims
// share one 128-column GEMM tile (K = 8: 5 dims, 125 of 128 columns used), so the workgroup that produced a 128x128 tile holds every
// parameter of DPT dims of its 128 rows and can evaluate the splines in its epilogue.
//   K = 4, 16: dim-major inside the tile, column(j, p) = (j / DPT) * 128 + (j % DPT) * (3K+1) + p.
//   K = 8: REGISTER-SLOT order of the transposed 32x32 MFMA product (parameters on the accumulator's row index, points on its lanes;
//     gemm.hip VAR 10).  A lane of half h = lane >> 5 owns, for its point, accumulator slot s = 16 * (block of 32 columns) + register
//     v, which is tile column  c(s, h) = (s / 16) * 32 + ((s % 16) / 4) * 8 + 4 h + s % 4.  Dims 0, 1 of the tile live in slots 0..49 of
//     half 0, dims 2, 3 in slots 0..49 of half 1, dim 4 in slots 50..63 of half 0 (parameters 0..13) and slots 50..60 of half 1
//     (parameters 14..24); slots 61..63 of half 1 (columns 125..127) are unused.  A lane pair therefore holds all 25 parameters of
//     each of its 5 dims in registers with compile-time indices, and 11 v_permlane32_swap move dim 4's upper part across.
__host__ __device__ inline int spline_dpt(int K) { return 128 / (3 * K + 1); }
__host__ __device__ inline int spline_slot_col(int s, int h) { return (s / 16) * 32 + ((s % 16) / 4) * 8 + 4 * h + s % 4; }
__host__ __device__ inline int spline_col(int j, int pp, int K) {
    const int dpt = spline_dpt(K), tile = j / dpt, dl = j % dpt;
    if (K != 8) return tile * 128 + dl * (3 * K + 1) + pp;
    if (dl < 4) return tile * 128 + spline_slot_col((dl & 1) * 25 + pp, dl >> 1);
    return tile * 128 + (pp < 14 ? spline_slot_col(50 + pp, 0) : spline_slot_col(50 + pp - 14, 1));
}
// inverse of the above inside one tile: column c -> dim-major position (dim in tile) * (3K+1) + parameter (the LDS-tile epilogues of the
// other GEMM variants store their accumulators there, so their readers see the parameters of a dim contiguously)
__host__ __device__ inline int spline_tile_pos(int c, int K) {
    if (K != 8) return c;
    const int h = (c >> 2) & 1, s = (c >> 5) * 16 + ((c >> 3) & 3) * 4 + (c & 3);
    if (s < 50) return (2 * h + s / 25) * 25 + s % 25;
    return h == 0 ? 100 + (s - 50) : 100 + 14 + (s - 50);          // h = 1, s = 61..63 -> 125..127 (unused)
}
__host__ __device__ inline int spline_ncols(int d2, int K) { return ((d2 + spline_dpt(K) - 1) / spline_dpt(K)) * 128; }

}  // namespace fc
