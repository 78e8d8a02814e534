// Activation math shared by the GEMM epilogues (gemm.hip) and the fused pre-attention MLP kernel (premlp.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fcflow.h"

namespace fc {

// Exact (erf) GELU in one branch-free chain: gelu(v) = v Phi(v), Phi(-|v|) = erfc(|v| / sqrt 2) / 2 = exp2(g(|v|)) with
// g(w) = log2 erfcx(w / sqrt 2) - (w^2 / 2) log2 e - 1: erfcx is smooth and slowly varying, so g is a parabola plus a small smooth term and is fitted
// by ONE degree-11 polynomial in w = min(|v|, 7.354) (Chebyshev fit, profiles/micro/fit_gelu.py).  15 VALU instructions: min, 11 fma, exp2, max, fma
// (round 4; the two-branch erf fit of round 1 took 28, round 3's form of this chain 20: the exponent's - u^2 log2 e and the 1/2 now ride in the
// polynomial, the argument is |v| itself instead of |v| / sqrt 2, and v - v e is one fma) -- the GEMM epilogues and the row-resident chains are
// VALU-bound (PMC: 7-12 VALU instructions per MFMA).
// fp32 accuracy against fp64: |error| <= 2.4e-7 (half an ulp of v at |v| ~ 5), 1.0e-7 relative to max(1, |v|), and 3.3e-6 RELATIVE
// in the negative tail, where the two-branch form lost all relative accuracy (1 - (1 - e^q)).
__device__ __forceinline__ float fc_gelu(float v) {
    const float w = fminf(fabsf(v), 7.353910524340095f);
    float g = 7.953413483363647e-10f;
    g = fmaf(g, w, -3.609737220244824e-08f);
    g = fmaf(g, w, 7.156736501201522e-07f);
    g = fmaf(g, w, -8.034509846766014e-06f);
    g = fmaf(g, w, 5.399647488957271e-05f);
    g = fmaf(g, w, -0.0001862359931692481f);
    g = fmaf(g, w, -0.0002285758382640779f);
    g = fmaf(g, w, 0.00727660721167922f);
    g = fmaf(g, w, -0.05270823836326599f);
    g = fmaf(g, w, -0.4591129422187805f);
    g = fmaf(g, w, -1.151120901107788f);
    g = fmaf(g, w, -0.9999995827674866f);
    const float e = __builtin_amdgcn_exp2f(g);                                        // Phi(-|v|)
    return fmaf(-fabsf(v), e, fmaxf(v, 0.f));                                         // = v > 0 ? v - v e : v e, without compare + select
}

// Two values -> their packed fp16 limb words: hi = [rn16(x0) | rn16(x1) << 16], lo = [rn16((x0 - hi0) * 2048) |
// rn16((x1 - hi1) * 2048) << 16].  Five VALU instructions for the pair -- v_cvt_pk_f16_f32, two v_fma_mix_f32 that subtract the fp16 halves
// from the fp32 inputs (exact), v_fma_mixlo / mixhi_f16 for the scaled remainders -- where the scalar formulation compiles to ten (convert,
// convert back, subtract, multiply-convert, pack, per value).  Same roundings, same bits.  The limb splits sit in VALU-bound epilogues
// (attention's P, the row-resident MLP kernels): round 3.
__device__ __forceinline__ void limb_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    float d0, d1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hi), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(lo) : "v"(d0), "s"(2048.0f));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(lo) : "v"(d1), "s"(2048.0f));
}
// One-accumulator form (common.h): hi = rn16(x), lo = rn16(x - hi) UNSCALED (the caller has scaled x so that lo stays a normal number where
// it matters).  Three VALU instructions for the pair: x - hi is exact in fp32 (hi is x's leading 11 bits), so v_fma_mixlo / mixhi_f16 round it
// to fp16 once, straight into the two halves of lo -- the bits of subtracting in fp32 and converting the pair (four instructions until round 4).
__device__ __forceinline__ void limb_split2u(float x0, float x1, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x1));
}
// The same with run-time (wave-uniform) scales: hi = rn16(x s1), lo = rn16((x s1 - hi) s2).  (1, 2048) gives limb_split2's bits exactly (the
// products by 1 are exact); (kOneAccActScale, 1) the one-accumulator form of spline_wide.hip (common.h).  Six VALU instructions for the pair.
__device__ __forceinline__ void limb_split2s(float x0, float x1, float s1, float s2, unsigned& hi, unsigned& lo) {
    float d0, d1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(x0), "s"(s1));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x1), "s"(s1));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(d0) : "v"(x0), "s"(s1), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d1) : "v"(x1), "s"(s1), "v"(hi));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(lo) : "v"(d0), "s"(s2));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(lo) : "v"(d1), "s"(s2));
}
// element SEL (0 = low half, 1 = high half) of packed limb words back to fp32, x = hi + lo'/2048, in one v_fma_mix_f32 (the scalar
// formulation is two converts and a multiply-add); the product is exact, the sum rounds once either way: same bits
template <int SEL>
__device__ __forceinline__ float limb_join(unsigned hi, unsigned lo) {
    float r;
    if constexpr (SEL == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(lo), "s"(1.0f / 2048.0f), "v"(hi));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(lo), "s"(1.0f / 2048.0f), "v"(hi));
    return r;
}
// eight values -> one MFMA operand fragment per limb
typedef _Float16 fc_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned fc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void limb_split8(const float (&x)[8], fc_f16x8& hi, fc_f16x8& lo) {
    fc_u32x4 h, l;
    unsigned a, b;
    limb_split2(x[0], x[1], a, b); h[0] = a; l[0] = b;
    limb_split2(x[2], x[3], a, b); h[1] = a; l[1] = b;
    limb_split2(x[4], x[5], a, b); h[2] = a; l[2] = b;
    limb_split2(x[6], x[7], a, b); h[3] = a; l[3] = b;
    hi = __builtin_bit_cast(fc_f16x8, h);
    lo = __builtin_bit_cast(fc_f16x8, l);
}
__device__ __forceinline__ void limb_split8_unscaled(const float (&x)[8], fc_f16x8& hi, fc_f16x8& lo) {
    fc_u32x4 h, l;
    unsigned a, b;
    limb_split2u(x[0], x[1], a, b); h[0] = a; l[0] = b;
    limb_split2u(x[2], x[3], a, b); h[1] = a; l[1] = b;
    limb_split2u(x[4], x[5], a, b); h[2] = a; l[2] = b;
    limb_split2u(x[6], x[7], a, b); h[3] = a; l[3] = b;
    hi = __builtin_bit_cast(fc_f16x8, h);
    lo = __builtin_bit_cast(fc_f16x8, l);
}
// and back: one fragment per limb -> eight values
__device__ __forceinline__ void limb_join8(const fc_f16x8& hi, const fc_f16x8& lo, float (&x)[8]) {
    const fc_u32x4 h = __builtin_bit_cast(fc_u32x4, hi), l = __builtin_bit_cast(fc_u32x4, lo);
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[2 * i] = limb_join<0>(h[i], l[i]); x[2 * i + 1] = limb_join<1>(h[i], l[i]); }
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case FC_ACT_GELU: return fc_gelu(v);
        case FC_ACT_RELU: return v > 0.f ? v : 0.f;
        case FC_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case FC_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
        default: return v;
    }
}

// d/dv [v Phi(v)] = Phi(v) + v phi(v) on fc_gelu's machinery: Phi(-|v|) = erfc(u) / 2 from the same fit, phi(v) = exp2(-u^2 log2 e) / sqrt(2 pi)
// with the same (clamped) u.  ~24 VALU instructions where erfcf + expf compile to ~60 (the epilogue of the training data-gradient GEMM evaluates
// it for every output value); |error| <= 2.3e-7 against fp64 on [-9, 9].
__device__ __forceinline__ float fc_gelu_grad(float v) {
    const float u = fminf(fabsf(v) * 0.70710678118654752440f, 5.2f);
    float g = 3.599303965984291e-08f;
    g = fmaf(g, u, -1.1551159104783437e-06f);
    g = fmaf(g, u, 1.6193846022360958e-05f);
    g = fmaf(g, u, -0.00012855215754825622f);
    g = fmaf(g, u, 0.0006109004025347531f);
    g = fmaf(g, u, -0.0014898879453539848f);
    g = fmaf(g, u, -0.00129302020650357f);
    g = fmaf(g, u, 0.02910642884671688f);
    g = fmaf(g, u, -0.14908140897750854f);
    g = fmaf(g, u, 0.5244691371917725f);
    g = fmaf(g, u, -1.627930760383606f);
    g = fmaf(g, u, 4.18458824924528e-07f - 1.0f);
    const float t = -1.4426950408889634f * u * u;
    const float e = __builtin_amdgcn_exp2f(t + g);                                    // Phi(-|v|)
    const float p = __builtin_amdgcn_exp2f(t);                                        // exp(-v^2 / 2)
    const float cdf = 0.5f + copysignf(0.5f - e, v);                                  // Phi(v)
    return fmaf(v, 0.39894228040143267794f * p, cdf);
}

// d act / du (training: fc_train_act_bwd_f32 and the activation-gradient epilogue of the data-gradient GEMM); GELU = exact erf form
__device__ __forceinline__ float fc_act_grad(float u, int act) {
    switch (act) {
        case FC_ACT_GELU: return fc_gelu_grad(u);
        case FC_ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case FC_ACT_ELU: return u > 0.f ? 1.f : expf(u);
        case FC_ACT_LRELU02: return u > 0.f ? 1.f : 0.2f;
        default: return 1.f;
    }
}

__device__ __forceinline__ float half_wave_sum(float v) {
    // sum over the 32 lanes that share (lane>>5): xor masks < 32 never cross the half
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    return v;
}

}  // namespace fc
