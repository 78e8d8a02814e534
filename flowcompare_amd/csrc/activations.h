// Activation math shared by the GEMM epilogues (gemm.hip) and the fused pre-attention MLP kernel (premlp.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fcflow.h"

namespace fc {

// Exact (erf) GELU in one branch-free chain: gelu(v) = v Phi(v), Phi(-|v|) = erfc(u)/2 with u = |v|/sqrt(2), and
// log2 erfc(u) = -u^2 log2(e) + log2 erfcx(u), where log2 erfcx is smooth and slowly varying (0 ... -3.3 on [0, 5.2]) and is fitted
// by a degree-11 polynomial (Chebyshev fit, profiles/micro/fit_gelu.py).  22 VALU instructions instead of the 28 of the two-branch erf fit it replaced -- the GEMM epilogues and the fused pre-attention kernel are VALU-bound (PMC: 8-12 VALU instructions per MFMA).
// fp32 accuracy against fp64: |error| <= 2.4e-7 (half an ulp of v at |v| ~ 5), 9.8e-8 relative to max(1, |v|), and 5e-6 RELATIVE
// in the negative tail, where the two-branch form lost all relative accuracy (1 - (1 - e^q)).
__device__ __forceinline__ float fc_gelu(float v) {
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
    g = fmaf(g, u, 4.18458824924528e-07f);
    const float e = __builtin_amdgcn_exp2f(fmaf(-1.4426950408889634f * u, u, g));      // erfc(u)
    const float h = (0.5f * v) * e;                                                   // v Phi(-|v|), signed like v
    return v > 0.f ? v - h : h;
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
