// Activation math shared by the GEMM epilogues (gemm.hip) and the fused pre-attention MLP kernel (premlp.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fcflow.h"

namespace fc {

// erf for the exact-GELU epilogue.  Own piecewise fit (profiles/micro/fit_erf.py): |x| <= 0.95: x * P5(x^2); 0.95 < |x| < 4:
// 1 - exp(P8(|x|)) (P8 fits log erfc); |x| >= 4: +-1.  Max abs error vs fp64 erf 1.5e-7 (about one fp32 ulp of the result),
// branch-light (both arms are short fma chains) where ocml's erff costs several times more VALU issue in the epilogue.
__device__ __forceinline__ float fc_erf(float x) {
    const float t = fabsf(x);
    const float s = x * x;
    float r = -0.0005881639663130045f;
    r = fmaf(r, s, 0.004971958696842194f);
    r = fmaf(r, s, -0.026752419769763947f);
    r = fmaf(r, s, 0.11281437426805496f);
    r = fmaf(r, s, -0.3761245906352997f);
    r = fmaf(r, s, 1.1283791065216064f);
    const float small = r * x;
    float q = 1.6150449937413214e-06f;
    q = fmaf(q, t, -4.561102105071768e-05f);
    q = fmaf(q, t, 0.0005929505568929017f);
    q = fmaf(q, t, -0.00474111782386899f);
    q = fmaf(q, t, 0.026367414742708206f);
    q = fmaf(q, t, -0.10998330265283585f);
    q = fmaf(q, t, -0.6319313645362854f);
    q = fmaf(q, t, -1.1301703453063965f);
    q = fmaf(q, t, 0.00030417676316574216f);
    float big = 1.0f - __expf(q);
    big = t >= 4.0f ? 1.0f : big;
    big = copysignf(big, x);
    return t <= 0.95f ? small : big;
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case FC_ACT_GELU: return 0.5f * v * (1.0f + fc_erf(v * 0.70710678118654752440f));
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
