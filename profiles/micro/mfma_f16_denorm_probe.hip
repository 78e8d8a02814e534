// Does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs?  (Round 3 question for a one-accumulator limb scheme: with an unscaled low
// limb lo = rn16(x - hi) the cross products land at their true scale and could share the main accumulator -- no second accumulator set, no
// fold -- but lo then lives in fp16's subnormal range for |x| < 0.125.)  A = 2^-20 (subnormal), B = 1: a row sum of 16 products is 16 * 2^-20
// if subnormals are honoured, 0 if they are flushed.  Also probes 2^-24 (the smallest subnormal) and a mixed sum.
//   hipcc --offload-arch=gfx950 -O2 mfma_f16_denorm_probe.hip -o mfma_f16_denorm_probe && ./mfma_f16_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void probe(float* out, float aval, float bval) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
    floatx16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float vals[][2] = {{ldexpf(1.f, -20), 1.f}, {ldexpf(1.f, -24), 1.f}, {ldexpf(1.f, -14), 1.f}, {ldexpf(1.f, -20), ldexpf(1.f, -3)}, {ldexpf(3.f, -24), 2.f}};
    for (auto& v : vals) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, v[0], v[1]);
        float h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a = %.10g (fp16 %s), b = %g: sum of 16 products = %.10g, expected %.10g -> %s\n", v[0], v[0] < 6.1e-5f ? "subnormal" : "normal", v[1], h, 16.0 * v[0] * v[1],
               h == 16.0f * v[0] * v[1] ? "honoured" : (h == 0.f ? "FLUSHED" : "other"));
    }
    return 0;
}
