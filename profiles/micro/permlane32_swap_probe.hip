#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) {
    float a = threadIdx.x, b = 100.f + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    o[threadIdx.x] = __builtin_bit_cast(float, r[0]);
    o[64 + threadIdx.x] = __builtin_bit_cast(float, r[1]);
}
int main() {
    float* d; hipMalloc(&d, 128 * 4);
    k<<<1, 64>>>(d);
    float h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("r0: lane0 %g lane31 %g lane32 %g lane63 %g\n", h[0], h[31], h[32], h[63]);
    printf("r1: lane0 %g lane31 %g lane32 %g lane63 %g\n", h[64], h[95], h[96], h[127]);
}
