// Probe of ds_read_b64_tr_b16 (gfx950): which LDS element lands in which lane/element for the address pattern attention.hip uses.
// LDS image: 8 rows x 64 columns of 16-bit values row*100 + col, row pitch 160 bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4i16 __attribute__((__vector_size__(4 * sizeof(short))));
constexpr int PITCH = 160;
__global__ void probe(short* out) {
    __shared__ __attribute__((aligned(16))) char lds[8 * PITCH];
    for (int i = threadIdx.x; i < 8 * 64; i += 64) *reinterpret_cast<short*>(lds + (i / 64) * PITCH + (i % 64) * 2) = (short)((i / 64) * 100 + (i % 64));
    __syncthreads();
    const int lane = threadIdx.x, lh = lane >> 5;
    const int off = (4 * lh + ((lane & 15) >> 2)) * PITCH + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(lds + off));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    return 0;
}
