// Issue rate of the conversion instructions the limb split uses, on one wave per SIMD (gfx950): cycles per instruction from s_memtime
// around an unrolled loop of 8 independent chains.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters) {
    float a[8], b[8];
    unsigned h[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.0f + i * 0.01f; h[i] = i; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define OPX(i)                                                                                                   \
        if constexpr (OP == 0) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h[i]) : "v"(a[i]), "v"(b[i]));                  \
        else if constexpr (OP == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]), "v"(h[i])); \
        else if constexpr (OP == 2) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h[i]) : "v"(a[i]));               \
        else if constexpr (OP == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(b[i]));   \
        else if constexpr (OP == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[i]), "v"(b[i]));  \
        else if constexpr (OP == 5) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(h[i]));               \
        else if constexpr (OP == 6) asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));                   \
        else if constexpr (OP == 7) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[i & 6]), "v"(*(double*)&b[i & 6])); \
        else if constexpr (OP == 8) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7])); \
        else if constexpr (OP == 9) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(b[i]));
        REP8(OPX) REP8(OPX) REP8(OPX) REP8(OPX)
#undef OPX
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[OP] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 16 * 8); hipMemset(cyc, 0, 16 * 8);
    const int iters = 2000;
    const char* names[] = {"v_fma_mixlo_f16", "v_fma_mix_f32", "v_cvt_f16_f32", "v_cvt_pk_f16_f32", "v_fma_f32", "v_cvt_f32_f16", "v_exp_f32", "v_pk_mul_f32", "v_max3_f32 |a| |b|", "v_cvt_pkrtz_f16_f32"};
    for (int waves = 1; waves <= 2; ++waves) {       // waves per SIMD: 256 threads = 1 per SIMD, 512 = 2 per SIMD
#define RUN(OP) hipLaunchKernelGGL(probe<OP>, dim3(1), dim3(256 * waves > 256 ? 256 : 256), 0, 0, out, cyc, iters);
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
        // s_memtime counts at 100 MHz: convert with the shader clock the run had is not possible here; report memtime ticks per 1000 instructions
        for (int i = 0; i < 10; ++i) printf("%-22s %8.3f memtime ticks per 1000 instr (one wave per SIMD)\n", names[i], (double)h[i] / (iters * 32) * 1000.0);
        break;
    }
    return 0;
}
