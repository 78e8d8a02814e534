// One-accumulator limb scheme: accuracy and MFMA rate (round 3 probe for DESIGN.md section 12.5).
//   C[32 x 32] = A[32 x K] W[32 x K]^T, K = 512, on v_mfma_f32_32x32x16_f16 with fp32-equivalent operands in three forms:
//     (0) shipped: lo' = rn16((x - hi) 2048), hi.hi into `acc`, the two cross products into `corr`, acc + corr / 2048 at the end;
//     (1) one accumulator: lo = rn16(x - hi) UNSCALED (subnormal in fp16 for |x| < 2^-3), all three products into `acc`;
//     (2) as (1) with W pre-scaled by 2^s (s chosen so that max |w| lands in [8, 16)) and the result scaled back by 2^-s (exact).
//   Compared with a double-precision host product, for activations ~ N(0, 1) through a GELU-like sparsifier and weights ~ U(-a, a).
//   Second part: the rate of a dependent MFMA chain on normal against subnormal fp16 operands (cycles per MFMA by s_memtime).
//   hipcc --offload-arch=gfx950 -O2 one_acc_probe.hip -o one_acc_probe && ./one_acc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// one wave: C tile; A and W row-major [32][K]; mode as above
__global__ void gemm32(const float* A, const float* W, float* C, int K, int mode, float wscale) {
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    floatx16 acc = {0}, corr = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 ah, al, wh, wl;
        for (int e = 0; e < 8; ++e) {
            const float a = A[li * K + k0 + 8 * lh + e], w = W[li * K + k0 + 8 * lh + e] * wscale;
            const _Float16 ahh = (_Float16)a, whh = (_Float16)w;
            ah[e] = ahh; wh[e] = whh;
            if (mode == 0) { al[e] = (_Float16)((a - (float)ahh) * 2048.0f); wl[e] = (_Float16)((w - (float)whh) * 2048.0f); }
            else { al[e] = (_Float16)(a - (float)ahh); wl[e] = (_Float16)(w - (float)whh); }
        }
        // D[i][j] = sum_k Aop[i][k] Bop[k][j]: A operand = activations (row i = lane & 31), B operand = weights (column j = lane & 31)
        if (mode == 0) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc, 0, 0, 0);
            corr = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, corr, 0, 0, 0);
            corr = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, corr, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = mode == 0 ? acc[r] + corr[r] * (1.0f / 2048.0f) : acc[r];
        C[row * 32 + li] = v / wscale;
    }
}

__global__ void rate(unsigned long long* out, float aval, float bval, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
    floatx16 acc = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(acc[0] != 12345.f); }
}

int main() {
    const int K = 512;
    std::mt19937 g(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    float *dA, *dW, *dC; hipMalloc(&dA, 32 * K * 4); hipMalloc(&dW, 32 * K * 4); hipMalloc(&dC, 32 * 32 * 4);
    for (float wamp : {0.04f, 0.4f, 0.004f}) {
        std::vector<float> A(32 * K), W(32 * K), C(32 * 32);
        std::uniform_real_distribution<float> ud(-wamp, wamp);
        for (auto& v : A) { float x = nd(g); v = x > 0 ? x : 0.05f * x; }            // GELU-like: many small values
        for (auto& v : W) v = ud(g);
        std::vector<double> ref(32 * 32);
        double scale = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * W[j * K + k]; ref[i * 32 + j] = s; scale = fmax(scale, fabs(s)); }
        hipMemcpy(dA, A.data(), 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), 32 * K * 4, hipMemcpyHostToDevice);
        const float wmax = wamp; int s = 0; while (wmax * ldexpf(1.f, s) < 8.f) ++s;
        const struct { int mode; float ws; const char* name; } runs[] = {{0, 1.f, "two accumulators, lo' = lo * 2048 (shipped)"}, {1, 1.f, "one accumulator, unscaled lo"},
                                                                         {2, ldexpf(1.f, s), "one accumulator, W pre-scaled by a power of two"}};
        // fp32 reference error (sequential fmaf chain on the host) for scale
        double e32 = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float sacc = 0.f; for (int k = 0; k < K; ++k) sacc = fmaf(A[i * K + k], W[j * K + k], sacc); e32 = fmax(e32, fabs((double)sacc - ref[i * 32 + j])); }
        printf("weights U(-%g, %g), max |C| %.3f; host fp32 fmaf chain: max |err| %.2e\n", wamp, wamp, scale, e32);
        for (auto& r : runs) {
            hipLaunchKernelGGL(gemm32, dim3(1), dim3(64), 0, 0, dA, dW, dC, K, r.mode, r.ws);
            hipMemcpy(C.data(), dC, 32 * 32 * 4, hipMemcpyDeviceToHost);
            double emax = 0, esum = 0;
            for (int i = 0; i < 1024; ++i) { const double e = fabs((double)C[i] - ref[i]); emax = fmax(emax, e); esum += e; }
            printf("   %-55s max |err| %.2e  mean %.2e\n", r.name, emax, esum / 1024);
        }
    }
    unsigned long long* dT; hipMalloc(&dT, 16);
    for (float av : {1.0f, 9.5367431640625e-07f, 5.9604644775390625e-08f}) {
        hipLaunchKernelGGL(rate, dim3(1), dim3(64), 0, 0, dT, av, 1.0f, 2000);
        unsigned long long h[2]; hipMemcpy(h, dT, 16, hipMemcpyDeviceToHost);
        printf("dependent MFMA chain, A = %.3g (%s): %.1f clock ticks per MFMA\n", av, av < 6.1e-5f ? "subnormal" : "normal", (double)h[0] / 8000.0);
    }
    return 0;
}
