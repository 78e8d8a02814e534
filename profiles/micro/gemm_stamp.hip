// Diagnostic build of the 128x128x32 fp32-MFMA GEMM main loop with s_memtime stamps (wave 0 of every workgroup):
// where do the cycles of one k-tile go?  Stamps are written to their own buffer; no output value depends on them.
//   hipcc --offload-arch=gfx950 -O3 gemm_stamp.hip -o gemm_stamp && ./gemm_stamp [rows] [K] [blocks_per_cu_cap]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int LDS_LD = 36, BM = 128, BN = 128, TM = 2, TN = 2, STAGE = (BM + BN) * LDS_LD;

template <int MODE, int LAYOUT>   // MODE 0: full loop; 1: no global loads in the loop; 2: no LDS fragment reads; 3: only A loaded; 4: only W loaded.  LAYOUT 1: k-tile-major panels
__global__ __launch_bounds__(256) void k(const float* A, const float* W, float* C, int K, int nbn, unsigned long long* stamps) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn, m0 = bm * BM, n0 = bn * BN;
    floatx16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
    float4 ra[4], rb[4];
    auto gload = [&](int kt) {
        const size_t sa = LAYOUT ? 32 : K, rowsA = (size_t)gridDim.x / nbn * BM, rowsW = (size_t)nbn * BN;
        const float* a = LAYOUT ? A + ((size_t)kt * rowsA + m0 + lrow) * 32 + lc4 : A + (size_t)(m0 + lrow) * K + kt * 32 + lc4;
        const float* w = LAYOUT ? W + ((size_t)kt * rowsW + n0 + lrow) * 32 + lc4 : W + (size_t)(n0 + lrow) * K + kt * 32 + lc4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE != 4) ra[i] = *reinterpret_cast<const float4*>(a + (size_t)(32 * i) * sa);
            if (MODE != 3) rb[i] = *reinterpret_cast<const float4*>(w + (size_t)(32 * i) * sa);
        }
    };
    auto lstore = [&](int st) {
        float* sA = smem + st * STAGE; float* sB = sA + BM * LDS_LD;
#pragma unroll
        for (int i = 0; i < 4; ++i) { *reinterpret_cast<float4*>(sA + (lrow + 32 * i) * LDS_LD + lc4) = ra[i]; *reinterpret_cast<float4*>(sB + (lrow + 32 * i) * LDS_LD + lc4) = rb[i]; }
    };
    gload(0); lstore(0); __syncthreads();
    const int KT = K / 32;
    unsigned long long t_mfma = 0, t_wait = 0, t_store = 0, t_bar = 0, t_all = 0;
    const unsigned long long tb = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < KT; ++kt) {
        const unsigned long long s0 = __builtin_amdgcn_s_memtime();
        const bool more = kt + 1 < KT;
        if (MODE != 1 && more) gload(kt + 1);
        const float* sA = smem + (kt & 1) * STAGE + (wr * 64 + li) * LDS_LD + 4 * lh;
        const float* sB = smem + (kt & 1) * STAGE + BM * LDS_LD + (wc * 64 + li) * LDS_LD + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = MODE == 2 ? make_float4(1.f, 2.f, 3.f, lane) : *reinterpret_cast<const float4*>(sA + i * 32 * LDS_LD + 8 * g);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = MODE == 2 ? make_float4(1.f, 2.f, 3.f, lane) : *reinterpret_cast<const float4*>(sB + j * 32 * LDS_LD + 8 * g);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        const unsigned long long s1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long s2 = __builtin_amdgcn_s_memtime();
        if (more) lstore((kt + 1) & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long s3 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        const unsigned long long s4 = __builtin_amdgcn_s_memtime();
        t_mfma += s1 - s0; t_wait += s2 - s1; t_store += s3 - s2; t_bar += s4 - s3; t_all += s4 - s0;
    }
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < TN; ++j) for (int i = 0; i < TM; ++i) for (int r = 0; r < 16; ++r)
        C[(size_t)(m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * (nbn * BN) + n0 + wc * 64 + j * 32 + li] = acc[i][j][r];
    if (lane == 0) {
        unsigned long long* s = stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
        s[0] = t_mfma; s[1] = t_wait; s[2] = t_store; s[3] = t_bar; s[4] = t_all; s[5] = te - tb; s[6] = __builtin_amdgcn_s_memrealtime(); s[7] = KT;
    }
}

template <int MODE, int LAYOUT>
void run(int rows, int N, int K, const char* name) {
    float *A, *W, *C; unsigned long long* st;
    const int nbm = rows / BM, nbn = N / BN, nb = nbm * nbn;
    hipMalloc(&A, (size_t)rows * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)rows * N * 4); hipMalloc(&st, (size_t)nb * 4 * 8 * 8);
    hipMemset(A, 0, (size_t)rows * K * 4); hipMemset(W, 0, (size_t)N * K * 4);
    std::vector<float> h((size_t)N * K); for (auto& v : h) v = (rand() % 17 - 8) * 0.125f;
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> ha((size_t)rows * K); for (auto& v : ha) v = (rand() % 17 - 8) * 0.125f;
    hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = 2 * STAGE * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, LAYOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, LAYOUT>), dim3(nb), dim3(256), lds, 0, A, W, C, K, nbn, st);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, LAYOUT>), dim3(nb), dim3(256), lds, 0, A, W, C, K, nbn, st);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> s((size_t)nb * 4 * 8);
    hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
    double sum[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < nb * 4; ++b) for (int q = 0; q < 6; ++q) sum[q] += (double)s[(size_t)b * 8 + q];
    const double kt = (double)nb * 4 * (K / 32);
    printf("L%d %-9s rows=%6d N=%4d K=%4d blocks=%5d (%.1f/CU) %8.1f us %6.1f TF | cycles per k-tile per wave: mfma-issue %6.0f  vmcnt-wait %6.0f  lds-store %5.0f  barrier %5.0f  total %6.0f | loop/wave %8.0f\n",
           LAYOUT, name, rows, N, K, nb, nb / 256.0, ms * 1e3, 2.0 * rows * N * K / ms / 1e9, sum[0] / kt, sum[1] / kt, sum[2] / kt, sum[3] / kt, sum[4] / kt, sum[5] / (nb * 4.0));
    hipFree(A); hipFree(W); hipFree(C); hipFree(st);
}
int main(int argc, char** argv) {
    run<0, 0>(65536, 128, 2048, "full");
    run<3, 0>(65536, 128, 2048, "A-only");
    run<4, 0>(65536, 128, 2048, "W-only");
    run<1, 0>(65536, 128, 2048, "no-gload");
    run<0, 1>(65536, 128, 2048, "full");
    run<3, 1>(65536, 128, 2048, "A-only");
    run<0, 0>(65536, 512, 512, "full");
    run<0, 1>(65536, 512, 512, "full");
    run<0, 0>(65536, 3840, 512, "full");
    run<0, 1>(65536, 3840, 512, "full");
    return 0;
}
