// Ablation of the split-bf16 128x128 GEMM main loop (same structure as csrc/gemm.hip VAR=3): which part of a k-tile costs what?
//   MODE 0 full | 1 no global loads in the loop | 2 no fp32->limb conversion (A limbs written from stale registers)
//   MODE 3 no LDS writes in the loop | 4 no LDS fragment reads (constant fragments) | 5 MFMAs only
// hipcc --offload-arch=gfx950 -O3 split_ablate.hip -o split_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BN = 128, TM = 2, TN = 2, ROWB = 112, STAGE3 = (BM + BN) * ROWB;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* A, const unsigned short* W3, float* C, int K, int nbn) {
    extern __shared__ char smc[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn, m0 = bm * BM, n0 = bn * BN;
    floatx16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lrow3 = tid >> 2, lc3 = (tid & 3) * 4, KT16 = K / 16;
    float4 ra3[2];
    uint4 rw3[3];
    bf16x4 sh[2], sm[2], sl[2];
#define GLOAD(KT_) { const float* a_ = A + (size_t)(m0 + lrow3) * K + (KT_) * 16 + lc3; \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) ra3[i] = *reinterpret_cast<const float4*>(a_ + (size_t)(64 * i) * K); \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) { const int c_ = tid + 256 * i, row_ = c_ / 6, part_ = c_ - row_ * 6; \
        rw3[i] = *reinterpret_cast<const uint4*>(W3 + ((size_t)(n0 + row_) * KT16 + (KT_)) * 48 + part_ * 8); } }
#define CONVERT() { _Pragma("unroll") for (int i = 0; i < 2; ++i) { const float x_[4] = {ra3[i].x, ra3[i].y, ra3[i].z, ra3[i].w}; \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { sh[i][e_] = (__bf16)x_[e_]; const float r1_ = x_[e_] - (float)sh[i][e_]; \
            sm[i][e_] = (__bf16)r1_; sl[i][e_] = (__bf16)(r1_ - (float)sm[i][e_]); } } }
#define LSTORE(ST_) { char* sa_ = smc + (ST_) * STAGE3 + lrow3 * ROWB + (tid & 3) * 8; \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) { *reinterpret_cast<bf16x4*>(sa_ + 64 * i * ROWB) = sh[i]; \
        *reinterpret_cast<bf16x4*>(sa_ + 64 * i * ROWB + 32) = sm[i]; *reinterpret_cast<bf16x4*>(sa_ + 64 * i * ROWB + 64) = sl[i]; } \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) { const int c_ = tid + 256 * i, row_ = c_ / 6, part_ = c_ - row_ * 6; \
        *reinterpret_cast<uint4*>(smc + (ST_) * STAGE3 + (BM + row_) * ROWB + part_ * 16) = rw3[i]; } }
    GLOAD(0) CONVERT() LSTORE(0)
    __syncthreads();
    bf16x8 cf;
    for (int e = 0; e < 8; ++e) cf[e] = (__bf16)(float)(lane + e);
    for (int kt = 0; kt < KT16; ++kt) {
        const int ktn = kt + 1 < KT16 ? kt + 1 : kt;
        if (MODE != 1 && MODE != 5) GLOAD(ktn)
        const char* sA = smc + (kt & 1) * STAGE3 + (wr * 64 + li) * ROWB + lh * 16;
        const char* sB = smc + (kt & 1) * STAGE3 + (BM + wc * 64 + li) * ROWB + lh * 16;
        bf16x8 af3[TM][3], bf3[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) af3[i][q] = (MODE == 4 || MODE == 5) ? cf : *reinterpret_cast<const bf16x8*>(sA + i * 32 * ROWB + q * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) bf3[j][q] = (MODE == 4 || MODE == 5) ? cf : *reinterpret_cast<const bf16x8*>(sB + j * 32 * ROWB + q * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[i][2], bf3[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[i][1], bf3[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[i][0], bf3[j][2], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[i][1], bf3[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[i][0], bf3[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[i][0], bf3[j][0], acc[i][j], 0, 0, 0);
            }
        if (MODE != 2 && MODE != 5) CONVERT()
        if (MODE != 3 && MODE != 5) LSTORE((kt + 1) & 1)
        __syncthreads();
    }
    for (int j = 0; j < TN; ++j) for (int i = 0; i < TM; ++i) for (int r = 0; r < 16; ++r)
        C[(size_t)(m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * (nbn * BN) + n0 + wc * 64 + j * 32 + li] = acc[i][j][r];
}
template <int MODE>
void run(int rows, int N, int K, const char* name) {
    float *A, *C; unsigned short* W3;
    const int nbn = N / BN, nb = rows / BM * nbn;
    hipMalloc(&A, (size_t)rows * K * 4); hipMalloc(&W3, (size_t)N * K * 6); hipMalloc(&C, (size_t)rows * N * 4);
    std::vector<float> ha((size_t)rows * K); for (auto& v : ha) v = (rand() % 2001 - 1000) * 1e-3f;
    hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned short> hw((size_t)N * K * 3); for (auto& v : hw) v = 0x3c00 + rand() % 512;
    hipMemcpy(W3, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    const size_t lds = 2 * STAGE3;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), lds, 0, A, W3, C, K, nbn);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), lds, 0, A, W3, C, K, nbn);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-22s rows=%6d N=%4d K=%4d  %8.1f us  %6.1f TF fp32-equivalent\n", name, rows, N, K, best * 1e3, 2.0 * rows * N * K / best / 1e9);
    hipFree(A); hipFree(W3); hipFree(C);
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(65536, 3840, 512, "full");
        run<1>(65536, 3840, 512, "no global loads");
        run<2>(65536, 3840, 512, "no conversion");
        run<3>(65536, 3840, 512, "no LDS writes");
        run<4>(65536, 3840, 512, "no LDS frag reads");
        run<5>(65536, 3840, 512, "MFMA only");
    }
    return 0;
}
