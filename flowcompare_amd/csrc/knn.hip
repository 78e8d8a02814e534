// k nearest neighbours in feature space for the DGCNN edge convolutions (models/pytorch_gcn.py:13-20).
//
// Ranking value, same expansion and fp32 operation order as the reference:
//     pd[i][j] = fl( fl(-|xj|^2 + 2 xi.xj) - |xi|^2 ),   top-k LARGEST over j (self included)
// The M x M matrix is never written (the reference materialises it; at 16384 points it is 1 GiB per
// scene per level): distances are produced tile by tile from LDS-staged candidate features and consumed
// immediately by a streaming top-k.
//
// Workgroup = 16 queries of one scene (4 waves x 4 queries).  Per 64-candidate tile each lane owns one
// candidate: it reads its feature row from LDS (ds_read_b128, padded rows -> conflict free) and the four
// query rows as LDS broadcasts.  Selection keeps, per query, an unordered candidate set in LDS:
// values above the current k-th best (tau) are appended with a ballot/popcount compaction; when the set
// exceeds 64 entries it is pruned back to k with a 32-step radix select over the two entries each lane holds.
// Output is the unordered SET of k indices (only max-pooling consumes it).
#include "common.h"

namespace fc {

__device__ __forceinline__ uint32_t sortable_key(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }

// prune the wave-private set (vals/idxs in LDS, cnt entries <= 128) down to its k largest; returns the k-th largest value
__device__ __forceinline__ float prune_set(float* vals, int32_t* idxs, int cnt, int k, int lane) {
    const bool h0 = lane < cnt, h1 = 64 + lane < cnt;
    const float v0 = h0 ? vals[lane] : 0.f, v1 = h1 ? vals[64 + lane] : 0.f;
    const int32_t i0 = h0 ? idxs[lane] : 0, i1 = h1 ? idxs[64 + lane] : 0;
    const uint32_t k0 = h0 ? sortable_key(v0) : 0u, k1 = h1 ? sortable_key(v1) : 0u;
    uint32_t prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t t = prefix | (1u << bit);
        const int c = popc64(__ballot(k0 >= t)) + popc64(__ballot(k1 >= t));
        if (c >= k) prefix = t;
    }
    // keep everything above the k-th key, and the first ties (slot order) to fill up to k
    const unsigned long long gt0 = __ballot(k0 > prefix), gt1 = __ballot(k1 > prefix);
    const unsigned long long eq0 = __ballot(h0 && k0 == prefix), eq1 = __ballot(h1 && k1 == prefix);
    const int n_gt = popc64(gt0) + popc64(gt1);
    const int need_eq = k - n_gt;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int eq_rank0 = popc64(eq0 & lt), eq_rank1 = popc64(eq0) + popc64(eq1 & lt);
    const bool keep0 = (k0 > prefix) || (h0 && k0 == prefix && eq_rank0 < need_eq);
    const bool keep1 = (k1 > prefix) || (h1 && k1 == prefix && eq_rank1 < need_eq);
    const unsigned long long m0 = __ballot(keep0), m1 = __ballot(keep1);
    if (keep0) { const int p = popc64(m0 & lt); vals[p] = v0; idxs[p] = i0; }
    if (keep1) { const int p = popc64(m0) + popc64(m1 & lt); vals[p] = v1; idxs[p] = i1; }
    return key_to_float(prefix);
}

constexpr int KNN_Q = 4;        // queries per wave
constexpr int KNN_CAP = 128;    // set capacity per query

__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ f, int ldf, int Cp, int32_t* __restrict__ idx_out, int M,
                                                  int m_stride, int k) {
    extern __shared__ float smem[];
    const int LDC = Cp + 4;
    float* s_cand = smem;                                   // [64][LDC]
    float* s_qry = s_cand + 64 * LDC;                       // [16][LDC]
    float* s_qxx = s_qry + 16 * LDC;                        // [16]
    float* s_vals = s_qxx + 16;                             // [16][CAP]
    int32_t* s_idx = reinterpret_cast<int32_t*>(s_vals + 16 * KNN_CAP);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, q_base = blockIdx.x * 16;
    const float* fb = f + (size_t)b * m_stride * ldf;
    const int f4r = Cp / 4;

    // ---- stage the 16 query rows, then |xq|^2 with the SAME fma order used for candidates
    for (int e = tid; e < 16 * f4r; e += 256) {
        const int r = e / f4r, c4 = (e - r * f4r) * 4;
        int qi = q_base + r;
        qi = qi < M ? qi : M - 1;
        *reinterpret_cast<float4*>(s_qry + r * LDC + c4) = *reinterpret_cast<const float4*>(fb + (size_t)qi * ldf + c4);
    }
    __syncthreads();
    if (tid < 16) {
        float xx = 0.f;
        for (int c = 0; c < Cp; ++c) { const float v = s_qry[tid * LDC + c]; xx = fmaf(v, v, xx); }
        s_qxx[tid] = xx;
    }
    float tau[KNN_Q];
    int cnt[KNN_Q];
#pragma unroll
    for (int qq = 0; qq < KNN_Q; ++qq) { tau[qq] = -INFINITY; cnt[qq] = 0; }
    __syncthreads();
    float qxx[KNN_Q];
#pragma unroll
    for (int qq = 0; qq < KNN_Q; ++qq) qxx[qq] = s_qxx[wave * KNN_Q + qq];

    const int ntiles = (M + 63) / 64;
    for (int t = 0; t < ntiles; ++t) {
        for (int e = tid; e < 64 * f4r; e += 256) {
            const int r = e / f4r, c4 = (e - r * f4r) * 4;
            int ci = t * 64 + r;
            ci = ci < M ? ci : M - 1;
            *reinterpret_cast<float4*>(s_cand + r * LDC + c4) = *reinterpret_cast<const float4*>(fb + (size_t)ci * ldf + c4);
        }
        __syncthreads();
        // ---- lane = candidate: dot with the wave's 4 queries + own squared norm
        float dot[KNN_Q] = {0.f, 0.f, 0.f, 0.f};
        float cxx = 0.f;
        const float* cr = s_cand + lane * LDC;
        const float* qr = s_qry + wave * KNN_Q * LDC;
        for (int c4 = 0; c4 < Cp; c4 += 4) {
            const float4 cv = *reinterpret_cast<const float4*>(cr + c4);
            cxx = fmaf(cv.x, cv.x, cxx); cxx = fmaf(cv.y, cv.y, cxx); cxx = fmaf(cv.z, cv.z, cxx); cxx = fmaf(cv.w, cv.w, cxx);
#pragma unroll
            for (int qq = 0; qq < KNN_Q; ++qq) {
                const float4 qv = *reinterpret_cast<const float4*>(qr + qq * LDC + c4);
                float d = dot[qq];
                d = fmaf(qv.x, cv.x, d); d = fmaf(qv.y, cv.y, d); d = fmaf(qv.z, cv.z, d); d = fmaf(qv.w, cv.w, d);
                dot[qq] = d;
            }
        }
        const int cand = t * 64 + lane;
        const bool cvalid = cand < M;
#pragma unroll
        for (int qq = 0; qq < KNN_Q; ++qq) {
            // pairwise_distance = -xx - inner - xx^T with inner = -2 x^T x   (pytorch_gcn.py:14-16)
            const float pd = (-cxx + 2.0f * dot[qq]) - qxx[qq];
            const bool hit = cvalid && pd > tau[qq];
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                float* vals = s_vals + (wave * KNN_Q + qq) * KNN_CAP;
                int32_t* idxs = s_idx + (wave * KNN_Q + qq) * KNN_CAP;
                if (hit) {
                    const int pos = cnt[qq] + popc64(mask & ((1ull << lane) - 1ull));
                    vals[pos] = pd;
                    idxs[pos] = cand;
                }
                cnt[qq] += popc64(mask);
                if (cnt[qq] > 64) {
                    tau[qq] = prune_set(vals, idxs, cnt[qq], k, lane);
                    cnt[qq] = k;
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int qq = 0; qq < KNN_Q; ++qq) {
        float* vals = s_vals + (wave * KNN_Q + qq) * KNN_CAP;
        int32_t* idxs = s_idx + (wave * KNN_Q + qq) * KNN_CAP;
        if (cnt[qq] > k) { prune_set(vals, idxs, cnt[qq], k, lane); cnt[qq] = k; }
        const int qi = q_base + wave * KNN_Q + qq;
        // fewer than k finite candidates (NaN / -inf features, e.g. in a pass whose fp16 range flag is already up and whose results
        // will be discarded): the open slots get the query itself, never an uninitialised index (the gathers downstream trust them)
        if (qi < M && lane < k) idx_out[((size_t)b * M + qi) * k + lane] = lane < cnt[qq] ? idxs[lane] : qi;
    }
}

void launch_knn(const float* f, int ldf, int C, int32_t* idx, int B, int M, int m_stride_rows, int k, hipStream_t s) {
    if (k > 64 || k < 1) throw Error(FC_ERR_UNSUPPORTED, "knn: k must be in [1, 64]");
    if (M < k) throw Error(FC_ERR_INVALID, "knn: fewer points than neighbours (torch.topk would raise as well)");
    const int Cp = round_up(C, 4);
    if (ldf < Cp || ldf % 4 != 0 || ((uintptr_t)f & 15)) throw Error(FC_ERR_INVALID, "knn: feature pitch must cover round_up(C,4) and be 16-byte aligned");
    const size_t lds = ((size_t)(64 + 16) * (Cp + 4) + 16 + 16 * KNN_CAP) * sizeof(float) + 16 * KNN_CAP * sizeof(int32_t);
    if (lds > 160 * 1024) throw Error(FC_ERR_UNSUPPORTED, "knn: feature dimension too large for the LDS tile");
    static PerDeviceOnce attr_once;
    attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); return 0; });
    ProfScope ps("fc::knn_kernel(float const*, int, int, int*, int, int, int)", 2.0 * B * (double)M * M * C, 4.0 * B * (double)M * (C + k), s);
    hipLaunchKernelGGL(knn_kernel, dim3((M + 15) / 16, B), dim3(256), lds, s, f, ldf, Cp, idx, M, m_stride_rows, k);
    FC_HIP(hipGetLastError());
}

}  // namespace fc
