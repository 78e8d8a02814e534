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
#include <cstdio>

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

// ---------------------------------------------------------------------------------------------------------------------------------
// The same selection on the matrix cores (round 3).  The reference literally computes the Gram matrix with `matmul`
// (models/pytorch_gcn.py:13-16); here a wave owns 32 queries and produces a 32 x 32 tile of inner products per step with
// v_mfma_f32_32x32x2_f32 -- exact fp32 products and sums, the fp32-input matrix rate is twice the scalar FMA rate and needs ONE LDS
// read per 4 products where the lane-per-candidate loop above needs five per 16 -- queries as the A operand (held in registers for the
// whole launch), candidates as B (staged through LDS in blocks of 64, double-buffered, one barrier per block, shared by the four waves
// of a workgroup).  The accumulator of a lane then holds 16 QUERIES of ONE candidate (column = lane & 31, rows 8 (r / 4) + 4 (lane >>
// 5) + r % 4), so `pd > tau` is one compare per register and a ballot tells which of the 32 candidates beat the current k-th best of the
// two queries a register stands for (lower / upper half-wave).
// The per-query top-k lives in REGISTERS, sorted: query (r, half) owns one value VGPR and one index VGPR, entry e on lane e (k <= 64).
// A hit is inserted with one whole-wave lane shift (v_mov_b32_dpp wave_shr:1): lanes holding smaller entries take min(left neighbour,
// new value), so the list stays sorted, the k-th best (lane k - 1) is always exact -- tau never lags, which keeps the hits at
// k (1 + ln(M / k)) per query instead of the ~2x of the lazily pruned LDS set above -- and nothing of it touches LDS.
// Ranking value as above: pd = fl( fl(2 xi.xj - |xj|^2) - |xi|^2 ), largest first, strict `>` against the k-th best (earlier candidates
// win ties); norms are summed per half-row and then across the two halves, queries and candidates alike.
typedef float knn_f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float knn_wave_shr1(float v, float edge) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ int knn_wave_shr1(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false); }

// inserts (v, ci) into the descending list (sv, si): entry e on lane e; what falls off lane 63 is lost (k <= 64 entries matter)
__device__ __forceinline__ void knn_insert(float& sv, int& si, float v, int ci) {
    const bool smaller = sv < v;                                  // a suffix of the lanes
    const float shv = knn_wave_shr1(sv, INFINITY);                // lane e <- entry e - 1; lane 0 <- +inf
    const int shi = knn_wave_shr1(si, 0);
    const bool moved = shv < v;                                   // lanes behind the insertion point
    const float nv = moved ? shv : v;
    const int ni = moved ? shi : ci;
    sv = smaller ? nv : sv;
    si = smaller ? ni : si;
}

constexpr int KM_TC = 64;       // candidates per LDS block

template <int CP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void knn_mfma_kernel(const float* __restrict__ f, int ldf, int C, int32_t* __restrict__ idx_out, int M, int m_stride,
                                                       int k, const int32_t* __restrict__ warm) {
    extern __shared__ float smem[];
    constexpr int LDC = CP + 4, H = CP / 2, NLD = (KM_TC * CP / 4 + 255) / 256;
    float* s_qxx = smem + 2 * KM_TC * LDC;                        // [4 waves][32]
    float* s_tau = s_qxx + 128;                                   // [4 waves][32]: warm-start thresholds
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * 128 + wave * 32;
    const float* fb = f + (size_t)b * m_stride * ldf;

    // ---- this wave's 32 queries as the A operand: lane (i, kh) holds features kh * H + 0 .. H - 1 of query i
    float aq[H];
    {
        int qi = q0 + li;
        qi = qi < M ? qi : M - 1;
        const float* src = fb + (size_t)qi * ldf + lh * H;
        float nrm = 0.f;
#pragma unroll
        for (int u = 0; u < H; u += 4) {
            float4 v = *reinterpret_cast<const float4*>(src + u);
            if (CP != C) {                                        // columns C .. CP - 1 are padding, whatever the memory holds
                const int c0 = lh * H + u;
                v.x = c0 < C ? v.x : 0.f; v.y = c0 + 1 < C ? v.y : 0.f; v.z = c0 + 2 < C ? v.z : 0.f; v.w = c0 + 3 < C ? v.w : 0.f;
            }
            aq[u] = v.x; aq[u + 1] = v.y; aq[u + 2] = v.z; aq[u + 3] = v.w;
            nrm = fmaf(v.x, v.x, nrm); nrm = fmaf(v.y, v.y, nrm); nrm = fmaf(v.z, v.z, nrm); nrm = fmaf(v.w, v.w, nrm);
        }
        nrm += __shfl_xor(nrm, 32);
        if (lh == 0) s_qxx[wave * 32 + li] = nrm;
        // ---- warm start (round 4): `warm` holds k neighbours of every query from the PREVIOUS DGCNN level (models/pytorch_gcn.py:43-60: the four
        // edge convolutions search the same cloud in successive feature spaces).  Any k distinct candidates bound the k-th best from below, so
        // the stream starts with tau0 = min_j pd(query, warm_j) instead of -inf and the ~k ln(M / k) insertions that only serve to raise the
        // threshold never happen (the previous level's neighbours are mostly near in this level's space too).  The selection stays EXACT:
        // tau0 is lowered by a bound on the rounding difference between this scalar fmaf chain and the MFMA's sums, every candidate above it
        // goes through the same sorted insertion in the same order, and tau only ever rises.  A set that holds an index twice (e.g. through the
        // "open slots get the query itself" rule below), an index outside the cloud or a non-finite value gives no bound: tau0 = -inf, the
        // plain stream.  Distinctness is checked for real (a wave looks at its 32 sets one after the other, entry j on lane j, against the
        // entries 1 .. k / 2 places further round the set: every unordered pair once).
        float tau0 = -INFINITY;
        if (warm) {
            unsigned dup_mask = 0;                                     // bit q: the set of this wave's query q repeats an index
            for (int q = 0; q < 32; ++q) {
                int qq = q0 + q;
                qq = qq < M ? qq : M - 1;
                const int mine = lane < k ? warm[((size_t)b * M + qq) * k + lane] : -1 - lane;
                bool dup = false;
                for (int d = 1; d <= k / 2; ++d) {
                    int o = lane + d;
                    o = o >= k ? o - k : o;
                    dup |= lane < k && __shfl(mine, o, 64) == mine;
                }
                dup_mask |= __ballot(dup) ? (1u << q) : 0u;
            }
            const int32_t* wl = warm + ((size_t)b * M + qi) * k;
            float tmin = INFINITY, cmax = 0.f;
            bool bad = ((dup_mask >> li) & 1u) != 0;
            for (int j = 0; j < k; ++j) {
                int ci = wl[j];
                bad |= ci < 0 || ci >= M;
                ci = ci < 0 ? 0 : (ci >= M ? M - 1 : ci);
                const float* row = fb + (size_t)ci * ldf + lh * H;
                float dot = 0.f, cx = 0.f;
#pragma unroll
                for (int u = 0; u < H; u += 4) {
                    float4 v = *reinterpret_cast<const float4*>(row + u);
                    if (CP != C) {
                        const int c0 = lh * H + u;
                        v.x = c0 < C ? v.x : 0.f; v.y = c0 + 1 < C ? v.y : 0.f; v.z = c0 + 2 < C ? v.z : 0.f; v.w = c0 + 3 < C ? v.w : 0.f;
                    }
                    dot = fmaf(aq[u], v.x, dot); dot = fmaf(aq[u + 1], v.y, dot); dot = fmaf(aq[u + 2], v.z, dot); dot = fmaf(aq[u + 3], v.w, dot);
                    cx = fmaf(v.x, v.x, cx); cx = fmaf(v.y, v.y, cx); cx = fmaf(v.z, v.z, cx); cx = fmaf(v.w, v.w, cx);
                }
                dot += __shfl_xor(dot, 32);
                cx += __shfl_xor(cx, 32);
                const float pd = fmaf(2.0f, dot, -cx) - nrm;
                bad |= !(pd == pd);
                tmin = fminf(tmin, pd);
                cmax = fmaxf(cmax, cx);
            }
            // |2 x.y computed two ways| differs by at most ~2 C eps |x| |y| <= C eps (|x|^2 + |y|^2); 2^-21 = 4 eps covers the two subtractions as well
            tau0 = tmin - (float)CP * 4.76837158e-7f * (nrm + cmax + fabsf(tmin));
            if (bad || !(tau0 == tau0)) tau0 = -INFINITY;
        }
        if (lh == 0) s_tau[wave * 32 + li] = tau0;
    }
    // ---- block loader: thread e covers float4 e of the 64 x CP block (register-staged, written to the other buffer after the block's products)
    // in two parts (one per 32-candidate tile of the block in flight), so that only half of a block's float4s are live at a time
    constexpr int NPART = NLD >= 2 ? 2 : 1, NPP = NLD / NPART;
    float4 stg[NPP];
    auto gload = [&](int blk, int part) {
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            const int e = tid + 256 * (part * NPP + i);
            if (NLD * 256 == KM_TC * CP / 4 || e < KM_TC * CP / 4) {
                const int r = e / (CP / 4), c4 = (e - r * (CP / 4)) * 4;
                int ci = blk * KM_TC + r;
                ci = ci < M ? ci : M - 1;
                float4 v = *reinterpret_cast<const float4*>(fb + (size_t)ci * ldf + c4);
                if (CP != C) { v.x = c4 < C ? v.x : 0.f; v.y = c4 + 1 < C ? v.y : 0.f; v.z = c4 + 2 < C ? v.z : 0.f; v.w = c4 + 3 < C ? v.w : 0.f; }
                stg[i] = v;
            }
        }
    };
    auto lstore = [&](float* dst, int part) {
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            const int e = tid + 256 * (part * NPP + i);
            if (NLD * 256 == KM_TC * CP / 4 || e < KM_TC * CP / 4) {
                const int r = e / (CP / 4), c4 = (e - r * (CP / 4)) * 4;
                *reinterpret_cast<float4*>(dst + r * LDC + c4) = stg[i];
            }
        }
    };
    float qxx[16], tau[16];
    float setv[16][2];
    int seti[16][2];
#pragma unroll
    for (int part = 0; part < NPART; ++part) { gload(0, part); lstore(smem, part); }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        qxx[r] = s_qxx[wave * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)];
        tau[r] = s_tau[wave * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)];
        setv[r][0] = setv[r][1] = -INFINITY;
        seti[r][0] = seti[r][1] = 0;
    }

    const int nblk = (M + KM_TC - 1) / KM_TC;
    for (int blk = 0; blk < nblk; ++blk) {
        const float* cur = smem + (blk & 1) * (KM_TC * LDC);
        float* nxt = smem + ((blk + 1) & 1) * (KM_TC * LDC);
        const bool more = blk + 1 < nblk;
#pragma unroll 1
        for (int t = 0; t < KM_TC / 32; ++t) {
            const int cbase = blk * KM_TC + t * 32;
            if (more && t < NPART) gload(blk + 1, t);            // (the next block is complete: cbase < M for both of this block's tiles)
            if (cbase >= M) break;
            const float* brow = cur + (t * 32 + li) * LDC + lh * H;
            knn_f16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            float cxx = 0.f;
#pragma unroll
            for (int u = 0; u < H; u += 4) {
                const float4 bv = *reinterpret_cast<const float4*>(brow + u);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u], bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u + 1], bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u + 2], bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u + 3], bv.w, acc, 0, 0, 0);
                cxx = fmaf(bv.x, bv.x, cxx); cxx = fmaf(bv.y, bv.y, cxx); cxx = fmaf(bv.z, bv.z, cxx); cxx = fmaf(bv.w, bv.w, cxx);
            }
            cxx += __shfl_xor(cxx, 32);
            if (cbase + li >= M) cxx = INFINITY;                  // candidates past the end: pd = -inf, never a hit
            // tau is refreshed once per register, behind its hits: a value below the true k-th best lands behind lane k - 1, where it is
            // ignored.  (Measured: inserting the hits of four lists side by side -- two registers x two half-waves, dummies of -inf for lists
            // without a pending hit -- is no faster, 5.4 vs 5.2 ms per C2 forward: the second wave of the SIMD already fills the gaps of
            // one insertion's dependent chain, so the dummy work costs what the interleaving gains.)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pd = fmaf(2.0f, acc[r], -cxx) - qxx[r];
                unsigned long long mask = __ballot(pd > tau[r]);
                if (mask) {                                       // (wave-uniform)
                    do {
                        const int bpos = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pd), bpos));
                        const int ci = cbase + (bpos & 31);
                        if (bpos < 32) knn_insert(setv[r][0], seti[r][0], v, ci);
                        else knn_insert(setv[r][1], seti[r][1], v, ci);
                    } while (mask);
                    const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, setv[r][0]), k - 1));
                    const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, setv[r][1]), k - 1));
                    tau[r] = fmaxf(tau[r], lh ? t1 : t0);        // (never below the warm-start threshold; without one the list's k-th entry only rises anyway)
                }
            }
            if (more && t < NPART) lstore(nxt, t);
        }
        __syncthreads();
    }
    // ---- entry e of query (r, half) sits on lane e.  Fewer than k finite candidates (NaN / -inf features in a pass whose range flag is
    // already up): the open slots get the query itself, never an uninitialised index (the gathers downstream trust them)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int qi = q0 + 8 * (r >> 2) + 4 * h + (r & 3);
            if (qi < M && lane < k) idx_out[((size_t)b * M + qi) * k + lane] = setv[r][h] > -INFINITY ? seti[r][h] : qi;
        }
}

template <int CP>
static void launch_knn_mfma(const float* f, int ldf, int C, int32_t* idx, int B, int M, int m_stride_rows, int k, hipStream_t s, const int32_t* warm) {
    constexpr size_t lds = (2 * (size_t)KM_TC * (CP + 4) + 256) * sizeof(float);
    static PerDeviceOnce attr_once;
    attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_mfma_kernel<CP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); return 0; });
    char name[112];
    snprintf(name, sizeof name, "void fc::knn_mfma_kernel<%d>(float const*, int, int, int*, int, int, int, int const*)", CP);
    ProfScope ps(name, 2.0 * B * (double)M * M * C, 4.0 * B * (double)M * (C + k), s);
    hipLaunchKernelGGL(knn_mfma_kernel<CP>, dim3((M + 127) / 128, B), dim3(256), lds, s, f, ldf, C, idx, M, m_stride_rows, k, warm);
    FC_HIP(hipGetLastError());
}

int g_knn_warm = 1;          // knob 32: 1 = a level's search starts from the previous level's neighbour sets where the caller hands them over (shipped), 0 = never
int g_knn_mfma = 1;          // knob 24: 1 = Gram tiles on the matrix cores + sorted register lists where the launch fills the chip (shipped), 2 = always (tests),
                             // 0 = the lane-per-candidate kernel above

// warm: k neighbours per query of the same cloud from another feature space (may alias idx: a workgroup reads its own queries' sets before it
// writes them), or null; used by the matrix-core kernel only -- the result is the exact top-k set either way
void launch_knn(const float* f, int ldf, int C, int32_t* idx, int B, int M, int m_stride_rows, int k, hipStream_t s, const int32_t* warm) {
    if (!g_knn_warm) warm = nullptr;
    if (k > 64 || k < 1) throw Error(FC_ERR_UNSUPPORTED, "knn: k must be in [1, 64]");
    if (M < k) throw Error(FC_ERR_INVALID, "knn: fewer points than neighbours (torch.topk would raise as well)");
    const int Cp = round_up(C, 4);
    if (ldf < Cp || ldf % 4 != 0 || ((uintptr_t)f & 15)) throw Error(FC_ERR_INVALID, "knn: feature pitch must cover round_up(C,4) and be 16-byte aligned");
    if (g_knn_mfma) {
        const int Cp8 = C <= 8 ? 8 : (C <= 16 ? 16 : (C <= 32 ? 32 : (C <= 64 ? 64 : (C <= 128 ? 128 : 0))));
        // (wider features or a pitch that does not cover the padded row: the kernel below; so do small scenes -- a workgroup of the MFMA kernel
        // owns 128 queries for ALL M candidates, which takes as long with two scenes in the batch as with sixteen, so the choice depends on
        // M alone (a scene's neighbour sets, near-ties included, must not depend on the batch it sits in); at M = 1024 the 16-query
        // workgroups of the kernel below finish a C1 batch in 0.37 ms against 2.7)
        if (Cp8 && ldf >= Cp8 && (g_knn_mfma == 2 || M >= 2048)) {
            switch (Cp8) {
                case 8: launch_knn_mfma<8>(f, ldf, C, idx, B, M, m_stride_rows, k, s, warm); break;
                case 16: launch_knn_mfma<16>(f, ldf, C, idx, B, M, m_stride_rows, k, s, warm); break;
                case 32: launch_knn_mfma<32>(f, ldf, C, idx, B, M, m_stride_rows, k, s, warm); break;
                case 64: launch_knn_mfma<64>(f, ldf, C, idx, B, M, m_stride_rows, k, s, warm); break;
                default: launch_knn_mfma<128>(f, ldf, C, idx, B, M, m_stride_rows, k, s, warm); break;
            }
            return;
        }
    }
    const size_t lds = ((size_t)(64 + 16) * (Cp + 4) + 16 + 16 * KNN_CAP) * sizeof(float) + 16 * KNN_CAP * sizeof(int32_t);
    if (lds > 160 * 1024) throw Error(FC_ERR_UNSUPPORTED, "knn: feature dimension too large for the LDS tile");
    static PerDeviceOnce attr_once;
    attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); return 0; });
    ProfScope ps("fc::knn_kernel(float const*, int, int, int*, int, int, int)", 2.0 * B * (double)M * M * C, 4.0 * B * (double)M * (C + k), s);
    hipLaunchKernelGGL(knn_kernel, dim3((M + 15) / 16, B), dim3(256), lds, s, f, ldf, Cp, idx, M, m_stride_rows, k);
    FC_HIP(hipGetLastError());
}

}  // namespace fc
