// Backward of the cross-attention core  O = softmax(Q K^T * scale) V  (models/perceiver.py:106-113), single head, per scene.
// Part of the training primitives.  Nothing of the [N, M] score matrix is stored by the forward or here:
// both kernels recompute S tile by tile (flash-attention style), with fp32-input MFMA (v_mfma_f32_32x32x2_f32).
//
//   dq kernel   one wave = 32 queries (a workgroup = 4 waves = 128 queries of one scene); loops over 32-key tiles staged in LDS.
//               pass A: S^T = K Q^T -> running max / sum -> LSE_i ;  D_i = dO_i . O_i
//               pass B: P^T = exp(S^T scale - LSE), dP^T = V dO^T, dS^T = P^T (dP^T - D) scale, dQ^T += K^T dS^T
//               writes dQ, LSE, D.
//   dkv kernel  one wave = 32 keys (workgroup = 128 keys); loops over 32-query tiles staged in LDS (Q, dO, LSE, D):
//               S = Q K^T, P, dP = dO V^T, dS;  dV^T += dO^T P,  dK^T += Q^T dS.
// Orientation trick (as in the forward kernel): the score tile is computed TRANSPOSED so that the softmax index a lane reduces over
// lives in its own registers, and the accumulator layout of one product (rows (r&3)+8(r>>2)+4h, column = lane) is used directly as
// the B operand of the next with the contraction index permuted to match -- no shuffles, no LDS round trip for P or dS.
#include <hip/hip_runtime.h>

#include "common.h"
#include "activations.h"

namespace fc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnBwdParams {
    const float* q; int ldq;
    const float* k; int ldk;
    const float* v; int ldv;
    const float* o; int ldo;
    const float* dout; int lddo;
    float* dq; int lddq;
    float* dk; int lddk;
    float* dv; int lddv;
    float* lse; float* dvec;       // [B * N] each
    int N, M;
    float scale;
    int have_lse;                  // lse was written by the forward kernel: the dq kernel skips its own log-sum-exp pass
};

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdParams p) {
    constexpr int LD = DH + 1, HALF = DH / 2, NB = DH / 32;
    __shared__ float sK[32 * LD];
    __shared__ float sV[32 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int qrow = blockIdx.x * 128 + wave * 32 + li;
    const bool qok = qrow < p.N;
    const size_t grow = (size_t)b * p.N + (qok ? qrow : 0);
    const float* kb = p.k + (size_t)b * p.M * p.ldk;
    const float* vb = p.v + (size_t)b * p.M * p.ldv;
    float Qr[HALF], dOr[HALF];
    float dsum = 0.f;
#pragma unroll
    for (int j = 0; j < HALF; ++j) {
        Qr[j] = qok ? p.q[grow * p.ldq + 2 * j + h] : 0.f;
        dOr[j] = qok ? p.dout[grow * p.lddo + 2 * j + h] : 0.f;
        dsum += dOr[j] * (qok ? p.o[grow * p.ldo + 2 * j + h] : 0.f);
    }
    dsum += __shfl_xor(dsum, 32, 64);
    const int ntiles = (p.M + 31) / 32;
    // cooperative staging: 32 x DH floats per tile, 256 threads
    auto stage = [&](const float* src, int ld, float* dst, int t) {
        for (int e = tid; e < 32 * DH; e += 256) {
            const int r = e / DH, c = e % DH;
            const int key = t * 32 + r;
            dst[r * LD + c] = key < p.M ? src[(size_t)key * ld + c] : 0.f;
        }
    };
    // ---- pass A: log-sum-exp of every query row
    float m = -INFINITY, l = 0.f;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        stage(kb, p.ldk, sK, t);
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < HALF; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[li * LD + 2 * kk + h], Qr[kk], acc, 0, 0, 0);
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool kok = t * 32 + acc_row(r, h) < p.M;
            acc[r] = kok ? acc[r] * p.scale : -INFINITY;
            tmax = fmaxf(tmax, acc[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);
        float ts = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ts += expf(acc[r] - mn);
        ts += __shfl_xor(ts, 32, 64);
        l = l * expf(m - mn) + ts;
        m = mn;
    }
    const float lse = m + logf(l);
    if (qok && h == 0) { p.lse[grow] = lse; p.dvec[grow] = dsum; }
    // ---- pass B: dQ
    f32x16 dqa[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqa[i][r] = 0.f;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        stage(kb, p.ldk, sK, t);
        stage(vb, p.ldv, sV, t);
        __syncthreads();
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < HALF; ++kk) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[li * LD + 2 * kk + h], Qr[kk], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[li * LD + 2 * kk + h], dOr[kk], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool kok = t * 32 + acc_row(r, h) < p.M;
            const float pr = kok ? expf(s[r] * p.scale - lse) : 0.f;
            s[r] = pr * (dp[r] - dsum) * p.scale;                       // dS^T[key][query]
        }
        // dQ^T[d][query] += sum_key K[key][d] dS^T[key][query]; contraction step r takes key acc_row(r, 0) from the low half-wave and
        // acc_row(r, 1) from the high one, which is where register r of the accumulator layout holds them
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < NB; ++i)
                dqa[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[acc_row(r, h) * LD + i * 32 + li], s[r], dqa[i], 0, 0, 0);
    }
    if (qok)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) p.dq[grow * p.lddq + i * 32 + acc_row(r, h)] = dqa[i][r];
}

template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdParams p) {
    constexpr int LD = DH + 1, HALF = DH / 2, NB = DH / 32;
    __shared__ float sQ[32 * LD];
    __shared__ float sdO[32 * LD];
    __shared__ float sLse[32];
    __shared__ float sD[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int krow = blockIdx.x * 128 + wave * 32 + li;
    const bool kok = krow < p.M;
    const size_t gk = (size_t)b * p.M + (kok ? krow : 0);
    float Kr[HALF], Vr[HALF];
#pragma unroll
    for (int j = 0; j < HALF; ++j) {
        Kr[j] = kok ? p.k[gk * p.ldk + 2 * j + h] : 0.f;
        Vr[j] = kok ? p.v[gk * p.ldv + 2 * j + h] : 0.f;
    }
    f32x16 dka[NB], dva[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dka[i][r] = 0.f; dva[i][r] = 0.f; }
    const int ntiles = (p.N + 31) / 32;
    const size_t q0 = (size_t)b * p.N;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        for (int e = tid; e < 32 * DH; e += 256) {
            const int r = e / DH, c = e % DH;
            const int qi = t * 32 + r;
            sQ[r * LD + c] = qi < p.N ? p.q[(q0 + qi) * p.ldq + c] : 0.f;
            sdO[r * LD + c] = qi < p.N ? p.dout[(q0 + qi) * p.lddo + c] : 0.f;
        }
        if (tid < 32) {
            const int qi = t * 32 + tid;
            sLse[tid] = qi < p.N ? p.lse[q0 + qi] : INFINITY;        // exp(s - inf) = 0: queries past the end contribute nothing
            sD[tid] = qi < p.N ? p.dvec[q0 + qi] : 0.f;
        }
        __syncthreads();
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < HALF; ++kk) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(sQ[li * LD + 2 * kk + h], Kr[kk], s, 0, 0, 0);       // S[query][key]
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sdO[li * LD + 2 * kk + h], Vr[kk], dp, 0, 0, 0);    // dP[query][key]
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = acc_row(r, h);
            const float pr = kok ? expf(s[r] * p.scale - sLse[qi]) : 0.f;
            dp[r] = pr * (dp[r] - sD[qi]) * p.scale;                    // dS[query][key]
            s[r] = pr;                                                  // P[query][key]
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                dva[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(sdO[acc_row(r, h) * LD + i * 32 + li], s[r], dva[i], 0, 0, 0);    // dV^T[d][key]
                dka[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(sQ[acc_row(r, h) * LD + i * 32 + li], dp[r], dka[i], 0, 0, 0);    // dK^T[d][key]
            }
    }
    if (kok)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p.dk[gk * p.lddk + i * 32 + acc_row(r, h)] = dka[i][r];
                p.dv[gk * p.lddv + i * 32 + acc_row(r, h)] = dva[i][r];
            }
}

// ================================================================ the same two kernels on the split-fp16 loop (head dim 64)
// Operands as fp16 limbs (x ~ hi + lo'/2048, gemm.hip), 3 x v_mfma_f32_32x32x16_f16 per product block: hi.hi into `main`, hi.lo' and lo'.hi into
// `cross`.  K / V (dq kernel) and Q / dO (dk/dv kernel) tiles are converted once per tile while they are staged into LDS, row-major
// [32 rows][hi 64 | lo' 64]; a product that contracts over the head dim reads them as they lie (16 bytes per lane), a product that
// contracts over the tile's rows reads them through ds_read_b64_tr_b16 (attention.hip, train.hip), whose row order per lane,
// {4h..4h+3, 8+4h..8+4h+3} + 16 j, is exactly the row order of accumulator registers 8j..8j+7 -- so P^T / dS^T go from the
// accumulators into the next MFMA's B operand after a limb split only.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short v4i16 __attribute__((ext_vector_type(4)));
constexpr int A16_PITCH = 256 + 16;        // bytes per staged row: [hi 64 halfs | lo' 64 halfs] + pad

__device__ __forceinline__ void split8(const float* v, f16x8& hi, f16x8& lo) {
    const float x[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    limb_split8(x, hi, lo);                                 // five VALU per pair of values (activations.h)
}
// 32 x 64 fp32 rows (row stride ld) -> limbs in LDS; rows >= valid come out as zeros.  256 threads, 8 consecutive columns each.
// Split in two halves so that a tile's rows can be REQUESTED one tile ahead and converted / stored behind the current tile's products
// (round 3: both backward kernels staged a tile between two barriers with nothing else in flight -- every tile paid a full memory latency
// for 36 MFMAs per wave).  The load goes through a buffer descriptor that ends behind row valid - 1: the range check zeroes the rows past
// the end, no branch, and the per-lane offset is a loop constant.
struct StageRegs { float4 a, b; };
__device__ __forceinline__ StageRegs stage_load(const float* __restrict__ src, int ld, int row0, int valid, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    const size_t bytes = (size_t)(valid - row0) * ld * 4;                       // row0 < valid at every call
    const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)row0 * ld), 0, bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes, 0x00020000);
    const unsigned off = (unsigned)((r * ld + c) * 4);
    StageRegs g;
    g.a = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d, off, 0, 0));
    g.b = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d, off + 16, 0, 0));
    return g;
}
__device__ __forceinline__ void stage_store(const StageRegs& g, char* dst, int tid, float& amax) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    const float v[8] = {g.a.x, g.a.y, g.a.z, g.a.w, g.b.x, g.b.y, g.b.z, g.b.w};
    f16x8 hi, lo;
    split8(v, hi, lo);
    *reinterpret_cast<f16x8*>(dst + r * A16_PITCH + c * 2) = hi;
    *reinterpret_cast<f16x8*>(dst + r * A16_PITCH + 128 + c * 2) = lo;
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v[0]), "v"(v[1]));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v[2]), "v"(v[3]));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v[4]), "v"(v[5]));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v[6]), "v"(v[7]));
}
#define FC_TR16(PTR_) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(PTR_))
#define FC_TR16x2(PTR_) __builtin_bit_cast(f16x8, __builtin_shufflevector(FC_TR16(PTR_), FC_TR16((PTR_) + 8 * A16_PITCH), 0, 1, 2, 3, 4, 5, 6, 7))
#define FC_MMA3(MAIN_, CROSS_, AH_, AL_, BH_, BL_)                                   \
    MAIN_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH_, BH_, MAIN_, 0, 0, 0);        \
    CROSS_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH_, BL_, CROSS_, 0, 0, 0);      \
    CROSS_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL_, BH_, CROSS_, 0, 0, 0);

// (two waves per SIMD: 247 registers, no scratch -- left to itself hipcc takes 276 and one wave per SIMD: 569 -> 388 us per C2 launch)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd16_dq_kernel(AttnBwdParams p, int* __restrict__ ovf) {
    constexpr int DH = 64;
    constexpr int ST = 32 * A16_PITCH;                                // one stage; two per operand
    __shared__ __attribute__((aligned(16))) char sK[2 * ST];
    __shared__ __attribute__((aligned(16))) char sV[2 * ST];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int qrow = blockIdx.x * 128 + wave * 32 + li;
    const bool qok = qrow < p.N;
    const size_t grow = (size_t)b * p.N + (qok ? qrow : 0);
    const float* kb = p.k + (size_t)b * p.M * p.ldk;
    const float* vb = p.v + (size_t)b * p.M * p.ldv;
    // this lane's query row as B operands: step ds covers head dims 16 ds + 8 h .. + 7
    f16x8 Qh[4], Ql[4], Gh[4], Gl[4];
    float dsum = 0.f, amax = 0.f;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
        float qv[8], gv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = 16 * ds + 8 * h + i;
            qv[i] = qok ? p.q[grow * p.ldq + d] : 0.f;
            gv[i] = qok ? p.dout[grow * p.lddo + d] : 0.f;
            dsum += gv[i] * (qok ? p.o[grow * p.ldo + d] : 0.f);
            amax = fmaxf(amax, fmaxf(fabsf(qv[i]), fabsf(gv[i])));
        }
        split8(qv, Qh[ds], Ql[ds]);
        split8(gv, Gh[ds], Gl[ds]);
    }
    dsum += __shfl_xor(dsum, 32, 64);
    const int ntiles = (p.M + 31) / 32;
    const float sl2 = p.scale * 1.4426950408889634f;
    const int a_off = li * A16_PITCH + 16 * h;                        // row li, 8 halfs at head dim 8 h of a 16-dim step
    const int tr_off = (4 * h + ((lane & 15) >> 2)) * A16_PITCH + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    // ---- pass A: log-sum-exp per query (skipped when the forward kernel left it in p.lse)
    float m = -INFINITY, l = 0.f;
    for (int t = 0; t < (p.have_lse ? 0 : ntiles); ++t) {
        __syncthreads();
        stage_store(stage_load(kb, p.ldk, t * 32, p.M, tid), sK, tid, amax);
        __syncthreads();
        f32x16 sm, sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sm[r] = 0.f; sc[r] = 0.f; }
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const f16x8 kh = *reinterpret_cast<const f16x8*>(sK + a_off + 32 * ds), kl = *reinterpret_cast<const f16x8*>(sK + a_off + 32 * ds + 128);
            FC_MMA3(sm, sc, kh, kl, Qh[ds], Ql[ds])
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool kok = t * 32 + acc_row(r, h) < p.M;
            sm[r] = kok ? (sm[r] + sc[r] * (1.0f / 2048.0f)) * sl2 : -INFINITY;      // log2 domain: exp(x) = exp2(x log2 e), one v_exp_f32
            tmax = fmaxf(tmax, sm[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);
        float ts = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ts += __builtin_amdgcn_exp2f(sm[r] - mn);
        ts += __shfl_xor(ts, 32, 64);
        l = l * __builtin_amdgcn_exp2f(m - mn) + ts;
        m = mn;
    }
    const float lse = p.have_lse ? p.lse[grow] : (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;      // natural log-sum-exp
    const float lse2 = lse * 1.4426950408889634f;
    if (qok && h == 0) { if (!p.have_lse) p.lse[grow] = lse; p.dvec[grow] = dsum; }
    // ---- pass B: dQ^T[d][query] += K^T dS^T
    f32x16 qm[2], qc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { qm[i][r] = 0.f; qc[i][r] = 0.f; }
    // software pipeline over the key tiles: tile t is multiplied out of LDS stage t & 1 while tile t + 1 (in registers since the previous
    // iteration) is converted and stored into the other stage behind the products, and tile t + 2 is requested; ONE barrier per tile
    StageRegs rk, rv;
    __syncthreads();                                                   // (pass A may still be reading stage 0)
    if (ntiles > 0) {
        stage_store(stage_load(kb, p.ldk, 0, p.M, tid), sK, tid, amax);
        stage_store(stage_load(vb, p.ldv, 0, p.M, tid), sV, tid, amax);
    }
    if (ntiles > 1) { rk = stage_load(kb, p.ldk, 32, p.M, tid); rv = stage_load(vb, p.ldv, 32, p.M, tid); }
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                                               // stage t & 1 is complete; every wave is done reading the other one
        const char* const cK = sK + (t & 1) * ST;
        const char* const cV = sV + (t & 1) * ST;
        f32x16 sm, sc, pm, pc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sm[r] = 0.f; sc[r] = 0.f; pm[r] = 0.f; pc[r] = 0.f; }
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const f16x8 kh = *reinterpret_cast<const f16x8*>(cK + a_off + 32 * ds), kl = *reinterpret_cast<const f16x8*>(cK + a_off + 32 * ds + 128);
            const f16x8 vh = *reinterpret_cast<const f16x8*>(cV + a_off + 32 * ds), vl = *reinterpret_cast<const f16x8*>(cV + a_off + 32 * ds + 128);
            FC_MMA3(sm, sc, kh, kl, Qh[ds], Ql[ds])
            FC_MMA3(pm, pc, vh, vl, Gh[ds], Gl[ds])
        }
        if (t + 1 < ntiles) {                                          // the next tile's limbs, under this tile's first products
            stage_store(rk, sK + ((t + 1) & 1) * ST, tid, amax);
            stage_store(rv, sV + ((t + 1) & 1) * ST, tid, amax);
        }
        if (t + 2 < ntiles) { rk = stage_load(kb, p.ldk, (t + 2) * 32, p.M, tid); rv = stage_load(vb, p.ldv, (t + 2) * 32, p.M, tid); }
        float dsv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool kok = t * 32 + acc_row(r, h) < p.M;
            const float pr = kok ? __builtin_amdgcn_exp2f((sm[r] + sc[r] * (1.0f / 2048.0f)) * sl2 - lse2) : 0.f;
            dsv[r] = pr * ((pm[r] + pc[r] * (1.0f / 2048.0f)) - dsum) * p.scale;          // dS^T[key][query]
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f16x8 sh, sl;
            split8(dsv + 8 * j, sh, sl);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* pk = cK + (16 * j) * A16_PITCH + (32 * i) * 2 + tr_off;
                const f16x8 kh = FC_TR16x2(pk), kl = FC_TR16x2(pk + 128);
                FC_MMA3(qm[i], qc[i], kh, kl, sh, sl)
            }
        }
    }
    if (amax >= 65504.0f || amax != amax) atomicOr(ovf, 1);
    if (qok)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) p.dq[grow * p.lddq + i * 32 + acc_row(r, h)] = qm[i][r] + qc[i][r] * (1.0f / 2048.0f);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd16_dkv_kernel(AttnBwdParams p, int* __restrict__ ovf) {
    constexpr int ST = 32 * A16_PITCH;                                // one stage; two per operand (pipeline as in the dq kernel)
    __shared__ __attribute__((aligned(16))) char sQ[2 * ST];
    __shared__ __attribute__((aligned(16))) char sG[2 * ST];
    __shared__ float sLse[2][32];
    __shared__ float sD[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int krow = blockIdx.x * 128 + wave * 32 + li;
    const bool kok = krow < p.M;
    const size_t gk = (size_t)b * p.M + (kok ? krow : 0);
    f16x8 Kh[4], Kl[4], Vh[4], Vl[4];
    float amax = 0.f;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
        float kv[8], vv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = 16 * ds + 8 * h + i;
            kv[i] = kok ? p.k[gk * p.ldk + d] : 0.f;
            vv[i] = kok ? p.v[gk * p.ldv + d] : 0.f;
            amax = fmaxf(amax, fmaxf(fabsf(kv[i]), fabsf(vv[i])));
        }
        split8(kv, Kh[ds], Kl[ds]);
        split8(vv, Vh[ds], Vl[ds]);
    }
    f32x16 km[2], kc[2], vm[2], vc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { km[i][r] = 0.f; kc[i][r] = 0.f; vm[i][r] = 0.f; vc[i][r] = 0.f; }
    const int ntiles = (p.N + 31) / 32;
    const float sl2 = p.scale * 1.4426950408889634f;
    const size_t q0 = (size_t)b * p.N;
    const int a_off = li * A16_PITCH + 16 * h;
    const int tr_off = (4 * h + ((lane & 15) >> 2)) * A16_PITCH + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const float* const qsrc = p.q + q0 * p.ldq;
    const float* const gsrc = p.dout + q0 * p.lddo;
    StageRegs rq, rg;
    float r_lse = INFINITY, r_d = 0.f;
    auto vec_load = [&](int t) {                                       // per-query log-sum-exp (log2 domain) and D of tile t, lanes 0..31 of wave 0
        if (tid < 32) {
            const int qi = t * 32 + tid;
            r_lse = qi < p.N ? p.lse[q0 + qi] * 1.4426950408889634f : INFINITY;          // exp2(s - inf) = 0: queries past the end contribute nothing
            r_d = qi < p.N ? p.dvec[q0 + qi] : 0.f;
        }
    };
    auto vec_store = [&](int st) { if (tid < 32) { sLse[st][tid] = r_lse; sD[st][tid] = r_d; } };
    if (ntiles > 0) {
        stage_store(stage_load(qsrc, p.ldq, 0, p.N, tid), sQ, tid, amax);
        stage_store(stage_load(gsrc, p.lddo, 0, p.N, tid), sG, tid, amax);
        vec_load(0); vec_store(0);
    }
    if (ntiles > 1) { rq = stage_load(qsrc, p.ldq, 32, p.N, tid); rg = stage_load(gsrc, p.lddo, 32, p.N, tid); vec_load(1); }
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                                               // stage t & 1 is complete; every wave is done reading the other one
        const int cur = t & 1;
        const char* const cQ = sQ + cur * ST;
        const char* const cG = sG + cur * ST;
        // The next tile's limbs first (the co-resident wave's products cover the conversion), then S, then dP: one pair of score accumulators
        // live at a time and the staging registers empty from here to the next request -- 256 registers hold the rest (two waves per SIMD)
        if (t + 1 < ntiles) {
            stage_store(rq, sQ + (cur ^ 1) * ST, tid, amax);
            stage_store(rg, sG + (cur ^ 1) * ST, tid, amax);
            vec_store(cur ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        float pv[16], dsv[16];
        {
            f32x16 sm, sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sm[r] = 0.f; sc[r] = 0.f; }
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const f16x8 qh = *reinterpret_cast<const f16x8*>(cQ + a_off + 32 * ds), ql = *reinterpret_cast<const f16x8*>(cQ + a_off + 32 * ds + 128);
                FC_MMA3(sm, sc, qh, ql, Kh[ds], Kl[ds])                  // S[query][key]
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                pv[r] = kok ? __builtin_amdgcn_exp2f((sm[r] + sc[r] * (1.0f / 2048.0f)) * sl2 - sLse[cur][acc_row(r, h)]) : 0.f;
        }
        {
            f32x16 pm, pc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { pm[r] = 0.f; pc[r] = 0.f; }
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const f16x8 gh = *reinterpret_cast<const f16x8*>(cG + a_off + 32 * ds), gl = *reinterpret_cast<const f16x8*>(cG + a_off + 32 * ds + 128);
                FC_MMA3(pm, pc, gh, gl, Vh[ds], Vl[ds])                  // dP[query][key]
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dsv[r] = pv[r] * ((pm[r] + pc[r] * (1.0f / 2048.0f)) - sD[cur][acc_row(r, h)]) * p.scale;
        }
        // dV from P, then dK from dS (each accumulator still receives j = 0 before j = 1): fewer operands live at once than with both interleaved
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f16x8 ph, pl;
            split8(pv + 8 * j, ph, pl);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* pg = cG + (16 * j) * A16_PITCH + (32 * i) * 2 + tr_off;
                const f16x8 gh = FC_TR16x2(pg), gl = FC_TR16x2(pg + 128);
                FC_MMA3(vm[i], vc[i], gh, gl, ph, pl)                    // dV^T[d][key] += dO^T P
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f16x8 sh, sl;
            split8(dsv + 8 * j, sh, sl);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* pq = cQ + (16 * j) * A16_PITCH + (32 * i) * 2 + tr_off;
                const f16x8 qh = FC_TR16x2(pq), ql = FC_TR16x2(pq + 128);
                FC_MMA3(km[i], kc[i], qh, ql, sh, sl)                    // dK^T[d][key] += Q^T dS
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < ntiles) { rq = stage_load(qsrc, p.ldq, (t + 2) * 32, p.N, tid); rg = stage_load(gsrc, p.lddo, (t + 2) * 32, p.N, tid); vec_load(t + 2); }
    }
    if (amax >= 65504.0f || amax != amax) atomicOr(ovf, 1);
    if (kok)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p.dk[gk * p.lddk + i * 32 + acc_row(r, h)] = km[i][r] + kc[i][r] * (1.0f / 2048.0f);
                p.dv[gk * p.lddv + i * 32 + acc_row(r, h)] = vm[i][r] + vc[i][r] * (1.0f / 2048.0f);
            }
}
#undef FC_MMA3
#undef FC_TR16x2
#undef FC_TR16

int g_train_attn16 = 1;       // tuning knob (fc_debug_set 12): attention backward on the split-fp16 loop inside a guard scope (head dim 64)

static void launch_attn_bwd16(const AttnBwdParams& p, int B, int* ovf, hipStream_t s) {
    const double fl = 2.0 * B * (double)p.N * p.M * 64;
    {
        ProfScope ps("fc::attn_bwd16_dq_kernel", 4.0 * fl, 0.0, s);
        hipLaunchKernelGGL(attn_bwd16_dq_kernel, dim3((p.N + 127) / 128, B), dim3(256), 0, s, p, ovf);
        FC_HIP(hipGetLastError());
    }
    ProfScope ps("fc::attn_bwd16_dkv_kernel", 4.0 * fl, 0.0, s);
    hipLaunchKernelGGL(attn_bwd16_dkv_kernel, dim3((p.M + 127) / 128, B), dim3(256), 0, s, p, ovf);
    FC_HIP(hipGetLastError());
}

template <int DH>
static void launch_attn_bwd(const AttnBwdParams& p, int B, hipStream_t s) {
    const double fl = 2.0 * B * (double)p.N * p.M * DH;
    {
        ProfScope ps("fc::attn_bwd_dq_kernel", 4.0 * fl, 0.0, s);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<DH>, dim3((p.N + 127) / 128, B), dim3(256), 0, s, p);
        FC_HIP(hipGetLastError());
    }
    ProfScope ps("fc::attn_bwd_dkv_kernel", 4.0 * fl, 0.0, s);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<DH>, dim3((p.M + 127) / 128, B), dim3(256), 0, s, p);
    FC_HIP(hipGetLastError());
}

bool launch_attention_scaled_op(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, int B, int N, int M,
                                int dh_pad, float scale, void* limb_ws, hipStream_t s, float* lse);      // attention.hip

}  // namespace fc

using namespace fc;


static void check_mat(const void* p, int ld, int D, const char* what) {
    if (!p || ld < D || ld % 4 != 0 || ((uintptr_t)p & 15)) throw Error(FC_ERR_INVALID, std::string("training attention: bad matrix ") + what);
}

extern "C" {

size_t fc_train_attention_ws_bytes(int32_t B, int32_t N, int32_t M, int32_t D) {
    (void)N;
    return attention_limb_ws_bytes((long)B * M, round_up(D, 32)) + 256;
}

int fc_train_attention_fwd_f32(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, float* out, int32_t ldo,
                               int32_t B, int32_t N, int32_t M, int32_t D, float scale, void* ws, size_t ws_bytes, float* stats, int32_t* stats_valid,
                               int32_t* ovf, void* stream) {
    FC_API_BEGIN
    if (B < 1 || N < 1 || M < 1 || (D != 32 && D != 64)) throw Error(FC_ERR_UNSUPPORTED, "fc_train_attention_fwd_f32: head dim (padded) must be 32 or 64");
    check_mat(q, ldq, D, "q"); check_mat(k, ldk, D, "k"); check_mat(v, ldv, D, "v"); check_mat(out, ldo, D, "out");
    const bool f16 = ovf && ws && ws_bytes >= fc_train_attention_ws_bytes(B, N, M, D) && !((uintptr_t)ws & 15);
    Fp16FlagScope scope(f16 ? (int*)ovf : nullptr);
    const bool wrote = launch_attention_scaled_op(q, ldq, k, ldk, v, ldv, out, ldo, B, N, M, D, scale, f16 ? ws : nullptr, (hipStream_t)stream,
                                                  (f16 && D == 64) ? stats : nullptr);
    if (stats_valid) *stats_valid = wrote ? 1 : 0;
    FC_API_END
}

int fc_train_attention_bwd_f32(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const float* out, int32_t ldo,
                               const float* dout, int32_t lddo, float* dq, int32_t lddq, float* dk, int32_t lddk, float* dv, int32_t lddv,
                               float* stats, int32_t stats_valid, int32_t B, int32_t N, int32_t M, int32_t D, float scale, int32_t* ovf, void* stream) {
    FC_API_BEGIN
    if (B < 1 || N < 1 || M < 1 || (D != 32 && D != 64)) throw Error(FC_ERR_UNSUPPORTED, "fc_train_attention_bwd_f32: head dim (padded) must be 32 or 64");
    check_mat(q, ldq, D, "q"); check_mat(k, ldk, D, "k"); check_mat(v, ldv, D, "v"); check_mat(out, ldo, D, "out"); check_mat(dout, lddo, D, "dout");
    check_mat(dq, lddq, D, "dq"); check_mat(dk, lddk, D, "dk"); check_mat(dv, lddv, D, "dv");
    if (!stats) throw Error(FC_ERR_INVALID, "fc_train_attention_bwd_f32: stats scratch [2 * B * N] is required");
    AttnBwdParams p{q, ldq, k, ldk, v, ldv, out, ldo, dout, lddo, dq, lddq, dk, lddk, dv, lddv, stats, stats + (size_t)B * N, N, M, scale, 0};
    p.have_lse = (stats_valid && D == 64 && ovf && g_train_attn16) ? 1 : 0;
    if (D == 64 && ovf && g_train_attn16) launch_attn_bwd16(p, B, (int*)ovf, (hipStream_t)stream);
    else if (D == 32) launch_attn_bwd<32>(p, B, (hipStream_t)stream);
    else launch_attn_bwd<64>(p, B, (hipStream_t)stream);
    FC_API_END
}

}  // extern "C"
