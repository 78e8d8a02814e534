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

void launch_attention_scaled_op(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, int B, int N, int M,
                                int dh_pad, float scale, void* limb_ws, hipStream_t s);      // attention.hip

}  // namespace fc

using namespace fc;

#define FC_API_BEGIN try {
#define FC_API_END                                                                            \
    }                                                                                         \
    catch (const fc::Error& e) { fc::set_last_error(e.what()); return e.code; }               \
    catch (const std::exception& e) { fc::set_last_error(e.what()); return FC_ERR_INVALID; }  \
    return FC_OK;

static void check_mat(const void* p, int ld, int D, const char* what) {
    if (!p || ld < D || ld % 4 != 0 || ((uintptr_t)p & 15)) throw Error(FC_ERR_INVALID, std::string("training attention: bad matrix ") + what);
}

extern "C" {

size_t fc_train_attention_ws_bytes(int32_t B, int32_t N, int32_t M, int32_t D) {
    (void)N;
    return attention_limb_ws_bytes((long)B * M, round_up(D, 32)) + 256;
}

int fc_train_attention_fwd_f32(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, float* out, int32_t ldo,
                               int32_t B, int32_t N, int32_t M, int32_t D, float scale, void* ws, size_t ws_bytes, int32_t* ovf, void* stream) {
    FC_API_BEGIN
    if (B < 1 || N < 1 || M < 1 || (D != 32 && D != 64)) throw Error(FC_ERR_UNSUPPORTED, "fc_train_attention_fwd_f32: head dim (padded) must be 32 or 64");
    check_mat(q, ldq, D, "q"); check_mat(k, ldk, D, "k"); check_mat(v, ldv, D, "v"); check_mat(out, ldo, D, "out");
    const bool f16 = ovf && ws && ws_bytes >= fc_train_attention_ws_bytes(B, N, M, D) && !((uintptr_t)ws & 15);
    Fp16FlagScope scope(f16 ? (int*)ovf : nullptr);
    launch_attention_scaled_op(q, ldq, k, ldk, v, ldv, out, ldo, B, N, M, D, scale, f16 ? ws : nullptr, (hipStream_t)stream);
    FC_API_END
}

int fc_train_attention_bwd_f32(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const float* out, int32_t ldo,
                               const float* dout, int32_t lddo, float* dq, int32_t lddq, float* dk, int32_t lddk, float* dv, int32_t lddv,
                               float* stats, int32_t B, int32_t N, int32_t M, int32_t D, float scale, void* stream) {
    FC_API_BEGIN
    if (B < 1 || N < 1 || M < 1 || (D != 32 && D != 64)) throw Error(FC_ERR_UNSUPPORTED, "fc_train_attention_bwd_f32: head dim (padded) must be 32 or 64");
    check_mat(q, ldq, D, "q"); check_mat(k, ldk, D, "k"); check_mat(v, ldv, D, "v"); check_mat(out, ldo, D, "out"); check_mat(dout, lddo, D, "dout");
    check_mat(dq, lddq, D, "dq"); check_mat(dk, lddk, D, "dk"); check_mat(dv, lddv, D, "dv");
    if (!stats) throw Error(FC_ERR_INVALID, "fc_train_attention_bwd_f32: stats scratch [2 * B * N] is required");
    AttnBwdParams p{q, ldq, k, ldk, v, ldv, out, ldo, dout, lddo, dq, lddq, dk, lddk, dv, lddv, stats, stats + (size_t)B * N, N, M, scale};
    if (D == 32) launch_attn_bwd<32>(p, B, (hipStream_t)stream); else launch_attn_bwd<64>(p, B, (hipStream_t)stream);
    FC_API_END
}

}  // extern "C"
