// Single-head cross attention of the flow's pre-conditioner, flash style, on fp32 MFMA.
//   out[b, i, :] = softmax_j( q[b,i,:] . k[b,j,:] ) v[b,j,:]        (models/perceiver.py:106-113)
// q arrives PRE-SCALED by inner_dim^-0.5 * log2(e) (folded into the packed q projection), so the softmax
// is exp2(S - max).  The [N, M] score matrix is never materialised (the reference materialises [B,N,M]).
//
// One workgroup = 128 queries of one scene (4 waves x 32 queries); K/V tiles of 64 keys are staged in LDS
// (register prefetch, double buffered) and shared by the 4 waves.
//   S^T tile = K Q^T   (A = K rows from LDS via ds_read_b128, B = Q held in registers for the whole kernel);
//     its C layout puts the query on the LANE and the 32 keys in the 16 registers x 2 half-waves, so the
//     softmax max/sum are in-lane reductions + one cross-half exchange, and
//   O tile  = P V      consumes those registers DIRECTLY as the A operand (lane (q,h) supplies key
//     (r&3)+8(r>>2)+4h of step r), with B = V[key][d] read row-wise from LDS (conflict-free ds_read_b32).
// No LDS round trip for P, no transposed V image.
#include "common.h"
#include <cstdio>

namespace fc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct AttnParams {
    const float* q; int ldq;
    const float* k; int ldk;
    const float* v; int ldv;
    float* out; int ldo;
    int N, n_stride, M, m_stride;
    float qscale;     // extra multiplier applied to q at load (1 when the projection is pre-scaled)
};

template <int DH>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
    constexpr int NG = DH / 8, DT = DH / 32, LD = DH + 4;
    constexpr int F4R = DH / 4, RPP = 256 / F4R, PASSES = 64 / RPP;
    constexpr int STAGE = 2 * 64 * LD;
    extern __shared__ float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;

    // ---- Q fragment of this lane's query (B operand of S^T = K Q^T): q[8g + 4h + e]
    float4 qf[NG];
    {
        int qi = q0 + li;
        qi = qi < p.N ? qi : p.N - 1;
        const float* qp = p.q + ((size_t)b * p.n_stride + qi) * p.ldq + 4 * lh;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float4 t = *reinterpret_cast<const float4*>(qp + 8 * g);
            qf[g] = make_float4(t.x * p.qscale, t.y * p.qscale, t.z * p.qscale, t.w * p.qscale);
        }
    }

    floatx16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // ---- staging: thread t moves float4 (t % F4R) of key rows (t / F4R) + RPP*pass, for K and for V.
    //      Straight-line macros on plain register arrays (lambdas / conditionals here sent the tile through scratch memory).
    const int srow = tid / F4R, sc4 = (tid % F4R) * 4;
    float4 rk[PASSES], rv[PASSES];
    const float* kb = p.k + (size_t)b * p.m_stride * p.ldk + sc4;
    const float* vb = p.v + (size_t)b * p.m_stride * p.ldv + sc4;
    float* const sKst = smem + srow * LD + sc4;
#define FC_GLOAD(T_)                                                                          \
    _Pragma("unroll") for (int i = 0; i < PASSES; ++i) {                                      \
        int key_ = (T_) * 64 + srow + RPP * i;                                                \
        key_ = key_ < p.M ? key_ : p.M - 1; /* clamped rows are masked to -inf below */       \
        rk[i] = *reinterpret_cast<const float4*>(kb + (size_t)key_ * p.ldk);                  \
        rv[i] = *reinterpret_cast<const float4*>(vb + (size_t)key_ * p.ldv);                  \
    }
#define FC_LSTORE(ST_)                                                                        \
    _Pragma("unroll") for (int i = 0; i < PASSES; ++i) {                                      \
        *reinterpret_cast<float4*>(sKst + (ST_) * STAGE + RPP * i * LD) = rk[i];              \
        *reinterpret_cast<float4*>(sKst + (ST_) * STAGE + 64 * LD + RPP * i * LD) = rv[i];    \
    }

    const int ntiles = (p.M + 63) / 64;
    FC_GLOAD(0)
    FC_LSTORE(0)
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int tn = t + 1 < ntiles ? t + 1 : t;          // the last iteration re-loads its own tile: branch-free loop
        FC_GLOAD(tn)
        const float* sK = smem + (t & 1) * STAGE;
        const float* sV = sK + 64 * LD;

        // ---- S^T = K Q^T for the two 32-key halves of the tile
        floatx16 s[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[h2][r] = 0.f;
            const float* kr = sK + (32 * h2 + li) * LD + 4 * lh;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const float4 kf = *reinterpret_cast<const float4*>(kr + 8 * g);
                s[h2] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[g].x, s[h2], 0, 0, 0);
                s[h2] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[g].y, s[h2], 0, 0, 0);
                s[h2] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[g].z, s[h2], 0, 0, 0);
                s[h2] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[g].w, s[h2], 0, 0, 0);
            }
        }
        // ---- mask the tail keys of the last tile
        if (t * 64 + 64 > p.M) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + 32 * h2 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= p.M) s[h2][r] = -INFINITY;
                }
        }
        // ---- online softmax (this lane's query = lane&31; the other half of its keys lives in lane^32)
        float mt = s[0][0];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[h2][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = exp2f(m_run - m_new);          // 0 on the first tile (m_run = -inf)
        float lt = 0.f;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp2f(s[h2][r] - m_new);
                s[h2][r] = pv;
                lt += pv;
            }
        lt += __shfl_xor(lt, 32, 64);
        l_run = l_run * alpha + lt;
        m_run = m_new;
        // ---- rescale O: its rows are queries (r&3)+8(r>>2)+4h, whose alpha lives in that LANE
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ar = __shfl(alpha, (r & 3) + 8 * (r >> 2) + 4 * lh, 64);
#pragma unroll
                for (int d = 0; d < DT; ++d) o[d][r] *= ar;
            }
        }
        // ---- O += P V : A = P registers (query on the lane, key per step), B = V[key][d] from LDS
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vr = sV + (32 * h2 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LD + li;
#pragma unroll
                for (int d = 0; d < DT; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[h2][r], vr[32 * d], o[d], 0, 0, 0);
            }
        }
        FC_LSTORE((t + 1) & 1)
        __syncthreads();
    }
#undef FC_GLOAD
#undef FC_LSTORE

    // ---- normalise and store: O rows are queries (r&3)+8(r>>2)+4h of this wave, columns d = 32*dt + lane&31
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float lr = __shfl(l_run, qr, 64);
        const int qi = q0 + qr;
        if (qi < p.N) {
            float* op = p.out + ((size_t)b * p.n_stride + qi) * p.ldo + li;
#pragma unroll
            for (int d = 0; d < DT; ++d) op[32 * d] = o[d][r] / lr;
        }
    }
}

template <int DH>
static void launch_attn_dh(const AttnParams& p, int B, hipStream_t s) {
    constexpr size_t lds = 2 * 2 * 64 * (size_t)(DH + 4) * sizeof(float);
    static bool attr_done = false;
    auto kern = attn_kernel<DH>;
    if (!attr_done) {
        FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    char name[64];
    snprintf(name, sizeof name, "void fc::attn_kernel<%d>(fc::AttnParams)", DH);
    ProfScope ps(name, 4.0 * B * (double)p.N * (double)p.M * DH, 0.0, s);
    hipLaunchKernelGGL(kern, dim3((p.N + 127) / 128, B), dim3(256), lds, s, p);
    FC_HIP(hipGetLastError());
}

static void launch_attention_scaled(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                                    int B, int N, int n_stride_rows, int M, int m_stride_rows, int dh_pad, float qscale, hipStream_t s) {
    if (B <= 0 || N <= 0 || M <= 0) throw Error(FC_ERR_INVALID, "attention: empty problem");
    if ((ldq | ldk | ldv) % 4 != 0 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15))
        throw Error(FC_ERR_INVALID, "attention: q/k/v must be 16-byte aligned with pitches that are multiples of 4 floats");
    AttnParams p{q, ldq, k, ldk, v, ldv, out, ldo, N, n_stride_rows, M, m_stride_rows, qscale};
    switch (dh_pad) {
        case 32: launch_attn_dh<32>(p, B, s); break;
        case 64: launch_attn_dh<64>(p, B, s); break;
        case 128: launch_attn_dh<128>(p, B, s); break;
        default: throw Error(FC_ERR_UNSUPPORTED, "attention: inner dim (padded) must be 32, 64 or 128");
    }
}

void launch_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                      int B, int N, int n_stride_rows, int M, int m_stride_rows, int dh_pad, hipStream_t s) {
    launch_attention_scaled(q, ldq, k, ldk, v, ldv, out, ldo, B, N, n_stride_rows, M, m_stride_rows, dh_pad, 1.0f, s);
}

void launch_attention_op(const float* q, const float* k, const float* v, float* out, int B, int N, int M, int dh_pad, float scale,
                         hipStream_t s) {
    launch_attention_scaled(q, dh_pad, k, dh_pad, v, dh_pad, out, dh_pad, B, N, N, M, M, dh_pad,
                            scale * 1.4426950408889634f, s);
}

}  // namespace fc
