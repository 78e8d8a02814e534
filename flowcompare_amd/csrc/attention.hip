// Single-head cross attention of the flow's pre-conditioner, flash style.  Two kernels: attn16_kernel (split-fp16 operands on
// the 16-bit matrix cores, the default for head dims <= 64 inside an Fp16Guard scope; further down) and attn_kernel (fp32-input
// MFMA; head dim 128, the range-fallback pass, and the first build's kernel), described here:
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
#include "activations.h"
#include <cstdio>

namespace fc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

int g_attn_fp16 = 1;      // tuning knob (fc_debug_set 5): 0 keeps the fp32-input MFMA kernel
bool attention_fp16_enabled() { return g_attn_fp16 != 0; }

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
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);          // 0 on the first tile (m_run = -inf)
        float lt = 0.f;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[h2][r] - m_new);   // raw v_exp_f32: arguments are <= 0, results below 2^-126 may flush to 0
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

// =====================================================================================================================
// Split-fp16 attention (the default when the caller is inside an Fp16Guard scope, DH <= 64).
// Same flash structure, but every product runs on the 16-bit matrix cores with fp32-equivalent operands in the ONE-ACCUMULATOR limb
// form of spline_wide.hip (round 4; rounds 2-3 kept x = hi + lo'/2048 with separate main / cross-product accumulators): q, k, v are
// x * 16 = hi + lo with lo = rn16(x 16 - hi) unscaled, the probabilities p * 2^12 = hi + lo, a*b ~= ah*bh + ah*bl + al*bh in one fp32
// accumulator.  12 MFMAs (32 cycles) per 32x32 score block instead of 32 fp32-input MFMAs (64 cycles); against the two-accumulator
// form: no fold per score, 3 instead of 5 VALU per pair of probabilities in the limb split, 64 fewer accumulator registers.
//   * K and V of the layer are split ONCE by kv_limbs_kernel into row images [key][hi DH | lo DH] (the fp32 kernel above
//     re-reads them per 128-query workgroup; here 32 workgroups per scene would redo the same conversion);
//   * S^T = K Q^T: A = K rows (ds_read_b128 per limb), B = Q limbs held in registers for the whole kernel;
//   * P: the S^T accumulator has the query on the lane and 32 keys in its registers, so registers 8s..8s+7 split into limbs
//     ARE the A operand of k-step s of O += P V; element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3) of the block;
//   * V: B operand wants, per lane (column d), those same 8 keys: two ds_read_b64_tr_b16 (hardware transposing read of a
//     4-key x 16-column block per 16-lane group) on the row-major V image.  Row pitches: K 4*DH+16 B (conflict-free b128 reads),
//     V 4*DH+64 B (the 4 rows x 64 B a half-wave's transposed read touches land on 64 distinct banks).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short v4i16 __attribute__((__vector_size__(4 * sizeof(short))));

struct Attn16Params {
    const float* q; int ldq;
    const unsigned short* k16;      // [B * m_stride][2][DH] halves
    const unsigned short* v16;
    float* out; int ldo;
    int N, n_stride, M, m_stride;
    float qscale;
    int* ovf;
    // optional LayerNorm -> q fold (common.h EPI_LNQ): q holds the UN-normalised projection; the kernel applies
    // q * rsqrt(sum_b q_sumsq[b][row] * q_inv_width + 1e-5) + q_bias while it loads its query (no separate finalize pass)
    const float* q_sumsq; int q_slots; size_t q_pitch; float q_inv_width; const float* q_bias;
    int kv_pitch;                   // row pitch of the k16 / v16 images in 16-byte chunks (DH / 4 for the packed images)
    int c16;                        // 1: rows are slices of a GEMM's limb-image output ([16 columns: hi 16 | lo' 16] tiles, GemmEpi::C16)
    float* lse = nullptr;           // optional [B * n_stride]: natural-log sum-exp of every query's scaled scores (training: the backward reuses it)
};

// fp32 K / V columns of the projected context -> limb row images in the ONE-ACCUMULATOR form (common.h kOneAccActScale): x s = hi + lo with
// lo = rn16(x s - hi) unscaled; raises *ovf when |x s| leaves fp16's range.
__global__ __launch_bounds__(256) void kv_limbs_kernel(const float* k, int ldk, const float* v, int ldv, unsigned short* k16, unsigned short* v16,
                                                       long rows, int DH, int* ovf) {
    const int c4 = DH / 4;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    float amax = 0.f;
    if (t < rows * c4) {
        const long row = t / c4;
        const int c = (int)(t - row * c4) * 4;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const float4 x = *reinterpret_cast<const float4*>((which ? v + row * ldv : k + row * ldk) + c);
            const float xs[4] = {x.x, x.y, x.z, x.w};
            uint2 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(xs[e]));
            limb_split2s(xs[0], xs[1], kOneAccActScale, 1.0f, h.x, l.x);
            limb_split2s(xs[2], xs[3], kOneAccActScale, 1.0f, h.y, l.y);
            unsigned short* dst = (which ? v16 : k16) + row * 2 * DH + c;
            *reinterpret_cast<uint2*>(dst) = h;
            *reinterpret_cast<uint2*>(dst + DH) = l;
        }
    }
    if (!(amax * kOneAccActScale < 65504.0f)) atomicOr(ovf, 1);
}

template <int DH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn16_kernel(const Attn16Params p) {
    constexpr int KS = DH / 16, DT = DH / 32;
    constexpr int KP = 4 * DH + 16, VP = 4 * DH + 64;          // LDS row pitches in bytes
    constexpr int VOFF = 64 * KP;
    constexpr int STAGE = 64 * (KP + VP);
    constexpr int CPR = DH / 4;                                // 16-byte chunks per image row
    constexpr int NCH = 64 * CPR / 256;                        // chunks per thread, tile and image
    extern __shared__ float smem[];
    char* smc = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;

    // ---- Q limbs of this lane's query (B operand of S^T = K Q^T): element j of k-step s is q[16 s + 8 h + j]
    f16x8 qh[KS], ql[KS];
    float amax = 0.f;
    {
        int qi = q0 + li;
        qi = qi < p.N ? qi : p.N - 1;
        const size_t qrow = (size_t)b * p.n_stride + qi;
        const float* qp = p.q + qrow * p.ldq + 8 * lh;
        float rstd = 1.0f;
        if (p.q_sumsq) {
            float ss = 0.f;
            for (int sb = 0; sb < p.q_slots; ++sb) ss += p.q_sumsq[(size_t)sb * p.q_pitch + qrow];
            rstd = 1.0f / sqrtf(ss * p.q_inv_width + 1e-5f);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 t0 = *reinterpret_cast<const float4*>(qp + 16 * s), t1 = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
            const float xs[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float bq = p.q_bias ? p.q_bias[16 * s + 8 * lh + e] : 0.f;
                const float x = (xs[e] * rstd + bq) * (p.qscale * kOneAccActScale);
                amax = fmaxf(amax, fabsf(x));
                qh[s][e] = (_Float16)x;
                ql[s][e] = (_Float16)(x - (float)qh[s][e]);
            }
        }
    }
    if (!(amax < 65504.0f)) atomicOr(p.ovf, 1);

    // ONE accumulator per output (round 4, the limb form of spline_wide.hip): q, k, v are held as hi + lo of x * 16 with lo UNSCALED and the
    // probabilities as hi + lo of p * 2^12, so that the three limb products hi.hi + hi.lo + lo.hi land at one scale in one fp32 accumulator:
    // no cross-product accumulator sets (64 registers) and no fold of S per score.  The scales cancel exactly: scores are read as
    // S / 256 inside the exponent's fma, 2^12 rides in the exponent's offset (row sums carry it too), the output divides by 16 l.
    floatx16 om[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) om[d][r] = 0.f;
    // LAZY reference (round 4, second half): m_run follows the running maximum of the true scores only when a tile's maximum exceeds it by more
    // than TAU (log2 domain), so p = exp2(s - m_run) <= 2^TAU and the probabilities are held as hi + lo of p 2^12 <= 2^15 (fp16's range).  The
    // result is the same quotient -- numerator and row sum carry the same reference -- but the rescaling of O (16 cross-lane reads + 32
    // multiplies per lane, taken in ~60 % of the tiles of a 4096-key scene with the exact running maximum) happens a handful of times per query.
    constexpr float S_INV = 1.0f / (kOneAccActScale * kOneAccActScale), P_EXP = 12.0f, TAU = 3.0f;
    float m_run = -INFINITY, l_run = 0.f;                      // m_run: the reference (log2 domain, within TAU of the running maximum of the true scores); l_run: sum of p 2^12

    // ---- staging: plain 16-byte copies of the limb images (whole-vector register values: arrays went through scratch)
    typedef unsigned int u32xs __attribute__((ext_vector_type(4 * NCH)));
    u32xs rk, rv;
    const uint4* kb = reinterpret_cast<const uint4*>(p.k16) + (size_t)b * p.m_stride * p.kv_pitch;
    const uint4* vb = reinterpret_cast<const uint4*>(p.v16) + (size_t)b * p.m_stride * p.kv_pitch;
#define FC_GLOAD(T_)                                                                          \
    _Pragma("unroll") for (int i = 0; i < NCH; ++i) {                                         \
        const int c_ = tid + 256 * i, kr_ = c_ / CPR, part_ = c_ - kr_ * CPR;                 \
        int key_ = (T_) * 64 + kr_;                                                           \
        key_ = key_ < p.M ? key_ : p.M - 1; /* clamped rows are masked to -inf below */       \
        /* chunk part_ of the [hi DH | lo' DH] row = limb part_/(CPR/2), columns 8 w .. 8 w + 7; in a C16 slice that is chunk   */ \
        /* (w >> 1) * 4 + limb * 2 + (w & 1)                                                                                    */ \
        const int w_ = part_ % (CPR / 2), lb_ = part_ / (CPR / 2);                             \
        const int sp_ = p.c16 ? (w_ >> 1) * 4 + lb_ * 2 + (w_ & 1) : part_;                    \
        const uint4 a_ = kb[(size_t)key_ * p.kv_pitch + sp_], b_ = vb[(size_t)key_ * p.kv_pitch + sp_];  \
        rk[4 * i] = a_.x; rk[4 * i + 1] = a_.y; rk[4 * i + 2] = a_.z; rk[4 * i + 3] = a_.w;     \
        rv[4 * i] = b_.x; rv[4 * i + 1] = b_.y; rv[4 * i + 2] = b_.z; rv[4 * i + 3] = b_.w;     \
    }
#define FC_LSTORE(ST_)                                                                        \
    _Pragma("unroll") for (int i = 0; i < NCH; ++i) {                                         \
        const int c_ = tid + 256 * i, kr_ = c_ / CPR, part_ = c_ - kr_ * CPR;                 \
        *reinterpret_cast<uint4*>(smc + (ST_) * STAGE + kr_ * KP + part_ * 16) =              \
            make_uint4(rk[4 * i], rk[4 * i + 1], rk[4 * i + 2], rk[4 * i + 3]);               \
        *reinterpret_cast<uint4*>(smc + (ST_) * STAGE + VOFF + kr_ * VP + part_ * 16) =       \
            make_uint4(rv[4 * i], rv[4 * i + 1], rv[4 * i + 2], rv[4 * i + 3]);               \
    }

    // transposed-read address of this lane inside a [4 keys][16 columns] block: lane 4q+p of a 16-lane group supplies row q,
    // columns 4p..4p+3; the group's 16 columns are 16*((lane>>4)&1) .. +15 of the 32-column block, its keys start at 4*lh
    const int tr_off = (4 * lh + ((lane & 15) >> 2)) * VP + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    const int ntiles = (p.M + 63) / 64;
    FC_GLOAD(0)
    FC_LSTORE(0)
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int tn = t + 1 < ntiles ? t + 1 : t;          // the last iteration re-loads its own tile: branch-free loop
        FC_GLOAD(tn)
        const char* sK = smc + (t & 1) * STAGE;
        const char* sV = sK + VOFF;

        // ---- S^T = K Q^T for the two 32-key halves of the tile
        floatx16 s[2];                                          // 256 x the scores
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[h2][r] = 0.f;
            const char* kr = sK + (32 * h2 + li) * KP + 16 * lh;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 kh = *reinterpret_cast<const f16x8*>(kr + 32 * ks);
                const f16x8 kl = *reinterpret_cast<const f16x8*>(kr + 2 * DH + 32 * ks);
                s[h2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s[h2], 0, 0, 0);
                s[h2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s[h2], 0, 0, 0);
                s[h2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s[h2], 0, 0, 0);
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
        const float mts = mt * S_INV;
        const float m_new = mts > m_run + TAU ? mts : m_run;                 // (first tile: m_run = -inf; a fully masked tail keeps the reference)
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);          // 0 on the first tile (m_run = -inf), 1 while the reference stands
        const float off = P_EXP - m_new;                                     // p 2^12 = exp2(S / 256 - m + 12)
        float lt = 0.f;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[h2][r], S_INV, off));   // raw v_exp_f32: arguments are <= 15, tiny results may flush to 0
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
                for (int d = 0; d < DT; ++d) om[d][r] *= ar;
            }
        }
        // ---- O += P V, 16 keys per MFMA k-step: A = limbs of P registers 8 s2 .. 8 s2 + 7, B = V via transposed reads
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8 ph, pl;
                {
                    const float x8[8] = {s[h2][8 * s2], s[h2][8 * s2 + 1], s[h2][8 * s2 + 2], s[h2][8 * s2 + 3],
                                         s[h2][8 * s2 + 4], s[h2][8 * s2 + 5], s[h2][8 * s2 + 6], s[h2][8 * s2 + 7]};
                    limb_split8_unscaled(x8, ph, pl);             // (3 VALU per pair of probabilities: this kernel is VALU-bound on exactly this)
                }
                const char* vr = sV + (32 * h2 + 16 * s2) * VP + tr_off;
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    // (assembled with one shufflevector + whole-vector bit_cast: element-wise extraction of the read's result made hipcc
                    //  emit v_perm/v_mov sequences that duplicated its first dword)
#define FC_TR(OFF_) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(vr + (OFF_)))
                    const f16x8 vh = __builtin_bit_cast(f16x8, __builtin_shufflevector(FC_TR(64 * d), FC_TR(8 * VP + 64 * d), 0, 1, 2, 3, 4, 5, 6, 7));
                    const f16x8 vl = __builtin_bit_cast(f16x8, __builtin_shufflevector(FC_TR(64 * d + 2 * DH), FC_TR(8 * VP + 64 * d + 2 * DH),
                                                                                      0, 1, 2, 3, 4, 5, 6, 7));
#undef FC_TR
                    om[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, om[d], 0, 0, 0);
                    om[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, om[d], 0, 0, 0);
                    om[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, om[d], 0, 0, 0);
                }
            }
        }
        FC_LSTORE((t + 1) & 1)
        __syncthreads();
    }
#undef FC_GLOAD
#undef FC_LSTORE

    if (p.lse && lane < 32 && q0 + lane < p.N)                   // scores are in the log2 domain (qscale carries log2 e); l_run carries 2^12 and the reference m_run
        p.lse[(size_t)b * p.n_stride + q0 + lane] = (m_run - P_EXP + __builtin_amdgcn_logf(l_run)) * 0.6931471805599453f;
    // ---- normalise and store: O rows are queries (r&3)+8(r>>2)+4h of this wave, columns d = 32*dt + lane&31
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float lr = __shfl(l_run, qr, 64);
        const int qi = q0 + qr;
        if (qi < p.N) {
            float* op = p.out + ((size_t)b * p.n_stride + qi) * p.ldo + li;
#pragma unroll
            for (int d = 0; d < DT; ++d) op[32 * d] = om[d][r] / (lr * kOneAccActScale);      // (v carries 16, p and l carry 2^12)
        }
    }
}

// (Round 4 also tried this attention as ONE 512-thread workgroup of two wave groups that run S / softmax / P V one slot apart -- the
//  explicit matrix | vector alternation of the fused spline GEMM.  Bit-identical and no faster, 26.6 against 26.2 ms per C2 step: the
//  softmax is VALU issue time that two waves per SIMD need more of than of MFMA time under any interleaving.  The kernel was removed when
//  this one went to a single accumulator; its measurements stay in DESIGN.md section 9 and profiles/r04q_*, r04r_attn_stamps.log.)

template <int DH>
static void launch_attn16_dh(const Attn16Params& p, int B, hipStream_t s) {
    constexpr size_t lds = 2 * 64 * (size_t)(8 * DH + 80);
    static PerDeviceOnce attr_once;
    auto kern = attn16_kernel<DH>;
    attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); return 0; });
    char name[64];
    snprintf(name, sizeof name, "void fc::attn16_kernel<%d>(fc::Attn16Params)", DH);
    ProfScope ps(name, 4.0 * B * (double)p.N * (double)p.M * DH, 0.0, s);
    hipLaunchKernelGGL(kern, dim3((p.N + 127) / 128, B), dim3(256), lds, s, p);
    FC_HIP(hipGetLastError());
}

template <int DH>
static void launch_attn_dh(const AttnParams& p, int B, hipStream_t s) {
    constexpr size_t lds = 2 * 2 * 64 * (size_t)(DH + 4) * sizeof(float);
    static PerDeviceOnce attr_once;
    auto kern = attn_kernel<DH>;
    attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); return 0; });
    char name[64];
    snprintf(name, sizeof name, "void fc::attn_kernel<%d>(fc::AttnParams)", DH);
    ProfScope ps(name, 4.0 * B * (double)p.N * (double)p.M * DH, 0.0, s);
    hipLaunchKernelGGL(kern, dim3((p.N + 127) / 128, B), dim3(256), lds, s, p);
    FC_HIP(hipGetLastError());
}

static void launch_attention_scaled(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                                    int B, int N, int n_stride_rows, int M, int m_stride_rows, int dh_pad, float qscale, void* limb_ws,
                                    hipStream_t s, const unsigned short* k_c16 = nullptr, const unsigned short* v_c16 = nullptr, int c16_pitch = 0,
                                    const AttnLnq* lnq = nullptr, float* lse = nullptr, bool* lse_written = nullptr) {
    if (B <= 0 || N <= 0 || M <= 0) throw Error(FC_ERR_INVALID, "attention: empty problem");
    if (k_c16) {
        // K / V arrive as slices of the projection GEMM's limb-image output: no fp32 K / V, no conversion pass
        int* flag16 = gemm_fp16_flag();
        if (!flag16 || dh_pad > 64 || !v_c16 || c16_pitch <= 0 || (ldq % 4) != 0)
            throw Error(FC_ERR_INVALID, "attention: limb-image K / V need a guard scope, head dim <= 64 and a pitch");
        Attn16Params p{q, ldq, k_c16, v_c16, out, ldo, N, n_stride_rows, M, m_stride_rows, qscale, flag16, lnq ? lnq->sumsq : nullptr,
                       lnq ? lnq->slots : 0, lnq ? lnq->pitch : 0, lnq ? lnq->inv_width : 0.f, lnq ? lnq->bias : nullptr, c16_pitch, 1};
        if (dh_pad == 32) launch_attn16_dh<32>(p, B, s); else launch_attn16_dh<64>(p, B, s);
        return;
    }
    if ((ldq | ldk | ldv) % 4 != 0 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15))
        throw Error(FC_ERR_INVALID, "attention: q/k/v must be 16-byte aligned with pitches that are multiples of 4 floats");
    int* flag = gemm_fp16_flag();
    if (flag && limb_ws && dh_pad <= 64 && g_attn_fp16) {
        // split-fp16 path (needs the caller's Fp16Guard scope for its range check and limb_ws for the K/V limb images)
        const long rows = (long)(B - 1) * m_stride_rows + M;
        unsigned short* k16 = (unsigned short*)limb_ws;
        unsigned short* v16 = k16 + (size_t)rows * 2 * dh_pad;
        {
            ProfScope ps("fc::kv_limbs_kernel", 0.0, (double)rows * dh_pad * 16.0, s);
            const long n = rows * (dh_pad / 4);
            hipLaunchKernelGGL(kv_limbs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, k, ldk, v, ldv, k16, v16, rows, dh_pad, flag);
            FC_HIP(hipGetLastError());
        }
        Attn16Params p{q, ldq, k16, v16, out, ldo, N, n_stride_rows, M, m_stride_rows, qscale, flag, nullptr, 0, 0, 0.f, nullptr, dh_pad / 4, 0};
        p.lse = lse;
        if (lse_written) *lse_written = lse != nullptr;
        if (dh_pad == 32) launch_attn16_dh<32>(p, B, s); else launch_attn16_dh<64>(p, B, s);
        return;
    }
    AttnParams p{q, ldq, k, ldk, v, ldv, out, ldo, N, n_stride_rows, M, m_stride_rows, qscale};
    switch (dh_pad) {
        case 32: launch_attn_dh<32>(p, B, s); break;
        case 64: launch_attn_dh<64>(p, B, s); break;
        case 128: launch_attn_dh<128>(p, B, s); break;
        default: throw Error(FC_ERR_UNSUPPORTED, "attention: inner dim (padded) must be 32, 64 or 128");
    }
}

size_t attention_limb_ws_bytes(long kv_rows, int dh_pad) { return dh_pad <= 64 ? (size_t)kv_rows * dh_pad * 8 : 0; }

void launch_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                      int B, int N, int n_stride_rows, int M, int m_stride_rows, int dh_pad, void* limb_ws, hipStream_t s) {
    launch_attention_scaled(q, ldq, k, ldk, v, ldv, out, ldo, B, N, n_stride_rows, M, m_stride_rows, dh_pad, 1.0f, limb_ws, s);
}

// K | V of this layer as columns [col0, col0 + 2 * dh_pad) of a GEMM limb-image output with n_pad columns per row (GemmEpi::C16)
void launch_attention_c16(const float* q, int ldq, const unsigned short* kv_c16, int n_pad, int col0, float* out, int ldo, int B, int N,
                          int n_stride_rows, int M, int m_stride_rows, int dh_pad, hipStream_t s, const AttnLnq* lnq) {
    if (col0 % 16 != 0 || n_pad % 16 != 0 || dh_pad % 16 != 0) throw Error(FC_ERR_INVALID, "attention: limb-image slices must start on 16-column tiles");
    const unsigned short* kp = kv_c16 + (size_t)(col0 / 16) * 32;
    const unsigned short* vp = kp + (size_t)(dh_pad / 16) * 32;
    launch_attention_scaled(q, ldq, nullptr, 4, nullptr, 4, out, ldo, B, N, n_stride_rows, M, m_stride_rows, dh_pad, 1.0f, nullptr, s, kp, vp, n_pad / 4, lnq);
}

// training path (train_attention.hip): strided q / k / v (columns of wider panels), explicit softmax scale
// lse (optional, [B * N]): filled by the split-fp16 kernel only; the return value says whether it was
bool launch_attention_scaled_op(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, int B, int N, int M,
                                int dh_pad, float scale, void* limb_ws, hipStream_t s, float* lse) {
    bool written = false;
    launch_attention_scaled(q, ldq, k, ldk, v, ldv, out, ldo, B, N, N, M, M, dh_pad, scale * 1.4426950408889634f, limb_ws, s, nullptr, nullptr, 0,
                            nullptr, lse, &written);
    return written;
}

void launch_attention_op(const float* q, const float* k, const float* v, float* out, int B, int N, int M, int dh_pad, float scale,
                         void* limb_ws, hipStream_t s) {
    launch_attention_scaled(q, dh_pad, k, dh_pad, v, dh_pad, out, dh_pad, B, N, N, M, M, dh_pad,
                            scale * 1.4426950408889634f, limb_ws, s);
}

}  // namespace fc
