// Fused pre-attention MLP -> LayerNorm -> q projection of the attention pre-conditioner
// (models/cif_block.py:14-20: x1 -> pre_attention_mlp; models/perceiver.py:18-35: PreNorm; :104-106: to_q), one launch per layer
// instead of six (in_layer, two hidden layers, out_layer, LayerNorm, q projection).
//
// A workgroup owns 64 point rows for the whole chain; the 64 x 256 activation tile never leaves the CU:
//   * it lives in LDS as the fp16 limb image the next layer's MFMAs read ([row][k/16][hi 16 | lo' 16], split-fp16 operands of
//     gemm.hip / DESIGN.md section 3), 65 KB;
//   * each layer streams its weight limb image (PackedLinear.W2) through a double-buffered 32-k LDS stage (2 x 36 KB) with a
//     two-deep register prefetch, 12 MFMAs per wave and barrier (8 waves: 2 row blocks x 4 column blocks of 32 x 64);
//   * the epilogue (bias, residual kept in registers, exact-erf GELU, limb split) writes the tile back in place;
//   * LayerNorm statistics are reduced across the 4 column waves through LDS, the normalised tile feeds the 256 -> 64 q
//     projection (gamma / beta / softmax scale / log2 e folded into it at create), and only q [rows, 64] is written to HBM.
// Removes per layer: 4 activation round trips through HBM (67 MB written + read each), the LayerNorm pass, 5 launches.
// Shapes: hidden width = attention input width = 256 exactly, q width 64, input width a multiple of 32 up to 256
// (the engine falls back to the separate kernels otherwise, and always on the bf16-limb range-fallback pass).
#include "common.h"
#include "activations.h"

namespace fc {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct PreMlpLayer {
    const unsigned short* W2;   // [n_alloc][K_pad/16][2][16]
    const float* bias;
    int K_pad;
};
struct PreMlpParams {
    const float* x; int ldx;            // input rows (first in_k columns)
    PreMlpLayer in, mid0, mid1, out, q; // q: LN-folded projection 256 -> 64
    int act;
    float* qout; int ldq;
    int rows;                           // rows allocated (multiple of 64)
    int* ovf;
};

constexpr int PM_ROWS = 64, PM_H = 256, PM_NT = 512;
constexpr int PM_APITCH = (PM_H / 16) * 64 + 16;           // 1040 B per activation row
constexpr int PM_WPITCH = 144;                             // [hi 32 k | lo' 32 k] + 16 B pad per weight row and stage
constexpr int PM_ACT_BYTES = PM_ROWS * PM_APITCH;          // 66560
constexpr int PM_WSTAGE = PM_H * PM_WPITCH;                // 36864
constexpr int PM_RED_OFF = PM_ACT_BYTES + 2 * PM_WSTAGE;   // [64 rows][4] floats for the LayerNorm reductions
constexpr int PM_LDS = PM_RED_OFF + PM_ROWS * 4 * 4;

// One dense layer on the resident tile: acc[j] (+ corr) = tile(64 x K) * W(NOUT x K)^T for this wave's 32 x (32*TNW) block.
// NOUT = 256 (TNW = 2, all 8 waves) or 64 (TNW = 1, waves with wc < 2).
// (Reading the weight fragments straight from the L2-resident limb image -- no LDS staging, no barrier inside a layer -- was
//  measured 60 % slower: 8 waves x 4 KB of 16-byte-per-lane loads per k-tile exceed what the CU's vector L1 delivers.)
template <int NOUT>
__device__ __forceinline__ void pm_gemm(const PreMlpLayer& L, char* smc, int tid, int li, int lh, int wr, int wc, floatx16 (&accm)[2], floatx16 (&accc)[2]) {
    constexpr int TNW = NOUT == 256 ? 2 : 1;
    constexpr int CHUNKS = NOUT * 8;                        // 16-byte chunks per 32-k stage
    constexpr int NCH = (CHUNKS + PM_NT - 1) / PM_NT;       // per thread (4 or 1)
    typedef unsigned int u32xs __attribute__((ext_vector_type(4 * NCH)));
    u32xs r0, r1;
    const int KT16 = L.K_pad / 16, KS = L.K_pad / 32;
    char* wst = smc + PM_ACT_BYTES;
#define PM_GLOAD(R_, S_)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < NCH; ++i) {                                                             \
        int c_ = tid + PM_NT * i;                                                                                 \
        c_ = c_ < CHUNKS ? c_ : CHUNKS - 1;                                                                       \
        const int row_ = c_ >> 3, part_ = c_ & 7, sub_ = part_ >> 2, q2_ = part_ & 3;                             \
        const uint4 t_ = *reinterpret_cast<const uint4*>(L.W2 + ((size_t)row_ * KT16 + 2 * (S_) + sub_) * 32 + q2_ * 8); \
        R_[4 * i] = t_.x; R_[4 * i + 1] = t_.y; R_[4 * i + 2] = t_.z; R_[4 * i + 3] = t_.w;                       \
    }
#define PM_LSTORE(R_, ST_)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NCH; ++i) {                                                             \
        const int c_ = tid + PM_NT * i, row_ = c_ >> 3, part_ = c_ & 7, sub_ = part_ >> 2, q2_ = part_ & 3;       \
        if (CHUNKS % PM_NT == 0 || c_ < CHUNKS)                                                                   \
            *reinterpret_cast<uint4*>(wst + (ST_) * PM_WSTAGE + row_ * PM_WPITCH + (q2_ >> 1) * 64 + sub_ * 32 + (q2_ & 1) * 16) = \
                make_uint4(R_[4 * i], R_[4 * i + 1], R_[4 * i + 2], R_[4 * i + 3]);                               \
    }
#define PM_MMA(ST_, S_)                                                                                           \
    if (NOUT == 256 || wc < 2) {                                                                                  \
        _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) {                                                     \
            const char* pa = smc + (32 * wr + li) * PM_APITCH + (2 * (S_) + sub) * 64 + lh * 16;                  \
            const f16x8 ah = *reinterpret_cast<const f16x8*>(pa), al = *reinterpret_cast<const f16x8*>(pa + 32); \
            _Pragma("unroll") for (int j = 0; j < TNW; ++j) {                                                     \
                const char* pb = wst + (ST_) * PM_WSTAGE + (32 * TNW * wc + 32 * j + li) * PM_WPITCH + sub * 32 + lh * 16; \
                const f16x8 bh = *reinterpret_cast<const f16x8*>(pb), bl = *reinterpret_cast<const f16x8*>(pb + 64); \
                accm[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accm[j], 0, 0, 0);                       \
                accc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accc[j], 0, 0, 0);                       \
                accc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accc[j], 0, 0, 0);                       \
            }                                                                                                     \
        }                                                                                                         \
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accm[j][r] = 0.f; accc[j][r] = 0.f; }
    // stage s & 1 of LDS holds k-step s; one register set holds step s + 1, the other s + 2 (two-deep prefetch, as in gemm.hip)
    PM_GLOAD(r0, 0)
    PM_LSTORE(r0, 0)
    if (KS > 1) { PM_GLOAD(r1, 1) }
    __syncthreads();
    for (int s = 0; s < KS; s += 2) {
        const int s2 = s + 2 < KS ? s + 2 : KS - 1, s3 = s + 3 < KS ? s + 3 : KS - 1;
        PM_GLOAD(r0, s2)
        PM_MMA(0, s)
        if (s + 1 < KS) {
            PM_LSTORE(r1, 1)
            __syncthreads();
            PM_GLOAD(r1, s3)
            PM_MMA(1, s + 1)
        }
        if (s + 2 < KS) { PM_LSTORE(r0, 0) }          // (an odd tail step has nothing left to store: stage 0 may still be read)
        __syncthreads();
    }
#undef PM_GLOAD
#undef PM_LSTORE
#undef PM_MMA
}

// v (this lane's 2 x 16 block values: column 64 wc + 32 j + li, rows 32 wr + (r&3) + 8 (r>>2) + 4 lh) -> limb image in the tile.
// Adjacent lanes (columns c, c+1) pair their halves so that every lane writes one 32-bit word per element.
__device__ __forceinline__ void pm_store_tile(char* smc, const float (&v)[2][16], int li, int lh, int wr, int wc, float& amax) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = 64 * wc + 32 * j + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = v[j][r];
            amax = fmaxf(amax, fabsf(x));
            const _Float16 h = (_Float16)x;
            const _Float16 l = (_Float16)((x - (float)h) * 2048.0f);
            const unsigned hb = __builtin_bit_cast(unsigned short, h), lb = __builtin_bit_cast(unsigned short, l);
            const unsigned mine = (li & 1) ? lb : hb, give = (li & 1) ? hb : lb;      // even lane keeps hi, odd lane keeps lo'
            const unsigned got = __shfl_xor(give, 1, 64);                              // even: neighbour's hi; odd: neighbour's lo'
            const unsigned word = (li & 1) ? (got | (mine << 16)) : (mine | (got << 16));
            const int row = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int c0 = col & ~1;                                                   // the pair's even column
            *reinterpret_cast<unsigned*>(smc + row * PM_APITCH + (c0 >> 4) * 64 + ((li & 1) ? 32 : 0) + (c0 & 15) * 2) = word;
        }
    }
}

__global__ __launch_bounds__(PM_NT) __attribute__((amdgpu_waves_per_eu(2))) void premlp_kernel(const PreMlpParams p) {
    extern __shared__ float smem[];
    char* smc = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5, wr = wave >> 2, wc = wave & 3;
    const int row0 = blockIdx.x * PM_ROWS;
    float amax = 0.f;

    // ---- input rows -> limb image (columns >= in.K_pad are never read by the in_layer)
    {
        const int c4n = p.in.K_pad / 4;
        for (int t = tid; t < PM_ROWS * c4n; t += PM_NT) {
            const int row = t / c4n, c = (t - row * c4n) * 4;
            const float4 x = *reinterpret_cast<const float4*>(p.x + (size_t)(row0 + row) * p.ldx + c);
            const float xs[4] = {x.x, x.y, x.z, x.w};
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                amax = fmaxf(amax, fabsf(xs[e]));
                h[e] = (_Float16)xs[e];
                l[e] = (_Float16)((xs[e] - (float)h[e]) * 2048.0f);
            }
            char* dst = smc + row * PM_APITCH + (c >> 4) * 64 + (c & 15) * 2;
            *reinterpret_cast<f16x4*>(dst) = h;
            *reinterpret_cast<f16x4*>(dst + 32) = l;
        }
    }
    __syncthreads();

    floatx16 accm[2], accc[2];
    float keep[2][16];                                     // h0: residual of the second hidden layer (models/nets.py:24-29)
    float v[2][16];

    // ---- in_layer, hidden layer 0 (keep = x; x = act(W x)), hidden layer 1 (x = act(keep + W x)), out_layer (no activation)
#pragma unroll 1
    for (int layer = 0; layer < 4; ++layer) {
        const PreMlpLayer& L = layer == 0 ? p.in : layer == 1 ? p.mid0 : layer == 2 ? p.mid1 : p.out;
        pm_gemm<256>(L, smc, tid, li, lh, wr, wc, accm, accc);      // ends with a barrier: every wave is done reading the tile
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float b = L.bias[64 * wc + 32 * j + li];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = accm[j][r] + accc[j][r] * (1.0f / 2048.0f) + b;
                if (layer == 2) t += keep[j][r];
                v[j][r] = layer == 3 ? t : act_apply(t, p.act);
                if (layer == 0) keep[j][r] = v[j][r];
            }
        }
        if (layer < 3) {
            pm_store_tile(smc, v, li, lh, wr, wc, amax);
            __syncthreads();
        }
    }

    // ---- LayerNorm over the 256 columns of each row (biased variance, eps 1e-5; gamma / beta live in the q projection)
    float* red = reinterpret_cast<float*>(smc + PM_RED_OFF);
    float mean[16], rstd[16];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float d = pass == 0 ? v[j][r] : v[j][r] - mean[r];
                t += pass == 0 ? d : d * d;
            }
            t = half_wave_sum(t);
            if (li == 0) red[(32 * wr + (r & 3) + 8 * (r >> 2) + 4 * lh) * 4 + wc] = t;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float4 q4 = *reinterpret_cast<const float4*>(red + (32 * wr + (r & 3) + 8 * (r >> 2) + 4 * lh) * 4);
            const float tot = (q4.x + q4.y) + (q4.z + q4.w);
            if (pass == 0) mean[r] = tot * (1.0f / PM_H); else rstd[r] = 1.0f / sqrtf(tot * (1.0f / PM_H) + 1e-5f);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[j][r] = (v[j][r] - mean[r]) * rstd[r];
    pm_store_tile(smc, v, li, lh, wr, wc, amax);
    __syncthreads();

    // ---- q projection 256 -> 64 (4 of the 8 waves multiply; all of them stage the weights)
    pm_gemm<64>(p.q, smc, tid, li, lh, wr, wc, accm, accc);
    if (wc < 2) {
        const int col = 32 * wc + li;
        const float b = p.q.bias ? p.q.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
            p.qout[(size_t)row * p.ldq + col] = accm[0][r] + accc[0][r] * (1.0f / 2048.0f) + b;
        }
    }
    if (amax >= 65504.0f) atomicOr(p.ovf, 1);
}

int g_premlp_fused = 0;       // tuning knob (fc_debug_set 8).  OFF by default: this kernel beat the six launches it replaces by 8 % when it was
                              // written, but the eight-wave GEMM tile and the cheaper GELU then made the separate kernels 2 % faster
                              // end to end (one 64-row workgroup per CU re-streams every layer's weights from L2: 16 % MFMA busy)

static bool premlp_layer_ok(const PackedLinear& L, int n, int kmax) {
    return L.W2 != nullptr && L.bias != nullptr && L.nseg == 1 && L.N_pad == n && L.n_true == n && L.K_pad % 32 == 0 && L.K_pad <= kmax &&
           L.n_alloc >= n;
}

// true when the fused kernel can run this pre-conditioner (shapes above, fp16 limb images present, inside a guard scope)
bool premlp_fusable(const PackedLinear& in, const std::vector<PackedLinear>& mid, const PackedLinear& out, const PackedLinear& q) {
    return gemm_fp16_flag() != nullptr && g_premlp_fused && mid.size() == 2 && premlp_layer_ok(in, PM_H, PM_H) &&
           premlp_layer_ok(mid[0], PM_H, PM_H) && mid[0].K_pad == PM_H && premlp_layer_ok(mid[1], PM_H, PM_H) && mid[1].K_pad == PM_H &&
           premlp_layer_ok(out, PM_H, PM_H) && out.K_pad == PM_H && out.k_true == PM_H && q.W2 != nullptr && q.nseg == 1 && q.N_pad == 64 &&
           q.K_pad == PM_H && q.k_true == PM_H;
}

void launch_premlp(const float* x, int ldx, const PackedLinear& in, const std::vector<PackedLinear>& mid, const PackedLinear& out,
                   const PackedLinear& q, int act, float* qout, int ldq, int rows_alloc, int rows_valid, hipStream_t s) {
    if (rows_alloc % PM_ROWS != 0 || ldx % 4 != 0 || ldx < in.K_pad || ((uintptr_t)x & 15))
        throw Error(FC_ERR_INVALID, "premlp: rows must be padded to 64, input pitch to 4 floats");
    static bool attr_done = false;
    if (!attr_done) {
        FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(premlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PM_LDS));
        attr_done = true;
    }
    PreMlpParams p{};
    p.x = x; p.ldx = ldx;
    auto L = [](const PackedLinear& l) { return PreMlpLayer{l.W2, l.bias, l.K_pad}; };
    p.in = L(in); p.mid0 = L(mid[0]); p.mid1 = L(mid[1]); p.out = L(out); p.q = L(q);
    p.act = act; p.qout = qout; p.ldq = ldq; p.rows = rows_alloc; p.ovf = gemm_fp16_flag();
    const double rv = rows_valid > 0 ? rows_valid : rows_alloc;
    const double flops = 2.0 * rv * ((double)in.k_true * PM_H + 3.0 * PM_H * PM_H + (double)PM_H * (q.n_true ? q.n_true : 64));
    ProfScope ps("fc::premlp_kernel(fc::PreMlpParams)", flops, 0.0, s);
    hipLaunchKernelGGL(premlp_kernel, dim3(rows_alloc / PM_ROWS), dim3(PM_NT), PM_LDS, s, p);
    FC_HIP(hipGetLastError());
}

}  // namespace fc
