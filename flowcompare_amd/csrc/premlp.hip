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
#include <type_traits>

#include "common.h"
#include <cstdio>
#include "activations.h"

namespace fc {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef __attribute__((address_space(3))) char pm_lds_char;
typedef const __attribute__((address_space(1))) char pm_glb_char;

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
    unsigned long long* stamps;         // diagnostic knob 20 = 3: 16 x u64 per workgroup (s_memtime at the layer boundaries; wall clock in 14 / 15)
    float* keep_ws;                     // row-resident kernel: rows x 256 floats of scratch (the second hidden layer's residual, parked as limb fragments)
    int* ovf;
    // row-resident kernel with the PREVIOUS layer's folded ActNorm + permuter matrix as a pre-layer (premlp_rows_kernel<.., .., NLU > 0>):
    // z = lu(xprev) is written to xnext (the latent this layer works on) and its first in.K_pad columns feed the in_layer from registers
    PreMlpLayer lu;
    const float* xprev; int ldxp;
    float* xnext; int ldxn;
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

#ifdef FC_DEV_VARIANTS      // (the LDS-tile form lost its A/B against the row-resident kernel below: developer builds only)
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
#endif      // FC_DEV_VARIANTS

// =====================================================================================================================================
// Row-resident variant (knob 8 = 2): the chain's activations never leave the REGISTERS.
//   The product is transposed (weights as the MFMA's A operand, points as B) on the 16x16x32 shape: a wave owns 16 points, a point is
//   spread over the four lanes n, n + 16, n + 32, n + 48 (lane row kg = lane >> 4 supplies in-features 32 s + 8 kg + 0..7 of k step s).
//   Per lane: the layer's input as B-operand fragments (8 k steps x [hi | lo'] x 8 fp16 = 64 registers), the output being assembled in
//   the same form (64, sharing registers with the residual of the second hidden layer as that is consumed), one 32-feature accumulator
//   chunk (16) -- about 200 registers, so eight waves (128 rows) run two per SIMD.  Between layers nothing is stored anywhere: an
//   output chunk goes from the accumulator's order (lane row kg holds features 16 mb + 4 kg + i) to the operand order (features
//   8 kg + e of the chunk) with one `v_permlane32_swap` + one `v_permlane16_swap` per register pair, gets bias / residual / activation
//   and is split into limbs in place.  LayerNorm is an in-lane sum plus two cross-row adds.
//   Only the weights move: a chunk = 32 out-features x K as one LDS stage (32 KB at K = 256: a 1 KiB DMA piece is one weight row;
//   16-byte chunks XOR-swizzled by (row & 15) on the source address and on the read), two stages, 48 MFMAs per wave and barrier.
//   The 1.3 MB of weights are streamed once per 128 rows (the LDS-tile kernel above: per 64) and, with no activation tile in LDS, a
//   stage is 4x as long per barrier.
constexpr int PR_ROWS = 128, PR_NT = 512, PR_CH = 32;
constexpr int PR_BUF = PR_CH * PM_H * 4;                    // 32 KB: one chunk of 32 weight rows at K = 256
constexpr int PR_BIAS_OFF = 2 * PR_BUF;                     // [5][256] floats: in, mid0, mid1, out, q
constexpr int PR_LDS = PR_BIAS_OFF + 5 * PM_H * 4;
// with the ActNorm + LU pre-layer (NLU chunks of 32 outputs, K = 32 NLU): a chunk is 32 rows x 32 NLU x 4 B, its bias follows the five others
constexpr int pr_buf(int nlu) { return nlu * 32 > PM_H ? PR_CH * nlu * 32 * 4 : PR_BUF; }
constexpr int pr_lds(int nlu) { return 2 * pr_buf(nlu) + 5 * PM_H * 4 + nlu * 32 * 4; }
typedef float floatx4 __attribute__((ext_vector_type(4)));

// rows of 16 lanes (a0,a1,a2,a3 | b0,b1,b2,b3):  swap32 -> a = (a0,a1,b0,b1), b = (a2,a3,b2,b3) ;  swap16 -> a = (a0,b0,a2,b2), b = (a1,b1,a3,b3)
__device__ __forceinline__ void pr_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void pr_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }

// DMA of chunk c (weight rows 32 c .. 32 c + 31, all K) of layer L into `dst`: 1 KiB pieces dealt round-robin to FOUR of the eight waves --
// waves 0..3 for even chunks of the stream, 4..7 for odd ones.  Waves w and w + 4 share a SIMD, and issuing a piece stalls its wave
// ~190 cycles: with all eight issuing behind the barrier every SIMD stood for 4 x 190 cycles per chunk; now one wave per SIMD issues
// (8 pieces) while its partner is already multiplying.
// CPRH: the layer's row length in 16-byte chunks when the call site knows it at compile time (64: the 256-wide layers; 40 / 80: the in_layer at
// K = 160 / the ActNorm + LU pre-layer at K = 320), 0 = read it from the layer.  Round 4, second half: the general form spends two 64-bit
// multiply-adds, a division and ~10 more instructions per piece -- about half of the ~190 cycles a piece holds its wave, and the issuing wave's
// pieces are on the chunk's critical path.  With the row length known a 256-wide row is one piece (lane l reads chunk l ^ (pc & 15) of row pc:
// one 64-bit add per piece on precomputed lane offsets), and for the other lengths the division is a multiply + shift in 32 bits.
template <int CPRH>
__device__ __forceinline__ void pr_dma(const PreMlpLayer& L, int c, char* dst, int wave, int lane, int grp) {
#ifdef FC_PREMLP_DMA_LATE
    const int first = wave, step = 8;                        // experiment: every wave issues its share, BEHIND its chunk's reads and MFMAs (see chunk_mma)
#else
    if ((wave >> 2) != grp) return;
    const int first = wave & 3, step = 4;
    if constexpr (CPRH == 64) {
        const char* ubase = reinterpret_cast<const char*>(L.W2) + (size_t)c * (PR_CH * 64 * 16) + first * 1024;
        const unsigned l0 = (unsigned)(lane ^ first) << 4;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((pm_glb_char*)(ubase + j * 4096 + (l0 ^ (unsigned)((j & 3) << 6))), (pm_lds_char*)(dst + (first + 4 * j) * 1024), 16, 0, 0);
        return;
    } else if constexpr (CPRH > 0) {
        constexpr int SW = (CPRH & 15) == 0 ? 15 : 7, NP = CPRH / 2;
        static_assert(NP % 4 == 0, "every issuing wave takes the same number of pieces");
        const char* ubase = reinterpret_cast<const char*>(L.W2) + (size_t)c * (PR_CH * CPRH * 16);
#pragma unroll
        for (int j = 0; j < NP / 4; ++j) {
            const int pc = first + 4 * j;
            const unsigned ci = (unsigned)(pc * 64 + lane), r = ci / CPRH, q = ci - r * CPRH;
            __builtin_amdgcn_global_load_lds((pm_glb_char*)(ubase + (r * CPRH + (q ^ (r & SW))) * 16u), (pm_lds_char*)(dst + pc * 1024), 16, 0, 0);
        }
        return;
    }
#endif
    const int cpr = L.K_pad >> 2;                            // 16-byte chunks per weight row (64 at K = 256, 40 at K = 160)
    const int sw = (cpr & 15) == 0 ? 15 : 7;
    const int npieces = cpr >> 1;                            // 32 rows * cpr chunks / 64 lanes
    const char* base = reinterpret_cast<const char*>(L.W2) + (size_t)c * PR_CH * cpr * 16;
    for (int pc = first; pc < npieces; pc += step) {
        const int ci = pc * 64 + lane;
        const int r = cpr == 64 ? ci >> 6 : ci / cpr;
        const int q = ci - r * cpr;
        __builtin_amdgcn_global_load_lds((pm_glb_char*)(base + ((size_t)r * cpr + (q ^ (r & sw))) * 16), (pm_lds_char*)(dst + pc * 1024), 16, 0, 0);
    }
}

// KSIN: the in_layer's number of 32-wide k steps when known at compile time (its k loop is then one basic block like the 256-wide
// layers'), 0 = read it from the layer (branches per k step)
// NLU: chunks (of 32 outputs) of the fused ActNorm + LU pre-layer, 0 = none (the input rows are read from p.x)
template <int ACT, int KSIN, int NLU = 0>
__global__ __launch_bounds__(PR_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void premlp_rows_kernel(const PreMlpParams p) {
    extern __shared__ float smem[];
    char* smc = reinterpret_cast<char*>(smem);
    constexpr int PRB = pr_buf(NLU);                          // bytes of one weight stage
    static_assert(NLU == 0 || (KSIN > 0 && KSIN <= NLU && KSIN <= 8), "the pre-layer's first KSIN chunks are the in_layer's input");
    float* biasbuf = reinterpret_cast<float*>(smc + 2 * PRB);
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = blockIdx.x * PR_ROWS + wave * 16 + n;
    float amax = 0.f;
    // Lane constants of the swizzled weight-fragment reads (round 4).  A lane reads 16-byte chunk (ch ^ sw) of its weight row with ch = 8 s + kgp
    // (k step s, hi limb; lo' = ch + 2) and sw = n (rows of 64 or 80 chunks) or n & 7 (40 chunks).  kgp = {0, 1, 4, 5} has bit 1 clear, so the
    // XOR splits: low three bits kgp ^ (n & 7) (lo': the same ^ 2), and -- four-bit swizzle only -- k steps 2 j and 2 j + 1 change places in rows
    // with bit 3 set.  Hence four bases per chunk and compile-time offsets 256 j + 16 KB mb instead of one address computation per read
    // (38 v_add_u32 per chunk of 48 MFMAs in a kernel bound by what its waves issue).
    const int pr_off3 = ((4 * (kg >> 1) + (kg & 1)) ^ (n & 7)) * 16;
    const int pr_off4 = pr_off3 + 128 * ((n >> 3) & 1);

#define PR_STAMP(K_)                                                                                                  \
    if (p.stamps && threadIdx.x == 0) {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        p.stamps[(size_t)blockIdx.x * 16 + (K_)] = __builtin_amdgcn_s_memtime();                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
    if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + 14] = wall_clock64();
    PR_STAMP(0)
    // biases of the five layers -> LDS (the epilogues read them as broadcasts)
    if (tid < PM_H) {
        const PreMlpLayer* Ls[5] = {&p.in, &p.mid0, &p.mid1, &p.out, &p.q};
#pragma unroll
        for (int l = 0; l < 5; ++l) biasbuf[l * PM_H + tid] = (Ls[l]->bias && (l < 4 || tid < 64)) ? Ls[l]->bias[tid] : 0.f;
    }
    if constexpr (NLU > 0) { if (tid < NLU * 32) biasbuf[5 * PM_H + tid] = p.lu.bias ? p.lu.bias[tid] : 0.f; }
    pr_dma<(NLU > 0 ? NLU * 8 : KSIN * 8)>(NLU > 0 ? p.lu : p.in, 0, smc, wave, lane, 0);
    int grp = 1;                                             // which half of the waves issues the next chunk's pieces

    f16x8 ah[8], al[8], nh[8], nl[8];
    // the residual of the second hidden layer (= the in_layer's output) is parked in global scratch while hidden layer 0 runs -- 64 registers
    // this kernel does not have: fragment (s, limb) of lane tid at keep_frag[(s * 2 + limb) * 512 + tid], 1 KiB per wave instruction
    uint4* keep_frag = reinterpret_cast<uint4*>(p.keep_ws) + (size_t)blockIdx.x * 16 * PR_NT + tid;
    // ---- input row -> operand fragments: lane (n, kg) supplies features 32 s + 8 kg + 0..7 of its point
    if constexpr (NLU == 0) {
        const int KS = p.in.K_pad >> 5;
        const float* xr = p.x + (size_t)row * p.ldx + 8 * kg;
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            float xs[8];
            if (s_ < KS) {
                const float4 x0 = *reinterpret_cast<const float4*>(xr + 32 * s_), x1 = *reinterpret_cast<const float4*>(xr + 32 * s_ + 4);
                xs[0] = x0.x; xs[1] = x0.y; xs[2] = x0.z; xs[3] = x0.w; xs[4] = x1.x; xs[5] = x1.y; xs[6] = x1.z; xs[7] = x1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xs[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                amax = fmaxf(amax, fabsf(xs[e]));
                const _Float16 h = (_Float16)xs[e];
                ah[s_][e] = h;
                al[s_][e] = (_Float16)((xs[e] - (float)h) * 2048.0f);
            }
        }
    }
    // ---- pre-layer input: the previous layer's latent row (all 32 NLU columns) -> operand fragments
    f16x8 uh[NLU > 0 ? NLU : 1], ul[NLU > 0 ? NLU : 1];
    if constexpr (NLU > 0) {
        const float* xr = p.xprev + (size_t)row * p.ldxp + 8 * kg;
#pragma unroll
        for (int s_ = 0; s_ < NLU; ++s_) {
            const float4 x0 = *reinterpret_cast<const float4*>(xr + 32 * s_), x1 = *reinterpret_cast<const float4*>(xr + 32 * s_ + 4);
            const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(xs[e]));
            limb_split8(xs, uh[s_], ul[s_]);
        }
    }
    __syncthreads();                                         // biasbuf visible
    PR_STAMP(1)

    int buf = 0;                                             // stage that holds (is receiving) the chunk about to be multiplied
    // diagnostic build (-DFC_PREMLP_WAIT_STAMPS, profiles/micro/premlp_rows_stamps.py): cycles this wave waited for its DMA pieces / at the chunk
    // barriers.  Not in the shipped build: the two counters cost the kernel its last registers (248 bytes of scratch per lane).
#ifdef FC_PREMLP_WAIT_STAMPS
    unsigned long long t_vm = 0, t_bar = 0, tw0 = 0, tw1 = 0;
#define PR_WAIT_T0 tw0 = __builtin_amdgcn_s_memtime();
#define PR_WAIT_T1 tw1 = __builtin_amdgcn_s_memtime();
#define PR_WAIT_T2 { const unsigned long long tw2 = __builtin_amdgcn_s_memtime(); t_vm += tw1 - tw0; t_bar += tw2 - tw1; }
#else
#define PR_WAIT_T0
#define PR_WAIT_T1
#define PR_WAIT_T2
#endif

    // MFMAs of one chunk: (am, ac) = W[32 c .. 32 c + 31][:] . act  (main / cross-product accumulators of the two 16-feature blocks)
    auto chunk_mma = [&](const PreMlpLayer& L, const PreMlpLayer* nextL, int nextc, auto fullk_tag, floatx4 (&am)[2], floatx4 (&ac)[2], auto&& after_dma) __attribute__((always_inline)) {
        constexpr int KSTAT = decltype(fullk_tag)::value;             // compile-time k steps (8 for the 256-wide layers): the k loop is ONE basic block; 0 = run time
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int i = 0; i < 4; ++i) { am[mb][i] = 0.f; ac[mb][i] = 0.f; }
        PR_WAIT_T0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // own pieces of this chunk have landed
        PR_WAIT_T1
        __builtin_amdgcn_s_barrier();                                   // ... everybody's; everybody is done reading the other stage
        PR_WAIT_T2
#ifndef FC_PREMLP_DMA_LATE
        if (nextL) {                                                    // (the next chunk is this layer's, or -- a layer's last chunk -- the first of a 256-wide layer: premlp_fusable)
            if (nextL == &L) pr_dma<KSTAT * 8>(L, nextc, smc + (buf ^ 1) * PRB, wave, lane, grp);
            else pr_dma<64>(*nextL, nextc, smc + (buf ^ 1) * PRB, wave, lane, grp);
        }
#endif
        grp ^= 1;
        after_dma();                                                    // (loads that must not sit in front of the wait above: they get this chunk's time to land)
        if constexpr (KSTAT > 0) {
            // compile-time row length (K = 32 KSTAT: launch_premlp instantiates KSIN = 5 for K_pad 160 only, the other layers are 256 wide)
            constexpr int CPR = KSTAT * 8;
            constexpr bool SW4 = (CPR & 15) == 0;
            const char* wb = smc + buf * PRB + n * (CPR * 16);
            const int o = SW4 ? pr_off4 : pr_off3;
            const char* bh0 = wb + o;
            const char* bl0 = wb + (o ^ 32);
            const char* bh1 = wb + (SW4 ? (o ^ 128) : o + 128);          // odd k steps
            const char* bl1 = wb + (SW4 ? (o ^ 160) : (o ^ 32) + 128);
#pragma unroll
            for (int s_ = 0; s_ < KSTAT; ++s_) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const int imm = 256 * (s_ >> 1) + mb * 16 * CPR * 16;
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(((s_ & 1) ? bh1 : bh0) + imm);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(((s_ & 1) ? bl1 : bl0) + imm);
                    am[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[s_], am[mb], 0, 0, 0);
                    ac[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[s_], ac[mb], 0, 0, 0);
                    ac[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[s_], ac[mb], 0, 0, 0);
                }
            }
#ifdef FC_PREMLP_DMA_LATE
            if (nextL) {                                                    // (the next chunk is this layer's, or -- a layer's last chunk -- the first of a 256-wide layer: premlp_fusable)
            if (nextL == &L) pr_dma<KSTAT * 8>(L, nextc, smc + (buf ^ 1) * PRB, wave, lane, grp);
            else pr_dma<64>(*nextL, nextc, smc + (buf ^ 1) * PRB, wave, lane, grp);
        }
#endif
            buf ^= 1;
            return;
        }
        const int cpr = L.K_pad >> 2, KS = L.K_pad >> 5;
        const int sw = (cpr & 15) == 0 ? n : (n & 7);
        const char* wrow = smc + buf * PRB + n * cpr * 16;
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            if (KSTAT ? s_ < KSTAT : s_ < KS) {
                const int ch = 4 * (2 * s_ + (kg >> 1)) + (kg & 1);      // 16-byte chunk of this lane's 8 k values (hi limb; lo' = + 2)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const char* wr = wrow + mb * 16 * cpr * 16;
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(wr + ((ch) ^ sw) * 16);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(wr + ((ch + 2) ^ sw) * 16);
                    am[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[s_], am[mb], 0, 0, 0);
                    ac[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[s_], ac[mb], 0, 0, 0);
                    ac[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[s_], ac[mb], 0, 0, 0);
                }
            }
        }
#ifdef FC_PREMLP_DMA_LATE
        if (nextL) {                                                    // (the next chunk is this layer's, or -- a layer's last chunk -- the first of a 256-wide layer: premlp_fusable)
            if (nextL == &L) pr_dma<KSTAT * 8>(L, nextc, smc + (buf ^ 1) * PRB, wave, lane, grp);
            else pr_dma<64>(*nextL, nextc, smc + (buf ^ 1) * PRB, wave, lane, grp);
        }
#endif
        buf ^= 1;
    };
    // accumulator order (block mb, register i = feature 16 mb + 4 kg + i) -> operand order: t[e] = out-feature 32 c + 8 kg + e (before bias)
    auto fold = [&](const floatx4 (&am)[2], const floatx4 (&ac)[2], float (&t)[8]) __attribute__((always_inline)) {
        // (am + ac / 2048 as ONE fma each: the scaling by a power of two is exact, so the sum rounds once either way -- the bits of multiply + add)
        constexpr float F = 1.0f / 2048.0f;
        float P0 = fmaf(ac[0][0], F, am[0][0]), P1 = fmaf(ac[0][1], F, am[0][1]), P2 = fmaf(ac[0][2], F, am[0][2]), P3 = fmaf(ac[0][3], F, am[0][3]);
        float Q0 = fmaf(ac[1][0], F, am[1][0]), Q1 = fmaf(ac[1][1], F, am[1][1]), Q2 = fmaf(ac[1][2], F, am[1][2]), Q3 = fmaf(ac[1][3], F, am[1][3]);
        // the eight lane swaps as one block: two wait states between a VALU write and the first swap that reads it; a pair's
        // permlane16 swap stands three instructions behind its permlane32 swap
        asm volatile("s_nop 1\n\t"
                     "v_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7\n\t"
                     "v_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\t"
                     "s_nop 1"
                     : "+v"(P0), "+v"(P1), "+v"(P2), "+v"(P3), "+v"(Q0), "+v"(Q1), "+v"(Q2), "+v"(Q3));
        t[0] = P0; t[1] = P1; t[2] = P2; t[3] = P3; t[4] = Q0; t[5] = Q1; t[6] = Q2; t[7] = Q3;
    };
    // bias, residual, activation, limb split of one chunk, pushed into the output FIFO nh / nl (slot 7 after a shift by one: every
    // chunk then runs the SAME code, so a layer is a loop of ~4 KB instead of 30 KB of straight-line code -- unrolled, the kernel
    // streamed 90 KB of instructions once per workgroup through a 64 KB instruction cache).  Flags are compile-time: straight-line
    // code the scheduler lays under the NEXT chunk's MFMAs (8 values x ~35 VALU instructions per lane and chunk are as much issue time
    // as the chunk's 48 MFMAs).
    auto epilogue = [&](const float (&t)[8], const f16x8& rh, const f16x8& rl, int bias_off, auto resid_tag, auto act_tag, auto shift_tag, auto slot_tag) __attribute__((always_inline)) {
        constexpr bool RESID = decltype(resid_tag)::value;
        constexpr int A = decltype(act_tag)::value;
        constexpr int SHIFT = decltype(shift_tag)::value, SLOT = decltype(slot_tag)::value;      // the FIFO moves up by SHIFT slots, then slot SLOT is written
        const float* bp = biasbuf + bias_off + 8 * kg;
        const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
        const float bs[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        if constexpr (SHIFT > 0) {
#pragma unroll
            for (int b = 0; b + SHIFT < 8; ++b) { nh[b] = nh[b + SHIFT]; nl[b] = nl[b + SHIFT]; }
        }
        float res[8], xv[8];
        if constexpr (RESID) limb_join8(rh, rl, res);                   // (one v_fma_mix_f32 per value, activations.h)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = t[e] + bs[e];
            if constexpr (RESID) x += res[e];
            if constexpr (A == FC_ACT_GELU) x = fc_gelu(x);
            else if constexpr (A == FC_ACT_RELU) x = x > 0.f ? x : 0.f;
            else if constexpr (A == FC_ACT_ELU) x = x > 0.f ? x : expm1f(x);
            else if constexpr (A == FC_ACT_LRELU02) x = x > 0.f ? x : 0.2f * x;
            amax = fmaxf(amax, fabsf(x));
            xv[e] = x;
        }
        limb_split8(xv, nh[SLOT], nl[SLOT]);                            // (five instructions per pair of values)
    };
    // one layer, software-pipelined: iteration c issues the MFMAs of chunk c and, behind them in the same basic block, the epilogue of
    // chunk c - 1 (iteration 0 pushes a dummy that the eight real pushes shift out again)
    auto layer = [&](const PreMlpLayer& L, const PreMlpLayer& nextL, int lidx, auto fullk_tag, auto resid_tag, auto act_tag) __attribute__((always_inline)) {
        constexpr bool RESID = decltype(resid_tag)::value;
        float tp[8];
        f16x8 rph, rpl, rch, rcl;                                       // residual fragments of the previous / this chunk
#pragma unroll
        for (int e = 0; e < 8; ++e) { tp[e] = 0.f; rph[e] = 0; rpl[e] = 0; rch[e] = 0; rcl[e] = 0; }
        // (round 4: the loop body is a PAIR of chunks and the FIFO moves by two slots per pair -- 12 fragment moves per pair instead of 14 per chunk;
        //  8 KB of code per layer instead of 4.  Pushes: [dummy, 0] [1, 2] [3, 4] [5, 6] into slots 6 and 7, then chunk 7 behind a move by one.)
        auto one = [&](int c, auto shift_tag, auto slot_tag) __attribute__((always_inline)) {
            floatx4 am[2], ac[2];
            chunk_mma(L, c + 1 < 8 ? &L : &nextL, c + 1 < 8 ? c + 1 : 0, fullk_tag, am, ac, [&]() __attribute__((always_inline)) {
                if constexpr (RESID) {
                    rch = __builtin_bit_cast(f16x8, keep_frag[(2 * c) * PR_NT]);
                    rcl = __builtin_bit_cast(f16x8, keep_frag[(2 * c + 1) * PR_NT]);
                }
            });
            epilogue(tp, rph, rpl, lidx * PM_H + 32 * (c > 0 ? c - 1 : 0), resid_tag, act_tag, shift_tag, slot_tag);
            fold(am, ac, tp);
            rph = rch; rpl = rcl;
        };
#pragma unroll 1
        for (int c = 0; c < 8; c += 2) {
            one(c, std::integral_constant<int, 2>{}, std::integral_constant<int, 6>{});
            one(c + 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 7>{});
        }
        epilogue(tp, rph, rpl, lidx * PM_H + 32 * 7, resid_tag, act_tag, std::integral_constant<int, 1>{}, std::integral_constant<int, 7>{});
#pragma unroll
        for (int b = 0; b < 8; ++b) { ah[b] = nh[b]; al[b] = nl[b]; }
    };
    using ActT = std::integral_constant<int, ACT>; using NoAct = std::integral_constant<int, FC_ACT_NONE>;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int e = 0; e < 8; ++e) { nh[b][e] = 0; nl[b][e] = 0; }
    // ---- pre-layer (NLU > 0): z = W_lu xprev + b_lu, the previous flow layer's folded ActNorm + permuter (models/act_norm.py:37-43,
    //      models/permuters.py:164-169; flow_engine.cpp build_lin) -- one launch per layer fewer, and its output never crosses HBM on the way
    //      to this kernel.  Same chunk pipeline as `layer`: iteration c multiplies chunk c (32 outputs, K = 32 NLU from the uh / ul fragments)
    //      and finishes chunk c - 1: bias, fp32 store of the lane's 8 consecutive columns of its row into xnext (every later kernel of the
    //      layer reads the latent there), and -- for the first KSIN chunks, x1' -- the limb split into the in_layer's input fragments.
    if constexpr (NLU > 0) {
        f16x8 fh[KSIN], fl[KSIN];                                       // FIFO of the in_layer's input fragments (chunk c ends up at index c)
#pragma unroll
        for (int b = 0; b < KSIN; ++b)
#pragma unroll
            for (int e = 0; e < 8; ++e) { fh[b][e] = 0; fl[b][e] = 0; }
        auto lu_mma = [&](int c, floatx4 (&am)[2], floatx4 (&ac)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int i = 0; i < 4; ++i) { am[mb][i] = 0.f; ac[mb][i] = 0.f; }
            PR_WAIT_T0
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // own pieces of this chunk have landed
            PR_WAIT_T1
            __builtin_amdgcn_s_barrier();                                   // ... everybody's; everybody is done reading the other stage
            PR_WAIT_T2
#ifndef FC_PREMLP_DMA_LATE
            if (c + 1 < NLU) pr_dma<NLU * 8>(p.lu, c + 1, smc + (buf ^ 1) * PRB, wave, lane, grp);
            else pr_dma<KSIN * 8>(p.in, 0, smc + (buf ^ 1) * PRB, wave, lane, grp);
#endif
            grp ^= 1;
            constexpr int cpr = NLU * 8;                                    // 16-byte chunks per weight row
            constexpr bool SW4 = (cpr & 15) == 0;
            const char* wb = smc + buf * PRB + n * (cpr * 16);              // (bases + compile-time offsets: see pr_off3 / pr_off4)
            const int o = SW4 ? pr_off4 : pr_off3;
            const char* bh0 = wb + o;
            const char* bl0 = wb + (o ^ 32);
            const char* bh1 = wb + (SW4 ? (o ^ 128) : o + 128);
            const char* bl1 = wb + (SW4 ? (o ^ 160) : (o ^ 32) + 128);
#pragma unroll
            for (int s_ = 0; s_ < NLU; ++s_) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const int imm = 256 * (s_ >> 1) + mb * 16 * cpr * 16;
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(((s_ & 1) ? bh1 : bh0) + imm);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(((s_ & 1) ? bl1 : bl0) + imm);
                    am[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, uh[s_], am[mb], 0, 0, 0);
                    ac[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, uh[s_], ac[mb], 0, 0, 0);
                    ac[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ul[s_], ac[mb], 0, 0, 0);
                }
            }
#ifdef FC_PREMLP_DMA_LATE
            if (c + 1 < NLU) pr_dma<NLU * 8>(p.lu, c + 1, smc + (buf ^ 1) * PRB, wave, lane, grp);
            else pr_dma<KSIN * 8>(p.in, 0, smc + (buf ^ 1) * PRB, wave, lane, grp);
#endif
            buf ^= 1;
        };
        auto lu_finish = [&](const float (&t)[8], int c) __attribute__((always_inline)) {       // chunk c >= 0 (wave-uniform)
            const float* bp = biasbuf + 5 * PM_H + 32 * c + 8 * kg;
            const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
            const float z[8] = {t[0] + b0.x, t[1] + b0.y, t[2] + b0.z, t[3] + b0.w, t[4] + b1.x, t[5] + b1.y, t[6] + b1.z, t[7] + b1.w};
            float* zp = p.xnext + (size_t)row * p.ldxn + 32 * c + 8 * kg;
            *reinterpret_cast<float4*>(zp) = make_float4(z[0], z[1], z[2], z[3]);
            *reinterpret_cast<float4*>(zp + 4) = make_float4(z[4], z[5], z[6], z[7]);
            if (c < KSIN) {
#pragma unroll
                for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(z[e]));
#pragma unroll
                for (int b = 0; b + 1 < KSIN; ++b) { fh[b] = fh[b + 1]; fl[b] = fl[b + 1]; }
                limb_split8(z, fh[KSIN - 1], fl[KSIN - 1]);
            }
        };
        float tp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) tp[e] = 0.f;
#pragma unroll 1
        for (int c = 0; c < NLU; ++c) {
            floatx4 am[2], ac[2];
            lu_mma(c, am, ac);
            if (c > 0) lu_finish(tp, c - 1);
            fold(am, ac, tp);
        }
        lu_finish(tp, NLU - 1);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (b < KSIN) { ah[b] = fh[b]; al[b] = fl[b]; }
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { ah[b][e] = 0; al[b][e] = 0; }
            }
        }
    }
    // ---- in_layer, hidden layer 0 (keep = x; x = act(W x)), hidden layer 1 (x = act(keep + W x)), out_layer (no activation)
    layer(p.in, p.mid0, 0, std::integral_constant<int, KSIN>{}, std::false_type{}, ActT{});
    PR_STAMP(2)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        keep_frag[(2 * b) * PR_NT] = __builtin_bit_cast(uint4, ah[b]);
        keep_frag[(2 * b + 1) * PR_NT] = __builtin_bit_cast(uint4, al[b]);
    }
    layer(p.mid0, p.mid1, 1, std::integral_constant<int, 8>{}, std::false_type{}, ActT{});
    PR_STAMP(3)
    layer(p.mid1, p.out, 2, std::integral_constant<int, 8>{}, std::true_type{}, ActT{});
    PR_STAMP(4)
    layer(p.out, p.q, 3, std::integral_constant<int, 8>{}, std::false_type{}, NoAct{});
    PR_STAMP(5)

    // ---- LayerNorm over the 256 features of the point (biased variance, eps 1e-5; gamma / beta live in the q projection): the four
    //      lanes of a point hold 64 features each, as limbs (x = hi + lo'/2048 to 2^-24)
    {
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)ah[b][e] + (float)al[b][e] * (1.0f / 2048.0f);
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / PM_H);
        float sq = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dlt = ((float)ah[b][e] + (float)al[b][e] * (1.0f / 2048.0f)) - mean;
                sq += dlt * dlt;
            }
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / PM_H) + 1e-5f);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float xv[8];
            limb_join8(ah[b], al[b], xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xv[e] = (xv[e] - mean) * rstd;
                amax = fmaxf(amax, fabsf(xv[e]));
            }
            limb_split8(xv, ah[b], al[b]);
        }
    }

    PR_STAMP(6)
    // ---- q projection 256 -> 64: two chunks; the lane stores 8 consecutive columns of its row
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        floatx4 am[2], ac[2];
        float t[8];
        chunk_mma(p.q, c == 0 ? &p.q : nullptr, 1, std::integral_constant<int, 8>{}, am, ac, []() {});
        fold(am, ac, t);
        const float* bp = biasbuf + 4 * PM_H + 32 * c + 8 * kg;
        const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
        float* qp = p.qout + (size_t)row * p.ldq + 32 * c + 8 * kg;
        *reinterpret_cast<float4*>(qp) = make_float4(t[0] + b0.x, t[1] + b0.y, t[2] + b0.z, t[3] + b0.w);
        *reinterpret_cast<float4*>(qp + 4) = make_float4(t[4] + b1.x, t[5] + b1.y, t[6] + b1.z, t[7] + b1.w);
    }
    if (amax >= 65504.0f) atomicOr(p.ovf, 1);
    PR_STAMP(7)
    if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + 15] = wall_clock64();
#ifdef FC_PREMLP_WAIT_STAMPS
    if (p.stamps && (threadIdx.x == 0 || threadIdx.x == 256)) {      // (wave 0: issues the even chunks' pieces; wave 4: the odd ones')
        p.stamps[(size_t)blockIdx.x * 16 + 8 + 2 * (threadIdx.x >> 8)] = t_vm;
        p.stamps[(size_t)blockIdx.x * 16 + 9 + 2 * (threadIdx.x >> 8)] = t_bar;
    }
#endif
#undef PR_WAIT_T0
#undef PR_WAIT_T1
#undef PR_WAIT_T2
#undef PR_STAMP
}

int g_premlp_fused = 2;       // tuning knob (fc_debug_set 8): 2 = the row-resident kernel (activations in registers; shipped: 190 us against ~250 us for the
                              // five launches it replaces, -1.4 ... -2 % per C2 step), 1 = the LDS-tile kernel (263 us: one 64-row workgroup per CU
                              // re-streams every layer's weights from L2, 16 % MFMA busy), 0 = separate GEMM launches + the LayerNorm -> q fold

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

// launch conditions of the row-resident kernel (the caller hands it 256 floats of scratch per row: a workspace planned for narrower coupling
// nets does not have them, and the pre-conditioner then runs as separate launches)
bool premlp_rows_ok(int rows_alloc, int ldq, const float* qout, const float* keep_ws, size_t keep_floats) {
    return rows_alloc % PR_ROWS == 0 && ldq % 4 == 0 && ((uintptr_t)qout & 15) == 0 && keep_ws && ((uintptr_t)keep_ws & 15) == 0 &&
           keep_floats >= (size_t)rows_alloc * PM_H;
}

extern int g_gemm_stamp;
unsigned long long* gemm_stamp_buffer(size_t n);      // gemm.hip: the knob-20 stamp buffer (grown on demand), read back by fc_debug_gemm_stamps

int g_premlp_lu = 1;          // knob 26: 1 = the previous layer's folded ActNorm + LU runs as a pre-layer of the row-resident kernel (shipped), 0 = as its own GEMM launch

// true when `lu` (the previous flow layer's folded ActNorm + permuter, latent pitch ldx) can run as the pre-layer of this pre-conditioner's
// row-resident kernel: square 320 x 320 in the latent's padded layout, the in_layer reading its first 160 columns, GELU (the instantiated case)
bool premlp_lu_fusable(const PackedLinear& lu, const PackedLinear& in, int act, int ldx) {
    return g_premlp_lu && (g_premlp_fused == 2 || !kDevVariants) && act == FC_ACT_GELU && lu.W2 != nullptr && lu.bias != nullptr && lu.nseg == 1 && lu.K_pad == 320 && lu.N_pad == 320 && ldx == 320 &&
           lu.n_alloc >= 320 && in.K_pad == 160;
}

void launch_premlp(const float* x, int ldx, const PackedLinear& in, const std::vector<PackedLinear>& mid, const PackedLinear& out,
                   const PackedLinear& q, int act, float* qout, int ldq, int rows_alloc, int rows_valid, hipStream_t s, float* keep_ws, size_t keep_floats,
                   const PackedLinear* lu, const float* xprev) {
    if (rows_alloc % PM_ROWS != 0 || ldx % 4 != 0 || ldx < in.K_pad || ((uintptr_t)x & 15))
        throw Error(FC_ERR_INVALID, "premlp: rows must be padded to 64, input pitch to 4 floats");
#ifdef FC_DEV_VARIANTS
    static PerDeviceOnce attr_once;
    attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(premlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PM_LDS)); return 0; });
#endif
    PreMlpParams p{};
    p.x = x; p.ldx = ldx;
    auto L = [](const PackedLinear& l) { return PreMlpLayer{l.W2, l.bias, l.K_pad}; };
    p.in = L(in); p.mid0 = L(mid[0]); p.mid1 = L(mid[1]); p.out = L(out); p.q = L(q);
    p.act = act; p.qout = qout; p.ldq = ldq; p.rows = rows_alloc; p.ovf = gemm_fp16_flag(); p.keep_ws = keep_ws;
    p.stamps = g_gemm_stamp == 3 ? gemm_stamp_buffer((size_t)(rows_alloc / PR_ROWS) * 16) : nullptr;
    const double rv = rows_valid > 0 ? rows_valid : rows_alloc;
    double flops = 2.0 * rv * ((double)in.k_true * PM_H + 3.0 * PM_H * PM_H + (double)PM_H * (q.n_true ? q.n_true : 64));
    if ((g_premlp_fused == 2 || !kDevVariants) && premlp_rows_ok(rows_alloc, ldq, qout, keep_ws, keep_floats)) {
        if (lu) {
            // the previous layer's ActNorm + LU as a pre-layer: x (this layer's latent buffer) is WRITTEN here, xprev is read
            if (!premlp_lu_fusable(*lu, in, act, ldx) || !xprev || ((uintptr_t)xprev & 15))
                throw Error(FC_ERR_INVALID, "launch_premlp: the ActNorm + LU pre-layer does not fit the row-resident kernel (callers check premlp_lu_fusable)");
            p.lu = L(*lu); p.xprev = xprev; p.ldxp = ldx; p.xnext = const_cast<float*>(x); p.ldxn = ldx;
            flops += 2.0 * rv * (double)(lu->k_true ? lu->k_true : lu->K_pad) * (double)(lu->n_true ? lu->n_true : lu->N_pad);
            auto kern = premlp_rows_kernel<FC_ACT_GELU, 5, 10>;
            static PerDeviceOnce attr_once;
            attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pr_lds(10))); return 0; });
            ProfScope ps("void fc::premlp_rows_kernel<1, 5, 10>(fc::PreMlpParams)", flops, 0.0, s);      // (the name rocprofv3 prints: profiles/pmc_traffic.json is keyed by it)
            hipLaunchKernelGGL(kern, dim3(rows_alloc / PR_ROWS), dim3(PR_NT), pr_lds(10), s, p);
            FC_HIP(hipGetLastError());
            return;
        }
        auto go = [&](auto kern, int act_code, int ksin) {
            static PerDeviceOnce attr_once;                             // (one per kernel instantiation: `go` is a generic lambda)
            attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PR_LDS)); return 0; });
            char name[96];
            snprintf(name, sizeof name, "void fc::premlp_rows_kernel<%d, %d, 0>(fc::PreMlpParams)", act_code, ksin);      // (the name rocprofv3 prints)
            ProfScope ps(name, flops, 0.0, s);
            hipLaunchKernelGGL(kern, dim3(rows_alloc / PR_ROWS), dim3(PR_NT), PR_LDS, s, p);
            FC_HIP(hipGetLastError());
        };
        const bool k5 = in.K_pad == 160;                                // latent 300: x1 = 150 columns (the shipped configurations)
        switch (act) {
            case FC_ACT_GELU: k5 ? go(premlp_rows_kernel<FC_ACT_GELU, 5>, FC_ACT_GELU, 5) : go(premlp_rows_kernel<FC_ACT_GELU, 0>, FC_ACT_GELU, 0); break;
            case FC_ACT_RELU: k5 ? go(premlp_rows_kernel<FC_ACT_RELU, 5>, FC_ACT_RELU, 5) : go(premlp_rows_kernel<FC_ACT_RELU, 0>, FC_ACT_RELU, 0); break;
            case FC_ACT_ELU: k5 ? go(premlp_rows_kernel<FC_ACT_ELU, 5>, FC_ACT_ELU, 5) : go(premlp_rows_kernel<FC_ACT_ELU, 0>, FC_ACT_ELU, 0); break;
            case FC_ACT_LRELU02: k5 ? go(premlp_rows_kernel<FC_ACT_LRELU02, 5>, FC_ACT_LRELU02, 5) : go(premlp_rows_kernel<FC_ACT_LRELU02, 0>, FC_ACT_LRELU02, 0); break;
            default: k5 ? go(premlp_rows_kernel<FC_ACT_NONE, 5>, FC_ACT_NONE, 5) : go(premlp_rows_kernel<FC_ACT_NONE, 0>, FC_ACT_NONE, 0); break;
        }
        return;
    }
    if (lu) throw Error(FC_ERR_INVALID, "launch_premlp: the ActNorm + LU pre-layer exists in the row-resident kernel only");
#ifdef FC_DEV_VARIANTS
    ProfScope ps("fc::premlp_kernel(fc::PreMlpParams)", flops, 0.0, s);
    hipLaunchKernelGGL(premlp_kernel, dim3(rows_alloc / PM_ROWS), dim3(PM_NT), PM_LDS, s, p);
    FC_HIP(hipGetLastError());
#else
    throw Error(FC_ERR_UNSUPPORTED, "launch_premlp: the row-resident kernel's launch conditions do not hold (callers check premlp_rows_ok)");
#endif
}

}  // namespace fc
