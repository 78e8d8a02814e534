// The GEMM of the flow on the CDNA4 matrix cores, with fused epilogues.
//
//   C[rows, N_pad] = epilogue( sum_seg A_seg[rows, k_seg] @ W[N_pad, K_pad]^T )
//
// Every Linear of the flow (coupling MLPs, q/kv projections, the folded ActNorm+LinearLU matrix, DGCNN / PAConv 1x1
// convolutions) goes through gemm_f32_kernel; operands and results are fp32 in memory, the products run in one of three
// main loops selected by the template parameter VAR:
//   VAR 5  split-fp16, the default: each operand as two fp16 limbs (hi + lo'/2048, 2^-24 relative), 3 v_mfma_f32_32x32x16_f16
//          per product block, main and cross-product fp32 accumulators; needs the caller's Fp16Guard scope (|x| < 65504);
//   VAR 3  split-bf16: three bf16 limbs, 6 MFMAs per block, unbounded range -- the pass a guarded call is repeated with;
//   VAR 0-2 fp32-input MFMA (v_mfma_f32_32x32x2_f32, an exact fmaf chain): the first build's loop, kept for A/B and tests.
// Epilogues (EPI): LINEAR (bias, rank-1 extra-context term, residual, activation), SPLINE (forward rational-quadratic spline
// coupling on the tile the workgroup just produced), AFFINE / AUGMENT / SLICE (pair-packed [first 32 | second 32] columns).
//
// Layout: both operands are K-contiguous in memory (activations [rows][K], weights [N][K] exactly like torch.nn.Linear.weight;
// limb images [N][K/16][limb][16]), so one ds_read_b128 per lane is one MFMA operand.  Global -> LDS goes through registers
// (split loops: two-deep prefetch, whole-vector staging values, double-buffered LDS rows with a 16-byte pad).
// Shipped tile for the split-fp16 loop: 128x128 on eight waves of 32x64 (118 VGPRs, 4 waves per SIMD).
#include "common.h"
#include "activations.h"
#include "spline.h"
#include <atomic>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

namespace fc {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct GemmParams {
    const float* A[3];
    int lda[3];
    int kt[3];          // 32-wide k tiles per segment
    int KT;             // total k tiles
    const float* W;     // [N_pad][K_pad]
    const unsigned short* W3;   // bf16 limb image [n_alloc][K_pad/16][3][16] (split-bf16 variant)
    const unsigned short* W2;   // fp16 limb image [n_alloc][K_pad/16][2][16]: hi, lo' = (w - hi) * 2048 (split-fp16 variant)
    int* ovf;                   // split-fp16 variant: set to 1 when an activation >= 65504 was met
    int K_pad;
    const float* bias;
    const float* colvec;
    int N_pad;
    int nbm, nbn;
    int col_group;      // > 0: row-band / column-group tile order for weight matrices that do not fit L2
    unsigned long long* stamps;   // diagnostic knob 20: 16 x u64 per workgroup (s_memtime at the phase boundaries, HW_ID, wall clock); null otherwise
    GemmEpi e;
};


constexpr int LDS_LD = 36;   // floats per LDS row (32 + 4 pad)

// In-kernel phase stamps of the LDS-DMA kernels (diagnostic knob 20; profiles/micro/spline_gemm_stamps.py): thread 0 of a workgroup stores
// the shader-clock counter.  Slot 0 entry, 1 prologue issued, 2 first k tile landed, 3 main loop done, 4 epilogue operands ready (LDS tile
// written / register exchange done), 5 splines evaluated, 6 results stored, 7 HW_ID | XCC_ID << 32, 8 / 9 wall clock (100 MHz) at entry / exit.
#define FC_STAMP(K_)                                                                                                 \
    if (p.stamps && threadIdx.x == 0) {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        p.stamps[(size_t)blockIdx.x * 16 + (K_)] = __builtin_amdgcn_s_memtime();                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }

// Lanes 0..31 of the result receive lanes 32..63 of v (v_permlane32_swap_b32 swaps the upper half of its first operand with the lower
// half of its second; lanes 32..63 of the result are unspecified).  Inline assembly: this hipcc's __builtin_amdgcn_permlane32_swap
// hands back its first result for both elements (profiles/micro/permlane32_swap_probe.hip); the s_nops cover the VALU <-> permlane-swap
// wait states the compiler cannot schedule around an asm block.
__device__ __forceinline__ float upper_to_lower(float v) {
    float a = v, b = 0.f;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return b;
}

template <int BM, int BN, int WM, int WN, int EPI, int VAR = 2>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(BN > 128 ? 1 : (BM == 128 && WM * WN == 8) ? 4 : 2)))   // resident waves per SIMD the register budget must allow
void gemm_f32_kernel(const GemmParams p) {
    constexpr int NT = WM * WN * 64;                       // 4 or 8 waves per workgroup
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_F4 = BM * 8 / NT, B_F4 = BN * 8 / NT, RPP = NT / 8;   // float4 per thread and tile; rows per staging pass
    constexpr int STAGE = (BM + BN) * LDS_LD;
    extern __shared__ float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware tile order.  Blocks b, b+8, ... share an XCD (and its 4 MiB L2).
    //  * small weight matrix (fits L2 beside the activations): each XCD takes a contiguous run of (row-tile, col-tile)
    //    pairs with the col-tile fastest, so an A row panel is fetched from HBM once and W stays L2 resident;
    //  * large weight matrix (p.col_group > 0, e.g. the 3750-wide spline parameter layer, W = 7.7 MB): each XCD owns a
    //    band of row tiles and walks it in groups of col_group column tiles, so that group of W tiles stays in L2 while
    //    the band's A panels stream past (measured before: 5x the algorithmic bytes were re-fetched through L2).
    int bm, bn;
    {
        const int nb = p.nbm * p.nbn, b = blockIdx.x;
        const int xcd = b & 7, loc = b >> 3;
        if (p.col_group > 0) {
            const int rows_x = p.nbm >> 3, G = p.col_group;           // launcher guarantees nbm % 8 == 0
            const int g = loc / (rows_x * G);
            const int rem = loc - g * rows_x * G;
            const int w = p.nbn - g * G < G ? p.nbn - g * G : G;
            const int r = rem / w;
            bm = xcd * rows_x + r;
            bn = g * G + (rem - r * w);
        } else {
            const int q = nb >> 3, r = nb & 7;
            const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
            bm = L / p.nbn;
            bn = L - bm * p.nbn;
        }
    }
    const int m0 = bm * BM, n0 = bn * BN;
    const int wave_n0 = n0 + wc * TN * 32;
    const int wave_m0 = m0 + wr * TM * 32;
    int nvalid = (p.N_pad - wave_n0) / 32;                 // wave-uniform number of live 32-col tiles (for the stores only:
    nvalid = nvalid < 0 ? 0 : (nvalid > TN ? TN : nvalid); // W / bias are allocated zero-padded to the grid, the k-loop is branch free)
    const GemmEpi& e = p.e;
    if constexpr (VAR == 8 || VAR == 9 || VAR == 10) {
        if (p.stamps && threadIdx.x == 0) {
            p.stamps[(size_t)blockIdx.x * 16 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
            p.stamps[(size_t)blockIdx.x * 16 + 8] = wall_clock64();
        }
        FC_STAMP(0)
    }


    // ---- accumulators start from the epilogue's additive terms (bias, rank-1 extra-context term, residual), so their
    //      global loads overlap the first tile's loads instead of forming a dependent tail after the last MFMA.
    //      Every runtime condition is hoisted OUTSIDE the unrolled element loops (a per-element "load or not" makes
    //      hipcc branch and wait vmcnt(0) around each load).
    // fused spline epilogue: the x2 values (and the log-det slot) this thread will update after the main loop are fetched NOW -- their
    // HBM latency then hides under the k loop instead of standing exposed between the tile's last MFMA and its spline evaluation
    constexpr int SPL_PER_THREAD = EPI == EPI_SPLINE ? (BM * 5 + NT - 1) / NT : 1;      // (K = 8: 5 dims per 128-column tile; K = 4 / 16 re-load below)
    float spl_x[SPL_PER_THREAD];
    float spl_ldj = 0.f;
    if constexpr (EPI == EPI_SPLINE && VAR == 10) {
        // transposed product (below): this lane evaluates dims 2 lh, 2 lh + 1 (and, lower half, dim 4) of ONE point
        const int row = m0 + wave * 32 + li, dim0 = bn * 5;
        const float* xr = e.xbuf + (size_t)row * e.ldx + e.x2_col0 + dim0;
        const bool rv = row < e.rows_valid;
        spl_x[0] = rv && dim0 + 2 * lh < e.d2 ? xr[2 * lh] : 0.f;
        spl_x[1] = rv && dim0 + 2 * lh + 1 < e.d2 ? xr[2 * lh + 1] : 0.f;
        spl_x[2] = rv && dim0 + 4 < e.d2 ? xr[4] : 0.f;
        if (lh == 0) spl_ldj = e.ldj_part[(size_t)bn * e.ldj_pitch + row];
    } else if constexpr (EPI == EPI_SPLINE) {
        const int per = 3 * e.spline_K + 1, DPT = BN / per, dim0 = bn * DPT;
#pragma unroll
        for (int k = 0; k < SPL_PER_THREAD; ++k) {
            const int it = tid + k * NT, row = it % BM, dl = it / BM;
            spl_x[k] = (DPT == 5 && it < BM * DPT && dim0 + dl < e.d2 && m0 + row < e.rows_valid)
                           ? e.xbuf[(size_t)(m0 + row) * e.ldx + e.x2_col0 + dim0 + dl] : 0.f;
        }
        if (tid < BM) spl_ldj = e.ldj_part[(size_t)bn * e.ldj_pitch + m0 + tid];
    }
    floatx16 acc[TM][TN];
    if constexpr (VAR == 10) {
        // transposed product: the accumulator's ROW index (register r, lane half) walks the tile's columns, so the bias varies per register
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n0 + j * 32 + 8 * g + 4 * lh);      // (the launcher requires a bias)
                acc[0][j][4 * g + 0] = b4.x; acc[0][j][4 * g + 1] = b4.y; acc[0][j][4 * g + 2] = b4.z; acc[0][j][4 * g + 3] = b4.w;
            }
    } else
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        float bv = 0.f;
        if constexpr (EPI == EPI_LINEAR || EPI == EPI_SPLINE || EPI == EPI_LNQ) bv = p.bias ? p.bias[wave_n0 + j * 32 + li] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
    }
    if constexpr (EPI == EPI_LINEAR) {
        if (e.rowscal && p.colvec) {
            float rs[TM][16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) rs[i][r] = e.rowscal[wave_m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float cv = p.colvec[wave_n0 + j * 32 + li];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += rs[i][r] * cv;
            }
        }
        // (a residual that arrives as a limb image, e.residual16, is added BEHIND the k loop -- in the epilogue below: the row-resident chain
        // kernel (mlprows.hip) adds it there, and the engine picks between that kernel and these per-layer launches by the row count, so the
        // two must round alike for a scene's log-probs not to depend on the batch it sits in; it is also the more accurate order)
        if (e.residual) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (j < nvalid) {
                    const float* rp = e.residual + (size_t)(wave_m0 + 4 * lh) * e.ldr + wave_n0 + j * 32 + li;
                    float t[TM][16];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) t[i][r] = rp[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * e.ldr];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] += t[i][r];
                }
            }
        }
    }

    // 64 x 64 tiles (one 32 x 32 block per wave): the limb-image residual is requested HERE, in front of the k loop, and added behind it as in
    // every other tile shape -- such a launch lasts one workgroup's life (~9 us), and 32 two-byte loads issued in the epilogue were ~0.7 us of it
    constexpr bool RES_EARLY = EPI == EPI_LINEAR && VAR == 9 && TM == 1 && TN == 1;
    unsigned short res_h[RES_EARLY ? 16 : 1], res_l[RES_EARLY ? 16 : 1];
    if constexpr (RES_EARLY) {
        if (e.residual16 && nvalid > 0) {
            const int blocks = e.ldr16 >> 4;
            const int col = wave_n0 + li;
            const unsigned short* rp = e.residual16 + ((size_t)(wave_m0 + 4 * lh) * blocks + (col >> 4)) * 32 + (col & 15);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned short* q = rp + (size_t)((r & 3) + 8 * (r >> 2)) * blocks * 32;
                res_h[r] = q[0]; res_l[r] = q[16];
            }
        }
    }

    if constexpr (VAR == 8 || VAR == 9 || VAR == 10) {
        // ================= split-fp16 main loop on LDS-DMA: BOTH operands arrive as fp16 limb images =================
        // A is the image its producer's epilogue wrote (e.A16, [rows][K/16][hi 16 | lo' 16]), W the host-packed one (p.W2): the main
        // loop converts nothing, so global -> LDS is a byte copy and goes through `global_load_lds_dwordx4` (no staging VGPRs, no
        // ds_write, whose VGPR -> LDS path was half busy in the register-staged loop).  256 x 128 tile on eight waves of 64 x 64 (128
        // accumulator registers: main + cross-product sets), two waves per SIMD, one workgroup per CU; k tile 32 = two 64-byte
        // (row, k16) blocks = 128 bytes per LDS row; three LDS stages of 48 KB; ONE raw s_barrier per k tile with a counted vmcnt
        // wait, so the DMA of tile t+1 stays in flight across the barrier while tile t is multiplied and tile t+2 is issued.
        //   LDS image: row r = 128 bytes = 8 chunks of 16 B; logical chunk c = 4*(k16 block) + 2*limb + (k half) sits at physical
        //   chunk c ^ ((r >> 1) & 7): with 128-byte rows two rows share a 256-byte bank row, and ds_read_b128's 16-lane groups
        //   ({0-3,12-15,20-27}, ...) then hit 16 distinct 16-byte slots.  The DMA writes LDS linearly (wave base + lane * 16), so the
        //   permutation is applied to the per-lane SOURCE address and again on the read (same involution on both sides).
        // VAR 9: the same loop on a 128 x 128 tile with FOUR waves of 64 x 64 and TWO LDS stages of 32 KB, so that two workgroups fit a
        // CU (2 x 64 KB of stages / 2 x 69 KB with the spline epilogue's parameter tile; 2 waves per SIMD): one workgroup's epilogue
        // then overlaps the other's main loop, which the one-workgroup-per-CU 256-row tile cannot do.
        // VAR 10 (fused spline layer): the VAR 9 tile with its four waves stacked along the rows (32 points x 128 columns each) and the
        // MFMA operands SWAPPED -- weights as the A operand, points as B -- so the accumulator holds, per lane, 64 parameters of ONE
        // point (the other 64 sit in lane ^ 32).  With the column order of spline.h that is every parameter of 2-3 transformed dims in
        // registers with compile-time indices: the spline is evaluated straight from the accumulators, the tile never goes through LDS
        // (no 66 KB parameter tile, no transposition, no epilogue barrier).
        static_assert((BN == 128 && ((VAR == 8 && BM == 256 && WM == 4 && WN == 2) || (VAR == 9 && BM == 128 && WM == 2 && WN == 2) ||
                                     (VAR == 10 && BM == 128 && WM == 4 && WN == 1 && EPI == EPI_SPLINE))) ||
                          (VAR == 9 && BM == 64 && BN == 64 && WM == 2 && ((WN == 2 && EPI == EPI_LINEAR) || (WN == 1 && EPI == EPI_AFFINE))),
                      "LDS-DMA loop: 256x128 on 4x2 waves (VAR 8), 128x128 on 2x2 waves (VAR 9) or on 4x1 waves, transposed (VAR 10); 64x64 on 2x2 "
                      "waves (EPI_LINEAR) / 2x1 waves (EPI_AFFINE: a wave's 64 columns are one pair block) for launches too small to fill the chip with 128x128 tiles");
        // 64 x 64 tiles (launches too small to fill the chip: ONE workgroup's k loop is the launch's duration, and with 6 MFMAs per wave and
        // k step that loop is pure DMA latency): EIGHT stages of 16 KB, seven k steps in flight, so the whole K = 512 operand is on its way
        // after one latency instead of one latency per k step (C1: 19 -> ~10 us per hidden-layer launch).  Same MFMAs in the same order.
        constexpr int NST8 = VAR == 8 ? 3 : (BM == 64 ? 8 : 2);
        constexpr int ROWB8 = 128, STAGE8 = (BM + BN) * ROWB8;        // launch_cfg reserves NST8 * STAGE8
        constexpr int PPW = STAGE8 / 1024 / (NT / 64);                       // 1-KB DMA pieces per wave and stage: 6
        typedef __attribute__((address_space(3))) char lds_char;
        typedef const __attribute__((address_space(1))) char glb_char;
        char* smc = reinterpret_cast<char*>(smem);
        const int KT = p.KT;                                                // k32 tiles
        const size_t rowbytes = (size_t)KT * 128;
        // piece pc = wave * PPW + i covers stage rows 8 pc .. 8 pc + 7 (rows 0..255: A, 256..383: W); lane l: row 8 pc + (l >> 3), physical chunk l & 7
        const char* gsrc[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pc = wave * PPW + i;
            const int r = pc * 8 + (lane >> 3);
            const int cl = (lane & 7) ^ ((r >> 1) & 7);
            const char* base = pc < BM / 8 ? reinterpret_cast<const char*>(e.A16) + (size_t)(m0 + r) * rowbytes
                                           : reinterpret_cast<const char*>(p.W2) + (size_t)(n0 + r - BM) * rowbytes;
            gsrc[i] = base + cl * 16;
        }
#define FC_DMA8(KT_, ST_)                                                                                          \
        {                                                                                                          \
            _Pragma("unroll") for (int i = 0; i < PPW; ++i)                                                        \
                __builtin_amdgcn_global_load_lds((glb_char*)(gsrc[i] + (size_t)(KT_) * 128),                       \
                                                 (lds_char*)(smc + (ST_) * STAGE8 + (wave * PPW + i) * 1024), 16, 0, 0); \
        }
        floatx16 corr[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) corr[i][j][r] = 0.f;
        const int xsw = (li >> 1) & 7;                                      // (row >> 1) & 7 of every row this lane reads (tiles are 32-row aligned)
        const int a_row = (wr * TM * 32 + li) * ROWB8, b_row = (BM + wc * TN * 32 + li) * ROWB8;
#define FC_MMA8_STAGE(ST_)                                                                                         \
        {                                                                                                          \
            const char* sA = smc + (ST_) * STAGE8 + a_row;                                                         \
            const char* sB = smc + (ST_) * STAGE8 + b_row;                                                         \
            _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) {                                                  \
                f16x8 af8[TM][2], bf8[TN][2];                                                                      \
                _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                    \
                    const int off = ((sub * 4 + q * 2 + lh) ^ xsw) * 16;                                           \
                    _Pragma("unroll") for (int i = 0; i < TM; ++i) af8[i][q] = *reinterpret_cast<const f16x8*>(sA + i * 32 * ROWB8 + off); \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j) bf8[j][q] = *reinterpret_cast<const f16x8*>(sB + j * 32 * ROWB8 + off); \
                }                                                                                                  \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                     \
                    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                               \
                        if constexpr (VAR == 10) {                                                                 \
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf8[j][0], af8[i][0], acc[i][j], 0, 0, 0);   \
                            corr[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf8[j][1], af8[i][0], corr[i][j], 0, 0, 0); \
                            corr[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf8[j][0], af8[i][1], corr[i][j], 0, 0, 0); \
                        } else {                                                                                   \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af8[i][0], bf8[j][0], acc[i][j], 0, 0, 0);     /* hi * hi */  \
                        corr[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af8[i][0], bf8[j][1], corr[i][j], 0, 0, 0);   /* hi * lo' */ \
                        corr[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af8[i][1], bf8[j][0], corr[i][j], 0, 0, 0);   /* lo' * hi */ \
                        }                                                                                          \
                    }                                                                                              \
            }                                                                                                      \
        }
        if constexpr (NST8 > 3) {
            constexpr int DEPTH = NST8 - 1;
            static_assert((DEPTH - 1) * PPW <= 63, "counted vmcnt wait");
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) FC_DMA8((d < KT ? d : KT - 1), d)
            int st = 0;
            for (int kt = 0; kt < KT; ++kt) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PPW) : "memory");   // this wave's pieces of tile kt have landed (DEPTH - 1 younger tiles may fly on)
                __builtin_amdgcn_s_barrier();                                   // ... and everybody's; everybody is done reading tile kt-1
                const int kn = kt + DEPTH < KT ? kt + DEPTH : KT - 1;           // (tail: harmless re-loads into the stage tile kt-1 just left)
                const int sn = st == 0 ? NST8 - 1 : st - 1;                     // (kt + DEPTH) % NST8
                FC_DMA8(kn, sn)
                FC_MMA8_STAGE(st)
                st = st == NST8 - 1 ? 0 : st + 1;
            }
        } else if constexpr (NST8 == 3) {
            FC_DMA8(0, 0)
            FC_DMA8((1 < KT ? 1 : KT - 1), 1)
            int st = 0;
            for (int kt = 0; kt < KT; ++kt) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");      // this wave's pieces of tile kt have landed (tile kt+1 may fly on)
                __builtin_amdgcn_s_barrier();                                   // ... and everybody's; everybody is done reading tile kt-1
                const int kn = kt + 2 < KT ? kt + 2 : KT - 1;                   // (tail: harmless re-loads into a stage nobody reads again)
                const int sn = st >= 1 ? st - 1 : 2;                            // (kt + 2) % 3
                FC_DMA8(kn, sn)
                FC_MMA8_STAGE(st)
                st = st == 2 ? 0 : st + 1;
            }
        } else {
            // two stages: tile kt+1 is in flight while tile kt is multiplied (issued right behind the barrier that frees its stage)
            FC_DMA8(0, 0)
            FC_STAMP(1)
            int st = 0;
            for (int kt = 0; kt < KT; ++kt) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (kt == 0) FC_STAMP(2)
                if (kt + 1 < KT) FC_DMA8(kt + 1, (st ^ 1))
                FC_MMA8_STAGE(st)
                st ^= 1;
            }
        }
#undef FC_MMA8_STAGE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the tail's re-loads: nothing may land in LDS once the epilogue owns it
        if constexpr (VAR != 10) __syncthreads();                           // (VAR 10's epilogue stays in registers: its waves finish independently)
        FC_STAMP(3)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += corr[i][j][r] * (1.0f / 2048.0f);
#undef FC_DMA8
    } else if constexpr (VAR >= 3) {
        // ================= split-bf16 main loop: fp32-equivalent products on the bf16 matrix cores =================
        // x = hi + mid + lo (three bf16 limbs, 24 significant bits);  a*b ~= ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh),
        // the dropped terms are below 2^-24 |a b|.  Every limb product is exact in the MFMA's fp32 accumulator, so the result
        // has fp32-GEMM accuracy (profiles/micro: 6e-9 rel. error with exact accumulation) at 6 bf16 MFMAs (32 cycles each)
        // per 32x32x16 block instead of 8 fp32 MFMAs (64 cycles each): 2.67x the matrix rate.  Weights are pre-split on the host
        // (PackedLinear.W3); activations are split while they are staged into LDS.  K tile = 16, LDS row = 3 limbs x 32 B + 16 B pad.
        //
        // VAR 5, the default: TWO fp16 limbs.  x = hi + lo'/2048 with hi = rn16(x), lo' = rn16((x - hi) * 2048): hi carries 11 bits,
        // the scaled remainder the next 11 (+ sign), so the pair represents x to 2^-24 relative -- fp32's own rounding unit -- and
        // the scaling keeps lo' out of fp16's subnormal range (abs. error floor 2^-36 per element).  a*b ~= ah*bh + (ah*bl' +
        // al'*bh)/2048: the h*h products go to the main accumulator, the two cross products to a second one that is scaled by
        // 2^-11 (exact) and added once after the k loop; the dropped l*l term is < 2^-24 |a b|.  3 MFMAs per block instead of 6
        // (5.3x the fp32-input matrix rate), 4 bytes per LDS element instead of 6.  fp16 overflows at 65504: every staged |x| is
        // max-reduced and a launch that met one >= 65504 raises *p.ovf; the entry point then repeats the whole call with the
        // bf16 limbs (unbounded range).  Weights with such entries never get an fp16 image (PackedLinear.W2 == nullptr).
        constexpr bool F16 = VAR >= 5;
        constexpr bool ALIMB = VAR == 7;                           // A arrives as the fp16 limb image its producer wrote (e.A16): plain copy
        constexpr int KS = 16;                                      // k extent of one LDS stage (32 with one-deep prefetch measured 13 % slower)
        constexpr int KSUB = KS / 16, U = 32 / KS;
        constexpr int NL = F16 ? 2 : 3;                             // limbs
        constexpr int LIMB_B = KS * 2;                              // bytes of one limb of a row
        constexpr int ROWB = NL * LIMB_B + 16;                      // bytes per LDS row (16 B pad: conflict-free 16-byte fragment reads)
        constexpr int CH = NL * KS / 8;                             // 16-byte chunks per (row, stage) of a limb image
        constexpr int STAGE3 = (BM + BN) * ROWB;
        constexpr int TPR = KS / 4;                                 // threads (float4s) per A row
        constexpr int RPP3 = NT / TPR, A3 = BM / RPP3, W3N = (BN * CH + NT - 1) / NT;   // float4 loads of A, 16-byte loads of W per thread and stage
        char* smc = reinterpret_cast<char*>(smem);
        // Row slots are dealt to lanes so that the lanes one LDS store cycle serves (16 for ds_write_b64, 8 for ds_write_b128;
        // stores see 32 banks) fall on distinct banks of the 80-byte-pitch image: with the natural order rows r and r+3 (b64)
        // or r and r+1 (b128) overlapped, a 2-way conflict on the CU's scarcest path (VGPR -> LDS, ~80 B/clk).
        const int rs3 = tid / TPR;
        const int lrow3 = F16 ? 8 * (rs3 >> 3) + ((rs3 >> 2) & 1) + 2 * (rs3 & 3) : rs3, lc3 = (tid % TPR) * 4;
#define FC_WROW(SLOT_) (F16 ? 8 * ((SLOT_) >> 3) + (((SLOT_) >> 1) & 3) + 4 * ((SLOT_) & 1) : (SLOT_))
        const int KT16 = p.KT * 2;
        const unsigned short* const Wl = F16 ? p.W2 : p.W3;
        float amax = 0.f;
        // two register sets: the tile loaded in iteration kt is only converted/stored in iteration kt+1, so a global load has a
        // whole iteration (the MFMAs of the other resident waves included) to land before anything waits for it
        float4 ra3_0[A3], ra3_1[A3], ra3_2[A3];            // (set 2: VAR 6 only, dead otherwise)
        constexpr int A4N = (BM * CH + NT - 1) / NT;                // 16-byte chunks of the A limb image per thread and stage
        typedef unsigned int u32xa __attribute__((ext_vector_type(4 * A4N)));
        u32xa ra4_0, ra4_1, ra4_2;
        typedef unsigned int u32xw __attribute__((ext_vector_type(4 * W3N)));      // whole-vector values: never an alloca, so never scratch
        u32xw rw3_0, rw3_1, rw3_2;
#define FC_GLOAD3(S_, KT_)                                                                                           \
        {                                                                                                          \
            const float* Ap_ = p.A[0];                                                                             \
            int lda_ = p.lda[0], kk_ = (KT_);                                                                      \
            if (kk_ >= U * p.kt[0]) {                                                                              \
                kk_ -= U * p.kt[0]; Ap_ = p.A[1]; lda_ = p.lda[1];                                                 \
                if (kk_ >= U * p.kt[1]) { kk_ -= U * p.kt[1]; Ap_ = p.A[2]; lda_ = p.lda[2]; }                     \
            }                                                                                                      \
            if constexpr (ALIMB) {                                                                                 \
                _Pragma("unroll") for (int i = 0; i < A4N; ++i) {                                                  \
                    int c_ = tid + NT * i;                                                                         \
                    c_ = c_ < BM * CH ? c_ : BM * CH - 1;                                                          \
                    const int slot_ = c_ / CH, part_ = c_ - slot_ * CH, row_ = FC_WROW(slot_);                     \
                    const uint4 t_ = *reinterpret_cast<const uint4*>(e.A16 + ((size_t)(m0 + row_) * KT16 + (KT_)) * (NL * 16) + part_ * 8); \
                    ra4_##S_[4 * i] = t_.x; ra4_##S_[4 * i + 1] = t_.y; ra4_##S_[4 * i + 2] = t_.z; ra4_##S_[4 * i + 3] = t_.w; \
                }                                                                                                  \
            } else {                                                                                               \
                const float* a_ = Ap_ + (size_t)(m0 + lrow3) * lda_ + kk_ * KS + lc3;                              \
                _Pragma("unroll") for (int i = 0; i < A3; ++i) ra3_##S_[i] = *reinterpret_cast<const float4*>(a_ + (size_t)(RPP3 * i) * lda_); \
            }                                                                                                      \
            _Pragma("unroll") for (int i = 0; i < W3N; ++i) {                                                      \
                int c_ = tid + NT * i;                                                                             \
                c_ = c_ < BN * CH ? c_ : BN * CH - 1;     /* unconditional load (a guarded one sends the staging registers through scratch) */ \
                const int slot_ = c_ / CH, part_ = c_ - slot_ * CH, row_ = FC_WROW(slot_);                         \
                const int sub_ = part_ / (NL * 2), q2_ = part_ - sub_ * (NL * 2);   /* k16 tile of the stage; limb*2 + half */ \
                const uint4 t_ = *reinterpret_cast<const uint4*>(Wl + ((size_t)(n0 + row_) * KT16 + (KT_) * KSUB + sub_) * (NL * 16) + q2_ * 8); \
                rw3_##S_[4 * i] = t_.x; rw3_##S_[4 * i + 1] = t_.y; rw3_##S_[4 * i + 2] = t_.z; rw3_##S_[4 * i + 3] = t_.w; \
            }                                                                                                      \
        }
#define FC_LSTORE3(S_, ST_)                                                                                          \
        {                                                                                                          \
            if constexpr (ALIMB) {                                                                                 \
                _Pragma("unroll") for (int i = 0; i < A4N; ++i) {                                                  \
                    const int c_ = tid + NT * i, slot_ = c_ / CH, part_ = c_ - slot_ * CH, row_ = FC_WROW(slot_);  \
                    if (BM * CH % NT == 0 || c_ < BM * CH)                                                         \
                        *reinterpret_cast<uint4*>(smc + (ST_) * STAGE3 + row_ * ROWB + part_ * 16) =               \
                            make_uint4(ra4_##S_[4 * i], ra4_##S_[4 * i + 1], ra4_##S_[4 * i + 2], ra4_##S_[4 * i + 3]); \
                }                                                                                                  \
            }                                                                                                      \
            char* sa_ = smc + (ST_) * STAGE3 + lrow3 * ROWB + (tid % TPR) * 8;                                      \
            _Pragma("unroll") for (int i = 0; i < (ALIMB ? 0 : A3); ++i) {                                         \
                const float x_[4] = {ra3_##S_[i].x, ra3_##S_[i].y, ra3_##S_[i].z, ra3_##S_[i].w};                  \
                if constexpr (F16) {                                                                               \
                    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(x_[0]), fabsf(x_[1]))), fmaxf(fabsf(x_[2]), fabsf(x_[3])));  \
                    uint2 h_, l_;                                  /* five VALU per pair of values (activations.h limb_split2) */ \
                    limb_split2(x_[0], x_[1], h_.x, l_.x);                                                         \
                    limb_split2(x_[2], x_[3], h_.y, l_.y);                                                         \
                    *reinterpret_cast<uint2*>(sa_ + RPP3 * i * ROWB) = h_;                                          \
                    *reinterpret_cast<uint2*>(sa_ + RPP3 * i * ROWB + LIMB_B) = l_;                                 \
                } else {                                                                                           \
                    bf16x4 h_, m_, l_;                                                                             \
                    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                             \
                        h_[e_] = (__bf16)x_[e_];                                                                   \
                        const float r1_ = x_[e_] - (float)h_[e_];                                                  \
                        m_[e_] = (__bf16)r1_;                                                                      \
                        l_[e_] = (__bf16)(r1_ - (float)m_[e_]);                                                    \
                    }                                                                                              \
                    *reinterpret_cast<bf16x4*>(sa_ + RPP3 * i * ROWB) = h_;                                         \
                    *reinterpret_cast<bf16x4*>(sa_ + RPP3 * i * ROWB + LIMB_B) = m_;                                \
                    *reinterpret_cast<bf16x4*>(sa_ + RPP3 * i * ROWB + 2 * LIMB_B) = l_;                            \
                }                                                                                                  \
            }                                                                                                      \
            _Pragma("unroll") for (int i = 0; i < W3N; ++i) {                                                      \
                const int c_ = tid + NT * i, slot_ = c_ / CH, part_ = c_ - slot_ * CH, row_ = FC_WROW(slot_);      \
                const int sub_ = part_ / (NL * 2), q2_ = part_ - sub_ * (NL * 2);                                  \
                if (BN * CH % NT == 0 || c_ < BN * CH)                                                             \
                    *reinterpret_cast<uint4*>(smc + (ST_) * STAGE3 + (BM + row_) * ROWB + (q2_ >> 1) * LIMB_B + sub_ * 32 + (q2_ & 1) * 16) = \
                        make_uint4(rw3_##S_[4 * i], rw3_##S_[4 * i + 1], rw3_##S_[4 * i + 2], rw3_##S_[4 * i + 3]);   \
            }                                                                                                      \
        }
#define FC_MMA3(ST_)                                                                                              \
        _Pragma("unroll") for (int sub = 0; sub < KSUB; ++sub) {                                                   \
            const char* sA = smc + (ST_) * STAGE3 + (wr * TM * 32 + li) * ROWB + lh * 16 + sub * 32;                \
            const char* sB = smc + (ST_) * STAGE3 + (BM + wc * TN * 32 + li) * ROWB + lh * 16 + sub * 32;           \
            if constexpr (F16) {                                                                                   \
                f16x8 af3[TM][2], bf3[TN][2];                                                                      \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                     \
                    _Pragma("unroll") for (int q = 0; q < 2; ++q) af3[i][q] = *reinterpret_cast<const f16x8*>(sA + i * 32 * ROWB + q * LIMB_B); \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                     \
                    _Pragma("unroll") for (int q = 0; q < 2; ++q) bf3[j][q] = *reinterpret_cast<const f16x8*>(sB + j * 32 * ROWB + q * LIMB_B); \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                     \
                    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                               \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af3[i][0], bf3[j][0], acc[i][j], 0, 0, 0);     /* hi * hi */  \
                        corr[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af3[i][0], bf3[j][1], corr[i][j], 0, 0, 0);   /* hi * lo' */ \
                        corr[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af3[i][1], bf3[j][0], corr[i][j], 0, 0, 0);   /* lo' * hi */ \
                    }                                                                                              \
            } else {                                                                                               \
                bf16x8 af3[TM][3], bf3[TN][3];                                                                     \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                     \
                    _Pragma("unroll") for (int q = 0; q < 3; ++q) af3[i][q] = *reinterpret_cast<const bf16x8*>(sA + i * 32 * ROWB + q * LIMB_B); \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                     \
                    _Pragma("unroll") for (int q = 0; q < 3; ++q) bf3[j][q] = *reinterpret_cast<const bf16x8*>(sB + j * 32 * ROWB + q * LIMB_B); \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                     \
                    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                               \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[i][2], bf3[j][0], acc[i][j], 0, 0, 0);   /* lo * hi */   \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[i][1], bf3[j][1], acc[i][j], 0, 0, 0);   /* mid * mid */ \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[i][0], bf3[j][2], acc[i][j], 0, 0, 0);   /* hi * lo */   \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[i][1], bf3[j][0], acc[i][j], 0, 0, 0);   /* mid * hi */  \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[i][0], bf3[j][1], acc[i][j], 0, 0, 0);   /* hi * mid */  \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[i][0], bf3[j][0], acc[i][j], 0, 0, 0);   /* hi * hi */   \
                    }                                                                                              \
            }                                                                                                      \
        }
        floatx16 corr[F16 ? TM : 1][F16 ? TN : 1];
        if constexpr (F16) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) corr[i][j][r] = 0.f;
        }
        if constexpr (VAR == 6) {
            // THREE register sets: a tile's global loads are issued three stages before its MFMAs (two in registers, one in LDS).  The
            // Linear launches of the flow stream their A operand from HBM / Infinity Cache once (no column-tile reuse to speak of at
            // N <= 512), 64 KB in flight per CU did not cover that latency (SQ_WAIT_ANY 43 % of the wave cycles at 25 % matrix-pipe busy).
            // LDS stays double-buffered: stage (kt + 1) & 1 was last read in iteration kt - 1, one barrier back.
            FC_GLOAD3(0, 0)
            FC_GLOAD3(1, (1 < KT16 ? 1 : KT16 - 1))
            FC_GLOAD3(2, (2 < KT16 ? 2 : KT16 - 1))
            FC_LSTORE3(0, 0)
            __syncthreads();
            int par = 0;
            for (int kt = 0; kt < KT16; kt += 3) {
                const int k3 = kt + 3 < KT16 ? kt + 3 : KT16 - 1, k4 = kt + 4 < KT16 ? kt + 4 : KT16 - 1, k5 = kt + 5 < KT16 ? kt + 5 : KT16 - 1;
                FC_GLOAD3(0, k3)
                FC_MMA3(par)
                FC_LSTORE3(1, (par ^ 1))
                __syncthreads();
                par ^= 1;
                if (kt + 1 < KT16) {
                    FC_GLOAD3(1, k4)
                    FC_MMA3(par)
                    FC_LSTORE3(2, (par ^ 1))
                    __syncthreads();
                    par ^= 1;
                }
                if (kt + 2 < KT16) {
                    FC_GLOAD3(2, k5)
                    FC_MMA3(par)
                    FC_LSTORE3(0, (par ^ 1))
                    __syncthreads();
                    par ^= 1;
                }
            }
        } else {
        // KT16 is even (K_pad is a multiple of 32).  Stage s of LDS holds tile kt (s = kt & 1); register set s holds tile kt+1 ... kt+2.
        FC_GLOAD3(0, 0)
        FC_LSTORE3(0, 0)
        FC_GLOAD3(1, 1)
        __syncthreads();
        for (int kt = 0; kt < KT16; kt += 2) {
            const int k2 = kt + 2 < KT16 ? kt + 2 : KT16 - 1, k3 = kt + 3 < KT16 ? kt + 3 : KT16 - 1;   // tail re-loads: branch-free loop
            FC_GLOAD3(0, k2)
            FC_MMA3(0)
            FC_LSTORE3(1, 1)
            __syncthreads();
            FC_GLOAD3(1, k3)
            FC_MMA3(1)
            FC_LSTORE3(0, 0)
            __syncthreads();
        }
        }
        if constexpr (F16) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += corr[i][j][r] * (1.0f / 2048.0f);
            if (amax >= 65504.0f) atomicOr(p.ovf, 1);               // some |x| does not fit fp16: the caller repeats with bf16 limbs
        }
#undef FC_MMA3
#undef FC_WROW
#undef FC_GLOAD3
#undef FC_LSTORE3
    } else {
    // ---- global -> register staging: thread t moves float4 (t&7) of rows (t>>3) + 32*i.  Straight-line code on plain
    //      register arrays (no lambdas / conditionals around them: those sent the staging tile through scratch memory).
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
    float4 ra[A_F4], rb[B_F4];
    const float* wbase = p.W + (size_t)(n0 + lrow) * p.K_pad + lc4;
    const size_t wstep = (size_t)RPP * p.K_pad;
    float* const sAst = smem + lrow * LDS_LD + lc4;
    float* const sBst = sAst + BM * LDS_LD;

#define FC_GLOAD(KT_)                                                                                              \
    {                                                                                                              \
        const float* Ap_ = p.A[0];                                                                                 \
        int lda_ = p.lda[0], kk_ = (KT_);                                                                          \
        if (kk_ >= p.kt[0]) {                                                                                      \
            kk_ -= p.kt[0]; Ap_ = p.A[1]; lda_ = p.lda[1];                                                         \
            if (kk_ >= p.kt[1]) { kk_ -= p.kt[1]; Ap_ = p.A[2]; lda_ = p.lda[2]; }                                 \
        }                                                                                                          \
        const float* a_ = Ap_ + (size_t)(m0 + lrow) * lda_ + kk_ * 32 + lc4;                                       \
        _Pragma("unroll") for (int i = 0; i < A_F4; ++i) ra[i] = *reinterpret_cast<const float4*>(a_ + (size_t)(RPP * i) * lda_); \
        const float* w_ = wbase + (KT_) * 32;                                                                      \
        _Pragma("unroll") for (int i = 0; i < B_F4; ++i) rb[i] = *reinterpret_cast<const float4*>(w_ + i * wstep);  \
    }
#define FC_LSTORE(STAGE_)                                                                                          \
    {                                                                                                              \
        float* sa_ = sAst + (STAGE_) * STAGE;                                                                      \
        float* sb_ = sBst + (STAGE_) * STAGE;                                                                      \
        _Pragma("unroll") for (int i = 0; i < A_F4; ++i) *reinterpret_cast<float4*>(sa_ + RPP * i * LDS_LD) = ra[i]; \
        _Pragma("unroll") for (int i = 0; i < B_F4; ++i) *reinterpret_cast<float4*>(sb_ + RPP * i * LDS_LD) = rb[i]; \
    }

    FC_GLOAD(0)
    FC_LSTORE(0)
    __syncthreads();

    float4 af[2][TM], bf[2][TN];
    for (int kt = 0; kt < p.KT; ++kt) {
        const int ktn = kt + 1 < p.KT ? kt + 1 : kt;        // last iteration re-loads its own tile: keeps the loop branch free
        FC_GLOAD(ktn)
        const float* sA = smem + (kt & 1) * STAGE + (wr * TM * 32 + li) * LDS_LD + 4 * lh;
        const float* sB = smem + (kt & 1) * STAGE + BM * LDS_LD + (wc * TN * 32 + li) * LDS_LD + 4 * lh;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const float4*>(sA + i * 32 * LDS_LD);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const float4*>(sB + j * 32 * LDS_LD);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cb = g & 1, nb = cb ^ 1;
            if (g < 3) {                                   // fragments of the next 8-deep k group fly while this group's MFMAs run
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nb][i] = *reinterpret_cast<const float4*>(sA + i * 32 * LDS_LD + 8 * (g + 1));
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nb][j] = *reinterpret_cast<const float4*>(sB + j * 32 * LDS_LD + 8 * (g + 1));
            } else if (VAR == 2) {
                FC_LSTORE((kt + 1) & 1)                     // next tile's LDS image is written under the last group's MFMAs
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][i].x, bf[cb][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][i].y, bf[cb][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][i].z, bf[cb][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][i].w, bf[cb][j].w, acc[i][j], 0, 0, 0);
                }
            }
        }
        if (VAR != 2) FC_LSTORE((kt + 1) & 1)
        __syncthreads();
    }
#undef FC_GLOAD
#undef FC_LSTORE

    }

    // ------------------------------------------------------------------ epilogues
    // C/D layout of the 32x32 MFMA: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5), r = 0..15
    if constexpr (EPI == EPI_LINEAR) {
        if ((VAR == 8 || VAR == 9) && e.inverse == 2) return;        // (diagnostic knob 14 = 2: main loop only, results invalid)
        if constexpr (RES_EARLY) {
            if (e.residual16 && nvalid > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[0][0][r] += (float)__builtin_bit_cast(_Float16, res_h[r]) + (float)__builtin_bit_cast(_Float16, res_l[r]) * (1.0f / 2048.0f);
            }
        } else if (e.residual16) {
            // residual from the limb image its producer wrote (hidden activations of a limb-chained MLP exist only in that form): v = (bias + sum) + residual
            const int blocks = e.ldr16 >> 4;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (j < nvalid) {
                    const int col = wave_n0 + j * 32 + li;
                    const unsigned short* rp = e.residual16 + ((size_t)(wave_m0 + 4 * lh) * blocks + (col >> 4)) * 32 + (col & 15);
                    unsigned short th[TM][16], tl[TM][16];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const unsigned short* q = rp + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * blocks * 32;
                            th[i][r] = q[0]; tl[i][r] = q[16];
                        }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[i][j][r] += (float)__builtin_bit_cast(_Float16, th[i][r]) + (float)__builtin_bit_cast(_Float16, tl[i][r]) * (1.0f / 2048.0f);
                }
            }
        }

        // The activation and the output format are wave-uniform run-time values: they are dispatched ONCE, outside the element loops
        // (a `switch (act)` per element compiled to ~12 branches per output value -- incl. the ELU path's expm1f -- and cost the
        // 256x128 tile 18 us per tile, 40 % of a 512 -> 512 layer; round 2).  Each body below is straight-line code over the tile.
        float omax = 0.f;
        // 64-bit bases once per wave, 32-bit offsets inside the tile (a size_t product per element cost two 64-bit multiply-adds each)
        float* const cbase = e.C ? e.C + (size_t)wave_m0 * e.ldc + wave_n0 : nullptr;
        // the training epilogues (pre-activation copy, activation gradient) exist on the fp32-A loops only -- the ones the fc_train_* entries launch
        constexpr bool TRAIN_EPI = VAR == 2 || VAR == 3 || VAR == 5;
        float* const pbase = e.Cpre ? e.Cpre + (size_t)wave_m0 * e.ldc + wave_n0 : nullptr;
        const float* const gbase = e.gradu ? e.gradu + (size_t)wave_m0 * e.ldgu + wave_n0 : nullptr;
        const int rp16 = (p.N_pad >> 4) * 32;                          // ushorts per row of the limb image
        unsigned short* const hbase = e.C16 ? e.C16 + (size_t)wave_m0 * rp16 + (size_t)(wave_n0 >> 4) * 32 : nullptr;
        const float c16_s1 = e.c16_scale > 0.f ? e.c16_scale : 1.0f, c16_s2 = e.c16_scale > 0.f ? 1.0f : 2048.0f;
        auto body = [&](auto act_tag, auto fmt_tag) {
            constexpr int ACT = decltype(act_tag)::value;             // 16 + a: no activation, the value is multiplied by act_a'(gradu[row][col]) instead
            constexpr int FMT = decltype(fmt_tag)::value;             // 1: fp32 C, 2: limb image C16, 3: both; 5: fp32 C + the pre-activation value in Cpre
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (j < nvalid) {
                    const int cl = j * 32 + li;                       // column inside the wave's strip
                    const int c0 = cl & ~1;
                    const int hoff = (c0 >> 4) * 32 + ((li & 1) ? 16 : 0) + (c0 & 15);
                    float gu[ACT >= 16 ? TM : 1][16];                 // the block's act' arguments, requested up front (independent loads)
                    if constexpr (ACT >= 16) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) gu[i][r] = gbase[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * e.ldgu + cl];
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;      // row inside the wave's strip
                            float v = acc[i][j][r];
                            if constexpr (ACT >= 16) v *= fc_act_grad(gu[i][r], ACT - 16);
                            if constexpr ((FMT & 4) != 0) pbase[rl * e.ldc + cl] = v;
                            if constexpr (ACT == FC_ACT_GELU) v = fc_gelu(v);
                            else if constexpr (ACT == FC_ACT_RELU) v = v > 0.f ? v : 0.f;
                            else if constexpr (ACT == FC_ACT_ELU) v = v > 0.f ? v : expm1f(v);
                            else if constexpr (ACT == FC_ACT_LRELU02) v = v > 0.f ? v : 0.2f * v;
                            if constexpr (FMT & 1) cbase[rl * e.ldc + cl] = v;
                            if constexpr (FMT & 2) {
                                // the output ALSO / ONLY as the fp16 limb image a following split-fp16 GEMM copies (its 30 column tiles
                                // would each re-split the same rows): lanes (c, c+1) swap one half through a DPP quad permute
                                // ([1,0,3,2]: a VALU move, not the LDS round trip __shfl_xor compiles to) and store one 32-bit word each
                                // (c16_s1, c16_s2) = (1, 2048): x = hi + lo'/2048; (kOneAccActScale, 1): the one-accumulator form hi + lo of x s1 (common.h)
                                const float vs = v * c16_s1;
                                omax = fmaxf(omax, fabsf(vs));
                                const _Float16 hb = (_Float16)vs;
                                const _Float16 lb = (_Float16)((vs - (float)hb) * c16_s2);
                                const unsigned hu = __builtin_bit_cast(unsigned short, hb), lu = __builtin_bit_cast(unsigned short, lb);
                                const unsigned mine = (li & 1) ? lu : hu, give = (li & 1) ? hu : lu;
                                const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)give, 0xB1, 0xF, 0xF, true);
                                const unsigned word = (li & 1) ? (got | (mine << 16)) : (mine | (got << 16));
                                *reinterpret_cast<unsigned*>(hbase + rl * rp16 + hoff) = word;
                            }
                        }
                    }
                }
            }
        };
        auto by_fmt = [&](auto act_tag) {
            if constexpr (TRAIN_EPI) { if (e.Cpre) { body(act_tag, std::integral_constant<int, 5>{}); return; } }   // (the launcher admits Cpre only beside C, without C16)
            if (e.C && e.C16) body(act_tag, std::integral_constant<int, 3>{});
            else if (e.C16) body(act_tag, std::integral_constant<int, 2>{});
            else body(act_tag, std::integral_constant<int, 1>{});
        };
        bool grad_done = false;
        if constexpr (TRAIN_EPI) {
            if (e.gradu) {                                           // (the launcher admits it with an fp32 C only and no activation)
                switch (e.gact) {
                    case FC_ACT_GELU: body(std::integral_constant<int, 16 + FC_ACT_GELU>{}, std::integral_constant<int, 1>{}); break;
                    case FC_ACT_RELU: body(std::integral_constant<int, 16 + FC_ACT_RELU>{}, std::integral_constant<int, 1>{}); break;
                    default: body(std::integral_constant<int, 16 + FC_ACT_ELU>{}, std::integral_constant<int, 1>{}); break;
                }
                grad_done = true;
            }
        }
        if (!grad_done)
        switch (e.act) {
            case FC_ACT_GELU: by_fmt(std::integral_constant<int, FC_ACT_GELU>{}); break;
            case FC_ACT_RELU: by_fmt(std::integral_constant<int, FC_ACT_RELU>{}); break;
            case FC_ACT_ELU: by_fmt(std::integral_constant<int, FC_ACT_ELU>{}); break;
            case FC_ACT_LRELU02: by_fmt(std::integral_constant<int, FC_ACT_LRELU02>{}); break;
            default: by_fmt(std::integral_constant<int, FC_ACT_NONE>{}); break;
        }
        if (omax >= 65504.0f) atomicOr(p.ovf, 1);                 // (omax stays 0 without a limb-image output)
        if constexpr (VAR == 8 || VAR == 9) {
            FC_STAMP(6)
            if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + 9] = wall_clock64();
        }
    } else if constexpr (EPI == EPI_LNQ) {
        // ---- LayerNorm folded through the layer (common.h): a wave's 64 columns are either hidden columns (sum of squares per row
        //      into the block's slot) or the 64 q columns (stored un-normalised)
        static_assert(TN == 2, "LNQ epilogue: a wave owns one 64-column block (hidden columns or the q columns)");
        if (wave_n0 < e.d2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float t = acc[i][0][r] * acc[i][0][r] + acc[i][1][r] * acc[i][1][r];
                    t = half_wave_sum(t);
                    if (li == 0) e.ldj_part[(size_t)(wave_n0 >> 6) * e.ldj_pitch + wave_m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = t;
                }
        } else if (wave_n0 < e.d2 + 64) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        e.C[(size_t)(wave_m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * e.ldc + (wave_n0 - e.d2) + j * 32 + li] = acc[i][j][r];
        }
    } else if constexpr (EPI == EPI_SPLINE && VAR == 10) {
        // ---- fused rational-quadratic spline coupling evaluated from the accumulator registers (transposed product, K = 8).
        //      Slot s = 16 j + r of this lane is tile column spline_slot_col(s, lh): slots 0..24 / 25..49 are dims 2 lh / 2 lh + 1,
        //      slots 50.. of the lower half are parameters 0..13 of dim 4, slots 50..60 of the upper half its parameters 14..24.
        if (e.inverse == 2) return;                                  // (diagnostic knob 14: main loop only)
        auto P = [&](int s) -> float { return acc[0][s >> 4][s & 15]; };
        float t4[11];
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            t4[i] = upper_to_lower(acc[0][3][2 + i]);                 // slot 50 + i
        }
        FC_STAMP(4)
        const int row = m0 + wave * 32 + li, dim0 = bn * 5;
        const bool rv = row < e.rows_valid;
        const bool vA = rv && dim0 + 2 * lh < e.d2, vB = rv && dim0 + 2 * lh + 1 < e.d2, vC = rv && lh == 0 && dim0 + 4 < e.d2;
        float yA, yB, yC, lA, lB, lC;
        if (e.inverse == 1) {                                        // (diagnostic knob 14 = 1: no spline evaluation)
            yA = spl_x[0] + P(0); lA = P(1); yB = spl_x[1] + P(25); lB = P(26); yC = spl_x[2] + P(50); lC = P(51);
        } else {
            rq_spline_fwd_regs<8>(spl_x[0], [&](int q) { return P(q); }, yA, lA);
            rq_spline_fwd_regs<8>(spl_x[1], [&](int q) { return P(25 + q); }, yB, lB);
            rq_spline_fwd_regs<8>(spl_x[2], [&](int q) { return q < 14 ? P(50 + q) : t4[q - 14]; }, yC, lC);
        }
        FC_STAMP(5)
        float* xr = e.xbuf + (size_t)row * e.ldx + e.x2_col0 + dim0;
        if (vA) xr[2 * lh] = yA;
        if (vB) xr[2 * lh + 1] = yB;
        if (vC) xr[4] = yC;
        lA = vA ? lA : 0.f; lB = vB ? lB : 0.f; lC = vC ? lC : 0.f;
        // log-dets of dims 2, 3 cross to the lower half; summed in dim order like the LDS-tile epilogues (bit-identical slot values)
        const float l2 = upper_to_lower(lA), l3 = upper_to_lower(lB);
        if (lh == 0) {
            float sum = 0.f;
            sum += lA; sum += lB; sum += l2; sum += l3; sum += lC;
            e.ldj_part[(size_t)bn * e.ldj_pitch + row] = spl_ldj + sum;
        }
        FC_STAMP(6)
        if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + 9] = wall_clock64();
    } else if constexpr (EPI == EPI_SPLINE) {
        // ---- fused rational-quadratic spline coupling (forward).  The parameter layer's columns are laid out so that this
        //      128-column tile holds all 3K+1 parameters of DPT transformed dims (spline.h): the tile goes through LDS (the
        //      accumulator layout has one parameter per lane), then each thread evaluates whole splines.  Nothing of the
        //      [rows, 25*d2] parameter matrix is written to or re-read from HBM.
        static_assert(BN == 128, "spline epilogue: the column layout is built for 128-column tiles");
        constexpr int TP = BN + 1;                                   // odd pitch: lanes walk rows conflict-free
        if (e.inverse == 2) return;                                  // (diagnostic knob 14: main loop only)
        float* tile = smem;                                          // aliases the staging buffers (all reads are behind the loop's last barrier)
        float* part = smem + BM * TP;                                // [DPT][BM] log-det terms
        if (e.inverse != 3) {                                        // (diagnostic knob 14 = 3: no parameter-tile write)
        int tpos[TN];                                                // column -> (dim, parameter) position (K = 8: spline.h's slot order)
#pragma unroll
        for (int j = 0; j < TN; ++j) tpos[j] = spline_tile_pos(wc * TN * 32 + j * 32 + li, e.spline_K);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tile[(wr * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TP + tpos[j]] = acc[i][j][r];
        } else { asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[TM - 1][TN - 1][15])); }
        __syncthreads();
        if constexpr (VAR == 8 || VAR == 9) FC_STAMP(4)
        if (e.inverse == 5) return;                                  // (diagnostic knob 14 = 5: main loop + parameter-tile write + barrier)
        const int K = e.spline_K, per = 3 * K + 1, DPT = BN / per;
        const int dim0 = bn * DPT;
        if (DPT == 5) {
#pragma unroll
            for (int k = 0; k < SPL_PER_THREAD; ++k) {                  // K = 8: x2 arrives from the prefetch at the top of the kernel
                const int it = tid + k * NT, row = it % BM, dl = it / BM;
                if (it < BM * DPT) {
                    float lad = 0.f;
                    if (dim0 + dl < e.d2 && m0 + row < e.rows_valid) {
                        float y;
                        if (e.inverse == 1) { y = spl_x[k] + tile[row * TP + dl * per]; lad = tile[row * TP + dl * per + 1]; }   // (diagnostic knob 14: no spline evaluation)
                        else rq_spline_fwd<8>(spl_x[k], tile + row * TP + dl * per, 1, y, lad);        // DPT == 5 <=> K == 8
                        if (e.inverse != 4) e.xbuf[(size_t)(m0 + row) * e.ldx + e.x2_col0 + dim0 + dl] = y;      // (diagnostic knob 14 = 4: no x2 store)
                        else asm volatile("" :: "v"(y));
                    }
                    part[dl * BM + row] = lad;
                }
            }
        } else {
            for (int it = tid; it < BM * DPT; it += NT) {
                const int row = it % BM, dl = it / BM;
                float lad = 0.f;
                if (dim0 + dl < e.d2 && m0 + row < e.rows_valid) {
                    float* xp = e.xbuf + (size_t)(m0 + row) * e.ldx + e.x2_col0 + dim0 + dl;
                    float y;
                    rq_any(K, *xp, tile + row * TP + dl * per, 1, false, y, lad);
                    *xp = y;
                }
                part[dl * BM + row] = lad;
            }
        }
        if constexpr (VAR == 8 || VAR == 9) FC_STAMP(5)
        __syncthreads();
        if (tid < BM) {
            float sum = 0.f;
            for (int dl = 0; dl < DPT; ++dl) sum += part[dl * BM + tid];
            e.ldj_part[(size_t)bn * e.ldj_pitch + m0 + tid] = spl_ldj + sum;      // this (tile, row) slot has one owner per launch: reproducible
        }
        if constexpr (VAR == 8 || VAR == 9) {
            FC_STAMP(6)
            if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + 9] = wall_clock64();
        }
    } else {
        static_assert(EPI == EPI_LINEAR || (TN % 2 == 0), "pair-packed epilogues need an even number of column tiles");
        float lsum[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) lsum[i][r] = 0.f;
#pragma unroll
        for (int pr = 0; pr < TN / 2; ++pr) {
            if (2 * pr + 1 < nvalid) {
                const int col_s = wave_n0 + (2 * pr) * 32 + li, col_t = col_s + 32;
                const int j = (wave_n0 / 64 + pr) * 32 + li;        // index of the transformed / noise dim
                const float bs = p.bias[col_s], bt = p.bias[col_t];
                if (j < e.d2) {
                    // The x2 operands of the whole pair block FIRST, as independent loads: read inside the element loop (`*xp = *xp * s + t`)
                    // every load sat behind the previous element's store to the same buffer -- hipcc cannot tell the rows apart -- so a lane
                    // waited out one memory latency per element, 32 in a row: 17 of an affine tile's 53 us (round 3; C4 -5 %, C3 -4 %).
                    float xv[TM][16];
                    float* xcol = nullptr;
                    float gsc = 1.0f;
                    if constexpr (EPI == EPI_AFFINE) {
                        xcol = e.xbuf + e.x2_col0 + (j < e.split ? j : e.split_pad + (j - e.split));
                        gsc = e.post_scale ? e.post_scale[j] : 1.0f;
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                xv[i][r] = xcol[(size_t)(wave_m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * e.ldx];
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = wave_m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            const float u = acc[i][2 * pr][r] + bs, t = acc[i][2 * pr + 1][r] + bt;
                            if constexpr (EPI == EPI_AFFINE) {
                                // models/affine_coupling.py:23-62: s = exp(u) | (2 sigmoid(u) - 1)(1 - 1e-8) + 1 ; y2 = x2 s + t ; x2 = (y2 - t)/s
                                // (round 3: the hardware transcendentals the spline path uses, ~8 instead of ~40 VALU per pair: with e = exp(-u),
                                // (2 sigmoid(u) - 1)(1 - 1e-8) + 1 = 2 / (1 + e) in fp32 -- (float)(1 - 1e-8) IS 1.0f -- and log s = ln 2 - ln(1 + e);
                                // e = inf (u < -88.7) gives s = 0, log s = -inf like the reference's sigmoid)
                                float sc, lsc;
                                if (e.scale_fn == FC_SCALE_EXP) { sc = __builtin_amdgcn_exp2f(u * 1.4426950408889634f); lsc = __builtin_amdgcn_logf(sc) * 0.69314718055994530942f; }   // (log of the ROUNDED s, +-inf included, as the reference takes it)
                                else {
                                    const float ope = 1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f);
                                    sc = 2.0f * __builtin_amdgcn_rcpf(ope);
                                    lsc = (1.0f - __builtin_amdgcn_logf(ope)) * 0.69314718055994530942f;
                                }
                                const float g = gsc;
                                float* xp = xcol + (size_t)row * e.ldx;
                                if (e.inverse) *xp = (xv[i][r] - t) / (sc * g);
                                else { *xp = xv[i][r] * (sc * g) + t; lsum[i][r] += lsc; }
                            } else if constexpr (EPI == EPI_AUGMENT) {
                                // models/augmenter.py:49-63 + distributions.py:128-153: z2 = mu + eps*sigma, ldj = -log N(z2; mu, sigma)
                                float sigma = expf(t);
                                if (e.clamp > 0.f) sigma = fminf(sigma, e.clamp);
                                const float ev = row < e.rows_valid ? e.eps[(size_t)row * e.d2 + j] : 0.f;
                                float z = u + ev * sigma;
                                const float dz = z - u;
                                const float lp = -(dz * dz) / (2.0f * sigma * sigma) - logf(sigma) - 0.91893853320467274178f;
                                if (e.val_scale) z = z / e.val_scale[j] + e.val_shift[j];     // CIF Slice.inverse: undo the ActNorm of the z2 part
                                const int idx = e.d_in + j;
                                const int col = idx < e.d1 ? idx : e.d1_pad + (idx - e.d1);
                                e.xbuf[(size_t)row * e.ldx + col] = z;
                                if (!e.inverse) lsum[i][r] -= lp;
                            } else {
                                // models/slice.py:31-44 + distributions.py:140-142: ldj = +log N(x2; mu(z), sigma(z))
                                float sigma = expf(t);
                                if (e.clamp > 0.f) sigma = fminf(sigma, e.clamp);
                                float v = e.val[(size_t)row * e.ldval + j];
                                if (e.val_scale) v = (v - e.val_shift[j]) * e.val_scale[j];
                                const float dz = v - u;
                                lsum[i][r] += -(dz * dz) / (2.0f * sigma * sigma) - logf(sigma) - 0.91893853320467274178f;
                            }
                        }
                    }
                }
            }
        }
        if (e.inverse) return;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float tot = half_wave_sum(lsum[i][r]);
                const int row = wave_m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (li == 0 && row < e.rows_valid) {
                    if (WN > 1 || BN < 320) e.ldj_part[(size_t)(bn * WN + wc) * e.ldj_pitch + row] += tot;     // own slot: reproducible
                    else if (p.nbn * WN > 1) atomicAdd(e.logprob + row, tot);
                    else e.logprob[row] += tot;
                }
            }
        }
    }
}

// ===================================================================================================================================
// VAR 11: the fused spline layer as a PERSISTENT transposed LDS-DMA GEMM (K = 8 bins).  Same tile, operands, MFMA order and register
// epilogue as VAR 10 (bit-identical results); what changes is the tile boundary, which the in-kernel stamps (knob 20,
// profiles/micro/spline_gemm_stamps.py) priced at 6.5 of a workgroup's 25.5 us per tile: 2.6 us from entry until the first k tile has
// landed, 0.8 us between a workgroup's exit and its successor's entry, 3.1 us of epilogue during which the slot fetches nothing.
//   * grid = 2 workgroups per CU, each walks tiles t = blockIdx.x, + gridDim.x, ... (the XCD-aware order of the one-tile-per-workgroup
//     launch: gridDim.x is a multiple of 8, so a workgroup's tiles stay on its XCD's band);
//   * ONE continuous DMA stream: behind the barrier of a tile's LAST k step the workgroup issues the NEXT tile's first k step into the
//     free stage (plus its 512 bytes of bias into LDS and its x2 / log-det operands into registers), so that data crosses the
//     epilogue in flight and the next tile's first barrier finds it landed;
//   * the epilogue never touches LDS and has no barrier: the four waves evaluate their splines independently, results stay in four
//     registers and are STORED behind the next tile's first barrier, so no wave waits for a store acknowledgement (stores count in
//     vmcnt on gfx9) before it may start multiplying again.
// LDS: 2 stages x 32 KB + 2 x 512 B of bias = 66560 B (two workgroups per CU).
__device__ __forceinline__ void spline_gemm_persistent(const GemmParams& p, float* smem) {
    constexpr int BM = 128, BN = 128, ROWB = 128, STAGE = (BM + BN) * ROWB, PPW = 8;
    typedef __attribute__((address_space(3))) char lds_char;
    typedef const __attribute__((address_space(1))) char glb_char;
    const GemmEpi& e = p.e;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* smc = reinterpret_cast<char*>(smem);
    float* biasbuf = smem + 2 * STAGE / 4;                              // [2][128]
    const int KT = p.KT;
    const unsigned rowbytes = (unsigned)KT * 128u;
    const int ntiles = p.nbm * p.nbn, G = gridDim.x;
    int t = blockIdx.x;
    if (t >= ntiles) return;

    auto tile_of = [&](int b, int& bm, int& bn) {                       // the XCD-aware order of gemm_f32_kernel
        const int xcd = b & 7, loc = b >> 3;
        if (p.col_group > 0) {
            const int rows_x = p.nbm >> 3, Gc = p.col_group;
            const int g = loc / (rows_x * Gc);
            const int rem = loc - g * rows_x * Gc;
            const int w = p.nbn - g * Gc < Gc ? p.nbn - g * Gc : Gc;
            const int r = rem / w;
            bm = xcd * rows_x + r;
            bn = g * Gc + (rem - r * w);
        } else {
            const int q = ntiles >> 3, r = ntiles & 7;
            const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
            bm = L / p.nbn;
            bn = L - bm * p.nbn;
        }
    };
    // DMA pieces: piece pc = wave * 8 + i covers stage rows 8 pc .. 8 pc + 7 (rows 0..127: points, 128..255: weight rows); waves 0, 1 fetch
    // the points, waves 2, 3 the weights, so a wave's source base is scalar and its eight per-lane byte offsets never change
    unsigned poff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int r = (wave * PPW + i) * 8 + (lane >> 3);
        const int cl = (lane & 7) ^ ((r >> 1) & 7);
        poff[i] = (unsigned)(r & (BM - 1)) * rowbytes + cl * 16;
    }
    auto src_of = [&](int bm, int bn) -> const char* {
        return wave < 2 ? reinterpret_cast<const char*>(e.A16) + (size_t)bm * BM * rowbytes : reinterpret_cast<const char*>(p.W2) + (size_t)bn * BN * rowbytes;
    };
#define FC_PDMA(SRC_, KT_, ST_)                                                                                       \
    {                                                                                                                 \
        const char* src_ = (SRC_) + (size_t)(KT_) * 128;                                                             \
        _Pragma("unroll") for (int i = 0; i < PPW; ++i)                                                              \
            __builtin_amdgcn_global_load_lds((glb_char*)(src_ + poff[i]), (lds_char*)(smc + (ST_) * STAGE + (wave * PPW + i) * 1024), 16, 0, 0); \
    }
    auto bias_dma = [&](int bn, int par) {                              // 128 floats: waves 0 and 1, 4 bytes per lane
        if (wave < 2)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) float*)(p.bias + bn * BN + wave * 64 + lane),
                                             (__attribute__((address_space(3))) float*)(biasbuf + par * 128 + wave * 64), 4, 0, 0);
    };
    auto load_x = [&](int bm, int bn, float (&x)[3], float& ldj) {
        const int row = bm * BM + wave * 32 + li, dim0 = bn * 5;
        const float* xr = e.xbuf + (size_t)row * e.ldx + e.x2_col0 + dim0;
        const bool rv = row < e.rows_valid;
        x[0] = rv && dim0 + 2 * lh < e.d2 ? xr[2 * lh] : 0.f;
        x[1] = rv && dim0 + 2 * lh + 1 < e.d2 ? xr[2 * lh + 1] : 0.f;
        x[2] = rv && dim0 + 4 < e.d2 ? xr[4] : 0.f;
        ldj = lh == 0 ? e.ldj_part[(size_t)bn * e.ldj_pitch + row] : 0.f;
    };
#define FC_PSTAMP(K_)                                                                                                 \
    if (p.stamps && threadIdx.x == 0) {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        p.stamps[(size_t)t * 16 + (K_)] = __builtin_amdgcn_s_memtime();                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }

    int bm, bn;
    tile_of(t, bm, bn);
    const char* src = src_of(bm, bn);
    if (p.stamps && threadIdx.x == 0) {
        p.stamps[(size_t)t * 16 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        p.stamps[(size_t)t * 16 + 8] = wall_clock64();
    }
    FC_PSTAMP(0)
    FC_PDMA(src, (e.prefetch_dist != 0 ? (int)(((unsigned)(bn % 10) * 3u + (unsigned)(bm & 15) * 5u) % (unsigned)p.KT) : 0), 0)
    bias_dma(bn, 0);
    float spl_x[3], spl_ldj;
    load_x(bm, bn, spl_x, spl_ldj);
    FC_PSTAMP(1)

    floatx16 acc[4], corr[4];
    const int xsw = (li >> 1) & 7;
    const int a_row = (wave * 32 + li) * ROWB, b_row = (BM + li) * ROWB;
    int st = 0, par = 0;
    // results of the previous tile, stored behind this tile's first barrier
    bool pend = false;
    int pbm = 0, pbn = 0;
    float pyA = 0.f, pyB = 0.f, pyC = 0.f, pldj = 0.f;
    auto flush = [&]() {
        const int row = pbm * BM + wave * 32 + li, dim0 = pbn * 5;
        const bool rv = row < e.rows_valid;
        float* xr = e.xbuf + (size_t)row * e.ldx + e.x2_col0 + dim0;
        if (rv && dim0 + 2 * lh < e.d2) xr[2 * lh] = pyA;
        if (rv && dim0 + 2 * lh + 1 < e.d2) xr[2 * lh + 1] = pyB;
        if (rv && lh == 0 && dim0 + 4 < e.d2) xr[4] = pyC;
        if (lh == 0) e.ldj_part[(size_t)pbn * e.ldj_pitch + row] = pldj;
    };

    // K rotation (knob 21): the ~10 workgroups of an XCD that share a 128-row panel (same row tile, the column tiles of one column group)
    // run concurrently and, started together, walk its k steps together: every step's first touch of the panel misses L2 for all of
    // them at once, and a miss holds back the hits queued behind it in the CU's in-order return path (PMC: 92 % L2 hits, yet the texture
    // data unit waits on the cache a third of the time and a DMA issued a whole k step earlier still kept its wave waiting).  A tile
    // therefore starts its k loop at step rot(column tile) and wraps around: the sharers are spread over the panel's k range, each k
    // step is missed by one of them and hit by the others.  fp32 accumulation order changes with it (not bit-identical to VAR 9 / 10).
    const bool rotate = e.prefetch_dist != 0;
    auto rot_of = [&](int bm_, int bn_) -> int { return rotate ? (int)(((unsigned)(bn_ % 10) * 3u + (unsigned)(bm_ & 15) * 5u) % (unsigned)KT) : 0; };
    for (;;) {
        const int tn = t + G;
        const bool has_next = tn < ntiles;
        int nbm = 0, nbn = 0;
        if (has_next) tile_of(tn, nbm, nbn);
        float nx[3] = {0.f, 0.f, 0.f}, nldj = 0.f;
        int kidx = rot_of(bm, bn);                                          // k step being multiplied; the DMA runs one ahead
        for (int kt = 0; kt < KT; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of this k step (and anything older) have landed
            __builtin_amdgcn_s_barrier();                               // ... everybody's have; everybody is done reading the other stage
            kidx = kidx + 1 < KT ? kidx + 1 : 0;
            if (kt + 1 < KT) {
                FC_PDMA(src, kidx, (st ^ 1))
            } else if (has_next) {                                      // the stream runs on into the next tile
                src = src_of(nbm, nbn);
                FC_PDMA(src, rot_of(nbm, nbn), (st ^ 1))
                bias_dma(nbn, par ^ 1);
                load_x(nbm, nbn, nx, nldj);
            }
            if (kt == 0) {
                FC_PSTAMP(2)
                if (pend) flush();
                // accumulators start from the bias (transposed product: it varies along the accumulator's registers)
                const float* bb = biasbuf + par * 128 + 4 * lh;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 b4 = *reinterpret_cast<const float4*>(bb + j * 32 + 8 * g);
                        acc[j][4 * g + 0] = b4.x; acc[j][4 * g + 1] = b4.y; acc[j][4 * g + 2] = b4.z; acc[j][4 * g + 3] = b4.w;
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) corr[j][r] = 0.f;
            }
            {
                const char* sA = smc + st * STAGE + a_row;
                const char* sB = smc + st * STAGE + b_row;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    f16x8 xf[2], wf[4][2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int off = ((sub * 4 + q * 2 + lh) ^ xsw) * 16;
                        xf[q] = *reinterpret_cast<const f16x8*>(sA + off);
#pragma unroll
                        for (int j = 0; j < 4; ++j) wf[j][q] = *reinterpret_cast<const f16x8*>(sB + j * 32 * ROWB + off);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][0], xf[0], acc[j], 0, 0, 0);      // hi * hi
                        corr[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][1], xf[0], corr[j], 0, 0, 0);    // lo' * hi
                        corr[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][0], xf[1], corr[j], 0, 0, 0);    // hi * lo'
                    }
                }
            }
            st ^= 1;
        }
        FC_PSTAMP(3)
        // ---- epilogue in registers (see VAR 10 in gemm_f32_kernel): slot s = 16 j + r of this lane is tile column spline_slot_col(s, lh)
        if (e.inverse != 2) {                                           // (diagnostic knob 14 = 2: main loop only)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] += corr[j][r] * (1.0f / 2048.0f);
            auto P = [&](int s) -> float { return acc[s >> 4][s & 15]; };
            float t4[11];
#pragma unroll
            for (int i = 0; i < 11; ++i) t4[i] = upper_to_lower(acc[3][2 + i]);      // slots 50..60 of the upper half: dim 4, parameters 14..24
            FC_PSTAMP(4)
            const int row = bm * BM + wave * 32 + li, dim0 = bn * 5;
            const bool rv = row < e.rows_valid;
            const bool vA = rv && dim0 + 2 * lh < e.d2, vB = rv && dim0 + 2 * lh + 1 < e.d2, vC = rv && lh == 0 && dim0 + 4 < e.d2;
            float lA, lB, lC;
            if (e.inverse == 1) {                                       // (diagnostic knob 14 = 1: no spline evaluation)
                pyA = spl_x[0] + P(0); lA = P(1); pyB = spl_x[1] + P(25); lB = P(26); pyC = spl_x[2] + P(50); lC = P(51);
            } else {
                rq_spline_fwd_regs<8>(spl_x[0], [&](int q) { return P(q); }, pyA, lA);
                rq_spline_fwd_regs<8>(spl_x[1], [&](int q) { return P(25 + q); }, pyB, lB);
                rq_spline_fwd_regs<8>(spl_x[2], [&](int q) { return q < 14 ? P(50 + q) : t4[q - 14]; }, pyC, lC);
            }
            lA = vA ? lA : 0.f; lB = vB ? lB : 0.f; lC = vC ? lC : 0.f;
            const float l2 = upper_to_lower(lA), l3 = upper_to_lower(lB);
            float sum = 0.f;
            sum += lA; sum += lB; sum += l2; sum += l3; sum += lC;      // dim order, like the LDS-tile epilogues (bit-identical slot values)
            pldj = spl_ldj + sum;
            pbm = bm; pbn = bn; pend = true;
            FC_PSTAMP(5)
        }
        if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)t * 16 + 9] = wall_clock64();
        FC_PSTAMP(6)
        if (!has_next) break;
        t = tn; bm = nbm; bn = nbn; par ^= 1;
        spl_x[0] = nx[0]; spl_x[1] = nx[1]; spl_x[2] = nx[2]; spl_ldj = nldj;
        if (p.stamps && threadIdx.x == 0) {
            p.stamps[(size_t)t * 16 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
            p.stamps[(size_t)t * 16 + 8] = wall_clock64();
        }
        FC_PSTAMP(0)
        FC_PSTAMP(1)
    }
    if (pend) flush();
#undef FC_PDMA
#undef FC_PSTAMP
}
template <>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void gemm_f32_kernel<128, 128, 4, 1, EPI_SPLINE, 11>(const GemmParams p) {
    extern __shared__ float smem[];
    spline_gemm_persistent(p, smem);
}

// tuning knobs (fc_debug_set), defaults = shipped configuration.  Every alternative below is kept because a test pins it against the
// shipped path (tests/test_gpu_flow.py::test_every_kernel_variant_agrees...) and DESIGN.md section 6 quotes its measurement.
int g_gemm_variant = 5, g_gemm_colgroup = 10, g_gemm_bigtile = 3, g_limb_chain = 1, g_lnq_fold = 1, g_fused_spline = 1;
int g_gemm_stamp = 0;        // knob 20: the LDS-DMA fused-spline launches record in-kernel phase stamps (read back with gemm_read_stamps)
static unsigned long long* g_stamp_buf = nullptr;
static size_t g_stamp_cap = 0, g_stamp_n = 0;
unsigned long long* gemm_stamp_buffer(size_t n) {
    if (n > g_stamp_cap) {
        if (g_stamp_buf) FC_HIP(hipFree(g_stamp_buf));
        FC_HIP(hipMalloc(&g_stamp_buf, n * sizeof(unsigned long long)));
        g_stamp_cap = n;
    }
    g_stamp_n = n;
    return g_stamp_buf;
}
size_t gemm_read_stamps(unsigned long long* host, size_t max_n) {
    FC_HIP(hipDeviceSynchronize());
    const size_t n = g_stamp_n < max_n ? g_stamp_n : max_n;
    if (n) FC_HIP(hipMemcpy(host, g_stamp_buf, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return n;
}
int g_spline_prefetch = 0;   // knob 21: persistent fused spline GEMM (VAR 11): 1 = a tile's k loop starts at a column-tile dependent step and wraps around (measured: no gain, other summation order); 0 = every tile starts at k = 0 (shipped, bit-identical to VAR 7-10)
int g_gemm_dma = 5;          // knob 13: fused spline GEMM: 5 = 256 x 256 one-accumulator tile on 16x16x32 MFMAs (spline_wide.hip, shipped round 4; K = 8 bins, limb-chained input; other launches fall to 4), 4 = persistent transposed LDS-DMA loop, splines evaluated from the accumulator registers (VAR 11, shipped; K = 8 bins), 3 = the same, one tile per workgroup (VAR 10), 2 = LDS-DMA loop on the 128x128 four-wave tile with the LDS parameter tile (VAR 9), 1 = on the 256x128 tile (VAR 8), 0 = register-staged (VAR 7); bit-identical results
int g_spline_ablate = 0;     // knob 14: diagnostics, results invalid (1 = no spline evaluation, 2 = main loop only, 3 = no parameter-tile write, 4 = no x2 store, 5 = stop behind the tile write)
int g_gemm_small_tiles = 1;  // knob 22: limb-chained Linear launches with at most 256 tiles of 128x128 run on 64x64 tiles
int g_gemm_dma_linear = 2;   // knob 15: limb-image A in a Linear layer: 2 = LDS-DMA loop on the 128x128 four-wave tile (shipped), 1 = on the 256x128 tile, 0 = register-staged
int g_limb_chain_all = 1;    // knob 16: every hidden activation of the coupling MLP exists only as a limb image (A16 / residual16 / C16)
int g_gemm_prefetch3 = 0;    // knob 17: three register sets of prefetch in the Linear loop (VAR 6): bit-identical, measured 3.5 % slower

static thread_local int* t_fp16_flag = nullptr;
static std::atomic<long> g_fp16_fallbacks{0};

bool gemm_fp16_enabled() { return g_gemm_variant == 5; }
bool gemm_lnq_ok() { return g_gemm_variant == 5 && t_fp16_flag != nullptr && g_gemm_bigtile == 3 && g_lnq_fold; }
bool gemm_limb_chain_all_ok() { return g_gemm_variant == 5 && t_fp16_flag != nullptr && g_gemm_bigtile == 3 && g_fused_spline && g_limb_chain && g_limb_chain_all; }
bool gemm_limb_chain_ok() { return g_gemm_variant == 5 && t_fp16_flag != nullptr && g_gemm_bigtile == 3 && g_fused_spline && g_limb_chain; }
bool gemm_split_enabled() { return (g_gemm_variant == 5 || g_gemm_variant == 3) && g_fused_spline; }
int* gemm_fp16_flag() { return g_gemm_variant == 5 ? t_fp16_flag : nullptr; }
long gemm_fp16_fallbacks() { return g_fp16_fallbacks.load(); }
Fp16Guard::Fp16Guard(int* dev_flag, hipStream_t s) : flag(dev_flag), stream(s), open(true) {
    FC_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
    t_fp16_flag = flag;
}
Fp16Guard::~Fp16Guard() { t_fp16_flag = nullptr; }
Fp16FlagScope::Fp16FlagScope(int* dev_flag) : prev(t_fp16_flag) { t_fp16_flag = dev_flag; }
Fp16FlagScope::~Fp16FlagScope() { t_fp16_flag = prev; }
bool Fp16Guard::overflowed() {
    t_fp16_flag = nullptr;
    open = false;
    int h = 0;
    FC_HIP(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    FC_HIP(hipStreamSynchronize(stream));
    if (h) g_fp16_fallbacks.fetch_add(1);
    return h != 0;
}

// ---- deferred range check (common.h).  Per thread: the switch, the queued passes and a pool of pinned flag slots / events.
namespace {
struct DeferredPass { std::function<void()> rerun; int* dev_flag; hipStream_t stream; int* host_flag; hipEvent_t ev; int device; };
struct DeferState {
    bool on = false;
    std::vector<DeferredPass> pending;
    struct Slot { int* host_flag; hipEvent_t ev; int device; };      // (an event belongs to the device that was current when it was created)
    std::vector<Slot> pool;
    ~DeferState() {
        for (auto& pe : pool) { (void)hipHostFree(pe.host_flag); (void)hipEventDestroy(pe.ev); }
        for (auto& d : pending) { (void)hipHostFree(d.host_flag); (void)hipEventDestroy(d.ev); }
    }
};
thread_local DeferState t_defer;
}  // namespace
bool guard_deferred() { return t_defer.on; }
int guard_pending() { return (int)t_defer.pending.size(); }
void guard_set_deferred(bool on) { t_defer.on = on; }
void Fp16Guard::defer(std::function<void()> rerun) {
    t_fp16_flag = nullptr;
    open = false;
    DeferredPass d{std::move(rerun), flag, stream, nullptr, nullptr, 0};
    FC_HIP(hipGetDevice(&d.device));                 // the pass may be repeated from a call made with another device current
    for (size_t i = t_defer.pool.size(); i-- > 0;)
        if (t_defer.pool[i].device == d.device) {
            d.host_flag = t_defer.pool[i].host_flag; d.ev = t_defer.pool[i].ev;
            t_defer.pool.erase(t_defer.pool.begin() + (long)i);
            break;
        }
    if (!d.host_flag) {
        FC_HIP(hipHostMalloc((void**)&d.host_flag, sizeof(int), hipHostMallocDefault));
        FC_HIP(hipEventCreateWithFlags(&d.ev, hipEventDisableTiming));
    }
    *d.host_flag = 0;
    FC_HIP(hipMemcpyAsync(d.host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    FC_HIP(hipEventRecord(d.ev, stream));
    t_defer.pending.push_back(std::move(d));
}
int guard_resolve() {
    int repeated = 0;
    std::vector<DeferredPass> todo;
    todo.swap(t_defer.pending);
    std::exception_ptr err;
    int dev_entry = 0;
    (void)hipGetDevice(&dev_entry);
    for (DeferredPass& d : todo) {
        try {
            if (err) (void)hipEventSynchronize(d.ev);        // error path: the flag copy behind this event still targets d.host_flag -- wait before the slot is pooled
            if (!err) {
                FC_HIP(hipSetDevice(d.device));              // re-launches go to the device (and pointers) the pass was queued on
                FC_HIP(hipEventSynchronize(d.ev));
                if (*d.host_flag) {                          // the fast pass left fp16's range: the whole pass again on the bf16-limb loops
                    g_fp16_fallbacks.fetch_add(1);
                    d.rerun();
                    ++repeated;
                } else if (repeated) {                       // an earlier pass was repeated and may feed this one: fast pass again, checked at once
                    bool over;
                    { Fp16Guard g(d.dev_flag, d.stream); d.rerun(); over = g.overflowed(); }
                    if (over) d.rerun();
                    ++repeated;
                }
            }
        } catch (...) { err = std::current_exception(); }
        t_defer.pool.push_back({d.host_flag, d.ev, d.device});
    }
    (void)hipSetDevice(dev_entry);
    if (err) std::rethrow_exception(err);
    return repeated;
}

template <int BM, int BN, int WM, int WN, int EPI, int VAR = 2>
static void launch_cfg(const GemmParams& p, hipStream_t s) {
    constexpr size_t lds_main = VAR == 8 ? 3 * (size_t)(BM + BN) * 128 : VAR == 11 ? 2 * (size_t)(BM + BN) * 128 + 1024 : (VAR == 9 || VAR == 10) ? (BM == 64 ? 8 : 2) * (size_t)(BM + BN) * 128 : (VAR == 5 || VAR == 6 || VAR == 7) ? 2 * (size_t)(BM + BN) * 80 : VAR >= 3 ? 2 * (size_t)(BM + BN) * 112 : 2 * (size_t)(BM + BN) * LDS_LD * sizeof(float);
    static PerDeviceOnce attr_once;
    constexpr size_t lds_epi = EPI == EPI_SPLINE && VAR != 10 && VAR != 11 ? ((size_t)BM * (BN + 1) + (size_t)BM * 9) * sizeof(float) : 0;   // tile + <= 9 dims of log-dets
    constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    auto kern = gemm_f32_kernel<BM, BN, WM, WN, EPI, VAR>;
    attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); return 0; });
    GemmParams q = p;
    q.nbn = (p.N_pad + BN - 1) / BN;
    q.col_group = 0;
    if ((size_t)p.N_pad * p.K_pad * sizeof(float) > (size_t)(3u << 19) && q.nbm % 8 == 0 && q.nbn > g_gemm_colgroup && g_gemm_colgroup > 0)
        q.col_group = g_gemm_colgroup;
    q.stamps = nullptr;
    if constexpr ((EPI == EPI_SPLINE || EPI == EPI_LINEAR) && (VAR == 8 || VAR == 9 || VAR == 10 || VAR == 11)) {
        if (g_gemm_stamp == (EPI == EPI_SPLINE ? 1 : 2)) {                  // knob 20: 1 = the fused spline launches, 2 = the limb-chained Linear launches
            const size_t n = (size_t)q.nbm * q.nbn * 16;
            if (n > g_stamp_cap) {
                if (g_stamp_buf) FC_HIP(hipFree(g_stamp_buf));
                FC_HIP(hipMalloc(&g_stamp_buf, n * sizeof(unsigned long long)));
                g_stamp_cap = n;
            }
            q.stamps = g_stamp_buf;
            g_stamp_n = n;
        }
    }
    char name[96];
    snprintf(name, sizeof name, "void fc::gemm_f32_kernel<%d, %d, %d, %d, %d, %d>(fc::GemmParams)", BM, BN, WM, WN, EPI, VAR);
    ProfScope ps(name, p.e.flops_hint, 0.0, s);
    int grid = q.nbm * q.nbn;
    if constexpr (VAR == 11) {                                          // persistent: two workgroups per CU, a multiple of 8 (XCD order)
        static PerDeviceOnce slots_once;
        const int slots = slots_once.run([](int dev) {
            int cus = 0;
            FC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            const int n = (2 * cus) & ~7;
            return n < 8 ? 8 : n;
        });
        if (grid > slots) grid = slots;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, s, q);
    FC_HIP(hipGetLastError());
}

bool gemm_dev_variants() { return kDevVariants; }
bool gemm_spline_wide_on() { return g_gemm_dma == 5 && g_gemm_variant == 5 && g_gemm_bigtile == 3 && g_fused_spline && g_limb_chain && g_spline_ablate != 3 && g_spline_ablate != 4 && g_spline_ablate != 5; }

static void launch_gemm_impl(const PackedLinear& L, const ASeg* segs, int rows_alloc, const GemmEpi& e_in, int epi_kind, hipStream_t s);
void launch_gemm(const PackedLinear& L, const ASeg* segs, int rows_alloc, const GemmEpi& e_in, int epi_kind, hipStream_t s) {
    static const bool trace = getenv("FC_FLAG_TRACE") != nullptr;      // diagnostic: which launch raises the split-fp16 range flag
    int before = 0, after = 0;
    if (trace && t_fp16_flag) { FC_HIP(hipStreamSynchronize(s)); FC_HIP(hipMemcpy(&before, t_fp16_flag, 4, hipMemcpyDeviceToHost)); }
    launch_gemm_impl(L, segs, rows_alloc, e_in, epi_kind, s);
    if (trace && t_fp16_flag) {
        FC_HIP(hipStreamSynchronize(s)); FC_HIP(hipMemcpy(&after, t_fp16_flag, 4, hipMemcpyDeviceToHost));
        if (after != before) fprintf(stderr, "[flag trace] launch_gemm epi %d rows %d N %d K %d c16 %d (scale %g) a16 %d: flag %d -> %d\n", epi_kind, rows_alloc, L.N_pad, L.K_pad, e_in.C16 != nullptr,
                                     e_in.c16_scale, e_in.A16 != nullptr, before, after);
    }
}
static void launch_gemm_impl(const PackedLinear& L, const ASeg* segs, int rows_alloc, const GemmEpi& e_in, int epi_kind, hipStream_t s) {
    if (rows_alloc % ROW_PAD != 0) throw Error(FC_ERR_INVALID, "launch_gemm: rows must be padded to ROW_PAD");
    const int v_bigtile = kDevVariants ? g_gemm_bigtile : 3;
    const int v_dma_linear = kDevVariants ? g_gemm_dma_linear : 2;
    const int v_dma = kDevVariants ? g_gemm_dma : (g_gemm_dma == 2 ? 2 : 4);      // (5, the shipped wide kernel, takes one-accumulator images only: spline_wide.hip)      // (2: the LDS-tile epilogue on the four-wave DMA tile -- in every build: it serves 4 and 16 bins)
    const int v_variant = kDevVariants ? g_gemm_variant : (g_gemm_variant < 2 ? 2 : g_gemm_variant);
    (void)v_bigtile; (void)v_dma_linear; (void)v_dma;
    if (L.K_pad % 32 != 0 || L.N_pad % 32 != 0 || L.nseg < 1 || L.nseg > 3) throw Error(FC_ERR_INVALID, "launch_gemm: bad packing");
    GemmParams p{};
    int kt = 0;
    for (int i = 0; i < 3; ++i) {
        p.A[i] = i < L.nseg ? segs[i].ptr : nullptr;
        p.lda[i] = i < L.nseg ? segs[i].lda : 0;
        p.kt[i] = i < L.nseg ? L.seg_k[i] / 32 : 0;
        if (i < L.nseg && (L.seg_k[i] % 32 != 0 || segs[i].lda % 4 != 0 || ((uintptr_t)segs[i].ptr & 15)))
            throw Error(FC_ERR_INVALID, "launch_gemm: A segment must be 16-byte aligned with a 32-multiple width");
        kt += p.kt[i];
    }
    if (kt * 32 != L.K_pad) throw Error(FC_ERR_INVALID, "launch_gemm: segment widths do not add up to K_pad");
    if (L.n_alloc < round_up(L.N_pad, gemm_bn(L.N_pad, epi_kind != EPI_LINEAR && epi_kind != EPI_SPLINE && epi_kind != EPI_LNQ)) || L.n_alloc < round_up(L.N_pad, 128))
        throw Error(FC_ERR_INVALID, "launch_gemm: W is not zero-padded to the column-tile grid (PackedLinear.n_alloc)");
    p.KT = kt;
    GemmEpi e = e_in;
    e.flops_hint = 2.0 * (double)(e.rows_valid > 0 ? e.rows_valid : rows_alloc) * (double)(L.n_true ? L.n_true : L.N_pad) *
                   (double)(L.k_true ? L.k_true : L.K_pad);
    p.W = L.W; p.W3 = L.W3; p.W2 = L.W2; p.ovf = t_fp16_flag; p.K_pad = L.K_pad; p.bias = L.bias; p.colvec = L.colvec; p.N_pad = L.N_pad;
    p.e = e;
    const bool split = (v_variant == 3 || v_variant == 5) && L.W3 != nullptr;
    const bool f16 = v_variant == 5 && L.W2 != nullptr && t_fp16_flag != nullptr;
    if (epi_kind == EPI_LINEAR) {
        if ((!e.C && !e.C16) || (e.C && e.ldc < L.N_pad)) throw Error(FC_ERR_INVALID, "launch_gemm: output pitch smaller than N_pad");
        if ((e.gradu || e.Cpre) && e.A16) throw Error(FC_ERR_INVALID, "launch_gemm: the training epilogues exist on the fp32-A loops only");
        if (e.gradu && (!e.C || e.C16 || e.Cpre || e.act != FC_ACT_NONE || e.ldgu < L.N_pad || (e.gact != FC_ACT_GELU && e.gact != FC_ACT_RELU && e.gact != FC_ACT_ELU)))
            throw Error(FC_ERR_INVALID, "launch_gemm: an activation-gradient epilogue goes with an fp32 C, no activation, and GELU / RELU / ELU");
        if (e.Cpre && (!e.C || e.C16)) throw Error(FC_ERR_INVALID, "launch_gemm: a pre-activation output goes with an fp32 C and no limb image");
        if ((e.gradu || e.Cpre) && kDevVariants && (v_variant == 0 || v_variant == 1 || (v_variant == 5 && v_bigtile == 3 && g_gemm_prefetch3)))
            throw Error(FC_ERR_UNSUPPORTED, "launch_gemm: the training epilogues (Cpre / gradu) exist on VAR 2 / 3 / 5 only; this developer variant would ignore them");
        if (e.C16 && !(f16 && v_bigtile == 3 && L.N_pad > 64 && L.N_pad % 16 == 0))
            throw Error(FC_ERR_UNSUPPORTED, "launch_gemm: limb-image output exists on the eight-wave split-fp16 tile only");
        if (e.a16_scale != 0.f) {                                       // a one-accumulator activation image: the 256 x 256 Linear kernel (spline_wide.hip EPI 1) only
            if (!(f16 && linear_wide_eligible(L, e, rows_alloc))) throw Error(FC_ERR_INVALID, "launch_gemm: a one-accumulator activation image needs the wide Linear kernel (GELU layer, images in and out, N % 256 == 0)");
            launch_linear_wide(L, e, rows_alloc, s);
            return;
        }
        if (e.r16_scale != 0.f) throw Error(FC_ERR_INVALID, "launch_gemm: a one-accumulator residual image goes with a one-accumulator A image");
        if (e.A16) {
            p.e.inverse = g_spline_ablate;
            // A arrives as the limb image of the producing layer (limb-chained MLP): the copy-only main loops
            if (!(f16 && v_bigtile == 3 && L.nseg == 1 && L.N_pad > 64 && L.n_alloc >= round_up(L.N_pad, 128)))
                throw Error(FC_ERR_UNSUPPORTED, "launch_gemm: a limb-image A operand needs the split-fp16 loop, one segment and N > 64");
            if (v_dma_linear == 2 && g_gemm_small_tiles && (rows_alloc / 128) * ((L.N_pad + 127) / 128) <= 256 && L.N_pad % 64 == 0) {
                // fewer 128x128 tiles than workgroup slots (C1: 2 x 1024 points = 16 row tiles): four times as many 64x64 tiles, each a
                // quarter of the MFMA work per k step -- the launch is bound by one workgroup's k loop, not by throughput
                p.nbm = rows_alloc / 64;
                launch_cfg<64, 64, 2, 2, EPI_LINEAR, 9>(p, s);
            }
            else if (v_dma_linear == 2) { p.nbm = rows_alloc / 128; launch_cfg<128, 128, 2, 2, EPI_LINEAR, 9>(p, s); }
            FC_DEV(else if (v_dma_linear && rows_alloc % 256 == 0) { p.nbm = rows_alloc / 256; launch_cfg<256, 128, 4, 2, EPI_LINEAR, 8>(p, s); }
                   else { p.nbm = rows_alloc / 128; launch_cfg<128, 128, 4, 2, EPI_LINEAR, 7>(p, s); })
        } else if (L.N_pad <= 64 || (f16 && g_gemm_small_tiles && !e.C16 && (rows_alloc / 128) * ((L.N_pad + 127) / 128) <= 128 && L.n_alloc >= round_up(L.N_pad, 64))) {
            // (64-wide layers; and fp32-A launches with at most 128 tiles of 128x128: twice as many 128x64 tiles)
            p.nbm = rows_alloc / 128;
            if (f16) launch_cfg<128, 64, 4, 1, EPI_LINEAR, 5>(p, s);
            else if (split) launch_cfg<128, 64, 4, 1, EPI_LINEAR, 3>(p, s);
            else launch_cfg<128, 64, 4, 1, EPI_LINEAR>(p, s);
        } else if (L.N_pad % 128 == 0 || L.N_pad > 320 || ((split || f16) && L.n_alloc >= round_up(L.N_pad, 128))) {
            // (with the split-bf16 loop two co-resident 128x128 workgroups beat the one-wave-per-SIMD 128x320 tile even at N = 320)
            // Default for the split-fp16 loop: 128x128 tile on EIGHT waves of 32x64 (64 accumulator registers per lane instead of
            // 128 -> 118 VGPRs -> 4 waves per SIMD instead of 2): +4 ... +19 % over four waves of 64x64 on every layer shape, and
            // better than the 8-wave 256x128 tile on the wide layers.  knob 3: 3 = that (default), 0 = four 64x64 waves,
            // 1 = 256x128 for N >= 1024, 2 = 256x128 everywhere
            FC_DEV(const bool big = v_bigtile == 2 || (v_bigtile == 1 && L.N_pad >= 1024);
                   if (f16 && big && rows_alloc % 256 == 0) { p.nbm = rows_alloc / 256; launch_cfg<256, 128, 4, 2, EPI_LINEAR, 5>(p, s); }
                   else if (split && v_bigtile == 2 && rows_alloc % 256 == 0) { p.nbm = rows_alloc / 256; launch_cfg<256, 128, 4, 2, EPI_LINEAR, 3>(p, s); }
                   else)
            {
                p.nbm = rows_alloc / 128;
                FC_DEV(if (f16 && v_bigtile == 3 && g_gemm_prefetch3) launch_cfg<128, 128, 4, 2, EPI_LINEAR, 6>(p, s); else)
                if (f16 && v_bigtile == 3) launch_cfg<128, 128, 4, 2, EPI_LINEAR, 5>(p, s);
                FC_DEV(else if (f16) launch_cfg<128, 128, 2, 2, EPI_LINEAR, 5>(p, s);)
                else if (split) launch_cfg<128, 128, 2, 2, EPI_LINEAR, 3>(p, s);
                FC_DEV(else if (v_variant == 0) launch_cfg<128, 128, 2, 2, EPI_LINEAR, 0>(p, s);
                       else if (v_variant == 1) launch_cfg<128, 128, 2, 2, EPI_LINEAR, 1>(p, s);)
                else launch_cfg<128, 128, 2, 2, EPI_LINEAR, 2>(p, s);
            }
        } else {
            p.nbm = rows_alloc / 128;
            if (split) launch_cfg<128, 320, 4, 1, EPI_LINEAR, 3>(p, s); else launch_cfg<128, 320, 4, 1, EPI_LINEAR>(p, s);
        }
    } else if (epi_kind == EPI_LNQ) {
        if (!(f16 && v_bigtile == 3)) throw Error(FC_ERR_UNSUPPORTED, "launch_gemm: the LayerNorm -> q fold runs on the eight-wave split-fp16 tile only");
        if (!e.C || !e.ldj_part || e.d2 % 64 != 0 || L.N_pad != e.d2 + 64 || e.ldc < 64 || e.ldj_pitch < (size_t)rows_alloc || !L.bias)
            throw Error(FC_ERR_INVALID, "launch_gemm: bad LayerNorm -> q fold arguments");
        p.nbm = rows_alloc / 128;
        if (e.A16) {
            if (L.nseg != 1 || L.n_alloc < round_up(L.N_pad, 128)) throw Error(FC_ERR_INVALID, "launch_gemm: a limb-image A operand must be the only segment");
            launch_cfg<128, 128, 2, 2, EPI_LNQ, 9>(p, s);
        } else launch_cfg<128, 128, 4, 2, EPI_LNQ, 5>(p, s);
    } else if (epi_kind == EPI_SPLINE) {
        const int K = e.spline_K;
        p.e.inverse = g_spline_ablate;
        p.e.prefetch_dist = g_spline_prefetch;
        if (!split) throw Error(FC_ERR_UNSUPPORTED, "launch_gemm: the fused spline epilogue exists for the split GEMM loops only");
        if ((K != 4 && K != 8 && K != 16) || L.N_pad != spline_ncols(e.d2, K) || !e.xbuf || !e.ldj_part || e.ldj_pitch < (size_t)rows_alloc)
            throw Error(FC_ERR_INVALID, "launch_gemm: bad fused-spline arguments (layout of spline.h, per-tile log-det buffer)");
        p.nbm = rows_alloc / 128;
        if (e.a16_scale != 0.f && !(f16 && e.A16 && v_bigtile == 3)) throw Error(FC_ERR_INVALID, "launch_gemm: a one-accumulator activation image outside the split-fp16 guard scope");
        if (f16 && e.A16 && v_bigtile == 3) {
            if (L.nseg != 1) throw Error(FC_ERR_INVALID, "launch_gemm: a limb-image A operand must be the only segment");
            if (L.n_alloc < round_up(L.N_pad, 128)) throw Error(FC_ERR_INVALID, "launch_gemm: fused spline layer not padded to the 128-column tile grid");
            if (e.a16_scale != 0.f) {                                   // the one-accumulator image: only spline_wide.hip reads it
                if (!(g_gemm_dma == 5 && spline_wide_eligible(L, K))) throw Error(FC_ERR_INVALID, "launch_gemm: a one-accumulator activation image needs the wide fused spline kernel (knob 13 = 5)");
                launch_spline_wide(L, p.e, rows_alloc, s);
            }
            else if (v_dma >= 4 && K == 8 && L.bias) launch_cfg<128, 128, 4, 1, EPI_SPLINE, 11>(p, s);
            FC_DEV(else if (v_dma == 3 && K == 8 && L.bias) launch_cfg<128, 128, 4, 1, EPI_SPLINE, 10>(p, s);)
            else if (v_dma >= 2) launch_cfg<128, 128, 2, 2, EPI_SPLINE, 9>(p, s);      // (4 and 16 bins: the LDS-tile epilogue on the four-wave DMA tile)
            FC_DEV(else if (v_dma == 1 && rows_alloc % 256 == 0) { p.nbm = rows_alloc / 256; launch_cfg<256, 128, 4, 2, EPI_SPLINE, 8>(p, s); }
                   else launch_cfg<128, 128, 4, 2, EPI_SPLINE, 7>(p, s);)
        }
        else if (f16 && v_bigtile == 3) launch_cfg<128, 128, 4, 2, EPI_SPLINE, 5>(p, s);
        FC_DEV(else if (f16) launch_cfg<128, 128, 2, 2, EPI_SPLINE, 5>(p, s);)
        else launch_cfg<128, 128, 2, 2, EPI_SPLINE, 3>(p, s);
    } else {
        if (!L.bias || L.N_pad % 64 != 0) throw Error(FC_ERR_INVALID, "launch_gemm: pair-packed epilogue needs bias and N_pad % 64 == 0");
        p.nbm = rows_alloc / 128;
        // forward direction inside a guard scope: 128x128 tile on eight waves with the split-fp16 loop (a wave's 64 columns are one
        // [first 32 | second 32] pair block); log-dets go to the caller's slot buffer.  Otherwise (inverse, bf16-limb fallback
        // pass, fp32 variants): the 128x320 tile whose workgroup owns whole rows.
        if (e.A16 && !(epi_kind == EPI_AFFINE && f16 && e.ldj_part && !e.inverse && v_bigtile == 3 && L.nseg == 1 && L.n_alloc >= round_up(L.N_pad, 128)))
            throw Error(FC_ERR_UNSUPPORTED, "launch_gemm: a limb-image A operand in a pair-packed epilogue exists for the forward affine coupling only");
        if (f16 && e.ldj_part && !e.inverse && v_bigtile == 3) {
            if (e.ldj_pitch < (size_t)rows_alloc) throw Error(FC_ERR_INVALID, "launch_gemm: log-det slot pitch smaller than the row count");
            if (e.A16 && g_gemm_small_tiles && (rows_alloc / 128) * ((L.N_pad + 127) / 128) <= 256) { p.nbm = rows_alloc / 64; launch_cfg<64, 64, 2, 1, EPI_AFFINE, 9>(p, s); }   // (small launch: 64x64 tiles, see EPI_LINEAR)
            else if (e.A16) launch_cfg<128, 128, 2, 2, EPI_AFFINE, 9>(p, s);       // limb-chained MLP: copy-only LDS-DMA loop (a wave's 64 columns = one pair block)
            else if (epi_kind == EPI_AFFINE) launch_cfg<128, 128, 4, 2, EPI_AFFINE, 5>(p, s);
            else if (epi_kind == EPI_AUGMENT) launch_cfg<128, 128, 4, 2, EPI_AUGMENT, 5>(p, s);
            else launch_cfg<128, 128, 4, 2, EPI_SLICE, 5>(p, s);
        }
        else if (epi_kind == EPI_AFFINE) { if (split) launch_cfg<128, 320, 4, 1, EPI_AFFINE, 3>(p, s); else launch_cfg<128, 320, 4, 1, EPI_AFFINE>(p, s); }
        else if (epi_kind == EPI_AUGMENT) { if (split) launch_cfg<128, 320, 4, 1, EPI_AUGMENT, 3>(p, s); else launch_cfg<128, 320, 4, 1, EPI_AUGMENT>(p, s); }
        else { if (split) launch_cfg<128, 320, 4, 1, EPI_SLICE, 3>(p, s); else launch_cfg<128, 320, 4, 1, EPI_SLICE>(p, s); }
    }
}

}  // namespace fc
