// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32) with fused epilogues.
//
//   C[rows, N_pad] = epilogue( sum_seg A_seg[rows, k_seg] @ W[N_pad, K_pad]^T )
//
// Every Linear of the flow (coupling MLPs, pre-attention MLPs, q/kv projections, the folded
// ActNorm+LinearLU matrix, DGCNN 1x1 convs) goes through this kernel.  fp32-in / fp32-accumulate MFMA is
// bit-for-bit an fmaf chain, which is what the 1e-4 nats parity gate over 115 chained layers needs
// (DESIGN.md §numerics); its peak is 157.3 TFLOP/s, the roofline this kernel is measured against.
//
// Layout: both operands are K-contiguous in memory (activations [rows][K], weights [N][K] exactly like
// torch.nn.Linear.weight), so one ds_read_b128 per lane feeds FOUR MFMA k-steps of an operand tile:
// lane (i = lane&31, h = lane>>5) holds element e of its float4 as the k = 8g + 4h + e operand.
// LDS rows are 32 floats + 4 pad (144 B): conflict-free for ds_read_b128 (16 lanes -> 16 distinct 16-B slots).
// Global -> LDS goes through registers (prefetch of tile t+1 is issued before the MFMAs of tile t, written
// after them): one barrier per 32-deep K tile.
#include "common.h"
#include <cstdio>

namespace fc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct GemmParams {
    const float* A[3];
    int lda[3];
    int kt[3];          // 32-wide k tiles per segment
    int KT;             // total k tiles
    const float* W;     // [N_pad][K_pad]
    int K_pad;
    const float* bias;
    const float* colvec;
    int N_pad;
    int nbm, nbn;
    GemmEpi e;
};

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case FC_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case FC_ACT_RELU: return v > 0.f ? v : 0.f;
        case FC_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case FC_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
        default: return v;
    }
}

__device__ __forceinline__ float half_wave_sum(float v) {
    // sum over the 32 lanes that share (lane>>5): xor masks < 32 never cross the half
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    return v;
}

constexpr int LDS_LD = 36;   // floats per LDS row (32 + 4 pad)

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_F4 = BM * 8 / 256, B_F4 = BN * 8 / 256;
    constexpr int STAGE = (BM + BN) * LDS_LD;
    extern __shared__ float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware tile order: blocks b, b+8, ... share an XCD (L2); give each XCD a contiguous run of
    // (row-tile, col-tile) pairs with the col-tile fastest so the A row panel is fetched from HBM once.
    int bm, bn;
    {
        const int nb = p.nbm * p.nbn, b = blockIdx.x;
        const int xcd = b & 7, q = nb >> 3, r = nb & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        bm = L / p.nbn;
        bn = L - bm * p.nbn;
    }
    const int m0 = bm * BM, n0 = bn * BN;
    const int wave_n0 = n0 + wc * TN * 32;
    int nvalid = (p.N_pad - wave_n0) / 32;                 // wave-uniform number of live 32-col tiles
    nvalid = nvalid < 0 ? 0 : (nvalid > TN ? TN : nvalid);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- global -> register staging: thread t loads float4 (t&7) of rows (t>>3) + 32*i
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
    float4 ra[A_F4], rb[B_F4];
    auto gload = [&](int kt) {
        const float* Ap = p.A[0];
        int lda = p.lda[0], kk = kt;
        if (kk >= p.kt[0]) {
            kk -= p.kt[0];
            Ap = p.A[1]; lda = p.lda[1];
            if (kk >= p.kt[1]) { kk -= p.kt[1]; Ap = p.A[2]; lda = p.lda[2]; }
        }
        const float* a = Ap + (size_t)(m0 + lrow) * lda + kk * 32 + lc4;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) ra[i] = *reinterpret_cast<const float4*>(a + (size_t)(32 * i) * lda);
        const float* w = p.W + (size_t)(n0 + lrow) * p.K_pad + kt * 32 + lc4;
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            if (n0 + lrow + 32 * i < p.N_pad) rb[i] = *reinterpret_cast<const float4*>(w + (size_t)(32 * i) * p.K_pad);
            else rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int stage) {
        float* sA = smem + stage * STAGE;
        float* sB = sA + BM * LDS_LD;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) *reinterpret_cast<float4*>(sA + (lrow + 32 * i) * LDS_LD + lc4) = ra[i];
#pragma unroll
        for (int i = 0; i < B_F4; ++i) *reinterpret_cast<float4*>(sB + (lrow + 32 * i) * LDS_LD + lc4) = rb[i];
    };

    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < p.KT; ++kt) {
        const bool more = kt + 1 < p.KT;
        if (more) gload(kt + 1);
        const float* sA = smem + (kt & 1) * STAGE + (wr * TM * 32 + li) * LDS_LD + 4 * lh;
        const float* sB = smem + (kt & 1) * STAGE + BM * LDS_LD + (wc * TN * 32 + li) * LDS_LD + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 a[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(sA + i * 32 * LDS_LD + 8 * g);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (j < nvalid) {
                    const float4 b = *reinterpret_cast<const float4*>(sB + j * 32 * LDS_LD + 8 * g);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b.x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b.y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b.z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b.w, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        if (more) lstore((kt + 1) & 1);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogues
    // C/D layout of the 32x32 MFMA: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5), r = 0..15
    const GemmEpi& e = p.e;
    const int wave_m0 = m0 + wr * TM * 32;
    if constexpr (EPI == EPI_LINEAR) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (j < nvalid) {
                const int col = wave_n0 + j * 32 + li;
                const float bv = p.bias ? p.bias[col] : 0.f;
                const float cv = (p.colvec && e.rowscal) ? p.colvec[col] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wave_m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        float v = acc[i][j][r] + bv;
                        if (e.rowscal) v += e.rowscal[row] * cv;
                        if (e.residual) v += e.residual[(size_t)row * e.ldr + col];
                        e.C[(size_t)row * e.ldc + col] = act_apply(v, e.act);
                    }
                }
            }
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
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = wave_m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            const float u = acc[i][2 * pr][r] + bs, t = acc[i][2 * pr + 1][r] + bt;
                            if constexpr (EPI == EPI_AFFINE) {
                                // models/affine_coupling.py:23-46: s = exp(u) | (2 sigmoid(u) - 1)(1 - 1e-8) + 1
                                float s;
                                if (e.scale_fn == FC_SCALE_EXP) s = expf(u);
                                else s = (2.0f * (1.0f / (1.0f + expf(-u))) - 1.0f) * (float)(1.0 - 1e-8) + 1.0f;
                                float* xp = e.xbuf + (size_t)row * e.ldx + e.x2_col0 + j;
                                *xp = *xp * s + t;
                                lsum[i][r] += logf(s);
                            } else {
                                // models/augmenter.py:49-63 + distributions.py:128-153: z2 = mu + eps*sigma, ldj = -log N(z2; mu, sigma)
                                float sigma = expf(t);
                                if (e.clamp > 0.f) sigma = fminf(sigma, e.clamp);
                                const float ev = row < e.rows_valid ? e.eps[(size_t)row * e.d2 + j] : 0.f;
                                const float z = u + ev * sigma;
                                const float dz = z - u;
                                const float lp = -(dz * dz) / (2.0f * sigma * sigma) - logf(sigma) - 0.91893853320467274178f;
                                const int idx = e.d_in + j;
                                const int col = idx < e.d1 ? idx : e.d1_pad + (idx - e.d1);
                                e.xbuf[(size_t)row * e.ldx + col] = z;
                                lsum[i][r] -= lp;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float tot = half_wave_sum(lsum[i][r]);
                const int row = wave_m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (li == 0 && row < e.rows_valid) {
                    if (p.nbn * WN > 1) atomicAdd(e.logprob + row, tot);
                    else e.logprob[row] += tot;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI>
static void launch_cfg(const GemmParams& p, hipStream_t s) {
    constexpr size_t lds = 2 * (size_t)(BM + BN) * LDS_LD * sizeof(float);
    static bool attr_done = false;
    auto kern = gemm_f32_kernel<BM, BN, WM, WN, EPI>;
    if (!attr_done) {
        FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    GemmParams q = p;
    q.nbn = (p.N_pad + BN - 1) / BN;
    char name[96];
    snprintf(name, sizeof name, "void fc::gemm_f32_kernel<%d, %d, %d, %d, %d>(fc::GemmParams)", BM, BN, WM, WN, EPI);
    ProfScope ps(name, p.e.flops_hint, 0.0, s);
    hipLaunchKernelGGL(kern, dim3(q.nbm * q.nbn), dim3(256), lds, s, q);
    FC_HIP(hipGetLastError());
}

void launch_gemm(const PackedLinear& L, const ASeg* segs, int rows_alloc, const GemmEpi& e_in, int epi_kind, hipStream_t s) {
    if (rows_alloc % ROW_PAD != 0) throw Error(FC_ERR_INVALID, "launch_gemm: rows must be padded to ROW_PAD");
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
    p.KT = kt;
    GemmEpi e = e_in;
    e.flops_hint = 2.0 * (double)(e.rows_valid > 0 ? e.rows_valid : rows_alloc) * (double)(L.n_true ? L.n_true : L.N_pad) *
                   (double)(L.k_true ? L.k_true : L.K_pad);
    p.W = L.W; p.K_pad = L.K_pad; p.bias = L.bias; p.colvec = L.colvec; p.N_pad = L.N_pad;
    p.e = e;
    if (epi_kind == EPI_LINEAR) {
        if (!e.C || e.ldc < L.N_pad) throw Error(FC_ERR_INVALID, "launch_gemm: output pitch smaller than N_pad");
        if (L.N_pad <= 64) { p.nbm = rows_alloc / 128; launch_cfg<128, 64, 4, 1, EPI_LINEAR>(p, s); }
        else if (L.N_pad % 128 == 0 || L.N_pad > 320) { p.nbm = rows_alloc / 128; launch_cfg<128, 128, 2, 2, EPI_LINEAR>(p, s); }
        else { p.nbm = rows_alloc / 128; launch_cfg<128, 320, 4, 1, EPI_LINEAR>(p, s); }
    } else {
        if (!L.bias || L.N_pad % 64 != 0) throw Error(FC_ERR_INVALID, "launch_gemm: pair-packed epilogue needs bias and N_pad % 64 == 0");
        p.nbm = rows_alloc / 128;
        if (epi_kind == EPI_AFFINE) launch_cfg<128, 320, 4, 1, EPI_AFFINE>(p, s);
        else launch_cfg<128, 320, 4, 1, EPI_AUGMENT>(p, s);
    }
}

}  // namespace fc
