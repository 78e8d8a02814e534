// Training primitives: the backward of the hot path, first slice = the residual GELU MLP (models/nets.py:19-30),
// which is ~61 % of the reference's forward time and two thirds of its backward FLOPs.
//
// The training path keeps parameters as ordinary dense fp32 tensors (they change every optimiser step), so nothing is folded at
// create time as in the inference engine.  Per step and per Linear the caller
//   * packs  W [N, K] -> the operand images of the split-fp16 GEMM loop (gemm.hip), for W and for W^T           (train_pack_kernel)
//   * forward   u = x W^T + b (+ residual),  y = act(u)             launch_gemm on the packed W   + act kernel
//   * backward  du = dy . act'(u)                                    act_bwd_kernel
//               dx = du W                                            launch_gemm on the packed W^T (same split-fp16 tile as the forward)
//               dW = du^T x ,  db = column sums of du                wgrad_kernel (fp32-input MFMA, rows split over workgroups,
//                                                                    partial tiles reduced in a fixed order: bit-reproducible)
// Activations are "panels": row-major fp32 [rows_pad, width_pad], rows padded to ROW_PAD, widths to 32 with ZERO pad columns.
// An input may be up to three panels side by side (the coupling MLP reads cat(x1, attention output), models/affine_coupling.py:33),
// exactly like launch_gemm's A segments.
//
// Range: the split-fp16 loop needs |operand| < 65504.  The caller owns ONE device flag per optimisation step (forward and backward
// run on different host threads under torch.autograd, so no thread-local scope can span them); every primitive ORs into it and the
// caller repeats the step with flag == NULL (fp32-input MFMA loop, any range) if it came back set.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <mutex>

#include "activations.h"
#include "common.h"

namespace fc {

struct TrainLinearLayout {
    int N, N_pad, n_alloc;                 // output features: true, padded to 32, rows of the packed W (column-tile grid)
    int nseg, seg[3], seg_pad[3];          // input panels: true and padded widths
    int K, K_pad, k_alloc;                 // sums; rows of the packed W^T
    size_t off_W, off_W2, off_bias, off_WT, off_WT2, bytes;
    // wide layers (one input panel, at least 1024 outputs, K and N multiples of 64: the spline parameter layer): one-accumulator images of
    // W and W^T for the 256 x 256 loop of spline_wide.hip (EPI 3) and the bias pre-scaled for it
    int wide, n256, k256;
    size_t off_W1, off_bias1, off_WT1;
};

static TrainLinearLayout train_layout(int N, const int32_t* seg, int nseg) {
    if (N < 1 || nseg < 1 || nseg > 3 || !seg) throw Error(FC_ERR_INVALID, "training Linear: need N >= 1 and 1..3 input segments");
    TrainLinearLayout L{};
    L.N = N; L.N_pad = round_up(N, 32); L.n_alloc = gemm_n_alloc(L.N_pad);
    L.nseg = nseg;
    for (int i = 0; i < nseg; ++i) {
        if (seg[i] < 1) throw Error(FC_ERR_INVALID, "training Linear: empty input segment");
        L.seg[i] = seg[i]; L.seg_pad[i] = round_up(seg[i], 32);
        L.K += seg[i]; L.K_pad += L.seg_pad[i];
    }
    L.k_alloc = gemm_n_alloc(L.K_pad);
    size_t o = 0;
    auto take = [&](size_t b) { const size_t r = o; o = round_up_sz(o + b, 256); return r; };
    L.off_W = take((size_t)L.n_alloc * L.K_pad * 4);
    L.off_W2 = take((size_t)L.n_alloc * L.K_pad * 4);          // [n_alloc][K_pad/16][hi 16 | lo' 16] halfs
    L.off_bias = take((size_t)L.n_alloc * 4);
    L.off_WT = take((size_t)L.k_alloc * L.N_pad * 4);
    L.off_WT2 = take((size_t)L.k_alloc * L.N_pad * 4);
    L.wide = nseg == 1 && L.N_pad >= 1024 && L.N_pad % 64 == 0 && L.K_pad % 64 == 0;
    L.n256 = round_up(L.N_pad, 256); L.k256 = round_up(L.K_pad, 256);
    if (L.wide) {
        L.off_W1 = take((size_t)L.n256 * L.K_pad * 4);             // [n256][K_pad/16][hi 16 | lo 16] halfs of w 2^kTrainWideWExp
        L.off_bias1 = take((size_t)L.n256 * 4);
        L.off_WT1 = take((size_t)L.k256 * L.N_pad * 4);            // [k256][N_pad/16][hi 16 | lo 16]
    }
    L.bytes = o;
    return L;
}

__device__ __forceinline__ int padded_to_true_k(const TrainLinearLayout& d, int kp) {
    int pad0 = 0, true0 = 0;
    for (int i = 0; i < d.nseg; ++i) {
        if (kp < pad0 + d.seg_pad[i]) {
            const int l = kp - pad0;
            return l < d.seg[i] ? true0 + l : -1;
        }
        pad0 += d.seg_pad[i]; true0 += d.seg[i];
    }
    return -1;
}

__device__ __forceinline__ void store_limbs(unsigned short* image, size_t row, int kt16, int k, float x, int* ovf) {
    if (x != 0.f && !(fabsf(x) < 65504.0f) && ovf) atomicOr(ovf, 1);
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)((x - (float)h) * 2048.0f);
    unsigned short* dst = image + (row * kt16 + (k >> 4)) * 32 + (k & 15);
    dst[0] = __builtin_bit_cast(unsigned short, h);
    dst[16] = __builtin_bit_cast(unsigned short, l);
}

// W [N, K] dense -> zero-padded fp32 + fp16 limb images of W ([n_alloc][K_pad]) and of W^T ([k_alloc][N_pad]), padded bias
__global__ void train_pack_kernel(const float* __restrict__ W, const float* __restrict__ bias, TrainLinearLayout d, char* pack, int* ovf) {
    float* Wp = (float*)(pack + d.off_W);
    unsigned short* W2 = (unsigned short*)(pack + d.off_W2);
    float* bp = (float*)(pack + d.off_bias);
    float* WT = (float*)(pack + d.off_WT);
    unsigned short* WT2 = (unsigned short*)(pack + d.off_WT2);
    const size_t nA = (size_t)d.n_alloc * d.K_pad, nB = (size_t)d.k_alloc * d.N_pad;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nA; i += stride) {
        const int n = (int)(i / d.K_pad), kp = (int)(i % d.K_pad);
        const int k = n < d.N ? padded_to_true_k(d, kp) : -1;
        const float x = k >= 0 ? W[(size_t)n * d.K + k] : 0.f;
        Wp[i] = x;
        store_limbs(W2, n, d.K_pad >> 4, kp, x, ovf);
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nB; i += stride) {
        const int kp = (int)(i / d.N_pad), n = (int)(i % d.N_pad);
        const int k = (kp < d.K_pad && n < d.N) ? padded_to_true_k(d, kp) : -1;
        const float x = k >= 0 ? W[(size_t)n * d.K + k] : 0.f;
        WT[i] = x;
        store_limbs(WT2, kp, d.N_pad >> 4, n, x, ovf);
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)d.n_alloc; i += stride)
        bp[i] = (bias && (int)i < d.N) ? bias[i] : 0.f;
    if (d.wide) {
        // one-accumulator images (spline_wide.hip): w 2^e = hi + lo with lo = rn16(w 2^e - hi) unscaled; one segment, so padded k = true k
        unsigned short* W1 = (unsigned short*)(pack + d.off_W1);
        unsigned short* WT1 = (unsigned short*)(pack + d.off_WT1);
        float* b1 = (float*)(pack + d.off_bias1);
        const float ws = (float)(1 << kTrainWideWExp);
        auto put = [&](unsigned short* image, size_t row, int kt16, int k, float x) {
            const float xs = x * ws;
            if (x != 0.f && !(fabsf(xs) < 65504.0f) && ovf) atomicOr(ovf, 1);
            const _Float16 h = (_Float16)xs;
            const _Float16 l = (_Float16)(xs - (float)h);
            unsigned short* dst = image + (row * kt16 + (k >> 4)) * 32 + (k & 15);
            dst[0] = __builtin_bit_cast(unsigned short, h);
            dst[16] = __builtin_bit_cast(unsigned short, l);
        };
        const size_t n1 = (size_t)d.n256 * d.K_pad, n2 = (size_t)d.k256 * d.N_pad;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += stride) {
            const int n = (int)(i / d.K_pad), k = (int)(i % d.K_pad);
            put(W1, n, d.K_pad >> 4, k, (n < d.N && k < d.K) ? W[(size_t)n * d.K + k] : 0.f);
        }
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
            const int k = (int)(i / d.N_pad), n = (int)(i % d.N_pad);
            put(WT1, k, d.N_pad >> 4, n, (n < d.N && k < d.K) ? W[(size_t)n * d.K + k] : 0.f);
        }
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)d.n256; i += stride)
            b1[i] = (bias && (int)i < d.N) ? bias[i] * (kOneAccActScale * ws) : 0.f;
    }
}

// ---------------------------------------------------------------- activations (models/nets.py:21-29; GELU = exact erf form)
__device__ __forceinline__ float act_grad(float u, int act) { return fc_act_grad(u, act); }      // (activations.h: shared with the GEMM epilogue)

__global__ void act_fwd_kernel(const float4* __restrict__ u, float4* __restrict__ y, size_t n4, int act) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = u[i];
        v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
        y[i] = v;
    }
}

// du = dy * act'(u) on a dense [rows_pad, ld] panel; rows >= rows_valid come out as zeros (they must not reach weight gradients)
__global__ void act_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ u, float4* __restrict__ du, size_t n4, size_t valid4,
                               int act) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < valid4) {
            const float4 a = dy[i], b = u[i];
            g.x = a.x * act_grad(b.x, act); g.y = a.y * act_grad(b.y, act); g.z = a.z * act_grad(b.z, act); g.w = a.w * act_grad(b.w, act);
        }
        du[i] = g;
    }
}

// ---------------------------------------------------------------- weight gradient  dW[n][k] = sum_p du[p][n] x[p][k]
// Both operands are row-major with the contraction index p as the slow one, which is exactly the operand order of
// v_mfma_f32_32x32x2_f32 (lane l supplies A[m = l % 32][k = l / 32] and B[k = l / 32][n = l % 32]): 32 consecutive floats of one row
// per half-wave, no transposition anywhere.  Workgroup = 4 waves, tile 128 (n) x 128 (k), each wave 64 x 64; grid.y splits the rows.
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_PS = 16;                  // rows per staged slab
constexpr int WG_LD = 128 + 4;             // LDS pitch in floats

__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ du, int ldu, int du_cols, const float* __restrict__ x, int ldx,
                                                    int x_cols, int rows_valid, int chunk_rows, int tiles_k, float* __restrict__ part,
                                                    int part_rows, int part_ld, float* __restrict__ colpart, int colpart_ld) {
    __shared__ float sA[WG_PS * WG_LD];
    __shared__ float sB[WG_PS * WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int wn = wave >> 1, wk = wave & 1;
    const int p_begin = blockIdx.y * chunk_rows, p_end = min(rows_valid, p_begin + chunk_rows);
    // staging: thread t moves 2 float4 per operand per slab: rows (t / 32) and (t / 32) + 8, columns 4 (t % 32) ...
    const int lr = tid >> 5, lc = (tid & 31) * 4;
    const bool a_ok = n0 + lc < du_cols, b_ok = k0 + lc < x_cols;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[2], rb[2];
    auto gload = [&](int p0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = p0 + lr + 8 * h;
            const bool ok = p < p_end;
            ra[h] = (ok && a_ok) ? *reinterpret_cast<const float4*>(du + (size_t)p * ldu + n0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[h] = (ok && b_ok) ? *reinterpret_cast<const float4*>(x + (size_t)p * ldx + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // bias gradient for free: the du slab is in LDS anyway, so the workgroups of the first k tile also sum its columns
    // (thread t < 128 owns column n0 + t; rows beyond the chunk were staged as zeros)
    const bool do_cols = colpart != nullptr && tk == 0 && tid < 128;
    float csum = 0.f;
    if (p_begin < p_end) gload(p_begin);
    for (int p0 = p_begin; p0 < p_end; p0 += WG_PS) {
        __syncthreads();                                  // the previous slab's reads are done
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4*>(&sA[(lr + 8 * h) * WG_LD + lc]) = ra[h];
            *reinterpret_cast<float4*>(&sB[(lr + 8 * h) * WG_LD + lc]) = rb[h];
        }
        __syncthreads();
        if (p0 + WG_PS < p_end) gload(p0 + WG_PS);
        if (do_cols) {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int r = 0; r < WG_PS; r += 2) { t0 += sA[r * WG_LD + tid]; t1 += sA[(r + 1) * WG_LD + tid]; }
            csum += t0 + t1;
        }
#pragma unroll
        for (int kk = 0; kk < WG_PS / 2; ++kk) {
            const int row = 2 * kk + (lane >> 5);
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = sA[row * WG_LD + wn * 64 + i * 32 + (lane & 31)];
                b[i] = sB[row * WG_LD + wk * 64 + i * 32 + (lane & 31)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    if (do_cols && n0 + tid < du_cols) colpart[(size_t)blockIdx.y * colpart_ld + n0 + tid] = csum;
    float* out = part + (size_t)blockIdx.y * part_rows * part_ld;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int k = k0 + wk * 64 + j * 32 + (lane & 31);
                out[(size_t)n * part_ld + k] = acc[i][j][r];
            }
}

// ---------------------------------------------------------------- the same product on the split-fp16 loop (default inside a guard scope)
// Operands go global -> registers (fp32) -> fp16 limbs (hi, lo' = (x - hi) 2048, gemm.hip) -> LDS row-major [p][hi 128 | lo' 128];
// the MFMA operands (8 consecutive-in-p halfs of one column per lane) come out of LDS through ds_read_b64_tr_b16, the transposing
// read the forward attention uses for P.V: a 16-lane group reads a [4 rows][16 columns] block and lane i receives column i.  Each
// lane's k-group therefore holds rows {4h .. 4h+3, 8+4h .. 8+4h+3} of the 16-row step -- the same permutation for both operands, so
// the contraction is unchanged.  3 MFMAs (v_mfma_f32_32x32x16_f16) per block and step: hi.hi into `main`, hi.lo' + lo'.hi into `cross`.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short v4i16 __attribute__((ext_vector_type(4)));

constexpr int W16_ROWS = 32;               // rows per staged slab (two MFMA k-steps)
constexpr int W16_PITCH = 512 + 64;        // bytes per LDS row: [hi 128 halfs | lo' 128 halfs] + pad.  144 words = 16 banks (of the 64 a ds_read_b64_tr_b16 sees) from
                                           // row to row: the 32 lanes of one read cycle -- 4 rows x 2 column halves x 4 lanes of 8 bytes -- then cover all 64 banks once
                                           // (with 512 + 32 a row advanced 8 banks, the distance of the column halves: 2-way conflicts, SQ_LDS_BANK_CONFLICT 33 % of the LDS cycles)

typedef unsigned w16_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad16_kernel(const float* __restrict__ du, int ldu, int du_cols, const float* __restrict__ x, int ldx,
                                                      int x_cols, int rows_valid, int chunk_rows, int tiles_k, float* __restrict__ part,
                                                      int part_rows, int part_ld, float* __restrict__ colpart, int colpart_ld, int* __restrict__ ovf) {
    __shared__ __attribute__((aligned(16))) char sA[2 * W16_ROWS * W16_PITCH];      // two stages each
    __shared__ __attribute__((aligned(16))) char sB[2 * W16_ROWS * W16_PITCH];
    constexpr int STAGE = W16_ROWS * W16_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lh = lane >> 5;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int wn = wave >> 1, wk = wave & 1;
    const int p_begin = blockIdx.y * chunk_rows, p_end = min(rows_valid, p_begin + chunk_rows);
    const int lr = tid >> 5, lc = (tid & 31) * 4;
    const bool a_ok = n0 + lc < du_cols, b_ok = k0 + lc < x_cols;
    const bool do_cols = colpart != nullptr && tk == 0;
    f32x16 om[2][2], oc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { om[i][j][r] = 0.f; oc[i][j][r] = 0.f; }
    float4 ra[4], rb[4];
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    float amax = 0.f;
    // Operand loads through buffer descriptors (round 3: the SQ counters showed this loop bound by VALU issue, 11 VALU + 1.5 SALU per MFMA, and
    // ~5 of them were the bounds branches and 64-bit address arithmetic of eight `ok ? load : 0` per slab).  A slab's descriptor starts at its
    // first row and ends behind row p_end - 1, so rows past the chunk read as zero through the hardware range check; a lane whose four columns
    // lie beyond the panel's width carries an offset no descriptor reaches.  Per-lane offsets are loop constants: a load is ONE instruction.
    unsigned va[4], vb[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        va[h] = a_ok ? (unsigned)(((lr + 8 * h) * ldu + n0 + lc) * 4) : 0x80000000u;
        vb[h] = b_ok ? (unsigned)(((lr + 8 * h) * ldx + k0 + lc) * 4) : 0x80000000u;
    }
    auto gload = [&](int p0) {
        const int left = p_end - p0;                      // > 0 at every call
        const __amdgpu_buffer_rsrc_t ra_d = __builtin_amdgcn_make_buffer_rsrc((void*)(du + (size_t)p0 * ldu), 0, left * ldu * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_d = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)p0 * ldx), 0, left * ldx * 4, 0x00020000);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            ra[h] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra_d, va[h], 0, 0));
            rb[h] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb_d, vb[h], 0, 0));
        }
    };
    auto split_store = [&](char* base, const float4& v) {
        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v.x), "v"(v.y));
        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v.z), "v"(v.w));
        uint2 hi, lo;                                       // five VALU per pair of values (activations.h limb_split2)
        limb_split2(v.x, v.y, hi.x, lo.x);
        limb_split2(v.z, v.w, hi.y, lo.y);
        *reinterpret_cast<uint2*>(base) = hi;
        *reinterpret_cast<uint2*>(base + 256) = lo;
    };
    // transposed-read address of this lane inside a [4 rows][16 columns] block (attention.hip): lane 4q+c of a 16-lane group supplies
    // row q, columns 4c .. 4c+3; the group's columns are 16 ((lane >> 4) & 1) .. +15 of the 32-column block, its rows start at 4 lh
    const int tr_off = (4 * lh + ((lane & 15) >> 2)) * W16_PITCH + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    // (the bias-gradient column sums exist in the workgroups of the first k tile only: two copies of the loop instead of a select per value)
    auto run = [&](auto cols_tag) {
        constexpr bool COLS = decltype(cols_tag)::value;
        auto lstore = [&](int st) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                if constexpr (COLS) { csum.x += ra[h].x; csum.y += ra[h].y; csum.z += ra[h].z; csum.w += ra[h].w; }
                split_store(sA + st * STAGE + (lr + 8 * h) * W16_PITCH + lc * 2, ra[h]);
                split_store(sB + st * STAGE + (lr + 8 * h) * W16_PITCH + lc * 2, rb[h]);
            }
        };
        // software pipeline: slab t is multiplied out of LDS stage t & 1 while slab t + 1 (already in registers) is converted and stored
        // into the other stage and slab t + 2 is fetched from HBM; one barrier per slab
        if (p_begin < p_end) { gload(p_begin); lstore(0); }
        if (p_begin + W16_ROWS < p_end) gload(p_begin + W16_ROWS);
        int st = 0;
        for (int p0 = p_begin; p0 < p_end; p0 += W16_ROWS, st ^= 1) {
            __syncthreads();                                  // stage st is complete; every wave has finished reading stage st ^ 1
#pragma unroll
            for (int ks = 0; ks < W16_ROWS / 16; ++ks) {
                f16x8 ah[2], al[2], bh[2], bl[2];
#define FC_TR(PTR_) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(PTR_))
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const char* pa = sA + st * STAGE + (16 * ks) * W16_PITCH + (wn * 64 + i * 32) * 2 + tr_off;
                    const char* pb = sB + st * STAGE + (16 * ks) * W16_PITCH + (wk * 64 + i * 32) * 2 + tr_off;
                    ah[i] = __builtin_bit_cast(f16x8, __builtin_shufflevector(FC_TR(pa), FC_TR(pa + 8 * W16_PITCH), 0, 1, 2, 3, 4, 5, 6, 7));
                    al[i] = __builtin_bit_cast(f16x8, __builtin_shufflevector(FC_TR(pa + 256), FC_TR(pa + 256 + 8 * W16_PITCH), 0, 1, 2, 3, 4, 5, 6, 7));
                    bh[i] = __builtin_bit_cast(f16x8, __builtin_shufflevector(FC_TR(pb), FC_TR(pb + 8 * W16_PITCH), 0, 1, 2, 3, 4, 5, 6, 7));
                    bl[i] = __builtin_bit_cast(f16x8, __builtin_shufflevector(FC_TR(pb + 256), FC_TR(pb + 256 + 8 * W16_PITCH), 0, 1, 2, 3, 4, 5, 6, 7));
                }
#undef FC_TR
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        om[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], om[i][j], 0, 0, 0);
                        oc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], oc[i][j], 0, 0, 0);
                        oc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], oc[i][j], 0, 0, 0);
                    }
                if (ks == 0 && p0 + W16_ROWS < p_end) lstore(st ^ 1);        // the next slab's limbs, under this slab's MFMAs
                if (ks == 0 && p0 + 2 * W16_ROWS < p_end) gload(p0 + 2 * W16_ROWS);      // ... and the slab after it requested at once: a whole slab to land
            }
        }
    };
    if (do_cols) run(std::true_type{}); else run(std::false_type{});
    if (amax >= 65504.0f || amax != amax) atomicOr(ovf, 1);
    if (do_cols) {
        // the 8 thread rows (tid >> 5) of the staging grid each summed their own rows of the chunk: add them in a fixed order
        __syncthreads();
        float* red = reinterpret_cast<float*>(sA);
        *reinterpret_cast<float4*>(red + lr * 128 + lc) = csum;
        __syncthreads();
        if (tid < 128 && n0 + tid < du_cols) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += red[r * 128 + tid];
            colpart[(size_t)blockIdx.y * colpart_ld + n0 + tid] = t;
        }
    }
    float* out = part + (size_t)blockIdx.y * part_rows * part_ld;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int k = k0 + wk * 64 + j * 32 + (lane & 31);
                out[(size_t)n * part_ld + k] = om[i][j][r] + oc[i][j][r] * (1.0f / 2048.0f);
            }
}

// dW[n][k_off + k] (=|+=) sum_s part[s][n][k]   for n < N, k < k_true: fixed summation order
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int S, int part_rows, int part_ld, float* __restrict__ dW, int N, int K,
                                    int k_off, int k_true, int accumulate, const float* __restrict__ colpart, int colpart_ld, float* __restrict__ db) {
    // entries [0, N k_true) are dW, entries [N k_true, N k_true + N) the bias gradient (column partials of the same launch);
    // the S partials are summed in chunk order with 8 independent loads in flight
    const size_t nw = (size_t)N * k_true, total = nw + (db ? N : 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float* src;
        size_t stride;
        float* dst;
        if (i < nw) {
            const int n = (int)(i / k_true), k = (int)(i % k_true);
            src = part + (size_t)n * part_ld + k; stride = (size_t)part_rows * part_ld; dst = dW + (size_t)n * K + k_off + k;
        } else {
            const int n = (int)(i - nw);
            src = colpart + n; stride = colpart_ld; dst = db + n;
        }
        float s = 0.f;
        int c = 0;
        for (; c + 8 <= S; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(c + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < S; ++c) s += src[(size_t)c * stride];
        *dst = accumulate ? *dst + s : s;
    }
}

// column sums over rows [0, rows_valid): part[s][col] then a fixed-order reduce (bias gradients).  One float4 of 4 columns per lane,
// 4 rows in flight per workgroup step; grid.y = row chunks.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, int lda, int cols, int rows_valid, int chunk_rows,
                                                     float* __restrict__ part, int part_ld) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    const int p_begin = blockIdx.y * chunk_rows, p_end = min(rows_valid, p_begin + chunk_rows);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cols) {
        if (c + 3 < cols && (lda & 3) == 0 && (((uintptr_t)a) & 15) == 0) {
#pragma unroll 8
            for (int p = p_begin + q; p < p_end; p += 4) {
                const float4 v = *reinterpret_cast<const float4*>(a + (size_t)p * lda + c);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        } else {
            for (int p = p_begin + q; p < p_end; p += 4) {
                const float* r = a + (size_t)p * lda + c;
                s.x += r[0];
                if (c + 1 < cols) s.y += r[1];
                if (c + 2 < cols) s.z += r[2];
                if (c + 3 < cols) s.w += r[3];
            }
        }
    }
    red[q][lane] = s;
    __syncthreads();
    if (q == 0 && c < cols) {
        const float4 t0 = red[0][lane], t1 = red[1][lane], t2 = red[2][lane], t3 = red[3][lane];
        float* o = part + (size_t)blockIdx.y * part_ld + c;
        o[0] = (t0.x + t1.x) + (t2.x + t3.x);
        if (c + 1 < cols) o[1] = (t0.y + t1.y) + (t2.y + t3.y);
        if (c + 2 < cols) o[2] = (t0.z + t1.z) + (t2.z + t3.z);
        if (c + 3 < cols) o[3] = (t0.w + t1.w) + (t2.w + t3.w);
    }
}
__global__ void colsum_reduce_kernel(const float* __restrict__ part, int S, int part_ld, float* __restrict__ out, int n, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    float s = 0.f;
    int i = 0;
    for (; i + 8 <= S; i += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(i + u) * part_ld + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; i < S; ++i) s += part[(size_t)i * part_ld + c];
    out[c] = accumulate ? out[c] + s : s;
}

// ---------------------------------------------------------------- host side
int g_train_wgrad16 = 1;      // tuning knob (fc_debug_set 11): weight gradients on the split-fp16 loop inside a guard scope
static int grid_for(size_t n, int block) { return (int)std::min<size_t>((n + block - 1) / block, 256 * 16); }

struct WgradPlan { int S, chunk, n128, k128max, S2, chunk2; size_t part_floats, colsum_floats, bytes; };
static int colsum_chunks(int rows) { return std::max(1, std::min(256, rows / 128)); }
static WgradPlan wgrad_plan(const TrainLinearLayout& L, int rows) {
    WgradPlan w{};
    w.n128 = round_up(L.N_pad, 128);
    int tiles_max = 1;
    for (int i = 0; i < L.nseg; ++i) {
        w.k128max = std::max(w.k128max, round_up(L.seg_pad[i], 128));
        tiles_max = std::max(tiles_max, (w.n128 / 128) * (round_up(L.seg_pad[i], 128) / 128));
    }
    // enough workgroups for 256 CUs x 2, slabs of at least 512 rows
    w.S = std::max(1, std::min(std::max(1, 512 / tiles_max), std::max(1, rows / 512)));
    w.chunk = round_up((rows + w.S - 1) / w.S, W16_ROWS);
    w.part_floats = (size_t)w.S * w.n128 * w.k128max;
    w.S2 = colsum_chunks(rows);
    w.chunk2 = (rows + w.S2 - 1) / w.S2;
    w.colsum_floats = (size_t)std::max(w.S, w.S2) * round_up(L.N_pad, 128);
    w.bytes = round_up_sz(w.part_floats * 4, 256) + round_up_sz(w.colsum_floats * 4, 256);
    return w;
}

static PackedLinear packed_forward(const TrainLinearLayout& L, const void* pack, bool f16) {
    const char* b = (const char*)pack;
    PackedLinear P;
    P.W = (float*)(b + L.off_W);
    P.W2 = f16 ? (unsigned short*)(b + L.off_W2) : nullptr;
    P.bias = (float*)(b + L.off_bias);
    P.N_pad = L.N_pad; P.K_pad = L.K_pad; P.nseg = L.nseg;
    for (int i = 0; i < L.nseg; ++i) P.seg_k[i] = L.seg_pad[i];
    P.n_true = L.N; P.k_true = L.K; P.n_alloc = L.n_alloc;
    return P;
}
static PackedLinear packed_transposed(const TrainLinearLayout& L, const void* pack, bool f16) {
    const char* b = (const char*)pack;
    PackedLinear P;
    P.W = (float*)(b + L.off_WT);
    P.W2 = f16 ? (unsigned short*)(b + L.off_WT2) : nullptr;
    P.bias = nullptr;
    P.N_pad = L.K_pad; P.K_pad = L.N_pad; P.nseg = 1; P.seg_k[0] = L.N_pad;
    P.n_true = L.K; P.k_true = L.N; P.n_alloc = L.k_alloc;
    return P;
}

// ---------------------------------------------------------------- row maxima of a gradient panel (producer -> data-gradient GEMM)
// The training spline backward holds a whole row of d(parameters) in LDS when it writes it, so it also writes max |row|; the data gradient of
// the parameter layer (the next launch that reads the panel, same stream) scales every row by an exact power of two with it before the
// limb split (spline_wide.hip EPI 3).  One entry per device; taking it consumes it (a panel that reached the GEMM by another route finds none
// and runs on the fp32-A loop).
namespace {
struct RowMaxSlot { float* buf = nullptr; int cap = 0; const float* tensor = nullptr; int rows = 0; hipStream_t s = nullptr; unsigned long long stamp = 0; };
std::mutex g_rowmax_mu;
RowMaxSlot g_rowmax[16];
// an entry is good for the next few training-Linear calls only (the parameter layer's weight and data gradient follow its spline backward
// directly): a panel that is merely allocated where an earlier gradient panel lived never meets that panel's row maxima
std::atomic<unsigned long long> g_train_calls{0};
constexpr unsigned long long kRowMaxLifetime = 4;
}
void train_call_tick() { g_train_calls.fetch_add(1); }
float* train_rowmax_reserve(const float* tensor, int rows, hipStream_t s) {
    int dev = 0;
    FC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(g_rowmax_mu);
    RowMaxSlot& r = g_rowmax[dev];
    if (r.cap < rows) {
        if (r.buf) { FC_HIP(hipStreamSynchronize(r.s)); FC_HIP(hipFree(r.buf)); r.buf = nullptr; r.cap = 0; }
        FC_HIP(hipMalloc((void**)&r.buf, (size_t)rows * 4));
        r.cap = rows;
    }
    r.tensor = tensor; r.rows = rows; r.s = s; r.stamp = g_train_calls.fetch_add(1) + 1;
    return r.buf;
}
const float* train_rowmax_take(const float* tensor, int rows, hipStream_t s) {
    int dev = 0;
    FC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(g_rowmax_mu);
    RowMaxSlot& r = g_rowmax[dev];
    const unsigned long long now = g_train_calls.load();
    if (!r.buf || r.tensor != tensor || r.rows < rows || r.s != s || now - r.stamp > kRowMaxLifetime) return nullptr;
    r.tensor = nullptr;
    return r.buf;
}

// the wide layers on the 256 x 256 one-accumulator loop (forward: constant activation scale; data gradient: per-row scales)
static bool train_wide_fwd_ok(const TrainLinearLayout& L, int rows_pad, const int32_t* ovf, const float* residual) {
    return g_train_wide && L.wide && ovf && !residual && rows_pad % 256 == 0 && gemm_fp16_enabled();
}

static void check_panel(const void* p, int ld, int width_pad, const char* what) {
    if (!p || ld < width_pad || ld % 4 != 0 || ((uintptr_t)p & 15)) throw Error(FC_ERR_INVALID, std::string("training Linear: bad panel for ") + what);
}

}  // namespace fc


using namespace fc;

extern "C" {

size_t fc_train_linear_pack_bytes(int32_t N, const int32_t* seg_widths, int32_t nseg) {
    try { return train_layout(N, seg_widths, nseg).bytes; } catch (const std::exception& e) { set_last_error(e.what()); return 0; }
}

int fc_train_linear_pack_f32(const float* W, const float* bias, int32_t N, const int32_t* seg_widths, int32_t nseg, void* pack, size_t pack_bytes,
                             int32_t* ovf, void* stream) {
    FC_API_BEGIN
    const TrainLinearLayout L = train_layout(N, seg_widths, nseg);
    if (!W || !pack || pack_bytes < L.bytes || ((uintptr_t)pack & 255)) throw Error(FC_ERR_INVALID, "fc_train_linear_pack_f32: bad argument (pack must be 256-byte aligned, fc_train_linear_pack_bytes long)");
    hipStream_t s = (hipStream_t)stream;
    size_t n = std::max((size_t)L.n_alloc * L.K_pad, (size_t)L.k_alloc * L.N_pad);
    if (L.wide) n = std::max(n, std::max((size_t)L.n256 * L.K_pad, (size_t)L.k256 * L.N_pad));
    ProfScope ps("fc::train_pack_kernel", 0.0, (double)n * 16.0, s);
    hipLaunchKernelGGL(train_pack_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, W, bias, L, (char*)pack, (int*)ovf);
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_linear_fwd_f32(const void* pack, int32_t N, const int32_t* seg_widths, int32_t nseg, const float* const* x, const int32_t* ldx,
                            int32_t rows_pad, const float* residual, int32_t ldr, float* u, int32_t ldu, int32_t* ovf, void* stream) {
    FC_API_BEGIN
    train_call_tick();
    const TrainLinearLayout L = train_layout(N, seg_widths, nseg);
    if (!pack || !x || !ldx || rows_pad < 1 || rows_pad % ROW_PAD != 0) throw Error(FC_ERR_INVALID, "fc_train_linear_fwd_f32: bad argument (rows_pad must be a multiple of 256)");
    ASeg a[3] = {};
    for (int i = 0; i < L.nseg; ++i) { check_panel(x[i], ldx[i], L.seg_pad[i], "x"); a[i] = ASeg{x[i], ldx[i]}; }
    check_panel(u, ldu, L.N_pad, "u");
    if (residual) check_panel(residual, ldr, L.N_pad, "residual");
    if (train_wide_fwd_ok(L, rows_pad, ovf, residual)) {
        TrainWideArgs w;
        w.A = x[0]; w.lda = ldx[0]; w.W1 = (const unsigned short*)((const char*)pack + L.off_W1); w.bias1 = (const float*)((const char*)pack + L.off_bias1);
        w.K_pad = L.K_pad; w.rows_pad = rows_pad; w.n_cols = L.N_pad; w.C = u; w.ldc = ldu; w.ovf = (int*)ovf;
        w.flops = 2.0 * rows_pad * (double)L.N * (double)L.K;
        launch_train_wide(w, (hipStream_t)stream);
        return FC_OK;
    }
    const PackedLinear P = packed_forward(L, pack, ovf != nullptr);
    GemmEpi e{};
    e.C = u; e.ldc = ldu; e.residual = residual; e.ldr = ldr; e.rows_valid = rows_pad;
    Fp16FlagScope scope((int*)ovf);
    launch_gemm(P, a, rows_pad, e, EPI_LINEAR, (hipStream_t)stream);
    FC_API_END
}

// the same with the activation in the GEMM's epilogue: u = cat(x) W^T + b (+ residual) AND y = act(u) from one launch (the separate
// fc_train_act_fwd_f32 pass re-read u: 6 % of a C2 training step went into the activation passes)
int fc_train_linear_act_fwd_f32(const void* pack, int32_t N, const int32_t* seg_widths, int32_t nseg, const float* const* x, const int32_t* ldx,
                                int32_t rows_pad, const float* residual, int32_t ldr, float* u, float* y, int32_t ldu, int32_t act, int32_t* ovf,
                                void* stream) {
    FC_API_BEGIN
    train_call_tick();
    const TrainLinearLayout L = train_layout(N, seg_widths, nseg);
    if (!pack || !x || !ldx || rows_pad < 1 || rows_pad % ROW_PAD != 0) throw Error(FC_ERR_INVALID, "fc_train_linear_act_fwd_f32: bad argument (rows_pad must be a multiple of 256)");
    if (act != FC_ACT_GELU && act != FC_ACT_RELU && act != FC_ACT_ELU) throw Error(FC_ERR_INVALID, "fc_train_linear_act_fwd_f32: act must be GELU, RELU or ELU");
    ASeg a[3] = {};
    for (int i = 0; i < L.nseg; ++i) { check_panel(x[i], ldx[i], L.seg_pad[i], "x"); a[i] = ASeg{x[i], ldx[i]}; }
    check_panel(u, ldu, L.N_pad, "u");
    check_panel(y, ldu, L.N_pad, "y");
    if (residual) check_panel(residual, ldr, L.N_pad, "residual");
    const PackedLinear P = packed_forward(L, pack, ovf != nullptr);
    GemmEpi e{};
    e.C = y; e.Cpre = u; e.ldc = ldu; e.act = act; e.residual = residual; e.ldr = ldr; e.rows_valid = rows_pad;
    Fp16FlagScope scope((int*)ovf);
    launch_gemm(P, a, rows_pad, e, EPI_LINEAR, (hipStream_t)stream);
    FC_API_END
}

int fc_train_linear_dgrad_f32(const void* pack, int32_t N, const int32_t* seg_widths, int32_t nseg, const float* du, int32_t ldu, int32_t rows_pad,
                              float* dx, int32_t lddx, int32_t* ovf, void* stream) {
    FC_API_BEGIN
    train_call_tick();
    const TrainLinearLayout L = train_layout(N, seg_widths, nseg);
    if (!pack || rows_pad < 1 || rows_pad % ROW_PAD != 0) throw Error(FC_ERR_INVALID, "fc_train_linear_dgrad_f32: bad argument (rows_pad must be a multiple of 256)");
    check_panel(du, ldu, L.N_pad, "du");
    check_panel(dx, lddx, L.K_pad, "dx");
    if (g_train_wide && L.wide && ovf && rows_pad % 256 == 0 && gemm_fp16_enabled()) {
        if (const float* rmax = train_rowmax_take(du, rows_pad, (hipStream_t)stream)) {
            TrainWideArgs w;
            w.A = du; w.lda = ldu; w.W1 = (const unsigned short*)((const char*)pack + L.off_WT1); w.K_pad = L.N_pad; w.rows_pad = rows_pad; w.n_cols = L.K_pad;
            w.row_absmax = rmax; w.C = dx; w.ldc = lddx; w.ovf = (int*)ovf;
            w.flops = 2.0 * rows_pad * (double)L.N * (double)L.K;
            launch_train_wide(w, (hipStream_t)stream);
            return FC_OK;
        }
    }
    const PackedLinear P = packed_transposed(L, pack, ovf != nullptr);
    ASeg a{du, ldu};
    GemmEpi e{};
    e.C = dx; e.ldc = lddx; e.rows_valid = rows_pad;
    Fp16FlagScope scope((int*)ovf);
    launch_gemm(P, &a, rows_pad, e, EPI_LINEAR, (hipStream_t)stream);
    FC_API_END
}

// dx = (du . W + addend) * act'(u_prev): the data gradient of a hidden Linear together with the backward of the activation in front of it
// (and the residual branch's gradient, `addend`): the result is the PRE-activation gradient of the previous layer, what its weight and data
// gradients consume -- no separate fc_train_act_bwd_f32 pass over the panel (6 % of a C2 training step went into the activation passes).
// One input segment; u_prev and addend are [rows_pad, lddx] panels like dx.
int fc_train_linear_dgrad_act_f32(const void* pack, int32_t N, const int32_t* seg_widths, int32_t nseg, const float* du, int32_t ldu, int32_t rows_pad,
                                  float* dx, int32_t lddx, const float* addend, const float* u_prev, int32_t act, int32_t* ovf, void* stream) {
    FC_API_BEGIN
    train_call_tick();
    const TrainLinearLayout L = train_layout(N, seg_widths, nseg);
    if (!pack || rows_pad < 1 || rows_pad % ROW_PAD != 0 || nseg != 1) throw Error(FC_ERR_INVALID, "fc_train_linear_dgrad_act_f32: bad argument (one input segment, rows_pad a multiple of 256)");
    if (act != FC_ACT_GELU && act != FC_ACT_RELU && act != FC_ACT_ELU) throw Error(FC_ERR_INVALID, "fc_train_linear_dgrad_act_f32: act must be GELU, RELU or ELU");
    check_panel(du, ldu, L.N_pad, "du");
    check_panel(dx, lddx, L.K_pad, "dx");
    check_panel(u_prev, lddx, L.K_pad, "u_prev");
    if (addend) check_panel(addend, lddx, L.K_pad, "addend");
    if (g_train_wide && L.wide && ovf && rows_pad % 256 == 0 && act == FC_ACT_GELU && gemm_fp16_enabled()) {
        if (const float* rmax = train_rowmax_take(du, rows_pad, (hipStream_t)stream)) {
            TrainWideArgs w;
            w.A = du; w.lda = ldu; w.W1 = (const unsigned short*)((const char*)pack + L.off_WT1); w.K_pad = L.N_pad; w.rows_pad = rows_pad; w.n_cols = L.K_pad;
            w.row_absmax = rmax; w.C = dx; w.ldc = lddx; w.addend = addend; w.gradu = u_prev; w.ldgu = lddx; w.gact = act; w.ovf = (int*)ovf;
            w.flops = 2.0 * rows_pad * (double)L.N * (double)L.K;
            launch_train_wide(w, (hipStream_t)stream);
            return FC_OK;
        }
    }
    const PackedLinear P = packed_transposed(L, pack, ovf != nullptr);
    ASeg a{du, ldu};
    GemmEpi e{};
    e.C = dx; e.ldc = lddx; e.rows_valid = rows_pad; e.residual = addend; e.ldr = lddx; e.gradu = u_prev; e.ldgu = lddx; e.gact = act;
    Fp16FlagScope scope((int*)ovf);
    launch_gemm(P, &a, rows_pad, e, EPI_LINEAR, (hipStream_t)stream);
    FC_API_END
}

size_t fc_train_linear_wgrad_ws_bytes(int32_t N, const int32_t* seg_widths, int32_t nseg, int32_t rows) {
    try { return wgrad_plan(train_layout(N, seg_widths, nseg), rows).bytes; } catch (const std::exception& e) { set_last_error(e.what()); return 0; }
}

int fc_train_linear_wgrad_f32(int32_t N, const int32_t* seg_widths, int32_t nseg, const float* du, int32_t ldu, const float* const* x,
                              const int32_t* ldx, int32_t rows, float* dW, float* db, int32_t accumulate, void* ws, size_t ws_bytes, int32_t* ovf,
                              void* stream) {
    FC_API_BEGIN
    train_call_tick();
    const TrainLinearLayout L = train_layout(N, seg_widths, nseg);
    if (!x || !ldx || rows < 1 || (!dW && !db)) throw Error(FC_ERR_INVALID, "fc_train_linear_wgrad_f32: bad argument");
    check_panel(du, ldu, L.N_pad, "du");
    const WgradPlan w = wgrad_plan(L, rows);
    if (!ws || ws_bytes < w.bytes || ((uintptr_t)ws & 255)) throw Error(FC_ERR_WORKSPACE, "fc_train_linear_wgrad_f32: workspace too small (fc_train_linear_wgrad_ws_bytes) or not 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)ws;
    float* cpart = (float*)((char*)ws + round_up_sz(w.part_floats * 4, 256));
    if (dW) {
        int k_off = 0;
        for (int i = 0; i < L.nseg; ++i) {
            check_panel(x[i], ldx[i], L.seg_pad[i], "x");
            const int k128 = round_up(L.seg_pad[i], 128), tiles_k = k128 / 128, tiles_n = w.n128 / 128;
            if (ovf && g_train_wgrad16) {
                ProfScope ps("fc::wgrad16_kernel", 2.0 * rows * (double)L.N * L.seg[i], 0.0, s);
                hipLaunchKernelGGL(wgrad16_kernel, dim3(tiles_n * tiles_k, w.S), dim3(256), 0, s, du, ldu, L.N_pad, x[i], ldx[i], L.seg_pad[i], rows,
                                   w.chunk, tiles_k, part, w.n128, k128, (db && i == 0) ? cpart : nullptr, w.n128, (int*)ovf);
                FC_HIP(hipGetLastError());
            } else {
                ProfScope ps("fc::wgrad_kernel", 2.0 * rows * (double)L.N * L.seg[i], 0.0, s);
                hipLaunchKernelGGL(wgrad_kernel, dim3(tiles_n * tiles_k, w.S), dim3(256), 0, s, du, ldu, L.N_pad, x[i], ldx[i], L.seg_pad[i], rows,
                                   w.chunk, tiles_k, part, w.n128, k128, (db && i == 0) ? cpart : nullptr, w.n128);
                FC_HIP(hipGetLastError());
            }
            const size_t total = (size_t)L.N * L.seg[i];
            ProfScope ps("fc::wgrad_reduce_kernel", 0.0, (double)total * 4.0 * (w.S + 1), s);
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(total + L.N, 256)), dim3(256), 0, s, part, w.S, w.n128, k128, dW, L.N, L.K, k_off,
                               L.seg[i], accumulate, cpart, w.n128, (db && i == 0) ? db : nullptr);
            FC_HIP(hipGetLastError());
            k_off += L.seg[i];
        }
    }
    if (db && !dW) {
        ProfScope ps("fc::colsum_kernel", 0.0, (double)rows * L.N * 4.0, s);
        hipLaunchKernelGGL(colsum_kernel, dim3((L.N + 255) / 256, w.S2), dim3(256), 0, s, du, ldu, L.N, rows, w.chunk2, cpart, L.N_pad);
        FC_HIP(hipGetLastError());
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3((L.N + 255) / 256), dim3(256), 0, s, cpart, w.S2, L.N_pad, db, L.N, accumulate);
        FC_HIP(hipGetLastError());
    }
    FC_API_END
}

int fc_train_act_fwd_f32(const float* u, float* y, int32_t rows_pad, int32_t ld, int32_t act, void* stream) {
    FC_API_BEGIN
    if (!u || !y || rows_pad < 1 || ld < 4 || ld % 4 != 0 || (((uintptr_t)u | (uintptr_t)y) & 15)) throw Error(FC_ERR_INVALID, "fc_train_act_fwd_f32: bad argument");
    const size_t n4 = (size_t)rows_pad * ld / 4;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::act_fwd_kernel", 0.0, (double)n4 * 32.0, s);
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, s, (const float4*)u, (float4*)y, n4, act);
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_act_bwd_f32(const float* dy, const float* u, float* du, int32_t rows_pad, int32_t rows, int32_t ld, int32_t act, void* stream) {
    FC_API_BEGIN
    if (!dy || !u || !du || rows_pad < 1 || rows < 0 || rows > rows_pad || ld < 4 || ld % 4 != 0 || (((uintptr_t)u | (uintptr_t)dy | (uintptr_t)du) & 15))
        throw Error(FC_ERR_INVALID, "fc_train_act_bwd_f32: bad argument");
    const size_t n4 = (size_t)rows_pad * ld / 4, v4 = (size_t)rows * ld / 4;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::act_bwd_kernel", 0.0, (double)n4 * 48.0, s);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, s, (const float4*)dy, (const float4*)u, (float4*)du, n4, v4, act);
    FC_HIP(hipGetLastError());
    FC_API_END
}

size_t fc_train_colsum_ws_bytes(int32_t cols, int32_t rows) {
    return (size_t)colsum_chunks(std::max(rows, 1)) * round_up(std::max(cols, 1), 32) * 4 + 256;
}

/* out[c] (=|+=) sum over rows [0, rows) of a[row][c], fixed summation order */
int fc_train_colsum_f32(const float* a, int32_t lda, int32_t cols, int32_t rows, float* out, int32_t accumulate, void* ws, size_t ws_bytes,
                        void* stream) {
    FC_API_BEGIN
    if (!a || !out || cols < 1 || rows < 1 || lda < cols) throw Error(FC_ERR_INVALID, "fc_train_colsum_f32: bad argument");
    if (!ws || ws_bytes < fc_train_colsum_ws_bytes(cols, rows)) throw Error(FC_ERR_WORKSPACE, "fc_train_colsum_f32: workspace too small (fc_train_colsum_ws_bytes)");
    const int S = colsum_chunks(rows), chunk = (rows + S - 1) / S, ld = round_up(cols, 32);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::colsum_kernel", 0.0, (double)rows * cols * 4.0, s);
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 255) / 256, S), dim3(256), 0, s, a, lda, cols, rows, chunk, (float*)ws, ld);
    FC_HIP(hipGetLastError());
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, (const float*)ws, S, ld, out, cols, accumulate);
    FC_HIP(hipGetLastError());
    FC_API_END
}

}  // extern "C"
