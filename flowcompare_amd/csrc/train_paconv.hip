// Training primitives for the PAConv context embedder: forward AND backward of what the PointNet++ /
// PAConv network adds to the Linear / BatchNorm / MLP primitives the DGCNN path already has.  Reference (models/scene_seg_PAConv/):
//   ScoreNet softmax over the m = 8 weight-bank kernels            model/pointnet2/paconv.py:31-54
//   assign_score: out[e, o] = sum_m score[e, m] G[e, m Cout + o]    util/paconv_util.py:52-56 (G = kernel_input @ weightbank, paconv.py:127-139)
//   kernel_input 'neighbor': E[e] = [x_e - x_centre | x_e]          paconv.py:118-123 (the centre is neighbour 0 of the group)
//   grouping backward (gradient of gathered rows)                   lib/pointops/src/grouping/grouping_cuda_kernel.cu:28-46
//   3-NN inverse-distance interpolation, forward and backward       lib/pointops/src/interpolation/interpolation_cuda_kernel.cu:90-195
// The reference's backward kernels scatter with atomicAdd; here every gather backward is an owner-computes sum over the edges sorted by
// the row they point at (the sort is index plumbing on the host side of the ABI): fixed summation order, bit-reproducible steps.
// Index kernels (FPS, sorted 32-NN, grouping of the first layer's input) are the inference ones (paconv.hip): indices carry no gradient.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "common.h"

namespace fc {
void launch_knn_xyz(const float* xyz, int ld, const float* qxyz, int32_t* out, int B, int n, int m, int k, hipStream_t s);
void launch_paconv_group(const float* xyz, int ldxyz, const float* feat, int ldf, int C, const float* qxyz, const int32_t* nidx, float* E, int ldE,
                         float* gdiff, int B, int n, int m, int K, hipStream_t s);

__device__ __forceinline__ float sqd3(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));       // nvcc -O2 contracts the kernels' sum of squares (paconv.hip sqdist3, oracle/pointops_oracle.c)
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------- softmax over the first `width` (<= 32) columns of a row; one thread per row
__global__ void softmax_fwd_kernel(const float* __restrict__ x, int ldx, int width, int rows, float* __restrict__ y, int ldy) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * ldx;
    float mx = -INFINITY;
    for (int c = 0; c < width; ++c) mx = fmaxf(mx, xr[c]);
    float sum = 0.f;
    for (int c = 0; c < width; ++c) sum += expf(xr[c] - mx);            // (same expression again below: bit-identical, no per-thread array)
    float* yr = y + (size_t)r * ldy;
    for (int c = 0; c < width; ++c) yr[c] = expf(xr[c] - mx) / sum;
    for (int c = width; c < ldy; ++c) yr[c] = 0.f;
}
__global__ void softmax_bwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ dy, int lddy, int width, int rows, int rows_pad,
                                   float* __restrict__ dx, int lddx) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows_pad) return;
    float* dr = dx + (size_t)r * lddx;
    if (r >= rows) { for (int c = 0; c < lddx; ++c) dr[c] = 0.f; return; }
    const float* yr = y + (size_t)r * ldy;
    const float* gr = dy + (size_t)r * lddy;
    float dot = 0.f;
    for (int c = 0; c < width; ++c) dot = fmaf(yr[c], gr[c], dot);
    for (int c = 0; c < width; ++c) dr[c] = yr[c] * (gr[c] - dot);
    for (int c = width; c < lddx; ++c) dr[c] = 0.f;
}

// ---------------------------------------------------------------- assign_score; one wave per edge row
__global__ __launch_bounds__(256) void assign_fwd_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ S, int lds, int m, int Cout,
                                                         int rows, int rows_pad, float* __restrict__ out, int ldo) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= rows_pad) return;
    float* o = out + (size_t)e * ldo;
    if (e >= rows) { for (int c = lane; c < ldo; c += 64) o[c] = 0.f; return; }
    const float* g = G + (size_t)e * ldg;
    const float* sc = S + (size_t)e * lds;
    for (int c = lane; c < Cout; c += 64) {
        float acc = 0.f;
        for (int mm = 0; mm < m; ++mm) acc = fmaf(sc[mm], g[(size_t)mm * Cout + c], acc);      // same order as the inference kernel (score_reduce_kernel)
        o[c] = acc;
    }
    for (int c = Cout + lane; c < ldo; c += 64) o[c] = 0.f;
}
__global__ __launch_bounds__(256) void assign_bwd_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ S, int lds,
                                                         const float* __restrict__ dout, int lddo, int m, int Cout, int rows, int rows_pad,
                                                         float* __restrict__ dG, int lddg, float* __restrict__ dS, int ldds) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= rows_pad) return;
    float* dg = dG + (size_t)e * lddg;
    float* ds = dS + (size_t)e * ldds;
    if (e >= rows) {
        for (int c = lane; c < lddg; c += 64) dg[c] = 0.f;
        for (int c = lane; c < ldds; c += 64) ds[c] = 0.f;
        return;
    }
    const float* g = G + (size_t)e * ldg;
    const float* sc = S + (size_t)e * lds;
    const float* go = dout + (size_t)e * lddo;
    for (int mm = 0; mm < m; ++mm) {
        const float s = sc[mm];
        float part = 0.f;
        for (int c = lane; c < Cout; c += 64) {
            const float d = go[c];
            dg[(size_t)mm * Cout + c] = s * d;
            part = fmaf(g[(size_t)mm * Cout + c], d, part);
        }
        part = wave_sum(part);
        if (lane == 0) ds[mm] = part;
    }
    for (int c = m * Cout + lane; c < lddg; c += 64) dg[c] = 0.f;
    for (int c = m + lane; c < ldds; c += 64) ds[c] = 0.f;
}

// ---------------------------------------------------------------- kernel_input 'neighbor'; one wave per group of K consecutive rows
__global__ __launch_bounds__(256) void centerdiff_fwd_kernel(const float* __restrict__ x, int ldx, int C, int K, int groups, float* __restrict__ E, int ldE) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= groups) return;
    const float* x0 = x + (size_t)q * K * ldx;
    for (int t = lane; t < K * ldE; t += 64) {
        const int e = t / ldE, c = t - e * ldE;
        float v = 0.f;
        if (c < C) v = x0[(size_t)e * ldx + c] - x0[c];
        else if (c < 2 * C) v = x0[(size_t)e * ldx + c - C];
        E[((size_t)q * K + e) * ldE + c] = v;
    }
}
// dx[e] = dE[e][0:C] + dE[e][C:2C] - [e is the centre] sum over the group of dE[.][0:C]
__global__ __launch_bounds__(256) void centerdiff_bwd_kernel(const float* __restrict__ dE, int ldE, int C, int K, int groups, float* __restrict__ dx, int lddx) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= groups) return;
    const float* d0 = dE + (size_t)q * K * ldE;
    float* o0 = dx + (size_t)q * K * lddx;
    for (int c = lane; c < lddx; c += 64) {
        if (c >= C) { for (int e = 0; e < K; ++e) o0[(size_t)e * lddx + c] = 0.f; continue; }
        float tot = 0.f;
        for (int e = 0; e < K; ++e) tot += d0[(size_t)e * ldE + c];
        for (int e = 0; e < K; ++e) {
            const float v = d0[(size_t)e * ldE + c] + d0[(size_t)e * ldE + C + c];
            o0[(size_t)e * lddx + c] = e == 0 ? v - tot : v;
        }
    }
}

// ---------------------------------------------------------------- gradient of gathered rows: dsrc[p] = sum over the edges pointing at p (sorted order)
__global__ __launch_bounds__(256) void rows_gather_bwd_kernel(const float* __restrict__ dout, int ldo, int col0, int C, const int32_t* __restrict__ order,
                                                              const int32_t* __restrict__ offsets, const float* __restrict__ wts, int div, int n_src,
                                                              int n_src_pad, float* __restrict__ dsrc, int lds) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= n_src_pad) return;
    float* d = dsrc + (size_t)p * lds;
    if (p >= n_src) { for (int c = lane; c < lds; c += 64) d[c] = 0.f; return; }
    const int t0 = offsets[p], t1 = offsets[p + 1];
    for (int c = lane; c < lds; c += 64) {
        float acc = 0.f;
        if (c < C)
            for (int t = t0; t < t1; ++t) {
                const int e = order[t];
                const float g = dout[(size_t)(e / div) * ldo + col0 + c];
                acc = wts ? fmaf(wts[e], g, acc) : acc + g;
            }
        d[c] = acc;
    }
}

// ---------------------------------------------------------------- 3-NN inverse-distance weights (as three_nn_interp_kernel, paconv.hip) and the weighted gather
__global__ __launch_bounds__(256) void three_nn_kernel(const float* __restrict__ uxyz, int ldu, const float* __restrict__ kxyz, int ldk, int nu, int mk,
                                                       int total, int32_t* __restrict__ idx, float* __restrict__ wout) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= total) return;
    const int b = p / nu;
    const float ux = uxyz[(size_t)p * ldu], uy = uxyz[(size_t)p * ldu + 1], uz = uxyz[(size_t)p * ldu + 2];
    const float* ks = kxyz + (size_t)b * mk * ldk;
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0x7fffffff, i1 = 0x7fffffff, i2 = 0x7fffffff;
    for (int c = lane; c < mk; c += 64) {
        const float d = sqd3(ux, uy, uz, ks[(size_t)c * ldk], ks[(size_t)c * ldk + 1], ks[(size_t)c * ldk + 2]);
        if (d < d0) { d2 = d1; i2 = i1; d1 = d0; i1 = i0; d0 = d; i0 = c; }
        else if (d < d1) { d2 = d1; i2 = i1; d1 = d; i1 = c; }
        else if (d < d2) { d2 = d; i2 = c; }
    }
    float bd[3];
    int bi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float v = d0;
        int i = i0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(v, off, 64);
            const int oi = __shfl_xor(i, off, 64);
            if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
        }
        if (i0 == i && i != 0x7fffffff) { d0 = d1; i0 = i1; d1 = d2; i1 = i2; d2 = INFINITY; i2 = 0x7fffffff; }
        bd[r] = v;
        bi[r] = i == 0x7fffffff ? 0 : i;
    }
    float w[3], ws = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) { w[r] = 1.0f / (sqrtf(bd[r]) + 1e-8f); ws += w[r]; }
    if (lane < 3) {
        idx[(size_t)p * 3 + lane] = b * mk + bi[lane];
        wout[(size_t)p * 3 + lane] = w[lane] / ws;
    }
}
__global__ __launch_bounds__(256) void interp_fwd_kernel(const float* __restrict__ Fk, int ldfk, int C, const int32_t* __restrict__ idx,
                                                         const float* __restrict__ w, int rows, int rows_pad, float* __restrict__ out, int ldo) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= rows_pad) return;
    float* o = out + (size_t)p * ldo;
    if (p >= rows) { for (int c = lane; c < ldo; c += 64) o[c] = 0.f; return; }
    const float w0 = w[(size_t)p * 3], w1 = w[(size_t)p * 3 + 1], w2 = w[(size_t)p * 3 + 2];
    const float* f0 = Fk + (size_t)idx[(size_t)p * 3] * ldfk;
    const float* f1 = Fk + (size_t)idx[(size_t)p * 3 + 1] * ldfk;
    const float* f2 = Fk + (size_t)idx[(size_t)p * 3 + 2] * ldfk;
    for (int c = lane; c < ldo; c += 64) o[c] = c < C ? (w0 * f0[c] + w1 * f1[c]) + w2 * f2[c] : 0.f;
}

}  // namespace fc

using namespace fc;

extern "C" {

/* xyz k-NN of the PAConv grouper (pointops knnquery_heap, ascending distance, local indices; n < k: the tail keeps index 0). xyz / qxyz pitch 4. */
int fc_op_paconv_knn_f32(const float* xyz, const float* qxyz, int32_t* out, int32_t B, int32_t n, int32_t m, int32_t k, void* stream) {
    FC_API_BEGIN
    if (!xyz || !qxyz || !out || B < 1 || n < 1 || m < 1 || k < 1) throw Error(FC_ERR_INVALID, "fc_op_paconv_knn_f32: bad argument");
    launch_knn_xyz(xyz, 4, qxyz, out, B, n, m, k, (hipStream_t)stream);
    FC_API_END
}
/* first PAConv layer's input of a set-abstraction level: E [B*m*K, ldE] = [x_e - x_0 | x_e | 0] with x_e = [xyz[idx_e] - qxyz (3) | feat[idx_e] (C)],
 * gdiff [B*m*K, 4] = xyz[idx_e] - xyz[idx_0] (ScoreNet input).  xyz / qxyz pitch 4, nidx local indices [B*m, K]. */
int fc_train_paconv_group_f32(const float* xyz, const float* feat, int32_t ldf, int32_t C, const float* qxyz, const int32_t* nidx, float* E, int32_t ldE,
                              float* gdiff, int32_t B, int32_t n, int32_t m, int32_t K, void* stream) {
    FC_API_BEGIN
    if (!xyz || !feat || !qxyz || !nidx || !E || !gdiff || ldE < 2 * (C + 3) || ldf < C || B < 1 || n < 1 || m < 1 || K < 1)
        throw Error(FC_ERR_INVALID, "fc_train_paconv_group_f32: bad argument");
    launch_paconv_group(xyz, 4, feat, ldf, C, qxyz, nidx, E, ldE, gdiff, B, n, m, K, (hipStream_t)stream);
    FC_API_END
}
int fc_train_softmax_fwd_f32(const float* x, int32_t ldx, int32_t width, int32_t rows, float* y, int32_t ldy, void* stream) {
    FC_API_BEGIN
    if (!x || !y || width < 1 || width > 32 || rows < 1 || ldx < width || ldy < width) throw Error(FC_ERR_INVALID, "fc_train_softmax_fwd_f32: bad argument (width <= 32)");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::softmax_fwd_kernel", 0.0, (double)rows * width * 8.0, s);
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, x, ldx, width, rows, y, ldy);
    FC_HIP(hipGetLastError());
    FC_API_END
}
int fc_train_softmax_bwd_f32(const float* y, int32_t ldy, const float* dy, int32_t lddy, int32_t width, int32_t rows, int32_t rows_pad, float* dx,
                             int32_t lddx, void* stream) {
    FC_API_BEGIN
    if (!y || !dy || !dx || width < 1 || width > 32 || rows < 1 || rows_pad < rows || ldy < width || lddy < width || lddx < width)
        throw Error(FC_ERR_INVALID, "fc_train_softmax_bwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::softmax_bwd_kernel", 0.0, (double)rows * width * 12.0, s);
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((rows_pad + 255) / 256), dim3(256), 0, s, y, ldy, dy, lddy, width, rows, rows_pad, dx, lddx);
    FC_HIP(hipGetLastError());
    FC_API_END
}
/* out [rows_pad, ldo] (pad rows / columns zeroed) = sum_m S[e, m] G[e, m Cout + o] */
int fc_train_assign_fwd_f32(const float* G, int32_t ldg, const float* S, int32_t lds, int32_t m, int32_t Cout, int32_t rows, int32_t rows_pad, float* out,
                            int32_t ldo, void* stream) {
    FC_API_BEGIN
    if (!G || !S || !out || m < 1 || Cout < 1 || rows < 1 || rows_pad < rows || ldg < m * Cout || lds < m || ldo < Cout)
        throw Error(FC_ERR_INVALID, "fc_train_assign_fwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::assign_fwd_kernel", 2.0 * rows * m * Cout, (double)rows * (m * Cout + Cout + m) * 4.0, s);
    hipLaunchKernelGGL(assign_fwd_kernel, dim3((rows_pad + 3) / 4), dim3(256), 0, s, G, ldg, S, lds, m, Cout, rows, rows_pad, out, ldo);
    FC_HIP(hipGetLastError());
    FC_API_END
}
int fc_train_assign_bwd_f32(const float* G, int32_t ldg, const float* S, int32_t lds, const float* dout, int32_t lddo, int32_t m, int32_t Cout, int32_t rows,
                            int32_t rows_pad, float* dG, int32_t lddg, float* dS, int32_t ldds, void* stream) {
    FC_API_BEGIN
    if (!G || !S || !dout || !dG || !dS || m < 1 || Cout < 1 || rows < 1 || rows_pad < rows || ldg < m * Cout || lds < m || lddo < Cout || lddg < m * Cout || ldds < m)
        throw Error(FC_ERR_INVALID, "fc_train_assign_bwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::assign_bwd_kernel", 4.0 * rows * m * Cout, (double)rows * (2.0 * m * Cout + Cout + 2.0 * m) * 4.0, s);
    hipLaunchKernelGGL(assign_bwd_kernel, dim3((rows_pad + 3) / 4), dim3(256), 0, s, G, ldg, S, lds, dout, lddo, m, Cout, rows, rows_pad, dG, lddg, dS, ldds);
    FC_HIP(hipGetLastError());
    FC_API_END
}
/* E [groups*K, ldE] = [x_e - x_centre | x_e | 0] for groups of K consecutive rows of x [groups*K, ldx] (centre = first row of the group) */
int fc_train_centerdiff_fwd_f32(const float* x, int32_t ldx, int32_t C, int32_t K, int32_t groups, float* E, int32_t ldE, void* stream) {
    FC_API_BEGIN
    if (!x || !E || C < 1 || K < 1 || groups < 1 || ldx < C || ldE < 2 * C) throw Error(FC_ERR_INVALID, "fc_train_centerdiff_fwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::centerdiff_fwd_kernel", 0.0, (double)groups * K * (C + ldE) * 4.0, s);
    hipLaunchKernelGGL(centerdiff_fwd_kernel, dim3((groups + 3) / 4), dim3(256), 0, s, x, ldx, C, K, groups, E, ldE);
    FC_HIP(hipGetLastError());
    FC_API_END
}
int fc_train_centerdiff_bwd_f32(const float* dE, int32_t ldE, int32_t C, int32_t K, int32_t groups, float* dx, int32_t lddx, void* stream) {
    FC_API_BEGIN
    if (!dE || !dx || C < 1 || K < 1 || groups < 1 || lddx < C || ldE < 2 * C) throw Error(FC_ERR_INVALID, "fc_train_centerdiff_bwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::centerdiff_bwd_kernel", 0.0, (double)groups * K * (3.0 * C + lddx) * 4.0, s);
    hipLaunchKernelGGL(centerdiff_bwd_kernel, dim3((groups + 3) / 4), dim3(256), 0, s, dE, ldE, C, K, groups, dx, lddx);
    FC_HIP(hipGetLastError());
    FC_API_END
}
/* dsrc [n_src_pad, lds] (everything beyond [n_src, C) zeroed): dsrc[p, c] = sum over t in [offsets[p], offsets[p+1]) of
 * wts[order[t]] * dout[order[t] / div, col0 + c]  (wts may be NULL = 1; div = edges per dout row: 1 for grouped rows, 3 for the 3-NN
 * interpolation).  order = edge ids sorted (stable) by the source row they gathered from, offsets = start of each source row's segment. */
int fc_train_rows_gather_bwd_f32(const float* dout, int32_t ldo, int32_t col0, int32_t C, const int32_t* order, const int32_t* offsets, const float* wts,
                                 int32_t div, int32_t n_src, int32_t n_src_pad, float* dsrc, int32_t lds, void* stream) {
    FC_API_BEGIN
    if (!dout || !order || !offsets || !dsrc || col0 < 0 || C < 1 || div < 1 || n_src < 1 || n_src_pad < n_src || ldo < col0 + C || lds < C)
        throw Error(FC_ERR_INVALID, "fc_train_rows_gather_bwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::rows_gather_bwd_kernel", 0.0, (double)n_src * C * 8.0, s);
    hipLaunchKernelGGL(rows_gather_bwd_kernel, dim3((n_src_pad + 3) / 4), dim3(256), 0, s, dout, ldo, col0, C, order, offsets, wts, div, n_src, n_src_pad, dsrc, lds);
    FC_HIP(hipGetLastError());
    FC_API_END
}
/* idx [B*nu, 3] = GLOBAL rows (b * mk + i) of the 3 nearest known points, w [B*nu, 3] = normalised inverse-distance weights. xyz pitch 4. */
int fc_train_three_nn_f32(const float* uxyz, const float* kxyz, int32_t B, int32_t nu, int32_t mk, int32_t* idx, float* w, void* stream) {
    FC_API_BEGIN
    if (!uxyz || !kxyz || !idx || !w || B < 1 || nu < 1 || mk < 1) throw Error(FC_ERR_INVALID, "fc_train_three_nn_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int total = B * nu;
    ProfScope ps("fc::three_nn_kernel", 8.0 * total * (double)mk, 24.0 * total, s);
    hipLaunchKernelGGL(three_nn_kernel, dim3((total + 3) / 4), dim3(256), 0, s, uxyz, 4, kxyz, 4, nu, mk, total, idx, w);
    FC_HIP(hipGetLastError());
    FC_API_END
}
/* out [rows_pad, ldo] (pads zeroed) = (w0 Fk[idx0] + w1 Fk[idx1]) + w2 Fk[idx2] on the first C columns */
int fc_train_interp_fwd_f32(const float* Fk, int32_t ldfk, int32_t C, const int32_t* idx, const float* w, int32_t rows, int32_t rows_pad, float* out,
                            int32_t ldo, void* stream) {
    FC_API_BEGIN
    if (!Fk || !idx || !w || !out || C < 1 || rows < 1 || rows_pad < rows || ldfk < C || ldo < C) throw Error(FC_ERR_INVALID, "fc_train_interp_fwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::interp_fwd_kernel", 0.0, (double)rows * C * 16.0, s);
    hipLaunchKernelGGL(interp_fwd_kernel, dim3((rows_pad + 3) / 4), dim3(256), 0, s, Fk, ldfk, C, idx, w, rows, rows_pad, out, ldo);
    FC_HIP(hipGetLastError());
    FC_API_END
}

}  // extern "C"
