// Training primitives for the DGCNN context embedder (models/pytorch_gcn.py:23-47, 81-107): one EdgeConv level
//     y_ij = W cat(f_j - f_i, f_i)  ->  BatchNorm2d with BATCH statistics over (scene, point, neighbour)  ->  LeakyReLU(slope)  ->  max over the k neighbours
// (slope 0.2 for the DGCNN; slope 0 = ReLU for the PAConv embedder's BatchNorm + ReLU (+ max over the 32 neighbours), train_paconv.hip)
// forward and backward.  By linearity y_ij = P[idx_ij] + Q[i] with P = f Wa^T, Q = f (Wb - Wa)^T (W = [Wa | Wb]), so the [points, k, C]
// edge tensor is never formed by a GEMM: the two per-point products come from the training Linear (train.hip) and these kernels gather.
// With idx == NULL (k = 1, identity) and Q == NULL the same kernels are BatchNorm1d + LeakyReLU on a [points, C] matrix (conv5).
//
//   edge_stats    per-channel sum and sum of squares of y over all (i, j), fp64 accumulators, chunked + fixed-order reduce
//   edge_fwd      out[i][c] = max_j lrelu(gamma (y_ij - mean) rstd + beta), arg[i][c] = argmax j
//   edge_bwd_prep t1 = g lrelu'(u*) and t2 = t1 xhat* at the argmax: their column sums are d beta and d gamma
//   edge_bwd_scatter  dy_ij = gamma rstd ([j = j*] t1 - d beta / n - xhat_ij d gamma / n) for EVERY (i, j) (batch statistics couple all
//                 of them), dQ[i] = sum_j dy_ij, dP[idx_ij] += dy_ij by float atomics -- or, given the edges sorted by target
//                 (edge_bwd_gather), as an owner-computes sum in a fixed order: the path the training embedder takes
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "common.h"

namespace fc {

struct EdgeParams {
    const float* P; int ldp;        // [rows, C] (gathered operand)
    const float* Q; int ldq;        // [rows, C] or null
    const int* idx;                 // [rows, k] global row indices or null (identity, k == 1)
    int rows, k, C;
};

__device__ __forceinline__ float edge_y(const EdgeParams& e, int i, int j, int c) {
    const int src = e.idx ? e.idx[(size_t)i * e.k + j] : i;
    return e.P[(size_t)src * e.ldp + c] + (e.Q ? e.Q[(size_t)i * e.ldq + c] : 0.f);
}

// grid (ceil(C / 64), chunks); block 256 = 64 channels x 4 point lanes
__global__ __launch_bounds__(256) void edge_stats_kernel(EdgeParams e, int chunk_rows, double* __restrict__ part, int part_ld) {
    __shared__ double red[2][4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int i0 = blockIdx.y * chunk_rows, i1 = min(e.rows, i0 + chunk_rows);
    double s1 = 0.0, s2 = 0.0;
    if (c < e.C)
        for (int i = i0 + q; i < i1; i += 4)
            for (int j = 0; j < e.k; ++j) {
                const double y = (double)edge_y(e, i, j, c);
                s1 += y; s2 += y * y;
            }
    red[0][q][lane] = s1; red[1][q][lane] = s2;
    __syncthreads();
    if (q == 0 && c < e.C) {
        part[((size_t)blockIdx.y * 2 + 0) * part_ld + c] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        part[((size_t)blockIdx.y * 2 + 1) * part_ld + c] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
    }
}
// stats[c] = mean, stats[C + c] = rstd = 1 / sqrt(biased var + eps), stats[2C + c] = biased var
__global__ void edge_stats_reduce_kernel(const double* __restrict__ part, int S, int part_ld, float* __restrict__ stats, int C, double n, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < S; ++s) { s1 += part[((size_t)s * 2) * part_ld + c]; s2 += part[((size_t)s * 2 + 1) * part_ld + c]; }
    const double mean = s1 / n;
    const double var = fmax(s2 / n - mean * mean, 0.0);
    stats[c] = (float)mean;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    stats[2 * C + c] = (float)var;
}

__global__ __launch_bounds__(256) void edge_fwd_kernel(EdgeParams e, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ out, int ldo, unsigned char* __restrict__ arg,
                                                       float slope) {
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + q;
    if (i >= e.rows) return;
    for (int c = lane; c < e.C; c += 64) {
        const float a = gamma[c] * stats[e.C + c], b = beta[c] - a * stats[c];
        float best = -INFINITY;
        int bj = 0;
        for (int j = 0; j < e.k; ++j) {
            const float u = a * edge_y(e, i, j, c) + b;
            const float z = u > 0.f ? u : slope * u;
            if (z > best) { best = z; bj = j; }
        }
        out[(size_t)i * ldo + c] = best;
        arg[(size_t)i * e.C + c] = (unsigned char)bj;
    }
}

__global__ __launch_bounds__(256) void edge_bwd_prep_kernel(EdgeParams e, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const unsigned char* __restrict__ arg,
                                                            const float* __restrict__ g, int ldg, float* __restrict__ t1, float* __restrict__ t2, int ldt,
                                                            int rows_pad, float slope) {
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + q;
    if (i >= rows_pad) return;
    for (int c = lane; c < e.C; c += 64) {
        float a1 = 0.f, a2 = 0.f;
        if (i < e.rows) {
            const float xh = (edge_y(e, i, arg[(size_t)i * e.C + c], c) - stats[c]) * stats[e.C + c];
            const float u = gamma[c] * xh + beta[c];
            a1 = g[(size_t)i * ldg + c] * (u > 0.f ? 1.0f : slope);
            a2 = a1 * xh;
        }
        t1[(size_t)i * ldt + c] = a1;
        t2[(size_t)i * ldt + c] = a2;
    }
}

__global__ __launch_bounds__(256) void edge_bwd_scatter_kernel(EdgeParams e, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                               const unsigned char* __restrict__ arg, const float* __restrict__ t1, int ldt,
                                                               const float* __restrict__ dbeta, const float* __restrict__ dgamma, float inv_n,
                                                               float* __restrict__ dP, int lddp, float* __restrict__ dQ, int lddq) {
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + q;
    if (i >= e.rows) return;
    for (int c = lane; c < e.C; c += 64) {
        const float mean = stats[c], rstd = stats[e.C + c];
        const float a = gamma[c] * rstd, mb = dbeta[c] * inv_n, mg = dgamma[c] * inv_n;
        const float tv = t1[(size_t)i * ldt + c];
        const int js = arg[(size_t)i * e.C + c];
        float dq = 0.f;
        for (int j = 0; j < e.k; ++j) {
            const int src = e.idx ? e.idx[(size_t)i * e.k + j] : i;
            const float xh = (e.P[(size_t)src * e.ldp + c] + (e.Q ? e.Q[(size_t)i * e.ldq + c] : 0.f) - mean) * rstd;
            const float dy = a * ((j == js ? tv : 0.f) - mb - xh * mg);
            dq += dy;
            if (!dP) continue;                                   // dP comes from edge_bwd_gather_kernel (fixed summation order)
            if (e.idx) atomicAdd(dP + (size_t)src * lddp + c, dy);
            else dP[(size_t)src * lddp + c] = dy;
        }
        if (dQ) dQ[(size_t)i * lddq + c] = dq;
    }
}

// dP[m][c] = sum of dy_ij over the edges (i, j) that point at row m, visited in the order of `order` (edge ids i k + j sorted by target,
// stable): every target row is owned by one wave and summed in a fixed order -- the deterministic counterpart of the atomics above
__global__ __launch_bounds__(256) void edge_bwd_gather_kernel(EdgeParams e, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                              const unsigned char* __restrict__ arg, const float* __restrict__ t1, int ldt,
                                                              const float* __restrict__ dbeta, const float* __restrict__ dgamma, float inv_n,
                                                              const int* __restrict__ order, const int* __restrict__ offsets, float* __restrict__ dP,
                                                              int lddp) {
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + q;
    if (m >= e.rows) return;
    const int e0 = offsets[m], e1 = offsets[m + 1];
    for (int c = lane; c < e.C; c += 64) {
        const float mean = stats[c], rstd = stats[e.C + c];
        const float a = gamma[c] * rstd, mb = dbeta[c] * inv_n, mg = dgamma[c] * inv_n;
        const float pm = e.P[(size_t)m * e.ldp + c];
        float acc = 0.f;
        for (int t = e0; t < e1; ++t) {
            const int ed = order[t], i = ed / e.k, j = ed - i * e.k;
            const float xh = (pm + (e.Q ? e.Q[(size_t)i * e.ldq + c] : 0.f) - mean) * rstd;
            acc += a * ((j == arg[(size_t)i * e.C + c] ? t1[(size_t)i * ldt + c] : 0.f) - mb - xh * mg);
        }
        dP[(size_t)m * lddp + c] = acc;
    }
}

// ---------------------------------------------------------------- global embedder pooling (models/pytorch_gcn.py:178-182)
// out[b] = [max_i t[b, i, :] | mean_i t[b, i, :]] with the arg-max kept for the backward; one workgroup per (64 channels, scene)
__global__ __launch_bounds__(256) void pool_train_fwd_kernel(const float* __restrict__ t, int ldt, int width, int M, float* __restrict__ out, int ldo,
                                                             int* __restrict__ arg) {
    __shared__ float smx[4][64], ssum[4][64];
    __shared__ int sarg[4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, part = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
    float mx = -INFINITY, sum = 0.f;
    int am = 0;
    if (c < width)
        for (int i = part; i < M; i += 4) {
            const float v = t[((size_t)b * M + i) * ldt + c];
            if (v > mx) { mx = v; am = i; }
            sum += v;
        }
    smx[part][lane] = mx; ssum[part][lane] = sum; sarg[part][lane] = am;
    __syncthreads();
    if (part == 0 && c < width) {
        float best = smx[0][lane];
        int bi = sarg[0][lane];
        for (int p = 1; p < 4; ++p)
            if (smx[p][lane] > best || (smx[p][lane] == best && sarg[p][lane] < bi)) { best = smx[p][lane]; bi = sarg[p][lane]; }
        out[(size_t)b * ldo + c] = best;
        out[(size_t)b * ldo + width + c] = ((ssum[0][lane] + ssum[1][lane]) + (ssum[2][lane] + ssum[3][lane])) / (float)M;
        arg[(size_t)b * width + c] = bi;
    }
}
// dt[b, i, c] = g[b, width + c] / M + (i == arg[b, c] ? g[b, c] : 0)
__global__ void pool_train_bwd_kernel(const float* __restrict__ g, int ldg, const int* __restrict__ arg, int width, int M, int B, float* __restrict__ dt,
                                      int lddt) {
    const size_t total = (size_t)B * M * width;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % width);
        const size_t r = e / width;
        const int b = (int)(r / M), i = (int)(r % M);
        dt[r * lddt + c] = g[(size_t)b * ldg + width + c] / (float)M + (i == arg[(size_t)b * width + c] ? g[(size_t)b * ldg + c] : 0.f);
    }
}

static int edge_chunks(int rows) { return std::max(1, std::min(512, rows / 64)); }

}  // namespace fc

using namespace fc;


static EdgeParams edge_params(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, int rows, int k, int C, const char* who) {
    if (!P || rows < 1 || C < 1 || ldp < C || (Q && ldq < C) || k < 1 || k > 255 || (!idx && k != 1))
        throw Error(FC_ERR_INVALID, std::string(who) + ": bad argument (k <= 255; k == 1 without an index)");
    return EdgeParams{P, ldp, Q, ldq, idx, rows, k, C};
}

extern "C" {

size_t fc_train_edge_ws_bytes(int32_t rows, int32_t C) { return (size_t)edge_chunks(std::max(rows, 1)) * 2 * round_up(std::max(C, 1), 64) * sizeof(double) + 256; }

/* stats [3C] = per-channel mean, rstd, biased variance of y_ij = P[idx_ij] + Q[i] over all (i, j) */
int fc_train_edge_stats_f32(const float* P, int32_t ldp, const float* Q, int32_t ldq, const int32_t* idx, int32_t rows, int32_t k, int32_t C, float eps,
                            float* stats, void* ws, size_t ws_bytes, void* stream) {
    FC_API_BEGIN
    const EdgeParams e = edge_params(P, ldp, Q, ldq, idx, rows, k, C, "fc_train_edge_stats_f32");
    if (!stats || !ws || ws_bytes < fc_train_edge_ws_bytes(rows, C) || ((uintptr_t)ws & 7)) throw Error(FC_ERR_WORKSPACE, "fc_train_edge_stats_f32: workspace too small (fc_train_edge_ws_bytes)");
    hipStream_t s = (hipStream_t)stream;
    const int S = edge_chunks(rows), chunk = (rows + S - 1) / S, ld = round_up(C, 64);
    ProfScope ps("fc::edge_stats_kernel", 0.0, (double)rows * k * C * 4.0, s);
    hipLaunchKernelGGL(edge_stats_kernel, dim3((C + 63) / 64, S), dim3(256), 0, s, e, chunk, (double*)ws, ld);
    FC_HIP(hipGetLastError());
    hipLaunchKernelGGL(edge_stats_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const double*)ws, S, ld, stats, C, (double)rows * k, eps);
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_edge_fwd_f32(const float* P, int32_t ldp, const float* Q, int32_t ldq, const int32_t* idx, int32_t rows, int32_t k, int32_t C,
                          const float* stats, const float* gamma, const float* beta, float slope, float* out, int32_t ldo, uint8_t* arg, void* stream) {
    FC_API_BEGIN
    const EdgeParams e = edge_params(P, ldp, Q, ldq, idx, rows, k, C, "fc_train_edge_fwd_f32");
    if (!stats || !gamma || !beta || !out || !arg || ldo < C) throw Error(FC_ERR_INVALID, "fc_train_edge_fwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::edge_fwd_kernel", 0.0, (double)rows * k * C * 4.0, s);
    hipLaunchKernelGGL(edge_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, e, stats, gamma, beta, out, ldo, arg, slope);
    FC_HIP(hipGetLastError());
    FC_API_END
}

/* t1, t2 [rows_pad, ldt]: g lrelu'(u*) and that times xhat*, zeros beyond `rows` (column sums = d beta, d gamma) */
int fc_train_edge_bwd_prep_f32(const float* P, int32_t ldp, const float* Q, int32_t ldq, const int32_t* idx, int32_t rows, int32_t k, int32_t C,
                               const float* stats, const float* gamma, const float* beta, float slope, const uint8_t* arg, const float* g, int32_t ldg,
                               float* t1, float* t2, int32_t ldt, int32_t rows_pad, void* stream) {
    FC_API_BEGIN
    const EdgeParams e = edge_params(P, ldp, Q, ldq, idx, rows, k, C, "fc_train_edge_bwd_prep_f32");
    if (!stats || !gamma || !beta || !arg || !g || !t1 || !t2 || ldg < C || ldt < C || rows_pad < rows) throw Error(FC_ERR_INVALID, "fc_train_edge_bwd_prep_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::edge_bwd_prep_kernel", 0.0, (double)rows * C * 16.0, s);
    hipLaunchKernelGGL(edge_bwd_prep_kernel, dim3((rows_pad + 3) / 4), dim3(256), 0, s, e, stats, gamma, beta, arg, g, ldg, t1, t2, ldt, rows_pad, slope);
    FC_HIP(hipGetLastError());
    FC_API_END
}

/* dP (ZEROED by the caller when idx != NULL: float atomics accumulate into it) and dQ (may be NULL) */
int fc_train_edge_bwd_scatter_f32(const float* P, int32_t ldp, const float* Q, int32_t ldq, const int32_t* idx, int32_t rows, int32_t k, int32_t C,
                                  const float* stats, const float* gamma, const uint8_t* arg, const float* t1, int32_t ldt, const float* dbeta,
                                  const float* dgamma, float* dP, int32_t lddp, float* dQ, int32_t lddq, void* stream) {
    FC_API_BEGIN
    const EdgeParams e = edge_params(P, ldp, Q, ldq, idx, rows, k, C, "fc_train_edge_bwd_scatter_f32");
    if (!stats || !gamma || !arg || !t1 || !dbeta || !dgamma || (!dP && !dQ) || ldt < C || (dP && lddp < C) || (dQ && lddq < C))
        throw Error(FC_ERR_INVALID, "fc_train_edge_bwd_scatter_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::edge_bwd_scatter_kernel", 0.0, (double)rows * k * C * 8.0, s);
    hipLaunchKernelGGL(edge_bwd_scatter_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, e, stats, gamma, arg, t1, ldt, dbeta, dgamma,
                       (float)(1.0 / ((double)rows * k)), dP, lddp, dQ, lddq);
    FC_HIP(hipGetLastError());
    FC_API_END
}

/* Global-embedder pooling over a scene's M points: out [B, ldo >= 2 width] = [max | mean], arg [B, width] (int32) for the backward. */
int fc_train_pool_fwd_f32(const float* t, int32_t ldt, int32_t width, int32_t B, int32_t M, float* out, int32_t ldo, int32_t* arg, void* stream) {
    FC_API_BEGIN
    if (!t || !out || !arg || width < 1 || B < 1 || M < 1 || ldt < width || ldo < 2 * width) throw Error(FC_ERR_INVALID, "fc_train_pool_fwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::pool_train_fwd_kernel", 0.0, (double)B * M * width * 4.0, s);
    hipLaunchKernelGGL(pool_train_fwd_kernel, dim3((width + 63) / 64, B), dim3(256), 0, s, t, ldt, width, M, out, ldo, (int*)arg);
    FC_HIP(hipGetLastError());
    FC_API_END
}
int fc_train_pool_bwd_f32(const float* g, int32_t ldg, const int32_t* arg, int32_t width, int32_t B, int32_t M, float* dt, int32_t lddt, void* stream) {
    FC_API_BEGIN
    if (!g || !arg || !dt || width < 1 || B < 1 || M < 1 || ldg < 2 * width || lddt < width) throw Error(FC_ERR_INVALID, "fc_train_pool_bwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)B * M * width;
    ProfScope ps("fc::pool_train_bwd_kernel", 0.0, (double)total * 4.0, s);
    hipLaunchKernelGGL(pool_train_bwd_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, g, ldg, (const int*)arg, width, M, B, dt, lddt);
    FC_HIP(hipGetLastError());
    FC_API_END
}

/* Deterministic dP: order [rows * k] = edge ids (i * k + j) sorted by the row they point at (stable), offsets [rows + 1] = start of each
 * row's segment.  Use together with fc_train_edge_bwd_scatter_f32(dP = NULL) for dQ. */
int fc_train_edge_bwd_gather_f32(const float* P, int32_t ldp, const float* Q, int32_t ldq, const int32_t* idx, int32_t rows, int32_t k, int32_t C,
                                 const float* stats, const float* gamma, const uint8_t* arg, const float* t1, int32_t ldt, const float* dbeta,
                                 const float* dgamma, const int32_t* order, const int32_t* offsets, float* dP, int32_t lddp, void* stream) {
    FC_API_BEGIN
    const EdgeParams e = edge_params(P, ldp, Q, ldq, idx, rows, k, C, "fc_train_edge_bwd_gather_f32");
    if (!idx || !stats || !gamma || !arg || !t1 || !dbeta || !dgamma || !order || !offsets || !dP || ldt < C || lddp < C)
        throw Error(FC_ERR_INVALID, "fc_train_edge_bwd_gather_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::edge_bwd_gather_kernel", 0.0, (double)rows * k * C * 8.0, s);
    hipLaunchKernelGGL(edge_bwd_gather_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, e, stats, gamma, arg, t1, ldt, dbeta, dgamma,
                       (float)(1.0 / ((double)rows * k)), order, offsets, dP, lddp);
    FC_HIP(hipGetLastError());
    FC_API_END
}

}  // extern "C"
