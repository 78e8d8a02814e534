// HBM-bound helper kernels of the flow engine: layout packing, LayerNorm, base density, the
// rational-quadratic spline, DGCNN gather-max and global pooling.  All are one-wave-per-row streaming
// kernels (coalesced row reads, wave reductions via DPP shuffles); none of them is reshaped into a GEMM.
#include "common.h"
#include "spline.h"

namespace fc {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// ---------------------------------------------------------------- limb images of a packed weight matrix (fc_*_create)
// One thread per (row, k16 block, element): W3 [rows][K/16][3][16] bf16 limbs (hi, mid, lo: x = hi + mid + lo to 24 bits), W2
// [rows][K/16][2][16] fp16 limbs (hi, lo' = (x - hi) * 2048) -- the roundings of hostpack.cpp make_bf16_limbs / make_f16_limbs.
__global__ void limb_images_kernel(const float* __restrict__ W, size_t n, int K_pad, unsigned short* __restrict__ W3, unsigned short* __restrict__ W2) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const size_t row = t / K_pad;
    const int k = (int)(t - row * K_pad);
    const float x = W[t];
    const size_t blk = row * (K_pad / 16) + k / 16;
    const int e = k & 15;
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    W3[blk * 48 + e] = __builtin_bit_cast(unsigned short, h);
    W3[blk * 48 + 16 + e] = __builtin_bit_cast(unsigned short, m);
    W3[blk * 48 + 32 + e] = __builtin_bit_cast(unsigned short, l);
    if (W2) {
        const _Float16 h16 = (_Float16)x;
        const _Float16 l16 = (_Float16)((x - (float)h16) * 2048.0f);
        W2[blk * 32 + e] = __builtin_bit_cast(unsigned short, h16);
        W2[blk * 32 + 16 + e] = __builtin_bit_cast(unsigned short, l16);
    }
}
void launch_limb_images(const float* W, int rows, int K_pad, unsigned short* W3, unsigned short* W2, hipStream_t s) {
    const size_t n = (size_t)rows * K_pad;
    if (!n) return;
    hipLaunchKernelGGL(limb_images_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, n, K_pad, W3, W2);
    FC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- pack / fill
// dst[row, dst_col0 + c] = src[row, c] for c < src_cols ; zero for src_cols <= c < zero_to
__global__ void pack_rows_kernel(const float* __restrict__ src, int src_ld, int src_cols, float* __restrict__ dst, int dst_ld,
                                 int dst_col0, int zero_to, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    for (int c = lane; c < zero_to; c += 64)
        dst[(size_t)row * dst_ld + dst_col0 + c] = c < src_cols ? src[(size_t)row * src_ld + c] : 0.f;
}
void launch_pack_rows(const float* src, int src_ld, int src_cols, float* dst, int dst_ld, int dst_col0, int zero_to, int rows,
                      hipStream_t s) {
    if (rows <= 0) return;
    ProfScope ps("fc::pack_rows_kernel", 0.0, 4.0 * rows * ((double)src_cols + zero_to), s);
    hipLaunchKernelGGL(pack_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, src, src_ld, src_cols, dst, dst_ld, dst_col0, zero_to, rows);
    FC_HIP(hipGetLastError());
}

__global__ void fill_kernel(float* p, float v, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) p[i] = v;
}
void launch_fill(float* p, float v, size_t n, hipStream_t s) {
    if (!n) return;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ProfScope ps("fc::fill_kernel", 0.0, 4.0 * n, s);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, v, n);
    FC_HIP(hipGetLastError());
}

// extra context [B, X] -> per-point scalar rowscal[b*N + i] = extra[b, 0]   (inner_loop's einops.repeat, X == 1)
__global__ void repeat_extra_kernel(const float* extra, int X, float* rowscal, int B, int N) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B * N) rowscal[i] = extra[(size_t)(i / N) * X];
}
void launch_repeat_extra(const float* extra, int X, float* rowscal, int B, int N, hipStream_t s) {
    hipLaunchKernelGGL(repeat_extra_kernel, dim3((B * N + 255) / 256), dim3(256), 0, s, extra, X, rowscal, B, N);
    FC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- LayerNorm (no affine: gamma/beta are folded into the q projection)
// models/perceiver.py:18-26 -> torch.nn.LayerNorm(width), biased variance, eps 1e-5.  In place.
__global__ void layernorm_kernel(float* h, int ld, int width, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = h + (size_t)row * ld;
    float v[16];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < width ? p[c] : 0.f;
        sum += v[i];
    }
    const float mean = wave_sum(sum) / (float)width;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        const float d = c < width ? v[i] - mean : 0.f;
        sq += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)width + 1e-5f);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < width) p[c] = (v[i] - mean) * rstd;
    }
}
void launch_layernorm(float* h, int ld, int width, int rows, hipStream_t s) {
    if (width > 1024) throw Error(FC_ERR_UNSUPPORTED, "layernorm: width > 1024");
    ProfScope ps("fc::layernorm_kernel(float*, int, int, int)", 0.0, 8.0 * rows * width, s);
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, h, ld, width, rows);
    FC_HIP(hipGetLastError());
}

// q[row, :] = q_unnorm[row, :] * rsqrt(sum_blocks(sumsq[b][row]) / width + 1e-5) + q_bias   (LayerNorm -> q fold, common.h EPI_LNQ)
__global__ __launch_bounds__(256) void lnq_finalize_kernel(float* __restrict__ q, int ldq, const float* __restrict__ sumsq, int nslots, size_t pitch,
                                                           float inv_width, const float* __restrict__ q_bias, int rows) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;          // 16 threads per row, one float4 of the 64 q columns each
    const int row = (int)(t >> 4), c4 = ((int)t & 15) * 4;
    if (row >= rows) return;
    float ss = 0.f;
    for (int b = 0; b < nslots; ++b) ss += sumsq[(size_t)b * pitch + row];
    const float rstd = 1.0f / sqrtf(ss * inv_width + 1e-5f);
    float4* qp = reinterpret_cast<float4*>(q + (size_t)row * ldq + c4);
    const float4 v = *qp, bq = *reinterpret_cast<const float4*>(q_bias + c4);
    *qp = make_float4(v.x * rstd + bq.x, v.y * rstd + bq.y, v.z * rstd + bq.z, v.w * rstd + bq.w);
}
void launch_lnq_finalize(float* q, int ldq, const float* sumsq, int nslots, size_t pitch, int width, const float* q_bias, int rows, hipStream_t s) {
    ProfScope ps("fc::lnq_finalize_kernel", 0.0, 4.0 * rows * (128.0 + nslots), s);
    hipLaunchKernelGGL(lnq_finalize_kernel, dim3((unsigned)(((long)rows * 16 + 255) / 256)), dim3(256), 0, s, q, ldq, sumsq, nslots, pitch,
                       1.0f / (float)width, q_bias, rows);
    FC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- base density + latent export
// logprob[row] += sum_d(-0.5 log 2pi - 0.5 x_d^2) + log_const   (models/distributions.py:192-195)
// x is in the engine layout [x1 (d1) | pad | x2 (d2) | pad]; z_out (optional) gets the dense [rows, D] latent.
__global__ void base_density_kernel(const float* __restrict__ x, int ldx, int d1, int d1_pad, int d2, float* logprob, float log_const,
                                    float* z_out, int D, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + (size_t)row * ldx;
    float acc = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float v = c < d1 ? p[c] : p[d1_pad + (c - d1)];
        acc += -0.91893853320467274178f - 0.5f * v * v;
        if (z_out) z_out[(size_t)row * D + c] = v;
    }
    acc = wave_sum(acc);
    if (lane == 0) logprob[row] += acc + log_const;
}
void launch_base_density(const float* x, int ldx, int d1, int d1_pad, int d2, float* logprob, float log_const, float* z_out, int D,
                         int rows, hipStream_t s) {
    ProfScope ps("fc::base_density_kernel", 0.0, 4.0 * rows * (D + 2.0 + (z_out ? D : 0)), s);
    hipLaunchKernelGGL(base_density_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, d1, d1_pad, d2, logprob, log_const, z_out, D, rows);
    FC_HIP(hipGetLastError());
}

// One wave per point row; lane j handles dims j, j+64, ...  Parameters arrive in the tile-grouped dim-major layout of spline.h
// (the producing GEMM normally evaluates the spline in its own epilogue; this kernel serves the inverse direction and the
// non-split GEMM variants).  y2 overwrites x2 in place; forward adds the row's sum of logabsdet to logprob.
__global__ __launch_bounds__(256) void spline_rows_kernel(const float* __restrict__ params, int ldp, float* xbuf, int ldx, int x2_col0,
                                                          int d2, int K, float* logprob, int rows, int inverse) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float* pr = params + (size_t)row * ldp;
    float acc = 0.f;
    float* xr = xbuf + (size_t)row * ldx + x2_col0;
    for (int j = lane; j < d2; j += 64) {
        float y, lad;
        if (K == 8) {                                            // K = 8: register-slot column order (spline.h), gathered per dim
            float u[25];
#pragma unroll
            for (int pp = 0; pp < 25; ++pp) u[pp] = pr[spline_col(j, pp, 8)];
            rq_spline_elem<8>(xr[j], u, 1, inverse != 0, y, lad);
        } else {
            rq_any(K, xr[j], pr + spline_col(j, 0, K), 1, inverse != 0, y, lad);
        }
        xr[j] = y;
        acc += lad;
    }
    acc = wave_sum(acc);
    if (lane == 0 && !inverse) logprob[row] += acc;
}
void launch_spline(const float* params, int ldp, float* xbuf, int ldx, int x2_col0, int d2, int K, float* logprob, int rows, int inverse,
                   hipStream_t s) {
    if (K != 4 && K != 8 && K != 16) throw Error(FC_ERR_UNSUPPORTED, "spline: num_bins_spline must be 4, 8 or 16");
    if (ldp < spline_ncols(d2, K)) throw Error(FC_ERR_INVALID, "spline: parameter pitch too small");
    ProfScope ps("fc::spline_rows_kernel", 0.0, 4.0 * rows * ((3.0 * K + 1) * d2 + 2.0 * d2 + 2.0), s);
    hipLaunchKernelGGL(spline_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, params, ldp, xbuf, ldx, x2_col0, d2, K, logprob, rows, inverse);
    FC_HIP(hipGetLastError());
}

// logprob[row] += sum over the column tiles of the fused spline epilogue's per-tile log-det partials (fixed order: reproducible)
__global__ __launch_bounds__(256) void ldj_reduce_kernel(const float* __restrict__ part, int ntiles, size_t pitch, float* __restrict__ logprob, int rows) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float s = 0.f;
    for (int t = 0; t < ntiles; ++t) s += part[(size_t)t * pitch + row];
    logprob[row] += s;
}
void launch_ldj_reduce(const float* part, int ntiles, size_t pitch, float* logprob, int rows, hipStream_t s) {
    ProfScope ps("fc::ldj_reduce_kernel", 0.0, 4.0 * rows * (ntiles + 2.0), s);
    hipLaunchKernelGGL(ldj_reduce_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, part, ntiles, pitch, logprob, rows);
    FC_HIP(hipGetLastError());
}

__global__ void spline_flat_kernel(const float* x, const float* params, float* y, float* lad, int64_t n, int K, int inverse) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float yy, ll;
    rq_any(K, x[i], params + i * (3 * K + 1), 1, inverse != 0, yy, ll);
    y[i] = yy;
    lad[i] = ll;
}
void launch_spline_flat(const float* x, const float* params, float* y, float* lad, int64_t n, int K, int inverse, hipStream_t s) {
    if (K != 4 && K != 8 && K != 16) throw Error(FC_ERR_UNSUPPORTED, "spline: num_bins_spline must be 4, 8 or 16");
    if (n <= 0) return;
    hipLaunchKernelGGL(spline_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, params, y, lad, n, K, inverse);
    FC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- exponential coupling
// models/exponential_coupling.py:44-75: per point, W = rescale*tanh(scale*raw + shift) + reshift + 1e-8 (d2 x d2),
// y2 = expm(W) x2 + b, ldj = tr W ; inverse x2 = expm(-W)(y2 - b).  expm acts on the vector directly (scaling by 2^-s so that
// |W|_inf <= 1/2, 12-term Taylor action applied 2^s times): both of the reference's algorithms ('torch' = matrix_exp,
// 'original' = truncated series with a batch-global stopping rule, utils.py:294-327) converge to this value.
// One thread per point; only feasible for small d2 (the layer emits d2^2 + d2 numbers per point), like in the reference.
template <int DMAX>
__global__ void expm_coupling_kernel(const float* __restrict__ params, int ldp, float* xbuf, int ldx, int x2_col0, int d2,
                                     const float* __restrict__ scal4, float* logprob, int rows, int inverse) {
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    const float sc = scal4[0], sh = scal4[1], rs = scal4[2], rsh = scal4[3];
    const float* pr = params + (size_t)row * ldp;
    float w[DMAX * DMAX], v[DMAX], term[DMAX], acc[DMAX];
    float nrm = 0.f, tr = 0.f;
    for (int i = 0; i < d2; ++i) {
        float rsum = 0.f;
        for (int j = 0; j < d2; ++j) {
            float wij = rs * tanhf(sc * pr[i * d2 + j] + sh) + rsh + 1e-8f;
            if (i == j) tr += wij;
            if (inverse) wij = -wij;
            w[i * DMAX + j] = wij;
            rsum += fabsf(wij);
        }
        nrm = fmaxf(nrm, rsum);
    }
    float* xr = xbuf + (size_t)row * ldx + x2_col0;
    for (int i = 0; i < d2; ++i) v[i] = inverse ? xr[i] - pr[d2 * d2 + i] : xr[i];
    int s = 0;
    while (nrm > 0.5f && s < 24) { nrm *= 0.5f; ++s; }
    const float f = ldexpf(1.0f, -s);
    for (int rep = 0; rep < (1 << s); ++rep) {
        for (int i = 0; i < d2; ++i) { acc[i] = v[i]; term[i] = v[i]; }
        for (int k = 1; k <= 12; ++k) {
            float nt[DMAX];
            const float fk = f / (float)k;
            for (int i = 0; i < d2; ++i) {
                float a = 0.f;
                for (int j = 0; j < d2; ++j) a = fmaf(w[i * DMAX + j], term[j], a);
                nt[i] = a * fk;
            }
            for (int i = 0; i < d2; ++i) { term[i] = nt[i]; acc[i] += nt[i]; }
        }
        for (int i = 0; i < d2; ++i) v[i] = acc[i];
    }
    for (int i = 0; i < d2; ++i) xr[i] = inverse ? v[i] : v[i] + pr[d2 * d2 + i];
    if (!inverse) logprob[row] += tr;
}
void launch_expm_coupling(const float* params, int ldp, float* xbuf, int ldx, int x2_col0, int d2, const float* scal4, float* logprob,
                          int rows, int inverse, hipStream_t s) {
    if (d2 > 16) throw Error(FC_ERR_UNSUPPORTED, "ExponentialCoupling: latent_dim - latent_dim/2 > 16 is not supported (the layer emits d2^2 numbers per point)");
    if (ldp < d2 * d2 + d2) throw Error(FC_ERR_INVALID, "expm coupling: parameter pitch too small");
    ProfScope ps("fc::expm_coupling_kernel", 0.0, 4.0 * rows * (d2 * d2 + 3.0 * d2), s);
    hipLaunchKernelGGL(expm_coupling_kernel<16>, dim3((rows + 63) / 64), dim3(64), 0, s, params, ldp, xbuf, ldx, x2_col0, d2, scal4, logprob, rows, inverse);
    FC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- DGCNN edge-conv tail
// out[p, c] = LeakyReLU_0.2( max_j u[idx[p, j], c] + v[p, c] ), uv rows = [u (c_out) | v (c_out)].
// Equals conv(cat(nbr - x, x)) -> BN -> LeakyReLU -> max_k of models/pytorch_gcn.py:23-47,85-99 because the 1x1 conv is
// linear (W1 nbr + (W2 - W1) x), eval BN is a per-channel affine map folded into u and v, and LeakyReLU is monotone.
__global__ void gather_max_kernel(const float* __restrict__ uv, int lduv, int c_out, const int32_t* __restrict__ idx, int k,
                                  float* out, int ldo, int out_col0, int M, int m_stride, int total) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= total) return;
    const int b = p / M, i = p - b * M;
    const size_t base = (size_t)b * m_stride;
    const int32_t* ip = idx + ((size_t)b * M + i) * k;
    for (int c = lane; c < c_out; c += 64) {
        float mx = -INFINITY;
        for (int j = 0; j < k; ++j) mx = fmaxf(mx, uv[(base + ip[j]) * lduv + c]);
        const float t = mx + uv[(base + i) * lduv + c_out + c];
        out[(base + i) * ldo + out_col0 + c] = t > 0.f ? t : 0.2f * t;
    }
}
void launch_gather_max(const float* uv, int lduv, int c_out, const int32_t* idx, int k, float* out, int ldo, int out_col0, int B, int M,
                       int m_stride_rows, hipStream_t s) {
    const int total = B * M;
    ProfScope ps("fc::gather_max_kernel", 0.0, 4.0 * total * ((double)k * c_out + 2.0 * c_out + k), s);
    hipLaunchKernelGGL(gather_max_kernel, dim3((total + 3) / 4), dim3(256), 0, s, uv, lduv, c_out, idx, k, out, ldo, out_col0, M, m_stride_rows, total);
    FC_HIP(hipGetLastError());
}

// global embedder pooling: out[b] = [max_i t[b,i,:] | mean_i t[b,i,:]]  (models/pytorch_gcn.py:178-182)
__global__ void pool_max_mean_kernel(const float* __restrict__ t, int ldt, int width, float* out, int ldo, int M, int m_stride) {
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;   // 4 row-partitions per block
    __shared__ float smx[4][64], ssum[4][64];
    float mx = -INFINITY, sum = 0.f;
    if (c < width)
        for (int i = part; i < M; i += 4) {
            const float v = t[((size_t)b * m_stride + i) * ldt + c];
            mx = fmaxf(mx, v);
            sum += v;
        }
    smx[part][threadIdx.x & 63] = mx;
    ssum[part][threadIdx.x & 63] = sum;
    __syncthreads();
    if (part == 0 && c < width) {
        const int l = threadIdx.x;
        out[(size_t)b * ldo + c] = fmaxf(fmaxf(smx[0][l], smx[1][l]), fmaxf(smx[2][l], smx[3][l]));
        out[(size_t)b * ldo + width + c] = ((ssum[0][l] + ssum[1][l]) + (ssum[2][l] + ssum[3][l])) / (float)M;
    }
}
void launch_pool_max_mean(const float* t, int ldt, int width, float* out, int ldo, int B, int M, int m_stride_rows, hipStream_t s) {
    hipLaunchKernelGGL(pool_max_mean_kernel, dim3((width + 63) / 64, B), dim3(256), 0, s, t, ldt, width, out, ldo, M, m_stride_rows);
    FC_HIP(hipGetLastError());
}

}  // namespace fc
