// Row-wise training primitives: rational-quadratic spline coupling element and LayerNorm, forward + backward.
//   spline   models/spline_coupling.py:24-169 in the REFERENCE's parameter layout: the coupling MLP's output row is [d2][3K+1] =
//            per transformed dim [K width logits | K height logits | K+1 derivative logits] (spline_coupling.py:196-203).  One
//            workgroup per point; forward = the inference element function (spline.h), backward = its analytic gradient w.r.t. x and
//            all 3K+1 logits (through the softmax of the bin widths / heights, the cumulative knots and the softplus derivatives).
//   layernorm  torch.nn.LayerNorm(width, eps 1e-5) of PreNorm (models/perceiver.py:18-27); backward returns dx and the panel
//            dy * xhat whose column sums are d gamma (d beta = column sums of dy), reduced by fc_train_colsum_f32 in a fixed order.
#include <hip/hip_runtime.h>

#include "common.h"
#include "spline.h"

namespace fc {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    // deterministic: xor tree inside each wave, then waves in order
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}

// One workgroup per point.  The point's d2 (3K+1) logits are staged through LDS with coalesced 16-byte loads (a thread evaluating one
// dim reads 3K+1 consecutive floats: straight from HBM that is a 100-byte stride across the wave); 3K+1 is odd, so the per-thread
// reads from LDS are conflict-free.
template <int K>
__global__ __launch_bounds__(256) void spline_train_fwd_kernel(const float* __restrict__ x2, int ldx, const float* __restrict__ params, int ldp,
                                                               float* __restrict__ y2, int ldy, float* __restrict__ ldj, int d2, int d2_pad) {
    extern __shared__ float sp[];                        // [d2 (3K+1)] rounded up to 4
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const int np = d2 * (3 * K + 1), np4 = (np + 3) >> 2;
    const float4* src = reinterpret_cast<const float4*>(params + row * ldp);          // panel rows are 16-byte aligned, ldp >= round_up(np, 32)
    for (int i = threadIdx.x; i < np4; i += 256) reinterpret_cast<float4*>(sp)[i] = src[i];
    __syncthreads();
    float part = 0.f;
    for (int dim = threadIdx.x; dim < d2_pad; dim += 256) {
        float y = 0.f, lad = 0.f;
        if (dim < d2) rq_spline_elem<K>(x2[row * ldx + dim], sp + dim * (3 * K + 1), 1, false, y, lad);
        y2[row * ldy + dim] = y;
        part += lad;
    }
    const float s = block_sum_256(part, red);
    if (threadIdx.x == 0) ldj[row] = s;
}

// gradient of (y, lad) of ONE spline element: gx = dL/dx, gu[3K+1] = dL/d logits, given gy = dL/dy and gl = dL/dlad
template <int K>
__device__ __forceinline__ void rq_spline_bwd_elem(float x, const float* __restrict__ u, float gy, float gl, float& gx, float* __restrict__ gu) {
    constexpr float B = 3.0f, MINW = 1e-3f, MINH = 1e-3f, MIND = 1e-3f;
    if (!(x >= -B && x <= B)) {
        gx = gy;
#pragma unroll
        for (int i = 0; i < 3 * K + 1; ++i) gu[i] = 0.f;
        return;
    }
    float pw[K], ph[K], cw[K + 1], ch[K + 1];
    {
        float mx = u[0];
#pragma unroll
        for (int i = 0; i < K; ++i) { pw[i] = u[i]; mx = fmaxf(mx, pw[i]); }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) { pw[i] = expf(pw[i] - mx); sum += pw[i]; }
        float c = 0.f;
        cw[0] = -B;
#pragma unroll
        for (int i = 0; i < K; ++i) { pw[i] = pw[i] / sum; c += MINW + (1.0f - MINW * K) * pw[i]; cw[i + 1] = 2.0f * B * c - B; }
        cw[K] = B;
    }
    {
        float mx = u[K];
#pragma unroll
        for (int i = 0; i < K; ++i) { ph[i] = u[K + i]; mx = fmaxf(mx, ph[i]); }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) { ph[i] = expf(ph[i] - mx); sum += ph[i]; }
        float c = 0.f;
        ch[0] = -B;
#pragma unroll
        for (int i = 0; i < K; ++i) { ph[i] = ph[i] / sum; c += MINH + (1.0f - MINH * K) * ph[i]; ch[i + 1] = 2.0f * B * c - B; }
        ch[K] = B;
    }
    int bin = 0;
#pragma unroll
    for (int i = 1; i <= K; ++i) bin += (x >= cw[i] + (i == K ? 1e-6f : 0.f)) ? 1 : 0;
    float in_cw = cw[0], w = cw[1] - cw[0], in_ch = ch[0], h = ch[1] - ch[0];
    float raw0 = -1e-3f, raw1 = u[2 * K];
#pragma unroll
    for (int i = 1; i < K; ++i)
        if (bin == i) { in_cw = cw[i]; w = cw[i + 1] - cw[i]; in_ch = ch[i]; h = ch[i + 1] - ch[i]; raw0 = u[2 * K + i - 1]; raw1 = u[2 * K + i]; }
    auto softplus = [](float v) { return v > 20.f ? v : log1pf(expf(v)); };
    auto sigmoid = [](float v) { return 1.0f / (1.0f + expf(-v)); };
    const float d0 = MIND + softplus(raw0), d1 = MIND + softplus(raw1);
    const float s = h / w;
    const float th = (x - in_cw) / w, omt = 1.0f - th, tt = th * omt;
    const float A = s * th * th + d0 * tt;
    const float E = d0 + d1 - 2.0f * s;
    const float den = s + E * tt;
    const float G = d1 * th * th + 2.0f * s * tt + d0 * omt * omt;
    const float rden = 1.0f / den, rG = 1.0f / G;
    const float dtt = 1.0f - 2.0f * th;
    // y = ch + h A / den ; lad = 2 log s + log G - 2 log den
    const float y_A = h * rden, y_den = -h * A * rden * rden;
    const float g_th = gy * (y_A * (2.0f * s * th + d0 * dtt) + y_den * (E * dtt)) +
                       gl * ((2.0f * d1 * th + 2.0f * s * dtt - 2.0f * d0 * omt) * rG - 2.0f * E * dtt * rden);
    const float g_s = gy * (y_A * th * th + y_den * (1.0f - 2.0f * tt)) + gl * (2.0f / s + 2.0f * tt * rG - 2.0f * (1.0f - 2.0f * tt) * rden);
    const float g_d0 = gy * (y_A * tt + y_den * tt) + gl * (omt * omt * rG - 2.0f * tt * rden);
    const float g_d1 = gy * (y_den * tt) + gl * (th * th * rG - 2.0f * tt * rden);
    const float g_h = gy * A * rden + g_s / w;                              // direct + through s = h / w
    const float g_w = -g_s * s / w - g_th * th / w;                         // through s and through th = (x - cw) / w
    const float g_cw = -g_th / w;
    gx = g_th / w;
    // widths: bin b owns w_b = 2B p_b and its left knot cw_b = -B + sum_{k<b} w_k ; p = MIN + (1 - K MIN) softmax(u)
    {
        float gs[K], dot = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float gwk = (k < bin ? g_cw : 0.f) + (k == bin ? g_w : 0.f);
            gs[k] = 2.0f * B * (1.0f - MINW * K) * gwk;
            dot += pw[k] * gs[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) gu[k] = pw[k] * (gs[k] - dot);
    }
    {
        float gs[K], dot = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float ghk = (k < bin ? gy : 0.f) + (k == bin ? g_h : 0.f);       // d y / d ch_b = 1
            gs[k] = 2.0f * B * (1.0f - MINH * K) * ghk;
            dot += ph[k] * gs[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) gu[K + k] = ph[k] * (gs[k] - dot);
    }
    // derivative logits: knot j >= 1 uses logit j-1 (knot 0 is the padded constant, the last logit is never read)
    const float gr0 = g_d0 * sigmoid(raw0), gr1 = g_d1 * sigmoid(raw1);
#pragma unroll
    for (int j = 0; j <= K; ++j) gu[2 * K + j] = (j == bin - 1 ? gr0 : 0.f) + (j == bin ? gr1 : 0.f);
}

template <int K>
__global__ __launch_bounds__(256) void spline_train_bwd_kernel(const float* __restrict__ x2, int ldx, const float* __restrict__ params, int ldp,
                                                               const float* __restrict__ dy2, int lddy, const float* __restrict__ dldj,
                                                               float* __restrict__ dx2, int lddx, float* __restrict__ dparams, int lddp, int d2,
                                                               int d2_pad, int np_pad, float* __restrict__ rowmax) {
    extern __shared__ float sp[];                        // logits in, their gradients out (in place): [round_up(d2 (3K+1), 32)]
    const size_t row = blockIdx.x;
    const int np = d2 * (3 * K + 1), np4 = (np + 3) >> 2;
    const float4* src = reinterpret_cast<const float4*>(params + row * ldp);
    for (int i = threadIdx.x; i < np4; i += 256) reinterpret_cast<float4*>(sp)[i] = src[i];
    __syncthreads();
    const float gl = dldj[row];
    for (int dim = threadIdx.x; dim < d2_pad; dim += 256) {
        float gx = 0.f;
        if (dim < d2) {
            float gu[3 * K + 1];
            float* u = sp + dim * (3 * K + 1);
            rq_spline_bwd_elem<K>(x2[row * ldx + dim], u, dy2[row * lddy + dim], gl, gx, gu);
#pragma unroll
            for (int i = 0; i < 3 * K + 1; ++i) u[i] = gu[i];                  // this thread's own logits: nobody else reads them
        }
        dx2[row * lddx + dim] = gx;
    }
    for (int c = np + threadIdx.x; c < np_pad; c += 256) sp[c] = 0.f;
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(dparams + row * lddp);
    float m = 0.f;
    for (int i = threadIdx.x; i < (np_pad >> 2); i += 256) {
        const float4 v = reinterpret_cast<const float4*>(sp)[i];
        dst[i] = v;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) m = INFINITY;      // (fmaxf drops a NaN: the consumer must see it)
    }
    if (rowmax) {
        // max |row| for the data-gradient GEMM of the parameter layer (train.hip: train_rowmax_reserve / _take)
        __shared__ float red[4];
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) rowmax[row] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
}

// ---------------------------------------------------------------- LayerNorm: one wave per row
__global__ __launch_bounds__(256) void layernorm_train_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ y, int ldy,
                                                                  float* __restrict__ stats, int rows, int width, int width_pad, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int c = lane; c < width; c += 64) s += xr[c];
    for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
    const float mean = s / width;
    float v = 0.f;
    for (int c = lane; c < width; c += 64) { const float d = xr[c] - mean; v += d * d; }
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    const float rstd = 1.0f / sqrtf(v / width + eps);
    for (int c = lane; c < width_pad; c += 64) y[(size_t)row * ldy + c] = c < width ? (xr[c] - mean) * rstd * gamma[c] + beta[c] : 0.f;
    if (lane == 0) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
}

// dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma ; t = dy xhat (column sums = d gamma)
__global__ __launch_bounds__(256) void layernorm_train_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                                  const float* __restrict__ dy, int lddy, const float* __restrict__ stats,
                                                                  float* __restrict__ dx, int lddx, float* __restrict__ t, int ldt, int rows,
                                                                  int rows_pad, int width, int width_pad) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows_pad) return;
    if (row >= rows) {
        for (int c = lane; c < width_pad; c += 64) { dx[(size_t)row * lddx + c] = 0.f; t[(size_t)row * ldt + c] = 0.f; }
        return;
    }
    const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
    const float* xr = x + (size_t)row * ldx;
    const float* gr = dy + (size_t)row * lddy;
    float a = 0.f, b = 0.f;
    for (int c = lane; c < width; c += 64) {
        const float g = gr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
        a += g; b += g * xh;
    }
    for (int m = 1; m < 64; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
    a /= width; b /= width;
    for (int c = lane; c < width_pad; c += 64) {
        float o = 0.f, tv = 0.f;
        if (c < width) {
            const float xh = (xr[c] - mean) * rstd;
            o = rstd * (gr[c] * gamma[c] - a - xh * b);
            tv = gr[c] * xh;
        }
        dx[(size_t)row * lddx + c] = o;
        t[(size_t)row * ldt + c] = tv;
    }
}


// ---------------------------------------------------------------- the three per-element closures of the flow, one workgroup per point
// affine coupling (models/affine_coupling.py:23-46): st = [raw scale d2 | shift d2];  y2 = x2 s + t,  ldj = sum log s
__device__ __forceinline__ float affine_scale(float raw, int scale_fn, float& ds) {
    if (scale_fn == FC_SCALE_EXP) { const float s = expf(raw); ds = s; return s; }
    const float sg = 1.0f / (1.0f + expf(-raw));
    ds = 2.0f * sg * (1.0f - sg) * (float)(1.0 - 1e-8);
    return (2.0f * sg - 1.0f) * (float)(1.0 - 1e-8) + 1.0f;
}
__global__ __launch_bounds__(256) void affine_train_fwd_kernel(const float* __restrict__ x2, int ldx, const float* __restrict__ st, int ldst,
                                                               float* __restrict__ y2, int ldy, float* __restrict__ ldj, int d2, int d2_pad, int scale_fn) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    float part = 0.f;
    for (int j = threadIdx.x; j < d2_pad; j += 256) {
        float y = 0.f;
        if (j < d2) {
            float ds;
            const float s = affine_scale(st[row * ldst + j], scale_fn, ds);
            y = x2[row * ldx + j] * s + st[row * ldst + d2 + j];
            part += logf(s);
        }
        y2[row * ldy + j] = y;
    }
    const float tot = block_sum_256(part, red);
    if (threadIdx.x == 0) ldj[row] = tot;
}
__global__ __launch_bounds__(256) void affine_train_bwd_kernel(const float* __restrict__ x2, int ldx, const float* __restrict__ st, int ldst,
                                                               const float* __restrict__ dy2, int lddy, const float* __restrict__ dldj,
                                                               float* __restrict__ dx2, int lddx, float* __restrict__ dst, int lddst, int d2,
                                                               int d2_pad, int nst_pad, int scale_fn) {
    const size_t row = blockIdx.x;
    const float gl = dldj[row];
    for (int j = threadIdx.x; j < d2_pad; j += 256) {
        float gx = 0.f;
        if (j < d2) {
            float ds;
            const float s = affine_scale(st[row * ldst + j], scale_fn, ds);
            const float gy = dy2[row * lddy + j];
            gx = gy * s;
            dst[row * lddst + j] = (gy * x2[row * ldx + j] + gl / s) * ds;
            dst[row * lddst + d2 + j] = gy;
        }
        dx2[row * lddx + j] = gx;
    }
    for (int c = 2 * d2 + threadIdx.x; c < nst_pad; c += 256) dst[row * lddst + c] = 0.f;
}

// augmenter draw (models/augmenter.py:49-63, distributions.py:128-153): p = [mean nz | log std nz];  z = mean + eps exp(log std),
// ldj = -log N(z; mean, std) summed = sum (eps^2 / 2 + log std + log(2 pi) / 2)
__global__ __launch_bounds__(256) void gauss_train_fwd_kernel(const float* __restrict__ p, int ldp, const float* __restrict__ eps, float* __restrict__ z,
                                                              int ldz, float* __restrict__ ldj, int nz, int nz_pad, float clamp) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    float part = 0.f;
    for (int j = threadIdx.x; j < nz_pad; j += 256) {
        float v = 0.f;
        if (j < nz) {
            const float e = eps[row * nz + j], ls = p[row * ldp + nz + j];
            float sc = expf(ls), lsc = ls;
            if (clamp > 0.f && sc > clamp) { sc = clamp; lsc = logf(clamp); }          // distributions.py:134-137 (clamp_max on the std)
            v = p[row * ldp + j] + e * sc;
            part += 0.5f * e * e + lsc + 0.91893853320467274178f;
        }
        z[row * ldz + j] = v;
    }
    const float tot = block_sum_256(part, red);
    if (threadIdx.x == 0) ldj[row] = tot;
}
__global__ __launch_bounds__(256) void gauss_train_bwd_kernel(const float* __restrict__ p, int ldp, const float* __restrict__ eps,
                                                              const float* __restrict__ dz, int lddz, const float* __restrict__ dldj,
                                                              float* __restrict__ dp, int lddp, int nz, int np_pad, float clamp) {
    const size_t row = blockIdx.x;
    const float gl = dldj[row];
    for (int j = threadIdx.x; j < nz; j += 256) {
        const float g = dz[row * lddz + j];
        const float sc = expf(p[row * ldp + nz + j]);
        dp[row * lddp + j] = g;
        dp[row * lddp + nz + j] = (clamp > 0.f && sc > clamp) ? 0.f : g * eps[row * nz + j] * sc + gl;
    }
    for (int c = 2 * nz + threadIdx.x; c < np_pad; c += 256) dp[row * lddp + c] = 0.f;
}

// Slice (models/slice.py:31-44 + distributions.py:140-142): out[row] = sum_j log N(v_j; mean_j, std_j), p = [mean nz | log std nz]
__global__ __launch_bounds__(256) void normlp_train_fwd_kernel(const float* __restrict__ v, int ldv, const float* __restrict__ p, int ldp,
                                                               float* __restrict__ out, int nz, float clamp) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    float part = 0.f;
    for (int j = threadIdx.x; j < nz; j += 256) {
        const float ls = p[row * ldp + nz + j];
        float sc = expf(ls), lsc = ls;
        if (clamp > 0.f && sc > clamp) { sc = clamp; lsc = logf(clamp); }
        const float d = (v[row * ldv + j] - p[row * ldp + j]) / sc;
        part += -0.5f * d * d - lsc - 0.91893853320467274178f;
    }
    const float tot = block_sum_256(part, red);
    if (threadIdx.x == 0) out[row] = tot;
}
__global__ __launch_bounds__(256) void normlp_train_bwd_kernel(const float* __restrict__ v, int ldv, const float* __restrict__ p, int ldp,
                                                               const float* __restrict__ g, float* __restrict__ dv, int lddv, float* __restrict__ dp,
                                                               int lddp, int nz, int nz_pad, int np_pad, float clamp) {
    const size_t row = blockIdx.x;
    const float gr = g[row];
    for (int j = threadIdx.x; j < nz_pad; j += 256) {
        float gv = 0.f;
        if (j < nz) {
            const float sc0 = expf(p[row * ldp + nz + j]);
            const bool cl = clamp > 0.f && sc0 > clamp;
            const float sc = cl ? clamp : sc0;
            const float d = (v[row * ldv + j] - p[row * ldp + j]) / sc;
            gv = -gr * d / sc;
            dp[row * lddp + j] = -gv;
            dp[row * lddp + nz + j] = cl ? 0.f : gr * (d * d - 1.0f);
        }
        dv[row * lddv + j] = gv;
    }
    for (int c = 2 * nz + threadIdx.x; c < np_pad; c += 256) dp[row * lddp + c] = 0.f;
}

// base density (models/distributions.py:192-195): out[row] = sum_c (-x^2 / 2 - log(2 pi) / 2) over `width` columns
__global__ __launch_bounds__(256) void base_train_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int width) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    float part = 0.f;
    for (int j = threadIdx.x; j < width; j += 256) { const float v = x[row * ldx + j]; part -= 0.5f * v * v; }
    const float tot = block_sum_256(part, red);
    if (threadIdx.x == 0) out[row] = tot - 0.91893853320467274178f * width;
}
__global__ __launch_bounds__(256) void base_train_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, float* __restrict__ dx,
                                                             int lddx, int width, int width_pad) {
    const size_t row = blockIdx.x;
    const float gr = g[row];
    for (int j = threadIdx.x; j < width_pad; j += 256) dx[row * lddx + j] = j < width ? -x[row * ldx + j] * gr : 0.f;
}


// ---------------------------------------------------------------- ExponentialCoupling (models/exponential_coupling.py:44-58, utils.py:294-327)
// o = [d2 x d2 raw matrix | d2 shift] per point, W = rescale tanh(scale raw + shift) + reshift + 1e-8, y2 = expm(W) x2 + b, ldj = tr W.
// Forward = the inference kernel's arithmetic (misc.hip): A = W 2^-s with |A|_inf <= 1/2, the 12-term Taylor ACTION u = sum_k A^k v / k!
// applied 2^s times.  Backward = exact reverse mode through that recurrence: the intermediate vectors v_rep are kept, each action is
// replayed for its terms t_k and differentiated with mu_12 = lambda, mu_{k-1} = lambda + A^T mu_k / k, dA += mu_k t_{k-1}^T / k.
// One thread per point with per-thread arrays (scratch): the layer only exists for small d2, like in the reference.
constexpr int EX_D = 16, EX_REP = 64;
__device__ __forceinline__ int expm_prepare(const float* pr, int d2, const float* scal4, float* w, float& tr) {
    const float sc = scal4[0], sh = scal4[1], rs = scal4[2], rsh = scal4[3];
    float nrm = 0.f;
    tr = 0.f;
    for (int i = 0; i < d2; ++i) {
        float rsum = 0.f;
        for (int j = 0; j < d2; ++j) {
            const float wij = rs * tanhf(sc * pr[i * d2 + j] + sh) + rsh + 1e-8f;
            if (i == j) tr += wij;
            w[i * EX_D + j] = wij;
            rsum += fabsf(wij);
        }
        nrm = fmaxf(nrm, rsum);
    }
    int s = 0;
    while (nrm > 0.5f && s < 6) { nrm *= 0.5f; ++s; }          // 2^6 = EX_REP stored states at most
    return s;
}
__global__ void expm_train_fwd_kernel(const float* __restrict__ x2, int ldx, const float* __restrict__ o, int ldo, const float* __restrict__ scal4,
                                      float* __restrict__ y2, int ldy, float* __restrict__ ldj, int rows, int d2, int d2_pad, int* __restrict__ status) {
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    const float* pr = o + (size_t)row * ldo;
    float w[EX_D * EX_D], v[EX_D], term[EX_D], acc[EX_D];
    float tr;
    const int s = expm_prepare(pr, d2, scal4, w, tr);
    {
        float nrm = 0.f;
        for (int i = 0; i < d2; ++i) { float r = 0.f; for (int j = 0; j < d2; ++j) r += fabsf(w[i * EX_D + j]); nrm = fmaxf(nrm, r); }
        if (ldexpf(nrm, -s) > 0.5f) atomicOr(status, 1);         // |W| too large for the stored-state budget: reported by the host
    }
    const float f = ldexpf(1.0f, -s);
    for (int i = 0; i < d2; ++i) v[i] = x2[(size_t)row * ldx + i];
    for (int rep = 0; rep < (1 << s); ++rep) {
        for (int i = 0; i < d2; ++i) { acc[i] = v[i]; term[i] = v[i]; }
        for (int k = 1; k <= 12; ++k) {
            float nt[EX_D];
            const float fk = f / (float)k;
            for (int i = 0; i < d2; ++i) {
                float a = 0.f;
                for (int j = 0; j < d2; ++j) a = fmaf(w[i * EX_D + j], term[j], a);
                nt[i] = a * fk;
            }
            for (int i = 0; i < d2; ++i) { term[i] = nt[i]; acc[i] += nt[i]; }
        }
        for (int i = 0; i < d2; ++i) v[i] = acc[i];
    }
    for (int i = 0; i < d2_pad; ++i) y2[(size_t)row * ldy + i] = i < d2 ? v[i] + pr[d2 * d2 + i] : 0.f;
    ldj[row] = tr;
}

// do [rows, ldo] gets d raw matrix | d shift (pads zero); dscal [rows, 4] the per-point parts of d scale, d shift, d rescale, d reshift
__global__ void expm_train_bwd_kernel(const float* __restrict__ x2, int ldx, const float* __restrict__ o, int ldo, const float* __restrict__ scal4,
                                      const float* __restrict__ dy2, int lddy, const float* __restrict__ dldj, float* __restrict__ dx2, int lddx,
                                      float* __restrict__ dout, int lddo, float* __restrict__ dscal, int rows, int d2, int d2_pad, int no_pad) {
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    const float* pr = o + (size_t)row * ldo;
    float w[EX_D * EX_D], dA[EX_D * EX_D], V[EX_REP + 1][EX_D], T[13][EX_D], lam[EX_D], mu[EX_D];
    float tr;
    const int s = expm_prepare(pr, d2, scal4, w, tr);
    const float f = ldexpf(1.0f, -s);
    const int R = 1 << s;
    for (int i = 0; i < d2; ++i) V[0][i] = x2[(size_t)row * ldx + i];
    for (int rep = 0; rep < R; ++rep) {                           // forward again, keeping every state
        float term[EX_D], acc[EX_D];
        for (int i = 0; i < d2; ++i) { acc[i] = V[rep][i]; term[i] = V[rep][i]; }
        for (int k = 1; k <= 12; ++k) {
            float nt[EX_D];
            const float fk = f / (float)k;
            for (int i = 0; i < d2; ++i) {
                float a = 0.f;
                for (int j = 0; j < d2; ++j) a = fmaf(w[i * EX_D + j], term[j], a);
                nt[i] = a * fk;
            }
            for (int i = 0; i < d2; ++i) { term[i] = nt[i]; acc[i] += nt[i]; }
        }
        for (int i = 0; i < d2; ++i) V[rep + 1][i] = acc[i];
    }
    for (int i = 0; i < d2 * EX_D; ++i) dA[i] = 0.f;
    for (int i = 0; i < d2; ++i) lam[i] = dy2[(size_t)row * lddy + i];
    for (int rep = R - 1; rep >= 0; --rep) {
        // terms of this action: t_0 = v, t_k = (f / k) W t_{k-1}
        for (int i = 0; i < d2; ++i) T[0][i] = V[rep][i];
        for (int k = 1; k <= 12; ++k) {
            const float fk = f / (float)k;
            for (int i = 0; i < d2; ++i) {
                float a = 0.f;
                for (int j = 0; j < d2; ++j) a = fmaf(w[i * EX_D + j], T[k - 1][j], a);
                T[k][i] = a * fk;
            }
        }
        for (int i = 0; i < d2; ++i) mu[i] = lam[i];              // mu_12
        for (int k = 12; k >= 1; --k) {
            const float fk = f / (float)k;
            // d W += (f / k) mu_k t_{k-1}^T ; mu_{k-1} = lambda + (f / k) W^T mu_k
            float nm[EX_D];
            for (int j = 0; j < d2; ++j) nm[j] = lam[j];
            for (int i = 0; i < d2; ++i) {
                const float mi = mu[i] * fk;
                for (int j = 0; j < d2; ++j) {
                    dA[i * EX_D + j] = fmaf(mi, T[k - 1][j], dA[i * EX_D + j]);
                    nm[j] = fmaf(w[i * EX_D + j], mi, nm[j]);
                }
            }
            for (int j = 0; j < d2; ++j) mu[j] = nm[j];
        }
        for (int i = 0; i < d2; ++i) lam[i] = mu[i];              // adjoint of this action's input
    }
    for (int i = 0; i < d2_pad; ++i) dx2[(size_t)row * lddx + i] = i < d2 ? lam[i] : 0.f;
    // through W = rescale tanh(scale raw + shift) + reshift + 1e-8 (and ldj = tr W)
    const float sc = scal4[0], sh = scal4[1], rs = scal4[2];
    const float gl = dldj[row];
    float g_sc = 0.f, g_sh = 0.f, g_rs = 0.f, g_rsh = 0.f;
    float* dr = dout + (size_t)row * lddo;
    for (int i = 0; i < d2; ++i)
        for (int j = 0; j < d2; ++j) {
            const float gw = dA[i * EX_D + j] + (i == j ? gl : 0.f);
            const float raw = pr[i * d2 + j];
            const float t = tanhf(sc * raw + sh);
            const float gt = gw * rs * (1.0f - t * t);
            dr[i * d2 + j] = gt * sc;
            g_sc += gt * raw; g_sh += gt; g_rs += gw * t; g_rsh += gw;
        }
    for (int i = 0; i < d2; ++i) dr[d2 * d2 + i] = dy2[(size_t)row * lddy + i];
    for (int c = d2 * d2 + d2; c < no_pad; ++c) dr[c] = 0.f;
    dscal[(size_t)row * 4 + 0] = g_sc; dscal[(size_t)row * 4 + 1] = g_sh; dscal[(size_t)row * 4 + 2] = g_rs; dscal[(size_t)row * 4 + 3] = g_rsh;
}

template <int K>
static void spline_fwd_k(const float* x2, int ldx, const float* params, int ldp, float* y2, int ldy, float* ldj, int rows, int d2, hipStream_t s) {
    ProfScope ps("fc::spline_train_fwd_kernel", 0.0, (double)rows * d2 * (3 * K + 3) * 4.0, s);
    const size_t lds = (size_t)round_up(d2 * (3 * K + 1), 32) * sizeof(float);
    if (lds > 60 * 1024) throw Error(FC_ERR_UNSUPPORTED, "training spline: more than 15360 logits per point");
    hipLaunchKernelGGL(spline_train_fwd_kernel<K>, dim3(rows), dim3(256), lds, s, x2, ldx, params, ldp, y2, ldy, ldj, d2, round_up(d2, 32));
    FC_HIP(hipGetLastError());
}
template <int K>
static void spline_bwd_k(const float* x2, int ldx, const float* params, int ldp, const float* dy2, int lddy, const float* dldj, float* dx2, int lddx,
                         float* dparams, int lddp, int rows, int d2, hipStream_t s) {
    // rows beyond `rows` of the row-maximum buffer (the GEMM reads a multiple of 256) are zero: scale 1 on zero rows
    const int rows_pad = round_up(rows, ROW_PAD);
    float* rowmax = lddp % 64 == 0 ? train_rowmax_reserve(dparams, rows_pad, s) : nullptr;
    if (rowmax && rows_pad > rows) FC_HIP(hipMemsetAsync(rowmax + rows, 0, (size_t)(rows_pad - rows) * 4, s));
    ProfScope ps("fc::spline_train_bwd_kernel", 0.0, (double)rows * d2 * (6 * K + 5) * 4.0, s);
    const size_t lds = (size_t)round_up(d2 * (3 * K + 1), 32) * sizeof(float);
    if (lds > 60 * 1024) throw Error(FC_ERR_UNSUPPORTED, "training spline: more than 15360 logits per point");
    hipLaunchKernelGGL(spline_train_bwd_kernel<K>, dim3(rows), dim3(256), lds, s, x2, ldx, params, ldp, dy2, lddy, dldj, dx2, lddx, dparams, lddp, d2,
                       round_up(d2, 32), round_up(d2 * (3 * K + 1), 32), rowmax);
    FC_HIP(hipGetLastError());
}

}  // namespace fc

using namespace fc;


extern "C" {

int fc_train_rqspline_fwd_f32(const float* x2, int32_t ldx, const float* params, int32_t ldp, float* y2, int32_t ldy, float* ldj, int32_t rows,
                              int32_t d2, int32_t K, void* stream) {
    FC_API_BEGIN
    if (!x2 || !params || !y2 || !ldj || rows < 1 || d2 < 1 || ldx < d2 || ldy < round_up(d2, 32) || ldp < round_up(d2 * (3 * K + 1), 4) || ldp % 4 != 0 ||
        ((uintptr_t)params & 15))
        throw Error(FC_ERR_INVALID, "fc_train_rqspline_fwd_f32: bad argument (params: 16-byte aligned rows, pitch a multiple of 4)");
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
        case 4: spline_fwd_k<4>(x2, ldx, params, ldp, y2, ldy, ldj, rows, d2, s); break;
        case 8: spline_fwd_k<8>(x2, ldx, params, ldp, y2, ldy, ldj, rows, d2, s); break;
        case 16: spline_fwd_k<16>(x2, ldx, params, ldp, y2, ldy, ldj, rows, d2, s); break;
        default: throw Error(FC_ERR_UNSUPPORTED, "fc_train_rqspline_fwd_f32: num_bins must be 4, 8 or 16");
    }
    FC_API_END
}

int fc_train_rqspline_bwd_f32(const float* x2, int32_t ldx, const float* params, int32_t ldp, const float* dy2, int32_t lddy, const float* dldj,
                              float* dx2, int32_t lddx, float* dparams, int32_t lddp, int32_t rows, int32_t d2, int32_t K, void* stream) {
    FC_API_BEGIN
    if (!x2 || !params || !dy2 || !dldj || !dx2 || !dparams || rows < 1 || d2 < 1 || ldx < d2 || lddy < d2 || lddx < round_up(d2, 32) ||
        ldp < round_up(d2 * (3 * K + 1), 4) || ldp % 4 != 0 || lddp < round_up(d2 * (3 * K + 1), 32) || lddp % 4 != 0 || (((uintptr_t)params | (uintptr_t)dparams) & 15))
        throw Error(FC_ERR_INVALID, "fc_train_rqspline_bwd_f32: bad argument (params / dparams: 16-byte aligned rows, pitches multiples of 4)");
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
        case 4: spline_bwd_k<4>(x2, ldx, params, ldp, dy2, lddy, dldj, dx2, lddx, dparams, lddp, rows, d2, s); break;
        case 8: spline_bwd_k<8>(x2, ldx, params, ldp, dy2, lddy, dldj, dx2, lddx, dparams, lddp, rows, d2, s); break;
        case 16: spline_bwd_k<16>(x2, ldx, params, ldp, dy2, lddy, dldj, dx2, lddx, dparams, lddp, rows, d2, s); break;
        default: throw Error(FC_ERR_UNSUPPORTED, "fc_train_rqspline_bwd_f32: num_bins must be 4, 8 or 16");
    }
    FC_API_END
}

int fc_train_layernorm_fwd_f32(const float* x, int32_t ldx, const float* gamma, const float* beta, float* y, int32_t ldy, float* stats,
                               int32_t rows, int32_t width, float eps, void* stream) {
    FC_API_BEGIN
    if (!x || !gamma || !beta || !y || !stats || rows < 1 || width < 1 || ldx < width || ldy < round_up(width, 32))
        throw Error(FC_ERR_INVALID, "fc_train_layernorm_fwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::layernorm_train_fwd_kernel", 0.0, (double)rows * width * 8.0, s);
    hipLaunchKernelGGL(layernorm_train_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, stats, rows, width,
                       round_up(width, 32), eps);
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_layernorm_bwd_f32(const float* x, int32_t ldx, const float* gamma, const float* dy, int32_t lddy, const float* stats, float* dx,
                               int32_t lddx, float* dy_xhat, int32_t ldt, int32_t rows_pad, int32_t rows, int32_t width, void* stream) {
    FC_API_BEGIN
    if (!x || !gamma || !dy || !stats || !dx || !dy_xhat || rows < 1 || rows_pad < rows || width < 1 || ldx < width || lddy < width ||
        lddx < round_up(width, 32) || ldt < round_up(width, 32))
        throw Error(FC_ERR_INVALID, "fc_train_layernorm_bwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::layernorm_train_bwd_kernel", 0.0, (double)rows * width * 16.0, s);
    hipLaunchKernelGGL(layernorm_train_bwd_kernel, dim3((rows_pad + 3) / 4), dim3(256), 0, s, x, ldx, gamma, dy, lddy, stats, dx, lddx, dy_xhat, ldt,
                       rows, rows_pad, width, round_up(width, 32));
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_affine_fwd_f32(const float* x2, int32_t ldx, const float* st, int32_t ldst, float* y2, int32_t ldy, float* ldj, int32_t rows, int32_t d2,
                            int32_t scale_fn, void* stream) {
    FC_API_BEGIN
    if (!x2 || !st || !y2 || !ldj || rows < 1 || d2 < 1 || ldx < d2 || ldst < 2 * d2 || ldy < round_up(d2, 32)) throw Error(FC_ERR_INVALID, "fc_train_affine_fwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::affine_train_fwd_kernel", 0.0, (double)rows * d2 * 16.0, s);
    hipLaunchKernelGGL(affine_train_fwd_kernel, dim3(rows), dim3(256), 0, s, x2, ldx, st, ldst, y2, ldy, ldj, d2, round_up(d2, 32), scale_fn);
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_affine_bwd_f32(const float* x2, int32_t ldx, const float* st, int32_t ldst, const float* dy2, int32_t lddy, const float* dldj, float* dx2,
                            int32_t lddx, float* dst, int32_t lddst, int32_t rows, int32_t d2, int32_t scale_fn, void* stream) {
    FC_API_BEGIN
    if (!x2 || !st || !dy2 || !dldj || !dx2 || !dst || rows < 1 || d2 < 1 || ldx < d2 || ldst < 2 * d2 || lddy < d2 || lddx < round_up(d2, 32) ||
        lddst < round_up(2 * d2, 32))
        throw Error(FC_ERR_INVALID, "fc_train_affine_bwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::affine_train_bwd_kernel", 0.0, (double)rows * d2 * 28.0, s);
    hipLaunchKernelGGL(affine_train_bwd_kernel, dim3(rows), dim3(256), 0, s, x2, ldx, st, ldst, dy2, lddy, dldj, dx2, lddx, dst, lddst, d2, round_up(d2, 32),
                       round_up(2 * d2, 32), scale_fn);
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_gauss_fwd_f32(const float* p, int32_t ldp, const float* eps, float* z, int32_t ldz, float* ldj, int32_t rows, int32_t nz, float clamp,
                           void* stream) {
    FC_API_BEGIN
    if (!p || !eps || !z || !ldj || rows < 1 || nz < 1 || ldp < 2 * nz || ldz < round_up(nz, 32)) throw Error(FC_ERR_INVALID, "fc_train_gauss_fwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::gauss_train_fwd_kernel", 0.0, (double)rows * nz * 16.0, s);
    hipLaunchKernelGGL(gauss_train_fwd_kernel, dim3(rows), dim3(256), 0, s, p, ldp, eps, z, ldz, ldj, nz, round_up(nz, 32), clamp);
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_gauss_bwd_f32(const float* p, int32_t ldp, const float* eps, const float* dz, int32_t lddz, const float* dldj, float* dp, int32_t lddp,
                           int32_t rows, int32_t nz, float clamp, void* stream) {
    FC_API_BEGIN
    if (!p || !eps || !dz || !dldj || !dp || rows < 1 || nz < 1 || ldp < 2 * nz || lddz < nz || lddp < round_up(2 * nz, 32))
        throw Error(FC_ERR_INVALID, "fc_train_gauss_bwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::gauss_train_bwd_kernel", 0.0, (double)rows * nz * 24.0, s);
    hipLaunchKernelGGL(gauss_train_bwd_kernel, dim3(rows), dim3(256), 0, s, p, ldp, eps, dz, lddz, dldj, dp, lddp, nz, round_up(2 * nz, 32), clamp);
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_normlp_fwd_f32(const float* v, int32_t ldv, const float* p, int32_t ldp, float* out, int32_t rows, int32_t nz, float clamp, void* stream) {
    FC_API_BEGIN
    if (!v || !p || !out || rows < 1 || nz < 1 || ldv < nz || ldp < 2 * nz) throw Error(FC_ERR_INVALID, "fc_train_normlp_fwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::normlp_train_fwd_kernel", 0.0, (double)rows * nz * 12.0, s);
    hipLaunchKernelGGL(normlp_train_fwd_kernel, dim3(rows), dim3(256), 0, s, v, ldv, p, ldp, out, nz, clamp);
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_normlp_bwd_f32(const float* v, int32_t ldv, const float* p, int32_t ldp, const float* g, float* dv, int32_t lddv, float* dp, int32_t lddp,
                            int32_t rows, int32_t nz, float clamp, void* stream) {
    FC_API_BEGIN
    if (!v || !p || !g || !dv || !dp || rows < 1 || nz < 1 || ldv < nz || ldp < 2 * nz || lddv < round_up(nz, 32) || lddp < round_up(2 * nz, 32))
        throw Error(FC_ERR_INVALID, "fc_train_normlp_bwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::normlp_train_bwd_kernel", 0.0, (double)rows * nz * 24.0, s);
    hipLaunchKernelGGL(normlp_train_bwd_kernel, dim3(rows), dim3(256), 0, s, v, ldv, p, ldp, g, dv, lddv, dp, lddp, nz, round_up(nz, 32), round_up(2 * nz, 32),
                       clamp);
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_expm_fwd_f32(const float* x2, int32_t ldx, const float* o, int32_t ldo, const float* scal4, float* y2, int32_t ldy, float* ldj, int32_t rows,
                          int32_t d2, int32_t* status, void* stream) {
    FC_API_BEGIN
    if (!x2 || !o || !scal4 || !y2 || !ldj || !status || rows < 1 || d2 < 1 || d2 > EX_D || ldx < d2 || ldo < d2 * d2 + d2 || ldy < round_up(d2, 32))
        throw Error(FC_ERR_INVALID, "fc_train_expm_fwd_f32: bad argument (d2 <= 16)");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::expm_train_fwd_kernel", 0.0, 0.0, s);
    hipLaunchKernelGGL(expm_train_fwd_kernel, dim3((rows + 63) / 64), dim3(64), 0, s, x2, ldx, o, ldo, scal4, y2, ldy, ldj, rows, d2, round_up(d2, 32), (int*)status);
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_expm_bwd_f32(const float* x2, int32_t ldx, const float* o, int32_t ldo, const float* scal4, const float* dy2, int32_t lddy, const float* dldj,
                          float* dx2, int32_t lddx, float* dout, int32_t lddo, float* dscal, int32_t rows, int32_t d2, void* stream) {
    FC_API_BEGIN
    if (!x2 || !o || !scal4 || !dy2 || !dldj || !dx2 || !dout || !dscal || rows < 1 || d2 < 1 || d2 > EX_D || ldx < d2 || ldo < d2 * d2 + d2 || lddy < d2 ||
        lddx < round_up(d2, 32) || lddo < round_up(d2 * d2 + d2, 32))
        throw Error(FC_ERR_INVALID, "fc_train_expm_bwd_f32: bad argument (d2 <= 16)");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::expm_train_bwd_kernel", 0.0, 0.0, s);
    hipLaunchKernelGGL(expm_train_bwd_kernel, dim3((rows + 63) / 64), dim3(64), 0, s, x2, ldx, o, ldo, scal4, dy2, lddy, dldj, dx2, lddx, dout, lddo, dscal, rows,
                       d2, round_up(d2, 32), round_up(d2 * d2 + d2, 32));
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_base_fwd_f32(const float* x, int32_t ldx, float* out, int32_t rows, int32_t width, void* stream) {
    FC_API_BEGIN
    if (!x || !out || rows < 1 || width < 1 || ldx < width) throw Error(FC_ERR_INVALID, "fc_train_base_fwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::base_train_fwd_kernel", 0.0, (double)rows * width * 4.0, s);
    hipLaunchKernelGGL(base_train_fwd_kernel, dim3(rows), dim3(256), 0, s, x, ldx, out, width);
    FC_HIP(hipGetLastError());
    FC_API_END
}

int fc_train_base_bwd_f32(const float* x, int32_t ldx, const float* g, float* dx, int32_t lddx, int32_t rows, int32_t width, void* stream) {
    FC_API_BEGIN
    if (!x || !g || !dx || rows < 1 || width < 1 || ldx < width || lddx < round_up(width, 32)) throw Error(FC_ERR_INVALID, "fc_train_base_bwd_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::base_train_bwd_kernel", 0.0, (double)rows * width * 8.0, s);
    hipLaunchKernelGGL(base_train_bwd_kernel, dim3(rows), dim3(256), 0, s, x, ldx, g, dx, lddx, width, round_up(width, 32));
    FC_HIP(hipGetLastError());
    FC_API_END
}

}  // extern "C"
