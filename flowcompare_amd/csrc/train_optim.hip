// Optimiser step of the training loop (train.py:112-120: clip_grad_norm_ -> Adam.step) on the gradient reducer's flat buffers
// (flowcompare_amd/shard.py): the gradients of a bucket are one contiguous fp32 array, the Adam moments mirror that layout, and the
// parameters stay the reference's separate nn.Parameter tensors, reached through a device table of pointers.  One launch per bucket
// instead of torch's foreach kernels over ~3000 tensors; HBM-bound (reads g, m, v, p; writes m, v, p: 28 bytes per weight).
//   sqnorm_kernel / sqnorm_reduce_kernel   sum of squares of a flat gradient buffer, fp64 partials, fixed order (bit-reproducible)
//   adam_kernel                            torch.optim.Adam semantics (no amsgrad; L2 weight decay added to the gradient), with the
//                                          clip coefficient read from device memory so that no host synchronisation sits between
//                                          the norm and the update
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "common.h"

namespace fc {

constexpr int OPT_CHUNK = 4096;      // elements per workgroup

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long n, double* __restrict__ part) {
    __shared__ double red[4];
    const long base = (long)blockIdx.x * OPT_CHUNK;
    double s = 0.0;
    for (int i = threadIdx.x; i < OPT_CHUNK; i += 256) {
        const long t = base + i;
        if (t < n) { const double v = (double)g[t]; s += v * v; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// out[slot] = sum of the partials, summed by ONE thread block in a fixed order
__global__ __launch_bounds__(256) void sqnorm_reduce_kernel(const double* __restrict__ part, int nparts, double* __restrict__ out, int slot) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[slot] = red[0];
}

struct AdamArgs {
    float* const* p;            // [T] parameter tensors
    const long* off;            // [T + 1] offset of each tensor inside the flat buffers
    const int* chunk_tensor;    // [chunks] tensor a chunk belongs to
    const long* chunk_off;      // [chunks] offset of the chunk inside its tensor
    const float* g;
    float* m;
    float* v;
    const float* coef;          // device scalar: gradients are multiplied by it (clip coefficient), or null
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt;
};
__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
    const int t = a.chunk_tensor[blockIdx.x];
    const long o = a.chunk_off[blockIdx.x];
    const long n = a.off[t + 1] - a.off[t];
    float* p = a.p[t];
    const float* g = a.g + a.off[t];
    float* m = a.m + a.off[t];
    float* v = a.v + a.off[t];
    const float c = a.coef ? *a.coef : 1.0f;
    const float step = a.lr / a.bc1;
    for (int i = threadIdx.x; i < OPT_CHUNK; i += 256) {
        const long k = o + i;
        if (k >= n) break;
        float gk = g[k] * c;
        const float pk = p[k];
        if (a.wd != 0.f) gk = fmaf(a.wd, pk, gk);
        const float mk = a.beta1 * m[k] + (1.0f - a.beta1) * gk;
        const float vk = a.beta2 * v[k] + (1.0f - a.beta2) * gk * gk;
        m[k] = mk;
        v[k] = vk;
        const float denom = sqrtf(vk) / a.bc2_sqrt + a.eps;
        p[k] = pk - step * (mk / denom);
    }
}

}  // namespace fc

using namespace fc;

extern "C" {

size_t fc_train_sqnorm_ws_bytes(int64_t n) { return (size_t)((n + OPT_CHUNK - 1) / OPT_CHUNK) * sizeof(double) + 256; }
/* out[slot] (device, fp64) = sum of squares of g[0..n): the global gradient norm of clip_grad_norm_ is sqrt of the sum over the buckets */
int fc_train_sqnorm_f32(const float* g, int64_t n, double* out, int32_t slot, void* ws, size_t ws_bytes, void* stream) {
    FC_API_BEGIN
    if (!g || !out || n < 1 || slot < 0 || !ws || ws_bytes < fc_train_sqnorm_ws_bytes(n) || ((uintptr_t)ws & 7)) throw Error(FC_ERR_INVALID, "fc_train_sqnorm_f32: bad argument / workspace");
    hipStream_t s = (hipStream_t)stream;
    const int nparts = (int)((n + OPT_CHUNK - 1) / OPT_CHUNK);
    ProfScope ps("fc::sqnorm_kernel", 0.0, (double)n * 4.0, s);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(nparts), dim3(256), 0, s, g, (long)n, (double*)ws);
    FC_HIP(hipGetLastError());
    hipLaunchKernelGGL(sqnorm_reduce_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, nparts, out, slot);
    FC_HIP(hipGetLastError());
    FC_API_END
}
/* One Adam step (torch.optim.Adam semantics, amsgrad = False) for the T parameter tensors of one bucket: params [T] device array of
 * pointers, offsets [T + 1] (int64) into the flat g / m / v arrays, chunk_tensor / chunk_off [n_chunks] = the 4096-element chunks of
 * every tensor (built once by the caller), coef = device scalar the gradients are multiplied by (clip coefficient) or NULL, step = t >= 1. */
int fc_train_adam_f32(float* const* params, const int64_t* offsets, const int32_t* chunk_tensor, const int64_t* chunk_off, int32_t n_chunks, const float* g,
                      float* m, float* v, const float* coef, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream) {
    FC_API_BEGIN
    if (!params || !offsets || !chunk_tensor || !chunk_off || n_chunks < 1 || !g || !m || !v || step < 1) throw Error(FC_ERR_INVALID, "fc_train_adam_f32: bad argument");
    AdamArgs a{params, (const long*)offsets, chunk_tensor, (const long*)chunk_off, g, m, v, coef, lr, beta1, beta2, eps, weight_decay,
               (float)(1.0 - std::pow((double)beta1, (double)step)), (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step))};
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("fc::adam_kernel", 0.0, (double)n_chunks * OPT_CHUNK * 28.0, s);
    hipLaunchKernelGGL(adam_kernel, dim3(n_chunks), dim3(256), 0, s, a);
    FC_HIP(hipGetLastError());
    FC_API_END
}

}  // extern "C"
