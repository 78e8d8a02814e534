// The steps either side of the log-prob path:
//   * input staging of the voxel loader (dataloaders/ams_voxel_loader.py:298-307,357-358): farthest point subsampling over the
//     full feature rows (torch_cluster.fps, random_start=False) and the joint unit-sphere normalisation (utils.py:259-280);
//   * change-map post-processing (test_flow.py:241-275): clamp_infs, per-scene mean/std threshold, min-max scaling.
// Small streaming / reduction kernels: one workgroup per scene, LDS tree reductions in a fixed order (bit-reproducible).
#include "common.h"

#include <cfloat>

namespace fc {

// ---------------------------------------------------------------- farthest point sampling over C-dimensional rows
// torch_cluster 1.5.9 fps (cpu/fps_cpu.cpp), random_start = False: out[0] = 0; dist = min(dist, |src - src[last]|^2) summed over
// ALL columns in column order; next = argmax(dist) with the FIRST maximum (lowest index) winning ties.
// The running min-distance lives in LDS up to 24576 points per cloud and in a caller-provided global scratch beyond.
template <bool LDS_DIST>
__global__ __launch_bounds__(1024) void fps_nd_kernel(const float* __restrict__ pts, int ld, int C, int64_t* __restrict__ idx, int n, int m,
                                                      float* __restrict__ dist_scratch) {
    extern __shared__ float sm[];
    __shared__ float red_v[16];
    __shared__ int red_i[16];
    __shared__ int s_last;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = pts + (size_t)b * n * ld;
    float* dist = LDS_DIST ? sm : dist_scratch + (size_t)b * n;
    for (int k = tid; k < n; k += 1024) dist[k] = INFINITY;
    if (tid == 0) { idx[(size_t)b * m] = 0; s_last = 0; }
    __syncthreads();
    for (int j = 1; j < m; ++j) {
        const int last = s_last;
        float ref[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) ref[c] = c < C ? src[(size_t)last * ld + c] : 0.f;
        float best = -1.f;
        int besti = 0x7fffffff;
        for (int k = tid; k < n; k += 1024) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < C) { const float t = src[(size_t)k * ld + c] - ref[c]; d += t * t; }
            const float d2 = fminf(d, dist[k]);
            dist[k] = d2;
            if (d2 > best) { best = d2; besti = k; }          // ascending k: keeps this thread's lowest index among equals
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(besti, off, 64);
            if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        if (lane == 0) { red_v[wave] = best; red_i[wave] = besti; }
        __syncthreads();
        if (wave == 0) {
            float v = lane < 16 ? red_v[lane] : -2.f;
            int i = lane < 16 ? red_i[lane] : 0x7fffffff;
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                const float ov = __shfl_xor(v, off, 64);
                const int oi = __shfl_xor(i, off, 64);
                if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
            }
            if (lane == 0) { s_last = i; idx[(size_t)b * m + j] = i; }
        }
        __syncthreads();
    }
}

void launch_fps_nd(const float* pts, int ld, int C, int64_t* idx, int B, int n, int m, float* dist_scratch, hipStream_t s) {
    if (m <= 0) return;
    if (C < 1 || C > 8 || ld < C) throw Error(FC_ERR_UNSUPPORTED, "fps: 1..8 feature columns supported");
    if (m > n) throw Error(FC_ERR_INVALID, "fps: more samples than points");
    ProfScope ps("fc::fps_nd_kernel", 0.0, 4.0 * B * ((double)n * C + m), s);
    if (n <= 24576) {
        static bool attr_done = false;
        if (!attr_done) {
            FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_nd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4));
            attr_done = true;
        }
        hipLaunchKernelGGL(fps_nd_kernel<true>, dim3(B), dim3(1024), (size_t)n * sizeof(float), s, pts, ld, C, idx, n, m, nullptr);
    } else {
        if (!dist_scratch) throw Error(FC_ERR_WORKSPACE, "fps: clouds above 24576 points need a [B*n] float scratch");
        hipLaunchKernelGGL(fps_nd_kernel<false>, dim3(B), dim3(1024), 0, s, pts, ld, C, idx, n, m, dist_scratch);
    }
    FC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- block reductions (fixed order: reproducible run to run)
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, red[w]);
    return t;
}
__device__ __forceinline__ float block_min(float v, float* red) { return -block_max(-v, red); }

// ---------------------------------------------------------------- joint unit-sphere normalisation (utils.py:259-280)
// joint = cat(points_0, points_1): xyz -= mean(xyz); xyz /= max |xyz|; other columns pass through.  One workgroup per pair.
__global__ __launch_bounds__(1024) void co_unit_sphere_kernel(const float* __restrict__ p0, int n0, const float* __restrict__ p1, int n1, int ld,
                                                              float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ inverse) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x, n = n0 + n1;
    const float* a = p0 + (size_t)b * n0 * ld;
    const float* c = p1 + (size_t)b * n1 * ld;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int k = tid; k < n; k += 1024) {
        const float* r = k < n0 ? a + (size_t)k * ld : c + (size_t)(k - n0) * ld;
        sx += r[0]; sy += r[1]; sz += r[2];
    }
    const float mx = block_sum(sx, red) / (float)n, my = block_sum(sy, red) / (float)n, mz = block_sum(sz, red) / (float)n;
    float far2 = 0.f;
    for (int k = tid; k < n; k += 1024) {
        const float* r = k < n0 ? a + (size_t)k * ld : c + (size_t)(k - n0) * ld;
        const float x = r[0] - mx, y = r[1] - my, z = r[2] - mz;
        far2 = fmaxf(far2, sqrtf((x * x + y * y) + z * z));          // torch.linalg.norm per row, then max
    }
    const float far = block_max(far2, red);
    for (int k = tid; k < n; k += 1024) {
        const float* r = k < n0 ? a + (size_t)k * ld : c + (size_t)(k - n0) * ld;
        float* w = k < n0 ? o0 + ((size_t)b * n0 + k) * ld : o1 + ((size_t)b * n1 + (k - n0)) * ld;
        w[0] = (r[0] - mx) / far; w[1] = (r[1] - my) / far; w[2] = (r[2] - mz) / far;
        for (int col = 3; col < ld; ++col) w[col] = r[col];
    }
    if (tid == 0) { inverse[4 * b] = far; inverse[4 * b + 1] = mx; inverse[4 * b + 2] = my; inverse[4 * b + 3] = mz; }
}

void launch_co_unit_sphere(const float* p0, int n0, const float* p1, int n1, int ld, float* o0, float* o1, float* inverse, int B, hipStream_t s) {
    if (B <= 0 || n0 < 0 || n1 < 0 || n0 + n1 < 1 || ld < 3) throw Error(FC_ERR_INVALID, "co_unit_sphere: bad shape");
    ProfScope ps("fc::co_unit_sphere_kernel", 0.0, 4.0 * B * (double)(n0 + n1) * ld * 2, s);
    hipLaunchKernelGGL(co_unit_sphere_kernel, dim3(B), dim3(1024), 0, s, p0, n0, p1, n1, ld, o0, o1, inverse);
    FC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- change map (test_flow.py:241-275)
// stats[0] = min over the non-inf entries (NaN never compares smaller), stats[1] = 1.0 when an inf was seen.
__device__ __forceinline__ void atomic_min_float(float* addr, float v) {
    // monotone map float -> int: non-negative floats compare as ints, negative ones in reverse
    if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__global__ __launch_bounds__(256) void inf_stats_kernel(const float* __restrict__ t, long n, float* __restrict__ stats) {
    __shared__ float red[4];
    float mn = INFINITY;
    bool inf = false;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) {
        const float v = t[k];
        if (isinf(v)) inf = true; else mn = fminf(mn, v);        // fminf drops NaN: torch's min would return NaN, and so does the
    }                                                            // row kernel below through its own NaN propagation
    const float bm = block_min(mn, red);
    if (threadIdx.x == 0 && bm < INFINITY) atomic_min_float(stats, bm);
    if (__syncthreads_or(inf) && threadIdx.x == 0) stats[1] = 1.0f;
}

__global__ void change_init_kernel(float* stats4, int* status) {
    if (threadIdx.x < 4) stats4[threadIdx.x] = (threadIdx.x & 1) ? 0.f : INFINITY;
    if (threadIdx.x == 0) *status = 0;
}

// One workgroup per scene.  lp10 / lp00 are clamped IN PLACE when their tensor held an inf (the reference's clamp_infs mutates
// its argument).  out = 1 - (lp10 - min)/(max - min) where lp10 < mean(lp00) - multiple * std(lp00) (unbiased) or < hard_cutoff,
// else 0.  status[0] is raised when a result is NaN / inf (the reference asserts is_valid).
__global__ __launch_bounds__(256) void change_map_kernel(float* __restrict__ lp10, int N, float* __restrict__ lp00, int N0, float* __restrict__ out,
                                                         const float* __restrict__ st10, const float* __restrict__ st00, float multiple,
                                                         float hard_cutoff, int use_cutoff, int* __restrict__ status) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float* r1 = lp10 + (size_t)b * N;
    float* r0 = lp00 + (size_t)b * N0;
    const bool c1 = st10[1] != 0.f, c0 = st00[1] != 0.f;
    const float m1 = st10[0], m0 = st00[0];
    float thr = hard_cutoff;
    if (!use_cutoff) {
        float s = 0.f;
        for (int k = tid; k < N0; k += 256) {
            float v = r0[k];
            if (c0 && isinf(v)) { v = m0; r0[k] = v; }
            s += v;
        }
        const float mean = block_sum(s, red) / (float)N0;
        float q = 0.f;
        for (int k = tid; k < N0; k += 256) { const float d = r0[k] - mean; q += d * d; }
        const float sd = sqrtf(block_sum(q, red) / (float)(N0 - 1));
        thr = mean - multiple * sd;
    } else if (c0) {
        for (int k = tid; k < N0; k += 256) if (isinf(r0[k])) r0[k] = m0;
    }
    float mx = -INFINITY, mn = INFINITY;
    bool nan = false;
    for (int k = tid; k < N; k += 256) {
        float v = r1[k];
        if (c1 && isinf(v)) { v = m1; r1[k] = v; }
        nan |= v != v;
        mx = fmaxf(mx, v); mn = fminf(mn, v);
    }
    mx = block_max(mx, red);
    mn = block_min(mn, red);
    bool bad = nan;
    for (int k = tid; k < N; k += 256) {
        const float v = r1[k];
        const float scaled = 1.0f - (v - mn) / (mx - mn);
        const float o = v < thr ? scaled : 0.0f;
        bad |= !(fabsf(o) <= FLT_MAX);
        out[(size_t)b * N + k] = o;
    }
    if (bad) atomicOr(status, 1);
}

__global__ __launch_bounds__(256) void clamp_infs_kernel(float* __restrict__ t, long n, const float* __restrict__ stats) {
    if (stats[1] == 0.f) return;
    const float m = stats[0];
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256)
        if (isinf(t[k])) t[k] = m;
}

// clamp_infs alone (test_flow.py:241-247); stats4 / status as for launch_change_map
void launch_clamp_infs(float* t, long n, float* stats4, int* status, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(change_init_kernel, dim3(1), dim3(64), 0, s, stats4, status);
    const unsigned g = (unsigned)std::min<long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(inf_stats_kernel, dim3(g), dim3(256), 0, s, t, n, stats4);
    hipLaunchKernelGGL(clamp_infs_kernel, dim3(g), dim3(256), 0, s, t, n, stats4);
    FC_HIP(hipGetLastError());
}

void launch_change_map(float* lp10, int N, float* lp00, int N0, float* out, int B, float multiple, float hard_cutoff, int use_cutoff,
                       float* stats4, int* status, hipStream_t s) {
    if (B <= 0 || N <= 0 || N0 <= 0) throw Error(FC_ERR_INVALID, "change map: empty input");
    hipLaunchKernelGGL(change_init_kernel, dim3(1), dim3(64), 0, s, stats4, status);
    ProfScope ps("fc::change_map_kernel", 0.0, 4.0 * B * (3.0 * N + 3.0 * N0), s);
    const long n1 = (long)B * N, n0 = (long)B * N0;
    hipLaunchKernelGGL(inf_stats_kernel, dim3((unsigned)std::min<long>((n1 + 255) / 256, 1024)), dim3(256), 0, s, lp10, n1, stats4);
    hipLaunchKernelGGL(inf_stats_kernel, dim3((unsigned)std::min<long>((n0 + 255) / 256, 1024)), dim3(256), 0, s, lp00, n0, stats4 + 2);
    hipLaunchKernelGGL(change_map_kernel, dim3(B), dim3(256), 0, s, lp10, N, lp00, N0, out, stats4, stats4 + 2, multiple, hard_cutoff, use_cutoff,
                       status);
    FC_HIP(hipGetLastError());
}

}  // namespace fc
