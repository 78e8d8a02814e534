// Kernels of the PAConv (PointNet++ SSG) context embedder that are not GEMMs: native replacements of the reference's six
// pointops_cuda kernels plus the PAConv-specific glue.  All channels-last, one wave per query point, no atomics.
//
//   fps_kernel            lib/pointops/src/sampling/sampling_cuda_kernel.cu:58-168   (furthestsampling_cuda)
//   gather_xyz_kernel     .../sampling_cuda_kernel.cu:6-20                           (gathering_forward_cuda)
//   knn_xyz_kernel        .../knnquery_heap/knnquery_heap_cuda_kernel.cu:53-89       (knnquery_heap_cuda, ascending order)
//   paconv_group_kernel   .../grouping/grouping_cuda_kernel.cu:60-74 + QueryAndGroup (functions/pointops.py:557-594)
//                         + the first PAConv layer's cat(f - f_centre, f)            (model/pointnet2/paconv.py:129-133)
//   scorenet_kernel       model/pointnet2/paconv.py:31-54 (3 -> 16 -> 8, BN folded, ReLU, softmax over the 8 kernels)
//   score_reduce_kernel   util/paconv_util.py:52-56 (assign_score) + BN + ReLU + either the next layer's cat(.) or the max over K
//   three_nn_interp_kernel .../interpolation/interpolation_cuda_kernel.cu:134-195 + PointNet2FPModule.forward
//                         (model/pointnet2/pointnet2_paconv_modules.py:224-236): 3-NN, inverse-distance weights, interpolation, skip concat
#include "common.h"

namespace fc {

// Squared distance as the reference's kernels compute it: lib/pointops/setup.py:32-33 builds them with `nvcc -O2`, whose default -fmad=true
// contracts `(dx)*(dx) + (dy)*(dy) + (dz)*(dz)` (sampling_cuda_kernel.cu:93, knnquery_heap_cuda_kernel.cu:77, interpolation_cuda_kernel.cu:155)
// into one multiply and two fused multiply-adds.  This file is compiled with -ffp-contract=off: the fusions are explicit, and identical to
// oracle/pointops_oracle.c's.  (Round 3 rounded every square separately: FPS picks, the 32-NN boundary and 3-NN weights could differ from
// the kernels' at near-ties.)
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// ---------------------------------------------------------------- furthest point sampling
// One workgroup (1024 threads) per scene; xyz and the running min-distance live in LDS.  The arg-max reproduces the CUDA
// kernel's tie rule: virtual thread t (t < T = opt_n_threads(n)) scans k = t, t+T, ... keeping its FIRST maximum; among threads the
// kernel's shared-memory tree (sampling_cuda_kernel.cu:47-53, 100-160: levels s = T/2 ... 1, slot t takes slot t + s only when STRICTLY
// greater) decides.  Two tied threads meet at the level of their LOWEST differing bit and the one whose bit is 0 stays: the winner is the
// thread with the smallest BIT-REVERSED id -- not the lowest id (round 3's reading; the literal simulation in oracle/pointops_oracle.c
// showed the difference on a lattice: after picks {0, 255} of an 8 x 8 x 4 grid the maxima sit at threads 29, 30, 225, 226 and the kernel
// takes 226).  (value, bit-reversed id) is a total order, so any reduction order finds that winner.
// BIG (more than 8192 points per scene: the LDS image would not fit): coordinates are read from the input rows and the running
// min-distance lives in a caller-provided global scratch [B][n]; both stay L2-resident over the m sweeps (same arithmetic, same ties).
template <bool BIG>
__global__ __launch_bounds__(1024) void fps_kernel(const float* __restrict__ xyz, int ld, int32_t* __restrict__ idx, int n, int m, int T,
                                                   float* __restrict__ scratch) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = xyz + (size_t)b * n * ld;
    const float* sx = BIG ? src : sm;                       // [n][3] (pitch 3 in LDS, ld in global)
    const int pitch = BIG ? ld : 3;
    float* st = BIG ? scratch + (size_t)b * n : sm + 3 * n; // [n]
    __shared__ float red_v[16];
    __shared__ int red_i[16];
    __shared__ int red_t[16];
    __shared__ int s_old;
    for (int k = tid; k < n; k += 1024) {
        if (!BIG) { sm[3 * k] = src[(size_t)k * ld]; sm[3 * k + 1] = src[(size_t)k * ld + 1]; sm[3 * k + 2] = src[(size_t)k * ld + 2]; }
        st[k] = 1e10f;
    }
    if (tid == 0) { idx[(size_t)b * m] = 0; s_old = 0; }
    __syncthreads();
    for (int j = 1; j < m; ++j) {
        const int old = s_old;
        const float x1 = sx[(size_t)pitch * old], y1 = sx[(size_t)pitch * old + 1], z1 = sx[(size_t)pitch * old + 2];
        float best = -1.f;
        int besti = 0;
        if (tid < T)
            for (int k = tid; k < n; k += T) {
                const float d = sqdist3(sx[(size_t)pitch * k], sx[(size_t)pitch * k + 1], sx[(size_t)pitch * k + 2], x1, y1, z1);
                const float d2 = fminf(d, st[k]);
                st[k] = d2;
                if (d2 > best) { best = d2; besti = k; }
            }
        unsigned bt = __brev((unsigned)tid);                // owning thread, bit-reversed: the CUDA tree's tie order (above)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(besti, off, 64);
            const unsigned ot = (unsigned)__shfl_xor((int)bt, off, 64);
            if (ov > best || (ov == best && ot < bt)) { best = ov; besti = oi; bt = ot; }
        }
        if (lane == 0) { red_v[wave] = best; red_i[wave] = besti; red_t[wave] = (int)bt; }
        __syncthreads();
        if (wave == 0) {
            float v = lane < 16 ? red_v[lane] : -2.f;
            int i = lane < 16 ? red_i[lane] : 0;
            unsigned w2 = lane < 16 ? (unsigned)red_t[lane] : 0xffffffffu;
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                const float ov = __shfl_xor(v, off, 64);
                const int oi = __shfl_xor(i, off, 64);
                const unsigned ow = (unsigned)__shfl_xor((int)w2, off, 64);
                if (ov > v || (ov == v && ow < w2)) { v = ov; i = oi; w2 = ow; }
            }
            if (lane == 0) { s_old = i; idx[(size_t)b * m + j] = i; }
        }
        __syncthreads();
    }
}
static int opt_n_threads_host(int n) {
    int p = 0;
    while ((2 << p) <= n) ++p;
    int t = 1 << p;
    return t > 1024 ? 1024 : (t < 1 ? 1 : t);
}
size_t fps_scratch_floats(int B, int n) { return n > 8192 ? (size_t)B * n : 0; }
void launch_fps(const float* xyz, int ld, int32_t* idx, int B, int n, int m, float* scratch, hipStream_t s) {
    if (m <= 0) return;
    if (m > n) throw Error(FC_ERR_INVALID, "furthest point sampling: m > n");
    ProfScope ps("fc::fps_kernel", 0.0, 4.0 * B * (3.0 * n + m), s);
    if (n > 8192) {
        if (!scratch) throw Error(FC_ERR_WORKSPACE, "furthest point sampling: more than 8192 points per scene need the [B][n] min-distance scratch (fps_scratch_floats)");
        hipLaunchKernelGGL(fps_kernel<true>, dim3(B), dim3(1024), 0, s, xyz, ld, idx, n, m, opt_n_threads_host(n), scratch);
    } else {
        const size_t lds = (size_t)n * 4 * sizeof(float);
        static bool attr_done = false;
        if (!attr_done) {
            FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16));
            attr_done = true;
        }
        hipLaunchKernelGGL(fps_kernel<false>, dim3(B), dim3(1024), lds, s, xyz, ld, idx, n, m, opt_n_threads_host(n), (float*)nullptr);
    }
    FC_HIP(hipGetLastError());
}

// dst[b, j, 0:3] = src[b, idx[b, j], 0:3] ; dst pitch 4 (last component 0)
__global__ void gather_xyz_kernel(const float* __restrict__ src, int ld, const int32_t* __restrict__ idx, float* dst, int n, int m, int total) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int b = t / m;
    const float* p = src + ((size_t)b * n + idx[t]) * ld;
    float* d = dst + (size_t)t * 4;
    d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = 0.f;
}
void launch_gather_xyz(const float* src, int ld, const int32_t* idx, float* dst, int B, int n, int m, hipStream_t s) {
    const int total = B * m;
    if (total <= 0) return;
    hipLaunchKernelGGL(gather_xyz_kernel, dim3((total + 255) / 256), dim3(256), 0, s, src, ld, idx, dst, n, m, total);
    FC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- k nearest neighbours in xyz, ascending distance
// One wave per query.  Each lane keeps the best candidates of its strided subset in a small sorted register list, then the
// wave extracts the global k best by k rounds of (distance, index)-lexicographic arg-min: the result is exactly the first k
// entries of a stable ascending sort, i.e. the heap kernel's output whenever distances are distinct.  n < k: the tail keeps
// index 0 like the reference's untouched heap slots.
// EQUAL distances (lattice clouds, duplicated points): the reference's max-heap (knnquery_heap_cuda_kernel.cu:21-49: strict `d2 < root`
// insertion, reheap, heap sort) returns them in the HEAP's order, and which of several entries tied at the k-th distance survives an
// eviction is the heap's choice too -- neither is an index order.  The wave therefore extracts one candidate more than k, and when any two
// neighbouring distances among those k + 1 are equal its lane 0 re-runs the query through the kernel's own heap, literally (LDS arrays):
// same set, same order as the reference on every input.  Clouds without exact ties never take that path.
template <int LK>   // per-lane list length
__global__ __launch_bounds__(256) void knn_xyz_kernel(const float* __restrict__ xyz, int ld, const float* __restrict__ qxyz, int32_t* __restrict__ out,
                                                      int n, int m, int k, int total) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= total) return;
    const int b = q / m;
    const float* src = xyz + (size_t)b * n * ld;
    const float qx = qxyz[(size_t)q * 4], qy = qxyz[(size_t)q * 4 + 1], qz = qxyz[(size_t)q * 4 + 2];
    float bd[LK];
    int bi[LK];
#pragma unroll
    for (int t = 0; t < LK; ++t) { bd[t] = INFINITY; bi[t] = 0x7fffffff; }
    for (int c = lane; c < n; c += 64) {
        const float d = sqdist3(qx, qy, qz, src[(size_t)c * ld], src[(size_t)c * ld + 1], src[(size_t)c * ld + 2]);
        if (d < bd[LK - 1]) {                              // insert into the ascending list (indices arrive ascending: ties keep the earlier)
            float cd = d;
            int ci = c;
#pragma unroll
            for (int t = 0; t < LK; ++t) {
                if (cd < bd[t]) { const float td = bd[t]; const int ti = bi[t]; bd[t] = cd; bi[t] = ci; cd = td; ci = ti; }
            }
        }
    }
    int32_t* o = out + (size_t)q * k;
    const int kk = k < n ? k : n;
    const int kx = kk < n ? kk + 1 : kk;                   // one more than k: a tie across the k-th boundary counts
    bool tie = false;
    float prev = -1.f;
    for (int r = 0; r < kx; ++r) {
        float v = bd[0];
        int i = bi[0];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(v, off, 64);
            const int oi = __shfl_xor(i, off, 64);
            if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
        }
        if (bi[0] == i) {                                  // the winning lane pops its head
#pragma unroll
            for (int t = 0; t + 1 < LK; ++t) { bd[t] = bd[t + 1]; bi[t] = bi[t + 1]; }
            bd[LK - 1] = INFINITY; bi[LK - 1] = 0x7fffffff;
        }
        tie = tie || v == prev;
        prev = v;
        if (lane == 0 && r < kk) o[r] = i;
    }
    if (lane >= kk && lane < k) o[lane] = 0;
    if (tie) {                                             // (wave-uniform) the reference's heap, literally, by lane 0
        __shared__ float hd_all[4][32];
        __shared__ int hi_all[4][32];
        if (lane == 0) {
            float* hd = hd_all[threadIdx.x >> 6];
            int* hi = hi_all[threadIdx.x >> 6];
            auto reheap = [&](int kh) {
                int root = 0, child = 1;
                while (child < kh) {
                    if (child + 1 < kh && hd[child + 1] > hd[child]) child++;
                    if (hd[root] > hd[child]) return;
                    const float td = hd[root]; hd[root] = hd[child]; hd[child] = td;
                    const int ti = hi[root]; hi[root] = hi[child]; hi[child] = ti;
                    root = child;
                    child = root * 2 + 1;
                }
            };
            for (int t = 0; t < k; ++t) { hd[t] = 1e10f; hi[t] = 0; }
            for (int c = 0; c < n; ++c) {
                const float d2 = sqdist3(qx, qy, qz, src[(size_t)c * ld], src[(size_t)c * ld + 1], src[(size_t)c * ld + 2]);
                if (d2 < hd[0]) { hd[0] = d2; hi[0] = c; reheap(k); }
            }
            for (int t = k - 1; t > 0; --t) {
                const float td = hd[0]; hd[0] = hd[t]; hd[t] = td;
                const int ti = hi[0]; hi[0] = hi[t]; hi[t] = ti;
                reheap(t);
            }
            for (int t = 0; t < k; ++t) o[t] = hi[t];
        }
    }
}
void launch_knn_xyz(const float* xyz, int ld, const float* qxyz, int32_t* out, int B, int n, int m, int k, hipStream_t s) {
    const int total = B * m;
    if (total <= 0) return;
    if (k > 32) throw Error(FC_ERR_UNSUPPORTED, "xyz k-NN: nsample > 32");
    // a lane sees ceil(n/64) candidates; its list must be able to hold all of them or k of them, whichever is smaller
    const int need = (n + 63) / 64 < k ? (n + 63) / 64 : k;
    ProfScope ps("fc::knn_xyz_kernel", 8.0 * total * (double)n, 4.0 * total * k, s);
    if (need <= 8) hipLaunchKernelGGL(knn_xyz_kernel<8>, dim3((total + 3) / 4), dim3(256), 0, s, xyz, ld, qxyz, out, n, m, k, total);
    else if (need <= 16) hipLaunchKernelGGL(knn_xyz_kernel<16>, dim3((total + 3) / 4), dim3(256), 0, s, xyz, ld, qxyz, out, n, m, k, total);
    else hipLaunchKernelGGL(knn_xyz_kernel<32>, dim3((total + 3) / 4), dim3(256), 0, s, xyz, ld, qxyz, out, n, m, k, total);
    FC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- grouping + first PAConv layer input
// Per query q (one wave) and neighbour e: f_e = [xyz_src[idx_e] - new_xyz[q] (3) | F_src[idx_e] (C)]  (QueryAndGroup, use_xyz)
//   E[q*K + e] = [f_e - f_0 (C+3) | f_e (C+3) | 0-pad]   (PAConv kernel_input 'neighbor': the centre is neighbour 0)
//   gdiff[q*K + e] = xyz_src[idx_e] - xyz_src[idx_0]      (ScoreNet input, score_input 'identity')
__global__ __launch_bounds__(256) void paconv_group_kernel(const float* __restrict__ xyz, int ldxyz, const float* __restrict__ feat, int ldf, int C,
                                                           const float* __restrict__ qxyz, const int32_t* __restrict__ nidx, float* E, int ldE,
                                                           float* gdiff, int n, int m, int K, int total) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= total) return;
    const int b = q / m;
    const size_t base = (size_t)b * n;
    const int32_t* ip = nidx + (size_t)q * K;
    const int i0 = ip[0];
    const float cx = qxyz[(size_t)q * 4], cy = qxyz[(size_t)q * 4 + 1], cz = qxyz[(size_t)q * 4 + 2];
    const int Cin = C + 3;
    for (int t = lane; t < K * ldE; t += 64) {
        const int e = t / ldE, c = t - e * ldE;
        const int ie = ip[e];
        float v = 0.f;
        if (c < 2 * Cin) {
            const int ch = c < Cin ? c : c - Cin;
            float fe, f0;
            if (ch < 3) {
                const float ctr = ch == 0 ? cx : (ch == 1 ? cy : cz);
                fe = xyz[(base + ie) * ldxyz + ch] - ctr;
                f0 = xyz[(base + i0) * ldxyz + ch] - ctr;
            } else {
                fe = feat[(base + ie) * ldf + ch - 3];
                f0 = feat[(base + i0) * ldf + ch - 3];
            }
            v = c < Cin ? fe - f0 : fe;
        }
        E[((size_t)q * K + e) * ldE + c] = v;
    }
    for (int t = lane; t < K * 4; t += 64) {
        const int e = t >> 2, c = t & 3;
        gdiff[((size_t)q * K + e) * 4 + c] = c < 3 ? xyz[(base + ip[e]) * ldxyz + c] - xyz[(base + i0) * ldxyz + c] : 0.f;
    }
}
void launch_paconv_group(const float* xyz, int ldxyz, const float* feat, int ldf, int C, const float* qxyz, const int32_t* nidx, float* E, int ldE,
                         float* gdiff, int B, int n, int m, int K, hipStream_t s) {
    const int total = B * m;
    if (total <= 0) return;
    ProfScope ps("fc::paconv_group_kernel", 0.0, 4.0 * total * K * (ldE + 4.0 + C), s);
    hipLaunchKernelGGL(paconv_group_kernel, dim3((total + 3) / 4), dim3(256), 0, s, xyz, ldxyz, feat, ldf, C, qxyz, nidx, E, ldE, gdiff, n, m, K, total);
    FC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- ScoreNet: one thread per edge
// w = [W0' (16x3, BN scale folded) | b0' (16, BN shift) | W1 (8x16) | b1 (8)] = 200 floats
__global__ void scorenet_kernel(const float* __restrict__ gdiff, const float* __restrict__ w, float* scores, int edges) {
    __shared__ float sw[200];
    if (threadIdx.x < 200) sw[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= edges) return;
    const float4 g = *reinterpret_cast<const float4*>(gdiff + (size_t)e * 4);
    float h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = ((sw[3 * i] * g.x + sw[3 * i + 1] * g.y) + sw[3 * i + 2] * g.z) + sw[48 + i];
        h[i] = v > 0.f ? v : 0.f;
    }
    float sc[8], mx = -INFINITY;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float v = sw[64 + 128 + o];
#pragma unroll
        for (int i = 0; i < 16; ++i) v = fmaf(sw[64 + 16 * o + i], h[i], v);
        sc[o] = v;
        mx = fmaxf(mx, v);
    }
    float sum = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o) { sc[o] = expf(sc[o] - mx); sum += sc[o]; }
    float* out = scores + (size_t)e * 8;
#pragma unroll
    for (int o = 0; o < 8; ++o) out[o] = sc[o] / sum;
}
void launch_scorenet(const float* gdiff, const float* w, float* scores, int edges, hipStream_t s) {
    if (edges <= 0) return;
    ProfScope ps("fc::scorenet_kernel", 0.0, 48.0 * edges, s);
    hipLaunchKernelGGL(scorenet_kernel, dim3((edges + 255) / 256), dim3(256), 0, s, gdiff, w, scores, edges);
    FC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- score-weighted kernel assembly + BN + ReLU
// O[e, o] = ReLU( bn_s[o] * sum_m score[e, m] * G[e, m*Cout + o] + bn_t[o] )        (assign_score + BN2d + ReLU)
//   mode 0: next layer input  E[e] = [O_e - O_0 | O_e | 0-pad]      mode 1: pooled[q, o] = max_e O[e, o]
__global__ __launch_bounds__(256) void score_reduce_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ scores,
                                                           const float* __restrict__ bn_s, const float* __restrict__ bn_t, int Cout, int K,
                                                           float* dst, int ldd, int dst_col0, int mode, int total) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= total) return;
    const float* sq = scores + (size_t)q * K * 8;
    for (int o = lane; o < Cout; o += 64) {
        const float bs = bn_s[o], bt = bn_t[o];
        float o0 = 0.f, mx = -INFINITY;
        for (int e = 0; e < K; ++e) {
            const float* g = G + ((size_t)q * K + e) * ldg + o;
            float acc = 0.f;
#pragma unroll
            for (int mm = 0; mm < 8; ++mm) acc = fmaf(sq[e * 8 + mm], g[(size_t)mm * Cout], acc);
            float v = acc * bs + bt;
            v = v > 0.f ? v : 0.f;
            if (mode == 0) {
                if (e == 0) o0 = v;
                float* d = dst + ((size_t)q * K + e) * ldd;
                d[o] = v - o0;
                d[Cout + o] = v;
            } else {
                mx = fmaxf(mx, v);
            }
        }
        if (mode == 1) dst[(size_t)q * ldd + dst_col0 + o] = mx;
    }
    if (mode == 0)                                          // zero the pad columns of the next layer's input
        for (int t = lane; t < K * (ldd - 2 * Cout); t += 64) {
            const int e = t / (ldd - 2 * Cout), c = t - e * (ldd - 2 * Cout);
            dst[((size_t)q * K + e) * ldd + 2 * Cout + c] = 0.f;
        }
}
void launch_score_reduce(const float* G, int ldg, const float* scores, const float* bn_s, const float* bn_t, int Cout, int K, float* dst, int ldd,
                         int dst_col0, int mode, int total_queries, hipStream_t s) {
    if (total_queries <= 0) return;
    ProfScope ps("fc::score_reduce_kernel", 0.0, 4.0 * total_queries * K * (8.0 * Cout + 8.0 + (mode == 0 ? 2.0 * Cout : 0.0)), s);
    hipLaunchKernelGGL(score_reduce_kernel, dim3((total_queries + 3) / 4), dim3(256), 0, s, G, ldg, scores, bn_s, bn_t, Cout, K, dst, ldd, dst_col0,
                       mode, total_queries);
    FC_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- feature propagation front end
// Per unknown point (one wave): the 3 nearest known points (strict '<' updates: the earliest index wins ties; bests kept as the
// float distance like the CUDA kernel's double copies of float values), weights (1/(sqrt(d2)+1e-8)) normalised, then
//   X[p] = [ sum_i w_i Fk[idx_i, 0:C2] | Fu[p, 0:C1] | 0-pad ]
__global__ __launch_bounds__(256) void three_nn_interp_kernel(const float* __restrict__ uxyz, int ldu, const float* __restrict__ kxyz, int ldk,
                                                              const float* __restrict__ Fk, int ldfk, int C2, const float* __restrict__ Fu, int ldfu,
                                                              int C1, float* X, int ldX, int nu, int mk, int total) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= total) return;
    const int b = p / nu;
    const float ux = uxyz[(size_t)p * ldu], uy = uxyz[(size_t)p * ldu + 1], uz = uxyz[(size_t)p * ldu + 2];
    const float* ks = kxyz + (size_t)b * mk * ldk;
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0x7fffffff, i1 = 0x7fffffff, i2 = 0x7fffffff;
    for (int c = lane; c < mk; c += 64) {
        const float d = sqdist3(ux, uy, uz, ks[(size_t)c * ldk], ks[(size_t)c * ldk + 1], ks[(size_t)c * ldk + 2]);
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
        bi[r] = i == 0x7fffffff ? 0 : i;                   // fewer than 3 known points: distance inf, index 0 (weight 0)
    }
    float w[3], ws = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) { w[r] = 1.0f / (sqrtf(bd[r]) + 1e-8f); ws += w[r]; }
#pragma unroll
    for (int r = 0; r < 3; ++r) w[r] = w[r] / ws;
    const float* f0 = Fk + ((size_t)b * mk + bi[0]) * ldfk;
    const float* f1 = Fk + ((size_t)b * mk + bi[1]) * ldfk;
    const float* f2 = Fk + ((size_t)b * mk + bi[2]) * ldfk;
    float* x = X + (size_t)p * ldX;
    for (int c = lane; c < C2; c += 64) x[c] = (w[0] * f0[c] + w[1] * f1[c]) + w[2] * f2[c];
    for (int c = lane; c < C1; c += 64) x[C2 + c] = Fu[(size_t)p * ldfu + c];
    for (int c = C2 + C1 + lane; c < ldX; c += 64) x[c] = 0.f;
}
void launch_three_nn_interp(const float* uxyz, int ldu, const float* kxyz, int ldk, const float* Fk, int ldfk, int C2, const float* Fu, int ldfu, int C1,
                            float* X, int ldX, int B, int nu, int mk, hipStream_t s) {
    const int total = B * nu;
    if (total <= 0) return;
    if (mk < 1) throw Error(FC_ERR_INVALID, "feature propagation: no known points");
    ProfScope ps("fc::three_nn_interp_kernel", 8.0 * total * (double)mk, 4.0 * total * (ldX + 3.0 * C2 + C1), s);
    hipLaunchKernelGGL(three_nn_interp_kernel, dim3((total + 3) / 4), dim3(256), 0, s, uxyz, ldu, kxyz, ldk, Fk, ldfk, C2, Fu, ldfu, C1, X, ldX, nu, mk, total);
    FC_HIP(hipGetLastError());
}

}  // namespace fc
