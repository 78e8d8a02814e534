/* CPU restatement of three pointops CUDA kernels of the reference's PAConv embedder.  TEST INFRASTRUCTURE ONLY: nothing under flowcompare_amd/
 * loads this; tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.  Parity status: "parity UNPINNED at reference level" -- the
 * kernels are CUDA-only, the reference ships no vectors for them and this image has no nvcc; what is restated is the SOURCE,
 * loop for loop, and the arithmetic nvcc's documented default generates for it.
 *
 * Paths below are relative to models/scene_seg_PAConv/lib/pointops/ in the reference.
 *   fps      src/sampling/sampling_cuda_kernel.cu:47-53 (__update), :58-168 (furthestsampling_cuda_kernel), launcher :170-, block size
 *            opt_n_threads(n) (src/cuda_utils.h:15-18); temp starts at 1e10 (functions/pointops.py FurthestSampling.forward)
 *   knn      src/knnquery_heap/knnquery_heap_cuda_kernel.cu:21-36 (reheap), :39-49 (heap_sort), :53-89 (kernel): a MAX-heap of nsample
 *            (distance, index) pairs filled by strict `d2 < root`, then heap-sorted ascending -- the ORDER among equal distances is the
 *            heap's, not an index order
 *   three_nn src/interpolation/interpolation_cuda_kernel.cu:134-176: three running minima kept in DOUBLE, strict `<` updates
 * Distances: setup.py:32-33 builds with `nvcc -O2`, i.e. nvcc's default -fmad=true, which contracts a multiply feeding an add into one fused
 * multiply-add.  `(dx)*(dx) + (dy)*(dy) + (dz)*(dz)` is therefore evaluated as fma(dz, dz, fma(dy, dy, dx * dx)): the first square rounded, the
 * other two fused (round 3's review prescribes this form; which of the first two squares nvcc leaves as the separate multiply cannot be
 * checked without the compiler -- the two choices differ in the last bit of a near-tie only).  Built with -ffp-contract=off: every fusion
 * below is explicit.
 *
 * Every function exists for float (the kernels' own type) and for double (the oracle's fp64 runs: same loops, same tie rules). */
#include <math.h>
#include <stdlib.h>
#include <stdint.h>

static int opt_n_threads(int work_size) {
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int t = 1 << pow_2;
    if (t > 1024) t = 1024;
    return t < 1 ? 1 : t;
}

#define DEFINE_POINTOPS(SUF, T, FMA)                                                                                          \
    static inline T sqdist_##SUF(T ax, T ay, T az, T bx, T by, T bz) {                                                        \
        const T dx = ax - bx, dy = ay - by, dz = az - bz;                                                                     \
        return FMA(dz, dz, FMA(dy, dy, dx * dx));                                                                             \
    }                                                                                                                         \
    /* xyz [b][n][3] -> idx [b][m]; literal simulation of the block: thread t scans k = t, t + T, ..., then the shared-memory tree */ \
    void fps_##SUF(int b, int n, int m, const T* xyz, int32_t* idx) {                                                         \
        if (m <= 0) return;                                                                                                   \
        const int TT = opt_n_threads(n);                                                                                      \
        T* temp = (T*)malloc(sizeof(T) * (size_t)n);                                                                          \
        T* dists = (T*)malloc(sizeof(T) * (size_t)TT);                                                                        \
        int* dists_i = (int*)malloc(sizeof(int) * (size_t)TT);                                                                \
        for (int bi = 0; bi < b; ++bi) {                                                                                      \
            const T* d = xyz + (size_t)bi * n * 3;                                                                            \
            int32_t* out = idx + (size_t)bi * m;                                                                              \
            for (int k = 0; k < n; ++k) temp[k] = (T)1e10;                                                                    \
            int old = 0;                                                                                                      \
            out[0] = old;                                                                                                     \
            for (int j = 1; j < m; ++j) {                                                                                     \
                const T x1 = d[old * 3 + 0], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];                                        \
                for (int tid = 0; tid < TT; ++tid) {                                                                          \
                    int besti = 0;                                                                                            \
                    T best = (T)-1;                                                                                           \
                    for (int k = tid; k < n; k += TT) {                                                                       \
                        const T dd = sqdist_##SUF(d[k * 3 + 0], d[k * 3 + 1], d[k * 3 + 2], x1, y1, z1);                      \
                        const T d2 = dd < temp[k] ? dd : temp[k];                                                             \
                        temp[k] = d2;                                                                                         \
                        besti = d2 > best ? k : besti;                                                                        \
                        best = d2 > best ? d2 : best;                                                                         \
                    }                                                                                                         \
                    dists[tid] = best;                                                                                        \
                    dists_i[tid] = besti;                                                                                     \
                }                                                                                                             \
                for (int s = TT >> 1; s >= 1; s >>= 1)                                                                        \
                    for (int tid = 0; tid < s; ++tid) {                                                                       \
                        const T v1 = dists[tid], v2 = dists[tid + s];                                                         \
                        const int i1 = dists_i[tid], i2 = dists_i[tid + s];                                                   \
                        dists[tid] = v1 > v2 ? v1 : v2;                                                                       \
                        dists_i[tid] = v2 > v1 ? i2 : i1;                                                                     \
                    }                                                                                                         \
                old = dists_i[0];                                                                                             \
                out[j] = old;                                                                                                 \
            }                                                                                                                 \
        }                                                                                                                     \
        free(temp); free(dists); free(dists_i);                                                                               \
    }                                                                                                                         \
    static void reheap_##SUF(T* dist, int* idx, int k) {                                                                      \
        int root = 0, child = 1;                                                                                              \
        while (child < k) {                                                                                                   \
            if (child + 1 < k && dist[child + 1] > dist[child]) child++;                                                      \
            if (dist[root] > dist[child]) return;                                                                             \
            const T td = dist[root]; dist[root] = dist[child]; dist[child] = td;                                              \
            const int ti = idx[root]; idx[root] = idx[child]; idx[child] = ti;                                                \
            root = child;                                                                                                     \
            child = root * 2 + 1;                                                                                             \
        }                                                                                                                     \
    }                                                                                                                         \
    /* xyz [b][n][3], new_xyz [b][m][3] -> idx [b][m][nsample], dist2 [b][m][nsample]; nsample <= 100 as in the kernel */      \
    int knn_heap_##SUF(int b, int n, int m, int nsample, const T* xyz, const T* new_xyz, int32_t* idx, T* dist2) {            \
        if (nsample > 100 || nsample < 1) return -1;                                                                          \
        for (int bi = 0; bi < b; ++bi)                                                                                        \
            for (int pt = 0; pt < m; ++pt) {                                                                                  \
                const T* q = new_xyz + ((size_t)bi * m + pt) * 3;                                                             \
                const T* x = xyz + (size_t)bi * n * 3;                                                                        \
                T best_dist[100];                                                                                             \
                int best_idx[100];                                                                                            \
                for (int i = 0; i < nsample; ++i) { best_dist[i] = (T)1e10; best_idx[i] = 0; }                                \
                for (int i = 0; i < n; ++i) {                                                                                 \
                    const T d2 = sqdist_##SUF(q[0], q[1], q[2], x[i * 3 + 0], x[i * 3 + 1], x[i * 3 + 2]);                    \
                    if (d2 < best_dist[0]) {                                                                                  \
                        best_dist[0] = d2;                                                                                    \
                        best_idx[0] = i;                                                                                      \
                        reheap_##SUF(best_dist, best_idx, nsample);                                                           \
                    }                                                                                                         \
                }                                                                                                             \
                for (int i = nsample - 1; i > 0; --i) {                         /* heap_sort */                               \
                    const T td = best_dist[0]; best_dist[0] = best_dist[i]; best_dist[i] = td;                                \
                    const int ti = best_idx[0]; best_idx[0] = best_idx[i]; best_idx[i] = ti;                                  \
                    reheap_##SUF(best_dist, best_idx, i);                                                                     \
                }                                                                                                             \
                for (int i = 0; i < nsample; ++i) {                                                                           \
                    idx[((size_t)bi * m + pt) * nsample + i] = best_idx[i];                                                   \
                    dist2[((size_t)bi * m + pt) * nsample + i] = best_dist[i];                                                \
                }                                                                                                             \
            }                                                                                                                 \
        return 0;                                                                                                             \
    }                                                                                                                         \
    /* unknown [b][n][3], known [b][m][3] -> dist2 [b][n][3] (the kernel's float store of its double minima: 1e40 becomes +inf), idx */ \
    void three_nn_##SUF(int b, int n, int m, const T* unknown, const T* known, T* dist2, int32_t* idx) {                      \
        for (int bi = 0; bi < b; ++bi)                                                                                        \
            for (int pt = 0; pt < n; ++pt) {                                                                                  \
                const T* u = unknown + ((size_t)bi * n + pt) * 3;                                                             \
                const T* kn = known + (size_t)bi * m * 3;                                                                     \
                double best1 = 1e40, best2 = 1e40, best3 = 1e40;                                                              \
                int besti1 = 0, besti2 = 0, besti3 = 0;                                                                       \
                for (int k = 0; k < m; ++k) {                                                                                 \
                    const T d = sqdist_##SUF(u[0], u[1], u[2], kn[k * 3 + 0], kn[k * 3 + 1], kn[k * 3 + 2]);                  \
                    if (d < best1) { best3 = best2; besti3 = besti2; best2 = best1; besti2 = besti1; best1 = d; besti1 = k; } \
                    else if (d < best2) { best3 = best2; besti3 = besti2; best2 = d; besti2 = k; }                            \
                    else if (d < best3) { best3 = d; besti3 = k; }                                                            \
                }                                                                                                             \
                T* o = dist2 + ((size_t)bi * n + pt) * 3;                                                                     \
                int32_t* oi = idx + ((size_t)bi * n + pt) * 3;                                                                \
                /* the kernel stores its double minima into FLOAT dist2: an unfilled slot (1e40) becomes +inf in either build */   \
                o[0] = best1 > 3.0e38 ? (T)INFINITY : (T)best1; o[1] = best2 > 3.0e38 ? (T)INFINITY : (T)best2; o[2] = best3 > 3.0e38 ? (T)INFINITY : (T)best3; \
                oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;                                                               \
            }                                                                                                                 \
    }

DEFINE_POINTOPS(f32, float, fmaf)
DEFINE_POINTOPS(f64, double, fma)
