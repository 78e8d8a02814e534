// Round-4 probe for the fused spline GEMM: a 256 x 256 workgroup tile on EIGHT waves of 64 points x 128 parameters with ONE fp32 accumulator
// per output block (unscaled low limbs, power-of-two pre-scales), two wave groups that alternate LOAD and MFMA segments (the guide's
// 8-phase template: cdna_hip_programming.md section 5), persistent workgroups, one continuous LDS-DMA stream.
//   C[rows, 3840] = A[rows, 512] W[3840, 512]^T   (C2's spline parameter layer: rows = 65536)
// What it measures: the MAIN LOOP of that structure against the shipped VAR 11 loop (0.59-0.65 ms with knob 14 = 2), and the accuracy of the
// one-accumulator limb form against fp64 / an fp32 fmaf chain on the three weight scales of one_acc_probe.
//   hipcc --offload-arch=gfx950 -O3 wide_gemm_probe.hip -o wide_gemm_probe && ./wide_gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

struct WideParams {
    const unsigned short* A16;   // [rows][KT][4 x 16]: per k16 block hi 16 | lo 16 (fp16), lo unscaled; values pre-scaled by Sa
    const unsigned short* W1;    // [cols][KT][4 x 16]: the same for the weights, pre-scaled by Sw
    float* C; int ldc;           // optional fp32 output (validation)
    int KT;                      // k32 steps (even)
    int nbm, nbn;                // 256-row tiles, 256-column tiles
    int col_group;
    float out_scale;             // 1 / (Sa Sw)
    unsigned long long* stamps;
};

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// DMA pieces per LOAD segment of a k step (phases 0..3), for the group that fetches the points (waves 0-3) and the weights (waves 4-7)
// ABL (timing ablations, results invalid): 1 no DMA in the loop, 2 no fragment reads, 3 no MFMAs, 4 no barriers at all (one wave group races the other)
template <int STAGGER, int P0, int P1, int P2, int P3, int Q0, int Q1, int Q2, int Q3, int PRIO, int ABL = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2)))
void wide_kernel(const WideParams p) {
    static_assert(P0 + P1 + P2 + P3 == 8 && Q0 + Q1 + Q2 + Q3 == 8 && (STAGGER == 0 || Q3 == 0), "eight pieces per wave and k step; the lagging group may not issue in its last LOAD segment");
    extern __shared__ char smc[];
    typedef __attribute__((address_space(3))) char lds_char;
    typedef const __attribute__((address_space(1))) char glb_char;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, pw = wave & 3;
    const int KT = p.KT;
    const unsigned rowbytes = (unsigned)KT * 128u;
    const int ntiles = p.nbm * p.nbn, G = gridDim.x;
    int t = blockIdx.x;
    if (t >= ntiles) return;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    auto tile_of = [&](int b, int& bm, int& bn) {
        const int xcd = b & 7, loc = b >> 3;
        if (p.col_group > 0) {
            const int rows_x = p.nbm >> 3, Gc = p.col_group;
            const int g = loc / (rows_x * Gc);
            const int rem = loc - g * rows_x * Gc;
            const int w = p.nbn - g * Gc < Gc ? p.nbn - g * Gc : Gc;
            const int r = rem / w;
            bm = xcd * rows_x + r;
            bn = g * Gc + (rem - r * w);
        } else {
            const int q = ntiles >> 3, r = ntiles & 7;
            const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
            bm = L / p.nbn;
            bn = L - bm * p.nbn;
        }
    };
    // LDS: [stage 0 points 32 KB | stage 1 points 32 KB | stage 0 weights 32 KB | stage 1 weights 32 KB]; a row = 128 B = one k32 step of one point /
    // weight row = 8 chunks of 16 B, logical chunk c = 4 (k16 block) + 2 limb + (k half) at physical chunk c ^ ((row >> 1) & 7)
    // DMA piece i of this wave: operand rows (wave & 3) * 64 + 8 i + (lane >> 3); the swizzle term depends on the parity of i only
    unsigned poff[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int r = pw * 64 + par * 8 + (lane >> 3);
        const int cl = (lane & 7) ^ ((r >> 1) & 7);
        poff[par] = (unsigned)r * rowbytes + cl * 16;
    }
    auto src_of = [&](int bm, int bn) -> const char* {
        return grp == 0 ? reinterpret_cast<const char*>(p.A16) + (size_t)bm * 256 * rowbytes : reinterpret_cast<const char*>(p.W1) + (size_t)bn * 256 * rowbytes;
    };
    const int dst0 = grp * 65536 + pw * 8192;                           // LDS byte offset of this wave's first piece in stage 0
#define WG_DMA(SRC_, ST_, I0_, N_)                                                                                                      \
    {                                                                                                                                     \
        _Pragma("unroll") for (int i_ = (I0_); i_ < (I0_) + (N_); ++i_)                                                                 \
            __builtin_amdgcn_global_load_lds((glb_char*)((SRC_) + (size_t)(i_ >> 1) * 16 * rowbytes + poff[i_ & 1]),                      \
                                             (lds_char*)(smc + dst0 + (ST_) * 32768 + i_ * 1024), 16, 0, 0);                            \
    }
    // fragment reads: lane (li, lh) reads chunk (sub * 4 + q * 2 + lh) ^ xsw of its row
    const int xsw = (li >> 1) & 7;
    int abase[2][2], bbase[2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = ((sub * 4 + q * 2 + lh) ^ xsw) * 16;
            abase[sub][q] = (pw * 64 + li) * 128 + c;
            bbase[sub][q] = 65536 + (grp * 128 + li) * 128 + c;
        }

    int bm, bn;
    tile_of(t, bm, bn);
    const char* src = src_of(bm, bn);
    WG_DMA(src, 0, 0, 8)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (STAGGER && grp == 1) __builtin_amdgcn_s_barrier();                // the second group runs one segment behind the first

    floatx16 acc[2][4];
    for (;;) {
        const int tn = t + G;
        const bool has_next = tn < ntiles;
        int nbm = bm, nbn = bn;
        if (has_next) tile_of(tn, nbm, nbn);
        const char* nsrc = src_of(nbm, nbn);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define WG_PHASE(ST_, SUB_, H_, NP_, NQ_, I0P_, I0Q_, LAST_)                                                                             \
        {                                                                                                                                 \
            if (ABL != 2) {                                                                                                               \
            if ((H_) == 0) {                                                                                                              \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                            \
                    _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                        \
                        xf[i][q] = *reinterpret_cast<const f16x8*>(smc + abase[SUB_][q] + (ST_) * 32768 + i * 4096);                      \
            }                                                                                                                             \
            _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                             \
                _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                            \
                    wf[jj][q] = *reinterpret_cast<const f16x8*>(smc + bbase[SUB_][q] + (ST_) * 32768 + (2 * (H_) + jj) * 4096);           \
            }                                                                                                                             \
            if (ABL != 1) {                                                                                                               \
            if (grp == 0) { if ((NP_) > 0) WG_DMA(dsrc, (ST_) ^ 1, I0P_, NP_) }                                                          \
            else { if ((NQ_) > 0) WG_DMA(dsrc, (ST_) ^ 1, I0Q_, NQ_) }                                                                   \
            }                                                                                                                             \
            if ((LAST_) && STAGGER && grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            if (ABL != 4) __builtin_amdgcn_s_barrier();                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                                      \
            if (ABL != 3)                                                                                                                 \
            _Pragma("unroll") for (int pr = 0; pr < 3; ++pr)                                                                             \
                _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                         \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                        \
                        acc[i][2 * (H_) + jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[jj][pr == 1 ? 1 : 0], xf[i][pr == 2 ? 1 : 0],  \
                                                                                       acc[i][2 * (H_) + jj], 0, 0, 0);                   \
            if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                                      \
            if ((LAST_) && !(STAGGER && grp == 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            if (ABL != 4) __builtin_amdgcn_s_barrier();                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
        }
#define WG_STEP(ST_)                                                                                                                      \
        {                                                                                                                                 \
            WG_PHASE(ST_, 0, 0, P0, Q0, 0, 0, 0)                                                                                          \
            WG_PHASE(ST_, 0, 1, P1, Q1, P0, Q0, 0)                                                                                        \
            WG_PHASE(ST_, 1, 0, P2, Q2, P0 + P1, Q0 + Q1, 0)                                                                              \
            WG_PHASE(ST_, 1, 1, P3, Q3, P0 + P1 + P2, Q0 + Q1 + Q2, 1)                                                                    \
        }
        f16x8 xf[2][2], wf[2][2];
        if (ABL == 2) { for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int c = 0; c < 8; ++c) { xf[a][b][c] = (_Float16)(float)(lane + c); wf[a][b][c] = (_Float16)(float)(li - c); } }
        for (int kt = 0; kt < KT; kt += 2) {
            const char* dsrc = src + (size_t)(kt + 1) * 128;
            WG_STEP(0)
            dsrc = kt + 2 < KT ? src + (size_t)(kt + 2) * 128 : nsrc;
            WG_STEP(1)
        }
        // ---- epilogue
        if (p.C) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = bm * 256 + pw * 64 + i * 32 + li;
                float* cr = p.C + (size_t)row * p.ldc + bn * 256 + grp * 128 + 4 * lh;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float4 v;
                        v.x = acc[i][j][4 * g + 0] * p.out_scale; v.y = acc[i][j][4 * g + 1] * p.out_scale;
                        v.z = acc[i][j][4 * g + 2] * p.out_scale; v.w = acc[i][j][4 * g + 3] * p.out_scale;
                        *reinterpret_cast<float4*>(cr + j * 32 + 8 * g) = v;
                    }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        }
        if (!has_next) break;
        t = tn; bm = nbm; bn = nbn; src = nsrc;
    }
    if (STAGGER && grp == 0) __builtin_amdgcn_s_barrier();
    if (p.stamps && tid == 0) { p.stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; p.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

// ---- the same tile on v_mfma_f32_16x16x32_f16 (4 point blocks x 8 parameter blocks of 16 x 16 per wave): the guide's DVFS item 7 says the chip may
// hold a higher clock on this shape.  A k32 step is ONE MFMA k extent: lane l supplies row l & 15, k quarter l >> 4 (chunks 0, 1, 4, 5 of the hi limb,
// 2, 3, 6, 7 of the lo limb).  Phases: parameter blocks 2 f, 2 f + 1 against all four point blocks (24 MFMAs of 16 cycles).
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int STAGGER, int P0, int P1, int P2, int P3, int Q0, int Q1, int Q2, int Q3>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2)))
void wide16_kernel(const WideParams p) {
    extern __shared__ char smc[];
    typedef __attribute__((address_space(3))) char lds_char;
    typedef const __attribute__((address_space(1))) char glb_char;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, pw = wave & 3;
    const int KT = p.KT;
    const unsigned rowbytes = (unsigned)KT * 128u;
    const int ntiles = p.nbm * p.nbn, G = gridDim.x;
    int t = blockIdx.x;
    if (t >= ntiles) return;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    auto tile_of = [&](int b, int& bm, int& bn) {
        const int xcd = b & 7, loc = b >> 3;
        const int rows_x = p.nbm >> 3, Gc = p.col_group;
        const int g = loc / (rows_x * Gc);
        const int rem = loc - g * rows_x * Gc;
        const int w = p.nbn - g * Gc < Gc ? p.nbn - g * Gc : Gc;
        const int r = rem / w;
        bm = xcd * rows_x + r;
        bn = g * Gc + (rem - r * w);
    };
    unsigned poff[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int r = pw * 64 + par * 8 + (lane >> 3);
        const int cl = (lane & 7) ^ ((r >> 1) & 7);
        poff[par] = (unsigned)r * rowbytes + cl * 16;
    }
    auto src_of = [&](int bm, int bn) -> const char* {
        return grp == 0 ? reinterpret_cast<const char*>(p.A16) + (size_t)bm * 256 * rowbytes : reinterpret_cast<const char*>(p.W1) + (size_t)bn * 256 * rowbytes;
    };
    const int dst0 = grp * 65536 + pw * 8192;
    const int xsw = (l15 >> 1) & 7;
    int abase[2], bbase[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = (((kq >> 1) * 4 + q * 2 + (kq & 1)) ^ xsw) * 16;
        abase[q] = (pw * 64 + l15) * 128 + c;
        bbase[q] = 65536 + (grp * 128 + l15) * 128 + c;
    }
    int bm, bn;
    tile_of(t, bm, bn);
    const char* src = src_of(bm, bn);
    WG_DMA(src, 0, 0, 8)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (STAGGER && grp == 1) __builtin_amdgcn_s_barrier();
    floatx4 acc[4][8];
    for (;;) {
        const int tn = t + G;
        const bool has_next = tn < ntiles;
        int nbm = bm, nbn = bn;
        if (has_next) tile_of(tn, nbm, nbn);
        const char* nsrc = src_of(nbm, nbn);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
#define W16_PHASE(ST_, F_, NP_, NQ_, I0P_, I0Q_, LAST_)                                                                                  \
        {                                                                                                                                 \
            if ((F_) == 0) {                                                                                                              \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                            \
                    _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                        \
                        xf[i][q] = *reinterpret_cast<const f16x8*>(smc + abase[q] + (ST_) * 32768 + i * 2048);                            \
            }                                                                                                                             \
            _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                             \
                _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                            \
                    wf[jj][q] = *reinterpret_cast<const f16x8*>(smc + bbase[q] + (ST_) * 32768 + (2 * (F_) + jj) * 2048);                 \
            if (grp == 0) { if ((NP_) > 0) WG_DMA(dsrc, (ST_) ^ 1, I0P_, NP_) }                                                          \
            else { if ((NQ_) > 0) WG_DMA(dsrc, (ST_) ^ 1, I0Q_, NQ_) }                                                                   \
            if ((LAST_) && STAGGER && grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            __builtin_amdgcn_s_barrier();                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            __builtin_amdgcn_s_setprio(1);                                                                                                \
            _Pragma("unroll") for (int pr = 0; pr < 3; ++pr)                                                                             \
                _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                         \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                        \
                        acc[i][2 * (F_) + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jj][pr == 1 ? 1 : 0], xf[i][pr == 2 ? 1 : 0],  \
                                                                                       acc[i][2 * (F_) + jj], 0, 0, 0);                   \
            __builtin_amdgcn_s_setprio(0);                                                                                                \
            if ((LAST_) && !(STAGGER && grp == 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
            __builtin_amdgcn_s_barrier();                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                            \
        }
#define W16_STEP(ST_)                                                                                                                     \
        {                                                                                                                                 \
            W16_PHASE(ST_, 0, P0, Q0, 0, 0, 0)                                                                                            \
            W16_PHASE(ST_, 1, P1, Q1, P0, Q0, 0)                                                                                          \
            W16_PHASE(ST_, 2, P2, Q2, P0 + P1, Q0 + Q1, 0)                                                                                \
            W16_PHASE(ST_, 3, P3, Q3, P0 + P1 + P2, Q0 + Q1 + Q2, 1)                                                                      \
        }
        f16x8 xf[4][2], wf[2][2];
        for (int kt = 0; kt < KT; kt += 2) {
            const char* dsrc = src + (size_t)(kt + 1) * 128;
            W16_STEP(0)
            dsrc = kt + 2 < KT ? src + (size_t)(kt + 2) * 128 : nsrc;
            W16_STEP(1)
        }
        if (p.C) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = bm * 256 + pw * 64 + i * 16 + l15;
                float* cr = p.C + (size_t)row * p.ldc + bn * 256 + grp * 128 + 4 * kq;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float4 v;
                    v.x = acc[i][j][0] * p.out_scale; v.y = acc[i][j][1] * p.out_scale; v.z = acc[i][j][2] * p.out_scale; v.w = acc[i][j][3] * p.out_scale;
                    *reinterpret_cast<float4*>(cr + j * 16) = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
        }
        if (!has_next) break;
        t = tn; bm = nbm; bn = nbn; src = nsrc;
    }
    if (STAGGER && grp == 0) __builtin_amdgcn_s_barrier();
    if (p.stamps && tid == 0) { p.stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; p.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

static void make_image(const std::vector<float>& X, int rows, int K, float scale, std::vector<unsigned short>& img) {
    img.assign((size_t)rows * K * 2, 0);
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < K; ++k) {
            const float x = X[(size_t)r * K + k] * scale;
            const _Float16 h = (_Float16)x;
            const _Float16 l = (_Float16)(x - (float)h);
            const size_t blk = (size_t)r * (K / 16) + k / 16;
            img[blk * 32 + (k & 15)] = __builtin_bit_cast(unsigned short, h);
            img[blk * 32 + 16 + (k & 15)] = __builtin_bit_cast(unsigned short, l);
        }
}

template <class KERN>
static float time_kernel(KERN kern, const WideParams& p, int grid, int reps) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 133120, 0, p);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 133120, 0, p);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char** argv) {
    const int K = 512, rows = argc > 1 ? atoi(argv[1]) : 65536, vrows = 2048, N = argc > 2 ? atoi(argv[2]) : 3840;      // (N = 512: a hidden layer of the coupling MLP)
    const bool timing_only = argc > 3;
    std::mt19937 g(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    int cus = 0;
    CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int slots = cus & ~7;
    printf("CUs %d, persistent grid %d\n", cus, slots);

    auto k_stag = wide_kernel<1, 2, 2, 2, 2, 3, 3, 2, 0, 1>;
    auto k_stag_b = wide_kernel<1, 3, 3, 2, 0, 3, 3, 2, 0, 1>;
    auto k_stag_np = wide_kernel<1, 2, 2, 2, 2, 3, 3, 2, 0, 0>;
    auto k_lock = wide_kernel<0, 2, 2, 2, 2, 2, 2, 2, 2, 1>;
    auto k16_lock = wide16_kernel<0, 1, 2, 3, 2, 1, 2, 3, 2>;
    auto k16_stag = wide16_kernel<1, 1, 3, 3, 1, 2, 3, 3, 0>;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_lock), hipFuncAttributeMaxDynamicSharedMemorySize, 133120));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k16_stag), hipFuncAttributeMaxDynamicSharedMemorySize, 133120));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stag), hipFuncAttributeMaxDynamicSharedMemorySize, 133120));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stag_b), hipFuncAttributeMaxDynamicSharedMemorySize, 133120));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stag_np), hipFuncAttributeMaxDynamicSharedMemorySize, 133120));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lock), hipFuncAttributeMaxDynamicSharedMemorySize, 133120));

    // ---------------- accuracy: 2048 rows, three weight scales, against fp64 and an fp32 fmaf chain (sampled columns)
    for (float wamp : {0.04f, 0.4f, 0.004f}) {
        if (timing_only) break;
        std::vector<float> A((size_t)vrows * K), W((size_t)N * K);
        std::uniform_real_distribution<float> ud(-wamp, wamp);
        for (auto& v : A) { float x = nd(g); v = x > 0 ? x : 0.05f * x; }
        for (auto& v : W) v = ud(g);
        float wmax = 0.f;
        for (float v : W) wmax = std::max(wmax, std::fabs(v));
        int sw = 0;
        while (wmax * ldexpf(1.f, sw) < 16384.f) ++sw;
        const float Sa = 16.f, Sw = ldexpf(1.f, sw);
        std::vector<unsigned short> ia, iw;
        make_image(A, vrows, K, Sa, ia);
        make_image(W, N, K, Sw, iw);
        unsigned short *dA, *dW; float* dC;
        CHK(hipMalloc(&dA, ia.size() * 2)); CHK(hipMalloc(&dW, iw.size() * 2)); CHK(hipMalloc(&dC, (size_t)vrows * N * 4));
        CHK(hipMemcpy(dA, ia.data(), ia.size() * 2, hipMemcpyHostToDevice)); CHK(hipMemcpy(dW, iw.data(), iw.size() * 2, hipMemcpyHostToDevice));
        WideParams p{dA, dW, dC, N, K / 32, vrows / 256, N / 256, 0, 1.0f / (Sa * Sw), nullptr};
        std::vector<float> C((size_t)vrows * N);
        p.col_group = p.nbn;
        for (int variant = 0; variant < 3; ++variant) {
            CHK(hipMemset(dC, 0xff, (size_t)vrows * N * 4));
            if (variant == 0) hipLaunchKernelGGL(k_stag, dim3(std::min(slots, p.nbm * p.nbn)), dim3(512), 133120, 0, p);
            else if (variant == 1) hipLaunchKernelGGL(k_lock, dim3(std::min(slots, p.nbm * p.nbn)), dim3(512), 133120, 0, p);
            else hipLaunchKernelGGL(k16_lock, dim3(std::min(slots, p.nbm * p.nbn)), dim3(512), 133120, 0, p);
            CHK(hipDeviceSynchronize());
            CHK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
            double emax = 0, esum = 0, e32max = 0, e32sum = 0, cmax = 0; size_t n = 0;
            for (int r = 0; r < vrows; r += 7)
                for (int c = (r * 13) % 11; c < N; c += 37) {
                    double s = 0; float s32 = 0.f;
                    for (int k = 0; k < K; ++k) { s += (double)A[(size_t)r * K + k] * W[(size_t)c * K + k]; s32 = fmaf(A[(size_t)r * K + k], W[(size_t)c * K + k], s32); }
                    const double e = std::fabs((double)C[(size_t)r * N + c] - s), e32 = std::fabs((double)s32 - s);
                    emax = std::max(emax, e); esum += e; e32max = std::max(e32max, e32); e32sum += e32; cmax = std::max(cmax, std::fabs(s)); ++n;
                }
            printf("weights U(-%g, %g) Sw 2^%d %s: max |C| %.3f, %zu samples: one-acc max %.2e mean %.2e | fp32 fmaf chain max %.2e mean %.2e\n", wamp, wamp, sw,
                   variant == 0 ? "staggered" : variant == 1 ? "lockstep " : "16x16x32 ", cmax, n, emax, esum / n, e32max, e32sum / n);
        }
        CHK(hipFree(dA)); CHK(hipFree(dW)); CHK(hipFree(dC));
    }

    // ---------------- main-loop time at the C2 size
    {
        std::vector<float> A((size_t)rows * K), W((size_t)N * K);
        std::uniform_real_distribution<float> ud(-0.04f, 0.04f);
        for (auto& v : A) { float x = nd(g); v = x > 0 ? x : 0.05f * x; }
        for (auto& v : W) v = ud(g);
        std::vector<unsigned short> ia, iw;
        make_image(A, rows, K, 16.f, ia);
        make_image(W, N, K, ldexpf(1.f, 18), iw);
        unsigned short *dA, *dW;
        CHK(hipMalloc(&dA, ia.size() * 2)); CHK(hipMalloc(&dW, iw.size() * 2));
        CHK(hipMemcpy(dA, ia.data(), ia.size() * 2, hipMemcpyHostToDevice)); CHK(hipMemcpy(dW, iw.data(), iw.size() * 2, hipMemcpyHostToDevice));
        const double flop = 2.0 * rows * (double)N * K;
        unsigned long long* dS; CHK(hipMalloc(&dS, 4096 * 16));
        auto clock_of = [&](int grid) {
            std::vector<unsigned long long> h(2 * grid);
            CHK(hipMemcpy(h.data(), dS, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> c;
            for (int i = 0; i < grid; ++i) c.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
            std::sort(c.begin(), c.end());
            return c[c.size() / 2];
        };
        struct V { const char* name; void (*k)(const WideParams); };
        const V vs[] = {{"staggered 3320/3320", k_stag_b}, {"lockstep 2222", k_lock}, {"16x16x32 lockstep", k16_lock}, {"16x16x32 staggered", k16_stag}, {"lockstep 2222 again", k_lock}, {"16x16x32 lock again", k16_lock},
                        {"stag, no DMA", wide_kernel<1, 3, 3, 2, 0, 3, 3, 2, 0, 1, 1>}, {"stag, no ds_read", wide_kernel<1, 3, 3, 2, 0, 3, 3, 2, 0, 1, 2>},
                        {"stag, no MFMA", wide_kernel<1, 3, 3, 2, 0, 3, 3, 2, 0, 1, 3>}, {"stag, no barriers", wide_kernel<1, 3, 3, 2, 0, 3, 3, 2, 0, 1, 4>},
                        {"lock, no DMA", wide_kernel<0, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1>}, {"lock, no ds_read", wide_kernel<0, 2, 2, 2, 2, 2, 2, 2, 2, 1, 2>},
                        {"lock, no MFMA", wide_kernel<0, 2, 2, 2, 2, 2, 2, 2, 2, 1, 3>}};
        for (auto& v : vs) CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.k), hipFuncAttributeMaxDynamicSharedMemorySize, 133120));
        for (int cg : {5}) {
            if (N / 256 <= cg) cg = N / 256;
            WideParams p{dA, dW, nullptr, 0, K / 32, rows / 256, N / 256, cg, 1.f, dS};
            if (p.nbm % 8 != 0) { printf("rows %d: row tiles not a multiple of 8, skipped\n", rows); continue; }
            const int grid = std::min(slots, p.nbm * p.nbn);
            for (auto& v : vs) {
                const float a = time_kernel(v.k, p, grid, 20);
                const double ghz = clock_of(grid);
                printf("rows %d col_group %2d %-22s %.3f ms  %.0f TF-eq  %.3f of 2500 issued  clock %.2f GHz  matrix pipes %.1f %% busy (if all MFMAs issued)\n", rows, cg, v.name, a, flop / a * 1e-9,
                       3.0 * flop / a * 1e-9 / 2500.0, ghz, 100.0 * 3.0 * flop / (a * 1e-3) / (1024.0 * 2.0 * 32 * 32 * 16 / 32.0 * ghz * 1e9));
            }
        }
        CHK(hipFree(dA)); CHK(hipFree(dW));
    }
    return 0;
}
