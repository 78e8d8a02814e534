// Shared declarations of the fcflow HIP engine (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <map>
#include <stdexcept>
#include <mutex>
#include <functional>

#include "../../include/fcflow.h"

namespace fc {

// ---------------------------------------------------------------- errors
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
void set_last_error(const std::string& m);

#define FC_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            throw fc::Error(FC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// Every extern "C" entry point: exceptions -> status code + fc_last_error() text (the C ABI never throws, never exits).
#define FC_API_BEGIN try {
#define FC_API_END                                                                            \
    }                                                                                         \
    catch (const fc::Error& e) { fc::set_last_error(e.what()); return e.code; }               \
    catch (const std::exception& e) { fc::set_last_error(e.what()); return FC_ERR_INVALID; }  \
    return FC_OK;

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

// One-accumulator limb form (spline_wide.hip): activations are stored as hi + lo of x * kOneAccActScale with lo = rn16(x s - hi) UNSCALED.  The
// exact power-of-two pre-scale keeps lo a normal fp16 number for |x| >= 2^-7 (absolute floor 2^-29 below that) and leaves fp16's range at
// |x| < 4094: a producer that meets more raises the range flag like any other split-fp16 kernel (the pass then repeats on the bf16 limbs).
constexpr float kOneAccActScale = 16.0f;

constexpr int ROW_PAD = 256;   // row counts of every workspace matrix are padded to this (largest GEMM BM)
constexpr int COL_PAD = 32;    // feature widths are padded to this (GEMM BK / MFMA tile)

// ---------------------------------------------------------------- in-library kernel timing (fc_profile_*)
// When enabled every launch is bracketed by HIP events ON THE LAUNCH STREAM; the report aggregates, per kernel
// name (as rocprofv3 prints it), launches, total milliseconds and the useful FLOPs / algorithmic bytes the launches
// processed.  bench.py derives its `roofline` object from this over the timed region.
bool prof_enabled();
void prof_filter(const char* substr);
void prof_stride(int n);
struct ProfScope {
    bool on;
    hipStream_t s;
    int slot = -1;
    ProfScope(const char* name, double flops, double bytes, hipStream_t stream);
    ~ProfScope();
};

// ---------------------------------------------------------------- device memory owned by a handle
// Thread-safe (the flow's layers are packed by several host threads): 64 MiB chunks, 256-byte aligned sub-allocations, one hipMalloc
// per chunk instead of one per tensor (a 115-layer flow packs ~4000 tensors).
struct DeviceArena {
    std::vector<void*> blocks;
    size_t total = 0;
    void* alloc_bytes(size_t n);
    float* alloc_floats(size_t n) { return (float*)alloc_bytes((n ? n : 1) * sizeof(float)); }
    float* upload(const std::vector<float>& host);
    ~DeviceArena();
private:
    std::mutex mu;
    char* cur = nullptr;
    size_t cur_left = 0;
};
// fp16 / bf16 limb images of a packed fp32 weight matrix already on the device (misc.hip): [rows][K_pad/16][2 or 3][16]
void launch_limb_images(const float* W, int rows, int K_pad, unsigned short* W3, unsigned short* W2, hipStream_t s);

// ---------------------------------------------------------------- workspace carving (caller-owned memory)
struct WsCarver {
    char* base;
    size_t off = 0, cap;
    bool dry;
    WsCarver(void* b, size_t c, bool dry_run) : base((char*)b), cap(c), dry(dry_run) {}
    float* floats(size_t n) { return (float*)bytes(n * sizeof(float)); }
    void* bytes(size_t n) {
        size_t o = round_up_sz(off, 256);
        off = o + n;
        if (!dry && off > cap) throw Error(FC_ERR_WORKSPACE, "workspace too small");
        return dry ? nullptr : (void*)(base + o);
    }
};

// ---------------------------------------------------------------- packed Linear (see gemm.h)
struct PackedLinear {
    float* W = nullptr;        // [N_alloc][K_pad] row-major, zero padded
    float* bias = nullptr;     // [N_alloc]
    float* colvec = nullptr;   // [N_alloc] or null: rank-1 term  C += rowscal[row] * colvec[n]
    unsigned short* W2 = nullptr;  // the same matrix as 2 fp16 limbs, [N_alloc][K_pad/16][2][16]: w ~= hi + lo'/2048 to 2^-24 (null when an entry
                                   // does not fit fp16): operand of the default split-fp16 GEMM loop
    unsigned short* W3 = nullptr;  // the same matrix as 3 bf16 limbs, [N_alloc][K_pad/16][3][16]: x = hi + mid + lo exactly to 24 bits
                                   // (operand image of the split-bf16 GEMM variant, see gemm.hip)
    unsigned short* Wf = nullptr;  // K <= 512 -> 512 layers of a coupling MLP: the fp16 limb image pre-tiled in MFMA-fragment order,
                                   // [N/32][ks][2][64 lanes][8] (mlprows.hip: one LDS-DMA piece = one linear 1 KiB read); null otherwise
    unsigned short* W1 = nullptr;  // spline parameter layer (K = 8 bins) only: the ONE-ACCUMULATOR fp16 limb image of spline_wide.hip,
                                   // [round_up(N_pad, 256)][K_pad/16][2][16]: w 2^w1_exp = hi + lo with lo UNSCALED, rows in that kernel's column order
    float* bias1 = nullptr;        // its bias in the same order, times kOneAccActScale 2^w1_exp
    int w1_exp = 0;
    bool w1_permuted = false;      // W1 rows in the fused spline kernel's register-slot order (else natural order: a Linear layer of the coupling MLP)
    float wmax = 0.f;              // max |w| of the packed matrix (pack_linear)
    int N_pad = 0;             // columns written (multiple of 32)
    int K_pad = 0;             // multiple of 32 (sum of segment widths)
    int seg_k[3] = {0, 0, 0};  // padded K of each A segment
    int nseg = 0;
    int n_true = 0, k_true = 0;  // un-padded extents (useful-FLOP accounting of the profiler)
    int n_alloc = 0;             // rows of W / bias / colvec actually allocated (zero padded to the GEMM grid)
};

// Column-tile width the GEMM launcher picks for a layer, and the row count W must be zero-padded to so that the
// k-loop needs no bounds checks (pair = pair-packed affine / augment epilogue, always the 128x320 tile).
inline int gemm_bn(int N_pad, bool pair) {
    if (pair) return 320;
    if (N_pad <= 64) return 64;
    if (N_pad % 128 == 0 || N_pad > 320) return 128;
    return 320;
}
inline int gemm_n_alloc(int N_pad) {
    const int a = round_up(N_pad, gemm_bn(N_pad, false)), b = round_up(N_pad, 320), c = round_up(N_pad, 128);
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

// A operand segment: rows x seg_k floats starting at ptr with row pitch lda
struct ASeg { const float* ptr; int lda; };

enum Epilogue { EPI_LINEAR = 0, EPI_AFFINE = 1, EPI_AUGMENT = 2, EPI_SLICE = 3, EPI_SPLINE = 4, EPI_LNQ = 5 };

struct GemmEpi {
    // EPI_LINEAR
    int act = FC_ACT_NONE;
    const float* residual = nullptr; int ldr = 0;
    const unsigned short* residual16 = nullptr; int ldr16 = 0;   // the residual given as an fp16 limb image [rows][ldr16/16][hi 16 | lo' 16] (x = hi + lo'/2048)
    const float* rowscal = nullptr;          // [rows] (extra context per point), used with PackedLinear.colvec
    float* C = nullptr; int ldc = 0;          // may be null when only the limb image C16 is wanted
    const float* gradu = nullptr; int ldgu = 0; int gact = FC_ACT_NONE;   // LINEAR, optional (training data gradient): C = value * act'(gradu[row][col]) for gact
    float* Cpre = nullptr;                    // LINEAR, optional (training forward): the PRE-activation value goes here (pitch ldc), act(value) to C
    unsigned short* C16 = nullptr;            // optional: the output as fp16 limb image [rows][N_pad/16][hi 16 | lo' 16] (operand image of a
                                              // following split-fp16 GEMM, which then copies it instead of re-splitting it per column tile)
    const unsigned short* A16 = nullptr;      // input: the A operand given as such an image (single segment of K_pad columns)
    float c16_scale = 0.f;                    // C16 in the one-accumulator form instead: hi + lo of value * c16_scale, lo unscaled (0: the hi + lo'/2048 form)
    float a16_scale = 0.f;                    // A16 arrives in that form (spline_wide.hip: the fused spline layer and the 512-wide Linear layers)
    float r16_scale = 0.f;                    // residual16 arrives in that form
    // EPI_AFFINE (W pair-packed [s 32 | t 32] x pairs): in-place y2 = x2*s + t on xbuf, logprob[row] += sum log s
    // EPI_AUGMENT (W pair-packed [mu 32 | logsigma 32]): z2 = mu + eps*sigma scattered into xbuf, logprob += -log N(z2)
    float* xbuf = nullptr; int ldx = 0;
    int x2_col0 = 0;           // AFFINE: first column of x2 inside xbuf
    int d2 = 0;                // AFFINE: number of transformed dims; AUGMENT: number of noise dims
    int scale_fn = FC_SCALE_SIGMOID;
    float* logprob = nullptr;  // [rows_valid]
    const float* eps = nullptr; int d_in = 0, d1 = 0, d1_pad = 0;   // AUGMENT: latent index = d_in + q ; x1|x2 split
    float clamp = 0.f;         // AUGMENT/SLICE: std clamp (0 = none)
    int inverse = 0;           // AFFINE: x2 = (y2 - t) / (s*g), no log-det ; AUGMENT: sample only (no log-det)
    int prefetch_dist = 0;     // SPLINE (persistent kernel): != 0 rotates each tile's k loop by a column-tile dependent offset (launch_gemm fills it from knob 21)
    int split = 1 << 30, split_pad = 0;   // AFFINE: transformed dim j lives at column x2_col0 + (j < split ? j : split_pad + j - split)
    const float* post_scale = nullptr;    // AFFINE: optional per-dim factor g folded behind s (CIF: ActNorm of the x part)
    const float* val = nullptr; int ldval = 0;             // SLICE: values whose log N(.; mu, sigma) is ADDED to logprob
    const float* val_shift = nullptr; const float* val_scale = nullptr;   // SLICE: v = (val - shift) * scale ; AUGMENT(+inverse): z = z / scale + shift
    int rows_valid = 0;        // rows that exist in user-visible outputs
    // EPI_LNQ (LayerNorm folded through a linear layer, flow_engine.cpp build_lnq): columns [0, d2) are the centred layer's outputs --
    // only their per-row sums of squares leave the kernel, ldj_part[(64-column block) * ldj_pitch + row] = sum -- and columns
    // [d2, d2 + 64) go to C (pitch ldc) as the un-normalised q projection
    // EPI_SPLINE (forward rational-quadratic spline coupling evaluated by the workgroup that produced the parameters; uses
    // xbuf / ldx / x2_col0 / d2 / rows_valid above): y2 overwrites x2, per-tile log-dets are ACCUMULATED into ldj_part[tile * ldj_pitch + row]
    // (the caller zeroes the buffer before the first layer and reduces it over the tiles once after the last, launch_ldj_reduce)
    int spline_K = 0;
    float* ldj_part = nullptr;
    size_t ldj_pitch = 0;
    double flops_hint = 0.0;   // filled by launch_gemm for the profiler
};

// The split-fp16 GEMM loop (the default) cannot represent |activation| >= 65504.  It therefore only runs inside a guard scope of
// the calling thread: the scope hands the kernels a device flag they raise on such a value, and the entry point that opened the
// scope repeats its whole computation with the bf16-limb loop (unbounded range) when the flag came back set.  Outside a scope
// launch_gemm always takes the bf16-limb loop.
// Kernel variants that lost a same-box A/B are compiled only with -DFC_DEV_VARIANTS (`python -m flowcompare_amd.build --dev`):
// the default library holds, per role, the shipped kernel, the bf16-limb range fallback and ONE fp32-input reference loop.  In a default build
// fc_debug_set refuses the knob values that would select a developer variant.
#ifdef FC_DEV_VARIANTS
#define FC_DEV(...) __VA_ARGS__
constexpr bool kDevVariants = true;
#else
#define FC_DEV(...)
constexpr bool kDevVariants = false;
#endif

// One-time setup per (call site, device) -- hipFuncSetAttribute for > 64 KB of dynamic LDS, the CU count behind a persistent grid: a process
// may drive several devices and pack from a pool of host threads, so a process-wide `static bool done` is not enough.
struct PerDeviceOnce {
    std::mutex mu;
    bool done[64] = {};
    int value[64] = {};
    // runs f(dev) the first time the calling thread's current device is seen here; returns what that call returned
    template <class F>
    int run(F&& f) {
        int dev = 0;
        FC_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        if (dev < 0 || dev >= 64) return f(dev);
        if (!done[dev]) { value[dev] = f(dev); done[dev] = true; }
        return value[dev];
    }
};

struct Fp16Guard {
    Fp16Guard(int* dev_flag, hipStream_t s);
    ~Fp16Guard();
    bool overflowed();            // closes the scope: waits for the stream and reads the flag
    void defer(std::function<void()> rerun);   // closes the scope WITHOUT waiting: the flag is copied to pinned host memory behind the pass and
                                               // the pass is queued for fc_range_check_resolve (deferred range check, below)
    int* flag; hipStream_t stream; bool open;
};
// Deferred range check (fc_range_check_defer / _resolve, include/fcflow.h): with the calling thread's switch on, a guarded entry point only
// ENQUEUES its fast pass, a 4-byte copy of the flag into a pinned host slot and an event -- no stream synchronisation, so any number of
// forwards can be queued back to back.  guard_resolve() then walks the queued passes in order: it waits for a pass's event, and repeats the
// pass on the unbounded-range loops if its flag came back set (or if an earlier pass was repeated: it may have consumed that pass's output).
bool guard_deferred();
int guard_pending();
void guard_set_deferred(bool on);
int guard_resolve();              // returns the number of passes it repeated
// Training primitives (train.hip): the CALLER owns the overflow flag for a whole optimisation step (forward and backward run on
// different host threads under torch.autograd), so the scope only lends it to launch_gemm for one call -- no reset, no wait.
struct Fp16FlagScope {
    explicit Fp16FlagScope(int* dev_flag);
    ~Fp16FlagScope();
    int* prev;
};
bool gemm_fp16_enabled();
bool gemm_limb_chain_ok();
bool gemm_limb_chain_all_ok();   // every hidden activation of a limb-chained MLP as a limb image (A16 in, C16 out, residual16)
bool gemm_lnq_ok();             // the LayerNorm -> q fold (EPI_LNQ) can run: guard scope open, default tile
void launch_lnq_finalize(float* q, int ldq, const float* sumsq, int nslots, size_t pitch, int width, const float* q_bias, int rows, hipStream_t s);      // inside a guard scope on the default tile: producers may emit / consumers may take limb images
bool gemm_split_enabled();        // a split (limb) GEMM loop is the active variant: the fused spline epilogue is available
int* gemm_fp16_flag();         // the open scope's device flag of the calling thread, or null
template <class F>
inline void run_fp16_guarded(int* dev_flag, hipStream_t s, F&& fn, bool deferrable = false) {
    if (!dev_flag || !gemm_fp16_enabled()) { fn(); return; }
    if (deferrable && guard_deferred()) {           // (fn must own its arguments: it may run again after this call has returned)
        Fp16Guard g(dev_flag, s);
        fn();
        g.defer(std::function<void()>(fn));
        return;
    }
    bool over;
    { Fp16Guard g(dev_flag, s); fn(); over = g.overflowed(); }
    if (over) fn();
}
void launch_gemm(const PackedLinear& L, const ASeg* segs, int rows_alloc, const GemmEpi& e, int epi_kind, hipStream_t s);

// ---------------------------------------------------------------- other kernels (misc.hip / attention.hip / knn.hip)
void launch_pack_rows(const float* src, int src_ld, int src_cols, float* dst, int dst_ld, int dst_col0, int dst_cols_zero_to,
                      int rows, hipStream_t s);
void launch_fill(float* p, float v, size_t n, hipStream_t s);
void launch_repeat_extra(const float* extra, int X, float* rowscal, int B, int N, hipStream_t s);
void launch_layernorm(float* h, int ld, int width, int rows, hipStream_t s);   // in place, no affine (folded into q-proj)
// limb_ws: scratch of attention_limb_ws_bytes(rows of k/v, dh_pad) bytes for the split-fp16 kernel's K/V limb images; with a null
// limb_ws, outside an Fp16Guard scope or for dh_pad > 64 the fp32-input MFMA kernel runs
size_t attention_limb_ws_bytes(long kv_rows, int dh_pad);
// premlp.hip: fused pre-attention MLP -> LayerNorm -> q projection (one launch instead of six) when the shapes allow it
bool premlp_rows_ok(int rows_alloc, int ldq, const float* qout, const float* keep_ws, size_t keep_floats);   // the row-resident kernel's launch conditions
bool premlp_fusable(const PackedLinear& in, const std::vector<PackedLinear>& mid, const PackedLinear& out, const PackedLinear& q);
void launch_premlp(const float* x, int ldx, const PackedLinear& in, const std::vector<PackedLinear>& mid, const PackedLinear& out,
                   const PackedLinear& q, int act, float* qout, int ldq, int rows_alloc, int rows_valid, hipStream_t s, float* keep_ws = nullptr,
                   size_t keep_floats = 0, const PackedLinear* lu = nullptr, const float* xprev = nullptr);
// the previous flow layer's folded ActNorm + permuter `lu` as a pre-layer of the row-resident kernel: z = lu(xprev) is written to x (premlp.hip)
bool premlp_lu_fusable(const PackedLinear& lu, const PackedLinear& in, int act, int ldx);
// mlprows.hip: in_layer + hidden layers of a 512-wide coupling MLP in one launch, activations resident in registers
size_t mlp_rows_image_bytes(int K_pad);
void launch_mlp_rows_image(const PackedLinear& L, unsigned short* Wf, hipStream_t s);      // fills L's fragment-major image from L.W2
bool mlp_rows_eligible(const PackedLinear& in, const std::vector<PackedLinear>& mid, int act);
bool mlp_rows_fills_the_chip(int rows_alloc);           // at least 3/4 of the CUs get a 128-row workgroup
void launch_mlp_rows(const PackedLinear& in, const std::vector<PackedLinear>& mid, const ASeg* segs, const float* rowscal, int act,
                     float* const h[3], unsigned short* out16, int rows_alloc, int rows_valid, hipStream_t s, float out16_scale = 0.f);   // out16_scale: GemmEpi::c16_scale of the last layer
void launch_limb_decode(const unsigned short* img, float* out, int ldo, int rows, int width, hipStream_t s);   // row-major limb image -> fp32 (tests)
// spline_wide.hip: the fused spline parameter layer on 256 x 256 tiles with one accumulator per output (K = 8 bins, limb-chained input)
bool spline_wide_eligible(const PackedLinear& L, int K_bins);
void spline_wide_attach(DeviceArena& arena, PackedLinear& L, float wmax, hipStream_t s, bool permute = true);
void launch_spline_wide(const PackedLinear& L, const GemmEpi& e, int rows_alloc, hipStream_t s);
bool linear_wide_eligible(const PackedLinear& L, const GemmEpi& e, int rows_alloc);   // a GELU Linear layer with one-accumulator images in and out
void launch_linear_wide(const PackedLinear& L, const GemmEpi& e, int rows_alloc, hipStream_t s);
bool gemm_linear_wide_on();      // knob 29 (1: for scenes of at least 2048 target points, 2: at any size)
int gemm_linear_wide_knob();
bool gemm_spline_wide_on();      // knob 13 = 5 (shipped): launch_gemm routes eligible EPI_SPLINE launches there
// the same 256 x 256 main loop for the wide Linear layers of a TRAINING step (train.hip: the spline parameter layer's forward and its data
// gradient): the point operand is an fp32 panel, split into limbs after its LDS read; C = (A W^T + bias (+ addend)) (* act'(gradu))
constexpr int kTrainWideWExp = 11;         // weights of those layers are stored as hi + lo of w 2^11 (|w| < 32; the pack kernel raises the range flag beyond)
struct TrainWideArgs {
    const float* A = nullptr; int lda = 0;          // [rows_pad][lda] fp32, lda >= K_pad
    const unsigned short* W1 = nullptr;              // [round_up(n_cols, 256)][K_pad / 16][hi 16 | lo 16], scaled by 2^kTrainWideWExp
    const float* bias1 = nullptr;                    // [round_up(n_cols, 256)] times a_scale 2^kTrainWideWExp, or null
    int K_pad = 0, rows_pad = 0, n_cols = 0;
    float a_scale = kOneAccActScale;                 // scale of A when row_absmax is null
    const float* row_absmax = nullptr;               // [rows_pad] max |A[row, :]|: per-row power-of-two scales (gradients)
    float* C = nullptr; int ldc = 0;
    const float* addend = nullptr;                   // [rows_pad][ldc] or null
    const float* gradu = nullptr; int ldgu = 0, gact = 0;
    int* ovf = nullptr;
    double flops = 0.0;
};
void launch_train_wide(const TrainWideArgs& a, hipStream_t s);
extern int g_train_wide;         // knob 31: 1 = training Linear layers with at least 1024 outputs run on it (shipped), 0 = on the fp32-A 128 x 128 loop
// row maxima of a gradient panel, handed from the kernel that writes it (training spline backward) to the data-gradient GEMM that reads it
float* train_rowmax_reserve(const float* tensor, int rows, hipStream_t s);
const float* train_rowmax_take(const float* tensor, int rows, hipStream_t s);
// staging.hip: the steps either side of the path
void launch_fps_nd(const float* pts, int ld, int C, int64_t* idx, int B, int n, int m, float* dist_scratch, hipStream_t s);
void launch_co_unit_sphere(const float* p0, int n0, const float* p1, int n1, int ld, float* o0, float* o1, float* inverse, int B, hipStream_t s);
void launch_clamp_infs(float* t, long n, float* stats4, int* status, hipStream_t s);
void launch_change_map(float* lp10, int N, float* lp00, int N0, float* out, int B, float multiple, float hard_cutoff, int use_cutoff,
                       float* stats4, int* status, hipStream_t s);
void launch_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                      int B, int N, int n_stride_rows, int M, int m_stride_rows, int dh_pad, void* limb_ws, hipStream_t s);
bool attention_fp16_enabled();
// un-normalised q of the LayerNorm -> q fold: the attention kernel finishes it (rstd from the per-row sums of squares, bias) on load
struct AttnLnq { const float* sumsq; int slots; size_t pitch; float inv_width; const float* bias; };
void launch_attention_c16(const float* q, int ldq, const unsigned short* kv_c16, int n_pad, int col0, float* out, int ldo, int B, int N,
                          int n_stride_rows, int M, int m_stride_rows, int dh_pad, hipStream_t s, const AttnLnq* lnq = nullptr);
void launch_attention_op(const float* q, const float* k, const float* v, float* out, int B, int N, int M, int dh_pad, float scale,
                         void* limb_ws, hipStream_t s);
void launch_base_density(const float* x, int ldx, int d1, int d1_pad, int d2, float* logprob, float log_const,
                         float* z_out, int D, int rows, hipStream_t s);
void launch_spline(const float* params, int ldp, float* xbuf, int ldx, int x2_col0, int d2, int K, float* logprob, int rows, int inverse,
                   hipStream_t s);       // parameters in the tile-grouped dim-major layout of spline.h
void launch_ldj_reduce(const float* part, int ntiles, size_t pitch, float* logprob, int rows, hipStream_t s);   // params[row, p*d2s + j] = parameter p of dim j
void launch_expm_coupling(const float* params, int ldp, float* xbuf, int ldx, int x2_col0, int d2, const float* scal4, float* logprob,
                          int rows, int inverse, hipStream_t s);
void launch_spline_flat(const float* x, const float* params, float* y, float* lad, int64_t n, int K, int inverse, hipStream_t s);
void launch_knn(const float* f, int ldf, int C, int32_t* idx, int B, int M, int m_stride_rows, int k, hipStream_t s, const int32_t* warm = nullptr);
void launch_gather_max(const float* uv, int lduv, int c_out, const int32_t* idx, int k, float* out, int ldo, int out_col0,
                       int B, int M, int m_stride_rows, hipStream_t s);
void launch_pool_max_mean(const float* t, int ldt, int width, float* out, int ldo, int B, int M, int m_stride_rows, hipStream_t s);

// ---------------------------------------------------------------- host-side weight table
struct HostTensor { const float* data; std::vector<int64_t> shape; int64_t numel() const; };
struct WeightTable {
    std::map<std::string, HostTensor> t;
    WeightTable(const fc_tensor* tensors, int n);
    bool has(const std::string& name) const { return t.count(name) != 0; }
    const HostTensor& get(const std::string& name) const;
    const HostTensor& get(const std::string& name, std::initializer_list<int64_t> shape) const;
};

}  // namespace fc
