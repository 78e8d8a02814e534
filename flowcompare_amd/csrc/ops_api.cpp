// Single-operator entry points of the C ABI (fc_op_*): the SAME kernels the engines launch, wrapped so that
// unit-level parity tests can drive them with dense tensors.  These wrappers allocate temporary device
// memory for the padded layouts (they are test/diagnostic conveniences, not the hot path).
#include <memory>

#include "hostpack.h"
#include "spline.h"

namespace fc {

struct TmpBuf {
    void* p = nullptr;
    explicit TmpBuf(size_t bytes) { FC_HIP(hipMalloc(&p, bytes ? bytes : 4)); }
    ~TmpBuf() { (void)hipFree(p); }
    float* f() const { return (float*)p; }
};
}  // namespace fc


namespace fc { void one_acc_gemm_debug(const float*, const float*, const float*, float, float*, int, int, int, hipStream_t); long gemm_fp16_fallbacks(); extern int g_train_wgrad16; extern int g_train_attn16; extern int g_gemm_dma; extern int g_spline_ablate; extern int g_gemm_dma_linear; extern int g_limb_chain_all; extern int g_gemm_prefetch3; extern int g_premlp_chain; extern int g_gemm_stamp; extern int g_gemm_small_tiles; extern int g_spline_prefetch; extern int g_mlp_rows; extern int g_knn_mfma; extern int g_premlp_lu; extern int g_spline_wide_dma; extern int g_spline_wide_colgroup; extern int g_linear_wide; extern int g_knn_warm; size_t gemm_read_stamps(unsigned long long*, size_t); void flow_set_trace(float*, size_t); }
namespace fc { extern int g_gemm_variant, g_gemm_colgroup, g_gemm_bigtile, g_attn_fp16, g_fused_spline, g_premlp_fused, g_limb_chain, g_lnq_fold; }

extern "C" {

/* tuning knobs for profiles/kernel_bench.py (not part of the stable ABI surface in fcflow.h on purpose) */
int fc_debug_set(int32_t key, int32_t value) {
    if (!fc::kDevVariants && ((key == 0 && (value == 0 || value == 1)) || (key == 3 && value != 3) || (key == 8 && value == 1) || (key == 13 && value != 5 && value != 4 && value != 2) ||
                              (key == 15 && value != 2) || (key == 17 && value != 0) || (key == 27 && value != 0)))
        return FC_ERR_UNSUPPORTED;       /* a developer variant: compiled only with -DFC_DEV_VARIANTS (python -m flowcompare_amd.build --dev) */
    if (key == 0) fc::g_gemm_variant = value;
    else if (key == 2) fc::g_gemm_colgroup = value;
    else if (key == 3) fc::g_gemm_bigtile = value;
    else if (key == 5) fc::g_attn_fp16 = value;
    else if (key == 7) fc::g_fused_spline = value;
    else if (key == 8) fc::g_premlp_fused = value;
    else if (key == 9) fc::g_limb_chain = value;
    else if (key == 10) fc::g_lnq_fold = value;
    else if (key == 11) fc::g_train_wgrad16 = value;
    else if (key == 12) fc::g_train_attn16 = value;
    else if (key == 13) fc::g_gemm_dma = value;
    else if (key == 15) fc::g_gemm_dma_linear = value;   /* limb-image A in EPI_LINEAR: 1 = LDS-DMA loop on the 256x128 tile (default), 0 = register-staged 128x128 */
    else if (key == 16) fc::g_limb_chain_all = value;    /* 1 = the coupling MLP's hidden activations exist only as limb images (default 0) */
    else if (key == 17) fc::g_gemm_prefetch3 = value;   /* Linear GEMM: three register sets of prefetch (VAR 6) instead of two (VAR 5) */
    else if (key == 19) fc::g_premlp_chain = value;
    else if (key == 21) fc::g_spline_prefetch = value;
    else if (key == 22) fc::g_gemm_small_tiles = value;
    else if (key == 23) fc::g_mlp_rows = value;          /* 1 = row-resident coupling MLP chain (mlprows.hip, default), 0 = one GEMM launch per layer */
    else if (key == 26) fc::g_premlp_lu = value;         /* 1 = ActNorm + LU as a pre-layer of the row-resident pre-attention kernel (default), 0 = its own GEMM launch */
    else if (key == 24) fc::g_knn_mfma = value;          /* 1 = k-NN Gram tiles on the matrix cores (default), 0 = lane-per-candidate kernel */
    else if (key == 27) fc::g_spline_wide_dma = value;   /* developer builds: DMA pieces per phase of the wide fused spline kernel (spline_wide.hip) */
    else if (key == 29) fc::g_linear_wide = value;       /* hidden layers of the coupling MLP on the 256 x 256 one-accumulator kernel: 0 = off (default: measured no faster than the chain), 1 = for scenes of >= 2048 target points, 2 = at any size */
    else if (key == 32) fc::g_knn_warm = value;         /* 1 = DGCNN levels 1-3 start their k-NN stream from the previous level's neighbour sets (default; exact either way), 0 = from -inf */
    else if (key == 31) fc::g_train_wide = value;       /* 1 = training Linear layers with >= 1024 outputs (the spline parameter layer) on the 256 x 256 one-accumulator loop (default), 0 = on the fp32-A 128 x 128 loop, 3 = 1 with non-temporal stores of a GB-sized output (measured slower) */
    else if (key == 30) { if (value != 0) return FC_ERR_UNSUPPORTED; }      /* (was: attention as two staggered wave groups -- measured no faster, removed with the one-accumulator attention kernel) */
    else if (key == 28) fc::g_spline_wide_colgroup = value;   /* column-group size of its tile order (-1 = shipped) */
    else if (key == 20) fc::g_gemm_stamp = value;        /* diagnostic: in-kernel phase stamps of the LDS-DMA fused-spline launches */
    else if (key == 14) fc::g_spline_ablate = value;     /* diagnostic: 1 = fused spline epilogue without the spline evaluation, 2 = main loop only (results invalid) */
    else return FC_ERR_INVALID;
    return FC_OK;
}

/* diagnostic / test entry (not part of fcflow.h): out[rows, N] = x[rows, K] W[N, K]^T + bias through the ONE-ACCUMULATOR limb form on the 256 x 256 main loop of
   spline_wide.hip, nothing else -- the product the fused spline layer is built on, measurable against fp64 by itself; device pointers, wmax = max |W| */
int fc_debug_one_acc_gemm_f32(const float* x, const float* W, const float* bias, float wmax, float* out, int32_t rows, int32_t N, int32_t K, void* stream) {
    FC_API_BEGIN
    fc::one_acc_gemm_debug(x, W, bias, wmax, out, rows, N, K, (hipStream_t)stream);
    FC_API_END
}

/* 1 when the library was built with the developer kernel variants (-DFC_DEV_VARIANTS) */
int32_t fc_debug_dev_variants(void) { return fc::kDevVariants ? 1 : 0; }

/* host-side view of the spline parameter layer's column layout (csrc/spline.h) for the CPU tests: column of (transformed dim j, parameter
   pp) and the dim-major position inside a tile that the LDS-tile epilogues store a column at; no device call */
int32_t fc_debug_spline_col(int32_t j, int32_t pp, int32_t K) { return fc::spline_col(j, pp, K); }
int32_t fc_debug_spline_tile_pos(int32_t c, int32_t K) { return fc::spline_tile_pos(c, K); }

/* diagnostic (knob 20): copies the phase stamps of the last stamped fused-spline launch (16 x u64 per workgroup) to host memory; returns the count */
int64_t fc_debug_gemm_stamps(uint64_t* host, int64_t max_n) {
    try { return (int64_t)fc::gemm_read_stamps(reinterpret_cast<unsigned long long*>(host), (size_t)max_n); } catch (...) { return -1; }
}

/* deferred range check (include/fcflow.h) */
int fc_range_check_defer(int32_t on) {
    FC_API_BEGIN
    if (!on && fc::guard_pending()) fc::guard_resolve();
    fc::guard_set_deferred(on != 0);
    FC_API_END
}
int fc_range_check_resolve(int32_t* n_repeated) {
    FC_API_BEGIN
    const int n = fc::guard_resolve();
    if (n_repeated) *n_repeated = n;
    FC_API_END
}
int32_t fc_range_check_pending(void) { return fc::guard_pending(); }

/* diagnostic: trace of every coupling's x2 input of the calling thread's next fc_flow_logprob_f32 calls into a DEVICE buffer
   [n_flow_layers][B * N][d2] (flow_engine.cpp flow_set_trace); NULL switches it off.  Test infrastructure, not part of fcflow.h. */
int fc_debug_flow_trace(float* device_buf, int64_t capacity_floats) {
    fc::flow_set_trace(device_buf, device_buf && capacity_floats > 0 ? (size_t)capacity_floats : 0);
    return FC_OK;
}

/* number of calls that were repeated with the bf16-limb GEMMs because an activation left fp16's range (tests, diagnostics) */
int64_t fc_debug_fp16_fallbacks(void) { return (int64_t)fc::gemm_fp16_fallbacks(); }

int fc_op_linear_f32(const float* x, const float* W, const float* bias, const float* residual, float* y, int32_t rows, int32_t N, int32_t K,
                     int32_t act, void* stream) {
    FC_API_BEGIN
    using namespace fc;
    if (!x || !W || !y || rows < 1 || N < 1 || K < 1) throw Error(FC_ERR_INVALID, "fc_op_linear_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int rp = round_up(rows, ROW_PAD), np = round_up(N, 32), kp = round_up(K, 32);
    const int na = gemm_n_alloc(np);
    TmpBuf xp((size_t)rp * kp * 4), wp((size_t)na * kp * 4), bp((size_t)na * 4), cp((size_t)rp * np * 4), rpad(residual ? (size_t)rp * np * 4 : 4);
    launch_fill(xp.f(), 0.f, (size_t)rp * kp, s);
    launch_fill(wp.f(), 0.f, (size_t)na * kp, s);
    launch_fill(bp.f(), 0.f, (size_t)na, s);
    launch_pack_rows(x, K, K, xp.f(), kp, 0, K, rows, s);
    launch_pack_rows(W, K, K, wp.f(), kp, 0, K, N, s);
    if (bias) launch_pack_rows(bias, N, N, bp.f(), np, 0, N, 1, s);
    if (residual) {
        launch_fill(rpad.f(), 0.f, (size_t)rp * np, s);
        launch_pack_rows(residual, N, N, rpad.f(), np, 0, N, rows, s);
    }
    PackedLinear L;
    L.W = wp.f(); L.bias = bp.f(); L.N_pad = np; L.K_pad = kp; L.nseg = 1; L.seg_k[0] = kp; L.n_alloc = na;
    L.n_true = N; L.k_true = K;
    std::unique_ptr<TmpBuf> w3buf, w2buf;
    TmpBuf flag(sizeof(int));
    if (g_gemm_variant == 3 || g_gemm_variant == 5) {           // split variants: limb images via a host round trip (test path only)
        FC_HIP(hipStreamSynchronize(s));
        std::vector<float> hw((size_t)na * kp);
        FC_HIP(hipMemcpy(hw.data(), wp.f(), hw.size() * 4, hipMemcpyDeviceToHost));
        const std::vector<unsigned short> w3 = make_bf16_limbs(hw, na, kp);
        w3buf.reset(new TmpBuf(w3.size() * 2));
        FC_HIP(hipMemcpy(w3buf->p, w3.data(), w3.size() * 2, hipMemcpyHostToDevice));
        L.W3 = (unsigned short*)w3buf->p;
        const std::vector<unsigned short> w2 = make_f16_limbs(hw, na, kp);
        if (!w2.empty()) {
            w2buf.reset(new TmpBuf(w2.size() * 2));
            FC_HIP(hipMemcpy(w2buf->p, w2.data(), w2.size() * 2, hipMemcpyHostToDevice));
            L.W2 = (unsigned short*)w2buf->p;
        }
    }
    GemmEpi e{};
    e.act = act; e.C = cp.f(); e.ldc = np;
    if (residual) { e.residual = rpad.f(); e.ldr = np; }
    ASeg a{xp.f(), kp};
    run_fp16_guarded((int*)flag.p, s, [&] { launch_gemm(L, &a, rp, e, EPI_LINEAR, s); });
    launch_pack_rows(cp.f(), np, N, y, N, 0, N, rows, s);
    FC_HIP(hipStreamSynchronize(s));
    FC_API_END
}

/* in_layer + hidden layers of one reference MLP (models/nets.py:19-30) at hidden width 512 over cat(x0, x1) (+ the rank-1 extra-context
   term rowscal[row] * net.colvec): the last hidden activation, decoded from the limb image the output GEMM would consume.  use_rows != 0:
   the row-resident chain kernel (mlprows.hip); 0: one GEMM launch per layer (limb-chained).  Tensors (host fp32): net.in_layer.{weight,bias},
   net.layers.i.{weight,bias}, net.out_layer.weight (shape check only), optional net.colvec [512]. */
int fc_op_mlp_hidden_f32(const float* x0, int32_t k0, const float* x1, int32_t k1, const float* rowscal, const fc_tensor* tensors, int32_t n_tensors,
                         float* out, int32_t rows, int32_t act, int32_t use_rows, void* stream) {
    FC_API_BEGIN
    using namespace fc;
    if (!x0 || !out || rows < 1 || k0 < 1 || k1 < 0 || (k1 > 0 && !x1)) throw Error(FC_ERR_INVALID, "fc_op_mlp_hidden_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    WeightTable wt(tensors, n_tensors);
    DeviceArena arena;
    PackedMLP m;
    pack_mlp_mid(arena, wt, "net", m);
    const int H = m.sizes[0], p0 = round_up(k0, 32), p1 = k1 > 0 ? round_up(k1, 32) : 0;
    for (int h : m.sizes) if (h != 512) throw Error(FC_ERR_UNSUPPORTED, "fc_op_mlp_hidden_f32: hidden width must be 512");
    {
        const HostTensor& w = wt.get("net.in_layer.weight", {H, k0 + k1});
        std::vector<int> km = map_prefix(k0, p0);
        for (int j = 0; j < p1; ++j) km.push_back(j < k1 ? k0 + j : -1);
        VecD cv;
        if (wt.has("net.colvec")) cv = vec_from(wt.get("net.colvec", {H}));
        std::vector<int> segk = {p0};
        if (p1) segk.push_back(p1);
        m.in_layer = pack_linear(arena, mat_from(w), vec_from(wt.get("net.in_layer.bias", {H})), cv, map_prefix(H, H), km, segk);
    }
    attach_mlp_rows_images(arena, m);
    const int rp = round_up(rows, ROW_PAD);
    TmpBuf xa((size_t)rp * p0 * 4), xb((size_t)rp * std::max(p1, 32) * 4), h0((size_t)rp * 512 * 4), h1((size_t)rp * 512 * 4), h2((size_t)rp * 512 * 4),
        h16((size_t)rp * 512 * 4), flag(sizeof(int)), rs((size_t)rp * 4);
    launch_fill(xa.f(), 0.f, (size_t)rp * p0, s);
    launch_fill(xb.f(), 0.f, (size_t)rp * std::max(p1, 32), s);
    launch_fill(rs.f(), 0.f, (size_t)rp, s);
    launch_pack_rows(x0, k0, k0, xa.f(), p0, 0, k0, rows, s);
    if (k1 > 0) launch_pack_rows(x1, k1, k1, xb.f(), p1, 0, k1, rows, s);
    if (rowscal) launch_pack_rows(rowscal, 1, 1, rs.f(), 1, 0, 1, rows, s);
    float* const h[3] = {h0.f(), h1.f(), h2.f()};
    ASeg segs[2] = {{xa.f(), p0}, {xb.f(), std::max(p1, 32)}};
    const float* rsp = rowscal && m.in_layer.colvec ? rs.f() : nullptr;
    FC_HIP(hipDeviceSynchronize());                       // (the images were built on the null stream)
    run_fp16_guarded((int*)flag.p, s, [&] {
        if (use_rows == 1) launch_mlp_rows(m.in_layer, m.mid, segs, rsp, act, h, (unsigned short*)h16.p, rp, rows, s);
        else if (run_mlp_hidden_generic(m, segs, rsp, act, h, 512, rp, s, rows, (unsigned short*)h16.p, 0.f, use_rows == 2) != -1)      // (2: hidden layers on the 256 x 256 one-accumulator kernel)
            throw Error(FC_ERR_INVALID, "fc_op_mlp_hidden_f32: the limb chain is off");
    });
    launch_limb_decode((const unsigned short*)h16.p, out, 512, rows, 512, s);
    FC_HIP(hipStreamSynchronize(s));
    FC_API_END
}

int fc_op_attention_f32(const float* q, const float* k, const float* v, float* out, int32_t B, int32_t N, int32_t M, int32_t D, float scale,
                        void* stream) {
    FC_API_BEGIN
    if (!q || !k || !v || !out) throw fc::Error(FC_ERR_INVALID, "fc_op_attention_f32: null pointer");
    if (D != 32 && D != 64 && D != 128) throw fc::Error(FC_ERR_UNSUPPORTED, "fc_op_attention_f32: D must be 32, 64 or 128");
    fc::TmpBuf limbs(fc::attention_limb_ws_bytes((long)B * M, D)), flag(sizeof(int));
    fc::run_fp16_guarded((int*)flag.p, (hipStream_t)stream,
                         [&] { fc::launch_attention_op(q, k, v, out, B, N, M, D, scale, limbs.p, (hipStream_t)stream); });
    FC_HIP(hipStreamSynchronize((hipStream_t)stream));
    FC_API_END
}

int fc_op_knn_f32(const float* f, int32_t* idx, int32_t B, int32_t M, int32_t C, int32_t k, void* stream) {
    FC_API_BEGIN
    using namespace fc;
    if (!f || !idx || B < 1 || M < 1 || C < 1) throw Error(FC_ERR_INVALID, "fc_op_knn_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int cp = round_up(C, 32);
    TmpBuf fp((size_t)B * M * cp * 4);
    launch_pack_rows(f, C, C, fp.f(), cp, 0, cp, B * M, s);
    launch_knn(fp.f(), cp, C, idx, B, M, M, k, s);
    FC_HIP(hipStreamSynchronize(s));
    FC_API_END
}

// the same search started from given neighbour sets (what the DGCNN engine does between its levels): idx_warm [B, M, k] int32, may equal idx
int fc_op_knn_warm_f32(const float* f, const int32_t* idx_warm, int32_t* idx, int32_t B, int32_t M, int32_t C, int32_t k, void* stream) {
    FC_API_BEGIN
    using namespace fc;
    if (!f || !idx || !idx_warm || B < 1 || M < 1 || C < 1) throw Error(FC_ERR_INVALID, "fc_op_knn_warm_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int cp = round_up(C, 32);
    TmpBuf fp((size_t)B * M * cp * 4);
    launch_pack_rows(f, C, C, fp.f(), cp, 0, cp, B * M, s);
    launch_knn(fp.f(), cp, C, idx, B, M, M, k, s, idx_warm);
    FC_HIP(hipStreamSynchronize(s));
    FC_API_END
}

int fc_stage_fps_f32(const float* pts, int32_t ld, int32_t C, int64_t* idx, int32_t B, int32_t n, int32_t m, void* stream) {
    FC_API_BEGIN
    if (!pts || !idx || B < 1 || n < 1 || m < 1) throw fc::Error(FC_ERR_INVALID, "fc_stage_fps_f32: bad argument");
    fc::TmpBuf scratch(n > 24576 ? (size_t)B * n * sizeof(float) : 4);
    fc::launch_fps_nd(pts, ld, C, idx, B, n, m, scratch.f(), (hipStream_t)stream);
    FC_HIP(hipStreamSynchronize((hipStream_t)stream));
    FC_API_END
}

int fc_stage_co_unit_sphere_f32(const float* p0, int32_t n0, const float* p1, int32_t n1, int32_t ld, float* out0, float* out1, float* inverse,
                                int32_t B, void* stream) {
    FC_API_BEGIN
    if (!p0 || !out0 || !inverse || (n1 > 0 && (!p1 || !out1))) throw fc::Error(FC_ERR_INVALID, "fc_stage_co_unit_sphere_f32: null pointer");
    fc::launch_co_unit_sphere(p0, n0, n1 > 0 ? p1 : p0, n1, ld, out0, n1 > 0 ? out1 : out0, inverse, B, (hipStream_t)stream);
    FC_API_END
}

int fc_clamp_infs_f32(float* t, int64_t n, void* stream) {
    FC_API_BEGIN
    if (!t || n < 0) throw fc::Error(FC_ERR_INVALID, "fc_clamp_infs_f32: bad argument");
    fc::TmpBuf tmp(4 * sizeof(float) + sizeof(int));
    fc::launch_clamp_infs(t, (long)n, tmp.f(), (int*)(tmp.f() + 4), (hipStream_t)stream);
    FC_HIP(hipStreamSynchronize((hipStream_t)stream));
    FC_API_END
}

int fc_change_map_f32(float* lp10, int32_t N, float* lp00, int32_t N0, float* out, int32_t B, float multiple, float hard_cutoff,
                      int32_t use_cutoff, int32_t* invalid, void* stream) {
    FC_API_BEGIN
    if (!lp10 || !lp00 || !out || !invalid) throw fc::Error(FC_ERR_INVALID, "fc_change_map_f32: null pointer");
    fc::TmpBuf tmp(4 * sizeof(float) + sizeof(int));
    int* status = (int*)(tmp.f() + 4);
    fc::launch_change_map(lp10, N, lp00, N0, out, B, multiple, hard_cutoff, use_cutoff, tmp.f(), status, (hipStream_t)stream);
    int h = 0;
    FC_HIP(hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    FC_HIP(hipStreamSynchronize((hipStream_t)stream));
    *invalid = h;
    FC_API_END
}

int fc_op_rqspline_f32(const float* x, const float* params, float* y, float* logabsdet, int64_t n, int32_t K, int32_t inverse, void* stream) {
    FC_API_BEGIN
    if (!x || !params || !y || !logabsdet || n < 0) throw fc::Error(FC_ERR_INVALID, "fc_op_rqspline_f32: bad argument");
    fc::launch_spline_flat(x, params, y, logabsdet, n, K, inverse, (hipStream_t)stream);
    FC_API_END
}

}  // extern "C"
