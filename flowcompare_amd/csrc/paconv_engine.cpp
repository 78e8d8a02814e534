// fc_paconv: the PAConv context embedder (PointNet++ SSG U-Net, models/scene_seg_PAConv/model/pointnet2/pointnet2_paconv_seg.py:14-82)
// as a launch schedule.  Per set-abstraction level l (n -> n/4 points, K = 32 neighbours, 3 PAConv layers):
//   fps -> gather new_xyz -> 32-NN (ascending) -> group + first-layer input E = [f - f_centre | f]
//   per layer: ScoreNet(xyz_nbr - xyz_centre) -> scores[edges, 8];  G = E @ weightbank  (fp32 MFMA GEMM, rows = edges, N = 8*Cout);
//              O = ReLU(BN(sum_m score_m G_m)) -> next E, or max over the 32 neighbours after the last layer.
// Per feature-propagation level: 3-NN inverse-distance interpolation + skip concat, then 1x1 conv + BN + ReLU as GEMMs
// with the BN folded into the weights.  Head: the shared MLP runner (GELU).
#include <algorithm>
#include <cmath>
#include <memory>

#include "hostpack.h"

namespace fc {
size_t fps_scratch_floats(int B, int n);
void launch_fps(const float* xyz, int ld, int32_t* idx, int B, int n, int m, float* scratch, hipStream_t s);
void launch_gather_xyz(const float* src, int ld, const int32_t* idx, float* dst, int B, int n, int m, hipStream_t s);
void launch_knn_xyz(const float* xyz, int ld, const float* qxyz, int32_t* out, int B, int n, int m, int k, hipStream_t s);
void launch_paconv_group(const float* xyz, int ldxyz, const float* feat, int ldf, int C, const float* qxyz, const int32_t* nidx, float* E, int ldE,
                         float* gdiff, int B, int n, int m, int K, hipStream_t s);
void launch_scorenet(const float* gdiff, const float* w, float* scores, int edges, hipStream_t s);
void launch_score_reduce(const float* G, int ldg, const float* scores, const float* bn_s, const float* bn_t, int Cout, int K, float* dst, int ldd,
                         int dst_col0, int mode, int total_queries, hipStream_t s);
void launch_three_nn_interp(const float* uxyz, int ldu, const float* kxyz, int ldk, const float* Fk, int ldfk, int C2, const float* Fu, int ldfu, int C1,
                            float* X, int ldX, int B, int nu, int mk, hipStream_t s);

struct PaLayer {
    int cin = 0, cout = 0;      // PAConv input channels (before the x2 of kernel_input 'neighbor') and output channels
    PackedLinear bank;          // [8*cout][round32(2*cin)]
    float* score_w = nullptr;   // 200 floats, see scorenet_kernel
    float* bn_s = nullptr;
    float* bn_t = nullptr;
};
struct FpLayer { PackedLinear lin; int cout = 0; };
}  // namespace fc

struct fc_paconv {
    int* fp16_flag = nullptr;   // device word raised by the split-fp16 GEMM loop on an activation >= 65504 (common.h: Fp16Guard)
    fc::DeviceArena arena;
    int c_feat = 0;                       // input feature channels (input_dim - 3)
    std::vector<fc::PaLayer> sa[4];
    std::vector<fc::FpLayer> fp[4];
    int sa_out[4] = {0, 0, 0, 0};
    fc::PackedMLP mlp;
    int E = 0, E_pad = 0, H_pad = 0;
    static constexpr int K = 32;
};

namespace fc {

static void bn_fold2(const WeightTable& wt, const std::string& p, int C, VecD& s, VecD& t) {
    const HostTensor& g = wt.get(p + ".weight", {C});
    const HostTensor& b = wt.get(p + ".bias", {C});
    const HostTensor& m = wt.get(p + ".running_mean", {C});
    const HostTensor& v = wt.get(p + ".running_var", {C});
    s.resize(C); t.resize(C);
    for (int i = 0; i < C; ++i) {
        s[i] = (double)g.data[i] / std::sqrt((double)v.data[i] + 1e-5);
        t[i] = (double)b.data[i] - (double)m.data[i] * s[i];
    }
}

static void build_paconv(fc_paconv& e, const WeightTable& wt) {
    int prev = -1;
    for (int l = 0; l < 4; ++l) {
        for (int j = 0;; ++j) {
            const std::string p = "SA_modules." + std::to_string(l) + ".mlps.0.layer" + std::to_string(j);
            if (!wt.has(p + ".weightbank")) break;
            const HostTensor& wb = wt.get(p + ".weightbank");
            if (wb.shape.size() != 2 || wb.shape[0] % 2 || wb.shape[1] % 8) throw Error(FC_ERR_SHAPE, p + ".weightbank: expected [2*C_in, 8*C_out]");
            PaLayer L;
            L.cin = (int)wb.shape[0] / 2;
            L.cout = (int)wb.shape[1] / 8;
            if (j == 0) {
                if (l == 0) { e.c_feat = L.cin - 3; if (e.c_feat < 0 || e.c_feat > 29) throw Error(FC_ERR_UNSUPPORTED, "PAConv: input feature channels out of range"); }
                else if (L.cin != prev + 3) throw Error(FC_ERR_SHAPE, p + ".weightbank: C_in does not chain (+3 xyz)");
            } else if (L.cin != prev) throw Error(FC_ERR_SHAPE, p + ".weightbank: C_in does not chain");
            MatD bank = mat_from(wb);                              // [2cin][8cout] -> GEMM wants [N = 8cout][K = 2cin]
            MatD W(8 * L.cout, 2 * L.cin);
            for (int k = 0; k < 2 * L.cin; ++k) for (int n = 0; n < 8 * L.cout; ++n) W.at(n, k) = bank.at(k, n);
            L.bank = pack_linear(e.arena, W, {}, {}, map_prefix(8 * L.cout, 8 * L.cout), map_prefix(2 * L.cin, round_up(2 * L.cin, 32)),
                                 {round_up(2 * L.cin, 32)});
            // ScoreNet 3 -> 16 (no bias, BN, ReLU) -> 8 (bias); hidden [16], m = 8 (paconv.py:64-65 defaults)
            const HostTensor& w0 = wt.get(p + ".scorenet.mlp_convs_hidden.0.weight", {16, 3, 1, 1});
            const HostTensor& w1 = wt.get(p + ".scorenet.mlp_convs_hidden.1.weight", {8, 16, 1, 1});
            const HostTensor& b1 = wt.get(p + ".scorenet.mlp_convs_hidden.1.bias", {8});
            VecD s0, t0;
            bn_fold2(wt, p + ".scorenet.mlp_bns_hidden.0", 16, s0, t0);
            std::vector<float> sw(200);
            for (int i = 0; i < 16; ++i) { for (int c = 0; c < 3; ++c) sw[3 * i + c] = (float)(s0[i] * w0.data[3 * i + c]); sw[48 + i] = (float)t0[i]; }
            for (int i = 0; i < 128; ++i) sw[64 + i] = w1.data[i];
            for (int i = 0; i < 8; ++i) sw[192 + i] = b1.data[i];
            L.score_w = e.arena.upload(sw);
            VecD s, t;
            bn_fold2(wt, p + ".bn", L.cout, s, t);
            std::vector<float> fs(s.begin(), s.end()), ft(t.begin(), t.end());
            L.bn_s = e.arena.upload(fs);
            L.bn_t = e.arena.upload(ft);
            prev = L.cout;
            e.sa[l].push_back(L);
        }
        if (e.sa[l].empty()) throw Error(FC_ERR_MISSING, "SA_modules." + std::to_string(l) + ": no PAConv layers found");
        e.sa_out[l] = prev;
    }
    // feature propagation: FP_modules.i consumes level i (unknown, skip) and level i+1 (known)
    for (int i = 3; i >= 0; --i) {
        const int c_skip = i == 0 ? e.c_feat : e.sa_out[i - 1];
        int cin = -1;
        for (int j = 0;; ++j) {
            const std::string p = "FP_modules." + std::to_string(i) + ".mlp.layer" + std::to_string(j);
            if (!wt.has(p + ".conv.weight")) break;
            const HostTensor& w = wt.get(p + ".conv.weight");
            if (w.shape.size() != 4 || w.shape[2] != 1 || w.shape[3] != 1) throw Error(FC_ERR_SHAPE, p + ".conv.weight: expected 1x1 conv");
            const int co = (int)w.shape[0], ci = (int)w.shape[1];
            if (j == 0) {
                const int c_known = i == 3 ? e.sa_out[3] : e.fp[i + 1].back().cout;
                if (ci != c_known + c_skip) throw Error(FC_ERR_SHAPE, p + ".conv.weight: expected " + std::to_string(c_known + c_skip) + " input channels");
            } else if (ci != cin) throw Error(FC_ERR_SHAPE, p + ".conv.weight: channels do not chain");
            VecD s, t;
            bn_fold2(wt, p + ".bn.bn", co, s, t);
            MatD W = mat_from(w);
            for (int o = 0; o < co; ++o) for (int c = 0; c < ci; ++c) W.at(o, c) *= s[o];
            FpLayer L;
            L.cout = co;
            L.lin = pack_linear(e.arena, W, t, {}, map_prefix(co, round_up(co, 32)), map_prefix(ci, round_up(ci, 32)), {round_up(ci, 32)});
            e.fp[i].push_back(L);
            cin = co;
        }
        if (e.fp[i].empty()) throw Error(FC_ERR_MISSING, "FP_modules." + std::to_string(i) + ": no layers found");
    }
    const int head_in = e.fp[0].back().cout;
    pack_mlp_mid(e.arena, wt, "out_mlp", e.mlp);
    const HostTensor& w = wt.get("out_mlp.in_layer.weight");
    if (w.shape[1] != head_in) throw Error(FC_ERR_SHAPE, "out_mlp.in_layer.weight: expected input width " + std::to_string(head_in));
    const int n = (int)w.shape[0];
    e.mlp.in_layer = pack_linear(e.arena, mat_from(w), vec_from(wt.get("out_mlp.in_layer.bias", {n})), {}, map_prefix(n, round_up(n, 32)),
                                 map_prefix(head_in, round_up(head_in, 32)), {round_up(head_in, 32)});
    const HostTensor& wo = wt.get("out_mlp.out_layer.weight");
    const int hl = e.mlp.sizes.back();
    e.E = (int)wo.shape[0];
    e.E_pad = round_up(e.E, 32);
    e.mlp.out_layer = pack_linear(e.arena, mat_from(wo), vec_from(wt.get("out_mlp.out_layer.bias", {e.E})), {}, map_prefix(e.E, e.E_pad),
                                  map_prefix(hl, round_up(hl, 32)), {round_up(hl, 32)});
    e.H_pad = std::max(max_hidden_pad(e.mlp), 32);
}

struct PaWs {
    int n[5];                 // points per level
    float* xyz[5];            // [B*n_l][4]
    float* feat[5];           // level features, channels-last, pitch ldf[l]
    int ldf[5];
    int32_t *fidx, *nidx;
    float *gdiff, *scores, *Ea, *Eb, *G, *X, *Ya, *Yb, *h[3], *otmp, *fps_tmp;
    int P_pad;
};
static PaWs plan_pa(const fc_paconv& e, int B, int M, void* ws, size_t bytes, bool dry, size_t* need) {
    PaWs w{};
    WsCarver c(ws, bytes, dry);
    w.n[0] = M;
    for (int l = 1; l <= 4; ++l) w.n[l] = w.n[l - 1] / 4;
    for (int l = 0; l <= 4; ++l) {
        const int rows = round_up(std::max(B * w.n[l], 1), ROW_PAD);
        w.xyz[l] = c.floats((size_t)rows * 4);
        // features of a level: SA output first, later overwritten by the FP output (both <= 512 wide); level 0 holds the raw features
        const int width = round_up(std::max({l == 0 ? e.c_feat : e.sa_out[l - 1], l < 4 ? e.fp[l].back().cout : 0, 1}), 32);
        w.ldf[l] = width;
        w.feat[l] = c.floats((size_t)rows * width);
    }
    size_t emax = 1, gmax = 1, xmax = 1, edges_max = 1;
    for (int l = 0; l < 4; ++l) {
        const size_t edges = round_up(std::max(B * w.n[l + 1] * fc_paconv::K, 1), ROW_PAD);
        edges_max = std::max(edges_max, edges);
        for (const auto& L : e.sa[l]) {
            emax = std::max(emax, edges * (size_t)round_up(2 * L.cin, 32));
            gmax = std::max(gmax, edges * (size_t)(8 * L.cout));
        }
        const size_t rows = round_up(std::max(B * w.n[l], 1), ROW_PAD);
        for (const auto& L : e.fp[l]) xmax = std::max(xmax, rows * (size_t)std::max(L.lin.K_pad, L.lin.N_pad));
    }
    w.fidx = (int32_t*)c.bytes((size_t)B * std::max(w.n[1], 1) * sizeof(int32_t));
    w.nidx = (int32_t*)c.bytes(edges_max * sizeof(int32_t));
    w.fps_tmp = fps_scratch_floats(B, M) ? c.floats(fps_scratch_floats(B, M)) : nullptr;     // > 8192 points per scene: global min-distance array
    w.gdiff = c.floats(edges_max * 4);
    w.scores = c.floats(edges_max * 8);
    w.Ea = c.floats(emax);
    w.Eb = c.floats(emax);
    w.G = c.floats(gmax);
    w.X = c.floats(xmax);
    w.Ya = c.floats(xmax);
    w.Yb = c.floats(xmax);
    w.P_pad = round_up(B * M, ROW_PAD);
    for (int i = 0; i < 3; ++i) w.h[i] = c.floats((size_t)w.P_pad * e.H_pad);
    w.otmp = c.floats((size_t)w.P_pad * e.E_pad);
    if (need) *need = c.off + 256;
    return w;
}

static void paconv_forward(fc_paconv& e, const float* pts, float* out, int B, int M, void* ws, size_t bytes, hipStream_t s) {
    if (!pts || !out || B < 1) throw Error(FC_ERR_INVALID, "fc_paconv_embed_f32: bad argument");
    if (M < 256) throw Error(FC_ERR_INVALID, "PAConv embedder needs at least 256 context points (four 4x farthest-point down-samplings)");
    constexpr int K = fc_paconv::K;
    PaWs w = plan_pa(e, B, M, ws, bytes, false, nullptr);
    const int ldp = 3 + e.c_feat;
    // level 0: xyz (pitch 4) and raw features (pitch 32)
    launch_fill(w.xyz[0], 0.f, (size_t)round_up(B * M, ROW_PAD) * 4, s);
    launch_pack_rows(pts, ldp, 3, w.xyz[0], 4, 0, 4, B * M, s);
    launch_fill(w.feat[0], 0.f, (size_t)round_up(B * M, ROW_PAD) * w.ldf[0], s);
    if (e.c_feat > 0) launch_pack_rows(pts + 3, ldp, e.c_feat, w.feat[0], w.ldf[0], 0, e.c_feat, B * M, s);
    int cprev = e.c_feat;
    // ---- set abstraction
    for (int l = 0; l < 4; ++l) {
        const int n = w.n[l], m = w.n[l + 1], edges = B * m * K, edges_pad = round_up(std::max(edges, 1), ROW_PAD);
        launch_fps(w.xyz[l], 4, w.fidx, B, n, m, w.fps_tmp, s);
        launch_gather_xyz(w.xyz[l], 4, w.fidx, w.xyz[l + 1], B, n, m, s);
        launch_knn_xyz(w.xyz[l], 4, w.xyz[l + 1], w.nidx, B, n, m, K, s);
        float* Ein = w.Ea;
        float* Eout = w.Eb;
        const int ldE0 = e.sa[l][0].bank.K_pad;
        launch_paconv_group(w.xyz[l], 4, w.feat[l], w.ldf[l], cprev, w.xyz[l + 1], w.nidx, Ein, ldE0, w.gdiff, B, n, m, K, s);
        for (size_t j = 0; j < e.sa[l].size(); ++j) {
            const PaLayer& L = e.sa[l][j];
            launch_scorenet(w.gdiff, L.score_w, w.scores, edges, s);
            GemmEpi g{};
            g.C = w.G; g.ldc = 8 * L.cout; g.rows_valid = edges;
            ASeg a{Ein, L.bank.K_pad};
            launch_gemm(L.bank, &a, edges_pad, g, EPI_LINEAR, s);
            const bool last = j + 1 == e.sa[l].size();
            if (!last) {
                launch_score_reduce(w.G, 8 * L.cout, w.scores, L.bn_s, L.bn_t, L.cout, K, Eout, e.sa[l][j + 1].bank.K_pad, 0, 0, B * m, s);
                std::swap(Ein, Eout);
            } else {
                launch_fill(w.feat[l + 1], 0.f, (size_t)round_up(std::max(B * m, 1), ROW_PAD) * w.ldf[l + 1], s);
                launch_score_reduce(w.G, 8 * L.cout, w.scores, L.bn_s, L.bn_t, L.cout, K, w.feat[l + 1], w.ldf[l + 1], 0, 1, B * m, s);
            }
        }
        cprev = e.sa_out[l];
    }
    // ---- feature propagation (level 3 <- 4, 2 <- 3, 1 <- 2, 0 <- 1); a level's features are replaced by the FP output
    int c_known = e.sa_out[3];
    for (int i = 3; i >= 0; --i) {
        const int nu = w.n[i], mk = w.n[i + 1], rows = B * nu, rows_pad = round_up(std::max(rows, 1), ROW_PAD);
        const int c_skip = i == 0 ? e.c_feat : e.sa_out[i - 1];
        const int ldX = e.fp[i][0].lin.K_pad;
        launch_three_nn_interp(w.xyz[i], 4, w.xyz[i + 1], 4, w.feat[i + 1], w.ldf[i + 1], c_known, w.feat[i], w.ldf[i], c_skip, w.X, ldX, B, nu, mk, s);
        const float* cur = w.X;
        int ldcur = ldX;
        for (size_t j = 0; j < e.fp[i].size(); ++j) {
            const FpLayer& L = e.fp[i][j];
            const bool last = j + 1 == e.fp[i].size();
            float* dst = last ? w.feat[i] : (cur == w.Ya ? w.Yb : w.Ya);
            const int ldd = last ? w.ldf[i] : L.lin.N_pad;
            if (last && w.ldf[i] < L.lin.N_pad) throw Error(FC_ERR_INVALID, "PAConv: level feature pitch too small");
            GemmEpi g{};
            g.C = dst; g.ldc = ldd; g.act = FC_ACT_RELU; g.rows_valid = rows;
            ASeg a{cur, ldcur};
            launch_gemm(L.lin, &a, rows_pad, g, EPI_LINEAR, s);
            cur = dst;
            ldcur = ldd;
        }
        c_known = e.fp[i].back().cout;
    }
    // ---- head MLP on level 0
    ASeg in{w.feat[0], w.ldf[0]};
    const int cur = run_mlp_hidden_generic(e.mlp, &in, nullptr, FC_ACT_GELU, w.h, e.H_pad, w.P_pad, s, B * M);
    GemmEpi g{};
    g.C = w.otmp; g.ldc = e.E_pad; g.rows_valid = B * M;
    ASeg a{w.h[cur], e.H_pad};
    launch_gemm(e.mlp.out_layer, &a, w.P_pad, g, EPI_LINEAR, s);
    launch_pack_rows(w.otmp, e.E_pad, e.E, out, e.E, 0, e.E, B * M, s);
}

}  // namespace fc


extern "C" {

int fc_paconv_create(const fc_tensor* tensors, int32_t n_tensors, fc_paconv** out) {
    FC_API_BEGIN
    if (!out) throw fc::Error(FC_ERR_INVALID, "fc_paconv_create: null out");
    *out = nullptr;
    std::unique_ptr<fc_paconv> e(new fc_paconv());
    fc::WeightTable wt(tensors, n_tensors);
    fc::build_paconv(*e, wt);
    e->fp16_flag = (int*)e->arena.alloc_floats(1);
    FC_HIP(hipDeviceSynchronize());
    *out = e.release();
    FC_API_END
}
void fc_paconv_destroy(fc_paconv* emb) { delete emb; }
int fc_paconv_out_dim(const fc_paconv* emb) { return emb ? emb->E : 0; }
int fc_paconv_workspace_bytes(const fc_paconv* emb, int32_t B, int32_t M, size_t* bytes) {
    FC_API_BEGIN
    if (!emb || !bytes || B < 1 || M < 1) throw fc::Error(FC_ERR_INVALID, "fc_paconv_workspace_bytes: bad argument");
    fc::plan_pa(*emb, B, M, nullptr, 0, true, bytes);
    FC_API_END
}
int fc_paconv_embed_f32(fc_paconv* emb, const float* pts, float* out, int32_t B, int32_t M, void* workspace, size_t workspace_bytes, void* stream) {
    FC_API_BEGIN
    if (!emb || !workspace) throw fc::Error(FC_ERR_INVALID, "fc_paconv_embed_f32: null handle / workspace");
    // fast split-fp16 GEMMs first; the whole pass is repeated with the bf16-limb GEMMs if an activation left fp16's range
    fc::run_fp16_guarded(emb->fp16_flag, (hipStream_t)stream, [=] {
        fc::paconv_forward(*emb, pts, out, B, M, workspace, workspace_bytes, (hipStream_t)stream);
    }, true);
    FC_API_END
}
int fc_op_fps_f32(const float* xyz, int32_t* idx, int32_t B, int32_t n, int32_t m, void* stream) {
    FC_API_BEGIN
    if (!xyz || !idx || B < 1 || n < 1 || m < 0) throw fc::Error(FC_ERR_INVALID, "fc_op_fps_f32: bad argument");
    /* (single-operator entry of the test / training paths: the one scratch array of the > 8192-point variant is allocated here, stream-ordered;
     * the engine's own calls carve it from the caller's workspace) */
    float* scratch = nullptr;
    const size_t nf = fc::fps_scratch_floats(B, n);
    if (nf) FC_HIP(hipMallocAsync((void**)&scratch, nf * sizeof(float), (hipStream_t)stream));
    try { fc::launch_fps(xyz, 3, idx, B, n, m, scratch, (hipStream_t)stream); }
    catch (...) { if (scratch) (void)hipFreeAsync(scratch, (hipStream_t)stream); throw; }
    if (scratch) FC_HIP(hipFreeAsync(scratch, (hipStream_t)stream));
    FC_API_END
}

}  // extern "C"
