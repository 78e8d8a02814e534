// fc_dgcnn: DGCNN context embedder (models/pytorch_gcn.py:50-188), eval mode, as a launch schedule.
//
// Per edge-conv level (C_in -> C_out, k neighbours):
//   reference:  knn -> gather -> cat(nbr - x, x) [B,2C,M,k] -> 1x1 conv -> BN2d -> LeakyReLU(0.2) -> max_k
//   here:       knn (fused distance + streaming top-k, no MxM matrix)
//               one GEMM  uv = f [W1' ; W2'-W1']^T + [0 ; t]      (conv is linear: W (nbr - x | x) = W1 nbr + (W2 - W1) x;
//                                                                  eval BN = per-channel scale s / shift t, s folded into both halves)
//               gather-max: out = LeakyReLU( max_j u[idx_j] + v )  (LeakyReLU is increasing, so it commutes with max)
//   which does k x fewer conv FLOPs and never builds the [B,2C,M,k] edge tensor.
// conv5 + BN1d fold into one GEMM with a LeakyReLU epilogue; out_mlp is the shared MLP runner (GELU).
#include <algorithm>
#include <cmath>
#include <memory>

#include "hostpack.h"

struct fc_dgcnn {
    int* fp16_flag = nullptr;   // device word raised by the split-fp16 GEMM loop on an activation >= 65504 (common.h: Fp16Guard)
    fc::DeviceArena arena;
    int k = 0, global_pool = 0, c_in = 0, E = 0, E_pad = 0, H_pad = 0;
    fc::PackedLinear level[4];
    int lvl_cin[4] = {0, 0, 0, 0}, lvl_cout[4] = {64, 64, 128, 256}, lvl_col[4] = {0, 64, 128, 256};
    fc::PackedLinear conv5;
    fc::PackedMLP mlp;
};

namespace fc {

static void bn_fold(const WeightTable& wt, const std::string& p, int C, VecD& s, VecD& t) {
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

static void build_dgcnn(fc_dgcnn& e, const WeightTable& wt) {
    int cin = -1;
    for (int l = 0; l < 4; ++l) {
        const std::string p = "conv" + std::to_string(l + 1);
        const HostTensor& w = wt.get(p + ".0.weight");
        const int co = e.lvl_cout[l];
        if (w.shape.size() != 4 || w.shape[0] != co || w.shape[2] != 1 || w.shape[3] != 1 || w.shape[1] % 2)
            throw Error(FC_ERR_SHAPE, p + ".0.weight: expected [C_out, 2*C_in, 1, 1]");
        const int ci = (int)w.shape[1] / 2;
        if (l == 0) { cin = ci; e.c_in = ci; if (ci > 32) throw Error(FC_ERR_UNSUPPORTED, "DGCNN input_dim > 32"); }
        else if (ci != e.lvl_cout[l - 1]) throw Error(FC_ERR_SHAPE, p + ".0.weight: C_in does not chain");
        e.lvl_cin[l] = ci;
        VecD s, t;
        bn_fold(wt, p + ".1", co, s, t);
        MatD W = mat_from(w);                       // [co][2ci] : first ci columns act on (nbr - x), last ci on x
        MatD F(2 * co, ci);
        VecD b(2 * co, 0.0);
        for (int o = 0; o < co; ++o) {
            for (int c = 0; c < ci; ++c) {
                F.at(o, c) = s[o] * W.at(o, c);
                F.at(co + o, c) = s[o] * (W.at(o, ci + c) - W.at(o, c));
            }
            b[co + o] = t[o];
        }
        e.level[l] = pack_linear(e.arena, F, b, {}, map_prefix(2 * co, 2 * co), map_prefix(ci, round_up(ci, 32)), {round_up(ci, 32)});
    }
    (void)cin;
    {
        const HostTensor& w = wt.get("conv5.0.weight", {512, 512, 1});
        VecD s, t;
        bn_fold(wt, "conv5.1", 512, s, t);
        MatD W = mat_from(w);
        for (int o = 0; o < 512; ++o) for (int c = 0; c < 512; ++c) W.at(o, c) *= s[o];
        e.conv5 = pack_linear(e.arena, W, t, {}, map_prefix(512, 512), map_prefix(512, 512), {512});
    }
    const int in_w = e.global_pool ? 1024 : 512;
    pack_mlp_mid(e.arena, wt, "out_mlp", e.mlp);
    {
        const HostTensor& w = wt.get("out_mlp.in_layer.weight");
        if (w.shape[1] != in_w) throw Error(FC_ERR_SHAPE, "out_mlp.in_layer.weight: expected input width " + std::to_string(in_w));
        const int n = (int)w.shape[0];
        e.mlp.in_layer = pack_linear(e.arena, mat_from(w), vec_from(wt.get("out_mlp.in_layer.bias", {n})), {}, map_prefix(n, round_up(n, 32)),
                                     map_prefix(in_w, in_w), {in_w});
        const HostTensor& wo = wt.get("out_mlp.out_layer.weight");
        const int hl = e.mlp.sizes.back();
        e.E = (int)wo.shape[0];
        e.E_pad = round_up(e.E, 32);
        e.mlp.out_layer = pack_linear(e.arena, mat_from(wo), vec_from(wt.get("out_mlp.out_layer.bias", {e.E})), {}, map_prefix(e.E, e.E_pad),
                                      map_prefix(hl, round_up(hl, 32)), {round_up(hl, 32)});
    }
    e.H_pad = std::max(max_hidden_pad(e.mlp), 32);
}

struct DgWs {
    float *pts, *cat, *uv, *t5, *h[3], *pool, *otmp;
    int32_t* idx;
    int P, P_pad, R_pad;     // R = rows of the MLP stage (P per-point, B global)
};
static DgWs plan_dg(const fc_dgcnn& e, int B, int M, void* ws, size_t bytes, bool dry, size_t* need) {
    DgWs w{};
    w.P = B * M; w.P_pad = round_up(w.P, ROW_PAD);
    w.R_pad = e.global_pool ? round_up(B, ROW_PAD) : w.P_pad;
    WsCarver c(ws, bytes, dry);
    w.pts = c.floats((size_t)w.P_pad * 32);
    w.cat = c.floats((size_t)w.P_pad * 512);
    w.uv = c.floats((size_t)w.P_pad * 512);
    w.t5 = c.floats((size_t)w.P_pad * 512);
    w.idx = (int32_t*)c.bytes((size_t)w.P * e.k * sizeof(int32_t));
    for (int i = 0; i < 3; ++i) w.h[i] = c.floats((size_t)w.R_pad * e.H_pad);
    w.pool = c.floats(e.global_pool ? (size_t)w.R_pad * 1024 : 1);
    w.otmp = c.floats((size_t)w.R_pad * e.E_pad);
    if (need) *need = c.off + 256;
    return w;
}

static void dgcnn_forward(fc_dgcnn& e, const float* pts, float* out, int B, int M, void* ws, size_t bytes, hipStream_t s) {
    if (!pts || !out || B < 1 || M < 1) throw Error(FC_ERR_INVALID, "fc_dgcnn_embed_f32: bad argument");
    if (M < e.k) throw Error(FC_ERR_INVALID, "fewer context points than n_neighbors (torch.topk raises in the reference)");
    DgWs w = plan_dg(e, B, M, ws, bytes, false, nullptr);
    launch_fill(w.pts, 0.f, (size_t)w.P_pad * 32, s);
    launch_pack_rows(pts, e.c_in, e.c_in, w.pts, 32, 0, 32, w.P, s);
    if (w.P_pad > w.P) launch_fill(w.cat + (size_t)w.P * 512, 0.f, (size_t)(w.P_pad - w.P) * 512, s);
    for (int l = 0; l < 4; ++l) {
        const float* f = l == 0 ? w.pts : w.cat + e.lvl_col[l - 1];
        const int ldf = l == 0 ? 32 : 512;
        launch_knn(f, ldf, e.lvl_cin[l], w.idx, B, M, M, e.k, s, l > 0 ? w.idx : nullptr);      // (levels 1-3: warm start from the previous level's sets, in place)
        GemmEpi g{};
        g.C = w.uv; g.ldc = 512;
        ASeg a{f, ldf};
        launch_gemm(e.level[l], &a, w.P_pad, g, EPI_LINEAR, s);
        launch_gather_max(w.uv, 512, e.lvl_cout[l], w.idx, e.k, w.cat, 512, e.lvl_col[l], B, M, M, s);
    }
    {
        GemmEpi g{};
        g.C = w.t5; g.ldc = 512; g.act = FC_ACT_LRELU02;
        ASeg a{w.cat, 512};
        launch_gemm(e.conv5, &a, w.P_pad, g, EPI_LINEAR, s);
    }
    ASeg in{w.t5, 512};
    int rows = w.P, rows_pad = w.P_pad;
    if (e.global_pool) {
        launch_fill(w.pool, 0.f, (size_t)w.R_pad * 1024, s);
        launch_pool_max_mean(w.t5, 512, 512, w.pool, 1024, B, M, M, s);
        in = {w.pool, 1024};
        rows = B; rows_pad = w.R_pad;
    }
    const int cur = run_mlp_hidden_generic(e.mlp, &in, nullptr, FC_ACT_GELU, w.h, e.H_pad, rows_pad, s, rows);
    GemmEpi g{};
    g.C = w.otmp; g.ldc = e.E_pad;
    ASeg a{w.h[cur], e.H_pad};
    launch_gemm(e.mlp.out_layer, &a, rows_pad, g, EPI_LINEAR, s);
    launch_pack_rows(w.otmp, e.E_pad, e.E, out, e.E, 0, e.E, rows, s);
}

const char* get_last_error();
}  // namespace fc


extern "C" {

int fc_dgcnn_create(int32_t n_neighbors, int32_t global_pool, const fc_tensor* tensors, int32_t n_tensors, fc_dgcnn** out) {
    FC_API_BEGIN
    if (!out) throw fc::Error(FC_ERR_INVALID, "fc_dgcnn_create: null out");
    *out = nullptr;
    if (n_neighbors < 1 || n_neighbors > 64) throw fc::Error(FC_ERR_UNSUPPORTED, "n_neighbors must be in [1, 64]");
    std::unique_ptr<fc_dgcnn> e(new fc_dgcnn());
    e->k = n_neighbors;
    e->global_pool = global_pool ? 1 : 0;
    fc::WeightTable wt(tensors, n_tensors);
    fc::build_dgcnn(*e, wt);
    e->fp16_flag = (int*)e->arena.alloc_floats(1);
    FC_HIP(hipDeviceSynchronize());
    *out = e.release();
    FC_API_END
}
void fc_dgcnn_destroy(fc_dgcnn* emb) { delete emb; }
int fc_dgcnn_out_dim(const fc_dgcnn* emb) { return emb ? emb->E : 0; }
int fc_dgcnn_workspace_bytes(const fc_dgcnn* emb, int32_t B, int32_t M, size_t* bytes) {
    FC_API_BEGIN
    if (!emb || !bytes || B < 1 || M < 1) throw fc::Error(FC_ERR_INVALID, "fc_dgcnn_workspace_bytes: bad argument");
    fc::plan_dg(*emb, B, M, nullptr, 0, true, bytes);
    FC_API_END
}
int fc_dgcnn_embed_f32(fc_dgcnn* emb, const float* pts, float* out, int32_t B, int32_t M, void* workspace, size_t workspace_bytes, void* stream) {
    FC_API_BEGIN
    if (!emb || !workspace) throw fc::Error(FC_ERR_INVALID, "fc_dgcnn_embed_f32: null handle / workspace");
    // fast split-fp16 GEMMs first; the whole pass is repeated with the bf16-limb GEMMs if an activation left fp16's range
    fc::run_fp16_guarded(emb->fp16_flag, (hipStream_t)stream, [=] {
        fc::dgcnn_forward(*emb, pts, out, B, M, workspace, workspace_bytes, (hipStream_t)stream);
    }, true);
    FC_API_END
}

}  // extern "C"
