// fc_flow: the conditional normalizing flow (augmenter -> n x [pre-conditioner -> coupling -> ActNorm -> permuter]
// -> base density) as a schedule of HIP kernel launches over a caller-owned workspace.
//
// Reference being replaced: models.Flow.log_prob (models/transform.py:70-76) over the transform list that
// initialize_flow assembles (model_initialization.py:136-160).  Weight folding done once at create (double precision):
//   * attn.fn.lin (I -> attn_dim) is folded INTO the coupling / augmenter in_layer:  W_ctx (W_lin a + b_lin) = (W_ctx W_lin) a + W_ctx b_lin
//   * LayerNorm gamma/beta, the softmax scale inner^-0.5 and log2(e) are folded into the q projection
//   * ActNorm and the permuter (LinearLU: L U; FullCombiner: w; ExponentialCombiner: expm; Permuter: P) become ONE matrix
//     z = W' x + b',  W' = P diag(e^-log_scale),  b' = -W' shift; their log-dets are data independent and summed into one constant
//   * extra context (one scalar per scene) enters every in_layer as a rank-1 epilogue term instead of a concatenated column
// Activation layout in HBM: x is [rows, d1_pad + d2_pad] = [x1 | 0-pad | x2 | 0-pad] (pads kept zero by construction), every
// other activation is [rows, round_up(width, 32)]; rows are padded to 256.
#include <algorithm>
#include <cmath>
#include <memory>
#include <cstring>

#include "hostpack.h"

namespace fc {

struct AttnPack {
    PackedLinear q;      // LN-folded, pre-scaled q projection  [I_pad][A_in_pad]
    MatD lin_w;          // [attn_dim][I]  (folded into the consumer's in_layer)
    VecD lin_b;
    int kv_col = 0;      // column of this layer's [K | V] block inside the kv buffer
};

struct BlockPack {
    bool has_attn = false;
    PackedMLP pre;       // pre_attention_mlp
    AttnPack attn;
    PackedMLP net;       // coupling MLP (in_layer has the folded context segment)
    bool has_lin = false;
    PackedLinear lin;    // folded ActNorm + permuter (absent after the last block)
};

struct Dims {
    int Din, D, d1, d2, d1_pad, d2_pad, ldx;
    int E, E_pad, X;
    int A_in = 0, A_in_pad = 0, I = 0, I_pad = 0;
    int H_pad = 0;       // widest hidden activation
    int ldp = 0;         // spline parameter pitch
};

}  // namespace fc

struct fc_flow {
    fc_flow_config cfg;
    fc::Dims d;
    fc::DeviceArena arena;
    bool has_augment = false;
    fc::PackedMLP aug_pre, aug_net;
    fc::AttnPack aug_attn;
    std::vector<fc::BlockPack> blocks;
    fc::PackedLinear kv_all;   // ctx -> [K|V] of every attention (augmenter first)
    int n_attn = 0;
    double log_const = 0.0;
};

namespace fc {

static int pad_inner(int I) {
    if (I <= 32) return 32;
    if (I <= 64) return 64;
    if (I <= 128) return 128;
    throw Error(FC_ERR_UNSUPPORTED, "attention inner dim (cross_heads*cross_dim_head) > 128 is not supported yet");
}

// q' = c * Wq (gamma . n + beta),  c = inner^-0.5 * log2(e)   (models/perceiver.py:18-26, 96-110)
static void build_attn(fc_flow& f, const WeightTable& wt, const std::string& p, AttnPack& out, std::vector<MatD>& kv_rows) {
    Dims& d = f.d;
    const HostTensor& wq_t = wt.get(p + ".fn.attention.to_q.weight");
    if (wq_t.shape.size() != 2) throw Error(FC_ERR_SHAPE, p + ".fn.attention.to_q.weight must be 2-D");
    const int I = (int)wq_t.shape[0], A_in = (int)wq_t.shape[1];
    if (d.I == 0) { d.I = I; d.I_pad = pad_inner(I); d.A_in = A_in; d.A_in_pad = round_up(A_in, 32); }
    if (I != d.I || A_in != d.A_in) throw Error(FC_ERR_SHAPE, p + ": all attention blocks must share inner / input dims");
    MatD wq = mat_from(wq_t);
    VecD gamma = vec_from(wt.get(p + ".norm.weight", {A_in})), beta = vec_from(wt.get(p + ".norm.bias", {A_in}));
    const double c = std::pow((double)I, -0.5) * 1.4426950408889634074;
    VecD bq(I, 0.0);
    for (int i = 0; i < I; ++i)
        for (int k = 0; k < A_in; ++k) {
            bq[i] += c * wq.at(i, k) * beta[k];
            wq.at(i, k) *= c * gamma[k];
        }
    out.q = pack_linear(f.arena, wq, bq, {}, map_prefix(I, d.I_pad), map_prefix(A_in, d.A_in_pad), {d.A_in_pad});
    const HostTensor& wkv_t = wt.get(p + ".fn.attention.to_kv.weight", {2 * I, d.E});
    MatD wkv = mat_from(wkv_t);                 // rows [0,I) = K, [I,2I) = V  (chunk(2, dim=-1))
    MatD blk(2 * d.I_pad, d.E);
    for (int i = 0; i < I; ++i)
        for (int k = 0; k < d.E; ++k) { blk.at(i, k) = wkv.at(i, k); blk.at(d.I_pad + i, k) = wkv.at(I + i, k); }
    out.kv_col = (int)kv_rows.size() * 2 * d.I_pad;
    kv_rows.push_back(blk);
    const HostTensor& wl = wt.get(p + ".fn.lin.weight");
    if (wl.shape.size() != 2 || wl.shape[1] != I) throw Error(FC_ERR_SHAPE, p + ".fn.lin.weight: expected [attn_dim, inner]");
    out.lin_w = mat_from(wl);
    out.lin_b = vec_from(wt.get(p + ".fn.lin.bias", {wl.shape[0]}));
}

// in_layer over cat(first(n_first), extra(X), ctxvec(C)) -> packed [first_pad | second_pad] + rank-1 extra column.
// With attention the context segment is the folded attention output (I_pad wide); in global mode it is the E-wide embedding.
static PackedLinear build_in_layer(fc_flow& f, const WeightTable& wt, const std::string& prefix, int n_first, int first_pad,
                                   const AttnPack* attn) {
    Dims& d = f.d;
    const HostTensor& wt_in = wt.get(prefix + ".in_layer.weight");
    MatD w = mat_from(wt_in);
    const int H = w.rows;
    VecD b = vec_from(wt.get(prefix + ".in_layer.bias", {H}));
    const int C = attn ? attn->lin_w.rows : d.E;
    if (w.cols != n_first + d.X + C) throw Error(FC_ERR_SHAPE, prefix + ".in_layer.weight: expected input width " +
                                                                  std::to_string(n_first + d.X + C) + ", got " + std::to_string(w.cols));
    const int c0 = n_first + d.X;
    const int second = attn ? d.I : d.E, second_pad = attn ? d.I_pad : d.E_pad;
    MatD fw(H, n_first + second);
    VecD colvec;
    if (d.X) colvec.assign(H, 0.0);
    for (int n = 0; n < H; ++n) {
        for (int k = 0; k < n_first; ++k) fw.at(n, k) = w.at(n, k);
        if (d.X) colvec[n] = w.at(n, n_first);
        if (attn) {
            for (int j = 0; j < d.I; ++j) {
                double s = 0;
                for (int c = 0; c < C; ++c) s += w.at(n, c0 + c) * attn->lin_w.at(c, j);
                fw.at(n, n_first + j) = s;
            }
            double sb = 0;
            for (int c = 0; c < C; ++c) sb += w.at(n, c0 + c) * attn->lin_b[c];
            b[n] += sb;
        } else {
            for (int c = 0; c < C; ++c) fw.at(n, n_first + c) = w.at(n, c0 + c);
        }
    }
    std::vector<int> k2(second_pad, -1);
    for (int j = 0; j < second; ++j) k2[j] = n_first + j;
    return pack_linear(f.arena, fw, b, colvec, map_prefix(H, round_up(H, 32)), map_concat({map_prefix(n_first, first_pad), k2}),
                       {first_pad, second_pad});
}

static PackedLinear build_plain(fc_flow& f, const WeightTable& wt, const std::string& name, int k_src, int k_pad) {
    const HostTensor& w = wt.get(name + ".weight");
    if (w.shape.size() != 2 || w.shape[1] != k_src) throw Error(FC_ERR_SHAPE, name + ".weight: unexpected input width");
    const int n = (int)w.shape[0];
    return pack_linear(f.arena, mat_from(w), vec_from(wt.get(name + ".bias", {n})), {}, map_prefix(n, round_up(n, 32)),
                       map_prefix(k_src, k_pad), {k_pad});
}

// ActNorm (models/act_norm.py:37-43) followed by the permuter (models/permuters.py) as one affine map on the x layout.
static void build_lin(fc_flow& f, const WeightTable& wt, int idx_actnorm, int idx_perm, BlockPack& blk) {
    const Dims& d = f.d;
    const int D = d.D;
    VecD shift(D, 0.0), ls(D, 0.0);
    if (idx_actnorm >= 0) {
        const std::string p = "transforms." + std::to_string(idx_actnorm);
        shift = vec_from(wt.get(p + ".shift", {1, D}));
        ls = vec_from(wt.get(p + ".log_scale", {1, D}));
        for (double v : ls) f.log_const -= v;
    }
    const std::string p = "transforms." + std::to_string(idx_perm);
    MatD Wp(D, D);
    switch (f.cfg.permuter_type) {
        case FC_PERM_LINEAR_LU: {
            const int ntri = D * (D - 1) / 2;
            const HostTensor& lo = wt.get(p + ".lower_entries", {ntri});
            const HostTensor& up = wt.get(p + ".upper_entries", {ntri});
            const HostTensor& ud = wt.get(p + ".unconstrained_upper_diag", {D});
            MatD L(D, D), U(D, D);
            int t = 0;
            for (int i = 0; i < D; ++i) { for (int j = 0; j < i; ++j) L.at(i, j) = lo.data[t++]; L.at(i, i) = 1.0; }
            t = 0;
            for (int i = 0; i < D; ++i) for (int j = i + 1; j < D; ++j) U.at(i, j) = up.data[t++];
            for (int i = 0; i < D; ++i) {
                const double dg = softplus_d(ud.data[i]) + (double)f.cfg.linear_lu_eps;
                U.at(i, i) = dg;
                f.log_const += std::log(dg);
            }
            Wp = matmul(L, U);               // z = L (U x)   (permuters.py:164-169)
            break;
        }
        case FC_PERM_RANDOM: {
            const HostTensor& pm = wt.get(p + ".permutation", {D});
            for (int i = 0; i < D; ++i) {
                const int src = (int)std::lround(pm.data[i]);
                if (src < 0 || src >= D) throw Error(FC_ERR_INVALID, p + ".permutation out of range");
                Wp.at(i, src) = 1.0;          // y = x.index_select(-1, permutation)
            }
            break;
        }
        case FC_PERM_FULL: {
            Wp = mat_from(wt.get(p + ".w", {D, D}));
            f.log_const += slogdet_abs(Wp);
            break;
        }
        case FC_PERM_EXPONENTIAL: {
            MatD w = mat_from(wt.get(p + ".w", {D, D}));
            const double sc = wt.get(p + ".scale", {1}).data[0], sh = wt.get(p + ".shift", {1}).data[0];
            const double rs = wt.get(p + ".rescale", {1}).data[0], rsh = wt.get(p + ".reshift", {1}).data[0];
            for (auto& e : w.v) e = rs * std::tanh(sc * e + sh) + rsh + 1e-8;
            for (int i = 0; i < D; ++i) f.log_const += w.at(i, i);
            Wp = expm_double(w);
            break;
        }
        default: throw Error(FC_ERR_INVALID, "unknown permuter_type");
    }
    VecD b(D, 0.0);
    for (int i = 0; i < D; ++i) {
        double s = 0;
        for (int k = 0; k < D; ++k) {
            Wp.at(i, k) *= std::exp(-ls[k]);
            s += Wp.at(i, k) * shift[k];
        }
        b[i] = -s;
    }
    const std::vector<int> xl = map_xlayout(d.d1, d.d1_pad, d.d2, d.d2_pad);
    blk.lin = pack_linear(f.arena, Wp, b, {}, xl, xl, {d.ldx});
    blk.has_lin = true;
}

static void build_out_layer(fc_flow& f, const WeightTable& wt, const std::string& prefix, PackedMLP& net) {
    const Dims& d = f.d;
    const int hl = net.sizes.back();
    const HostTensor& w = wt.get(prefix + ".out_layer.weight");
    const int n = (int)w.shape[0];
    VecD b = vec_from(wt.get(prefix + ".out_layer.bias", {n}));
    std::vector<int> nmap;
    if (f.cfg.flow_type == FC_FLOW_AFFINE) {
        if (n != 2 * d.d2) throw Error(FC_ERR_SHAPE, prefix + ".out_layer: affine coupling expects 2*(D - D/2) outputs");
        nmap = map_pairs(d.d2, d.d2);
    } else if (f.cfg.flow_type == FC_FLOW_SPLINE) {
        const int per = 3 * f.cfg.num_bins_spline + 1;
        if (n != per * d.d1) throw Error(FC_ERR_SHAPE, prefix + ".out_layer: spline coupling expects (3K+1)*(D/2) outputs");
        if (n != per * d.d2) throw Error(FC_ERR_UNSUPPORTED, "spline coupling with odd latent_dim fails in the reference too (reshape)");
        nmap = map_prefix(n, round_up(n, 32));
    } else {
        throw Error(FC_ERR_UNSUPPORTED, "ExponentialCoupling is not built yet");
    }
    net.out_layer = pack_linear(f.arena, mat_from(w), b, {}, nmap, map_prefix(hl, round_up(hl, 32)), {round_up(hl, 32)});
}

static void build_flow(fc_flow& f, const WeightTable& wt) {
    const fc_flow_config& c = f.cfg;
    Dims& d = f.d;
    if (c.struct_size != (int)sizeof(fc_flow_config)) throw Error(FC_ERR_INVALID, "fc_flow_config.struct_size mismatch (ABI)");
    if (c.latent_dim < c.input_dim) throw Error(FC_ERR_INVALID, "Latent dim < Input dim");
    if (c.cif_latent_dim < c.latent_dim) throw Error(FC_ERR_INVALID, "Augment dim smaller than main latent!");
    if (c.cif_latent_dim > c.latent_dim) throw Error(FC_ERR_UNSUPPORTED, "CIFblock (cif_latent_dim > latent_dim) is not built yet");
    if (c.n_flow_layers < 1 || c.latent_dim < 2) throw Error(FC_ERR_INVALID, "need n_flow_layers >= 1 and latent_dim >= 2");
    if (c.extra_context_dim < 0 || c.extra_context_dim > 1) throw Error(FC_ERR_UNSUPPORTED, "extra_context_dim must be 0 or 1");
    d.Din = c.input_dim; d.D = c.latent_dim; d.d1 = d.D / 2; d.d2 = d.D - d.d1;
    d.d1_pad = round_up(d.d1, 32); d.d2_pad = round_up(d.d2, 32); d.ldx = d.d1_pad + d.d2_pad;
    d.E = c.input_embedding_dim; d.E_pad = round_up(d.E, 32); d.X = c.extra_context_dim;
    if (d.Din > 32) throw Error(FC_ERR_UNSUPPORTED, "input_dim > 32");
    std::vector<MatD> kv_rows;

    // ---- transform 0: AugmentAttentionPreconditioner (models/augmenter.py:7-22) or IdentityTransform
    int idx = 1;
    f.has_augment = d.D > d.Din;
    if (f.has_augment) {
        const std::string p = "transforms.0";
        build_attn(f, wt, p + ".attn", f.aug_attn, kv_rows);
        pack_mlp_mid(f.arena, wt, p + ".pre_attn_mlp", f.aug_pre);
        f.aug_pre.in_layer = build_plain(f, wt, p + ".pre_attn_mlp.in_layer", d.Din, 32);
        f.aug_pre.out_layer = build_plain(f, wt, p + ".pre_attn_mlp.out_layer", f.aug_pre.sizes.back(), round_up(f.aug_pre.sizes.back(), 32));
        if (f.aug_pre.out_layer.N_pad != d.A_in_pad) throw Error(FC_ERR_SHAPE, "pre_attn_mlp output width != attn_input_dim");
        const std::string pn = p + ".augment.noise_dist.net";
        pack_mlp_mid(f.arena, wt, pn, f.aug_net);
        f.aug_net.in_layer = build_in_layer(f, wt, pn, d.Din, 32, &f.aug_attn);
        const int nz = d.D - d.Din, hl = f.aug_net.sizes.back();
        const HostTensor& wo = wt.get(pn + ".out_layer.weight", {2 * nz, hl});
        f.aug_net.out_layer = pack_linear(f.arena, mat_from(wo), vec_from(wt.get(pn + ".out_layer.bias", {2 * nz})), {}, map_pairs(nz, nz),
                                          map_prefix(hl, round_up(hl, 32)), {round_up(hl, 32)});
        d.H_pad = std::max({d.H_pad, max_hidden_pad(f.aug_pre), max_hidden_pad(f.aug_net), d.A_in_pad});
    }
    // ---- blocks
    f.blocks.resize(c.n_flow_layers);
    for (int l = 0; l < c.n_flow_layers; ++l) {
        BlockPack& b = f.blocks[l];
        const std::string p = "transforms." + std::to_string(idx++);
        b.has_attn = !c.global_context;
        if (b.has_attn) {
            build_attn(f, wt, p + ".pre_conditioner.attn", b.attn, kv_rows);
            const std::string pp = p + ".pre_conditioner.pre_attention_mlp";
            pack_mlp_mid(f.arena, wt, pp, b.pre);
            b.pre.in_layer = build_plain(f, wt, pp + ".in_layer", d.d1, d.d1_pad);
            b.pre.out_layer = build_plain(f, wt, pp + ".out_layer", b.pre.sizes.back(), round_up(b.pre.sizes.back(), 32));
            if (b.pre.out_layer.N_pad != d.A_in_pad) throw Error(FC_ERR_SHAPE, "pre_attention_mlp output width != attn_input_dim");
            d.H_pad = std::max({d.H_pad, max_hidden_pad(b.pre), d.A_in_pad});
        }
        const std::string pn = p + ".transform.nn";
        pack_mlp_mid(f.arena, wt, pn, b.net);
        b.net.in_layer = build_in_layer(f, wt, pn, d.d1, d.d1_pad, b.has_attn ? &b.attn : nullptr);
        build_out_layer(f, wt, pn, b.net);
        d.H_pad = std::max(d.H_pad, max_hidden_pad(b.net));
        if (c.flow_type == FC_FLOW_SPLINE) d.ldp = std::max(d.ldp, b.net.out_layer.N_pad);
        if (l != c.n_flow_layers - 1) {
            const int ia = c.act_norm ? idx++ : -1;
            const int ip = idx++;
            build_lin(f, wt, ia, ip, b);
        }
    }
    // ---- one stacked K|V projection for every attention
    f.n_attn = (int)kv_rows.size();
    if (f.n_attn) {
        MatD all(f.n_attn * 2 * d.I_pad, d.E);
        for (int a = 0; a < f.n_attn; ++a) std::copy(kv_rows[a].v.begin(), kv_rows[a].v.end(), all.v.begin() + (size_t)a * 2 * d.I_pad * d.E);
        f.kv_all = pack_linear(f.arena, all, {}, {}, map_prefix(all.rows, all.rows), map_prefix(d.E, d.E_pad), {d.E_pad});
    }
}

// ---------------------------------------------------------------- workspace plan
struct FlowWs {
    float *xa, *xb, *h[3], *q, *a, *ctxp, *kv, *xin, *rowscal, *spl;
    int P, P_pad, Pc, Pc_pad, ldkv;
};
static FlowWs plan_ws(const fc_flow& f, int B, int N, int M, void* ws, size_t bytes, bool dry, size_t* need) {
    const Dims& d = f.d;
    FlowWs w{};
    w.P = B * N; w.P_pad = round_up(w.P, ROW_PAD);
    w.Pc = B * M; w.Pc_pad = round_up(w.Pc, ROW_PAD);
    w.ldkv = f.n_attn * 2 * d.I_pad;
    WsCarver c(ws, bytes, dry);
    w.xa = c.floats((size_t)w.P_pad * d.ldx);
    w.xb = c.floats((size_t)w.P_pad * d.ldx);
    for (int i = 0; i < 3; ++i) w.h[i] = c.floats((size_t)w.P_pad * std::max(d.H_pad, 32));
    w.q = c.floats((size_t)w.P_pad * std::max(d.I_pad, 32));
    w.a = c.floats((size_t)w.P_pad * std::max(d.I_pad, 32));
    w.ctxp = c.floats((size_t)w.Pc_pad * d.E_pad);
    w.kv = c.floats((size_t)w.Pc_pad * std::max(w.ldkv, 32));
    w.xin = c.floats((size_t)w.P_pad * 32);
    w.rowscal = c.floats((size_t)w.P_pad);
    w.spl = c.floats(d.ldp ? (size_t)w.P_pad * d.ldp : 1);
    if (need) *need = c.off + 256;
    return w;
}

static int run_mlp_hidden(const fc_flow& f, const PackedMLP& m, const ASeg* in_segs, const float* rowscal, FlowWs& w, int rows, hipStream_t s) {
    return run_mlp_hidden_generic(m, in_segs, rowscal, f.cfg.nonlinearity, w.h, std::max(f.d.H_pad, 32), rows, s, w.P);
}

// pre-conditioner: pre-MLP -> LayerNorm -> q -> attention; result in w.a  (models/cif_block.py:14-20 / augmenter.py:15-16)
static void run_attention(const fc_flow& f, const PackedMLP& pre, const AttnPack& at, const ASeg& in, FlowWs& w, int B, int N, int M, hipStream_t s) {
    const Dims& d = f.d;
    const int ldh = std::max(d.H_pad, 32);
    const int cur = run_mlp_hidden(f, pre, &in, nullptr, w, w.P_pad, s);
    int o = 0;
    while (o == cur) ++o;
    GemmEpi e{};
    e.act = FC_ACT_NONE; e.C = w.h[o]; e.ldc = ldh; e.rows_valid = w.P;
    ASeg a{w.h[cur], ldh};
    launch_gemm(pre.out_layer, &a, w.P_pad, e, EPI_LINEAR, s);
    launch_layernorm(w.h[o], ldh, d.A_in, w.P, s);
    GemmEpi eq{};
    eq.act = FC_ACT_NONE; eq.C = w.q; eq.ldc = d.I_pad; eq.rows_valid = w.P;
    ASeg aq{w.h[o], ldh};
    launch_gemm(at.q, &aq, w.P_pad, eq, EPI_LINEAR, s);
    launch_attention(w.q, d.I_pad, w.kv + at.kv_col, w.ldkv, w.kv + at.kv_col + d.I_pad, w.ldkv, w.a, d.I_pad, B, N, N, M, M, d.I_pad, s);
}

static void flow_forward(fc_flow& f, const float* x, const float* ctx, const float* extra, const float* const* eps, int n_eps,
                         float* logprob, float* z_out, int B, int N, int M, void* ws, size_t ws_bytes, hipStream_t s) {
    const Dims& d = f.d;
    const fc_flow_config& c = f.cfg;
    if (B < 1 || N < 1 || M < 1) throw Error(FC_ERR_INVALID, "B, N, M must be positive");
    if (!x || !ctx || !logprob) throw Error(FC_ERR_INVALID, "null x / ctx / logprob");
    if (d.X && !extra) throw Error(FC_ERR_INVALID, "this flow was built with extra context: extra must not be NULL");
    if (c.global_context && M != N) throw Error(FC_ERR_INVALID, "global context is per target point: ctx must be [B,N,E] (M == N)");
    if (n_eps != (f.has_augment ? 1 : 0) || (n_eps && (!eps || !eps[0]))) throw Error(FC_ERR_INVALID, "wrong number of noise tensors");
    FlowWs w = plan_ws(f, B, N, M, ws, ws_bytes, false, nullptr);
    const int ldh = std::max(d.H_pad, 32);
    const float* rowscal = nullptr;

    launch_fill(logprob, 0.f, (size_t)w.P, s);
    launch_pack_rows(ctx, d.E, d.E, w.ctxp, d.E_pad, 0, d.E_pad, w.Pc, s);
    if (d.X) { launch_repeat_extra(extra, d.X, w.rowscal, B, N, s); rowscal = w.rowscal; }
    if (f.n_attn) {
        GemmEpi e{};
        e.C = w.kv; e.ldc = w.ldkv; e.rows_valid = w.Pc;
        ASeg a{w.ctxp, d.E_pad};
        launch_gemm(f.kv_all, &a, w.Pc_pad, e, EPI_LINEAR, s);
    }
    float* xc = w.xa;
    float* xn = w.xb;
    launch_fill(xc, 0.f, (size_t)w.P_pad * d.ldx, s);
    if (f.has_augment) {
        launch_pack_rows(x, d.Din, d.Din, w.xin, 32, 0, 32, w.P, s);
        const int n1 = std::min(d.Din, d.d1);                                       // latent[0:Din] = x, split over the x1 | x2 regions
        launch_pack_rows(x, d.Din, n1, xc, d.ldx, 0, n1, w.P, s);
        if (d.Din > n1) launch_pack_rows(x + n1, d.Din, d.Din - n1, xc, d.ldx, d.d1_pad, d.Din - n1, w.P, s);
        ASeg in{w.xin, 32};
        run_attention(f, f.aug_pre, f.aug_attn, in, w, B, N, M, s);
        ASeg segs[2] = {{w.xin, 32}, {w.a, d.I_pad}};
        const int cur = run_mlp_hidden(f, f.aug_net, segs, rowscal, w, w.P_pad, s);
        GemmEpi e{};
        e.xbuf = xc; e.ldx = d.ldx; e.d2 = d.D - d.Din; e.logprob = logprob; e.eps = eps[0];
        e.d_in = d.Din; e.d1 = d.d1; e.d1_pad = d.d1_pad; e.rows_valid = w.P;
        ASeg a{w.h[cur], ldh};
        launch_gemm(f.aug_net.out_layer, &a, w.P_pad, e, EPI_AUGMENT, s);
    } else {
        launch_pack_rows(x, d.D, d.d1, xc, d.ldx, 0, d.d1, w.P, s);
        launch_pack_rows(x + d.d1, d.D, d.d2, xc, d.ldx, d.d1_pad, d.d2, w.P, s);
    }
    for (int l = 0; l < c.n_flow_layers; ++l) {
        const BlockPack& b = f.blocks[l];
        ASeg segs[2];
        segs[0] = {xc, d.ldx};
        if (b.has_attn) {
            run_attention(f, b.pre, b.attn, segs[0], w, B, N, M, s);
            segs[1] = {w.a, d.I_pad};
        } else {
            segs[1] = {w.ctxp, d.E_pad};
        }
        const int cur = run_mlp_hidden(f, b.net, segs, rowscal, w, w.P_pad, s);
        ASeg a{w.h[cur], ldh};
        if (c.flow_type == FC_FLOW_AFFINE) {
            GemmEpi e{};
            e.xbuf = xc; e.ldx = d.ldx; e.x2_col0 = d.d1_pad; e.d2 = d.d2; e.scale_fn = c.affine_scale_fn;
            e.logprob = logprob; e.rows_valid = w.P;
            launch_gemm(b.net.out_layer, &a, w.P_pad, e, EPI_AFFINE, s);
        } else {
            GemmEpi e{};
            e.C = w.spl; e.ldc = d.ldp; e.rows_valid = w.P;
            launch_gemm(b.net.out_layer, &a, w.P_pad, e, EPI_LINEAR, s);
            launch_spline(w.spl, d.ldp, xc, d.ldx, d.d1_pad, d.d2, c.num_bins_spline, logprob, w.P, 0, s);
        }
        if (b.has_lin) {
            GemmEpi e{};
            e.C = xn; e.ldc = d.ldx; e.rows_valid = w.P;
            ASeg ax{xc, d.ldx};
            launch_gemm(b.lin, &ax, w.P_pad, e, EPI_LINEAR, s);
            std::swap(xc, xn);
        }
    }
    launch_base_density(xc, d.ldx, d.d1, d.d1_pad, d.d2, logprob, (float)f.log_const, z_out, d.D, w.P, s);
}

}  // namespace fc

// ================================================================== C ABI
namespace fc { const char* get_last_error(); void prof_set(bool); void prof_reset(); std::string prof_report_json(); }

#define FC_API_BEGIN try {
#define FC_API_END                                                    \
    }                                                                 \
    catch (const fc::Error& e) { fc::set_last_error(e.what()); return e.code; }          \
    catch (const std::exception& e) { fc::set_last_error(e.what()); return FC_ERR_INVALID; } \
    return FC_OK;

extern "C" {

int fc_abi_version(void) { return FC_ABI_VERSION; }
const char* fc_last_error(void) { return fc::get_last_error(); }

int fc_profile_enable(int32_t on) { fc::prof_set(on != 0); return FC_OK; }
int fc_profile_reset(void) { fc::prof_reset(); return FC_OK; }
int fc_profile_report(char* buf, size_t cap) {
    FC_API_BEGIN
    const std::string r = fc::prof_report_json();
    if (!buf || cap < r.size() + 1) throw fc::Error(FC_ERR_INVALID, "fc_profile_report: buffer too small");
    memcpy(buf, r.c_str(), r.size() + 1);
    FC_API_END
}

int fc_flow_create(const fc_flow_config* cfg, const fc_tensor* tensors, int32_t n_tensors, fc_flow** out) {
    FC_API_BEGIN
    if (!cfg || !out) throw fc::Error(FC_ERR_INVALID, "fc_flow_create: null argument");
    *out = nullptr;
    std::unique_ptr<fc_flow> f(new fc_flow());
    f->cfg = *cfg;
    fc::WeightTable wt(tensors, n_tensors);
    fc::build_flow(*f, wt);
    FC_HIP(hipDeviceSynchronize());
    *out = f.release();
    FC_API_END
}

void fc_flow_destroy(fc_flow* flow) { delete flow; }

int fc_flow_workspace_bytes(const fc_flow* flow, int32_t B, int32_t N, int32_t M, size_t* bytes) {
    FC_API_BEGIN
    if (!flow || !bytes || B < 1 || N < 1 || M < 1) throw fc::Error(FC_ERR_INVALID, "fc_flow_workspace_bytes: bad argument");
    fc::plan_ws(*flow, B, N, M, nullptr, 0, true, bytes);
    FC_API_END
}

int fc_flow_noise_count(const fc_flow* flow) { return flow ? (flow->has_augment ? 1 : 0) : 0; }
int fc_flow_noise_width(const fc_flow* flow, int32_t i) {
    if (!flow || i != 0 || !flow->has_augment) return 0;
    return flow->d.D - flow->d.Din;
}

int fc_flow_logprob_f32(fc_flow* flow, const float* x, const float* ctx, const float* extra, const float* const* eps, int32_t n_eps,
                        float* logprob, float* z_out, int32_t B, int32_t N, int32_t M, void* workspace, size_t workspace_bytes, void* stream) {
    FC_API_BEGIN
    if (!flow || !workspace) throw fc::Error(FC_ERR_INVALID, "fc_flow_logprob_f32: null flow / workspace");
    fc::flow_forward(*flow, x, ctx, extra, eps, n_eps, logprob, z_out, B, N, M, workspace, workspace_bytes, (hipStream_t)stream);
    FC_API_END
}

int fc_flow_inverse_f32(fc_flow*, const float*, const float*, const float*, const float* const*, int32_t, float*, int32_t, int32_t, int32_t,
                        void*, size_t, void*) {
    fc::set_last_error("fc_flow_inverse_f32: the sampling path (SURVEY.md §8f N2) is not built yet");
    return FC_ERR_UNSUPPORTED;
}

}  // extern "C"
