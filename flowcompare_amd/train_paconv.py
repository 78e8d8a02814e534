"""PAConv context embedder in TRAINING mode, differentiable through HIP kernels.

Mirrors PointNet2SSGSeg.forward (models/scene_seg_PAConv/model/pointnet2/pointnet2_paconv_seg.py:63-82) with BatchNorm batch statistics:
four set-abstraction levels (pointnet2_paconv_modules.py:20-61: farthest point sampling to n/4, sorted 32-NN grouping, three PAConv
layers -- paconv.py:107-153: ScoreNet softmax over the 8 weight-bank kernels, kernel_input 'neighbor' [x - x_centre | x], weight-bank
product, assign_score, BatchNorm2d, ReLU -- and the max over the 32 neighbours), four feature-propagation levels
(pointnet2_paconv_modules.py:206-238: 3-NN inverse-distance interpolation, skip concat, 1x1 conv + BatchNorm2d + ReLU) and the head MLP.

What runs where: index kernels (FPS, k-NN, 3-NN) are the inference kernels, indices carry no gradient; every product (weight bank,
ScoreNet convs, FP convs, head) is the training Linear of train_ops.py; BatchNorm (+ ReLU, + max over the neighbours) is the EdgeConv
BatchNorm kernel family with slope 0; softmax, assign_score, the centre difference, the gathered-row gradients and the interpolation
are csrc/train_paconv.hip.  torch on activations: padding / views (panels), nothing else; torch on indices: argsort / bincount of the
edge lists for the fixed-order gather backwards.
"""
import ctypes

import torch

from . import engine
from . import train_ops as T

K_NEIGHBOURS = 32
M_KERNELS = 8


def _r32(x):
    return (x + 31) // 32 * 32


def _sorted_edges(src_rows, n_src):
    """edge ids sorted (stable) by the source row they read + start offset of each source row's segment (index plumbing)."""
    flat = src_rows.reshape(-1).long()
    order = torch.argsort(flat, stable=True).to(torch.int32)
    offsets = torch.zeros(n_src + 1, dtype=torch.int32, device=flat.device)
    offsets[1:] = torch.cumsum(torch.bincount(flat, minlength=n_src), 0).to(torch.int32)
    return order, offsets


class GroupFn(torch.autograd.Function):
    """QueryAndGroup(use_xyz) + the first PAConv layer's kernel input of a level: feat panel [B*n (padded), ldf] -> E panel
    [B*m*K (padded), round32(2 (C + 3))] = [x_e - x_0 | x_e] with x_e = [xyz[idx_e] - new_xyz | feat[idx_e]]; also returns gdiff
    [edges, 4] = xyz[idx_e] - xyz[idx_0] (no gradient: coordinates are data)."""

    @staticmethod
    def forward(ctx, feat, xyz, qxyz, nidx, C, B, n, m):
        L = engine.lib()
        K = nidx.shape[1]
        edges = B * m * K
        ldE = _r32(2 * (C + 3))
        E = T._panel_out(T._round_up(edges, T.ROW_PAD), ldE, edges, feat.device)
        gdiff = torch.zeros(T._round_up(edges, T.ROW_PAD), 4, dtype=torch.float32, device=feat.device)
        with T._OnDevice(feat.device):
            engine._check(L.fc_train_paconv_group_f32(engine._ptr(xyz), engine._ptr(feat), feat.shape[1], C, engine._ptr(qxyz), engine._ptr(nidx),
                                                      engine._ptr(E), ldE, engine._ptr(gdiff), B, n, m, K, engine._stream()))
        ctx.save_for_backward(nidx)
        ctx.meta = (C, B, n, m, K, feat.shape)
        ctx.mark_non_differentiable(gdiff)
        return E, gdiff

    @staticmethod
    def backward(ctx, dE, _dg):
        L = engine.lib()
        (nidx,) = ctx.saved_tensors
        C, B, n, m, K, fshape = ctx.meta
        dev = dE.device
        dE = dE.contiguous()
        Cin = C + 3
        edges = B * m * K
        dx = torch.empty(edges, _r32(Cin), dtype=torch.float32, device=dev)
        src = nidx.view(B, m * K) + (torch.arange(B, device=dev, dtype=torch.int32) * n)[:, None]
        order, offsets = _sorted_edges(src, B * n)
        dfeat = torch.empty(fshape, dtype=torch.float32, device=dev)
        with T._OnDevice(dev):
            s = engine._stream()
            engine._check(L.fc_train_centerdiff_bwd_f32(engine._ptr(dE), dE.shape[1], Cin, K, B * m, engine._ptr(dx), dx.shape[1], s))
            engine._check(L.fc_train_rows_gather_bwd_f32(engine._ptr(dx), dx.shape[1], 3, C, engine._ptr(order), engine._ptr(offsets), ctypes.c_void_p(0), 1,
                                                         B * n, fshape[0], engine._ptr(dfeat), fshape[1], s))
        return dfeat, None, None, None, None, None, None, None


class CenterDiffFn(torch.autograd.Function):
    """kernel_input 'neighbor' (paconv.py:118-123) on groups of K consecutive rows: x panel [groups*K (padded), >= C] -> [x_e - x_centre | x_e]."""

    @staticmethod
    def forward(ctx, x, C, K, groups):
        L = engine.lib()
        rows = groups * K
        E = T._panel_out(x.shape[0], _r32(2 * C), rows, x.device)
        with T._OnDevice(x.device):
            engine._check(L.fc_train_centerdiff_fwd_f32(engine._ptr(x), x.shape[1], C, K, groups, engine._ptr(E), E.shape[1], engine._stream()))
        ctx.meta = (C, K, groups, x.shape)
        return E

    @staticmethod
    def backward(ctx, dE):
        L = engine.lib()
        C, K, groups, xshape = ctx.meta
        dE = dE.contiguous()
        dx = T._panel_out(xshape[0], xshape[1], groups * K, dE.device)
        with T._OnDevice(dE.device):
            engine._check(L.fc_train_centerdiff_bwd_f32(engine._ptr(dE), dE.shape[1], C, K, groups, engine._ptr(dx), dx.shape[1], engine._stream()))
        return dx, None, None, None


class SoftmaxFn(torch.autograd.Function):
    """Row softmax over the first `width` columns of a panel (ScoreNet's softmax over the weight-bank kernels, paconv.py:50-53)."""

    @staticmethod
    def forward(ctx, x, width, rows):
        L = engine.lib()
        y = T._panel_out(x.shape[0], x.shape[1], rows, x.device)
        with T._OnDevice(x.device):
            engine._check(L.fc_train_softmax_fwd_f32(engine._ptr(x), x.shape[1], width, rows, engine._ptr(y), y.shape[1], engine._stream()))
        ctx.save_for_backward(y)
        ctx.meta = (width, rows)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = engine.lib()
        (y,) = ctx.saved_tensors
        width, rows = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        with T._OnDevice(y.device):
            engine._check(L.fc_train_softmax_bwd_f32(engine._ptr(y), y.shape[1], engine._ptr(dy), dy.shape[1], width, rows, y.shape[0], engine._ptr(dx),
                                                     dx.shape[1], engine._stream()))
        return dx, None, None


class AssignFn(torch.autograd.Function):
    """assign_score (util/paconv_util.py:52-56): out[e, o] = sum_m S[e, m] G[e, m Cout + o]; G panel [edges (padded), m Cout], S panel."""

    @staticmethod
    def forward(ctx, G, S, m, Cout, rows):
        L = engine.lib()
        out = torch.empty(G.shape[0], _r32(Cout), dtype=torch.float32, device=G.device)
        with T._OnDevice(G.device):
            engine._check(L.fc_train_assign_fwd_f32(engine._ptr(G), G.shape[1], engine._ptr(S), S.shape[1], m, Cout, rows, G.shape[0], engine._ptr(out),
                                                    out.shape[1], engine._stream()))
        ctx.save_for_backward(G, S)
        ctx.meta = (m, Cout, rows)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = engine.lib()
        G, S = ctx.saved_tensors
        m, Cout, rows = ctx.meta
        dout = dout.contiguous()
        dG, dS = torch.empty_like(G), torch.empty_like(S)
        with T._OnDevice(G.device):
            engine._check(L.fc_train_assign_bwd_f32(engine._ptr(G), G.shape[1], engine._ptr(S), S.shape[1], engine._ptr(dout), dout.shape[1], m, Cout, rows,
                                                    G.shape[0], engine._ptr(dG), dG.shape[1], engine._ptr(dS), dS.shape[1], engine._stream()))
        return dG, dS, None, None, None


class BNActMaxFn(torch.autograd.Function):
    """BatchNorm (batch statistics over rows*k values per channel) + LeakyReLU(slope) (+ max over groups of k consecutive rows when
    k > 1) of a dense panel P [rows*k (padded), >= C]: the EdgeConv BatchNorm kernels (csrc/train_edge.hip) with identity indices and
    no per-query term.  `bn`: the BatchNorm module whose running statistics are updated as torch does in train mode (first c_real
    channels: a panel may carry zero-padded channels, e.g. ScoreNet's 16 hidden units in a 32-wide panel)."""

    @staticmethod
    def forward(ctx, P, gamma, beta, rows, C, k, slope, bn, c_real):
        L = engine.lib()
        dev = P.device
        if C % 32 != 0 or P.shape[1] < C or P.shape[0] < rows * k:
            raise RuntimeError("BNActMaxFn: panel too small or channel count not a multiple of 32")
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        idx = None if k == 1 else torch.arange(rows * k, dtype=torch.int32, device=dev).view(rows, k)
        stats = torch.empty(3 * C, dtype=torch.float32, device=dev)
        out = T._panel_out(T._round_up(rows, T.ROW_PAD), C, rows, dev)
        arg = torch.empty(rows, C, dtype=torch.uint8, device=dev)
        with T._OnDevice(dev):
            s = engine._stream()
            nb = L.fc_train_edge_ws_bytes(rows, C)
            ws = T._ws(nb, dev)
            engine._check(L.fc_train_edge_stats_f32(engine._ptr(P), P.shape[1], ctypes.c_void_p(0), 0, engine._ptr(idx), rows, k, C, ctypes.c_float(bn.eps),
                                                    engine._ptr(stats), engine._ptr(ws), ctypes.c_size_t(nb), s))
            engine._check(L.fc_train_edge_fwd_f32(engine._ptr(P), P.shape[1], ctypes.c_void_p(0), 0, engine._ptr(idx), rows, k, C, engine._ptr(stats),
                                                  engine._ptr(g32), engine._ptr(b32), ctypes.c_float(slope), engine._ptr(out), C, engine._ptr(arg), s))
        if bn.track_running_stats and bn.running_mean is not None:
            with torch.no_grad():
                n = rows * k
                mom = bn.momentum if bn.momentum is not None else 0.1
                bn.running_mean.mul_(1 - mom).add_(stats[:c_real].to(bn.running_mean.dtype), alpha=mom)
                bn.running_var.mul_(1 - mom).add_(stats[2 * C:2 * C + c_real].to(bn.running_var.dtype) * (n / max(n - 1, 1)), alpha=mom)
                bn.num_batches_tracked += 1
        ctx.save_for_backward(P, g32, b32, stats, arg, idx)
        ctx.meta = (rows, C, k, slope, gamma.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        L = engine.lib()
        P, g32, b32, stats, arg, idx = ctx.saved_tensors
        rows, C, k, slope, pdtype = ctx.meta
        dev = P.device
        g = g.contiguous()
        rows_pad = g.shape[0]
        t1 = torch.empty(rows_pad, C, dtype=torch.float32, device=dev)
        t2 = torch.empty(rows_pad, C, dtype=torch.float32, device=dev)
        dP = torch.zeros_like(P)
        with T._OnDevice(dev):
            s = engine._stream()
            engine._check(L.fc_train_edge_bwd_prep_f32(engine._ptr(P), P.shape[1], ctypes.c_void_p(0), 0, engine._ptr(idx), rows, k, C, engine._ptr(stats),
                                                       engine._ptr(g32), engine._ptr(b32), ctypes.c_float(slope), engine._ptr(arg), engine._ptr(g), g.shape[1],
                                                       engine._ptr(t1), engine._ptr(t2), C, rows_pad, s))
            dbeta, dgamma = T._colsum(t1, C, rows), T._colsum(t2, C, rows)
            # identity indices: every row of P is the target of exactly one edge, so the "scatter" writes each element once (deterministic)
            engine._check(L.fc_train_edge_bwd_scatter_f32(engine._ptr(P), P.shape[1], ctypes.c_void_p(0), 0, engine._ptr(idx), rows, k, C, engine._ptr(stats),
                                                          engine._ptr(g32), engine._ptr(arg), engine._ptr(t1), C, engine._ptr(dbeta), engine._ptr(dgamma),
                                                          engine._ptr(dP), dP.shape[1], ctypes.c_void_p(0), 0, s))
        return dP, dgamma.to(pdtype), dbeta.to(pdtype), None, None, None, None, None, None


def bn_act(P, bn, rows, C, k=1, slope=0.0):
    """BatchNorm(batch statistics) + (Leaky)ReLU (+ max over k consecutive rows) with the module's affine parameters, zero-padded to the
    panel's channel count when the module has fewer channels."""
    c_real = bn.weight.shape[0]
    gamma, beta = bn.weight, bn.bias
    if c_real < C:
        z = torch.zeros(C - c_real, dtype=gamma.dtype, device=gamma.device)
        gamma, beta = torch.cat((gamma, z)), torch.cat((beta, z))
    return BNActMaxFn.apply(P, gamma, beta, rows, C, k, slope, bn, c_real)


class InterpFn(torch.autograd.Function):
    """3-NN inverse-distance interpolation of known features (pointnet2_paconv_modules.py:225-229): Fk panel [B*mk (padded), >= C],
    idx [rows, 3] global known rows, w [rows, 3] -> panel [rows (padded), round32(C)]."""

    @staticmethod
    def forward(ctx, Fk, idx, w, C, rows):
        L = engine.lib()
        out = torch.empty(T._round_up(rows, T.ROW_PAD), _r32(C), dtype=torch.float32, device=Fk.device)
        with T._OnDevice(Fk.device):
            engine._check(L.fc_train_interp_fwd_f32(engine._ptr(Fk), Fk.shape[1], C, engine._ptr(idx), engine._ptr(w), rows, out.shape[0], engine._ptr(out),
                                                    out.shape[1], engine._stream()))
        ctx.save_for_backward(idx, w)
        ctx.meta = (C, rows, Fk.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = engine.lib()
        idx, w = ctx.saved_tensors
        C, rows, kshape = ctx.meta
        dout = dout.contiguous()
        n_known = int(kshape[0])
        order, offsets = _sorted_edges(idx, n_known)
        dFk = torch.empty(kshape, dtype=torch.float32, device=dout.device)
        with T._OnDevice(dout.device):
            engine._check(L.fc_train_rows_gather_bwd_f32(engine._ptr(dout), dout.shape[1], 0, C, engine._ptr(order), engine._ptr(offsets), engine._ptr(w), 3,
                                                         n_known, n_known, engine._ptr(dFk), kshape[1], engine._stream()))
        return dFk, None, None, None, None


def _scorenet(sn, gd_panel, edges):
    """ScoreNet (paconv.py:31-54; hidden [16], last_bn False): conv 3 -> 16 (no bias), BatchNorm2d, ReLU, conv 16 -> 8 (+ bias), softmax."""
    w0 = sn.mlp_convs_hidden[0].weight
    w1, b1 = sn.mlp_convs_hidden[1].weight, sn.mlp_convs_hidden[1].bias
    h = T.linear_act([gd_panel], [3], w0.reshape(w0.shape[0], 3), None, edges)
    h = bn_act(h, sn.mlp_bns_hidden[0], edges, _r32(w0.shape[0]))
    s = T.linear_act([h], [w0.shape[0]], w1.reshape(w1.shape[0], w0.shape[0]), b1, edges)
    return SoftmaxFn.apply(s, w1.shape[0], edges)


def paconv_embed(emb, pts):
    """pts [B, M, 3 + c] (xyz first) -> [B, M, E] with autograd to every embedder parameter."""
    L = engine.lib()
    B, M, Cin = pts.shape
    c = Cin - 3
    if M < 256:
        raise RuntimeError("PAConv embedder needs at least 256 context points (four 4x farthest-point down-samplings)")
    dev = pts.device
    K = K_NEIGHBOURS
    pts = pts.to(torch.float32)
    with torch.no_grad():
        xyz0 = torch.zeros(B * M, 4, dtype=torch.float32, device=dev)
        xyz0[:, :3] = pts[..., :3].reshape(B * M, 3)
    xyz = [xyz0]
    feats = [(T.to_panel(pts[..., 3:].reshape(B * M, c)), c)]
    n_pts = [M]
    # ---- set abstraction
    for l, sa in enumerate(emb.SA_modules):
        n, m = n_pts[l], n_pts[l] // 4
        f_panel, C = feats[l]
        with torch.no_grad(), T._OnDevice(dev):
            fidx = engine.op_fps(xyz[l].view(B, n, 4)[..., :3].contiguous(), m)                     # [B, m] local indices
            gi = (fidx.long() + (torch.arange(B, device=dev) * n)[:, None]).reshape(-1)
            qxyz = xyz[l][gi].contiguous()                                                           # [B*m, 4] (row gather: data movement)
            nidx = torch.empty(B * m, K, dtype=torch.int32, device=dev)
            engine._check(L.fc_op_paconv_knn_f32(engine._ptr(xyz[l]), engine._ptr(qxyz), engine._ptr(nidx), B, n, m, K, engine._stream()))
        edges = B * m * K
        E, gdiff = GroupFn.apply(f_panel, xyz[l], qxyz, nidx, C, B, n, m)
        gd_panel = T.to_panel(gdiff[:edges, :3])
        layers = list(sa.mlps[0])
        width = C + 3
        for j, layer in enumerate(layers):
            Cout = layer.output_dim
            S = _scorenet(layer.scorenet, gd_panel, edges)
            G = T.linear_act([E], [2 * width], layer.weightbank.t(), None, edges)                    # [edges, m * Cout]: column = kernel * Cout + channel
            o = AssignFn.apply(G, S, M_KERNELS, Cout, edges)
            if j + 1 < len(layers):
                y = bn_act(o, layer.bn, edges, Cout)
                E = CenterDiffFn.apply(y, Cout, K, B * m)
            else:
                nxt = bn_act(o, layer.bn, B * m, Cout, k=K)
            width = Cout
        xyz.append(qxyz)
        feats.append((nxt, width))
        n_pts.append(m)
    # ---- feature propagation: level i <- level i + 1
    for i in range(3, -1, -1):
        nu, mk = n_pts[i], n_pts[i + 1]
        rows = B * nu
        fk, C2 = feats[i + 1]
        fu, C1 = feats[i]
        with torch.no_grad(), T._OnDevice(dev):
            idx3 = torch.empty(rows, 3, dtype=torch.int32, device=dev)
            w3 = torch.empty(rows, 3, dtype=torch.float32, device=dev)
            engine._check(L.fc_train_three_nn_f32(engine._ptr(xyz[i]), engine._ptr(xyz[i + 1]), B, nu, mk, engine._ptr(idx3), engine._ptr(w3), engine._stream()))
        x = InterpFn.apply(fk, idx3, w3, C2, rows)
        segs, widths = [x, fu], [C2, C1]
        for blk in emb.FP_modules[i].mlp:
            w = blk.conv.weight
            y = T.linear_act(segs, widths, w.reshape(w.shape[0], w.shape[1]), None, rows)
            y = bn_act(y, blk.bn.bn, rows, _r32(w.shape[0]))
            segs, widths = [y], [w.shape[0]]
        feats[i] = (segs[0], widths[0])
    f0, C0 = feats[0]
    y = T.mlp_panels(emb.out_mlp, [f0], [C0], B * M, "GELU")
    return T.from_panel(y, B * M, emb.out_mlp.out_layer.out_features).reshape(B, M, -1)
