"""Training path, first slice: the residual MLP of models/nets.py:19-30 as torch.autograd Functions whose
forward AND backward are HIP kernels behind the C ABI (include/fcflow.h, fc_train_*; kernels in csrc/train.hip + the GEMM of the
inference path).  torch is plumbing: it owns the tensors, the stream and the autograd graph between the primitives; there is no
PyTorch arithmetic on activations here and no CPU fallback.

Activations travel as PANELS: contiguous fp32 [rows_pad, width_pad], rows padded to 256 and widths to 32 with zero pad columns
(`to_panel` / `from_panel`).  A Linear may read up to three panels side by side, so cat(x1, context) is never materialised.

Range guard: `step_guard()` hands every primitive one device flag for the whole optimisation step (forward and backward run on
different host threads under autograd); `guard.overflowed()` after backward tells the caller to repeat the step with
`step_guard(fp16=False)` (fp32-input MFMA loop, any range).  Without a guard the fp32-input loop runs.
"""
import ctypes
import os
import threading

import torch

from . import engine

ROW_PAD = 256
FUSED_ACT = os.environ.get("FC_TRAIN_FUSED_ACT", "1") != "0"       # 0: activation as its own pass behind the Linear (A/B runs)
ACT_IDS = {None: 0, "none": 0, "GELU": 1, "RELU": 2, "ELU": 3}


def _round_up(x, m):
    return (x + m - 1) // m * m


def to_panel(x2d):
    """[rows, C] -> zero-padded panel [round_up(rows, 256), round_up(C, 32)] (differentiable: plain torch padding)."""
    rows, c = x2d.shape
    return torch.nn.functional.pad(x2d.to(torch.float32), (0, _round_up(c, 32) - c, 0, _round_up(rows, ROW_PAD) - rows)).contiguous()


def from_panel(p, rows, c):
    return p[:rows, :c]


# ---------------------------------------------------------------- per-step state shared by forward and backward threads
class _Step:
    flag = None          # int32[1] device tensor or None (fp32-input loop)
    ws = {}              # device -> uint8 scratch tensor for the weight-gradient partial tiles
    lock = threading.Lock()


class step_guard:
    """with step_guard() as g: loss = ...; loss.backward()  ;  g.overflowed() -> repeat with step_guard(fp16=False)."""

    def __init__(self, fp16=True, device=None):
        self.fp16 = fp16
        self.device = device

    def __enter__(self):
        self.prev = _Step.flag
        _Step.flag = torch.zeros(1, dtype=torch.int32, device=self.device or "cuda") if self.fp16 else None
        self.mine = _Step.flag
        return self

    def __exit__(self, *a):
        _Step.flag = self.prev

    def overflowed(self):
        return self.mine is not None and bool(self.mine.item())


def _flag_ptr():
    return engine._ptr(_Step.flag)


def _ws(nbytes, device):
    with _Step.lock:
        t = _Step.ws.get(device)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            _Step.ws[device] = t
        return t


class _OnDevice:
    """`with _OnDevice(dev)` costs ~10 us of host time per entry, and a training step enters it ~20 000 times: the step is
    partly host-bound, so the switch is skipped when `dev` is already the current device (the normal one-process-per-GPU case)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        idx = dev.index if isinstance(dev, torch.device) else torch.device(dev).index
        self.ctx = None if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _panel_out(rows_pad, cols, rows, device):
    """Uninitialised [rows_pad, cols] panel whose pad ROWS are zeroed (the row-wise kernels write `rows` rows, pad columns included);
    at the bench sizes rows == rows_pad and nothing is filled."""
    t = torch.empty(rows_pad, cols, dtype=torch.float32, device=device)
    if rows < rows_pad:
        t[rows:].zero_()
    return t


def _vec_out(rows_pad, rows, device):
    t = torch.empty(rows_pad, dtype=torch.float32, device=device)
    if rows < rows_pad:
        t[rows:].zero_()
    return t


def _segs(widths):
    return (ctypes.c_int32 * len(widths))(*widths)


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _check_panel(t, width):
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous() and t.shape[0] % ROW_PAD == 0
            and t.shape[1] >= _round_up(width, 32)):
        raise RuntimeError("flowcompare_amd.train_ops: expected a contiguous fp32 HIP panel [rows_pad % 256 == 0, width padded to 32]")


class LinearActFn(torch.autograd.Function):
    """y = act(cat(x...) W^T + b (+ residual)) on panels; forward and backward are HIP kernels (csrc/train.hip)."""

    @staticmethod
    def forward(ctx, weight, bias, residual, act, rows, widths, *xs):
        L = engine.lib()
        N, K = weight.shape
        if sum(widths) != K or len(xs) != len(widths):
            raise RuntimeError(f"LinearActFn: segment widths {widths} do not add up to in_features {K}")
        for x, w in zip(xs, widths):
            _check_panel(x, w)
        rows_pad = xs[0].shape[0]
        dev = weight.device
        w32 = weight.detach().to(torch.float32).contiguous()
        b32 = None if bias is None else bias.detach().to(torch.float32).contiguous()
        segs = _segs(widths)
        N_pad = _round_up(N, 32)
        with _OnDevice(dev):
            nb = L.fc_train_linear_pack_bytes(N, segs, len(widths))
            pack = torch.empty(nb, dtype=torch.uint8, device=dev)
            s = engine._stream()
            engine._check(L.fc_train_linear_pack_f32(engine._ptr(w32), engine._ptr(b32), N, segs, len(widths), engine._ptr(pack),
                                                     ctypes.c_size_t(nb), _flag_ptr(), s))
            u = torch.empty(rows_pad, N_pad, dtype=torch.float32, device=dev)
            ldx = _segs([x.shape[1] for x in xs])
            if residual is not None:
                _check_panel(residual, N)
            if FUSED_ACT and act in (1, 2, 3):                    # GELU / RELU / ELU: u and y = act(u) from the GEMM's epilogue, one launch
                y = torch.empty_like(u)
                engine._check(L.fc_train_linear_act_fwd_f32(engine._ptr(pack), N, segs, len(widths), _ptr_array(xs), ldx, rows_pad,
                                                            engine._ptr(residual), 0 if residual is None else residual.shape[1],
                                                            engine._ptr(u), engine._ptr(y), N_pad, act, _flag_ptr(), s))
            else:
                engine._check(L.fc_train_linear_fwd_f32(engine._ptr(pack), N, segs, len(widths), _ptr_array(xs), ldx, rows_pad,
                                                        engine._ptr(residual), 0 if residual is None else residual.shape[1],
                                                        engine._ptr(u), N_pad, _flag_ptr(), s))
                if act:
                    y = torch.empty_like(u)
                    engine._check(L.fc_train_act_fwd_f32(engine._ptr(u), engine._ptr(y), rows_pad, N_pad, act, s))
                else:
                    y = u
        ctx.save_for_backward(pack, u if act else None, *xs)
        ctx.meta = (N, K, tuple(widths), act, rows, bias is not None, residual is not None, weight.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = engine.lib()
        pack, u, *xs = ctx.saved_tensors
        N, K, widths, act, rows, has_bias, has_res, wdtype = ctx.meta
        dev = dy.device
        rows_pad, N_pad = xs[0].shape[0], _round_up(N, 32)
        dy = dy.contiguous()
        segs = _segs(widths)
        need = ctx.needs_input_grad
        with _OnDevice(dev):
            s = engine._stream()
            if act:
                du = torch.empty_like(dy)
                engine._check(L.fc_train_act_bwd_f32(engine._ptr(dy), engine._ptr(u), engine._ptr(du), rows_pad, rows, N_pad, act, s))
            else:
                du = dy
            dW = db = None
            if need[0] or (has_bias and need[1]):
                nb = L.fc_train_linear_wgrad_ws_bytes(N, segs, len(widths), rows)
                ws = _ws(nb, dev)
                dW = torch.empty(N, K, dtype=torch.float32, device=dev) if need[0] else None
                db = torch.empty(N, dtype=torch.float32, device=dev) if (has_bias and need[1]) else None
                ldx = _segs([x.shape[1] for x in xs])
                engine._check(L.fc_train_linear_wgrad_f32(N, segs, len(widths), engine._ptr(du), N_pad, _ptr_array(xs), ldx, rows,
                                                          engine._ptr(dW), engine._ptr(db), 0, engine._ptr(ws), ctypes.c_size_t(nb), _flag_ptr(), s))
            dxs = [None] * len(xs)
            if any(need[6:]):
                K_pad = sum(_round_up(w, 32) for w in widths)
                dx = torch.empty(rows_pad, K_pad, dtype=torch.float32, device=dev)
                engine._check(L.fc_train_linear_dgrad_f32(engine._ptr(pack), N, segs, len(widths), engine._ptr(du), N_pad, rows_pad,
                                                          engine._ptr(dx), K_pad, _flag_ptr(), s))
                off = 0
                for i, (x, w) in enumerate(zip(xs, widths)):
                    wp = _round_up(w, 32)
                    if need[6 + i]:
                        g = dx[:, off:off + wp]
                        if x.shape[1] != wp:                      # a wider panel than the segment reads: its other columns get no gradient
                            g = torch.nn.functional.pad(g, (0, x.shape[1] - wp))
                        dxs[i] = g
                    off += wp
        if dW is not None:
            dW = dW.to(wdtype)
        return (dW, db, du if (has_res and need[2]) else None, None, None, None, *dxs)


def linear_act(xs, widths, weight, bias, rows, act=None, residual=None):
    """act(cat(xs) W^T + b + residual) on panels; `widths` = true widths of the input panels, `rows` = valid rows."""
    return LinearActFn.apply(weight, bias, residual, ACT_IDS[act] if not isinstance(act, int) else act, rows, tuple(widths), *xs)


class MlpFn(torch.autograd.Function):
    """The whole residual MLP of models/nets.py:19-30 as ONE autograd node (same kernels as a chain of LinearActFn nodes in forward).  Its
    backward walks the layers itself, so the data gradient of layer l + 1 can leave the GEMM already multiplied by act'(u_l) and with the
    residual branch's gradient added (fc_train_linear_dgrad_act_f32): no activation-backward pass and no gradient adds between the layers.
    Arguments: rows, act id, widths of the input panels, their count, then the input panels, then (weight, bias) of in_layer, the hidden
    layers and out_layer."""

    @staticmethod
    def forward(ctx, rows, act, widths, nx, *t):
        L = engine.lib()
        xs, params = t[:nx], t[nx:]
        nl = len(params) // 2
        for x, w in zip(xs, widths):
            _check_panel(x, w)
        rows_pad, dev = xs[0].shape[0], xs[0].device
        packs, us, ys, metas = [], [], [], []
        cur, cur_w, keep = list(xs), list(widths), None
        with _OnDevice(dev):
            s = engine._stream()
            for l in range(nl):
                W, b = params[2 * l], params[2 * l + 1]
                N, K = W.shape
                if sum(cur_w) != K:
                    raise RuntimeError(f"MlpFn: layer {l} reads {K} features, its input panels hold {cur_w}")
                w32 = W.detach().to(torch.float32).contiguous()
                b32 = None if b is None else b.detach().to(torch.float32).contiguous()
                segs = _segs(cur_w)
                N_pad = _round_up(N, 32)
                nb = L.fc_train_linear_pack_bytes(N, segs, len(cur_w))
                pack = torch.empty(nb, dtype=torch.uint8, device=dev)
                engine._check(L.fc_train_linear_pack_f32(engine._ptr(w32), engine._ptr(b32), N, segs, len(cur_w), engine._ptr(pack),
                                                         ctypes.c_size_t(nb), _flag_ptr(), s))
                h = l - 1                                             # hidden index: even -> keep = input, odd -> residual = keep
                last = l == nl - 1
                residual = keep if (0 < l < nl - 1 and h % 2 == 1) else None
                if 0 < l < nl - 1 and h % 2 == 0:
                    keep = cur[0]
                u = torch.empty(rows_pad, N_pad, dtype=torch.float32, device=dev)
                ldx = _segs([x.shape[1] for x in cur])
                if last:
                    engine._check(L.fc_train_linear_fwd_f32(engine._ptr(pack), N, segs, len(cur_w), _ptr_array(cur), ldx, rows_pad,
                                                            None, 0, engine._ptr(u), N_pad, _flag_ptr(), s))
                    y = u
                else:
                    y = torch.empty_like(u)
                    engine._check(L.fc_train_linear_act_fwd_f32(engine._ptr(pack), N, segs, len(cur_w), _ptr_array(cur), ldx, rows_pad,
                                                                engine._ptr(residual), 0 if residual is None else residual.shape[1],
                                                                engine._ptr(u), engine._ptr(y), N_pad, act, _flag_ptr(), s))
                packs.append(pack); us.append(None if last else u); ys.append(y)
                metas.append((N, K, tuple(cur_w), b is not None, W.dtype))
                cur, cur_w = [y], [N]
        ctx.save_for_backward(*xs, *packs, *[u for u in us if u is not None], *ys[:-1])
        ctx.meta = (rows, act, tuple(widths), nx, nl, metas)
        return ys[-1]

    @staticmethod
    def backward(ctx, dy):
        L = engine.lib()
        rows, act, widths, nx, nl, metas = ctx.meta
        sv = ctx.saved_tensors
        xs, packs = sv[:nx], sv[nx:nx + nl]
        us, ys = sv[nx + nl:nx + nl + nl - 1], sv[nx + 2 * nl - 1:]
        dev = dy.device
        rows_pad = xs[0].shape[0]
        need = ctx.needs_input_grad
        grads = [None] * (4 + nx + 2 * nl)
        du = dy.contiguous()                                             # out_layer has no activation
        du_next = None                                                   # du of layer l + 1 (the residual branch's gradient when l is an even hidden layer)
        with _OnDevice(dev):
            s = engine._stream()
            for l in range(nl - 1, -1, -1):
                N, K, in_w, has_bias, wdtype = metas[l]
                N_pad = _round_up(N, 32)
                ins = list(xs) if l == 0 else [ys[l - 1]]
                segs = _segs(in_w)
                ldx = _segs([x.shape[1] for x in ins])
                wi = 4 + nx + 2 * l
                if need[wi] or (has_bias and need[wi + 1]):
                    nb = L.fc_train_linear_wgrad_ws_bytes(N, segs, len(in_w), rows)
                    ws = _ws(nb, dev)
                    dW = torch.empty(N, K, dtype=torch.float32, device=dev) if need[wi] else None
                    db = torch.empty(N, dtype=torch.float32, device=dev) if (has_bias and need[wi + 1]) else None
                    engine._check(L.fc_train_linear_wgrad_f32(N, segs, len(in_w), engine._ptr(du), N_pad, _ptr_array(ins), ldx, rows,
                                                              engine._ptr(dW), engine._ptr(db), 0, engine._ptr(ws), ctypes.c_size_t(nb), _flag_ptr(), s))
                    grads[wi] = None if dW is None else dW.to(wdtype)
                    grads[wi + 1] = db
                if l == 0:
                    if any(need[4:4 + nx]):
                        K_pad = sum(_round_up(w, 32) for w in in_w)
                        dx = torch.empty(rows_pad, K_pad, dtype=torch.float32, device=dev)
                        engine._check(L.fc_train_linear_dgrad_f32(engine._ptr(packs[0]), N, segs, len(in_w), engine._ptr(du), N_pad, rows_pad,
                                                                  engine._ptr(dx), K_pad, _flag_ptr(), s))
                        off = 0
                        for i, (x, w) in enumerate(zip(xs, in_w)):
                            wp = _round_up(w, 32)
                            if need[4 + i]:
                                g = dx[:, off:off + wp]
                                if x.shape[1] != wp:
                                    g = torch.nn.functional.pad(g, (0, x.shape[1] - wp))
                                grads[4 + i] = g
                            off += wp
                    break
                # gradient w.r.t. the previous layer's pre-activation: (du . W + [residual branch]) * act'(u_{l-1})
                h = l - 1
                addend = du_next if (l < nl - 1 and h % 2 == 0 and l + 1 < nl - 1) else None
                K_pad = _round_up(K, 32)
                du_prev = torch.empty(rows_pad, K_pad, dtype=torch.float32, device=dev)
                engine._check(L.fc_train_linear_dgrad_act_f32(engine._ptr(packs[l]), N, segs, 1, engine._ptr(du), N_pad, rows_pad,
                                                              engine._ptr(du_prev), K_pad, engine._ptr(addend), engine._ptr(us[l - 1]), act,
                                                              _flag_ptr(), s))
                du_next, du = du, du_prev
        return tuple(grads)


FUSED_MLP = os.environ.get("FC_TRAIN_FUSED_MLP", "1") != "0"       # 0: the MLP as a chain of LinearActFn nodes (A/B runs, and any activation MlpFn does not take)


def mlp_panels(mlp, xs, widths, rows, act):
    """models/nets.py:19-30 on panels: act(in) ; even hidden layer: r = x, x = act(W x) ; odd: x = act(r + W x) ; out (no activation)."""
    act_id = ACT_IDS[act] if not isinstance(act, int) else act
    if FUSED_MLP and FUSED_ACT and act_id in (1, 2, 3) and all(l.in_features % 32 == 0 for l in list(mlp.layers) + [mlp.out_layer]):
        params = [mlp.in_layer.weight, mlp.in_layer.bias]
        for layer in mlp.layers:
            params += [layer.weight, layer.bias]
        params += [mlp.out_layer.weight, mlp.out_layer.bias]
        return MlpFn.apply(rows, act_id, tuple(widths), len(xs), *xs, *params)
    x = linear_act(xs, widths, mlp.in_layer.weight, mlp.in_layer.bias, rows, act)
    keep = None
    for i, layer in enumerate(mlp.layers):
        w = layer.in_features
        if i % 2 == 0:
            keep = x
            x = linear_act([x], [w], layer.weight, layer.bias, rows, act)
        else:
            x = linear_act([x], [w], layer.weight, layer.bias, rows, act, residual=keep)
    return linear_act([x], [mlp.out_layer.in_features], mlp.out_layer.weight, mlp.out_layer.bias, rows, None)


def mlp_forward(mlp, x, act="GELU"):
    """MLP.forward for an ordinary [..., in_dim] HIP tensor, differentiable w.r.t. x and every parameter of `mlp`."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    rows = x2.shape[0]
    y = mlp_panels(mlp, [to_panel(x2)], [x2.shape[1]], rows, act)
    return from_panel(y, rows, mlp.out_layer.out_features).reshape(*lead, -1)


class AttentionFn(torch.autograd.Function):
    """out = softmax(q k^T scale) v per scene (models/perceiver.py:106-113) on panels q [B*N (padded), D], k / v [B*M (padded), D];
    D = head dim padded to 32 or 64 with zero columns.  Backward recomputes the scores tile by tile (csrc/train_attention.hip)."""

    @staticmethod
    def forward(ctx, q, k, v, B, N, M, scale):
        L = engine.lib()
        D = q.shape[1]
        for t, rows in ((q, B * N), (k, B * M), (v, B * M)):
            _check_panel(t, D)
            if t.shape[0] < rows or t.shape[1] != D:
                raise RuntimeError("AttentionFn: panel smaller than B * points, or head dims differ")
        dev = q.device
        out = _panel_out(q.shape[0], D, B * N, dev)
        with _OnDevice(dev):
            nb = L.fc_train_attention_ws_bytes(B, N, M, D)
            ws = _ws(nb, dev) if _Step.flag is not None else None
            stats = torch.empty(2 * B * N, dtype=torch.float32, device=dev)
            valid = ctypes.c_int32(0)
            engine._check(L.fc_train_attention_fwd_f32(engine._ptr(q), D, engine._ptr(k), D, engine._ptr(v), D, engine._ptr(out), D, B, N, M, D,
                                                       ctypes.c_float(scale), engine._ptr(ws), ctypes.c_size_t(nb), engine._ptr(stats),
                                                       ctypes.byref(valid), _flag_ptr(), engine._stream()))
        ctx.save_for_backward(q, k, v, out, stats)
        ctx.meta = (B, N, M, D, scale, int(valid.value))
        return out

    @staticmethod
    def backward(ctx, dout):
        L = engine.lib()
        q, k, v, out, stats = ctx.saved_tensors
        B, N, M, D, scale, stats_valid = ctx.meta
        dout = dout.contiguous()
        dq = _panel_out(q.shape[0], D, B * N, q.device)
        dk, dv = _panel_out(k.shape[0], D, B * M, q.device), _panel_out(k.shape[0], D, B * M, q.device)
        with _OnDevice(q.device):
            engine._check(L.fc_train_attention_bwd_f32(engine._ptr(q), D, engine._ptr(k), D, engine._ptr(v), D, engine._ptr(out), D,
                                                       engine._ptr(dout), D, engine._ptr(dq), D, engine._ptr(dk), D, engine._ptr(dv), D,
                                                       engine._ptr(stats), stats_valid if _Step.flag is not None else 0, B, N, M, D,
                                                       ctypes.c_float(scale), _flag_ptr(), engine._stream()))
        return dq, dk, dv, None, None, None, None


def attention(q, k, v, B, N, M, scale):
    return AttentionFn.apply(q, k, v, B, N, M, float(scale))


class SplineFn(torch.autograd.Function):
    """Rational-quadratic spline coupling element (models/spline_coupling.py:24-169), reference parameter layout.
    x2 panel [rows_pad, round_up(d2, 32)], params panel [rows_pad, round_up(d2 (3K+1), 32)] -> (y2 panel, ldj [rows_pad])."""

    @staticmethod
    def forward(ctx, x2, params, rows, d2, K):
        L = engine.lib()
        _check_panel(x2, d2)
        _check_panel(params, d2 * (3 * K + 1))
        y2 = _panel_out(x2.shape[0], _round_up(d2, 32), rows, x2.device)
        ldj = _vec_out(x2.shape[0], rows, x2.device)
        with _OnDevice(x2.device):
            engine._check(L.fc_train_rqspline_fwd_f32(engine._ptr(x2), x2.shape[1], engine._ptr(params), params.shape[1], engine._ptr(y2),
                                                      y2.shape[1], engine._ptr(ldj), rows, d2, K, engine._stream()))
        ctx.save_for_backward(x2, params)
        ctx.meta = (rows, d2, K)
        return y2, ldj

    @staticmethod
    def backward(ctx, dy2, dldj):
        L = engine.lib()
        x2, params = ctx.saved_tensors
        rows, d2, K = ctx.meta
        dy2, dldj = dy2.contiguous(), dldj.contiguous()
        dx2 = _panel_out(x2.shape[0], x2.shape[1], rows, x2.device) if x2.shape[1] == _round_up(d2, 32) else torch.zeros_like(x2)
        dparams = (_panel_out(params.shape[0], params.shape[1], rows, x2.device) if params.shape[1] == _round_up(d2 * (3 * K + 1), 32)
                   else torch.zeros_like(params))
        with _OnDevice(x2.device):
            engine._check(L.fc_train_rqspline_bwd_f32(engine._ptr(x2), x2.shape[1], engine._ptr(params), params.shape[1], engine._ptr(dy2),
                                                      dy2.shape[1], engine._ptr(dldj), engine._ptr(dx2), dx2.shape[1], engine._ptr(dparams),
                                                      dparams.shape[1], rows, d2, K, engine._stream()))
        return dx2, dparams, None, None, None


def rq_spline(x2, params, rows, d2, K):
    return SplineFn.apply(x2, params, rows, d2, K)


def _colsum(a, cols, rows):
    L = engine.lib()
    out = torch.empty(cols, dtype=torch.float32, device=a.device)
    nb = L.fc_train_colsum_ws_bytes(cols, rows)
    ws = _ws(nb, a.device)
    engine._check(L.fc_train_colsum_f32(engine._ptr(a), a.shape[1], cols, rows, engine._ptr(out), 0, engine._ptr(ws), ctypes.c_size_t(nb),
                                        engine._stream()))
    return out


class LayerNormFn(torch.autograd.Function):
    """torch.nn.LayerNorm(width) (PreNorm, models/perceiver.py:18-27) on a panel."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rows, eps):
        L = engine.lib()
        width = gamma.shape[0]
        _check_panel(x, width)
        y = _panel_out(x.shape[0], _round_up(width, 32), rows, x.device)
        stats = torch.empty(2 * rows, dtype=torch.float32, device=x.device)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        with _OnDevice(x.device):
            engine._check(L.fc_train_layernorm_fwd_f32(engine._ptr(x), x.shape[1], engine._ptr(g32), engine._ptr(b32), engine._ptr(y), y.shape[1],
                                                       engine._ptr(stats), rows, width, ctypes.c_float(eps), engine._stream()))
        ctx.save_for_backward(x, g32, stats)
        ctx.meta = (rows, width, gamma.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = engine.lib()
        x, g32, stats = ctx.saved_tensors
        rows, width, pdtype = ctx.meta
        dy = dy.contiguous()
        wp = _round_up(width, 32)
        dx = torch.empty(x.shape[0], x.shape[1], dtype=torch.float32, device=x.device)
        if x.shape[1] != wp:
            dx.zero_()
        t = torch.empty(x.shape[0], wp, dtype=torch.float32, device=x.device)
        with _OnDevice(x.device):
            engine._check(L.fc_train_layernorm_bwd_f32(engine._ptr(x), x.shape[1], engine._ptr(g32), engine._ptr(dy), dy.shape[1], engine._ptr(stats),
                                                       engine._ptr(dx), dx.shape[1], engine._ptr(t), wp, x.shape[0], rows, width, engine._stream()))
            dgamma = _colsum(t, width, rows).to(pdtype)
            dbeta = _colsum(dy, width, rows).to(pdtype)
        return dx, dgamma, dbeta, None, None


def layer_norm(x, gamma, beta, rows, eps=1e-5):
    return LayerNormFn.apply(x, gamma, beta, rows, eps)


SCALE_IDS = {"exp": 0, "sigmoid": 1}


class AffineFn(torch.autograd.Function):
    """Affine coupling element (models/affine_coupling.py:23-46): x2 panel, st panel [raw scale d2 | shift d2] -> (y2 panel, ldj)."""

    @staticmethod
    def forward(ctx, x2, st, rows, d2, scale_fn):
        L = engine.lib()
        _check_panel(x2, d2)
        _check_panel(st, 2 * d2)
        y2 = _panel_out(x2.shape[0], _round_up(d2, 32), rows, x2.device)
        ldj = _vec_out(x2.shape[0], rows, x2.device)
        with _OnDevice(x2.device):
            engine._check(L.fc_train_affine_fwd_f32(engine._ptr(x2), x2.shape[1], engine._ptr(st), st.shape[1], engine._ptr(y2), y2.shape[1],
                                                    engine._ptr(ldj), rows, d2, scale_fn, engine._stream()))
        ctx.save_for_backward(x2, st)
        ctx.meta = (rows, d2, scale_fn)
        return y2, ldj

    @staticmethod
    def backward(ctx, dy2, dldj):
        L = engine.lib()
        x2, st = ctx.saved_tensors
        rows, d2, scale_fn = ctx.meta
        dy2, dldj = dy2.contiguous(), dldj.contiguous()
        dx2 = _panel_out(x2.shape[0], x2.shape[1], rows, x2.device) if x2.shape[1] == _round_up(d2, 32) else torch.zeros_like(x2)
        dst = _panel_out(st.shape[0], st.shape[1], rows, x2.device) if st.shape[1] == _round_up(2 * d2, 32) else torch.zeros_like(st)
        with _OnDevice(x2.device):
            engine._check(L.fc_train_affine_bwd_f32(engine._ptr(x2), x2.shape[1], engine._ptr(st), st.shape[1], engine._ptr(dy2), dy2.shape[1],
                                                    engine._ptr(dldj), engine._ptr(dx2), dx2.shape[1], engine._ptr(dst), dst.shape[1], rows, d2,
                                                    scale_fn, engine._stream()))
        return dx2, dst, None, None, None


def affine(x2, st, rows, d2, scale_fn_type):
    return AffineFn.apply(x2, st, rows, d2, SCALE_IDS[scale_fn_type])


class GaussDrawFn(torch.autograd.Function):
    """Reparameterised draw of the augmenter (models/augmenter.py:49-63): p panel [mean nz | log std nz], eps [rows, nz] ->
    (z panel, ldj = -log N(z; mean, std) summed over the nz dims); std = min(exp(log std), clamp) when clamp > 0."""

    @staticmethod
    def forward(ctx, p, eps, rows, nz, clamp):
        L = engine.lib()
        _check_panel(p, 2 * nz)
        eps = eps.to(torch.float32).contiguous()
        z = _panel_out(p.shape[0], _round_up(nz, 32), rows, p.device)
        ldj = _vec_out(p.shape[0], rows, p.device)
        with _OnDevice(p.device):
            engine._check(L.fc_train_gauss_fwd_f32(engine._ptr(p), p.shape[1], engine._ptr(eps), engine._ptr(z), z.shape[1], engine._ptr(ldj), rows, nz,
                                                   ctypes.c_float(clamp), engine._stream()))
        ctx.save_for_backward(p, eps)
        ctx.meta = (rows, nz, clamp)
        return z, ldj

    @staticmethod
    def backward(ctx, dz, dldj):
        L = engine.lib()
        p, eps = ctx.saved_tensors
        rows, nz, clamp = ctx.meta
        dz, dldj = dz.contiguous(), dldj.contiguous()
        dp = _panel_out(p.shape[0], p.shape[1], rows, p.device) if p.shape[1] == _round_up(2 * nz, 32) else torch.zeros_like(p)
        with _OnDevice(p.device):
            engine._check(L.fc_train_gauss_bwd_f32(engine._ptr(p), p.shape[1], engine._ptr(eps), engine._ptr(dz), dz.shape[1], engine._ptr(dldj),
                                                   engine._ptr(dp), dp.shape[1], rows, nz, ctypes.c_float(clamp), engine._stream()))
        return dp, None, None, None, None


def gauss_draw(p, eps, rows, nz, clamp=0.0):
    return GaussDrawFn.apply(p, eps, rows, nz, float(clamp or 0.0))


class NormalLogProbFn(torch.autograd.Function):
    """sum_j log N(v_j; mean_j, std_j) per row (Slice.forward, models/slice.py:31-44); p panel [mean nz | log std nz]."""

    @staticmethod
    def forward(ctx, v, p, rows, nz, clamp):
        L = engine.lib()
        _check_panel(v, nz)
        _check_panel(p, 2 * nz)
        out = _vec_out(v.shape[0], rows, v.device)
        with _OnDevice(v.device):
            engine._check(L.fc_train_normlp_fwd_f32(engine._ptr(v), v.shape[1], engine._ptr(p), p.shape[1], engine._ptr(out), rows, nz,
                                                    ctypes.c_float(clamp), engine._stream()))
        ctx.save_for_backward(v, p)
        ctx.meta = (rows, nz, clamp)
        return out

    @staticmethod
    def backward(ctx, g):
        L = engine.lib()
        v, p = ctx.saved_tensors
        rows, nz, clamp = ctx.meta
        g = g.contiguous()
        dv = _panel_out(v.shape[0], v.shape[1], rows, v.device) if v.shape[1] == _round_up(nz, 32) else torch.zeros_like(v)
        dp = _panel_out(p.shape[0], p.shape[1], rows, v.device) if p.shape[1] == _round_up(2 * nz, 32) else torch.zeros_like(p)
        with _OnDevice(v.device):
            engine._check(L.fc_train_normlp_bwd_f32(engine._ptr(v), v.shape[1], engine._ptr(p), p.shape[1], engine._ptr(g), engine._ptr(dv), dv.shape[1],
                                                    engine._ptr(dp), dp.shape[1], rows, nz, ctypes.c_float(clamp), engine._stream()))
        return dv, dp, None, None, None


def normal_log_prob(v, p, rows, nz, clamp=0.0):
    return NormalLogProbFn.apply(v, p, rows, nz, float(clamp or 0.0))


class BaseDensityFn(torch.autograd.Function):
    """Standard-normal log-density of a panel's `width` true columns per row (models/distributions.py:192-195)."""

    @staticmethod
    def forward(ctx, x, rows, width):
        L = engine.lib()
        _check_panel(x, width)
        out = _vec_out(x.shape[0], rows, x.device)
        with _OnDevice(x.device):
            engine._check(L.fc_train_base_fwd_f32(engine._ptr(x), x.shape[1], engine._ptr(out), rows, width, engine._stream()))
        ctx.save_for_backward(x)
        ctx.meta = (rows, width)
        return out

    @staticmethod
    def backward(ctx, g):
        L = engine.lib()
        (x,) = ctx.saved_tensors
        rows, width = ctx.meta
        g = g.contiguous()
        dx = _panel_out(x.shape[0], x.shape[1], rows, x.device) if x.shape[1] == _round_up(width, 32) else torch.zeros_like(x)
        with _OnDevice(x.device):
            engine._check(L.fc_train_base_bwd_f32(engine._ptr(x), x.shape[1], engine._ptr(g), engine._ptr(dx), dx.shape[1], rows, width, engine._stream()))
        return dx, None, None


def base_density(x, rows, width):
    return BaseDensityFn.apply(x, rows, width)


class EdgeBNMaxFn(torch.autograd.Function):
    """One EdgeConv level of the DGCNN embedder in training mode, after the two per-point products (csrc/train_edge.hip):
    pq panel [rows_pad, 2C] = [P | Q] (or [rows_pad, C] = P alone when idx is None: BatchNorm1d + LeakyReLU of conv5), idx [rows, k]
    int32 global row indices -> max_j lrelu(BN_batch(P[idx_ij] + Q[i])) as a panel [rows_pad, C].  `bn` is the BatchNorm module:
    its running statistics are updated exactly as torch does in train mode (momentum, unbiased variance)."""

    @staticmethod
    def forward(ctx, pq, gamma, beta, idx, rows, C, k, bn):
        L = engine.lib()
        dev = pq.device
        has_q = idx is not None
        ld = pq.shape[1]
        if ld < (2 * C if has_q else C) or C % 32 != 0:
            raise RuntimeError("EdgeBNMaxFn: panel narrower than the channels, or channel count not a multiple of 32")
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        stats = torch.empty(3 * C, dtype=torch.float32, device=dev)
        out = _panel_out(pq.shape[0], C, rows, dev)
        arg = torch.empty(rows, C, dtype=torch.uint8, device=dev)
        q_ptr = ctypes.c_void_p(pq.data_ptr() + 4 * C) if has_q else ctypes.c_void_p(0)
        with _OnDevice(dev):
            s = engine._stream()
            nb = L.fc_train_edge_ws_bytes(rows, C)
            ws = _ws(nb, dev)
            engine._check(L.fc_train_edge_stats_f32(engine._ptr(pq), ld, q_ptr, ld, engine._ptr(idx), rows, k, C, ctypes.c_float(bn.eps),
                                                    engine._ptr(stats), engine._ptr(ws), ctypes.c_size_t(nb), s))
            engine._check(L.fc_train_edge_fwd_f32(engine._ptr(pq), ld, q_ptr, ld, engine._ptr(idx), rows, k, C, engine._ptr(stats), engine._ptr(g32),
                                                  engine._ptr(b32), ctypes.c_float(0.2), engine._ptr(out), C, engine._ptr(arg), s))
        if bn.track_running_stats and bn.running_mean is not None:
            # torch.nn.BatchNorm train-mode side effect (parameter-sized vectors): running <- (1 - m) running + m batch, unbiased variance
            with torch.no_grad():
                n = rows * k
                m = bn.momentum if bn.momentum is not None else 0.1
                bn.running_mean.mul_(1 - m).add_(stats[:C].to(bn.running_mean.dtype), alpha=m)
                bn.running_var.mul_(1 - m).add_(stats[2 * C:].to(bn.running_var.dtype) * (n / max(n - 1, 1)), alpha=m)
                bn.num_batches_tracked += 1
        order = offsets = None
        if has_q:
            # edges sorted by the row they point at (index plumbing): the backward then sums each row's incoming gradients in a fixed order
            flat = idx.reshape(-1).long()
            order = torch.argsort(flat, stable=True).to(torch.int32)
            offsets = torch.zeros(rows + 1, dtype=torch.int32, device=dev)
            offsets[1:] = torch.cumsum(torch.bincount(flat, minlength=rows), 0).to(torch.int32)
        ctx.save_for_backward(pq, g32, b32, stats, arg, idx, order, offsets)
        ctx.meta = (rows, C, k, has_q, gamma.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        L = engine.lib()
        pq, g32, b32, stats, arg, idx, order, offsets = ctx.saved_tensors
        rows, C, k, has_q, pdtype = ctx.meta
        dev = pq.device
        g = g.contiguous()
        ld = pq.shape[1]
        q_ptr = ctypes.c_void_p(pq.data_ptr() + 4 * C) if has_q else ctypes.c_void_p(0)
        rows_pad = pq.shape[0]
        t1 = torch.empty(rows_pad, C, dtype=torch.float32, device=dev)
        t2 = torch.empty(rows_pad, C, dtype=torch.float32, device=dev)
        dpq = torch.zeros_like(pq)
        with _OnDevice(dev):
            s = engine._stream()
            engine._check(L.fc_train_edge_bwd_prep_f32(engine._ptr(pq), ld, q_ptr, ld, engine._ptr(idx), rows, k, C, engine._ptr(stats), engine._ptr(g32),
                                                       engine._ptr(b32), ctypes.c_float(0.2), engine._ptr(arg), engine._ptr(g), g.shape[1], engine._ptr(t1),
                                                       engine._ptr(t2), C, rows_pad, s))
            dbeta, dgamma = _colsum(t1, C, rows), _colsum(t2, C, rows)
            dq_ptr = ctypes.c_void_p(dpq.data_ptr() + 4 * C) if has_q else ctypes.c_void_p(0)
            if has_q:
                # dQ by row sums, dP by an owner-computes gather over the sorted edges: no atomics, bit-reproducible
                engine._check(L.fc_train_edge_bwd_scatter_f32(engine._ptr(pq), ld, q_ptr, ld, engine._ptr(idx), rows, k, C, engine._ptr(stats),
                                                              engine._ptr(g32), engine._ptr(arg), engine._ptr(t1), C, engine._ptr(dbeta),
                                                              engine._ptr(dgamma), ctypes.c_void_p(0), ld, dq_ptr, ld, s))
                engine._check(L.fc_train_edge_bwd_gather_f32(engine._ptr(pq), ld, q_ptr, ld, engine._ptr(idx), rows, k, C, engine._ptr(stats),
                                                             engine._ptr(g32), engine._ptr(arg), engine._ptr(t1), C, engine._ptr(dbeta),
                                                             engine._ptr(dgamma), engine._ptr(order), engine._ptr(offsets), engine._ptr(dpq), ld, s))
            else:
                engine._check(L.fc_train_edge_bwd_scatter_f32(engine._ptr(pq), ld, q_ptr, ld, engine._ptr(idx), rows, k, C, engine._ptr(stats),
                                                              engine._ptr(g32), engine._ptr(arg), engine._ptr(t1), C, engine._ptr(dbeta),
                                                              engine._ptr(dgamma), engine._ptr(dpq), ld, dq_ptr, ld, s))
        return dpq, dgamma.to(pdtype), dbeta.to(pdtype), None, None, None, None, None


def edge_bn_max(pq, bn, idx, rows, C, k):
    return EdgeBNMaxFn.apply(pq, bn.weight, bn.bias, idx, rows, C, k, bn)


class ExpmCouplingFn(torch.autograd.Function):
    """ExponentialCoupling element (models/exponential_coupling.py:44-58): x2 panel, o panel [d2*d2 raw matrix | d2 shift], scal4 =
    cat(scale, shift, rescale, reshift) -> (y2 panel, ldj).  d2 <= 16 (the layer emits d2^2 numbers per point)."""

    @staticmethod
    def forward(ctx, x2, o, scal4, rows, d2):
        L = engine.lib()
        _check_panel(x2, d2)
        _check_panel(o, d2 * d2 + d2)
        s4 = scal4.detach().to(torch.float32).contiguous()
        y2 = _panel_out(x2.shape[0], _round_up(d2, 32), rows, x2.device)
        ldj = _vec_out(x2.shape[0], rows, x2.device)
        status = torch.zeros(1, dtype=torch.int32, device=x2.device)
        with _OnDevice(x2.device):
            engine._check(L.fc_train_expm_fwd_f32(engine._ptr(x2), x2.shape[1], engine._ptr(o), o.shape[1], engine._ptr(s4), engine._ptr(y2), y2.shape[1],
                                                  engine._ptr(ldj), rows, d2, engine._ptr(status), engine._stream()))
        if int(status.item()):
            raise RuntimeError("ExponentialCoupling (training): a matrix norm exceeds 2^5; the backward keeps at most 64 squaring states")
        ctx.save_for_backward(x2, o, s4)
        ctx.meta = (rows, d2, scal4.dtype)
        return y2, ldj

    @staticmethod
    def backward(ctx, dy2, dldj):
        L = engine.lib()
        x2, o, s4 = ctx.saved_tensors
        rows, d2, sdtype = ctx.meta
        dy2, dldj = dy2.contiguous(), dldj.contiguous()
        dx2 = _panel_out(x2.shape[0], x2.shape[1], rows, x2.device) if x2.shape[1] == _round_up(d2, 32) else torch.zeros_like(x2)
        do = _panel_out(o.shape[0], o.shape[1], rows, x2.device) if o.shape[1] == _round_up(d2 * d2 + d2, 32) else torch.zeros_like(o)
        dscal = torch.zeros(x2.shape[0], 4, dtype=torch.float32, device=x2.device)
        with _OnDevice(x2.device):
            engine._check(L.fc_train_expm_bwd_f32(engine._ptr(x2), x2.shape[1], engine._ptr(o), o.shape[1], engine._ptr(s4), engine._ptr(dy2), dy2.shape[1],
                                                  engine._ptr(dldj), engine._ptr(dx2), dx2.shape[1], engine._ptr(do), do.shape[1], engine._ptr(dscal), rows,
                                                  d2, engine._stream()))
            ds4 = _colsum(dscal, 4, rows).to(sdtype)
        return dx2, do, ds4, None, None


def expm_coupling(x2, o, cp, rows, d2):
    scal4 = torch.cat((cp.scale.reshape(1), cp.shift.reshape(1), cp.rescale.reshape(1), cp.reshift.reshape(1)))
    return ExpmCouplingFn.apply(x2, o, scal4, rows, d2)


class PoolMaxMeanFn(torch.autograd.Function):
    """[max over a scene's M points | mean] of a panel t [B*M (padded), width] -> [B, 2 width] (DGCNNembedderGlobal, pytorch_gcn.py:178-182)."""

    @staticmethod
    def forward(ctx, t, B, M, width):
        L = engine.lib()
        out = torch.empty(B, 2 * width, dtype=torch.float32, device=t.device)
        arg = torch.empty(B, width, dtype=torch.int32, device=t.device)
        with _OnDevice(t.device):
            engine._check(L.fc_train_pool_fwd_f32(engine._ptr(t), t.shape[1], width, B, M, engine._ptr(out), 2 * width, engine._ptr(arg), engine._stream()))
        ctx.save_for_backward(arg)
        ctx.meta = (B, M, width, t.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        L = engine.lib()
        (arg,) = ctx.saved_tensors
        B, M, width, shape = ctx.meta
        g = g.contiguous()
        dt = torch.zeros(shape, dtype=torch.float32, device=g.device)
        with _OnDevice(g.device):
            engine._check(L.fc_train_pool_bwd_f32(engine._ptr(g), g.shape[1], engine._ptr(arg), width, B, M, engine._ptr(dt), shape[1], engine._stream()))
        return dt, None, None, None


def pool_max_mean(t, B, M, width):
    return PoolMaxMeanFn.apply(t, B, M, width)


def column_stats(panel, width, rows, eps=0.0):
    """Per-column mean and BIASED variance of a panel's first `rows` rows and `width` columns (fp64 accumulation on the device, the
    statistics kernel of the EdgeConv BatchNorm with k = 1): returns (mean [width], var [width]) -- parameter-sized vectors."""
    L = engine.lib()
    _check_panel(panel, width)
    stats = torch.empty(3 * width, dtype=torch.float32, device=panel.device)
    with _OnDevice(panel.device):
        nb = L.fc_train_edge_ws_bytes(rows, width)
        ws = _ws(nb, panel.device)
        engine._check(L.fc_train_edge_stats_f32(engine._ptr(panel), panel.shape[1], ctypes.c_void_p(0), 0, ctypes.c_void_p(0), rows, 1, width,
                                                ctypes.c_float(eps), engine._ptr(stats), engine._ptr(ws), ctypes.c_size_t(nb), engine._stream()))
    return stats[:width], stats[2 * width:]
