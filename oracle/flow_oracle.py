"""CPU oracle for the FlowCompare forward log-prob hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (eager PyTorch on the host, fp32 or fp64) of
the reference algorithm for the path BASELINE.json names: context embedder ->
augmenter -> n x [pre-conditioner -> coupling -> ActNorm -> permuter] -> base
density.  It is the CHECKER for the HIP engine: only tests/, __graft_entry__
.smoke() and bench.py's cpu_baseline leg may import it; the product package
flowcompare_amd never does.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function
below against golden vectors produced by running the reference itself in the
build container (tests/golden/gen_golden.py; fixtures tests/golden/*.npz).
Being eager PyTorch, it is differentiable: tests/test_oracle_grads.py pins its
autograd gradients (the backward of the path) to gradients
of the reference's own loss.backward() (tests/golden/gen_golden_grads.py).

It is written functionally over a flat state_dict (name -> tensor) whose key
names are the reference's checkpoint names, so a reference
checkpoint, the HIP engine and this oracle all consume the same weights.
Every function cites the reference file:line it restates
(paths relative to the reference repository root).
"""
import contextlib
import math
import threading

import torch
import torch.nn.functional as F

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------- helpers
def _act(name):
    """coupling_block_nonlinearity dispatch, model_initialization.py:54-61."""
    if name == "GELU":
        return F.gelu                      # exact erf form (nn.GELU default)
    if name == "RELU":
        return F.relu
    if name == "ELU":
        return F.elu
    raise ValueError(f"Invalid coupling_block_nonlinearity {name!r}")


def _n_mid(sd, prefix):
    n = 0
    while f"{prefix}.layers.{n}.weight" in sd:
        n += 1
    return n


def mlp(sd, prefix, x, act):
    """models/nets.py:19-30.  act(in) ; even hidden: r=x, x=act(Wx) ; odd: x=act(r+Wx) ; out.
    (`residual` flag of the reference is ignored there as well.)"""
    x = act(F.linear(x, sd[f"{prefix}.in_layer.weight"], sd[f"{prefix}.in_layer.bias"]))
    keep = None
    for i in range(_n_mid(sd, prefix)):
        y = F.linear(x, sd[f"{prefix}.layers.{i}.weight"], sd[f"{prefix}.layers.{i}.bias"])
        if i % 2 == 0:
            keep = x
            x = act(y)
        else:
            x = act(keep + y)
    return F.linear(x, sd[f"{prefix}.out_layer.weight"], sd[f"{prefix}.out_layer.bias"])


def cross_attention(sd, prefix, h, ctx):
    """models/perceiver.py:18-35 (PreNorm, query side only), :89-115 (AttentionMine +
    AttentionControlledOut).  Single head; scale = inner_dim**-0.5; context not normed."""
    wq = sd[f"{prefix}.fn.attention.to_q.weight"]
    wkv = sd[f"{prefix}.fn.attention.to_kv.weight"]
    inner = wq.shape[0]
    hn = F.layer_norm(h, (h.shape[-1],), sd[f"{prefix}.norm.weight"], sd[f"{prefix}.norm.bias"], 1e-5)
    q = hn @ wq.t()
    kv = ctx @ wkv.t()
    k, v = kv[..., :inner], kv[..., inner:]
    w = torch.softmax((q @ k.transpose(1, 2)) * (inner ** -0.5), dim=-1)
    return F.linear(w @ v, sd[f"{prefix}.fn.lin.weight"], sd[f"{prefix}.fn.lin.bias"])


# --------------------------------------------------------------------------- couplings
def _affine_scale(raw, kind, eps=1e-8):
    """models/affine_coupling.py:23-28."""
    if kind == "exp":
        return torch.exp(raw)
    if kind == "sigmoid":
        return (2 * torch.sigmoid(raw) - 1) * (1 - eps) + 1
    raise ValueError("Invalid scale_fn_type")


def affine_coupling(sd, prefix, x, cond, act, scale_fn, split=None, inverse=False):
    """models/affine_coupling.py:30-62.  Returns (y, ldj) forward, x inverse."""
    D = x.shape[-1]
    d1 = D // 2 if split is None else split
    a, b = x[..., :d1], x[..., d1:]
    inp = a if cond is None else torch.cat((a, cond), -1)
    st = mlp(sd, f"{prefix}.nn", inp, act)
    s = _affine_scale(st[..., : D - d1], scale_fn)
    t = st[..., D - d1:]
    if inverse:
        return torch.cat((a, (b - t) / s), -1)
    return torch.cat((a, b * s + t), -1), torch.log(s).sum(-1)


_SPLINE_TLS = threading.local()      # test helper of the full-depth parity tests: spline_decisions() below (per thread: the tests run independent oracle passes side by side)


@contextlib.contextmanager
def spline_decisions(forced=None):
    """Test helper (tests/fullsize_util.py).  Records the inside / outside decision `-3 <= x2 <= 3` (models/spline_coupling.py:35-48) of every
    forward spline evaluation inside the block, in call order (one [..., d2] bool mask per coupling layer), into the yielded list; with
    `forced` (a list of such masks, e.g. the decisions a HIP run took on its own fp32 latent) the evaluation uses THOSE decisions instead
    of its own: a forced-inside input that sits a rounding error beyond the boundary is evaluated at the boundary knot, a forced-outside
    one passes through with log-det 0.  The spline has derivative 0.6936 at the boundary knots but the identity outside, so log p jumps by
    0.366 nats at |x2| = 3: an fp32 run may land on either side where its latent is within its rounding error of 3, and only the run's
    own decisions make a like-for-like fp64 reference for it."""
    rec, it = [], (iter(forced) if forced is not None else None)

    def hook(x, inside):
        if it is not None:
            inside = next(it).to(torch.bool).reshape(inside.shape)
        rec.append(inside)
        return inside
    prev = getattr(_SPLINE_TLS, "hook", None)
    _SPLINE_TLS.hook = hook
    try:
        yield rec
    finally:
        _SPLINE_TLS.hook = prev


def rq_spline(x, uw, uh, ud, inverse=False, bound=3.0, min_w=1e-3, min_h=1e-3, min_d=1e-3):
    """models/spline_coupling.py:24-66 + :69-169 + :17-19, elementwise over x[...]
    with uw,uh [...,K] and ud [...,K+1].

    Quirks kept on purpose: the MLP emits K+1 derivative logits which
    are padded on both sides (K+3 values); only the first K+1 padded values are ever
    gathered, so knot 0 uses the constant log(exp(1-min_d-1)) = -min_d, knots 1..K use
    ud[0..K-1], ud[K] is dead.  searchsorted adds 1e-6 to the last knot in place, so
    an input exactly equal to +bound lands in bin K-1.  Outside [-bound, bound]: identity, ldj 0.
    """
    K = uw.shape[-1]
    inside = (x >= -bound) & (x <= bound)
    hook = getattr(_SPLINE_TLS, "hook", None)
    if hook is not None and not inverse:
        inside = hook(x, inside)                      # (test helper: record / force the decisions; a no-op clamp otherwise)
    xc = torch.where(inside, x.clamp(-bound, bound), torch.zeros_like(x))          # any in-range stand-in for masked lanes
    const = float(torch.tensor(math.log(math.exp((1 - min_d) - 1))))      # the reference builds it as a float32 tensor (:43)
    ud_p = torch.cat((torch.full_like(ud[..., :1], const), ud), -1)      # knots 0..K+1 (last unused)
    d = min_d + F.softplus(ud_p)

    def knots(u, m):
        p = m + (1 - m * K) * torch.softmax(u, -1)
        c = torch.cumsum(p, -1)
        c = F.pad(c, (1, 0))
        c = 2 * bound * c - bound
        c = c.clone()
        c[..., 0] = -bound
        c[..., -1] = bound
        return c, c[..., 1:] - c[..., :-1]
    cw, w = knots(uw, min_w)
    ch, h = knots(uh, min_h)
    loc = (ch if inverse else cw).clone()
    loc[..., -1] += 1e-6
    b = ((xc[..., None] >= loc).sum(-1) - 1)[..., None]
    g = lambda t: t.gather(-1, b)[..., 0]
    in_cw, in_w, in_ch, in_h = g(cw), g(w), g(ch), g(h)
    delta = h / w
    in_delta, d0, d1 = g(delta), g(d), g(d[..., 1:])
    if inverse:
        dy = xc - in_ch
        t3 = d0 + d1 - 2 * in_delta
        qa = dy * t3 + in_h * (in_delta - d0)
        qb = in_h * d0 - dy * t3
        qc = -in_delta * dy
        disc = qb.pow(2) - 4 * qa * qc
        root = (2 * qc) / (-qb - torch.sqrt(disc))
        out = root * in_w + in_cw
        tt = root * (1 - root)
        den = in_delta + t3 * tt
        num = in_delta.pow(2) * (d1 * root.pow(2) + 2 * in_delta * tt + d0 * (1 - root).pow(2))
        lad = -(torch.log(num) - 2 * torch.log(den))
    else:
        th = (xc - in_cw) / in_w
        tt = th * (1 - th)
        num = in_h * (in_delta * th.pow(2) + d0 * tt)
        den = in_delta + (d0 + d1 - 2 * in_delta) * tt
        out = in_ch + num / den
        dnum = in_delta.pow(2) * (d1 * th.pow(2) + 2 * in_delta * tt + d0 * (1 - th).pow(2))
        lad = torch.log(dnum) - 2 * torch.log(den)
    return torch.where(inside, out, x), torch.where(inside, lad, torch.zeros_like(lad))


def spline_coupling(sd, prefix, x, cond, act, num_bins, inverse=False):
    """models/spline_coupling.py:187-227: params reshaped [..., d2, 3K+1] split [K, K, K+1]."""
    D = x.shape[-1]
    d1 = D // 2
    a, b = x[..., :d1], x[..., d1:]
    inp = a if cond is None else torch.cat((a, cond), -1)
    p = mlp(sd, f"{prefix}.nn", inp, act).reshape(*x.shape[:-1], -1, 3 * num_bins + 1)
    y2, lad = rq_spline(b, p[..., :num_bins], p[..., num_bins:2 * num_bins], p[..., 2 * num_bins:], inverse=inverse)
    if inverse:
        return torch.cat((a, y2), -1)
    return torch.cat((a, y2), -1), lad.sum(-1)


def expm(w, eps, algo):
    """utils.py:321-327 and :294-312 (series with scaling-and-squaring, data-dependent stop)."""
    if algo == "torch":
        return torch.matrix_exp(w)
    if algo != "original":
        raise ValueError("Invalid expm algo!")
    nrm = torch.norm(w, p=1, dim=-1).max().item()
    scale = int(math.ceil(math.log2(max(nrm, 0.5))) + 1)
    w = w / (2 ** scale)
    s = torch.eye(w.shape[-1], dtype=w.dtype)
    t = w
    k = 2
    while torch.norm(t, p=1, dim=-1).max().item() > eps:
        s = s + t
        t = (w @ t) / k
        k += 1
    for _ in range(scale):
        s = s @ s
    return s


def exponential_coupling(sd, prefix, x, cond, act, algo, eps_expm, inverse=False):
    """models/exponential_coupling.py:44-75."""
    D = x.shape[-1]
    d1 = D // 2
    d2 = D - d1
    a, b = x[..., :d1], x[..., d1:]
    inp = a if cond is None else torch.cat((a, cond), -1)
    o = mlp(sd, f"{prefix}.nn", inp, act)
    wm, bv = o[..., : d2 * d2], o[..., d2 * d2:]
    wm = sd[f"{prefix}.rescale"] * torch.tanh(sd[f"{prefix}.scale"] * wm + sd[f"{prefix}.shift"]) + sd[f"{prefix}.reshift"] + 1e-8
    wm = wm.reshape(*wm.shape[:-1], d2, d2)
    if inverse:
        e = expm(-wm, eps_expm, algo)
        return torch.cat((a, (e @ (b - bv)[..., None])[..., 0]), -1)
    e = expm(wm, eps_expm, algo)
    y2 = (e @ b[..., None])[..., 0] + bv
    return torch.cat((a, y2), -1), wm.diagonal(dim1=-2, dim2=-1).sum(-1)


# --------------------------------------------------------------------------- ActNorm / permuters
ACTNORM_DATA_INIT = False      # set by actnorm_data_init(): the first TRAINING forward of un-initialised ActNorm layers


class actnorm_data_init:
    """Context manager: every ActNorm forward first sets its statistics from its input (act_norm.py:27-39, 72-88: shift = mean over
    batch and points, log_scale = log(unbiased std + 1e-6)), writing them into the state_dict -- what the reference does on the
    first batch when `training and not initialized`."""
    def __enter__(self):
        global ACTNORM_DATA_INIT
        self.prev, ACTNORM_DATA_INIT = ACTNORM_DATA_INIT, True

    def __exit__(self, *a):
        global ACTNORM_DATA_INIT
        ACTNORM_DATA_INIT = self.prev


def actnorm(sd, prefix, x, inverse=False):
    """models/act_norm.py:37-46 (already initialised, or data-dependent init inside actnorm_data_init())."""
    if ACTNORM_DATA_INIT and not inverse:
        with torch.no_grad():
            flat = x.reshape(-1, x.shape[-1])
            sd[f"{prefix}.shift"] = flat.mean(0, keepdim=True)
            sd[f"{prefix}.log_scale"] = torch.log(flat.std(0, keepdim=True) + 1e-6)
    sh, ls = sd[f"{prefix}.shift"], sd[f"{prefix}.log_scale"]
    if inverse:
        return sh + x * torch.exp(ls)
    return (x - sh) * torch.exp(-ls), (-ls).sum().expand(x.shape[:-1])


def lu_matrices(sd, prefix, eps):
    """models/permuters.py:148-162: unit-lower L, upper U with diag softplus(u)+eps; row-major tril/triu order."""
    diag = F.softplus(sd[f"{prefix}.unconstrained_upper_diag"]) + eps
    D = diag.shape[0]
    L = torch.eye(D, dtype=diag.dtype)
    U = torch.diag(diag)
    il = torch.tril_indices(D, D, -1)
    iu = torch.triu_indices(D, D, 1)
    L[il[0], il[1]] = sd[f"{prefix}.lower_entries"]
    U[iu[0], iu[1]] = sd[f"{prefix}.upper_entries"]
    return L, U, diag


def linear_lu(sd, prefix, x, eps, inverse=False):
    """models/permuters.py:164-177: z = (x U^T) L^T ; ldj = sum log diag(U)."""
    L, U, diag = lu_matrices(sd, prefix, eps)
    if inverse:
        t = torch.linalg.solve_triangular(L, x.transpose(-1, -2), upper=False, unitriangular=True)
        t = torch.linalg.solve_triangular(U, t, upper=True)
        return t.transpose(-1, -2)
    return (x @ U.t()) @ L.t(), torch.log(diag).sum().expand(x.shape[:-1])


def permuter(cfg, sd, prefix, x, inverse=False):
    """permuter dispatch of model_initialization.py:116-131 over models/permuters.py."""
    kind = cfg["permuter_type"]
    if kind == "LinearLU":
        return linear_lu(sd, prefix, x, cfg["linear_lu_eps"], inverse)
    if kind == "random_permute":                            # permuters.py:55-70
        if inverse:
            return x.index_select(-1, sd[f"{prefix}.inv_permutation"].long())
        return x.index_select(-1, sd[f"{prefix}.permutation"].long()), torch.zeros(x.shape[:-1], dtype=x.dtype)
    if kind == "FullCombiner":                              # permuters.py:15-30
        w = sd[f"{prefix}.w"]
        if inverse:
            return x @ torch.linalg.inv(w).t()
        return x @ w.t(), torch.linalg.slogdet(w)[1].expand(x.shape[:-1])
    if kind == "ExponentialCombiner":                       # permuters.py:34-53 (algo default 'original', eps_expm from config)
        wm = sd[f"{prefix}.rescale"] * torch.tanh(sd[f"{prefix}.scale"] * sd[f"{prefix}.w"] + sd[f"{prefix}.shift"]) \
            + sd[f"{prefix}.reshift"] + 1e-8
        if inverse:
            return x @ expm(-wm, cfg["eps_expm"], "original").t()
        return x @ expm(wm, cfg["eps_expm"], "original").t(), wm.diagonal().sum().expand(x.shape[:-1])
    raise ValueError(f"Invalid permuter type: {kind}")


# --------------------------------------------------------------------------- distributions / augment / slice
def std_normal_log_prob(x):
    """models/distributions.py:192-195 with utils.py:384-393 (sum over the feature dim)."""
    return (-0.5 * LOG_2PI - 0.5 * x ** 2).sum(-1)


def cond_normal_params(sd, prefix, cond, act, clamp=False):
    """models/distributions.py:128-138."""
    p = mlp(sd, f"{prefix}.net", cond, act)
    half = p.shape[-1] // 2
    mean, scale = p[..., :half], p[..., half:].exp()
    if clamp:
        scale = scale.clamp_max(clamp)
    return mean, scale


def normal_log_prob(z, mean, scale):
    """torch.distributions.Normal.log_prob as called from distributions.py:140-153."""
    return -((z - mean) ** 2) / (2 * scale ** 2) - scale.log() - 0.5 * LOG_2PI


# --------------------------------------------------------------------------- transforms of the flow
def _layout(cfg):
    """Transform list built by model_initialization.py:136-152: [augmenter] + per layer
    [cif_block, ActNorm?, permuter] (no ActNorm/permuter after the last layer)."""
    items = [("augment", 0)]
    i = 1
    L = cfg["n_flow_layers"]
    for layer in range(L):
        items.append(("block", i)); i += 1
        if layer != L - 1:
            if cfg["act_norm"]:
                items.append(("actnorm", i)); i += 1
            items.append(("permuter", i)); i += 1
    return items


def _derived(cfg):
    """config keys initialize_flow derives, model_initialization.py:33-45."""
    X = 1 if cfg["extra_z_value_context"] else 0
    return X, cfg["input_embedder"] in ("DGCNNembedderGlobal",)


def _coupling(cfg, sd, prefix, x, cond, act, inverse):
    ft = cfg["flow_type"]
    if ft == "AffineCoupling":
        return affine_coupling(sd, prefix, x, cond, act, cfg["affine_scale_fn"], inverse=inverse)
    if ft == "RationalQuadraticSplineCoupling":
        return spline_coupling(sd, prefix, x, cond, act, cfg["num_bins_spline"], inverse=inverse)
    if ft == "ExponentialCoupling":
        return exponential_coupling(sd, prefix, x, cond, act, cfg["coupling_expm_algo"], cfg["eps_expm"], inverse=inverse)
    raise ValueError("Invalid flow type")


def _precondition(cfg, sd, prefix, x, ctx, extra, act, is_global, mlp_act=None):
    """models/transform.py:47-52 + models/cif_block.py:14-27: conditioning vector of a coupling."""
    if is_global:
        c = ctx
    else:
        d1 = cfg["latent_dim"] // 2
        h = mlp(sd, f"{prefix}.pre_conditioner.pre_attention_mlp", x[..., :d1], mlp_act or act)
        c = cross_attention(sd, f"{prefix}.pre_conditioner.attn", h, ctx)
    if extra is not None:
        c = torch.cat((extra, c), -1)
    return c


def _block(cfg, sd, idx, x, ctx, extra, eps_iter, inverse=False):
    """One cif_helper product, models/cif_block.py:30-46."""
    act = _act(cfg["coupling_block_nonlinearity"])
    X, is_global = _derived(cfg)
    p = f"transforms.{idx}"
    D, Dc = cfg["latent_dim"], cfg["cif_latent_dim"]
    if D == Dc:
        if inverse:
            c = _precondition(cfg, sd, p, x, ctx, extra, act, is_global)
            return _coupling(cfg, sd, f"{p}.transform", x, c, act, True)
        c = _precondition(cfg, sd, p, x, ctx, extra, act, is_global)
        return _coupling(cfg, sd, f"{p}.transform", x, c, act, False)
    if D > Dc:
        raise ValueError("Augment dim smaller than main latent!")
    if X:
        raise ValueError("Not implemented extra context with cif")     # cif_block.py:33-34
    if is_global:
        raise ValueError("CIF + global embedding not implemented")
    # ---- CIFblock, models/cif_block.py:49-112 (GELU everywhere inside; extra_context ignored)
    gelu = F.gelu
    dist = f"{p}.augmenter.noise_dist"            # same weights as {p}.slicer.noise_dist (shared object)
    clamp = cfg["clamp_dist"]
    rev = lambda t: t.flip(-1)                     # Reverse(dim=-1), permuters.py:77-85
    if not inverse:
        mean, scale = cond_normal_params(sd, dist, x, gelu, clamp)     # augmenter.py:49-63, context=None -> cond on x
        z2 = mean + next(eps_iter) * scale
        ldj = -normal_log_prob(z2, mean, scale).sum(-1)
        y = rev(torch.cat((x, z2), -1))
        y, l = affine_coupling(sd, f"{p}.affine_cif", y, None, gelu, "sigmoid", split=Dc - D)
        ldj = ldj + l
        y, l = actnorm(sd, f"{p}.act_norm", y)
        ldj = ldj + l
        y = rev(y)
        z, x2 = y[..., :D], y[..., D:]                                 # slice.py:31-44
        mean, scale = cond_normal_params(sd, dist, z, gelu, clamp)
        ldj = ldj + normal_log_prob(x2, mean, scale).sum(-1)
        c = _precondition(cfg, sd, f"{p}.flow", z, ctx, None, act, False, mlp_act=gelu)
        z, l = _coupling(cfg, sd, f"{p}.flow.transform", z, c, act, False)
        return z, ldj + l
    c = _precondition(cfg, sd, f"{p}.flow", x, ctx, None, act, False, mlp_act=gelu)
    z = _coupling(cfg, sd, f"{p}.flow.transform", x, c, act, True)
    mean, scale = cond_normal_params(sd, dist, z, gelu, clamp)          # slice.py:46-58
    y = torch.cat((z, mean + next(eps_iter) * scale), -1)
    y = rev(y)
    y = actnorm(sd, f"{p}.act_norm", y, inverse=True)
    y = affine_coupling(sd, f"{p}.affine_cif", y, None, gelu, "sigmoid", split=Dc - D, inverse=True)
    return rev(y)[..., :D]


def _augment(cfg, sd, x, ctx, extra, eps_iter):
    """Transform 0: models/augmenter.py:15-19 + :49-63, or IdentityTransform (transform.py:86-92)."""
    D, Din = cfg["latent_dim"], cfg["input_dim"]
    if D == Din:
        return x, torch.zeros(x.shape[:-1], dtype=x.dtype)
    if D < Din:
        raise ValueError("Latent dim < Input dim")
    if cfg["augmenter_dist"] != "ConditionalNormal" or not cfg["use_attn_augment"]:
        raise NotImplementedError("only the attention-conditioned ConditionalNormal augmenter works in the reference (SURVEY F10)")
    act = _act(cfg["coupling_block_nonlinearity"])
    h = mlp(sd, "transforms.0.pre_attn_mlp", x, act)
    a = cross_attention(sd, "transforms.0.attn", h, ctx)
    if extra is not None:
        a = torch.cat((extra, a), -1)
    mean, scale = cond_normal_params(sd, "transforms.0.augment.noise_dist", torch.cat((x, a), -1), act)
    z2 = mean + next(eps_iter) * scale
    return torch.cat((x, z2), -1), -normal_log_prob(z2, mean, scale).sum(-1)


def flow_log_prob(cfg, sd, x, ctx, extra=None, eps=(), record=None):
    """models/transform.py:70-76.  x [B,N,Din], ctx [B,M,E] (or [B,N,E] global), extra [B,N,X] | None,
    eps = noise tensors in draw order (augmenter first, then one per CIF block)."""
    it = iter(eps)
    lp = torch.zeros(x.shape[:-1], dtype=x.dtype)
    for kind, idx in _layout(cfg):
        if kind == "augment":
            x, l = _augment(cfg, sd, x, ctx, extra, it)
        elif kind == "block":
            x, l = _block(cfg, sd, idx, x, ctx, extra, it)
        elif kind == "actnorm":
            x, l = actnorm(sd, f"transforms.{idx}", x)
        else:
            x, l = permuter(cfg, sd, f"transforms.{idx}", x)
        if record is not None:
            record.append((x, l))
        lp = lp + l
    return lp + std_normal_log_prob(x)


def spline_domain_margin(cfg, record):
    """Test helper for full-depth parity: per point, the smallest | |x2| - 3 | any RationalQuadraticSplineCoupling input of a forward
    pass came to; `record` is the list flow_log_prob(..., record=[]) filled (latent and log-det after every transform).
    The reference's spline is the identity with log-det 0 outside [-3, 3] but has derivative softplus(-1e-3) + 1e-3 = 0.6936 at the
    boundary knots (models/spline_coupling.py:43-45, 35-48), so log p jumps by 0.366 nats when an input crosses +-3: a point whose
    trajectory passes within rounding distance of the boundary may legitimately land on either side in fp32 (the reference's own fp32
    run does), exactly like a k-NN near-tie.  Returns [B, N] (inf when the flow has no spline coupling)."""
    margin = torch.full(record[0][0].shape[:-1], float("inf"), dtype=record[0][0].dtype)
    if cfg["flow_type"] != "RationalQuadraticSplineCoupling":
        return margin
    if cfg["latent_dim"] != cfg["cif_latent_dim"]:
        raise NotImplementedError("spline_domain_margin: plain PreConditionApplier stacks only")
    d1 = cfg["latent_dim"] // 2
    prev = None
    for (kind, _idx), (x, _l) in zip(_layout(cfg), record):
        if kind == "block" and prev is not None:
            margin = torch.minimum(margin, (prev[..., d1:].abs() - 3.0).abs().min(-1)[0])
        prev = x
    return margin


def flow_inverse(cfg, sd, z, ctx, extra=None, eps=()):
    """models/transform.py:79-84 from a given latent z (the draw from sample_dist is the caller's)."""
    it = iter(eps)
    for kind, idx in reversed(_layout(cfg)):
        if kind == "augment":
            z = z[..., : cfg["input_dim"]]                           # augmenter.py:65-67
        elif kind == "block":
            z = _block(cfg, sd, idx, z, ctx, extra, it, inverse=True)
        elif kind == "actnorm":
            z = actnorm(sd, f"transforms.{idx}", z, inverse=True)
        else:
            z = permuter(cfg, sd, f"transforms.{idx}", z, inverse=True)
    return z


# --------------------------------------------------------------------------- DGCNN embedders
def knn_indices(f, k):
    """models/pytorch_gcn.py:13-20 on channels-last f [B,M,C]: top-k of -|xi|^2 + 2 xi.xj - |xj|^2 (self included)."""
    inner = -2 * (f @ f.transpose(1, 2))
    sq = (f ** 2).sum(-1)
    pd = -sq[:, None, :] - inner - sq[:, :, None]
    return pd.topk(k, dim=-1)[1]


BN_BATCH_STATS = False      # set by train_mode(): BatchNorm normalises with the batch's own biased statistics (module.train())


class train_mode:
    """Context manager: the embedder's BatchNorms use batch statistics, as after `.train()` (train.py:46-47 builds the models in
    'train' mode).  The flow itself computes the same function in both modes once ActNorm is initialised (act_norm.py:38-39);
    torch.utils.checkpoint (cif_block.py:17-19) only changes what is stored.  Running statistics are not updated here."""
    def __enter__(self):
        global BN_BATCH_STATS
        self.prev, BN_BATCH_STATS = BN_BATCH_STATS, True

    def __exit__(self, *a):
        global BN_BATCH_STATS
        BN_BATCH_STATS = self.prev


def _bn(sd, prefix, y):
    """BatchNorm (eps 1e-5) on channels-last y: running statistics in eval mode, the batch's biased statistics over every leading
    axis in train mode (torch.nn.BatchNorm1d/2d semantics)."""
    if BN_BATCH_STATS:
        dims = tuple(range(y.dim() - 1))
        mean, var = y.mean(dims), y.var(dims, unbiased=False)
    else:
        mean, var = sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"]
    return (y - mean) * torch.rsqrt(var + 1e-5) * sd[f"{prefix}.weight"] + sd[f"{prefix}.bias"]


def edge_conv(sd, level, f, k):
    """models/pytorch_gcn.py:23-47 + conv{level} (1x1 conv, BN2d, LeakyReLU .2) + max over k (:85-99)."""
    B, M, C = f.shape
    idx = knn_indices(f, k)
    nb = torch.gather(f[:, None].expand(B, M, M, C), 2, idx[..., None].expand(B, M, k, C))
    e = torch.cat((nb - f[:, :, None], f[:, :, None].expand(B, M, k, C)), -1)
    w = sd[f"conv{level}.0.weight"].reshape(-1, 2 * C)
    y = F.leaky_relu(_bn(sd, f"conv{level}.1", e @ w.t()), 0.2)
    return y.max(dim=2)[0]


def dgcnn_trunk(sd, pts, k):
    """Shared trunk of models/pytorch_gcn.py:81-103 / :143-176 -> [B,M,512] after conv5."""
    f, outs = pts, []
    for level in (1, 2, 3, 4):
        f = edge_conv(sd, level, f, k)
        outs.append(f)
    cat = torch.cat(outs, -1)
    w5 = sd["conv5.0.weight"].reshape(512, 512)
    return F.leaky_relu(_bn(sd, "conv5.1", cat @ w5.t()), 0.2)


def dgcnn_embed(cfg, sd, pts):
    """DGCNNembedder.forward (pytorch_gcn.py:81-107) -> [B,M,E]; DGCNNembedderGlobal.forward (:143-188) -> [B,E]."""
    t = dgcnn_trunk(sd, pts, cfg["n_neighbors"])
    if cfg["input_embedder"] == "DGCNNembedderGlobal":
        t = torch.cat((t.max(dim=1)[0], t.mean(dim=1)), -1)
    return mlp(sd, "out_mlp", t, F.gelu)


def context_embed(cfg, sd, pts):
    """input_embedder dispatch of model_initialization.py:162-176."""
    if cfg["input_embedder"] == "PAConv":
        from . import paconv_oracle
        return paconv_oracle.paconv_embed(sd, pts)
    return dgcnn_embed(cfg, sd, pts)


# --------------------------------------------------------------------------- the boundary function
def inner_loop(cfg, sd_flow, sd_emb, batch, eps=()):
    """model_initialization.py:206-228 -> (loss, log_prob[B,N], bpd)."""
    e0, e1, extra = batch
    Din = cfg["input_dim"]
    e0, e1 = e0[..., :Din], e1[..., :Din]
    if extra is not None:
        if cfg["sample_size"] != e1.shape[1]:
            raise RuntimeError("extra_context is repeated to config['sample_size'], which must equal N")
        extra = extra[:, None, :].expand(-1, cfg["sample_size"], -1)
    emb = context_embed(cfg, sd_emb, e0)
    if emb.dim() == 2:
        emb = emb[:, None, :].expand(-1, e1.shape[1], -1)
    lp = flow_log_prob(cfg, sd_flow, e1, emb, extra, eps)
    loss = -lp.mean()
    return loss, lp, loss * math.log2(math.e) / Din


def make_sample(cfg, sd_flow, sd_emb, z, extract_0, extra=None, eps=()):
    """model_initialization.py:231-245 with the latent z [1,n,D] supplied by the caller."""
    n = z.shape[1]
    emb = context_embed(cfg, sd_emb, extract_0[..., : cfg["input_dim"]])
    if emb.dim() == 2:
        emb = emb[:, None, :].expand(-1, n, -1)
    if extra is not None:
        extra = extra[:, None, :].expand(-1, n, -1)
    return flow_inverse(cfg, sd_flow, z, emb, extra, eps).squeeze()
