"""Differentiable `Flow.log_prob` for training: the same transform stack as the inference engine
(models/transform.py:70-76), assembled from the HIP training primitives of train_ops.py so that `loss.backward()` (train.py:112)
produces gradients for every flow parameter and for the context embedding.

What runs where
  * every O(points x features) product or reduction -- the MLPs (nets.py:19-30), LayerNorm, q / k / v / output projections and the
    softmax attention core (perceiver.py:18-35, 89-115), the spline element (spline_coupling.py:24-169), ActNorm + LinearLU applied as
    one Linear (act_norm.py:37-43, permuters.py:164-169) -- is a HIP kernel, forward and backward;
  * parameter-space algebra (building L, U and W = L U diag(exp(-log_scale)) from the LinearLU / ActNorm parameters, 300 x 300) is
    ordinary differentiable torch on the parameters: torch.autograd carries dW back to lower_entries / upper_entries / log_scale / shift;
  * the per-element closures (affine scale-and-shift, the augmenter's reparameterised draw, the base density) are HIP kernels too;
    what torch does on activations is data movement only (pad, slice, cat of panels) and the running sum of the per-point log-dets.

Covered: AugmentAttentionPreconditioner or IdentityTransform; PreConditionApplier with the attention or the global-context
pre-conditioner; CIFblock (augment / affine_cif / ActNorm / Slice around the coupling); RationalQuadraticSplineCoupling,
AffineCoupling or ExponentialCoupling (latent_dim - latent_dim // 2 <= 16, as in the inference engine); ActNormBijectionCloud incl. its
first-batch init; LinearLU, random_permute, FullCombiner, ExponentialCombiner; extra context.
"""
import math
import os

import torch
import torch.utils.checkpoint

from . import modules as M
from . import train_ops as T

LOG_2PI = math.log(2.0 * math.pi)
ACTIVATION_BUDGET_BYTES = None      # None: ACTIVATION_BUDGET_FRACTION of the HBM available at call time; 0: checkpoint every layer (flow_log_prob)
# Round 4: 0.85 (was 0.70).  On the C2 training step (16 x 4096 points, 115 layers, MI355X 288 GB) 0.70 keeps 75 layers and recomputes 40 in backward
# (peak 208.6 GiB); 0.82 -> peak 240.8 GiB, -31 ms; 0.88 -> 255.7 GiB, -62 ms per step (profiles/r04w_train_*.json, same box).  0.85 leaves ~20 GiB of
# the 268 GiB for the allocator's slack, RCCL buffers and the transient gradient panels of the layer in flight.
ACTIVATION_BUDGET_FRACTION = float(os.environ.get("FC_TRAIN_BUDGET_FRACTION", "0.85"))


def _attention_block(pre, h_panel, h_width, ctx_k, ctx_v, rows, B, N, Mctx):
    """PreNorm(AttentionControlledOut) on the query side: LayerNorm -> q -> softmax(q k^T I^-1/2) v -> lin."""
    att = pre.fn.attention
    inner = att.inner_dim
    hn = T.layer_norm(h_panel, pre.norm.weight, pre.norm.bias, rows, pre.norm.eps)
    q = T.linear_act([hn], [h_width], att.to_q.weight, None, rows)
    a = T.attention(q, ctx_k, ctx_v, B, N, Mctx, inner ** -0.5)
    return T.linear_act([a], [inner], pre.fn.lin.weight, pre.fn.lin.bias, rows), pre.fn.lin.out_features


def _kv(pre, ctx_panel, E, ctx_rows):
    """K and V projections of the context as two zero-padded panels (to_kv.weight rows [0, I) and [I, 2I))."""
    att = pre.fn.attention
    inner = att.inner_dim
    w = att.to_kv.weight
    return (T.linear_act([ctx_panel], [E], w[:inner], None, ctx_rows), T.linear_act([ctx_panel], [E], w[inner:], None, ctx_rows))


_TRI_INDEX_CACHE = {}


def _tri_indices(D, dev):
    """Strict lower / upper triangle index pairs of a D x D matrix on `dev` (115 LinearLU layers ask for the same two tensors every step)."""
    key = (D, str(dev))
    hit = _TRI_INDEX_CACHE.get(key)
    if hit is None:
        hit = _TRI_INDEX_CACHE[key] = (torch.tril_indices(D, D, -1, device=dev), torch.triu_indices(D, D, 1, device=dev))
    return hit


def _perm_weight(perm, cfg):
    """The permuter between layers as a matrix W (z = W x) and its log|det|, built in parameter space (torch autograd carries dW back):
    LinearLU (models/permuters.py:148-169: W = L U, unit-lower L, diag(U) = softplus(u) + eps), Permuter / random_permute (:55-70: a
    fixed permutation), FullCombiner (:15-30: W = w, slogdet), ExponentialCombiner (:34-53: W = expm(rescale tanh(scale w + shift) +
    reshift + 1e-8), log|det| = trace; torch.matrix_exp stands in for the reference's truncated series, which is converged to eps_expm)."""
    dev = next(perm.parameters(), None)
    if isinstance(perm, M.LinearLU):
        D = perm.num_features
        dev = perm.lower_entries.device
        diag = torch.nn.functional.softplus(perm.unconstrained_upper_diag) + perm.eps
        il, iu = _tri_indices(D, dev)
        L = torch.eye(D, device=dev, dtype=diag.dtype).index_put((il[0], il[1]), perm.lower_entries)
        U = torch.diag(diag).index_put((iu[0], iu[1]), perm.upper_entries)
        # L U on the library's own training Linear (y = x W^T with x = L, W = U^T): no vendor BLAS on the training path
        LU = T.from_panel(T.linear_act([T.to_panel(L)], [D], U.t(), None, D), D, D)
        return LU, torch.log(diag).sum()
    if isinstance(perm, M.Permuter):
        D = perm.permutation.numel()
        return torch.eye(D, device=perm.permutation.device)[perm.permutation.long()], torch.zeros((), device=perm.permutation.device)
    if isinstance(perm, M.FullCombiner):
        return perm.w, torch.linalg.slogdet(perm.w)[1]
    if isinstance(perm, M.ExponentialCombiner):
        wm = perm.rescale * torch.tanh(perm.scale * perm.w + perm.shift) + perm.reshift + 1e-8
        return torch.matrix_exp(wm), wm.diagonal().sum()
    raise NotImplementedError(f"training path: permuter {type(perm).__name__}")


class actnorm_init_mode:
    """How the first-batch ActNorm initialisation (act_norm.py:27-39) writes its statistics while this context is active.

    in_place=False (default outside the context): like the reference, `shift` / `log_scale` are REPLACED by new Parameter objects.
    in_place=True: the existing Parameters are overwritten (`.copy_`), so optimisers, gradient reducers and `models_dict['parameters']`
    keep pointing at live tensors -- what the sharded training step and `conditioning.condition_flow` need.
    group: a torch.distributed process group (or True for the default group): column sums, sums of squares and row counts are
    all-reduced first, so every rank lands on the statistics of the GLOBAL batch -- bit-identical ActNorm weights on all ranks,
    equal to what one process computes on the whole batch."""
    current = (False, None)

    def __init__(self, in_place=True, group=None):
        self.mode = (bool(in_place), group)

    def __enter__(self):
        self.prev, actnorm_init_mode.current = actnorm_init_mode.current, self.mode
        return self

    def __exit__(self, *a):
        actnorm_init_mode.current = self.prev


def _actnorm_data_init(an, parts, rows):
    """act_norm.py:27-39, 72-88: on the first training batch shift = mean and log_scale = log(unbiased std + eps) of the layer's input
    over batch and points (column statistics by the HIP reduction kernel, fp64 accumulation).  By default, like the reference, it
    REPLACES the Parameter objects (an optimiser built before the first forward therefore keeps updating the old, orphaned tensors and
    these stay at their data-dependent values -- the reference's behaviour, kept on purpose); `actnorm_init_mode` switches to in-place
    writes and to statistics of the global batch of a process group."""
    in_place, group = actnorm_init_mode.current
    with torch.no_grad():
        stats = [T.column_stats(p.detach(), w, rows) for p, w in parts]              # HIP reduction; what follows is parameter-sized
        mean = torch.cat([m for m, _ in stats]).double()
        var = torch.cat([v for _, v in stats]).double()                              # biased
        n = float(rows)
        if group is not None:
            import torch.distributed as dist
            g = None if group is True else group
            acc = torch.cat((mean * n, (var + mean * mean) * n, torch.tensor([n], dtype=torch.float64, device=mean.device)))
            dist.all_reduce(acc, group=g)
            width = mean.numel()
            n = float(acc[-1].item())
            mean = acc[:width] / n
            var = (acc[width:2 * width] / n - mean * mean).clamp_min(0.0)
        var = var * (n / max(n - 1.0, 1.0))                                          # unbiased, as tensor.std() in act_norm.py:84-85
        shift = mean.to(torch.float32).reshape(1, -1)
        log_scale = torch.log(torch.sqrt(var).to(torch.float32) + an.eps).reshape(1, -1)
        if in_place:
            an.shift.copy_(shift)
            an.log_scale.copy_(log_scale)
        else:
            an.shift = torch.nn.Parameter(shift.clone())
            an.log_scale = torch.nn.Parameter(log_scale)
        an.initialized += 1.0


def flow_log_prob(flow, x, context, extra_context=None, eps=None, act=None, checkpoint=True, activation_budget_bytes=None):
    """log p(x | context) [B, N] with autograd through HIP kernels.  Arguments as Flow.log_prob (modules.py); `eps` pins the
    augmenter noise; `checkpoint` recomputes a layer's forward during backward for the layers whose saved activations do not fit
    `activation_budget_bytes` (default: ACTIVATION_BUDGET_FRACTION = 85 % of the HBM that is free at call time; 0 = checkpoint every layer).
    Call inside train_ops.step_guard() to run the split-fp16 loops with the range flag."""
    cfg = flow._config
    act = act or cfg["coupling_block_nonlinearity"]
    B, N, Din = x.shape
    D = cfg["latent_dim"]
    d1 = D // 2
    d2 = D - d1
    rows = B * N
    Mctx, E = context.shape[1], context.shape[2]
    ctx_rows = B * Mctx
    ctx_panel = T.to_panel(context.reshape(ctx_rows, E))
    X = 0 if extra_context is None else extra_context.shape[-1]
    extra_panel = None if X == 0 else T.to_panel(extra_context.reshape(rows, X).to(torch.float32))
    x_panel = T.to_panel(x.reshape(rows, Din))
    logp = torch.zeros(x_panel.shape[0], dtype=torch.float32, device=x.device)
    eps = list(eps) if eps is not None else None
    transforms = list(flow.transforms)

    # ---- transform 0: augmenter (models/augmenter.py:15-19, 49-63 + distributions.py:128-153) or identity
    t0 = transforms[0]
    if isinstance(t0, M.AugmentAttentionPreconditioner):
        nz = D - Din
        k, v = _kv(t0.attn, ctx_panel, E, ctx_rows)
        h = T.mlp_panels(t0.pre_attn_mlp, [x_panel], [Din], rows, act)
        a, a_w = _attention_block(t0.attn, h, t0.pre_attn_mlp.out_layer.out_features, k, v, rows, B, N, Mctx)
        segs, widths = [x_panel], [Din]
        if X:
            segs.append(extra_panel); widths.append(X)
        segs.append(a); widths.append(a_w)
        p = T.mlp_panels(t0.augment.noise_dist.net, segs, widths, rows, act)
        e = eps.pop(0) if eps else torch.randn(B, N, nz, device=x.device)
        z2, ldj = T.gauss_draw(p, e.reshape(rows, nz), rows, nz)                   # z2 = mean + eps std ; ldj = -log N(z2; mean, std)
        logp = logp + ldj
        latent = torch.cat((x_panel[:, :Din], z2[:, :nz]), -1)
    elif isinstance(t0, M.IdentityTransform):
        latent = x_panel[:, :Din]
    else:
        raise NotImplementedError(f"training path: augmenter {type(t0).__name__}")
    x1 = T.to_panel(latent[:, :d1])
    x2 = T.to_panel(latent[:, d1:D])

    # ---- layers: each one (pre-conditioner + coupling + ActNorm + permuter) is a function of (x1, x2, logp, context) so that it can be
    #      wrapped in torch.utils.checkpoint: only the layer inputs stay resident and the layer's forward is recomputed during
    #      backward, which is also what the reference does for the pre-conditioner (models/cif_block.py:17-19).  Without it the
    #      saved activations of C2 (115 layers x ~4 GB at 16 x 4096 points) would not fit even in 288 GB.
    def conditioned_coupling(pc, cp, x1, x2, logp, ctx_panel, extra_panel, mlp_act):
        """PreConditionApplier (models/transform.py:47-52): conditioning vector from x1 (attention over the context, or the
        per-point global embedding itself), then the coupling on x2."""
        if isinstance(pc, M.CouplingPreconditionerGlobal):
            if ctx_rows != rows:
                raise RuntimeError("global context must be per target point: context [B, N, E]")
            c, c_w = ctx_panel, E
        else:
            k, v = _kv(pc.attn, ctx_panel, E, ctx_rows)
            h = T.mlp_panels(pc.pre_attention_mlp, [x1], [d1], rows, mlp_act)
            c, c_w = _attention_block(pc.attn, h, pc.pre_attention_mlp.out_layer.out_features, k, v, rows, B, N, Mctx)
        segs, widths = [x1], [d1]
        if X:
            segs.append(extra_panel); widths.append(X)
        segs.append(c); widths.append(c_w)
        p = T.mlp_panels(cp.nn, segs, widths, rows, act)
        if isinstance(cp, M.RationalQuadraticSplineCoupling):
            x2, ldj = T.rq_spline(x2, p, rows, d2, cp.num_bins)
        elif isinstance(cp, M.ExponentialCoupling):
            x2, ldj = T.expm_coupling(x2, p, cp, rows, d2)
        else:
            x2, ldj = T.affine(x2, p, rows, d2, cp.scale_fn_type)
        return x2, logp + ldj

    def make_layer(blk, an, perm, init_an, init_cif):
        cif = isinstance(blk, M.CIFblock)

        def layer(x1, x2, logp, ctx_panel, extra_panel, e):
            if cif:
                # CIFblock.forward (models/cif_block.py:71-100): augment D -> Dc from x, Reverse, affine coupling of the x part
                # conditioned on the noise part, ActNorm(Dc), Reverse, Slice back to D (same ConditionalNormal), then the coupling
                Dc = cfg["cif_latent_dim"]
                nz = Dc - D
                net = blk.augmenter.noise_dist.net
                clamp = float(blk.augmenter.noise_dist.clamp or 0.0)
                p = T.mlp_panels(net, [x1, x2], [d1, d2], rows, "GELU")
                z2, ldj = T.gauss_draw(p, e, rows, nz, clamp)
                logp = logp + ldj
                a = T.to_panel(torch.flip(z2[:, :nz], [-1]))                                   # first nz dims of rev(cat(x, z2))
                b = T.to_panel(torch.flip(torch.cat((x1[:, :d1], x2[:, :d2]), -1), [-1]))      # last D dims: rev(x)
                st = T.mlp_panels(blk.affine_cif.nn, [a], [nz], rows, "GELU")
                b, ldj = T.affine(b, st, rows, D, "sigmoid")
                logp = logp + ldj
                # ActNorm(Dc) followed by Reverse, as one Linear on (a | b) whose weight is a flipped diagonal (parameter space)
                if init_cif:
                    if float(blk.act_norm.initialized.item()) == 0.0:
                        _actnorm_data_init(blk.act_norm, [(a, nz), (b, D)], rows)
                g = torch.exp(-blk.act_norm.log_scale.reshape(-1))
                W = torch.diag(g).flip(0)
                bias = (-blk.act_norm.shift.reshape(-1) * g).flip(0)
                logp = logp - blk.act_norm.log_scale.sum()
                x1 = T.linear_act([a, b], [nz, D], W[:d1], bias[:d1], rows)
                x2 = T.linear_act([a, b], [nz, D], W[d1:D], bias[d1:D], rows)
                xs = T.linear_act([a, b], [nz, D], W[D:], bias[D:], rows)
                p2 = T.mlp_panels(net, [x1, x2], [d1, d2], rows, "GELU")
                logp = logp + T.normal_log_prob(xs, p2, rows, nz, clamp)
                x2, logp = conditioned_coupling(blk.flow.pre_conditioner, blk.flow.transform, x1, x2, logp, ctx_panel, None, "GELU")
            else:
                x2, logp = conditioned_coupling(blk.pre_conditioner, blk.transform, x1, x2, logp, ctx_panel, extra_panel, act)
            # ActNorm and the permuter between layers, applied as ONE Linear on (x1 | x2)
            W, b = None, None
            if an is not None:
                if init_an and float(an.initialized.item()) == 0.0:          # (a checkpointed recompute finds it initialised)
                    _actnorm_data_init(an, [(x1, d1), (x2, d2)], rows)
                g = torch.exp(-an.log_scale.reshape(-1))
                W, b = torch.diag(g), -an.shift.reshape(-1) * g
                logp = logp - an.log_scale.sum()
            if perm is not None:
                Wlu, logdet = _perm_weight(perm, cfg)
                if W is None:
                    W, b = Wlu, None
                else:                                         # Wlu diag(g) and Wlu b: a column scaling and a row reduction (parameter space), no GEMM
                    W, b = Wlu * g[None, :], (Wlu * b[None, :]).sum(1)
                logp = logp + logdet
            if W is not None:
                z1 = T.linear_act([x1, x2], [d1, d2], W[:d1], None if b is None else b[:d1], rows)
                z2 = T.linear_act([x1, x2], [d1, d2], W[d1:], None if b is None else b[d1:], rows)
                x1, x2 = z1, z2
            return x1, x2, logp
        return layer

    # Activation budget: HBM is 288 GB and one layer's saved activations are ~2.6 GB at 16 x 4096 points, so a good part of the stack
    # can simply keep them; only the layers beyond the budget are checkpointed (recomputed in backward).  The first kept layer is
    # measured (allocator growth) and that figure plans the rest.
    budget, kept_bytes, layer_bytes, n_kept, n_ckpt = 0, 0, None, 0, 0
    if checkpoint and torch.is_grad_enabled() and x.is_cuda:
        if activation_budget_bytes is None:
            activation_budget_bytes = ACTIVATION_BUDGET_BYTES
        if activation_budget_bytes is None:
            free, _total = torch.cuda.mem_get_info(x.device)
            reusable = torch.cuda.memory_reserved(x.device) - torch.cuda.memory_allocated(x.device)      # cached by the allocator, free to us
            activation_budget_bytes = int(ACTIVATION_BUDGET_FRACTION * (free + reusable))
        budget = int(activation_budget_bytes)
    # which ActNorm layers still wait for their first-batch statistics: ONE device read for the whole stack (a per-layer .item() is a
    # stream synchronisation per layer, which keeps the host from running ahead of the GPU)
    actnorms = [m for m in flow.modules() if isinstance(m, M.ActNormBijectionCloud)]
    uninitialised = {}
    if actnorms:
        flags = torch.cat([m.initialized.reshape(1).to(torch.float32) for m in actnorms]).cpu()
        uninitialised = {id(m): bool(flags[j].item() == 0.0) for j, m in enumerate(actnorms)}
    i = 1
    while i < len(transforms):
        blk = transforms[i]
        if isinstance(blk, M.CIFblock):
            inner = blk.flow
            if X:
                raise Exception("Not implemented extra context with cif")
        elif isinstance(blk, M.PreConditionApplier):
            inner = blk
        else:
            raise NotImplementedError(f"training path: transform {type(blk).__name__}")
        if not isinstance(inner.transform, (M.RationalQuadraticSplineCoupling, M.AffineCoupling, M.ExponentialCoupling)):
            raise NotImplementedError(f"training path: coupling {type(inner.transform).__name__}")
        i += 1
        an = perm = None
        if i < len(transforms) and isinstance(transforms[i], M.ActNormBijectionCloud):
            an = transforms[i]
            i += 1
        if i < len(transforms) and isinstance(transforms[i], (M.LinearLU, M.Permuter, M.FullCombiner, M.ExponentialCombiner)):
            perm = transforms[i]
            i += 1
        elif i < len(transforms) and not isinstance(transforms[i], (M.PreConditionApplier, M.CIFblock)):
            raise NotImplementedError(f"training path: transform {type(transforms[i]).__name__} between layers")
        # un-initialised ActNorm layers take their statistics from this batch, in training mode only (act_norm.py:38-39)
        init_an = an is not None and uninitialised.get(id(an), False)
        init_cif = isinstance(blk, M.CIFblock) and uninitialised.get(id(blk.act_norm), False)
        if (init_an or init_cif) and not flow.training:
            init_an = init_cif = False
        e = None
        if isinstance(blk, M.CIFblock):
            nzc = cfg["cif_latent_dim"] - D
            e = eps.pop(0) if eps else torch.randn(B, N, nzc, device=x.device)
            e = e.reshape(rows, nzc)
        fn = make_layer(blk, an, perm, init_an, init_cif)
        keep = not (checkpoint and torch.is_grad_enabled())
        if not keep and budget > 0 and (layer_bytes is None or kept_bytes + layer_bytes <= budget):
            keep = True                                   # this layer's activations stay resident: no recompute in backward
        if keep:
            before = torch.cuda.memory_allocated(x.device) if budget > 0 else 0
            x1, x2, logp = fn(x1, x2, logp, ctx_panel, extra_panel, e)
            if budget > 0:
                grown = torch.cuda.memory_allocated(x.device) - before
                layer_bytes = grown if layer_bytes is None else max(layer_bytes, grown)
                kept_bytes += grown
                n_kept += 1
        else:
            n_ckpt += 1
            x1, x2, logp = torch.utils.checkpoint.checkpoint(fn, x1, x2, logp, ctx_panel, extra_panel, e, use_reentrant=False)

    if budget > 0 and os.environ.get("FC_TRAIN_DEBUG"):
        print(f"[train_flow] activation budget {budget / 2**30:.1f} GiB, kept {kept_bytes / 2**30:.1f} GiB, largest layer {(layer_bytes or 0) / 2**30:.2f} GiB, {n_kept} layers kept / {n_ckpt} recomputed in backward, "
              f"allocated {torch.cuda.memory_allocated(x.device) / 2**30:.1f} GiB", flush=True)
    # ---- base density (models/distributions.py:192-195)
    logp = logp + T.base_density(x1, rows, d1) + T.base_density(x2, rows, d2)
    return logp[:rows].reshape(B, N)


def _stateful_modules(models_dict):
    actnorms = [m for m in models_dict["flow"].modules() if isinstance(m, M.ActNormBijectionCloud)]
    bns = [m for m in models_dict["input_embedder"].modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    return actnorms, bns


def step_attempts(models_dict):
    """Which arithmetic a training step tries, in order.  Normally the split-fp16 loops first, then -- if any operand left the fp16
    range -- the fp32-input loops.  A step in which an ActNorm layer still takes its first-batch statistics runs on the fp32-input loops
    straight away: that step's side effect (the data-dependent initialisation, act_norm.py:27-39) must come from in-range arithmetic."""
    actnorms, _ = _stateful_modules(models_dict)
    if actnorms and models_dict["flow"].training:
        flags = torch.cat([m.initialized.reshape(1).to(torch.float32) for m in actnorms])
        if bool((flags == 0).any().item()):
            return (False,)
    return (True, False)


def snapshot_step_state(models_dict):
    """What a forward pass in train() mode changes besides gradients: ActNorm `initialized` flags (+ the statistics written on the first
    batch) and the BatchNorm running statistics of the embedder.  Parameter-sized."""
    actnorms, bns = _stateful_modules(models_dict)
    return ([(m, m.shift, m.log_scale, m.shift.detach().clone(), m.log_scale.detach().clone(), m.initialized.clone()) for m in actnorms],
            [(m, None if m.running_mean is None else m.running_mean.clone(), None if m.running_var is None else m.running_var.clone(),
              None if m.num_batches_tracked is None else m.num_batches_tracked.clone()) for m in bns])


def restore_step_state(models_dict, snap):
    """Undo the side effects of a rejected attempt (fcflow.h: results are invalid once the range flag is set) before the step is
    repeated: the BatchNorm running statistics would otherwise be updated twice -- or poisoned by an out-of-range activation -- and a
    first-batch ActNorm initialisation would keep statistics of invalid activations."""
    with torch.no_grad():
        for m, shift_p, ls_p, shift_v, ls_v, init in snap[0]:
            if m.shift is not shift_p:                 # replaced by the reference-style initialisation: put the old Parameters back
                m.shift, m.log_scale = shift_p, ls_p
            m.shift.copy_(shift_v)
            m.log_scale.copy_(ls_v)
            m.initialized.copy_(init)
        for m, rm, rv, nb in snap[1]:
            if rm is not None:
                m.running_mean.copy_(rm)
                m.running_var.copy_(rv)
            if nb is not None:
                m.num_batches_tracked.copy_(nb)


def training_step(batch, models_dict, config, optimizer=None, eps=None, grad_clip=None):
    """One optimisation step as train.py:108-120 runs it: inner_loop -> loss.backward() -> clip_grad_norm_ -> optimizer.step().
    The flow must be in train() mode (or the inputs require grad) so that Flow.log_prob takes the differentiable HIP path; a DGCNN
    context embedder in train() mode is differentiated too (train_embed.py, BatchNorm batch statistics); in eval() mode it runs its
    inference kernels and receives no gradient.
    Range guard: the step runs on the split-fp16 loops first and is repeated on the fp32-input loops if any operand left the fp16
    range; the rejected attempt's side effects (BatchNorm running statistics, ActNorm first-batch statistics) are rolled back first and
    every gradient -- also of Parameters an ActNorm initialisation created -- is dropped.  Returns (loss, log_prob, bpd, grad_norm)."""
    from .model_initialization import inner_loop
    params = [p for p in models_dict["parameters"] if p.requires_grad]
    device = batch[1].device
    snap = snapshot_step_state(models_dict)
    for k, fp16 in enumerate(step_attempts(models_dict)):
        if k:
            restore_step_state(models_dict, snap)
        for p in params:
            p.grad = None
        for mod in (models_dict["flow"], models_dict["input_embedder"]):
            for p in mod.parameters():
                p.grad = None
        with T.step_guard(fp16=fp16, device=device) as guard:
            loss, log_prob, bpd = inner_loop(batch, models_dict, config, eps=eps)
            loss.backward()
            if not guard.overflowed():
                break
    clip = config.get("grad_clip_val") if grad_clip is None else grad_clip
    with_grad = [p for p in params if p.grad is not None]
    norm = torch.nn.utils.clip_grad_norm_(with_grad, max_norm=clip if clip else float("inf"))
    if optimizer is not None:
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return loss.detach(), log_prob.detach(), bpd, norm
