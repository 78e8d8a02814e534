"""Conditioned synthetic weights for full-depth runs (bench.py, tests/test_gpu_fullsize.py, tests/test_gpu_configs.py).

There is no network, so benchmark and full-size parity runs use random-init weights of the reference's architecture.  The
constructor's init is a bad stand-in for a checkpoint at the reference's depth (115 layers): with `LinearLU` at identity, ActNorm
at identity and every spline layer a random monotone map far from the identity, the composition of 115 layers is close to singular --
log-probs of -1900 nats per point, and the reference's own fp32 arithmetic sits 26 nats per point away from its fp64 run (measured with
the oracle), so no implementation can be compared at 1e-4 on it.  `condition_flow` puts the stack into the state a
flow is in when training starts in earnest:

  1. each coupling net's output layer is scaled by `out_scale` and, for the spline coupling, the derivative logits get the bias
     log(exp(1 - 1e-3) - 1), so that an untrained layer is close to the identity map (the zero-initialised last layer of Glow /
     the identity-initialised derivatives of neural spline flows; models/spline_coupling.py:196-197 parameter layout);
  2. `LinearLU` gets small random off-diagonal entries (uniform +-lu_scale/sqrt(D): unit-determinant mixing of the two halves, so
     that errors of one layer reach the conditioning nets of all later ones);
  3. ActNorm takes its data-dependent statistics from the given batch, layer by layer -- the reference's own first-training-batch
     initialisation (models/act_norm.py:27-39, 72-88), computed by the HIP training kernels -- and is marked initialised.

The result is an ordinary state_dict (oracle and HIP engine both consume it): log-probs of a few tens of nats per point, and the
oracle's fp32-vs-fp64 gap on the logged scalar (bpd) drops to the reference's own noise floor (4.5e-5).  Kernel work is
unchanged: same shapes, same launches.  The latent entering the splines has a standard deviation of about 1.6, so about 6 % of the
spline inputs lie outside the +-3 domain (identity tails, log-det 0) and a quarter of the rows pass within 1e-4 of the boundary somewhere
in the stack -- the full-depth tests force the HIP run's own inside / outside decisions on the fp64 oracle (tests/fullsize_util.py).

What conditioning cannot remove (measured with the oracle, profiles/micro/depth_error_trace.py): a 115-layer stack of mixing layers
is a dynamical system with a positive Lyapunov exponent -- along single rows the fp32-vs-fp64 distance of the latent grows by ~1.04x
per layer (1e-6 after the first layer, 1e-3 around layer 70) before it decays again, for ANY fp32 arithmetic including the
reference's own, so the worst of 512 rows sits 5e-3 ... 2e-2 nats from fp64 in eager fp32 PyTorch while the mean over rows (the bpd
the reference logs) agrees to 5e-5.  The full-depth tests therefore gate the scalar absolutely (1e-4) and the per-row distance
against the reference arithmetic's own distance on the same rows.
"""
import math

import torch

from . import modules as M


def _couplings(flow):
    for t in flow.transforms:
        if isinstance(t, M.PreConditionApplier):
            yield t.transform
        elif isinstance(t, M.CIFblock):
            yield t.flow.transform


def condition_flow(models_dict, config, batch, eps=None, out_scale=0.1, lu_scale=0.03, seed=0):
    """In place on models_dict['flow'] (HIP device).  `batch` = (extract_0, extract_1, extra_context) as for inner_loop: the batch
    the ActNorm statistics are taken from (the embedder runs as it is, normally in eval mode).  Returns models_dict."""
    from . import train_flow
    from . import train_ops as T
    flow, emb = models_dict["flow"], models_dict["input_embedder"]
    dev = next(flow.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("condition_flow: the flow must be on a HIP device (the ActNorm statistics come from the HIP kernels)")
    if getattr(flow, "_fc_conditioned", False):
        raise RuntimeError("condition_flow: this flow is already conditioned (a second call would scale the coupling output layers again)")
    flow._fc_conditioned = True
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for cp in _couplings(flow):
            out = cp.nn.out_layer
            out.weight.mul_(out_scale)
            out.bias.mul_(out_scale)
            if isinstance(cp, M.RationalQuadraticSplineCoupling):
                K = cp.num_bins
                b = out.bias.view(-1, 3 * K + 1)                   # per transformed dim: [w x K | h x K | d x (K+1)]
                b[:, 2 * K:] += math.log(math.exp(1.0 - 1e-3) - 1.0)
        for t in flow.modules():
            if isinstance(t, M.LinearLU):
                a = lu_scale / math.sqrt(t.num_features)
                for p in (t.lower_entries, t.upper_entries):
                    p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * a).to(dev))
            if isinstance(t, M.ActNormBijectionCloud):
                t.initialized.zero_()
                t.shift.zero_()
                t.log_scale.zero_()
        # first-batch ActNorm initialisation through the HIP training forward (no autograd graph is kept under no_grad)
        e0, e1, extra = batch
        Din = config["input_dim"]
        e0, e1 = e0[:, :, :Din], e1[:, :, :Din]
        ctx = emb(e0)
        if config["global"]:
            ctx = ctx[:, None, :].expand(-1, e1.shape[1], -1).contiguous()
        if extra is not None:
            extra = extra[:, None, :].expand(-1, e1.shape[1], -1)
        was_training = flow.training
        flow.train()
        try:
            # split-fp16 kernels first (the step's range flag decides), fp32-input kernels if an activation left the fp16 range
            for fp16 in (True, False):
                for t in flow.modules():
                    if isinstance(t, M.ActNormBijectionCloud):
                        t.initialized.zero_()
                with T.step_guard(fp16=fp16, device=dev) as guard, train_flow.actnorm_init_mode(in_place=True):
                    train_flow.flow_log_prob(flow, e1, ctx, extra, eps, checkpoint=False)
                    if not guard.overflowed():
                        break
        finally:
            flow.train(was_training)
        for t in flow.modules():
            if isinstance(t, M.ActNormBijectionCloud) and float(t.initialized.item()) == 0.0:
                raise RuntimeError("condition_flow: an ActNorm layer was not reached by the initialisation forward")
    return models_dict
