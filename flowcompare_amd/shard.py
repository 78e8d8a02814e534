"""Scene-sharded data parallelism for the forward log-prob path.

Scenes are independent (eval-mode BatchNorm, initialised ActNorm): rank r of G takes a contiguous block of the batch,
runs the engine on its own GPU and no collective touches the data path.  The only exchange is the scalar reduction that
turns per-rank log-prob sums into the global loss / bpd of `inner_loop` (one all-reduce of 2 numbers), plus an optional
all-gather of the [B/G, N] log-probs when the caller wants the full tensor.  One process per GPU, torch.distributed
("nccl" = RCCL over xGMI on MI355X nodes, "gloo" in the CPU tests).
"""
import math

import torch
import torch.distributed as dist


def shard_bounds(batch_size, rank, world):
    """[lo, hi) scene range of `rank`; the first (batch_size % world) ranks take one extra scene."""
    base, rem = divmod(batch_size, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch, rank=None, world=None):
    """Slices (extract_0, extract_1, extra_context) — and anything else batch-major — to this rank's scenes."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(batch[0].shape[0], rank, world)
    return tuple(None if t is None else t[lo:hi] for t in batch)


def global_loss_bpd(log_prob_local, input_dim, group=None):
    """loss = -mean over ALL ranks' points, bpd = loss*log2(e)/input_dim (model_initialization.py:225-228)."""
    t = torch.stack((log_prob_local.double().sum(), torch.tensor(float(log_prob_local.numel()), dtype=torch.float64,
                                                                 device=log_prob_local.device)))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    loss = -(t[0] / t[1])
    return loss.to(log_prob_local.dtype), (loss * math.log2(math.e) / input_dim).to(log_prob_local.dtype)


def gather_log_prob(log_prob_local, batch_size, group=None):
    """All ranks' [b_r, N] log-probs concatenated in scene order -> [batch_size, N] on every rank."""
    world = dist.get_world_size(group)
    sizes = [shard_bounds(batch_size, r, world) for r in range(world)]
    n = log_prob_local.shape[1]
    pad = max(hi - lo for lo, hi in sizes)
    buf = log_prob_local.new_zeros(pad, n)
    buf[: log_prob_local.shape[0]] = log_prob_local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], 0)


def sharded_inner_loop(batch, models_dict, config, eps=None, group=None):
    """inner_loop over this rank's shard of a GLOBAL batch; returns (global loss, local log_prob, global bpd)."""
    from .model_initialization import inner_loop
    local = shard_batch(batch)
    lo, hi = shard_bounds(batch[0].shape[0], dist.get_rank(group), dist.get_world_size(group))
    eps_local = None if eps is None else [e[lo:hi] for e in eps]
    _, lp, _ = inner_loop(local, models_dict, config, eps=eps_local)
    loss, bpd = global_loss_bpd(lp, config["input_dim"], group)
    return loss, lp, bpd
