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


# ---------------------------------------------------------------- training: the one real exchange step
class GradientReducer:
    """Bucketed SUM all-reduce of parameter gradients, overlapped with backward (one process per GPU; "nccl" = RCCL over xGMI).

    Buckets are filled in REVERSE parameter order: the flow's layers finish their backward last-to-first, so a bucket's all-reduce
    starts as soon as its last gradient has been accumulated (post-accumulate-grad hooks) while earlier layers are still being
    differentiated.  xGMI is point-to-point (ring collectives are per-link bound), so buckets are large: 32 MB by default, i.e.
    about 45 collectives for the 369 M fp32 gradients of the spline flow.  The reduction is a SUM: each rank differentiates
    -sum(log_prob_local) / n_global_points (`local_loss`), so the summed gradients are those of the global mean loss exactly,
    also for uneven shards -- what nn.DataParallel (model_initialization.py:186-188) computes by gathering outputs on one device.
    """

    def __init__(self, params, bucket_bytes=32 << 20, group=None):
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self.hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.reset()

    def reset(self):
        self.pending = [len(b) for b in self.buckets]
        self.inflight = [None] * len(self.buckets)

    def _launch(self, i):
        b = self.buckets[i]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in b])
        work = dist.all_reduce(flat, group=self.group, async_op=True)
        self.inflight[i] = (flat, work)

    def _on_grad(self, p):
        i = self.bucket_of[id(p)]
        self.pending[i] -= 1
        if self.pending[i] == 0:
            self._launch(i)

    def finish(self):
        """Waits for every bucket (launching those whose parameters received no gradient this step, as zeros, so that all ranks
        issue the same collectives) and writes the summed gradients back."""
        for i, b in enumerate(self.buckets):
            if self.inflight[i] is None:
                self._launch(i)
            flat, work = self.inflight[i]
            work.wait()
            off = 0
            for p in b:
                n = p.numel()
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
        self.reset()

    def remove(self):
        for h in self.hooks:
            h.remove()


def local_loss(log_prob_local, n_global_points):
    """This rank's share of the global mean loss: summed over ranks it equals -mean over ALL points."""
    return -log_prob_local.sum() / n_global_points


def sharded_training_step(batch, models_dict, config, reducer, optimizer=None, eps=None, grad_clip=None, group=None):
    """train.py:108-120 over a GLOBAL batch sharded by scenes: every rank differentiates its scenes through the HIP training path,
    gradients are summed by `reducer` (bucketed, overlapped with backward), then clip_grad_norm_ / optimizer.step() run identically
    on every rank.  Returns (global loss, local log_prob, global bpd, grad_norm)."""
    from .model_initialization import inner_loop
    from . import train_ops as T
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(batch[0].shape[0], rank, world)
    local = shard_batch(batch, rank, world)
    eps_local = None if eps is None else [e[lo:hi] for e in eps]
    n_global = batch[1].shape[0] * batch[1].shape[1]
    params = reducer.params
    for fp16 in (True, False):
        for p in params:
            p.grad = None
        reducer.reset()
        with T.step_guard(fp16=fp16, device=local[1].device) as guard:
            _, lp, _ = inner_loop(local, models_dict, config, eps=eps_local)
            local_loss(lp, n_global).backward()
            over = torch.tensor([1.0 if guard.overflowed() else 0.0], device=lp.device)
        reducer.finish()
        dist.all_reduce(over, group=group)                   # every rank repeats the step if ANY rank left the fp16 range
        if over.item() == 0.0:
            break
    loss, bpd = global_loss_bpd(lp.detach(), config["input_dim"], group)
    clip = config.get("grad_clip_val") if grad_clip is None else grad_clip
    norm = torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], max_norm=clip if clip else float("inf"))
    if optimizer is not None:
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return loss, lp.detach(), bpd, norm
