"""torch.library registration of the inference entry points ("registered as torch.library custom ops so autograd / DDP see
ordinary tensors").  Two ops in the `flowcompare_amd` namespace, HIP devices only (there is no CPU kernel: the product path fails loudly
without the HIP extension):

    torch.ops.flowcompare_amd.flow_log_prob(x, context, extra, eps, handle)    -> log_prob [B, N]
    torch.ops.flowcompare_amd.context_embed(points, handle)                    -> embedding [B, M, E] / [B, E]

`handle` is the integer key of an engine handle (engine.FlowHandle / DgcnnHandle / PaconvHandle: the packed weights behind the C ABI) in the
registry below; the modules of modules.py register their handle and call the ops in eval mode.  Fake (meta) implementations give the output
shapes, so the ops trace under torch.compile / torch.export.  The training path has its own autograd nodes (train_ops.py); these two ops are
the gradient-free inference calls and carry no autograd formula: called on inputs that require grad they raise like any op without one.
"""
import weakref
from typing import List, Optional

import torch

_HANDLES = weakref.WeakValueDictionary()


def register(handle):
    """Integer key under which `handle` can be passed to the ops (weak: the key dies with the handle)."""
    key = id(handle)
    _HANDLES[key] = handle
    return key


def _get(key):
    h = _HANDLES.get(int(key))
    if h is None:
        raise RuntimeError(f"flowcompare_amd: no live engine handle under key {key} (the module that owned it was released or re-packed)")
    return h


@torch.library.custom_op("flowcompare_amd::flow_log_prob", mutates_args=(), device_types="cuda")
def flow_log_prob(x: torch.Tensor, context: Optional[torch.Tensor], extra: Optional[torch.Tensor], eps: List[torch.Tensor], handle: int) -> torch.Tensor:
    return _get(handle).log_prob(x, context, extra, list(eps))


@flow_log_prob.register_fake
def _(x, context, extra, eps, handle):
    return x.new_empty((x.shape[0], x.shape[1]), dtype=torch.float32)


@torch.library.custom_op("flowcompare_amd::context_embed", mutates_args=(), device_types="cuda")
def context_embed(points: torch.Tensor, handle: int) -> torch.Tensor:
    return _get(handle).embed(points)


@context_embed.register_fake
def _(points, handle):
    h = _get(handle)
    if getattr(h, "is_global", False):
        return points.new_empty((points.shape[0], h.out_dim), dtype=torch.float32)
    return points.new_empty((points.shape[0], points.shape[1], h.out_dim), dtype=torch.float32)
