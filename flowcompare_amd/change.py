"""Change-map post-processing, the step right after the log-prob path.

Drop-in for `clamp_infs` / `log_prob_to_change` of the reference's test_flow.py:241-275: same names, arguments, in-place
clamping of the inputs and the same AssertionError when the result is not finite.  The arithmetic runs in the HIP library
(`fc_change_map_f32`, csrc/staging.hip); there is no CPU fallback.
"""
import torch

from . import engine


def _as_rows(t):
    if t.dim() == 1:
        return t.unsqueeze(0), True
    if t.dim() != 2:
        raise RuntimeError(f"log-prob tensor must be [N] or [B, N], got {tuple(t.shape)}")
    return t, False


def clamp_infs(tensor):
    """test_flow.py:241-247: every +-inf becomes the smallest non-inf entry of the whole tensor, in place."""
    if tensor.isinf().any():
        work = tensor.contiguous()
        engine.clamp_infs(work)
        if work.data_ptr() != tensor.data_ptr():
            tensor.copy_(work)
        print('Clamping infs!')
    return tensor


def log_prob_to_change(log_prob_1_given_0, log_prob_0_given_0, multiple, hard_cutoff=None):
    """NLL to change scaled from 0 to 1 (test_flow.py:249-275)."""
    l10, squeeze = _as_rows(log_prob_1_given_0)
    l00, _ = _as_rows(log_prob_0_given_0)
    w10, w00 = l10.contiguous(), l00.contiguous()
    had_inf = bool(w10.isinf().any()) or bool(w00.isinf().any())
    out, invalid = engine.change_map(w10, w00, float(multiple), None if hard_cutoff is None else float(hard_cutoff))
    if had_inf:                       # the reference's clamp_infs mutates the caller's tensors
        if w10.data_ptr() != l10.data_ptr():
            l10.copy_(w10)
        if w00.data_ptr() != l00.data_ptr():
            l00.copy_(w00)
        print('Clamping infs!')
    assert not invalid
    return out[0] if squeeze else out
