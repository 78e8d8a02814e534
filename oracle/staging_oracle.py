"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy / torch, fp64 or fp32) of the steps either side of the hot path
(SURVEY.md §8f N3 / N4).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Pinning: `clamp_infs`, `log_prob_to_change` and `co_unit_sphere` are pinned by tests/golden/stage_*.npz, produced by running
the reference's own functions (tests/golden/gen_golden_staging.py).  `fps` is PARITY UNPINNED at reference level: the
algorithm lives in torch-cluster==1.5.9 (reference environment.yml:186), which is absent from /root/reference and from this
image; its published algorithm (csrc/cpu/fps_cpu.cpp) is restated here and anchored on the reference's call sites
(dataloaders/ams_voxel_loader.py:298-307: random_start=False, ratio = n_samples / n, result sliced to n_samples).
"""
import math

import numpy as np
import torch


def clamp_infs(t):
    """test_flow.py:241-247 (returns a new tensor; the reference clamps in place)."""
    t = t.clone()
    m = t.isinf()
    if m.any():
        t[m] = t[~m].min()
    return t


def log_prob_to_change(lp10, lp00, multiple, hard_cutoff=None):
    """test_flow.py:249-275.  lp10 [B,N], lp00 [B,N0] -> change map [B,N]; raises AssertionError like the reference when not finite."""
    lp10, lp00 = clamp_infs(lp10), clamp_infs(lp00)
    if hard_cutoff is None:
        base_mean = lp00.mean(dim=-1).unsqueeze(-1)
        base_std = lp00.std(dim=-1).unsqueeze(-1)                       # unbiased, torch default
        changed = lp10 < base_mean - multiple * base_std
    else:
        changed = lp10 < hard_cutoff
    mx = lp10.max(dim=-1)[0].unsqueeze(-1)
    mn = lp10.min(dim=-1)[0].unsqueeze(-1)
    out = 1 - (lp10 - mn) / (mx - mn)
    out[~changed] = 0.0
    assert not torch.logical_or(out.isnan(), out.isinf()).any()        # utils.py:416-420 is_valid
    return out


def unit_sphere(points):
    """utils.py:259-269 on a copy: returns (points, furthest_distance, mean)."""
    p = points.clone()
    mean = p[:, :3].mean(axis=0)
    p[:, :3] -= mean
    far = torch.max(torch.linalg.norm(p[:, :3], dim=-1))
    p[:, :3] = p[:, :3] / far
    return p, far, mean


def co_unit_sphere(p0, p1):
    """utils.py:271-280."""
    joint, far, mean = unit_sphere(torch.cat((p0, p1)))
    return joint[:p0.shape[0]], joint[p0.shape[0]:], far, mean


def fps(src, ratio):
    """torch_cluster 1.5.9 fps, one cloud, random_start=False: ceil(ratio * n) indices, first = 0, squared distance over all
    columns, argmax with the first maximum winning ties (numpy argmax)."""
    x = np.asarray(src)
    n = x.shape[0]
    m = int(math.ceil(ratio * n))
    out = np.zeros(m, dtype=np.int64)
    dist = np.full(n, np.inf, dtype=x.dtype)
    last = 0
    for j in range(1, m):
        d = np.zeros(n, dtype=x.dtype)
        for c in range(x.shape[1]):                                   # column order, like the kernel's fp32 accumulation
            t = x[:, c] - x[last, c]
            d = d + t * t
        dist = np.minimum(dist, d)
        last = int(np.argmax(dist))
        out[j] = last
    return out


def stage_pair(v0, v1, n_ctx, n_tgt):
    """ams_voxel_loader.py:298-307 + :357-358 for one pair."""
    s0 = v0[torch.from_numpy(fps(v0.numpy(), n_ctx / v0.shape[0]))][:n_ctx]
    s1 = v1[torch.from_numpy(fps(v1.numpy(), n_tgt / v1.shape[0]))][:n_tgt]
    return co_unit_sphere(s0, s1)
