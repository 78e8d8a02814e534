"""Input staging, the step right before the log-prob path: farthest point subsampling and the joint
unit-sphere normalisation of the voxel loader (dataloaders/ams_voxel_loader.py:298-307,357-358; utils.py:259-280).

`fps` has the call shape of torch_cluster.fps as the loader uses it (torch-cluster==1.5.9 in the reference's environment.yml;
the package is not in this image, so its published algorithm is restated: start at index 0, squared distances over ALL
columns, first maximum wins).  `unit_sphere` / `co_unit_sphere` mirror utils.py.  All arithmetic runs in the HIP library.
"""
import math

import torch

from . import engine


def fps(src, batch=None, ratio=0.5, random_start=True):
    """Indices of the farthest-point subsample of src [n, C] (or a batch of equal-sized clouds given by `batch`)."""
    if random_start:
        raise NotImplementedError("fps: only random_start=False (the loader's setting, ams_voxel_loader.py:298-307) is built")
    if src.dim() != 2:
        raise RuntimeError(f"fps: src must be [n, C], got {tuple(src.shape)}")
    n_total = src.shape[0]
    B = 1
    if batch is not None and batch.numel() > 0:
        B = int(batch.max().item()) + 1
        counts = torch.bincount(batch.to(torch.long), minlength=B)
        if not bool((counts == counts[0]).all()) or not bool((batch[1:] >= batch[:-1]).all()):
            raise NotImplementedError("fps: batches must be sorted and of equal size")
    n = n_total // B
    m = int(math.ceil(ratio * n))
    idx = engine.stage_fps(src.reshape(B, n, src.shape[1]), m)
    return (idx + torch.arange(B, device=idx.device)[:, None] * n).reshape(-1)


def unit_sphere(points, return_inverse=False):
    """utils.py:259-269 (zero mean, unit ball), in place like the reference."""
    empty = points[:0]
    out, _, inv = engine.stage_co_unit_sphere(points.unsqueeze(0), empty.unsqueeze(0))
    points.copy_(out[0])
    if return_inverse:
        return points, {'furthest_distance': inv[0, 0], 'mean': inv[0, 1:4]}
    return points


def co_unit_sphere(points_0, points_1, return_inverse=False):
    """utils.py:271-280: joint zero-mean unit-ball normalisation of a cloud pair."""
    o0, o1, inv = engine.stage_co_unit_sphere(points_0.unsqueeze(0), points_1.unsqueeze(0))
    if return_inverse:
        return o0[0], o1[0], {'furthest_distance': inv[0, 0], 'mean': inv[0, 1:4]}
    return o0[0], o1[0]


def stage_pair(voxel_0_large, voxel_1_small, n_samples_context, n_samples):
    """The loader's last steps for one (context, target) voxel pair: FPS both clouds to their sample counts
    (ams_voxel_loader.py:298-307), then `last_processing` = co_unit_sphere (:357-358)."""
    def sub(v, k):
        sel = fps(v, torch.zeros(v.shape[0], dtype=torch.long, device=v.device), ratio=k / v.shape[0], random_start=False)
        return v[sel][:k]
    return co_unit_sphere(sub(voxel_0_large, n_samples_context), sub(voxel_1_small, n_samples), return_inverse=True)
