"""The CPU restatements of the steps either side of the path (oracle/staging_oracle.py) against golden vectors produced by
the reference's own functions (tests/golden/gen_golden_staging.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import staging_oracle as S


def _cases():
    z = np.load(os.path.join(GOLDEN, "stage_change.npz"))
    return z, [tuple(r) for r in z["cases"]]


@pytest.mark.parametrize("dt,tag,tol", [(torch.float64, "f64", 1e-12), (torch.float32, "f32", 2e-6)])
def test_change_map_oracle_matches_reference(dt, tag, tol):
    z, cases = _cases()
    for ci, (B, N, N0, multiple, cutoff, use_cutoff) in enumerate(cases):
        a, b = torch.from_numpy(z[f"c{ci}_lp10"]).to(dt), torch.from_numpy(z[f"c{ci}_lp00"]).to(dt)
        out = S.log_prob_to_change(a, b, multiple, hard_cutoff=cutoff if use_cutoff else None)
        ref = torch.from_numpy(z[f"c{ci}_out_{tag}"])
        assert out.shape == ref.shape and (out - ref).abs().max().item() <= tol
        assert torch.equal(S.clamp_infs(a), torch.from_numpy(z[f"c{ci}_lp10_after_{tag}"]))
        assert torch.equal(S.clamp_infs(b), torch.from_numpy(z[f"c{ci}_lp00_after_{tag}"]))


def test_change_map_oracle_asserts_like_the_reference():
    with pytest.raises(AssertionError):                                 # max == min on a changed row: 0/0
        S.log_prob_to_change(torch.full((1, 8), -3.0), torch.zeros(1, 8) + torch.arange(8.0), 0.1)


@pytest.mark.parametrize("dt,tag,tol", [(torch.float64, "f64", 1e-13), (torch.float32, "f32", 1e-6)])
def test_co_unit_sphere_oracle_matches_reference(dt, tag, tol):
    z = np.load(os.path.join(GOLDEN, "stage_sphere.npz"))
    for ci in range(int(z["n_cases"])):
        o0, o1, far, mean = S.co_unit_sphere(torch.from_numpy(z[f"s{ci}_p0"]).to(dt), torch.from_numpy(z[f"s{ci}_p1"]).to(dt))
        assert (o0 - torch.from_numpy(z[f"s{ci}_o0_{tag}"])).abs().max().item() <= tol
        assert (o1 - torch.from_numpy(z[f"s{ci}_o1_{tag}"])).abs().max().item() <= tol
        assert abs(float(far) - float(z[f"s{ci}_far_{tag}"])) <= tol * 100 and (mean.numpy() - z[f"s{ci}_mean_{tag}"]).max() <= tol * 100
        joint = torch.cat((o0, o1))[:, :3]
        assert abs(joint.norm(dim=-1).max().item() - 1.0) < 1e-5 and joint.mean(0).abs().max().item() < 1e-5


def test_fps_oracle_is_farthest_point_sampling():
    """Unpinned at reference level (torch-cluster absent): check the defining property on the restatement instead."""
    g = np.random.default_rng(0)
    x = g.random((300, 6))
    idx = S.fps(x, 0.25)
    assert idx.shape == (75,) and idx[0] == 0 and len(set(idx.tolist())) == 75
    for j in range(1, 75):
        d = ((x[:, None, :] - x[idx[:j]][None]) ** 2).sum(-1).min(1)
        assert d[idx[j]] == d.max()
    grid = np.stack(np.meshgrid(np.arange(4.0), np.arange(4.0)), -1).reshape(-1, 2)      # ties: the lowest index must win
    assert S.fps(grid, 3 / 16).tolist() == [0, 15, 3]
