"""HIP kernels of the steps either side of the path, through the reference-shaped host API
(flowcompare_amd.change / flowcompare_amd.staging), against the reference's golden vectors and the pinned oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from flowcompare_amd import change, staging
from oracle import staging_oracle as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_change_map_matches_reference_golden():
    z = np.load(os.path.join(GOLDEN, "stage_change.npz"))
    for ci, (B, N, N0, multiple, cutoff, use_cutoff) in enumerate(z["cases"]):
        a, b = torch.from_numpy(z[f"c{ci}_lp10"]).to(DEV), torch.from_numpy(z[f"c{ci}_lp00"]).to(DEV)
        out = change.log_prob_to_change(a, b, float(multiple), hard_cutoff=float(cutoff) if use_cutoff else None).cpu().double()
        ref = torch.from_numpy(z[f"c{ci}_out_f64"])
        # the mask is a threshold decision: entries whose fp64 value sits within 1e-4 of the threshold may flip in fp32
        a64, b64 = torch.from_numpy(z[f"c{ci}_lp10_after_f64"]), torch.from_numpy(z[f"c{ci}_lp00_after_f64"])
        thr = torch.full((int(B), 1), float(cutoff), dtype=torch.float64) if use_cutoff else \
            b64.mean(-1, keepdim=True) - multiple * b64.std(-1, keepdim=True)
        safe = (a64 - thr).abs() > 1e-4
        assert safe.float().mean() > 0.99
        assert (out - ref)[safe].abs().max().item() < 2e-6
        assert (ref > 0).any()                                          # the fixture does contain a changed region
        # in-place clamping of the caller's tensors, like the reference
        assert torch.equal(a.cpu(), torch.from_numpy(z[f"c{ci}_lp10_after_f32"]))
        assert torch.equal(b.cpu(), torch.from_numpy(z[f"c{ci}_lp00_after_f32"]))


def test_change_map_shapes_and_errors():
    g = torch.Generator().manual_seed(0)
    l1, l0 = torch.randn(300, generator=g).to(DEV), torch.randn(300, generator=g).to(DEV)
    out = change.log_prob_to_change(l1.clone(), l0.clone(), 1.0)                      # 1-D in, 1-D out (test_flow.py squeezes B = 1)
    ref = S.log_prob_to_change(l1.cpu().double()[None], l0.cpu().double()[None], 1.0)[0]
    assert out.shape == (300,) and (out.cpu().double() - ref).abs().max().item() < 2e-6
    with pytest.raises(AssertionError):                                                # max == min on a changed row -> 0/0
        change.log_prob_to_change(torch.full((1, 8), -3.0, device=DEV), torch.arange(8.0, device=DEV)[None], 0.1)
    with pytest.raises(RuntimeError, match="GPU"):
        change.log_prob_to_change(torch.zeros(1, 8), torch.zeros(1, 8), 1.0)
    t = torch.tensor([1.0, float("inf"), -2.0, float("-inf")], device=DEV)
    assert change.clamp_infs(t) is t and t.tolist() == [1.0, -2.0, -2.0, -2.0]


def test_co_unit_sphere_matches_reference_golden():
    z = np.load(os.path.join(GOLDEN, "stage_sphere.npz"))
    for ci in range(int(z["n_cases"])):
        p0, p1 = torch.from_numpy(z[f"s{ci}_p0"]).to(DEV), torch.from_numpy(z[f"s{ci}_p1"]).to(DEV)
        o0, o1, inv = staging.co_unit_sphere(p0, p1, return_inverse=True)
        assert (o0.cpu().double() - torch.from_numpy(z[f"s{ci}_o0_f64"])).abs().max().item() < 1e-6
        assert (o1.cpu().double() - torch.from_numpy(z[f"s{ci}_o1_f64"])).abs().max().item() < 1e-6
        assert abs(float(inv["furthest_distance"]) - float(z[f"s{ci}_far_f64"])) < 1e-4
        assert np.abs(inv["mean"].cpu().numpy() - z[f"s{ci}_mean_f64"]).max() < 1e-4
        assert torch.equal(p0.cpu(), torch.from_numpy(z[f"s{ci}_p0"]))               # co_unit_sphere works on the concatenated copy
    p = torch.from_numpy(z["s1_p0"]).to(DEV)
    q, inv = staging.unit_sphere(p, return_inverse=True)                               # unit_sphere alone is in place (utils.py:262-265)
    ref, far, mean = S.unit_sphere(torch.from_numpy(z["s1_p0"]).double())
    assert q is p and (p.cpu().double() - ref).abs().max().item() < 1e-6 and abs(float(inv["furthest_distance"]) - float(far)) < 1e-4


@pytest.mark.parametrize("n,C,m", [(3000, 6, 1024), (517, 3, 517), (2000, 6, 1), (30000, 6, 64)])
def test_fps_matches_restatement(n, C, m):
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, C, generator=g)
    idx = staging.fps(x.to(DEV), torch.zeros(n, dtype=torch.long, device=DEV), ratio=m / n, random_start=False).cpu().numpy()
    ref = S.fps(x.numpy(), m / n)
    assert idx.dtype == np.int64 and idx.shape == ref.shape and np.array_equal(idx, ref)


def test_fps_ties_batches_and_errors():
    grid = torch.stack(torch.meshgrid(torch.arange(8.0), torch.arange(8.0), indexing="ij"), -1).reshape(-1, 2)
    idx = staging.fps(grid.to(DEV), None, ratio=6 / 64, random_start=False).cpu().numpy()
    assert np.array_equal(idx, S.fps(grid.numpy(), 6 / 64))                            # many exact ties: lowest index wins
    g = torch.Generator().manual_seed(1)
    x = torch.rand(3, 400, 6, generator=g)
    batch = torch.arange(3).repeat_interleave(400)
    idx = staging.fps(x.reshape(-1, 6).to(DEV), batch.to(DEV), ratio=0.1, random_start=False).cpu().numpy().reshape(3, -1)
    for b in range(3):
        assert np.array_equal(idx[b] - 400 * b, S.fps(x[b].numpy(), 0.1))
    with pytest.raises(NotImplementedError):
        staging.fps(grid.to(DEV), None, ratio=0.5)


def test_stage_pair_feeds_the_path():
    """The loader's last steps for one voxel pair, HIP vs restatement; the result is what inner_loop takes as extract_0 / extract_1."""
    g = torch.Generator().manual_seed(3)
    v0, v1 = torch.rand(5000, 6, generator=g) * 30 - 10, torch.rand(1500, 6, generator=g) * 8
    s0, s1, inv = staging.stage_pair(v0.to(DEV), v1.to(DEV), 1024, 512)
    r0, r1, far, mean = S.stage_pair(v0, v1, 1024, 512)
    assert s0.shape == (1024, 6) and s1.shape == (512, 6)
    assert (s0.cpu() - r0).abs().max().item() < 1e-6 and (s1.cpu() - r1).abs().max().item() < 1e-6
    assert abs(float(inv["furthest_distance"]) - float(far)) < 1e-4
