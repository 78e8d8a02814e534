"""Parity at BASELINE.json's full size (C2: 16 scenes x 4096 target + 4096 context points, 115 spline layers), where the
oracle cannot run the whole batch in test time: size-independent properties of the domain plus the oracle on rows of ONE full scene.

Weights: module-initialised under a fixed seed, then conditioned (flowcompare_amd/conditioning.py: near-identity coupling output
layers, LinearLU mixing, ActNorm first-batch statistics) so that the 115-layer stack is as well conditioned as a checkpoint and the
golden-fixture gates apply at full depth.

  * determinism: two runs are bit-identical;
  * scene independence: a 2-scene sub-batch reproduces its rows of the 16-scene batch bit for bit, so
  * the pinned oracle in fp64 on 512 target points of scene 0 (full 4096-point context) checks those rows of the 16-scene run with
    the gates of the golden fixtures: |bpd - bpd_fp64| < 1e-4, per point < 2e-3, mean < 3e-4;
  * invertibility: inverse(latent(x)) returns x (the augmented dims are dropped by the inverse, models/augmenter semantics).
"""
import time

import pytest
import torch

import flowcompare_amd as fa
from oracle import flow_oracle as O
from fullsize_util import (build_conditioned, check_spline_rows_against_fp64, hip_rows_with_decisions, oracle_flow_rows_forced, side_by_side, state_dicts,
                           synth_pairs)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, NPTS = 16, 4096


@pytest.fixture(scope="module")
def c2():
    cfg, md = build_conditioned("c2_dgcnn_attn_spline", NPTS, DEV)
    e0, e1, _, eps = synth_pairs(B, NPTS, NPTS, 12)
    return cfg, md, e0, e1, eps


def test_full_size_determinism_and_scene_independence(c2):
    cfg, md, e0, e1, eps = c2
    batch = (e0.to(DEV), e1.to(DEV), None)
    _, lp1, bpd1 = fa.inner_loop(batch, md, cfg, eps=[eps.to(DEV)])
    _, lp2, _ = fa.inner_loop(batch, md, cfg, eps=[eps.to(DEV)])
    assert lp1.shape == (B, NPTS) and torch.isfinite(lp1).all()
    assert torch.equal(lp1, lp2)
    sub = slice(5, 7)
    _, lps, _ = fa.inner_loop((e0[sub].to(DEV), e1[sub].to(DEV), None), md, cfg, eps=[eps[sub].to(DEV)])
    assert torch.equal(lps, lp1[sub])


def test_full_size_rows_match_fp64_oracle_at_golden_gates(c2):
    """Target points do not interact (each attends to the context only), so the oracle on the first 512 target points of scene 0
    against the FULL 4096-point context checks those rows of the 16-scene, 115-layer run -- at the tolerance north_star states, on
    EVERY row: bpd within 1e-4 (mean nats within 4.2e-4) of the fp64 oracle, every row within 2e-3 or the reference arithmetic's own gap.
    The spline's inside / outside decisions |x2| <= 3 (a 0.366-nat jump of log p, models/spline_coupling.py:35-48) are read from the HIP
    run (engine trace of the x2 the coupling kernels read) and forced on the fp64 oracle, so rows whose fp32 latent lands on the other
    side of the boundary than the fp64 latent are ordinary rows; the oracle's fp32 run is judged the same way against ITS decisions."""
    cfg, md, e0, e1, eps = c2
    n = 512
    _, lp, _ = fa.inner_loop((e0.to(DEV), e1.to(DEV), None), md, cfg, eps=[eps.to(DEV)])
    ctx_dev = md["input_embedder"](e0[:1].to(DEV))
    lp_rows, dec_hip = hip_rows_with_decisions(cfg, md, ctx_dev, e1[:1, :n], None, [eps[:1, :n]])
    assert torch.equal(lp_rows, lp[0, :n].cpu()), "rows of a 512-row run differ from the same rows of the 16-scene run"
    ctx = ctx_dev.cpu()
    t0 = time.time()
    args = (cfg, md, ctx, e1[:1, :n], None, [eps[:1, :n]])
    (lp64_nat, dec64), (lp64_hip, _), (lp32, dec32) = side_by_side(           # three independent passes on threads of their own
        lambda: oracle_flow_rows_forced(*args, torch.float64),
        lambda: oracle_flow_rows_forced(*args, torch.float64, forced=dec_hip),
        lambda: oracle_flow_rows_forced(*args, torch.float32))
    lp64_ref, _ = oracle_flow_rows_forced(*args, torch.float64, forced=dec32)
    print(f"oracle: {time.time() - t0:.0f} s of host time")
    check_spline_rows_against_fp64("C2 16 x 4096 x 115 spline layers, scene 0 rows 0..511", lp_rows, dec_hip, lp64_hip, lp64_nat, dec64, lp32, lp64_ref)


def test_full_size_rows_match_the_oracle_end_to_end_with_its_own_embedder(c2):
    """The same rows END TO END: HIP embedder + HIP flow against the oracle's OWN fp64 context embedding + fp64 flow (HIP spline decisions
    forced, as above).  k-NN near-ties flip single neighbours of single context points (test_full_size_embedder...: a handful of the
    4096 embedding rows move by up to a few 1e-3); a target row sees them through a softmax over all 4096 keys, so every row stays gated."""
    cfg, md, e0, e1, eps = c2
    n = 256
    ctx_dev = md["input_embedder"](e0[:1].to(DEV))
    lp_rows, dec_hip = hip_rows_with_decisions(cfg, md, ctx_dev, e1[:1, :n], None, [eps[:1, :n]])
    _, sd_e = state_dicts(md, torch.float64)
    t0 = time.time()
    with torch.no_grad():
        ctx64 = O.context_embed(cfg, sd_e, e0[:1].double())
    moved = (ctx_dev.cpu().double() - ctx64).abs().amax(-1)[0]
    print(f"context rows whose embedding differs from the fp64 oracle's by more than 1e-4 (k-NN near-ties): {int((moved > 1e-4).sum())} of {moved.numel()}")
    _, se32 = state_dicts(md, torch.float32)

    def fp32_pass():
        with torch.no_grad():
            ctx32 = O.context_embed(cfg, se32, e0[:1])
        return oracle_flow_rows_forced(cfg, md, ctx32, e1[:1, :n], None, [eps[:1, :n]], torch.float32)
    (lp64_hip, _), (lp32, dec32) = side_by_side(
        lambda: oracle_flow_rows_forced(cfg, md, ctx64, e1[:1, :n], None, [eps[:1, :n]], torch.float64, forced=dec_hip), fp32_pass)
    lp64_ref, _ = oracle_flow_rows_forced(cfg, md, ctx64, e1[:1, :n], None, [eps[:1, :n]], torch.float64, forced=dec32)
    print(f"oracle: {time.time() - t0:.0f} s of host time")
    check_spline_rows_against_fp64("C2 end to end (oracle's own fp64 embedder), scene 0 rows 0..255", lp_rows, dec_hip, lp64_hip, None, None, lp32, lp64_ref, end_to_end=True)


def test_full_size_embedder_matches_fp64_oracle(c2):
    """DGCNN context embedder on one full 4096-point scene against the oracle in fp64 (k-NN near-ties can flip a neighbour, which moves
    single rows: judged by quantile and by the worst row)."""
    cfg, md, e0, e1, eps = c2
    emb = md["input_embedder"](e0[:1].to(DEV)).cpu().double()
    _, sd_e = state_dicts(md, torch.float64)
    with torch.no_grad():
        ref = O.context_embed(cfg, sd_e, e0[:1].double())
    d = (emb - ref).abs().amax(-1)[0]
    print(f"C2 embedder, 4096 points: per-row max |hip - fp64| median {d.median():.2e} q99 {d.quantile(0.99):.2e} max {d.max():.2e}")
    assert d.quantile(0.99).item() < 2e-5 and d.max().item() < 5e-3


def test_full_size_inverse_round_trip(c2):
    cfg, md, e0, e1, eps = c2
    emb = md["input_embedder"](e0.to(DEV))
    h = md["flow"]._engine()
    x = e1.to(DEV)
    _, z = h.log_prob(x, emb, None, [eps.to(DEV)], return_latent=True)
    xr = h.inverse(z, emb, None, [])
    err = (xr - x).abs()
    print(f"full-size inverse(latent(x)) - x: max {err.max().item():.2e} mean {err.mean().item():.2e} over {x.numel()} coordinates")
    # 115 layers forward + 115 backward: the worst of 393 216 coordinates carries the row-wise error growth of conditioning.py twice
    assert torch.isfinite(xr).all() and err.max().item() < 3e-2 and err.mean().item() < 1e-3


def test_full_size_gradients_are_additive_over_scenes_and_reproducible(c2):
    """Backward at the full layer count and cloud size (115 spline layers, 4096 + 4096 points, 4 scenes, embedder frozen so that no
    BatchNorm batch statistic couples the scenes): the loss is a sum over scenes, so the gradient of the 4-scene batch equals the sum
    of the gradients of its two halves -- a size-independent check of every backward kernel at sizes the oracle cannot reach -- and two
    runs of the same step are bit-identical."""
    from flowcompare_amd import train_flow as TF
    from flowcompare_amd import train_ops as T
    cfg, md, e0, e1, eps = c2
    flow = md["flow"]
    for m in flow.modules():
        if hasattr(m, "initialized"):
            m.initialized.fill_(1.0)
    n_sc = 4
    with torch.no_grad():
        ctx = md["input_embedder"](e0[:n_sc].to(DEV))
    x, ee = e1[:n_sc].to(DEV), eps[:n_sc].to(DEV)
    n_pts = float(n_sc * NPTS)

    def grads(sl):
        flow.zero_grad()
        with T.step_guard(device=DEV) as guard:
            lp = TF.flow_log_prob(flow, x[sl], ctx[sl], None, [ee[sl]])
            (-lp.sum() / n_pts).backward()
            assert not guard.overflowed()
        return {n: p.grad.clone() for n, p in flow.named_parameters() if p.grad is not None}
    g_all, g_again = grads(slice(0, 4)), grads(slice(0, 4))
    assert all(torch.equal(g_all[n], g_again[n]) for n in g_all)
    g_a, g_b = grads(slice(0, 2)), grads(slice(2, 4))
    num = sum(float(((g_all[n] - g_a[n] - g_b[n]).double() ** 2).sum()) for n in g_all) ** 0.5
    den = sum(float((g_all[n].double() ** 2).sum()) for n in g_all) ** 0.5
    print(f"full-size backward, 4 scenes x 4096 points x 115 layers: |g(all) - g(first half) - g(second half)| / |g| = {num / den:.2e}; "
          f"{len(g_all)} parameter tensors, |g| = {den:.3e}")
    assert torch.isfinite(torch.tensor(den)) and num / den < 1e-5
    flow.zero_grad()
