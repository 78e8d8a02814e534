"""Shared pieces of the full-size GPU parity tests (tests/test_gpu_fullsize.py, tests/test_gpu_configs.py): synthetic cloud pairs
as bench.py draws them, conditioned 115-layer weights (flowcompare_amd/conditioning.py) and the comparison of rows of a full-size HIP
run against the pinned oracle in fp64.  Gates: |bpd - bpd_fp64| < 1e-4 on the logged scalar (absolute: north_star's tolerance, as in
tests/test_gpu_flow.py); per row the golden-fixture gates (|log p - log p_fp64| < 2e-3, mean < 3e-4) OR, where 115 chained layers put
the reference's own fp32 arithmetic beyond them (flowcompare_amd/conditioning.py: error growth along single rows), no further from
fp64 than the oracle's fp32 run is on the same rows -- both distances are printed side by side."""
import contextlib
import io
import math
import time

import torch

import flowcompare_amd as fa
from flowcompare_amd.conditioning import condition_flow
from oracle import flow_oracle as O

BPD_GATE, POINT_GATE, MEAN_GATE = 1e-4, 2e-3, 3e-4
SPLINE_MARGIN = 1e-4        # points whose fp64 trajectory passes this close to the spline's +-3 domain boundary: near-tie rule below


def synth_pairs(B, n_ctx, n_tgt, seed, noise_width=294):
    """xyz ~ U(-1,1)^3, pair-centred and scaled to the joint unit sphere; rgb ~ U[0,1); extra ~ U(0,15); eps ~ N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, n_ctx + n_tgt, 3, generator=g) * 2 - 1
    xyz = xyz - xyz.mean(1, keepdim=True)
    xyz = xyz / xyz.norm(dim=-1).amax(1)[:, None, None]
    pts = torch.cat((xyz, torch.rand(B, n_ctx + n_tgt, 3, generator=g)), -1)
    extra = torch.rand(B, 1, generator=g) * 15.0
    eps = torch.randn(B, n_tgt, noise_width, generator=g)
    return pts[:, :n_ctx].contiguous(), pts[:, n_ctx:].contiguous(), extra, eps


def build_conditioned(name, points, dev, weight_seed=11, cond_seed=999, cond_scenes=2, cond_points=None, **over):
    """initialize_flow under a fixed seed, then condition_flow on `cond_scenes` synthetic scenes (ActNorm first-batch statistics)."""
    cfg = fa.named_config(name, sample_size=points, **over)
    torch.manual_seed(weight_seed)
    with contextlib.redirect_stdout(io.StringIO()):
        md = fa.initialize_flow(cfg, device=dev, mode="test")
    n = cond_points or points
    c0, c1, cx, ce = synth_pairs(cond_scenes, n, n, cond_seed, cfg["latent_dim"] - cfg["input_dim"])
    ccfg = dict(cfg)
    ccfg["sample_size"] = n
    condition_flow(md, ccfg, (c0.to(dev), c1.to(dev), cx.to(dev) if cfg["extra_z_value_context"] else None), eps=[ce.to(dev)])
    return cfg, md


def state_dicts(md, dtype):
    f = {k: (v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu()) for k, v in md["flow"].state_dict().items()}
    e = {k: (v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu()) for k, v in md["input_embedder"].state_dict().items()}
    return f, e


def oracle_flow_rows(cfg, md, ctx, x, extra, eps, dtype):
    """The oracle's Flow.log_prob on given target rows x [1, n, 6] against a given context embedding ctx [1, M, E] (the HIP embedder's
    output, so that flow parity is judged on identical conditioning).  Returns (log_prob [n], spline boundary margin [n])."""
    sd_f, _ = state_dicts(md, dtype)
    n = x.shape[1]
    ex = None if extra is None else extra.to(dtype)[:, None, :].expand(-1, n, -1)
    rec = []
    with torch.no_grad():
        lp = O.flow_log_prob(cfg, sd_f, x.to(dtype), ctx.to(dtype), ex, [e.to(dtype) for e in eps], record=rec)
        margin = O.spline_domain_margin(cfg, rec)
    return lp[0], margin[0]


def check_rows_against_fp64(label, lp_hip, lp64, lp32, margin, input_dim=6):
    """Gates of tests/test_gpu_flow.py on rows of a full-size run.  Rows whose fp64 trajectory comes within SPLINE_MARGIN of the +-3
    spline boundary are judged by the near-tie rule: they may differ by whole multiples of the reference's 0.366-nat boundary jump."""
    lp_hip, lp64, lp32 = lp_hip.double(), lp64.double(), lp32.double()
    far = margin > SPLINE_MARGIN
    d_hip, d_ref = (lp_hip - lp64).abs(), (lp32 - lp64).abs()
    k = math.log2(math.e) / input_dim
    bpd_hip = abs(float((lp_hip[far].mean() - lp64[far].mean()) * k))
    bpd_ref = abs(float((lp32[far].mean() - lp64[far].mean()) * k))
    print(f"{label}: {int(far.sum())}/{far.numel()} rows away from the spline boundary; mean nats {float(lp64.mean()):.3f}\n"
          f"    |hip - fp64|          max {float(d_hip[far].max()):.2e} mean {float(d_hip[far].mean()):.2e} bpd {bpd_hip:.2e}\n"
          f"    |oracle fp32 - fp64|  max {float(d_ref[far].max()):.2e} mean {float(d_ref[far].mean()):.2e} bpd {bpd_ref:.2e}   (the reference arithmetic's own gap)")
    assert torch.isfinite(lp_hip).all()
    assert far.float().mean() > 0.6
    assert bpd_hip < BPD_GATE, f"{label}: bpd differs from fp64 by {bpd_hip:.2e}"
    assert float(d_hip[far].mean()) < max(MEAN_GATE, float(d_ref[far].mean()))
    assert float(d_hip[far].max()) < max(POINT_GATE, float(d_ref[far].max()))
    assert float(d_hip[far].median()) < MEAN_GATE
    near = ~far
    if near.any():                                           # near-tie rows: a whole number of boundary jumps (0.3659 nats) apart, at most
        jump = -math.log(math.log1p(math.exp(-1e-3)) + 1e-3)
        r = d_hip[near] / jump
        print(f"    {int(near.sum())} near-boundary rows: {int((r.round() > 0).sum())} of them differ by whole boundary jumps (at most {int(r.round().max())})")
        assert float((r - r.round()).abs().max() * jump) < max(4 * POINT_GATE, float(d_ref[far].max())) and float(r.max()) < 3.5
    return bpd_hip, float(d_hip[far].max())
