"""Pins the CPU oracle (oracle/flow_oracle.py) to golden vectors produced by RUNNING THE REFERENCE
(tests/golden/gen_golden.py).  CPU only; this is what makes the oracle a trustworthy checker for the HIP engine."""
import numpy as np
import pytest
import torch

from conftest import E2E_REAL, E2E_TINY, Fixture
from oracle import flow_oracle as O


def _batch(fx, dtype):
    return fx.t("extract_0", dtype), fx.t("extract_1", dtype), fx.t("extra", dtype)


@pytest.mark.parametrize("name", E2E_REAL + E2E_TINY)
def test_forward_fp64_matches_reference(name):
    fx = Fixture(name)
    cfg = fx.derived_cfg()
    sd_flow, sd_emb = fx.state_dicts(torch.float64)
    e0, e1, ex = _batch(fx, torch.float64)
    with torch.no_grad():
        emb = O.context_embed(cfg, sd_emb, e0)
        np.testing.assert_allclose(emb.numpy(), fx.a["emb_f64"], rtol=1e-9, atol=1e-9)
        loss, lp, bpd = O.inner_loop(cfg, sd_flow, sd_emb, (e0, e1, ex), fx.eps(torch.float64))
    np.testing.assert_allclose(lp.numpy(), fx.a["log_prob_f64"], rtol=1e-9, atol=1e-8)
    assert abs(float(bpd) - float(fx.a["bpd_f64"])) < 1e-10
    assert abs(float(loss) - float(fx.a["loss_f64"])) < 1e-9


@pytest.mark.parametrize("name", E2E_REAL + E2E_TINY)
def test_per_transform_records_fp64(name):
    fx = Fixture(name)
    cfg = fx.derived_cfg()
    sd_flow, sd_emb = fx.state_dicts(torch.float64)
    e0, e1, ex = _batch(fx, torch.float64)
    rec = []
    with torch.no_grad():
        emb = O.context_embed(cfg, sd_emb, e0)
        if emb.dim() == 2:
            emb = emb[:, None, :].expand(-1, e1.shape[1], -1)
        extra = None if ex is None else ex[:, None, :].expand(-1, e1.shape[1], -1)
        O.flow_log_prob(cfg, sd_flow, e1, emb, extra, fx.eps(torch.float64), record=rec)
    ldj = torch.stack([torch.as_tensor(r[1]).expand(e1.shape[:2]) for r in rec]).numpy()
    np.testing.assert_allclose(ldj, fx.a["ldj_f64"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rec[-1][0][:, :8].numpy(), fx.a["z_last_f64"], rtol=1e-9, atol=1e-9)
    if "z_sub_f64" in fx.a:
        z = torch.stack([r[0][:, :8] for r in rec]).numpy()
        np.testing.assert_allclose(z, fx.a["z_sub_f64"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name", E2E_REAL + E2E_TINY)
def test_forward_fp32_close_to_reference_fp32(name):
    """fp32 oracle vs the reference's own fp32 run: same op sequence, so only BLAS summation order differs."""
    fx = Fixture(name)
    cfg = fx.derived_cfg()
    sd_flow, sd_emb = fx.state_dicts(torch.float32)
    with torch.no_grad():
        _, lp, bpd = O.inner_loop(cfg, sd_flow, sd_emb, _batch(fx, torch.float32), fx.eps(torch.float32))
    assert np.abs(lp.numpy() - fx.a["log_prob_f32"]).max() < 2e-3
    assert abs(float(bpd) - float(fx.a["bpd_f64"])) < 1e-4          # the north-star gate, on the logged scalar


@pytest.mark.parametrize("name", E2E_REAL + E2E_TINY)
def test_inverse_fp64_matches_reference(name):
    fx = Fixture(name)
    cfg = fx.derived_cfg()
    sd_flow, sd_emb = fx.state_dicts(torch.float64)
    e0, _, ex = _batch(fx, torch.float64)
    z = fx.t("sample_z", torch.float64)
    with torch.no_grad():
        x = O.make_sample(cfg, sd_flow, sd_emb, z, e0[:1], None if ex is None else ex[:1], fx.eps(torch.float64, "inveps"))
    np.testing.assert_allclose(x.numpy(), fx.a["sample_x_f64"], rtol=1e-8, atol=1e-8)


def test_spline_op_fixture():
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "op_spline.npz"))
    x, w, h, d = (torch.from_numpy(z[k]) for k in ("x", "w", "h", "d"))
    y, lad = O.rq_spline(x, w, h, d)
    np.testing.assert_allclose(y.numpy(), z["y"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(lad.numpy(), z["logabsdet"], rtol=1e-11, atol=1e-12)
    xi, ladi = O.rq_spline(torch.from_numpy(z["y"]), w, h, d, inverse=True)
    np.testing.assert_allclose(xi.numpy(), z["x_inv"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(ladi.numpy(), z["logabsdet_inv"], rtol=1e-10, atol=1e-11)
    y32, lad32 = O.rq_spline(x.float(), w.float(), h.float(), d.float())
    assert np.abs(y32.numpy() - z["y_f32"]).max() < 1e-5
    assert np.abs(lad32.numpy() - z["logabsdet_f32"]).max() < 1e-4


def test_knn_op_fixture():
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "op_knn.npz"))
    for tag, k in (("xyzrgb", 40), ("feat64", 40)):
        x = torch.from_numpy(z[f"{tag}_x"])                      # [B,C,M] channels-first like the reference
        f = x.transpose(1, 2).contiguous()
        idx = O.knn_indices(f.double(), k)
        ref = torch.from_numpy(z[f"{tag}_idx_f64"].astype(np.int64))
        assert (idx.sort(-1)[0] == ref.sort(-1)[0]).all()
        idx32 = O.knn_indices(f, k)
        ref32 = torch.from_numpy(z[f"{tag}_idx_f32"].astype(np.int64))
        same = (idx32.sort(-1)[0] == ref32.sort(-1)[0]).all(-1)
        margin = torch.from_numpy(z[f"{tag}_margin_f64"])
        assert (same | (margin < 1e-5)).all()                    # only near-ties at the k-th boundary may differ


def test_paconv_embedder_fixture():
    """PAConv U-Net restatement vs the reference's own Python run on CPU (with the six CUDA pointops kernels substituted by
    oracle/paconv_oracle.py's restatements: those six are pinned only by restatement, see that file's header)."""
    from oracle import paconv_oracle as P
    fx = Fixture("emb_paconv")
    _, sd_emb = fx.state_dicts(torch.float64)
    with torch.no_grad():
        emb = P.paconv_embed(sd_emb, fx.t("pts", torch.float64))
    np.testing.assert_allclose(emb.numpy(), fx.a["emb_f64"], rtol=1e-9, atol=1e-9)
    _, sd32 = fx.state_dicts(torch.float32)
    with torch.no_grad():
        emb32 = P.paconv_embed(sd32, fx.t("pts", torch.float32))
    assert np.abs(emb32.numpy() - fx.a["emb_f64"]).max() < 1e-5


def test_pointops_restatements_basic_properties():
    from oracle import paconv_oracle as P
    g = torch.Generator().manual_seed(0)
    xyz = torch.rand(2, 200, 3, generator=g)
    idx = P.furthest_sampling(xyz, 50)
    assert (idx[:, 0] == 0).all() and all(len(set(r.tolist())) == 50 for r in idx)          # starts at 0, no repeats
    d = torch.cdist(xyz, xyz)
    for b in range(2):                                                                       # greedy max-min property
        chosen = [0]
        for j in range(1, 50):
            mind = d[b][:, chosen].min(1)[0]
            assert abs(float(mind[idx[b, j]]) - float(mind.max())) < 1e-6
            chosen.append(int(idx[b, j]))
    q = xyz[:, :7]
    nn = P.knnquery_heap(32, xyz, q)
    assert (nn[:, :, 0] == torch.arange(7)).all()                                            # a point is its own nearest neighbour
    nn_small = P.knnquery_heap(32, xyz[:, :5], q[:, :2])
    assert (nn_small[..., 5:] == 0).all()                                                    # unfilled heap slots keep index 0
    dist, i3 = P.nearest_neighbor3(xyz, xyz[:, :1])
    assert torch.isinf(dist[..., 1:]).all() and (i3 == 0).all()                              # fewer than 3 known points


def test_spline_decision_hook_records_and_forces_without_changing_the_natural_run():
    """oracle.spline_decisions (full-depth parity helper): recording changes nothing; forcing a run's own decisions reproduces it bit for
    bit; forcing an input just beyond the boundary inside evaluates it at the boundary knot (at -3: derivative 0.6936 -> log-det -0.366), and
    forcing one just inside the boundary outside passes it through with log-det 0."""
    import math
    from oracle import flow_oracle as O
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 7, generator=g, dtype=torch.float64) * 2.5
    uw, uh, ud = (torch.randn(4, 7, n, generator=g, dtype=torch.float64) for n in (8, 8, 9))
    y0, l0 = O.rq_spline(x, uw, uh, ud)
    with O.spline_decisions() as rec:
        y1, l1 = O.rq_spline(x, uw, uh, ud)
    assert torch.equal(y0, y1) and torch.equal(l0, l1) and len(rec) == 1 and rec[0].dtype == torch.bool
    assert torch.equal(rec[0], (x >= -3) & (x <= 3)) and (~rec[0]).any() and rec[0].any()
    with O.spline_decisions(forced=rec) as rec2:
        y2, l2 = O.rq_spline(x, uw, uh, ud)
    assert torch.equal(y0, y2) and torch.equal(l0, l2) and torch.equal(rec2[0], rec[0])
    xb = torch.tensor([[3.0 + 1e-7, 3.0 - 1e-7, -3.0 - 1e-7]], dtype=torch.float64)
    p = [torch.randn(1, 3, n, generator=g, dtype=torch.float64) for n in (8, 8, 9)]
    with O.spline_decisions(forced=[torch.tensor([[True, False, True]])]):
        yb, lb = O.rq_spline(xb, *p)
    jump = math.log(math.log1p(math.exp(-1e-3)) + 1e-3)
    # (the reference's derivative padding makes only knot 0 the constant: the upper boundary knot carries the last learned derivative)
    assert abs(float(yb[0, 0]) - 3.0) < 1e-9 and math.isfinite(float(lb[0, 0]))
    assert float(yb[0, 1]) == float(xb[0, 1]) and float(lb[0, 1]) == 0.0
    assert abs(float(yb[0, 2]) + 3.0) < 1e-9 and abs(float(lb[0, 2]) - jump) < 1e-6


def test_spline_decision_hook_is_per_thread():
    """The full-depth parity tests run independent oracle passes side by side (tests/fullsize_util.py::side_by_side): a thread's recorder /
    forced decisions must not be seen by another thread's pass."""
    import threading
    from oracle import flow_oracle as O
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 7, generator=g, dtype=torch.float64) * 2.5
    p = [torch.randn(64, 7, n, generator=g, dtype=torch.float64) for n in (8, 8, 9)]
    y0, l0 = O.rq_spline(x, *p)
    all_out = torch.zeros(64, 7, dtype=torch.bool)
    out, start = {}, threading.Barrier(2)

    def natural():
        start.wait()
        with O.spline_decisions() as rec:
            for _ in range(50):
                y, l = O.rq_spline(x, *p)
        out["nat"] = (y, l, len(rec))

    def forced():
        start.wait()
        with O.spline_decisions(forced=[all_out] * 50) as rec:
            for _ in range(50):
                y, l = O.rq_spline(x, *p)
        out["forced"] = (y, l, len(rec))
    ts = [threading.Thread(target=natural), threading.Thread(target=forced)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert torch.equal(out["nat"][0], y0) and torch.equal(out["nat"][1], l0) and out["nat"][2] == 50
    assert torch.equal(out["forced"][0], x) and float(out["forced"][1].abs().max()) == 0.0 and out["forced"][2] == 50
