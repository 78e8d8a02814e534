"""Every BASELINE.json configuration at its stated size (C2 is tests/test_gpu_fullsize.py), 115 layers, conditioned weights
(flowcompare_amd/conditioning.py), against the pinned oracle in fp64 with the golden-fixture gates (tests/fullsize_util.py):

  C1  DGCNN-global + affine, batch 2 x 1024 + 1024 points: the WHOLE batch against the oracle's inner_loop (embedder + flow) -- the one
      configuration the CPU oracle runs in full in seconds (BASELINE.json configs[0], the reference's CPU-runnable case);
  C3  PAConv + attention + affine, 16 x 4096: the PAConv embedder on one full scene against oracle/paconv_oracle.py (FPS 4096 -> 1024 ->
      256 -> 64 -> 16, 32-NN over 4096 points, 3-NN feature propagation), the 115-layer flow on 256 rows;
  C4  DGCNN + attention + extra context + affine, 8 scenes x 4096 per GPU: rows of scene 0 + determinism + scene independence;
  C5  C4's model at 16 scenes x 16384 + 16384 points per GPU: k-NN and attention at M = 16384 against fp64 (near-tie rule), the DGCNN
      embedder on one full scene, the flow on 256 rows of scene 0 against the full 16384-point context, determinism, scene independence.
"""
import time

import pytest
import torch

import flowcompare_amd as fa
from flowcompare_amd import engine
from oracle import flow_oracle as O
from fullsize_util import build_conditioned, check_rows_against_fp64, oracle_flow_rows, side_by_side, state_dicts, synth_pairs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(cfg, md, e0, e1, extra, eps, sl=slice(None)):
    batch = (e0[sl].to(DEV), e1[sl].to(DEV), extra[sl].to(DEV) if cfg["extra_z_value_context"] else None)
    _, lp, bpd = fa.inner_loop(batch, md, cfg, eps=[eps[sl].to(DEV)])
    return lp, bpd


def _rows_vs_oracle(label, cfg, md, e0, e1, extra, eps, lp, n):
    """Rows 0..n-1 of scene 0 of a HIP run against the oracle's flow in fp64 (and fp32, printed) on the HIP embedder's context."""
    ctx = md["input_embedder"](e0[:1].to(DEV)).cpu()
    ex = extra[:1] if cfg["extra_z_value_context"] else None
    c = dict(cfg)
    c["sample_size"] = n
    t0 = time.time()
    (lp64, margin), (lp32, _) = side_by_side(                              # the fp64 and the fp32 pass on threads of their own
        lambda: oracle_flow_rows(c, md, ctx, e1[:1, :n], ex, [eps[:1, :n]], torch.float64),
        lambda: oracle_flow_rows(c, md, ctx, e1[:1, :n], ex, [eps[:1, :n]], torch.float32))
    print(f"oracle: {time.time() - t0:.0f} s of host time")
    return check_rows_against_fp64(label, lp[0, :n].cpu(), lp64, lp32, margin)


# ------------------------------------------------------------------------------------------------ C1
def test_c1_whole_batch_against_the_oracle_in_full():
    B, N = 2, 1024
    cfg, md = build_conditioned("c1_dgcnn_global_affine", N, DEV)
    e0, e1, extra, eps = synth_pairs(B, N, N, 21)
    lp, bpd = _run(cfg, md, e0, e1, extra, eps)
    lp2, _ = _run(cfg, md, e0, e1, extra, eps)
    assert torch.equal(lp, lp2)
    sd_f, sd_e = state_dicts(md, torch.float64)
    sf32, se32 = state_dicts(md, torch.float32)
    t0 = time.time()
    def pass64():
        with torch.no_grad():                                              # (grad mode is per thread)
            return O.inner_loop(cfg, sd_f, sd_e, (e0.double(), e1.double(), None), [eps.double()])

    def pass32():
        with torch.no_grad():
            return O.inner_loop(cfg, sf32, se32, (e0, e1, None), [eps])
    (_, lp64, bpd64), (_, lp32, _) = side_by_side(pass64, pass32)
    print(f"oracle inner_loop (embedder + 115 layers, 2 x 1024 points, fp64 and fp32): {time.time() - t0:.0f} s of host time")
    margin = torch.full((B * N,), float("inf"), dtype=torch.float64)
    check_rows_against_fp64("C1 2 x 1024 x 115 affine layers, global context, whole batch", lp.cpu().reshape(-1), lp64.reshape(-1),
                            lp32.reshape(-1), margin)
    assert abs(float(bpd) - float(bpd64)) < 1e-4


# ------------------------------------------------------------------------------------------------ C3
def test_c3_paconv_embedder_and_flow_at_4096_points():
    from oracle import paconv_oracle as P
    B, N = 16, 4096
    cfg, md = build_conditioned("c3_paconv_attn_affine", N, DEV)
    e0, e1, extra, eps = synth_pairs(B, N, N, 31)
    lp, _ = _run(cfg, md, e0, e1, extra, eps)
    lp2, _ = _run(cfg, md, e0, e1, extra, eps)
    assert torch.equal(lp, lp2)
    lps, _ = _run(cfg, md, e0, e1, extra, eps, slice(3, 5))
    assert torch.equal(lps, lp[3:5])
    # embedder: one full scene.  The index kernels (FPS, 32-NN, 3-NN) work on fp32 coordinates with the reference kernels' own
    # arithmetic, so the restatement runs in fp32 too (an fp64 run may legitimately pick other samples at distance ties).
    emb = md["input_embedder"](e0[:1].to(DEV)).cpu()
    _, se32 = state_dicts(md, torch.float32)
    t0 = time.time()
    with torch.no_grad():
        ref = P.paconv_embed(se32, e0[:1])
    d = (emb - ref).abs().amax(-1)[0]
    scale = float(ref.abs().max())
    print(f"C3 PAConv embedder, 4096 points ({time.time() - t0:.0f} s of host time): per-row max |hip - oracle fp32| median {d.median():.2e} "
          f"q99 {d.quantile(0.99):.2e} max {d.max():.2e} (|emb| max {scale:.2f})")
    assert d.quantile(0.99).item() < 1e-4 * max(1.0, scale) and d.max().item() < 2e-3 * max(1.0, scale)
    fps = engine.op_fps(e0[:1, :, :3].to(DEV), 1024).cpu().long()
    assert torch.equal(fps, P.furthest_sampling(e0[:1, :, :3], 1024))
    _rows_vs_oracle("C3 16 x 4096 x 115 affine layers (PAConv context), scene 0 rows 0..255", cfg, md, e0, e1, extra, eps, lp, 256)


# ------------------------------------------------------------------------------------------------ C4
def test_c4_extra_context_at_8_scenes_of_4096_points():
    B, N = 8, 4096
    cfg, md = build_conditioned("c4_dgcnn_attn_extra_affine", N, DEV)
    e0, e1, extra, eps = synth_pairs(B, N, N, 41)
    lp, _ = _run(cfg, md, e0, e1, extra, eps)
    lp2, _ = _run(cfg, md, e0, e1, extra, eps)
    assert torch.equal(lp, lp2)
    lps, _ = _run(cfg, md, e0, e1, extra, eps, slice(2, 4))
    assert torch.equal(lps, lp[2:4])
    _rows_vs_oracle("C4 8 x 4096 x 115 affine layers + extra context, scene 0 rows 0..511", cfg, md, e0, e1, extra, eps, lp, 512)


def test_reference_native_training_shape_20_scenes_of_1024_target_and_1250_context_points():
    """The batch every FlowCompare user runs (config/dulcet-universe.yaml:1-3, 44-46, 185-187: batch_size 20, sample_size 1024, 1250 context
    points; N != M): dulcet's model (DGCNN + attention + extra context, 115 affine layers), the whole shape through the HIP path; rows of scene
    0 against the fp64 oracle on the full 1250-point context, determinism, and scenes 7..8 alone reproduce their rows of the 20-scene batch."""
    B, N, M = 20, 1024, 1250
    cfg, md = build_conditioned("c4_dgcnn_attn_extra_affine", N, DEV, cond_points=N)
    e0, e1, extra, eps = synth_pairs(B, M, N, 43)
    assert e0.shape == (B, M, 6) and e1.shape == (B, N, 6)
    lp, _ = _run(cfg, md, e0, e1, extra, eps)
    lp2, _ = _run(cfg, md, e0, e1, extra, eps)
    assert lp.shape == (B, N) and torch.isfinite(lp).all() and torch.equal(lp, lp2)
    lps, _ = _run(cfg, md, e0, e1, extra, eps, slice(7, 9))
    assert torch.equal(lps, lp[7:9])
    _rows_vs_oracle("native shape 20 x 1024 target / 1250 context x 115 affine layers + extra context, scene 0 rows 0..511", cfg, md, e0, e1, extra, eps, lp, 512)


# ------------------------------------------------------------------------------------------------ C5
M5 = 16384


def test_c5_knn_at_16384_points_against_fp64():
    for C, seed in ((6, 51), (64, 52)):                  # (M >= 2048: the matrix-core kernel, csrc/knn.hip)
        g = torch.Generator().manual_seed(seed)
        f = torch.rand(1, M5, C, generator=g) * 2 - 1
        idx = engine.op_knn(f.to(DEV), 40).cpu().long()
        assert idx.min() >= 0 and idx.max() < M5
        srt = idx.sort(-1)[0]
        assert (srt[..., 1:] != srt[..., :-1]).all(), "duplicate neighbours"
        fd = f.double()
        sq = (fd ** 2).sum(-1)
        pd = -sq[:, None, :] + 2 * fd @ fd.transpose(1, 2) - sq[:, :, None]
        top = pd.topk(40, dim=-1)
        kth = top.values[..., -1]
        got = torch.gather(pd, 2, idx)
        exact = (srt == top.indices.sort(-1)[0]).all(-1).float().mean().item()
        print(f"k-NN at M = {M5}, C = {C}: {100 * exact:.2f} % of the rows equal the fp64 top-40 set exactly; the rest differ at fp32 near-ties")
        assert (got >= kth[..., None] - 1e-5 * max(1.0, float(sq.max()))).all()      # every neighbour within fp32 noise of the true top-k
        assert (idx == torch.arange(M5)[None, :, None]).any(-1).all()               # self is always a neighbour
        assert exact > 0.98


def test_c5_attention_at_16384_keys_against_fp64():
    g = torch.Generator().manual_seed(53)
    q = torch.randn(1, 512, 64, generator=g) * 2.0
    k = torch.randn(1, M5, 64, generator=g) * 2.0
    v = torch.randn(1, M5, 64, generator=g)
    out = engine.op_attention(q.to(DEV), k.to(DEV), v.to(DEV), 0.125).cpu().double()
    ref = torch.softmax(q.double() @ k.double().transpose(1, 2) * 0.125, -1) @ v.double()
    err = (out - ref).abs().max().item()
    print(f"attention, 512 queries x {M5} keys: max |hip - fp64| {err:.2e}")
    assert err < 2e-5                     # 16384-term fp32 sums per output (the 4096-key cases of tests/test_gpu_ops.py hold 5e-6)


def test_c5_dulcet_at_16_scenes_of_16384_points():
    B, N = 16, M5
    cfg, md = build_conditioned("c4_dgcnn_attn_extra_affine", N, DEV)
    e0, e1, extra, eps = synth_pairs(B, N, N, 54)
    t0 = time.time()
    lp, bpd = _run(cfg, md, e0, e1, extra, eps)
    torch.cuda.synchronize()
    print(f"C5 forward 16 x {N} + {N} points, 115 layers: first call {time.time() - t0:.1f} s (with engine packing), bpd {float(bpd):.4f}, "
          f"peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    assert lp.shape == (B, N) and torch.isfinite(lp).all()
    lp2, _ = _run(cfg, md, e0, e1, extra, eps)
    assert torch.equal(lp, lp2)
    lps, _ = _run(cfg, md, e0, e1, extra, eps, slice(9, 11))
    assert torch.equal(lps, lp[9:11])
    # embedder on one full 16384-point scene against the oracle in fp64 (k-NN near-ties move single rows)
    emb = md["input_embedder"](e0[:1].to(DEV)).cpu().double()
    _, sd_e = state_dicts(md, torch.float64)
    t0 = time.time()
    with torch.no_grad():
        ref = O.context_embed(cfg, sd_e, e0[:1].double())
    d = (emb - ref).abs().amax(-1)[0]
    print(f"C5 embedder, {N} points ({time.time() - t0:.0f} s of host time): per-row max |hip - fp64| median {d.median():.2e} "
          f"q99 {d.quantile(0.99):.2e} max {d.max():.2e}")
    assert d.quantile(0.99).item() < 2e-5 and d.max().item() < 5e-3
    _rows_vs_oracle(f"C5 16 x {N} x 115 affine layers + extra context, scene 0 rows 0..255 against the full {N}-point context",
                    cfg, md, e0, e1, extra, eps, lp, 256)
