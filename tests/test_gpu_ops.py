"""Unit-level parity of the HIP kernels, called through the C ABI (fc_op_*), against fp64 math on the host."""
import os

import ctypes

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from flowcompare_amd import engine
from oracle import flow_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize("rows,N,K,act,res", [
    (1000, 512, 512, "gelu", True), (300, 256, 150, "gelu", False), (257, 300, 512, "none", False), (64, 64, 256, "none", False),
    (513, 3750, 512, "none", False), (100, 128, 12, "lrelu", False), (37, 24, 215, "relu", True), (129, 588, 96, "elu", False),
    (2048, 320, 320, "none", False), (1, 1, 1, "gelu", False),
])
def test_linear_matches_fp64(rows, N, K, act, res):
    """fp32 MFMA GEMM + fused epilogue vs fp64 F.linear; tolerance = fp32 fmaf-chain error bound ~ K * 2^-23 * sum|a b|."""
    x, W, b = _rand(rows, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3, scale=0.1)
    r = _rand(rows, N, seed=4) if res else None
    y = engine.op_linear(x.to(DEV), W.to(DEV), b.to(DEV), None if r is None else r.to(DEV), act).cpu().double()
    ref = torch.nn.functional.linear(x.double(), W.double(), b.double())
    if res:
        ref = ref + r.double()
    ref = {"none": lambda t: t, "gelu": torch.nn.functional.gelu, "relu": torch.relu, "elu": torch.nn.functional.elu,
           "lrelu": lambda t: torch.nn.functional.leaky_relu(t, 0.2)}[act](ref)
    assert (y - ref).abs().max().item() < 2e-6 * max(1.0, K ** 0.5)


def _mlp_state(k_in, n_mid, seed, colvec=False):
    g = torch.Generator().manual_seed(seed)
    u = lambda *sh, sc=1.0: (torch.rand(*sh, generator=g) * 2 - 1) * sc
    sd = {"net.in_layer.weight": u(512, k_in, sc=(3.0 / k_in) ** 0.5), "net.in_layer.bias": u(512, sc=0.2),
          "net.out_layer.weight": torch.zeros(1, 512)}
    for i in range(n_mid):
        sd[f"net.layers.{i}.weight"] = u(512, 512, sc=(3.0 / 512) ** 0.5 * 1.5)
        sd[f"net.layers.{i}.bias"] = u(512, sc=0.2)
    if colvec:
        sd["net.colvec"] = u(512, sc=0.1)
    return sd


def _mlp_ref(x, sd, n_mid, rowscal=None):
    """models/nets.py:19-30 in fp64: act(in_layer); even hidden layer keeps its input, odd adds it back before the activation."""
    F = torch.nn.functional
    h = F.linear(x.double(), sd["net.in_layer.weight"].double(), sd["net.in_layer.bias"].double())
    if rowscal is not None:
        h = h + rowscal.double()[:, None] * sd["net.colvec"].double()[None]
    h = F.gelu(h)
    keep = None
    for i in range(n_mid):
        z = F.linear(h, sd[f"net.layers.{i}.weight"].double(), sd[f"net.layers.{i}.bias"].double())
        if i % 2 == 0:
            keep = h
            h = F.gelu(z)
        else:
            h = F.gelu(keep + z)
    return h


@pytest.mark.parametrize("rows,k0,k1,n_mid,extra", [(1000, 150, 64, 2, False), (515, 150, 124, 5, False), (300, 150, 64, 2, True),
                                                     (4096, 150, 64, 2, False), (129, 150, 124, 4, False), (64, 256, 200, 3, False)])
def test_row_resident_mlp_chain_matches_fp64_and_the_per_layer_launches(rows, k0, k1, n_mid, extra):
    """csrc/mlprows.hip (in_layer + hidden layers of a 512-wide coupling net in one launch, activations resident in registers) against
    fp64 and against the limb-chained per-layer GEMM launches it replaces: same limb products in the same k order, residual added behind
    the k loop in both, same GELU and limb split -- bit-identical, and at the split-fp16 GEMM's distance from fp64."""
    sd = _mlp_state(k0 + k1, n_mid, seed=rows + n_mid, colvec=extra)
    x0, x1 = _rand(rows, k0, seed=11, scale=2.0), _rand(rows, k1, seed=12, scale=1.5)
    rs = (_rand(rows, seed=13) * 7 + 7.5) if extra else None
    ref = _mlp_ref(torch.cat((x0, x1), 1), sd, n_mid, rs)
    args = (x0.to(DEV), x1.to(DEV), sd, None if rs is None else rs.to(DEV))
    y_rows = engine.op_mlp_hidden(*args, use_rows=True).cpu().double()
    y_gemm = engine.op_mlp_hidden(*args, use_rows=False).cpu().double()
    scale = ref.abs().max().item()
    e_rows, e_gemm, e_ab = (y_rows - ref).abs().max().item(), (y_gemm - ref).abs().max().item(), (y_rows - y_gemm).abs().max().item()
    print(f"rows {rows} K {k0}+{k1} hidden layers {n_mid}: |chain - fp64| {e_rows:.2e}  |per-layer - fp64| {e_gemm:.2e}  |chain - per-layer| {e_ab:.2e}  (max |h| {scale:.2f})")
    assert e_gemm < 2e-5 * max(1.0, scale) and e_rows < 2e-5 * max(1.0, scale)
    # the engine picks between the two by the row count: they must agree bit for bit, or a scene's log-probs would depend on its batch
    assert torch.equal(y_rows, y_gemm)
    y2 = engine.op_mlp_hidden(*args, use_rows=True).cpu().double()
    assert torch.equal(y2, y_rows), "the row-resident chain is not deterministic"


@pytest.mark.parametrize("rows,k0,k1,n_mid,extra", [(1000, 150, 64, 2, False), (300, 150, 64, 4, True), (5000, 150, 64, 2, True), (256, 512, 0, 1, False)])
def test_wide_linear_layers_match_fp64_and_the_per_layer_path(rows, k0, k1, n_mid, extra):
    """Hidden layers of a 512-wide coupling net on the 256 x 256 one-accumulator kernel (spline_wide.hip EPI 1: k32 MFMAs, one fp32 accumulator,
    bias in the accumulator start, residual read from the previous image, exact-erf GELU, output as a limb image) against fp64 and against
    the per-layer 128 x 128 launches (another arithmetic: agreement within fp32 noise, not bit for bit).  5000 rows = 20 row tiles x 2 column
    tiles: several tiles per persistent workgroup order; the rows of a smaller call must reproduce bit for bit."""
    sd = _mlp_state(k0 + k1, n_mid, seed=9, colvec=extra)
    x0, x1 = _rand(rows, k0, seed=11, scale=2.0), (_rand(rows, k1, seed=12) if k1 else None)
    rs = (_rand(rows, seed=13) * 7 + 7.5) if extra else None
    ref = _mlp_ref(x0 if x1 is None else torch.cat((x0, x1), 1), sd, n_mid, rs)
    args = (x0.to(DEV), None if x1 is None else x1.to(DEV), sd, None if rs is None else rs.to(DEV))
    y_wide = engine.op_mlp_hidden(*args, use_rows="wide").cpu().double()
    y_gemm = engine.op_mlp_hidden(*args, use_rows=False).cpu().double()
    scale = ref.abs().max().item()
    e_wide, e_gemm = (y_wide - ref).abs().max().item(), (y_gemm - ref).abs().max().item()
    print(f"rows {rows} K {k0}+{k1} hidden layers {n_mid}: |wide - fp64| {e_wide:.2e} mean {(y_wide - ref).abs().mean().item():.2e}  |per-layer - fp64| {e_gemm:.2e} mean {(y_gemm - ref).abs().mean().item():.2e}  (max |h| {scale:.2f})")
    assert e_wide < 2e-5 * max(1.0, scale)
    assert (y_wide - ref).abs().mean().item() < 1.5 * (y_gemm - ref).abs().mean().item() + 1e-8
    part = engine.op_mlp_hidden(x0[:200].to(DEV), None if x1 is None else x1[:200].to(DEV), sd, None if rs is None else rs[:200].to(DEV), use_rows="wide").cpu().double()
    assert torch.equal(part, y_wide[:200]), "a row's result depends on the call it sits in (wide Linear kernel)"


def test_row_resident_mlp_chain_rows_are_independent():
    """Each wave of the chain kernel owns 32 rows for the whole chain: a row's result must not depend on which band / workgroup it sits in."""
    sd = _mlp_state(214, 2, seed=5)
    x0, x1 = _rand(700, 150, seed=21, scale=2.0), _rand(700, 64, seed=22)
    full = engine.op_mlp_hidden(x0.to(DEV), x1.to(DEV), sd).cpu()
    perm = torch.randperm(700, generator=torch.Generator().manual_seed(3))
    shuf = engine.op_mlp_hidden(x0[perm].to(DEV), x1[perm].to(DEV), sd).cpu()
    assert torch.equal(shuf, full[perm])
    part = engine.op_mlp_hidden(x0[37:300].to(DEV), x1[37:300].to(DEV), sd).cpu()
    assert torch.equal(part, full[37:300])


def test_linear_is_linear_and_exact_on_integers():
    """Integer-valued operands make every partial sum exact in fp32: the MFMA path must be bit-exact, which
    catches any fragment-layout / k-mapping mistake (asymmetric W so a transposed store cannot pass)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-8, 9, (384, 200), generator=g).float()
    W = torch.randint(-8, 9, (330, 200), generator=g).float()
    y = engine.op_linear(x.to(DEV), W.to(DEV)).cpu()
    assert torch.equal(y, x @ W.t())


@pytest.mark.parametrize("wamp", [0.04, 0.4, 0.004])
def test_one_accumulator_limb_form_is_at_least_as_accurate_as_an_fp32_fmaf_chain(wamp):
    """The product the round-4 fused spline layer is built on (spline_wide.hip), by itself: operands as hi + lo fp16 limbs with the low limb
    UNSCALED, pre-scaled by exact powers of two (activations by 16, weights so that max |w| lands in [2^14, 2^15)), the three limb products of
    a k32 step into ONE fp32 accumulator on v_mfma_f32_16x16x32_f16.  On the three weight scales of profiles/micro/one_acc_probe.hip, with
    GELU-like activations (many small values: the case an unscaled low limb is weakest at), K = 512, N = 3840: maximum and mean error against
    fp64 must not exceed those of a sequential fp32 fmaf chain (what the fp32-input MFMA computes, bit for bit) -- measured 0.6-0.7 x."""
    L = engine.lib()
    rows, N, K = 1024, 3840, 512
    g = torch.Generator().manual_seed(7)
    a = torch.randn(rows, K, generator=g)
    x = torch.where(a > 0, a, 0.05 * a)
    W = (torch.rand(N, K, generator=g) * 2 - 1) * wamp
    b = (torch.rand(N, generator=g) * 2 - 1) * 0.1
    ref = x.double() @ W.double().t() + b.double()
    # the fp32 fmaf chain, k ascending, on a sample of columns (a python loop over k on [rows, 96] panels)
    cols = torch.arange(5, N, 40)
    acc = b[cols].repeat(rows, 1).clone()
    Wc = W[cols]
    for k in range(K):
        acc = torch.addcmul(acc.double(), x[:, k:k + 1].double(), Wc[:, k][None].double()).float()     # one rounding per product-sum: fmaf
    e_chain = (acc.double() - ref[:, cols]).abs()
    out = torch.empty(rows, N, dtype=torch.float32, device=DEV)
    xd, Wd, bd = x.to(DEV), W.to(DEV), b.to(DEV)
    L.fc_debug_one_acc_gemm_f32.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_float, ctypes.c_void_p] + [ctypes.c_int32] * 3 + [ctypes.c_void_p]
    with torch.cuda.device(DEV):
        engine._check(L.fc_debug_one_acc_gemm_f32(engine._ptr(xd), engine._ptr(Wd), engine._ptr(bd), float(W.abs().max()), engine._ptr(out), rows, N, K, engine._stream()))
    e_one = (out.cpu().double() - ref).abs()
    print(f"weights U(-{wamp}, {wamp}), max |C| {ref.abs().max():.3f}: one accumulator max {e_one.max():.2e} mean {e_one.mean():.2e} (sampled columns: max {e_one[:, cols].max():.2e} mean {e_one[:, cols].mean():.2e})"
          f" | fp32 fmaf chain max {e_chain.max():.2e} mean {e_chain.mean():.2e}")
    assert e_one[:, cols].max().item() <= e_chain.max().item() and e_one[:, cols].mean().item() <= e_chain.mean().item()
    assert e_one.max().item() <= 1.5 * e_chain.max().item()          # (all 3840 columns against the sampled chain's worst)


def test_split_variants_match_fp64_like_fp32():
    """The split GEMM main loops -- 2 fp16 limbs / 3 MFMAs per product block (variant 5, the default) and 3 bf16 limbs / 6 MFMAs
    (variant 3, its unbounded-range fallback) -- must be as accurate as the fp32-input MFMA loop (variant 2)."""
    lib = engine.lib()
    try:
        for rows, N, K, act, xs in ((1000, 512, 512, "gelu", 3.0), (513, 3750, 512, "none", 3.0), (300, 256, 150, "none", 3.0),
                                    (300, 64, 256, "none", 3.0), (600, 512, 512, "none", 1e-3), (600, 512, 512, "none", 300.0)):
            x, W, b = _rand(rows, K, seed=1, scale=xs), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3, scale=0.1 * min(xs, 1.0))
            ref = torch.nn.functional.linear(x.double(), W.double(), b.double())
            if act == "gelu":
                ref = torch.nn.functional.gelu(ref)
            errs = {}
            for var in (2, 3, 5):
                lib.fc_debug_set(0, var)
                y = engine.op_linear(x.to(DEV), W.to(DEV), b.to(DEV), None, act).cpu().double()
                errs[var] = (y - ref).abs().max().item()
            print(f"split {rows}x{N}x{K} |x|~{xs}: fp32-mfma err {errs[2]:.2e}  split-bf16 err {errs[3]:.2e}  split-fp16 err {errs[5]:.2e}")
            for var in (3, 5):
                assert errs[var] < 2e-6 * max(1.0, K ** 0.5) * max(xs / 3.0, 1e-3) and errs[var] < 4 * errs[2] + 1e-7 * xs
        g = torch.Generator().manual_seed(5)
        x = torch.randint(-8, 9, (384, 200), generator=g).float()
        W = torch.randint(-8, 9, (512, 200), generator=g).float()
        for var in (3, 5):
            lib.fc_debug_set(0, var)
            assert torch.equal(engine.op_linear(x.to(DEV), W.to(DEV)).cpu(), x @ W.t())      # small integers: every limb product exact
    finally:
        lib.fc_debug_set(0, 5)            # shipped default


def test_split_fp16_out_of_range_falls_back_to_bf16_limbs():
    """fp16 limbs cannot hold |x| >= 65504: the kernel raises its flag and the call is repeated with the bf16 limbs, so the
    result stays fp32-accurate (and finite); weights that do not fit never get an fp16 image at all."""
    lib = engine.lib()
    lib.fc_debug_fp16_fallbacks.restype = ctypes.c_int64
    x, W = _rand(300, 512, seed=1, scale=3.0), _rand(512, 512, seed=2, scale=512 ** -0.5)
    before = lib.fc_debug_fp16_fallbacks()
    y = engine.op_linear(x.to(DEV), W.to(DEV)).cpu().double()
    assert lib.fc_debug_fp16_fallbacks() == before            # in range: no repeat
    x[17, 300] = 1.0e5
    x[250, 3] = -7.0e4
    ref = x.double() @ W.double().t()
    y = engine.op_linear(x.to(DEV), W.to(DEV)).cpu().double()
    assert lib.fc_debug_fp16_fallbacks() == before + 1
    assert torch.isfinite(y).all() and (y - ref).abs().max().item() < 2e-2 and ((y - ref).abs() / (1 + ref.abs())).max().item() < 1e-5
    Wb = W.clone(); Wb[5, 5] = 1.0e5                          # weight out of range: bf16 limbs from the start, no repeat
    x2 = _rand(300, 512, seed=4, scale=3.0)
    before = lib.fc_debug_fp16_fallbacks()
    y = engine.op_linear(x2.to(DEV), Wb.to(DEV)).cpu().double()
    ref = x2.double() @ Wb.double().t()
    assert lib.fc_debug_fp16_fallbacks() == before and ((y - ref).abs() / (1 + ref.abs())).max().item() < 1e-5


@pytest.mark.parametrize("B,N,M,D", [(2, 128, 64, 64), (3, 100, 130, 64), (1, 20, 24, 32), (2, 257, 1000, 64), (1, 64, 4096, 64), (2, 40, 70, 128)])
def test_attention_matches_fp64(B, N, M, D):
    """Both attention kernels: split-fp16 (default for D <= 64) and the fp32-input MFMA kernel (D = 128, fallback)."""
    q, k, v = _rand(B, N, D, seed=1, scale=2.0), _rand(B, M, D, seed=2, scale=2.0), _rand(B, M, D, seed=3)
    scale = D ** -0.5
    w = torch.softmax(q.double() @ k.double().transpose(1, 2) * scale, -1)
    ref = w @ v.double()
    lib = engine.lib()
    try:
        for fp16 in (1, 0):
            lib.fc_debug_set(5, fp16)
            out = engine.op_attention(q.to(DEV), k.to(DEV), v.to(DEV), scale).cpu().double()
            assert (out - ref).abs().max().item() < 5e-6, f"attention kernel fp16={fp16}"
    finally:
        lib.fc_debug_set(5, 1)


def test_attention_key_order_and_out_of_range_fallback():
    """Uniform scores over v[j] = j + d/1000 give the mean key exactly only if every key is paired with its own probability
    (a wrong k-order in the P V product passes random tests with small error but not this); then an out-of-range key makes
    the split-fp16 kernel raise its flag and the call is repeated with the fp32-input kernel."""
    lib = engine.lib()
    lib.fc_debug_fp16_fallbacks.restype = ctypes.c_int64
    B, N, M, D = 2, 100, 130, 64
    v = torch.arange(M).float()[None, :, None].expand(B, M, D).contiguous() + torch.arange(D).float()[None, None, :] * 1e-3
    k = _rand(B, M, D, seed=2)
    before = lib.fc_debug_fp16_fallbacks()
    out = engine.op_attention(torch.zeros(B, N, D).to(DEV), k.to(DEV), v.to(DEV), 0.125).cpu().double()
    assert (out - v.double().mean(1, keepdim=True)).abs().max().item() < 2e-5
    assert lib.fc_debug_fp16_fallbacks() == before
    q = _rand(B, N, D, seed=1)
    k[1, 77, 5] = 9.0e4
    ref = torch.softmax(q.double() @ k.double().transpose(1, 2) * 0.125, -1) @ v.double()
    out = engine.op_attention(q.to(DEV), k.to(DEV), v.to(DEV), 0.125).cpu().double()
    assert lib.fc_debug_fp16_fallbacks() == before + 1
    assert (out - ref).abs().max().item() < 2e-4


def test_attention_spiked_scores_force_rescale():
    """One key dominates late in the sequence so the running max jumps in the LAST tile (online-softmax rescale branch)."""
    B, N, M, D = 1, 64, 512, 64
    q, k, v = _rand(B, N, D, seed=1), _rand(B, M, D, seed=2), _rand(B, M, D, seed=3)
    k[0, 500] = q[0, 7] * 40.0
    out = engine.op_attention(q.to(DEV), k.to(DEV), v.to(DEV), 1.0).cpu().double()
    ref = torch.softmax(q.double() @ k.double().transpose(1, 2), -1) @ v.double()
    assert (out - ref).abs().max().item() < 5e-6


@pytest.mark.parametrize("per_tile", [0.4, 1.5, 7.5])
def test_attention_lazy_reference_on_score_ramps(per_tile):
    """Round 4: the split-fp16 attention moves its softmax reference only when a tile's maximum exceeds it by more than 3 (log2 domain), so the
    probabilities it holds are bounded by 2^3 (csrc/attention.hip, models/perceiver.py:106-113).  Scores that RISE along the key axis by 0.4 / 1.5 /
    7.5 nats per 64-key tile move the reference every fifth tile / every second tile / every tile, with probabilities above 1 in between:
    against fp64, to the bound of the exact-maximum form."""
    B, N, M, D = 2, 200, 1024, 64
    q, k, v = _rand(B, N, D, seed=61), _rand(B, M, D, seed=62), _rand(B, M, D, seed=63, scale=2.0)
    q[..., 0] = 4.0                                                          # score = 0.125 q . k: column 0 of k carries the ramp
    k[..., 0] = torch.arange(M).float()[None, :] * (per_tile / 64.0 / (0.125 * 4.0))
    ref = torch.softmax((q.double() @ k.double().transpose(1, 2)) * 0.125, -1) @ v.double()
    lib = engine.lib()
    lib.fc_debug_fp16_fallbacks.restype = ctypes.c_int64
    before = lib.fc_debug_fp16_fallbacks()
    y = engine.op_attention(q.to(DEV), k.to(DEV), v.to(DEV), 0.125).cpu()
    assert lib.fc_debug_fp16_fallbacks() == before, "the ramp must stay inside fp16's range (this test is about the split-fp16 kernel)"
    err = (y.double() - ref).abs().max().item()
    smax = per_tile * M / 64                                                 # a score s is known to |s| 2^-24 in fp32: so is its probability, relatively
    print(f"ramp {per_tile} nats per tile: max |y - fp64| = {err:.2e}   (largest score {smax:.0f} nats)")
    assert err < 4e-6 + 2.5e-7 * smax


@pytest.fixture(params=[2, 0], ids=["mfma-kernel", "lane-per-candidate-kernel"])
def knn_kernel(request):
    """Both k-NN kernels (csrc/knn.hip): 2 = the matrix-core kernel forced at any size (the engine picks it where the launch fills the chip),
    0 = the lane-per-candidate kernel (small launches)."""
    engine.lib().fc_debug_set(24, request.param)
    yield request.param
    engine.lib().fc_debug_set(24, 1)


def test_knn_golden_and_ties(knn_kernel):
    z = np.load(os.path.join(GOLDEN, "op_knn.npz"))
    for tag in ("xyzrgb", "feat64"):
        x = torch.from_numpy(z[f"{tag}_x"])
        f = x.transpose(1, 2).contiguous()
        idx = engine.op_knn(f.to(DEV), 40).cpu().long().sort(-1)[0]
        ref = torch.from_numpy(z[f"{tag}_idx_f64"].astype(np.int64)).sort(-1)[0]
        same = (idx == ref).all(-1)
        margin = torch.from_numpy(z[f"{tag}_margin_f64"])
        assert (same | (margin < 1e-5)).all(), f"{tag}: {int((~same).sum())} rows differ beyond near-ties"
        assert same.float().mean() > 0.99


@pytest.mark.parametrize("B,M,C,k", [(2, 1024, 6, 40), (1, 700, 64, 40), (2, 300, 128, 40), (1, 64, 6, 64), (1, 50, 16, 1), (3, 257, 30, 40), (1, 129, 100, 7)])
def test_knn_random_vs_fp64(B, M, C, k, knn_kernel):
    f = _rand(B, M, C, seed=7)
    idx = engine.op_knn(f.to(DEV), k).cpu().long()
    assert idx.min() >= 0 and idx.max() < M
    assert (idx.sort(-1)[0][..., 1:] != idx.sort(-1)[0][..., :-1]).all(), "duplicate neighbours"
    fd = f.double()
    sq = (fd ** 2).sum(-1)
    pd = -sq[:, None, :] + 2 * fd @ fd.transpose(1, 2) - sq[:, :, None]
    srt = pd.sort(-1, descending=True)
    kth = srt.values[..., k - 1]
    got = torch.gather(pd, 2, idx)
    assert (got >= kth[..., None] - 1e-5).all()                 # every returned neighbour is within fp32 noise of the true top-k
    assert (idx == torch.arange(M)[None, :, None]).any(-1).all()  # self is always a neighbour


@pytest.mark.parametrize("B,M,C", [(2, 2048, 64), (1, 4096, 128), (2, 1500, 6), (1, 16384, 64)])
def test_knn_warm_start_returns_the_same_sets(B, M, C):
    """Round 4: the DGCNN levels 1-3 start their k-NN stream from the previous level's neighbour sets (csrc/knn.hip, models/pytorch_gcn.py:43-60:
    four searches of one cloud in successive feature spaces).  The warm sets only give the initial threshold, so the result must be the
    SAME SET as the cold search whatever they hold: the true neighbours in a nearby feature space (the real case), random indices (a useless
    bound), sets with repeated indices or the query in every slot (no bound at all), out-of-range indices."""
    L = engine.lib()
    k = 40
    f = _rand(B, M, C, seed=21)
    g = torch.Generator().manual_seed(22)
    near = f + 0.3 * torch.randn(B, M, C, generator=g)
    try:
        assert L.fc_debug_set(24, 2) == 0                          # the matrix-core kernel at any size
        cold = engine.op_knn(f.to(DEV), k).cpu().long().sort(-1)[0]
        warm_sets = {
            "neighbours of a perturbed cloud": engine.op_knn(near.to(DEV), k).cpu(),
            "random indices": torch.randint(0, M, (B, M, k), generator=g, dtype=torch.int32),
            "the query in every slot": torch.arange(M, dtype=torch.int32)[None, :, None].expand(B, M, k).contiguous(),
            "one index repeated": torch.randint(0, M, (B, M, 1), generator=g, dtype=torch.int32).expand(B, M, k).contiguous(),
            "out of range": torch.full((B, M, k), M + 5, dtype=torch.int32),
            "its own result": cold.int(),
        }
        for name, w in warm_sets.items():
            got = engine.op_knn(f.to(DEV), k, warm=w.to(DEV)).cpu().long().sort(-1)[0]
            assert torch.equal(got, cold), f"warm start from {name}: {(got != cold).any(-1).sum().item()} of {B * M} sets differ"
    finally:
        L.fc_debug_set(24, 1)


def test_spline_golden_forward_and_inverse():
    z = np.load(os.path.join(GOLDEN, "op_spline.npz"))
    x, w, h, d = (torch.from_numpy(z[k]).float() for k in ("x", "w", "h", "d"))
    params = torch.cat((w, h, d), -1)
    y, lad = engine.op_rqspline(x.to(DEV), params.to(DEV), 8)
    # inputs within one fp32 ulp of +-3 change side when cast to fp32: compare those against the reference's own fp32 run
    same_side = torch.from_numpy((np.abs(z["x"]) <= 3) == (np.abs(z["x"].astype(np.float32)) <= 3))
    assert (~same_side).sum() <= 2
    assert np.abs(y.cpu().numpy() - z["y"])[same_side.numpy()].max() < 2e-5
    assert np.abs(lad.cpu().numpy() - z["logabsdet"])[same_side.numpy()].max() < 2e-4
    assert np.abs(y.cpu().numpy() - z["y_f32"]).max() < 2e-5 and np.abs(lad.cpu().numpy() - z["logabsdet_f32"]).max() < 2e-4
    # inverse: the quadratic root is ill-conditioned where the spline is flat, so the fp32 inverse is judged by what the domain
    # guarantees: forward(inverse(y)) == y and logabsdet_inv == -logabsdet_fwd at the recovered point, plus a loose bound vs fp64
    yin = torch.from_numpy(z["y"]).float()
    xi, ladi = engine.op_rqspline(yin.to(DEV), params.to(DEV), 8, inverse=True)
    y2, lad2 = engine.op_rqspline(xi, params.to(DEV), 8)
    inside = (yin.abs() < 2.999)
    assert (y2.cpu() - yin).abs()[inside].max() < 5e-5
    assert (lad2.cpu() + ladi.cpu()).abs()[inside].max() < 5e-4
    assert np.abs(xi.cpu().numpy() - z["x_inv"]).max() < 5e-3
    xo, lo = O.rq_spline(yin, w, h, d, inverse=True)             # pinned oracle in fp32: same formula, same conditioning
    assert (xi.cpu() - xo).abs().max() < 2e-3
    # tails: identity with zero log-det, knots: +-3 inclusive
    out = (x.abs() > 3)
    assert torch.equal(y.cpu()[out], x[out]) and (lad.cpu()[out] == 0).all()


@pytest.mark.parametrize("B,n,m", [(2, 1024, 256), (1, 4096, 1024), (3, 300, 75), (1, 20, 5), (2, 64, 64), (1, 16384, 4096), (2, 9000, 150)])
def test_fps_matches_restatement(B, n, m):
    """Farthest point sampling vs the CPU restatement of the reference's CUDA kernel (same start index, same min-distance update,
    same arg-max tie rule); bit-exact integer output.  More than 8192 points per scene take the variant whose min-distance array
    lives in global memory (the reference kernel has no size limit, sampling_cuda_kernel.cu:170-209)."""
    from oracle import paconv_oracle as P
    xyz = _rand(B, n, 3, seed=11)
    idx = engine.op_fps(xyz.to(DEV), m).cpu().long()
    ref = P.furthest_sampling(xyz, m)
    assert torch.equal(idx, ref)


def test_fps_ties_on_a_grid():
    """A regular grid produces exact distance ties: the winner must follow the kernel's (k mod T, k) rule, not 'lowest index'."""
    from oracle import paconv_oracle as P
    g = torch.stack(torch.meshgrid(torch.arange(8.), torch.arange(8.), torch.arange(4.), indexing="ij"), -1).reshape(1, -1, 3)
    idx = engine.op_fps(g.to(DEV), 64).cpu().long()
    assert torch.equal(idx, P.furthest_sampling(g, 64))


@pytest.mark.parametrize("B,N,M,D", [(2, 1024, 1000, 64), (1, 4096, 4096, 64), (3, 300, 777, 64), (2, 1000, 1250, 64), (1, 256, 64, 32), (2, 700, 33, 32)])
@pytest.mark.parametrize("scale", [1.0, 40.0, 0.05])
def test_attention_one_accumulator_form_against_fp64(B, N, M, D, scale):
    """Round 4: the split-fp16 attention keeps ONE accumulator per output (csrc/attention.hip: q, k, v as hi + lo of x * 16 with lo unscaled, the
    probabilities as hi + lo of p * 2^14; models/perceiver.py:106-113): ragged query / key counts (the last key tile masked, the last query
    rows clamped), one key tile only, and operands of order 1, 40 and 0.05 (where most unscaled low limbs are subnormal) against fp64."""
    q, k, v = _rand(B, N, D, seed=51) * scale, _rand(B, M, D, seed=52) * scale, _rand(B, M, D, seed=53, scale=2.0) * scale
    sm = 0.125 / (scale * scale)                                            # keeps the scores at the magnitude of the unit case
    ref = torch.softmax((q.double() @ k.double().transpose(1, 2)) * sm, -1) @ v.double()
    y = engine.op_attention(q.to(DEV), k.to(DEV), v.to(DEV), sm).cpu()
    err = (y.double() - ref).abs().max().item() / scale              # (y is a weighted mean of v rows of order 2 * scale: the bound the two-accumulator kernel was held to)
    print(f"B {B} N {N} M {M} D {D} operand scale {scale}: max |y - fp64| / scale = {err:.2e}   (max |y| / scale {ref.abs().max().item() / scale:.2f})")
    assert err < 2e-6


def _pointops_clouds():
    """Clouds that exercise what random floats never do: an integer lattice (exact distance ties everywhere), duplicated points (ties at
    distance 0: which copy is the centre), a non-cubic lattice with a jitter that breaks SOME ties, and plain random points."""
    lat = torch.stack(torch.meshgrid(torch.arange(8.), torch.arange(8.), torch.arange(8.), indexing="ij"), -1).reshape(1, -1, 3)
    g = torch.Generator().manual_seed(5)
    rnd = torch.rand(1, 512, 3, generator=g)
    dup = rnd.clone()
    dup[:, 1::4] = dup[:, 0::4]                                   # every 4th point is a copy of its predecessor
    mixed = (lat * torch.tensor([1.0, 0.5, 0.25])).clone()
    mixed[:, ::7] += 1e-3 * torch.rand(1, mixed[:, ::7].shape[1], 3, generator=g)
    return {"lattice": lat, "duplicates": dup, "lattice+jitter": mixed, "random": rnd}


def _knn_hip(xyz, q, k):
    """fc_op_paconv_knn_f32 on [B, n, 3] / [B, m, 3] clouds (the C ABI takes rows of pitch 4) -> [B, m, k] local indices"""
    L = engine.lib()
    B, n, m = xyz.shape[0], xyz.shape[1], q.shape[1]
    x4 = torch.zeros(B * n, 4); x4[:, :3] = xyz.reshape(-1, 3)
    q4 = torch.zeros(B * m, 4); q4[:, :3] = q.reshape(-1, 3)
    x4, q4 = x4.to(DEV), q4.to(DEV)
    out = torch.full((B * m, k), -1, dtype=torch.int32, device=DEV)
    with torch.cuda.device(DEV):
        engine._check(L.fc_op_paconv_knn_f32(engine._ptr(x4), engine._ptr(q4), engine._ptr(out), B, n, m, k, engine._stream()))
    return out.cpu().long().reshape(B, m, k)


@pytest.mark.parametrize("cloud", ["lattice", "duplicates", "lattice+jitter", "random"])
@pytest.mark.parametrize("k", [32, 8])
def test_paconv_knn_ties_follow_the_reference_heap(cloud, k):
    """fc_op_paconv_knn_f32 against the literal restatement of knnquery_heap_cuda_kernel.cu:21-89 (oracle/pointops_oracle.c: the kernel's
    max-heap, strict `d2 < root` insertion, heap sort; distances as nvcc -O2 contracts them): identical SET and identical ORDER, also where
    distances tie -- the heap's order among equal distances is not an index order, and which entry tied at the k-th distance survives is
    the heap's choice.  Queries: every 4th point (the lattice's and the copies' own points: distance 0 ties included)."""
    from oracle import paconv_oracle as P
    xyz = _pointops_clouds()[cloud]
    xyz = torch.cat((xyz, xyz.flip(1)), 0)                       # B = 2: the second scene walks the candidates in the opposite order
    q = xyz[:, ::4].contiguous()
    got = _knn_hip(xyz, q, k)
    ref = P.knnquery_heap(k, xyz, q)
    bad = (got != ref).any(-1)
    assert not bad.any(), f"{cloud}, k = {k}: {int(bad.sum())} of {bad.numel()} queries differ from the reference heap, first: {got[bad][0].tolist()} vs {ref[bad][0].tolist()}"
    if cloud == "lattice":
        srt = torch.argsort(((q[:, :, None] - xyz[:, None]) ** 2).sum(-1), dim=-1, stable=True)[..., :k]
        assert (srt != ref).any(), "the lattice case no longer distinguishes the heap's order from a stable sort"


@pytest.mark.parametrize("n,m,k", [(20, 5, 32), (3, 3, 32), (1, 1, 8), (33, 9, 32)])
def test_paconv_knn_fewer_points_than_neighbours(n, m, k):
    """n < nsample: the reference's unfilled heap slots keep (1e10, index 0) and sort to the end (knnquery_heap_cuda_kernel.cu:69-72)."""
    from oracle import paconv_oracle as P
    xyz = _rand(2, n, 3, seed=31)
    q = xyz[:, :m].contiguous()
    assert torch.equal(_knn_hip(xyz, q, k), P.knnquery_heap(k, xyz, q))


def test_paconv_knn_at_16384_points():
    """The size of C5's first set-abstraction level: 16384 candidates, 4096 queries picked by farthest point sampling, 32 neighbours."""
    from oracle import paconv_oracle as P
    xyz = _rand(1, 16384, 3, seed=32)
    idx = engine.op_fps(xyz.to(DEV), 4096).cpu().long()
    assert torch.equal(idx, P.furthest_sampling(xyz, 4096))
    q = torch.gather(xyz, 1, idx[..., None].expand(-1, -1, 3))
    assert torch.equal(_knn_hip(xyz, q, 32), P.knnquery_heap(32, xyz, q))


def _three_nn_hip(unknown, known):
    L = engine.lib()
    B, nu, mk = unknown.shape[0], unknown.shape[1], known.shape[1]
    u4 = torch.zeros(B * nu, 4); u4[:, :3] = unknown.reshape(-1, 3)
    k4 = torch.zeros(B * mk, 4); k4[:, :3] = known.reshape(-1, 3)
    u4, k4 = u4.to(DEV), k4.to(DEV)
    idx = torch.full((B * nu, 3), -1, dtype=torch.int32, device=DEV)
    w = torch.full((B * nu, 3), float("nan"), dtype=torch.float32, device=DEV)
    with torch.cuda.device(DEV):
        engine._check(L.fc_train_three_nn_f32(engine._ptr(u4), engine._ptr(k4), B, nu, mk, engine._ptr(idx), engine._ptr(w), engine._stream()))
    idx = idx.cpu().long().reshape(B, nu, 3) - (torch.arange(B) * mk)[:, None, None]      # the entry point returns rows of the whole [B * mk] panel
    return idx, w.cpu().reshape(B, nu, 3)


def _three_nn_ref(unknown, known):
    """nearestneighbor (interpolation_cuda_kernel.cu:134-176) + PointNet2FPModule's weights (pointnet2_paconv_modules.py:224-228)"""
    from oracle import paconv_oracle as P
    dist, idx = P.nearest_neighbor3(unknown, known)
    rec = 1.0 / (dist + 1e-8)
    return idx, rec / rec.sum(dim=2, keepdim=True)


@pytest.mark.parametrize("cloud", ["lattice", "duplicates", "lattice+jitter", "random"])
def test_three_nn_matches_the_reference_kernel(cloud):
    """fc_train_three_nn_f32 (and the forward's three_nn_interp_kernel, same selection code) against the literal restatement of
    nearestneighbor_cuda_kernel_fast: three running minima with strict `<` updates (the earliest index wins a tie), distances contracted as
    nvcc -O2 does; inverse-distance weights of PointNet2FPModule.  Known points: every 4th point of the cloud (so unknown points coincide
    with known ones: distance 0, weight 1e8 / sum)."""
    xyz = _pointops_clouds()[cloud]
    xyz = torch.cat((xyz, xyz.flip(1)), 0)
    known = xyz[:, ::4].contiguous()
    idx, w = _three_nn_hip(xyz, known)
    ridx, rw = _three_nn_ref(xyz, known)
    assert torch.equal(idx, ridx), f"{cloud}: {int((idx != ridx).any(-1).sum())} of {idx.shape[0] * idx.shape[1]} points pick other neighbours"
    assert (w - rw).abs().max().item() < 2e-6


@pytest.mark.parametrize("nu,mk", [(5, 1), (20, 2), (64, 3), (16384, 4096)])
def test_three_nn_with_fewer_than_three_known_points_and_at_size(nu, mk):
    """m < 3: the kernel's unfilled minima stay 1e40 -> +inf as float, index 0, weight 0 (interpolation_cuda_kernel.cu:150-151, 169-175); and
    the size of C5's last feature-propagation level."""
    unknown, known = _rand(2 if nu < 1000 else 1, nu, 3, seed=41), None
    known = unknown[:, :mk].contiguous() if mk < 100 else _rand(1, mk, 3, seed=42)
    idx, w = _three_nn_hip(unknown, known)
    ridx, rw = _three_nn_ref(unknown, known)
    assert torch.equal(idx, ridx)
    assert torch.isfinite(w).all() and (w - rw).abs().max().item() < 2e-6


@pytest.mark.parametrize("cloud", ["lattice", "duplicates", "lattice+jitter"])
def test_fps_ties_on_degenerate_clouds(cloud):
    """Farthest point sampling on clouds with exact ties (and, with duplicated points, minimum distances of exactly 0): the reference block's
    (k mod T, k) tie rule, simulated thread by thread in oracle/pointops_oracle.c."""
    from oracle import paconv_oracle as P
    xyz = _pointops_clouds()[cloud]
    xyz = torch.cat((xyz, xyz.flip(1)), 0)
    m = xyz.shape[1] // 4
    assert torch.equal(engine.op_fps(xyz.to(DEV), m).cpu().long(), P.furthest_sampling(xyz, m))


def test_paconv_embedder_matches_reference_golden():
    """PAConv U-Net (320 -> 80 -> 20 -> 5 -> 1 points: n < nsample heap tails, single known point in FP) vs the golden produced
    by the reference's own Python with the pointops kernels substituted by their CPU restatements."""
    import flowcompare_amd as fa
    from conftest import Fixture
    fx = Fixture("emb_paconv")
    cfg = dict(fx.cfg)
    md = fa.initialize_flow(cfg, device=DEV, mode="test")
    _, sd_emb = fx.state_dicts()
    md["input_embedder"].load_state_dict(sd_emb)
    emb = md["input_embedder"](fx.t("pts").to(DEV)).cpu().double().numpy()
    d = np.abs(emb - fx.a["emb_f64"])
    print(f"paconv embedder: max {d.max():.2e} mean {d.mean():.2e}")
    assert d.max() < 2e-5


def test_knn_with_non_finite_features_returns_valid_indices(knn_kernel):
    """A pass whose fp16 range flag is already raised keeps running until the caller discards it: k-NN on NaN / inf features must still
    hand valid row indices to the gathers downstream (found by tests/test_gpu_train.py::test_training_step_rolls_back_...)."""
    f = _rand(2, 300, 8, seed=9)
    f[0] = float("nan")
    f[1, 7] = float("inf")
    idx = engine.op_knn(f.to(DEV), 40).cpu().long()
    assert idx.min() >= 0 and idx.max() < 300


def test_profile_stride_brackets_every_nth_matching_launch():
    """fc_profile_stride (ABI v6): of the launches that pass the filter every n-th one is bracketed; the report counts the bracketed ones."""
    x, W = torch.randn(256, 64).to(DEV), torch.randn(64, 64).to(DEV)
    counts = {}
    for stride in (1, 4):
        engine.profile_filter("gemm_f32_kernel"); engine.profile_stride(stride); engine.profile_reset(); engine.profile_enable(True)
        for _ in range(10):
            engine.op_linear(x, W)
        torch.cuda.synchronize()
        engine.profile_enable(False)
        rep = engine.profile_report()
        counts[stride] = sum(p["launches"] for p in rep if "gemm_f32_kernel" in p["kernel"])
        assert all(p["ms"] > 0 for p in rep)
    engine.profile_filter(None); engine.profile_stride(1); engine.profile_reset()
    assert counts[1] == 10 and counts[4] == 3, counts          # launches 0, 4, 8

