"""Pins the oracle's BACKWARD (autograd through oracle/flow_oracle.py) to gradients the reference itself produced with
loss.backward() (train.py:112; tests/golden/gen_golden_grads.py): every parameter's gradient through a random projection,
its sum, its L1 norm and its first entries; the gradient w.r.t. the target points; the global norm clip_grad_norm_ sees.
Eval mode (BatchNorm running statistics) and train mode (batch statistics).  CPU only.  This is the checker for SURVEY.md
§8f row N1 (backward of the hot path)."""
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, Fixture
from oracle import flow_oracle as O
import synth

CASES = ["tiny_affine", "tiny_spline_relu", "tiny_cif", "tiny_global_extra", "tiny_random_permute", "spline_L2", "paconv_L2"]
HEAD = 8


def oracle_grads(fx, mode):
    cfg = fx.derived_cfg()
    sd_f, sd_e = fx.state_dicts(torch.float64)
    for sd in (sd_f, sd_e):
        for k, v in sd.items():
            if v.is_floating_point():
                v.requires_grad_(True)
    e0, e1, ex = fx.t("extract_0", torch.float64), fx.t("extract_1", torch.float64).requires_grad_(True), fx.t("extra", torch.float64)
    if mode == "train":
        with O.train_mode():
            loss, lp, _ = O.inner_loop(cfg, sd_f, sd_e, (e0, e1, ex), fx.eps(torch.float64))
    else:
        loss, lp, _ = O.inner_loop(cfg, sd_f, sd_e, (e0, e1, ex), fx.eps(torch.float64))
    loss.backward()
    return loss, {"flow": sd_f, "input_embedder": sd_e}, e1.grad


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("case", CASES)
def test_oracle_autograd_matches_reference_backward(case, mode):
    fx = Fixture("e2e_" + case)
    z = np.load(os.path.join(GOLDEN, "grad_" + case + ".npz"))
    names = json.loads(bytes(z["names_json"]).decode())[mode]
    loss, sds, d_e1 = oracle_grads(fx, mode)
    assert abs(loss.item() - float(z[f"{mode}/loss"])) < 1e-9 * max(1.0, abs(loss.item()))
    np.testing.assert_allclose(d_e1.numpy(), z[f"{mode}/d_extract_1"], rtol=1e-7, atol=1e-10)
    sq, worst = 0.0, 0.0
    for key in names:
        part, n = key.split("/", 1)
        # CIFblock: augmenter and slicer share ONE ConditionalNormal (cif_helper); the oracle reads the augmenter.* names.
        # DGCNN: bn{i} and conv{i}.1 are the same BatchNorm module (pytorch_gcn.py:63-78); named_parameters() lists it once, as bn{i}
        alias = re.sub(r"^bn(\d)\.", r"conv\1.1.", n).replace(".slicer.noise_dist.", ".augmenter.noise_dist.")
        t = sds[part][alias]
        assert t.grad is not None, key
        g = t.grad.double().reshape(-1)
        want = z[f"{mode}/{key}"]
        r = torch.from_numpy(synth.normal("gradproj/" + key, (g.numel(),), 0))
        got = np.concatenate([[g.sum().item(), g.abs().sum().item(), (g * r).sum().item()], np.pad(g[:HEAD].numpy(), (0, max(0, HEAD - g.numel())))])
        scale = max(want[1], 1e-9 * float(z[f"{mode}/grad_norm"]))    # L1 norm of the reference gradient (floor: identically-zero gradients, e.g. q of a one-key softmax)
        err = np.abs(got - want).max() / scale
        worst = max(worst, err)
        assert err < 1e-7, (key, got[:3], want[:3])
        sq += float((g ** 2).sum())
    assert abs(sq ** 0.5 - float(z[f"{mode}/grad_norm"])) < 1e-8 * float(z[f"{mode}/grad_norm"])
    print(f"{case}/{mode}: {len(names)} parameter gradients, worst error / L1 norm {worst:.1e}")


@pytest.mark.parametrize("case", CASES)
def test_oracle_actnorm_data_init_matches_reference(case):
    """First training forward with un-initialised ActNorm layers (act_norm.py:27-39): the statistics every layer sets from its input
    and the resulting log-probs, against the reference's own run."""
    fx = Fixture("e2e_" + case)
    z = np.load(os.path.join(GOLDEN, "grad_" + case + ".npz"))
    if "init/loss" not in z.files:
        pytest.skip("no ActNorm layer in this configuration")
    cfg = fx.derived_cfg()
    sd_f, sd_e = fx.state_dicts(torch.float64)
    e0, e1, ex = fx.t("extract_0", torch.float64), fx.t("extract_1", torch.float64), fx.t("extra", torch.float64)
    with torch.no_grad(), O.actnorm_data_init():
        loss, lp, _ = O.inner_loop(cfg, sd_f, sd_e, (e0, e1, ex), fx.eps(torch.float64))
    np.testing.assert_allclose(lp.numpy(), z["init/log_prob"], rtol=1e-9, atol=1e-9)
    n = 0
    for key in z.files:
        if key.startswith("init/transforms"):
            np.testing.assert_allclose(sd_f[key[len("init/"):]].numpy(), z[key], rtol=1e-9, atol=1e-10)
            n += 1
    assert n >= 2
