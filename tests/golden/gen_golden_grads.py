#!/usr/bin/env python3
"""Gradient fixtures for SURVEY.md §8f row N1 (backward of the hot path), produced by RUNNING THE REFERENCE.

Runs only in the build container (imports /root/reference through gen_golden.py's stubs).  For the existing end-to-end cases
(same synthesised weights, inputs and noise as e2e_<case>.npz) it runs the reference's `inner_loop` in fp64 with autograd on,
calls `loss.backward()` as train.py:112 does, and stores for every parameter a compact but binding summary of d loss / d param:

    sum(g), sum(|g|), g . r   (r = the deterministic N(0,1) vector synth.normal("gradproj/<name>"))   and the first 8 entries,

plus the gradient w.r.t. the target points' coordinates (extract_1, dense, small) and the global L2 norm that
`clip_grad_norm_` (train.py:115) computes.  A random projection pins the whole tensor: an error in any entry moves g . r.
Two modes: "eval" (BatchNorm running statistics — the function the forward fixtures pin) and "train" (`.train()`: BatchNorm
batch statistics in the DGCNN embedder, ActNorm already initialised, torch.utils.checkpoint recompute — numerically a no-op).
A third record, "init", is the first training forward with every ActNorm un-initialised (act_norm.py:27-39): the statistics each
layer sets from its input and the resulting log-probs (embedder in eval mode, as the HIP training path runs it this round).

    python tests/golden/gen_golden_grads.py            # writes tests/golden/grad_*.npz
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G      # noqa: E402  (imports the reference)
import synth                # noqa: E402

CASES = ["e2e_tiny_affine", "e2e_tiny_spline_relu", "e2e_tiny_cif", "e2e_tiny_global_extra", "e2e_tiny_random_permute", "e2e_spline_L2", "e2e_paconv_L2"]
HEAD = 8


def summarise(name, g):
    g = g.detach().double().reshape(-1)
    r = torch.from_numpy(synth.normal("gradproj/" + name, (g.numel(),), 0))
    return np.concatenate([[g.sum().item(), g.abs().sum().item(), (g * r).sum().item()],
                           np.pad(g[:HEAD].numpy(), (0, max(0, HEAD - g.numel())))])


def grad_case(case):
    z = np.load(os.path.join(HERE, case + ".npz"))
    cfg0 = json.loads(bytes(z["config_json"]).decode())
    meta = json.loads(bytes(z["meta_json"]).decode())
    arrays = {}
    info = {}
    for mode in ("eval", "train"):
        cfg = G.load_cfg(meta["cfg_name"], sample_size=meta["N"], **meta["over"])
        assert {k: cfg[k] for k in cfg0 if k in cfg} == {k: cfg0[k] for k in cfg0 if k in cfg}
        md = G.build(cfg, meta["seed"], torch.float64)
        if mode == "train":
            md["flow"].train()
            md["input_embedder"].train()
        e0 = torch.from_numpy(z["extract_0"]).double()
        e1 = torch.from_numpy(z["extract_1"]).double().requires_grad_(True)
        ex = torch.from_numpy(z["extra"]).double() if "extra" in z.files else None
        eps, i = [], 0
        while f"eps{i}" in z.files:
            eps.append(torch.from_numpy(z[f"eps{i}"]).double())
            i += 1
        G._EPS_QUEUE[:] = list(eps)
        loss, lp, bpd = G.mi.inner_loop((e0, e1, ex), md, cfg)
        assert not G._EPS_QUEUE
        if mode == "eval":
            assert np.allclose(lp.detach().numpy(), z["log_prob_f64"], rtol=1e-12, atol=1e-12)
        loss.backward()
        sq = 0.0
        names = []
        for part in ("flow", "input_embedder"):
            for n, p in md[part].named_parameters():
                if p.grad is None:
                    continue
                key = f"{part}/{n}"
                arrays[f"{mode}/{key}"] = summarise(key, p.grad)
                names.append(key)
                sq += float((p.grad.double() ** 2).sum())
        arrays[f"{mode}/d_extract_1"] = e1.grad.numpy()
        arrays[f"{mode}/loss"] = np.float64(loss.item())
        arrays[f"{mode}/grad_norm"] = np.float64(sq ** 0.5)
        info[mode] = names
        print(f"[{case}] {mode}: loss {loss.item():.6f}  |grad| {sq ** 0.5:.4e}  {len(names)} parameter tensors")
    # ---- data-dependent ActNorm initialisation (act_norm.py:27-39): first training forward with `initialized` == 0
    cfg = G.load_cfg(meta["cfg_name"], sample_size=meta["N"], **meta["over"])
    md = G.build(cfg, meta["seed"], torch.float64)
    md["flow"].train()
    md["input_embedder"].eval()                     # the HIP training path runs the embedder frozen in eval mode this round
    n_init = 0
    for m in md["flow"].modules():
        if hasattr(m, "initialized"):
            m.initialized.zero_()
            n_init += 1
    if n_init:
        G._EPS_QUEUE[:] = list(eps)
        with torch.no_grad():
            loss, lp, bpd = G.mi.inner_loop((e0, e1.detach(), ex), md, cfg)
        assert not G._EPS_QUEUE
        arrays["init/loss"] = np.float64(loss.item())
        arrays["init/log_prob"] = lp.numpy()
        for n, m in md["flow"].named_modules():
            if hasattr(m, "initialized"):
                assert float(m.initialized) == 1.0
                arrays[f"init/{n}.shift"] = m.shift.detach().numpy()
                arrays[f"init/{n}.log_scale"] = m.log_scale.detach().numpy()
        print(f"[{case}] data-dependent init of {n_init} ActNorm layers: loss {loss.item():.6f}")
    arrays["names_json"] = np.frombuffer(json.dumps(info).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "grad_" + case[len("e2e_"):] + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {os.path.basename(path)}  {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    G.patch_pointops()          # PAConv's six CUDA kernels -> the oracle's CPU restatements (as for the forward fixtures, gen_golden.py)
    for c in CASES:
        grad_case(c)
