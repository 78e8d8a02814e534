#!/usr/bin/env python3
"""Golden vectors for the steps either side of the path, produced by RUNNING THE REFERENCE's own
functions in this container:

  * `utils.co_unit_sphere` (utils.py:271-280) is imported from /root/reference (third-party imports stubbed as in gen_golden.py);
  * `clamp_infs` / `log_prob_to_change` live in test_flow.py, whose module imports (dataloaders -> numpy.lib.function_base)
    fail under numpy 2.x with an ordinary ModuleNotFoundError; the two function definitions are therefore taken from the
    file's syntax tree at generation time and executed with the reference's own `is_valid` -- nothing of the reference is
    copied into this repository, only inputs and outputs are stored.

Usage: python tests/golden/gen_golden_staging.py   (writes tests/golden/stage_change.npz, stage_sphere.npz)
"""
import ast
import contextlib
import io
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import synth

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_stub("laspy")
sys.modules["laspy"].file = _stub("laspy.file", File=object)
for n in ("open3d", "dash_core_components", "dash_html_components", "pykeops", "pointops_cuda"):
    _stub(n)
_stub("pykeops.torch", Vi=None, Vj=None)
sys.path.insert(0, REF)
import utils as ref_utils  # noqa: E402

tree = ast.parse(open(os.path.join(REF, "test_flow.py")).read())
ns = {"torch": torch, "is_valid": ref_utils.is_valid}
for node in tree.body:
    if isinstance(node, ast.FunctionDef) and node.name in ("clamp_infs", "log_prob_to_change"):
        exec(compile(ast.Module(body=[node], type_ignores=[]), "test_flow.py", "exec"), ns)
ref_change = ns["log_prob_to_change"]


def main():
    out = {}
    cases = []
    for ci, (B, N, N0, multiple, cutoff, n_inf) in enumerate([(3, 500, 500, 5.4, None, 0), (2, 1000, 700, 2.0, None, 3), (4, 257, 257, 1.0, -2.5, 2),
                                                              (1, 4096, 4096, 5.4, None, 0), (2, 64, 64, 0.5, None, 1)]):
        lp10 = (synth.normal(f"chg{ci}/lp10", (B, N), seed=1) * 3.0 - 4.0).astype(np.float32)      # fp32 inputs, as the path produces
        lp00 = (synth.normal(f"chg{ci}/lp00", (B, N0), seed=1) * 1.0 - 2.0).astype(np.float32)
        # a changed region: a block of much less likely points
        lp10[:, : N // 10] -= 15.0
        for k in range(n_inf):
            lp10[k % B, 7 + 11 * k] = -np.inf if k % 2 == 0 else np.inf
            lp00[(k + 1) % B, 3 + 5 * k] = -np.inf
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            a, b = torch.from_numpy(lp10).to(dt), torch.from_numpy(lp00).to(dt)
            with contextlib.redirect_stdout(io.StringIO()):
                r = ref_change(a, b, multiple, hard_cutoff=cutoff)
            out[f"c{ci}_out_{tag}"] = r.numpy()
            out[f"c{ci}_lp10_after_{tag}"] = a.numpy()          # the reference clamps its arguments in place
            out[f"c{ci}_lp00_after_{tag}"] = b.numpy()
        out[f"c{ci}_lp10"], out[f"c{ci}_lp00"] = lp10, lp00
        cases.append((B, N, N0, multiple, -1e30 if cutoff is None else cutoff, 0 if cutoff is None else 1))
    out["cases"] = np.array(cases, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "stage_change.npz"), **out)

    sph = {}
    for ci, (n0, n1) in enumerate([(4096, 2000), (100, 100), (1, 5)]):
        p0 = synth.uniform(f"sph{ci}/p0", (n0, 6), -40.0, 55.0, seed=2).astype(np.float32)
        p1 = synth.uniform(f"sph{ci}/p1", (n1, 6), -38.0, 50.0, seed=2).astype(np.float32)
        p0[:, 3:] = synth.uniform(f"sph{ci}/c0", (n0, 3), 0.0, 1.0, seed=2)
        p1[:, 3:] = synth.uniform(f"sph{ci}/c1", (n1, 3), 0.0, 1.0, seed=2)
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            a, b, inv = ref_utils.co_unit_sphere(torch.from_numpy(p0).to(dt), torch.from_numpy(p1).to(dt), return_inverse=True)
            sph[f"s{ci}_o0_{tag}"], sph[f"s{ci}_o1_{tag}"] = a.numpy(), b.numpy()
            sph[f"s{ci}_far_{tag}"], sph[f"s{ci}_mean_{tag}"] = inv["furthest_distance"].numpy(), inv["mean"].numpy()
        sph[f"s{ci}_p0"], sph[f"s{ci}_p1"] = p0, p1
    sph["n_cases"] = np.array(3)
    np.savez_compressed(os.path.join(HERE, "stage_sphere.npz"), **sph)
    print("wrote stage_change.npz, stage_sphere.npz")


if __name__ == "__main__":
    main()
