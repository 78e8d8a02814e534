"""Where does the gradient noise of the pre-attention MLP weights on the paconv_L2 fixture come from?  The same step on the split-fp16
kernels, on the fp32-input kernels, and with only the attention backward / only the weight gradient on fp32 inputs: worst three errors
(relative to each tensor's L1 norm) against the reference's fp64 gradients.  Run from the repo root on a GPU box."""
import sys, os, json, re
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_R, "tests")); sys.path.insert(0, os.path.join(_R, "tests", "golden")); sys.path.insert(0, _R)
import numpy as np, torch
import test_gpu_train as TG
from conftest import Fixture, GOLDEN
import synth
from flowcompare_amd import engine
lib = engine.lib()
fx = Fixture("e2e_paconv_L2")
z = np.load(os.path.join(GOLDEN, "grad_paconv_L2.npz"))
cfg, md = TG._build(fx)
names = [n for n in json.loads(bytes(z["names_json"]).decode())["eval"] if n.startswith("flow/")]
gn = float(z["eval/grad_norm"])
def run(label, fp16, knobs=()):
    for k, v in knobs: lib.fc_debug_set(k, v)
    TG._train_step(fx, cfg, md, fp16=fp16)
    for k, v in knobs: lib.fc_debug_set(k, 1 if k != 11 else 1)
    params = dict(md["flow"].named_parameters())
    errs = []
    for key in names:
        g = params[key.split("/",1)[1]].grad.double().cpu().reshape(-1)
        r = torch.from_numpy(synth.normal("gradproj/" + key, (g.numel(),), 0))
        got = np.array([g.sum().item(), g.abs().sum().item(), (g * r).sum().item()])
        want = z["eval/" + key]
        errs.append((np.abs(got - want[:3]).max() / max(want[1], 1e-6 * gn), key))
    errs.sort(reverse=True)
    print(label, [(f"{e:.1e}", k.split("flow/")[1][:60]) for e, k in errs[:3]])
run("split-fp16 (default)        ", True)
run("fp32-input everything        ", False)
run("split-fp16, attn bwd fp32    ", True, [(12, 0)])
run("split-fp16, wgrad fp32       ", True, [(11, 0)])
