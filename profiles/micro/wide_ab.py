"""Same-box A/B of the wide fused spline kernel (spline_wide.hip) under fc_debug_set settings: a short C2 stack (8 layers, 16 x 4096 points), the
in-library HIP-event profiler on, average launch duration of the fused spline kernel per setting, interleaved rounds.
    python profiles/micro/wide_ab.py "" "27=1" "27=2" "28=3" "13=4"
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import flowcompare_amd as fa  # noqa: E402
from flowcompare_amd import engine  # noqa: E402
from fullsize_util import build_conditioned, synth_pairs  # noqa: E402

DEV = "cuda:0"
B, N = 16, 4096
cfg, md = build_conditioned("c2_dgcnn_attn_spline", N, DEV, n_flow_layers=8)
e0, e1, _, eps = synth_pairs(B, N, N, 12)
batch = (e0.to(DEV), e1.to(DEV), None)
ep = [eps.to(DEV)]
lib = engine.lib()
specs = sys.argv[1:] or [""]
defaults = {13: 5, 27: 0, 28: -1, 2: 10, 14: 0}
res = {s: [] for s in specs}
for rnd in range(3):
    for spec in specs:
        sets = [tuple(int(x) for x in kv.split("=")) for kv in spec.split(",") if kv]
        ok = all(lib.fc_debug_set(k, v) == 0 for k, v in sets)
        if ok:
            for _ in range(2):
                fa.inner_loop(batch, md, cfg, eps=ep)
            torch.cuda.synchronize()
            engine.profile_reset(); engine.profile_enable(True)
            for _ in range(4):
                fa.inner_loop(batch, md, cfg, eps=ep)
            torch.cuda.synchronize()
            engine.profile_enable(False)
            for p in engine.profile_report():
                if "spline_wide" in p["kernel"] or ", 4, 11>" in p["kernel"] or ", 4, 9>" in p["kernel"]:
                    res[spec].append(p["ms"] / p["launches"])
        for k, _ in sets:
            lib.fc_debug_set(k, defaults[k])
for spec in specs:
    r = res[spec]
    print(f"{spec or '(shipped)':12s} " + ("refused by this build" if not r else "  ".join(f"{x * 1e3:7.1f}" for x in r) + f"   us per launch (min {min(r) * 1e3:.1f})"))
