#!/usr/bin/env python3
"""Per-shape microbenchmark of the library's kernels (GPU box only): drives the fc_op_* entry points with the in-library
HIP-event profiler on and prints achieved TFLOP/s per GEMM shape of the C2/C4 layer, attention and kNN shapes."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flowcompare_amd import engine

dev = "cuda:0"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
reps = 5


def run(fn):
    fn()
    torch.cuda.synchronize()
    engine.profile_reset(); engine.profile_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    engine.profile_enable(False)
    return engine.profile_report()


for kv in filter(None, os.environ.get("KNOBS", "").split(",")):      # e.g. KNOBS=2=6 (column group of the tile order)
    engine.lib().fc_debug_set(*map(int, kv.split("=")))
print(f"rows={rows} knobs={os.environ.get('KNOBS', '')}")
variants = [(int(v[0]), int(v[1])) for v in os.environ.get("VARIANTS", "51").split(",")]   # (loop variant 5/3/2, big tile 0/1/2)
shapes = [("pre_in", 256, 150, "gelu", False), ("pre_mid", 256, 256, "gelu", True), ("pre_out", 256, 256, "none", False),
          ("q_proj", 64, 256, "none", False), ("cpl_in", 512, 214, "gelu", False), ("cpl_mid", 512, 512, "gelu", True),
          ("cpl_mid_noact", 512, 512, "none", False), ("cpl_mid_gelu", 512, 512, "gelu", False), ("cpl_mid_res", 512, 512, "none", True), ("affine_out", 300, 512, "none", False), ("spline_out", 3750, 512, "none", False),
          ("lu", 300, 300, "none", False), ("kv", 1024, 64, "none", False)]
g = torch.Generator().manual_seed(0)
for name, N, K, act, res in shapes:
    if only and name not in only:
        continue
    x = torch.rand(rows, K, generator=g).to(dev) - 0.5
    W = ((torch.rand(N, K, generator=g) - 0.5) * K ** -0.5).to(dev)
    b = torch.rand(N, generator=g).to(dev)
    r = torch.rand(rows, N, generator=g).to(dev) if res else None
    for var, stag in variants:
        engine.lib().fc_debug_set(0, var); engine.lib().fc_debug_set(3, stag)
        rep = run(lambda: engine.op_linear(x, W, b, r, act))
        for p in rep:
            if "gemm" in p["kernel"]:
                ms = p["ms"] / p["launches"]
                print(f"{name:14s} N={N:5d} K={K:4d} act={act:5s} res={int(res)} var={var} bigtile={stag} {ms*1e3:9.1f} us  {2.0*rows*N*K/ms/1e9:7.1f} TF   {p['kernel'][21:52]}")

for B, N, M, D in [(16, 4096, 4096, 64), (2, 16384, 16384, 64), (16, 1024, 1250, 64)]:
    if only and "attention" not in only:
        continue
    q = torch.rand(B, N, D, generator=g).to(dev); k = torch.rand(B, M, D, generator=g).to(dev); v = torch.rand(B, M, D, generator=g).to(dev)
    rep = run(lambda: engine.op_attention(q, k, v, 0.125))
    for p in rep:
        if "attn" in p["kernel"]:
            ms = p["ms"] / p["launches"]
            print(f"attention B={B} N={N} M={M} D={D}: {ms*1e3:9.1f} us  {4.0*B*N*M*D/ms/1e9:7.1f} TF")

for B, M, C in [(16, 4096, 6), (16, 4096, 64), (16, 4096, 128)]:
    if only and "knn" not in only:
        continue
    f = torch.rand(B, M, C, generator=g).to(dev)
    rep = run(lambda: engine.op_knn(f, 40))
    for p in rep:
        if "knn" in p["kernel"]:
            print(f"knn B={B} M={M} C={C}: {p['ms']/p['launches']*1e3:9.1f} us")
