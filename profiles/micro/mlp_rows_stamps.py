"""In-kernel phase stamps of the row-resident coupling-MLP chain (csrc/mlprows.hip, knob 20 = 4).

    python profiles/micro/mlp_rows_stamps.py [rows] [hidden layers]

Prints, per stage class, the mean shader-clock cycles between the stamp points of a stage (entry, own DMA landed, barrier passed, 24 slots issued) for workgroup 0 (first round) and
workgroup 300 (second round on a 256-CU part)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowcompare_amd import engine  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n_mid = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g = torch.Generator().manual_seed(0)
u = lambda *sh, sc=1.0: (torch.rand(*sh, generator=g) * 2 - 1) * sc
sd = {"net.in_layer.weight": u(512, 214, sc=0.1), "net.in_layer.bias": u(512, sc=0.2), "net.out_layer.weight": torch.zeros(1, 512)}
for i in range(n_mid):
    sd[f"net.layers.{i}.weight"] = u(512, 512, sc=0.08)
    sd[f"net.layers.{i}.bias"] = u(512, sc=0.2)
x0, x1 = u(rows, 150, sc=2.0).cuda(), u(rows, 64).cuda()
lib = engine.lib()
engine.op_mlp_hidden(x0, x1, sd)
lib.fc_debug_set(20, 4)
engine.op_mlp_hidden(x0, x1, sd)
lib.fc_debug_set(20, 0)
buf = (ctypes.c_uint64 * (2 * 256 * 8))()
lib.fc_debug_gemm_stamps.restype = ctypes.c_int64
n = lib.fc_debug_gemm_stamps(buf, len(buf))
st = np.frombuffer(buf, dtype=np.uint64).reshape(2, 256, 8).astype(np.int64)
names = ["wait own DMA", "barrier", "24 slots (MFMA + epilogue + DMA issue)"]
for w, label in ((0, "workgroup 0"), (1, "workgroup 300")):
    s = st[w]
    live = np.where(s[:, 5] > 0)[0]
    if len(live) == 0:
        print(label, "no stamps")
        continue
    print(f"{label}: {len(live)} stages, total {int(s[live[-1], 5] - s[live[0], 0])} cycles")
    d = np.diff(s[live][:, [0, 1, 2, 5]], axis=1)
    gap = s[live][1:, 0] - s[live][:-1, 5]
    for lo, hi, nm in ((0, 32, "layer 0 (2 stages per block)"), (32, 96, "hidden layer 1"), (96, 160, "hidden layer 2 (residual)")):
        sel = [i for i, k in enumerate(live) if lo <= k < hi]
        if not sel:
            continue
        m = d[sel].mean(0)
        gsel = [i for i in sel if i < len(gap)]
        print(f"  {nm}: " + ", ".join(f"{names[i]} {m[i]:.0f}" for i in range(3)) + f", stage total {m.sum():.0f}, to next stage {gap[gsel].mean():.0f}")
    big = np.argsort(gap)[-6:]
    print("  largest gaps between stages (stage index, cycles):", [(int(live[i]), int(gap[i])) for i in big])
