"""Diagnostic: the wide Linear kernel at row counts whose 256-row tile count is / is not a multiple of 8 (the XCD band order), range-flag fallbacks counted."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from flowcompare_amd import engine
import test_gpu_ops as T
DEV = "cuda:0"
lib = engine.lib(); lib.fc_debug_fp16_fallbacks.restype = ctypes.c_int64
sd = T._mlp_state(214, 2, seed=9)
for rows in (2048, 6144, 4096, 5000, 1024, 256):
    x0, x1 = T._rand(rows, 150, seed=11, scale=2.0), T._rand(rows, 64, seed=12)
    ref = T._mlp_ref(torch.cat((x0, x1), 1), sd, 2)
    f0 = lib.fc_debug_fp16_fallbacks()
    yw = engine.op_mlp_hidden(x0.to(DEV), x1.to(DEV), sd, use_rows="wide").cpu().double()
    f1 = lib.fc_debug_fp16_fallbacks()
    yg = engine.op_mlp_hidden(x0.to(DEV), x1.to(DEV), sd, use_rows=False).cpu().double()
    print(f"rows {rows}: wide fallbacks {f1 - f0}  |wide - fp64| {(yw - ref).abs().max():.2e}  |per-layer - fp64| {(yg - ref).abs().max():.2e}  |wide - per-layer| {(yw - yg).abs().max():.2e}", flush=True)
