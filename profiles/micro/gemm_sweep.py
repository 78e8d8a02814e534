#!/usr/bin/env python3
"""Occupancy / K sweep of the 128x128 fp32-MFMA GEMM: does a second co-resident workgroup overlap with the first?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowcompare_amd import engine
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
def run(rows, N, K, reps=5):
    x = (torch.rand(rows, K, generator=g) - 0.5).to(dev); W = ((torch.rand(N, K, generator=g) - 0.5) * K ** -0.5).to(dev)
    engine.op_linear(x, W); torch.cuda.synchronize()
    engine.profile_reset(); engine.profile_enable(True)
    for _ in range(reps): engine.op_linear(x, W)
    torch.cuda.synchronize(); engine.profile_enable(False)
    for p in engine.profile_report():
        if "gemm" in p["kernel"]:
            ms = p["ms"] / p["launches"]
            blocks = (rows // 128) * ((N + 127) // 128)
            print(f"rows={rows:7d} N={N:5d} K={K:5d} blocks={blocks:6d} ({blocks/256:5.1f}/CU)  {ms*1e3:9.1f} us  {2.0*rows*N*K/ms/1e9:7.1f} TF")
for K in (512, 4096):
    for rows in (32768, 65536, 98304, 131072, 262144):
        run(rows, 128, K)
for K in (512, 2048):
    run(65536, 512, K); run(65536, 1024, K); run(65536, 4096, K)
