"""Error of the attention kernels (split-fp16 vs fp32-input MFMA) against fp64, per shape and per structured input (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowcompare_amd import engine
dev = "cuda:0"
lib = engine.lib()
def rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
for B, N, M, D in [(2, 128, 64, 64), (1, 32, 64, 64), (3, 100, 130, 64), (1, 20, 24, 32), (2, 257, 1000, 64)]:
    q, k, v = rand(B, N, D, seed=1, scale=2.0), rand(B, M, D, seed=2, scale=2.0), rand(B, M, D, seed=3)
    scale = D ** -0.5
    w = torch.softmax(q.double() @ k.double().transpose(1, 2) * scale, -1)
    ref = w @ v.double()
    for knob in (1, 0):
        lib.fc_debug_set(5, knob)
        out = engine.op_attention(q.to(dev), k.to(dev), v.to(dev), scale).cpu().double()
        print(f"B{B} N{N} M{M} D{D} fp16={knob}: max err {(out - ref).abs().max().item():.3e}")
    # structured: one-hot attention (huge logit on key j0) -> out = v[j0]; then v = key index -> shows which key is picked
    lib.fc_debug_set(5, 1)
    vv = torch.arange(M).float()[None, :, None].expand(B, M, D).contiguous() + torch.arange(D).float()[None, None, :] * 0.001
    out = engine.op_attention(torch.zeros(B, N, D).to(dev), k.to(dev), vv.to(dev), scale).cpu().double()
    print("   uniform attention over v=key index: expect", (M - 1) / 2, "got", out[0, 0, 0].item(), out[0, N - 1, D - 1].item() - 0.001 * (D - 1))
