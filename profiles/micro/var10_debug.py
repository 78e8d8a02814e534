"""Compares fused-spline GEMM variants (knob 13) / k rotation (knob 21) on a short C2 stack: max |log-prob difference| to the shipped kernel."""
import sys, torch
sys.path.insert(0, ".")
import flowcompare_amd as fa
from flowcompare_amd import engine
lib = engine.lib()
DEV = "cuda:0"
cfg = fa.named_config("c2_dgcnn_attn_spline", n_flow_layers=2, sample_size=2048)
torch.manual_seed(7)
md = fa.initialize_flow(cfg, device=DEV, mode="test")
g = torch.Generator().manual_seed(8)
B, N, M = 8, 2048, 2048
e0, e1 = torch.rand(B, M, 6, generator=g), torch.rand(B, N, 6, generator=g)
eps = [torch.randn(B, N, 294, generator=g).to(DEV)]
batch = (e0.to(DEV), e1.to(DEV), None)
out = {}
for v, r in ((2, 0), (3, 0), (4, 0), (4, 1)):
    lib.fc_debug_set(13, v); lib.fc_debug_set(21, r)
    _, lp, _ = fa.inner_loop(batch, md, cfg, eps=eps)
    out[(v, r)] = lp.clone()
    d = (out[(2, 0)] - lp).abs()
    print(f"knob 13 = {v}, 21 = {r}: max diff {d.max().item():.3e} mean {d.mean().item():.3e}")
lib.fc_debug_set(13, 2); lib.fc_debug_set(21, 1)
