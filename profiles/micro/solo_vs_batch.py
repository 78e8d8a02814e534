"""Diagnostic: a scene alone against the same scene inside a batch (test_persistent_spline_gemm_walks_several_tiles_per_workgroup's check) under knob settings."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import flowcompare_amd as fa
from flowcompare_amd import engine
DEV = "cuda:0"
lib = engine.lib()
import ctypes
lib.fc_debug_fp16_fallbacks.restype = ctypes.c_int64
B, N = 3, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = fa.named_config("c2_dgcnn_attn_spline", n_flow_layers=2, sample_size=N)
torch.manual_seed(21)
md = fa.initialize_flow(cfg, device=DEV, mode="test")
g = torch.Generator().manual_seed(22)
e0, e1 = torch.rand(B, 200, 6, generator=g), torch.rand(B, N, 6, generator=g)
eps = [torch.randn(B, N, 294, generator=g).to(DEV)]
batch = (e0.to(DEV), e1.to(DEV), None)
for spec in sys.argv[2:] or [""]:
    sets = [tuple(int(x) for x in kv.split("=")) for kv in spec.split(",") if kv]
    for k, v in sets:
        assert lib.fc_debug_set(k, v) == 0
    f0 = lib.fc_debug_fp16_fallbacks()
    _, ref, _ = fa.inner_loop(batch, md, cfg, eps=eps)
    torch.cuda.synchronize(); print('   batch run: fallbacks', lib.fc_debug_fp16_fallbacks() - f0)
    for sc in range(B):
        _, solo, _ = fa.inner_loop((batch[0][sc:sc + 1], batch[1][sc:sc + 1], None), md, cfg, eps=[eps[0][sc:sc + 1]])
        torch.cuda.synchronize(); fb = lib.fc_debug_fp16_fallbacks() - f0
        d = (solo[0] - ref[sc]).abs()
        print(f"knobs {spec or '(shipped)'} scene {sc}: max |solo - batch| {d.max().item():.3e}, rows differing {(d > 0).sum().item()} of {N}, first {(d > 0).nonzero()[:4].flatten().tolist()} fallbacks so far {fb}")
    for k, _ in sets:
        lib.fc_debug_set(k, {29: 0, 13: 5, 23: 1, 16: 1, 9: 1, 8: 2, 26: 1}[k])
