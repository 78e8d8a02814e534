"""Diagnostic: forward log-prob with out-of-fp16-range coordinates, per GEMM variant, vs the fp64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import flowcompare_amd as fa
from flowcompare_amd import engine
from oracle import flow_oracle as O
DEV = "cuda:0"
cfg = fa.named_config("c2_dgcnn_attn_spline", n_flow_layers=3, sample_size=150)
torch.manual_seed(5)
md = fa.initialize_flow(cfg, device=DEV, mode="test")
g = torch.Generator().manual_seed(6)
B, N, M = 2, 150, 200
e0, e1 = torch.rand(B, M, 6, generator=g), torch.rand(B, N, 6, generator=g)
eps = [torch.randn(B, N, 294, generator=g)]
sd_f = {k: v.cpu().double() for k, v in md["flow"].state_dict().items()}
sd_e = {k: v.cpu().double() for k, v in md["input_embedder"].state_dict().items()}
for scale in (1.0, 1e3, 7e4, 1e5):
    x = e1.clone(); x[:, :, :3] *= scale
    with torch.no_grad():
        _, lp_o, _ = O.inner_loop(cfg, sd_f, sd_e, (e0.double(), x.double(), None), [e.double() for e in eps])
        _, lp_32, _ = O.inner_loop(cfg, {k: v.float() for k, v in sd_f.items()}, {k: v.float() for k, v in sd_e.items()}, (e0, x, None), eps)
    for var in (5, 3, 2):
        engine.lib().fc_debug_set(0, var)
        _, lp, _ = fa.inner_loop((e0.to(DEV), x.to(DEV), None), md, cfg, eps=[e.to(DEV) for e in eps])
        lp = lp.cpu().double()
        print(f"scale {scale:g} var {var}: finite hip {bool(torch.isfinite(lp).all())} oracle64 {bool(torch.isfinite(lp_o).all())} oracle32 {bool(torch.isfinite(lp_32).all())} "
              f"nan count {int(torch.isnan(lp).sum())} max rel {((lp - lp_o).abs() / (1 + lp_o.abs())).nan_to_num(1e9).max().item():.2e} lp range {lp_o.min().item():.3e}")
engine.lib().fc_debug_set(0, 5)
