"""Diagnostic: HIP engine vs the fp64 oracle on the C2 configuration with module-initialised weights, by depth and cloud size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import flowcompare_amd as fa
from oracle import flow_oracle as O
DEV = "cuda:0"
for L, N in ((2, 256), (8, 256), (32, 256), (8, 2048), (115, 256)):
    cfg = fa.named_config("c2_dgcnn_attn_spline", sample_size=N, n_flow_layers=L)
    torch.manual_seed(11)
    md = fa.initialize_flow(cfg, device=DEV, mode="test")
    g = torch.Generator().manual_seed(12)
    B = 2
    xyz = torch.rand(B, 2 * N, 3, generator=g) * 2 - 1
    xyz = xyz - xyz.mean(1, keepdim=True)
    xyz = xyz / xyz.norm(dim=-1).amax(1)[:, None, None]
    pts = torch.cat((xyz, torch.rand(B, 2 * N, 3, generator=g)), -1)
    e0, e1 = pts[:, :N].contiguous(), pts[:, N:].contiguous()
    eps = torch.randn(B, N, 294, generator=g)
    _, lp, bpd = fa.inner_loop((e0.to(DEV), e1.to(DEV), None), md, cfg, eps=[eps.to(DEV)])
    sd_f = {k: v.cpu().double() for k, v in md["flow"].state_dict().items()}
    sd_e = {k: v.cpu().double() for k, v in md["input_embedder"].state_dict().items()}
    t0 = time.time()
    with torch.no_grad():
        _, lp_o, bpd_o = O.inner_loop(cfg, sd_f, sd_e, (e0.double(), e1.double(), None), [eps.double()])
        emb_o = O.context_embed(cfg, sd_e, e0.double()) if hasattr(O, "context_embed") else None
    emb = md["input_embedder"](e0.to(DEV)).cpu().double()
    d = (lp.cpu().double() - lp_o).abs()
    de = (emb - emb_o).abs().max().item() if emb_o is not None else float("nan")
    print(f"L={L} N={N}: lp mean {lp_o.mean():.3f} per-point max {d.max():.3e} mean {d.mean():.3e} bpd diff {abs(float(bpd) - float(bpd_o)):.3e} "
          f"emb max diff {de:.3e}  (oracle {time.time() - t0:.0f} s)", flush=True)
