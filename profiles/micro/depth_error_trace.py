"""Diagnostic (CPU, oracle only): fp32-vs-fp64 distance of the latent and of the log-det along the worst row of a conditioned 115-layer
C2 stack, layer by layer.  usage: python profiles/micro/depth_error_trace.py OUT_SCALE LU_SCALE"""
import os, sys, time, contextlib, io, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import flowcompare_amd as fa
from oracle import flow_oracle as O
torch.set_num_threads(8)
name = "c2_dgcnn_attn_spline"; L = 115; N = 512; M = 512; alpha = float(sys.argv[1]); lu = float(sys.argv[2])
cfg = fa.named_config(name, sample_size=N, n_flow_layers=L)
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    md = fa.initialize_flow(cfg, device="cpu", mode="test")
def draw(seed):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(1, N + M, 3, generator=g) * 2 - 1
    xyz = xyz - xyz.mean(1, keepdim=True); xyz = xyz / xyz.norm(dim=-1).amax(1)[:, None, None]
    pts = torch.cat((xyz, torch.rand(1, N + M, 3, generator=g)), -1)
    return pts[:, :M].contiguous(), pts[:, M:].contiguous(), torch.randn(1, N, 294, generator=g)
sd = {k: v.clone() for k, v in md["flow"].state_dict().items()}
se = {k: v.clone() for k, v in md["input_embedder"].state_dict().items()}
K = 8; d2 = 150; u0 = math.log(math.exp(1 - 1e-3) - 1)
gen = torch.Generator().manual_seed(5)
for k in list(sd):
    if k.endswith("transform.nn.out_layer.weight"): sd[k] = sd[k] * alpha
    if k.endswith("transform.nn.out_layer.bias"):
        b = (sd[k] * alpha).reshape(d2, 3 * K + 1); b[:, 2 * K:] += u0; sd[k] = b.reshape(-1)
    if lu > 0 and (k.endswith("lower_entries") or k.endswith("upper_entries")):
        sd[k] = (torch.rand(sd[k].shape, generator=gen) * 2 - 1) * lu / math.sqrt(300)
sd64 = {k: v.double() for k, v in sd.items()}; se64 = {k: v.double() for k, v in se.items()}
e0, e1, eps = draw(12)
with torch.no_grad(), O.actnorm_data_init():
    O.inner_loop(cfg, sd64, se64, (e0.double(), e1.double(), None), [eps.double()])
sd = {k: v.float() if v.is_floating_point() else v for k, v in sd64.items()}
sd64 = {k: v.double() for k, v in sd.items()}
e0, e1, eps = draw(99)
with torch.no_grad():
    emb = O.context_embed(cfg, se64, e0.double())
    r64, r32 = [], []
    lp64 = O.flow_log_prob(cfg, sd64, e1.double(), emb, None, [eps.double()], record=r64)
    lp32 = O.flow_log_prob(cfg, sd, e1, emb.float(), None, [eps], record=r32)
d = (lp32.double() - lp64).abs()[0]
margin = O.spline_domain_margin(cfg, r64)[0]
far = margin > 1e-4
print("kept", int(far.sum()), "max", float(d[far].max()), "mean", float(d[far].mean()))
worst = int((d * far).argmax())
print("worst row", worst, "err", float(d[worst]), "lp64", float(lp64[0, worst]))
lay = O._layout(cfg)
cum = 0.0
for i, ((kind, idx), (x64, l64), (x32, l32)) in enumerate(zip(lay, r64, r32)):
    dx = (x32[0, worst].double() - x64[0, worst]).abs().max().item()
    dl = (l32[0, worst].double() - l64[0, worst]).item() if l64.dim() > 1 else float(l32.reshape(-1)[0].double() - l64.reshape(-1)[0])
    cum += dl
    if kind == "block" and (i % 30 == 1 or abs(dl) > 1e-4):
        print(f"  t{idx} {kind}: max|dx| {dx:.2e}  dldj {dl:+.2e} cum {cum:+.2e}  |x64|max {x64[0, worst].abs().max():.2f}")
# overall stats of activation ranges
print("max |x| over all rows at last layer", float(r64[-1][0].abs().max()))
big = []
cum = 0
for i, ((kind, idx), (x64, l64), (x32, l32)) in enumerate(zip(lay, r64, r32)):
    if kind != "block": continue
    dl = (l32[0, worst].double() - l64[0, worst]).item()
    big.append((abs(dl), i, idx, dl))
big.sort(reverse=True)
print("largest per-layer ldj errors for worst row:", [(f"t{idx}", f"{dl:+.2e}") for _, i, idx, dl in big[:5]])
_, i, idx, dl = big[0]
xin64, xin32 = r64[i - 1][0][0, worst], r32[i - 1][0][0, worst]
# recompute the spline params for that layer in both precisions
act = O._act(cfg["coupling_block_nonlinearity"])
def params(sdx, x, ctx, dt):
    c = O._precondition(cfg, sdx, f"transforms.{idx}", x[None, None].to(dt), ctx.to(dt), None, act, False)
    p = O.mlp(sdx, f"transforms.{idx}.transform.nn", torch.cat((x[None, None, :150].to(dt), c), -1), act).reshape(150, 25)
    return p
p64 = params(sd64, xin64, emb, torch.float64); p32 = params(sd, xin32, emb, torch.float32)
y64, lad64 = O.rq_spline(xin64[150:], p64[:, :8], p64[:, 8:16], p64[:, 16:])
y32, lad32 = O.rq_spline(xin32[150:], p32[:, :8], p32[:, 8:16], p32[:, 16:])
dd = (lad32.double() - lad64).abs()
j = int(dd.argmax())
print(f"layer t{idx}: dim {j} lad err {dd[j]:.3e} x64 {xin64[150+j]:.8f} x32 {xin32[150+j]:.8f} lad64 {lad64[j]:.5f} lad32 {lad32[j]:.5f}; max param diff {(p32.double()-p64).abs().max():.2e}")
y32b, lad32b = O.rq_spline(xin64[150:].float(), p64.float()[:, :8], p64.float()[:, 8:16], p64.float()[:, 16:])
print(f"  same inputs rounded to fp32, fp32 spline arithmetic: lad err {(lad32b.double()-lad64).abs()[j]:.3e}")
print("trace of dim", j)
prevd = 0
for i, ((kind, idx), (x64, l64), (x32, l32)) in enumerate(zip(lay, r64, r32)):
    dxj = (x32[0, worst, 150 + j].double() - x64[0, worst, 150 + j]).item()
    if abs(dxj) > 3 * abs(prevd) + 1e-6 or abs(dxj) < abs(prevd) / 3:
        print(f"   t{idx} {kind}: x64 {x64[0, worst, 150+j]:.6f} dx {dxj:+.2e}")
    prevd = dxj
