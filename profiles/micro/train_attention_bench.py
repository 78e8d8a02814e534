"""Attention forward + backward (train_ops.AttentionFn) at the C2 size, 16 scenes x 4096 queries x 4096 keys, head dim 64, timed per kernel with
the in-library HIP events; also prints checksums of dq / dk / dv (same inputs every run: a changed kernel must reproduce them to rounding).
    python profiles/micro/train_attention_bench.py [B N M]"""
import sys
import torch
sys.path.insert(0, ".")
from flowcompare_amd import engine, train_ops

B, N, M = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (16, 4096, 4096)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
D = 64
q = train_ops.to_panel(torch.randn(B * N, D, generator=g)).to(dev).requires_grad_(True)
k = train_ops.to_panel(torch.randn(B * M, D, generator=g)).to(dev).requires_grad_(True)
v = train_ops.to_panel(torch.randn(B * M, D, generator=g)).to(dev).requires_grad_(True)
go = train_ops.to_panel(torch.randn(B * N, D, generator=g) * 1e-2).to(dev)
with train_ops.step_guard(device=dev) as guard:
    for it in range(5):
        if it == 2:
            torch.cuda.synchronize()
            engine.profile_filter(None); engine.profile_reset(); engine.profile_enable(True)
        out = train_ops.attention(q, k, v, B, N, M, D ** -0.5)
        out.backward(go)
        sums = [float(t.grad.double().abs().sum()) for t in (q, k, v)]
        q.grad = None; k.grad = None; v.grad = None
    torch.cuda.synchronize()
    engine.profile_enable(False)
    over = guard.overflowed()
print(f"B {B} N {N} M {M}: range flag {over}; sum|dq| {sums[0]:.9e} sum|dk| {sums[1]:.9e} sum|dv| {sums[2]:.9e}")
for p in sorted(engine.profile_report(), key=lambda p: -p["ms"]):
    tf = p["flops"] / (p["ms"] * 1e-3) / 1e12 if p["flops"] else 0.0
    print(f"    {p['kernel'][:70]:70s} {p['launches']:3d} launches  {p['ms'] / p['launches'] * 1e3:8.1f} us  {tf:6.1f} TFLOP/s")
