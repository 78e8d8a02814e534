"""One training Linear (forward + backward through train_ops.LinearActFn) at the C2 layer shapes, timed per kernel with the in-library
HIP events:   python profiles/micro/train_linear_bench.py [K=V knob ...] [only=hidden]
Shapes (65536 rows): hidden 512 -> 512 GELU, coupling in_layer 214 -> 512, spline parameter layer 512 -> 3750."""
import sys
import torch
sys.path.insert(0, ".")
from flowcompare_amd import engine, train_ops

lib = engine.lib()
only = [a[5:] for a in sys.argv[1:] if a.startswith("only=")]
for kv in sys.argv[1:]:
    if kv.startswith("only="):
        continue
    k, v = kv.split("=")
    assert lib.fc_debug_set(int(k), int(v)) == 0, kv
dev = torch.device("cuda", 0)
rows = 65536
g = torch.Generator().manual_seed(0)
for name, K, N, act in (("hidden 512->512 GELU", 512, 512, "GELU"), ("in_layer 214->512 GELU", 214, 512, "GELU"), ("spline out 512->3750", 512, 3750, None)):
    if only and not any(o in name for o in only):
        continue
    x = train_ops.to_panel((torch.randn(rows, K, generator=g) * 0.5)).to(dev).requires_grad_(True)
    W = (torch.randn(N, K, generator=g) * 0.04).to(dev).requires_grad_(True)
    b = torch.zeros(N, device=dev, requires_grad=True)
    gy = None
    with train_ops.step_guard(device=dev) as guard:
        for it in range(6):
            if it == 2:
                torch.cuda.synchronize()
                engine.profile_filter(None); engine.profile_reset(); engine.profile_enable(True)
            y = train_ops.linear_act([x], [K], W, b, rows, act)
            if gy is None:
                gy = torch.randn(y.shape, generator=g).to(dev) * 1e-3
            y.backward(gy)
            x.grad = None; W.grad = None; b.grad = None
        torch.cuda.synchronize()
        engine.profile_enable(False)
        over = guard.overflowed()
    rep = engine.profile_report()
    print(f"{name}: (range flag {over})")
    for p in sorted(rep, key=lambda p: -p["ms"]):
        tf = p["flops"] / (p["ms"] * 1e-3) / 1e12 if p["flops"] else 0.0
        print(f"    {p['kernel'][:80]:80s} {p['launches']:3d} launches  {p['ms'] / p['launches'] * 1e3:8.1f} us  {tf:6.1f} TFLOP/s")
    del x, W, b, y, gy
