"""Is the training step deterministic, and how much do its gradients move between the split-fp16 and the fp32-input MFMA loops?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import flowcompare_amd as fa                      # noqa: E402
from flowcompare_amd import train_flow, train_ops as T      # noqa: E402

dev = "cuda:0"
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = fa.named_config("c2_dgcnn_attn_spline", sample_size=1024, n_flow_layers=L)
torch.manual_seed(0)
md = fa.initialize_flow(cfg, device=dev, mode="test")
md["flow"].train()
for m in md["flow"].modules():
    if hasattr(m, "initialized"):
        m.initialized.fill_(1.0)
g = torch.Generator().manual_seed(1)
B, N = 4, 1024
pts = torch.rand(B, 2 * N, 6, generator=g).to(dev)
batch = (pts[:, :N].contiguous(), pts[:, N:].contiguous(), None)
eps = [torch.randn(B, N, 294, generator=g).to(dev)]


def grads(fp16, scale=1.0):
    for p in md["parameters"]:
        p.grad = None
    with T.step_guard(fp16=fp16, device=dev) as guard:
        loss, lp, bpd = fa.inner_loop(batch, md, cfg, eps=eps)
        (loss * scale).backward()
        over = guard.overflowed()
    return loss.item(), over, {n: p.grad.clone() / scale for n, p in md["flow"].named_parameters() if p.grad is not None}


l1, o1, g1 = grads(True)
l2, o2, g2 = grads(True)
l3, o3, g3 = grads(False)
same = all(torch.equal(g1[n], g2[n]) for n in g1)
norm = lambda g: sum(float((v.double() ** 2).sum()) for v in g.values()) ** 0.5
diff = sum(float(((g1[n] - g3[n]).double() ** 2).sum()) for n in g1) ** 0.5
print(f"layers {L}: loss {l1:.6f} / {l2:.6f} / fp32-input {l3:.6f}; overflow flags {o1} {o2} {o3}; two split-fp16 runs bit-identical: {same}; "
      f"|grad| {norm(g1):.6e} vs fp32-input {norm(g3):.6e}; |g_fp16 - g_fp32| / |g| = {diff / norm(g3):.2e}")

# the gradient panels of a mean loss over B * N points are O(1 / (B N)): far down in fp16's denormal range at real sizes.  A power-of-two
# loss scale moves them back (all backward operations are linear in the incoming gradient, the scaling is exact)
S = float(2 ** round(__import__("math").log2(B * N)))
l4, o4, g4 = grads(True, S)
d_scaled = sum(float(((g4[n] - g3[n]).double() ** 2).sum()) for n in g4) ** 0.5
print(f"layers {L}: with loss scale {S:.0f}: overflow {o4}; |g_fp16(scaled) - g_fp32| / |g| = {d_scaled / norm(g3):.2e}   (unscaled: {diff / norm(g3):.2e})")
