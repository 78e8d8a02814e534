"""Training step timing on one MI355X: forward + backward + Adam of the whole path on the HIP training primitives
at the bench workload's shape (C2: scenes x 4096 target + 4096 context points, spline layers): DGCNN embedder in train() mode
(BatchNorm batch statistics) and the flow; --frozen-embedder keeps the embedder in eval mode (inference kernels, no gradient).

    python profiles/train_bench.py --layers 115 --scenes 16 --steps 3        # prints one JSON line; kernel table with --profile
"""
import argparse
import json
import sys
import time
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flowcompare_amd as fa          # noqa: E402
from flowcompare_amd import engine, train_flow     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2_dgcnn_attn_spline", help="c2_dgcnn_attn_spline | c4_dgcnn_attn_extra_affine | c1_dgcnn_global_affine")
ap.add_argument("--layers", type=int, default=115)
ap.add_argument("--scenes", type=int, default=16)
ap.add_argument("--points", type=int, default=4096)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--profile", action="store_true")
ap.add_argument("--frozen-embedder", action="store_true")
ap.add_argument("--budget-gb", type=float, default=-1.0, help="activation budget in GB (-1 = default: 70 %% of free HBM, 0 = checkpoint every layer)")
a = ap.parse_args()
dev = "cuda:0"
if a.budget_gb >= 0:
    train_flow.ACTIVATION_BUDGET_BYTES = int(a.budget_gb * 2**30)
cfg = fa.named_config(a.config, sample_size=a.points, n_flow_layers=a.layers)
torch.manual_seed(0)
md = fa.initialize_flow(cfg, device=dev, mode="test")
md["flow"].train()
if not a.frozen_embedder:
    md["input_embedder"].train()
for m in md["flow"].modules():
    if hasattr(m, "initialized"):
        m.initialized.fill_(1.0)                       # ActNorm statistics as after the first batch / a checkpoint
g = torch.Generator().manual_seed(1)
B, N = a.scenes, a.points
xyz = torch.rand(B, 2 * N, 3, generator=g) * 2 - 1
xyz = xyz - xyz.mean(1, keepdim=True)
xyz = xyz / xyz.norm(dim=-1).amax(1)[:, None, None]
pts = torch.cat((xyz, torch.rand(B, 2 * N, 3, generator=g)), -1).to(dev)
extra = torch.rand(B, 1, generator=g).to(dev) * 15 if cfg["extra_z_value_context"] else None
batch = (pts[:, :N].contiguous(), pts[:, N:].contiguous(), extra)
eps = [torch.randn(B, N, 294, generator=g).to(dev)]
opt = torch.optim.Adam(md["parameters"], lr=1e-5)
lib = engine.lib()
times = []
for it in range(a.warmup + a.steps):
    if a.profile and it == a.warmup:
        lib.fc_profile_reset()
        lib.fc_profile_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, lp, bpd, norm = train_flow.training_step(batch, md, cfg, optimizer=opt, eps=eps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"step {it}: {dt * 1e3:.1f} ms  loss {loss.item():.4f}  |grad| {float(norm):.3e}  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    if it >= a.warmup:
        times.append(dt)
if a.profile:
    lib.fc_profile_enable(0)
    import ctypes
    buf = ctypes.create_string_buffer(1 << 16)
    lib.fc_profile_report(buf, ctypes.c_size_t(len(buf)))
    print(buf.value.decode())
ms = 1e3 * sum(times) / len(times)
print(json.dumps({"metric": "training step (forward + backward + Adam), points/s", "value": B * N / (ms / 1e3), "ms_per_step": ms, "layers": a.layers,
                  "config": a.config, "scenes": B, "points": N, "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2**30,
                  "embedder": "frozen (eval)" if a.frozen_embedder else "trained (BatchNorm batch statistics)"}))
