"""Layer-boundary stamps of the row-resident pre-attention chain kernel (knob 20 = 3) on the C2 workload: where a workgroup's time goes."""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
import flowcompare_amd as fa
from flowcompare_amd import engine
lib = engine.lib()
lib.fc_debug_gemm_stamps.restype = ctypes.c_int64
lib.fc_debug_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
DEV = torch.device("cuda", 0)
cfg = fa.named_config("c2_dgcnn_attn_spline", sample_size=4096, n_flow_layers=8)
torch.manual_seed(0)
md = fa.initialize_flow(cfg, device=DEV, mode="test")
g = torch.Generator().manual_seed(1)
B, N = 16, 4096
e0, e1 = torch.rand(B, N, 6, generator=g).to(DEV), torch.rand(B, N, 6, generator=g).to(DEV)
eps = [torch.randn(B, N, cfg["latent_dim"] - cfg["input_dim"], generator=g).to(DEV)]
fa.inner_loop((e0, e1, None), md, cfg, eps=eps)
lib.fc_debug_set(20, 3)
fa.inner_loop((e0, e1, None), md, cfg, eps=eps)
torch.cuda.synchronize()
buf = np.zeros(1 << 16, dtype=np.uint64)
n = lib.fc_debug_gemm_stamps(buf.ctypes.data, buf.size)
lib.fc_debug_set(20, 0)
st = buf[:n].reshape(-1, 16).astype(np.int64)
d = np.diff(st[:, 0:8], axis=1)
wall = (st[:, 15] - st[:, 14]) / 100.0
ghz = np.median((st[:, 7] - st[:, 0]) / wall) / 1e3
print(f"{len(st)} workgroups, span {(st[:, 15].max() - st[:, 14].min()) / 100.0:.1f} us, workgroup life {wall.mean():.1f} us, clock {ghz:.2f} GHz")
for k, nm in enumerate(["input split + biases", "in_layer (8 chunks, K 160)", "hidden 0", "hidden 1 (+ residual)", "out_layer", "LayerNorm", "q projection (2 chunks)"]):
    print(f"    {nm:28s} mean {d[:, k].mean():8.0f} cyc ({d[:, k].mean() / ghz / 1e3:6.2f} us)  p10 {np.percentile(d[:, k], 10):8.0f} p90 {np.percentile(d[:, k], 90):8.0f}")
tot = (st[:, 7] - st[:, 0]).mean()
for w, nm in ((0, "wave 0"), (1, "wave 4")):
    print(f"    {nm}: waited for its DMA pieces {st[:, 8 + 2 * w].mean():8.0f} cyc ({100 * st[:, 8 + 2 * w].mean() / tot:4.1f} %), at the chunk barriers {st[:, 9 + 2 * w].mean():8.0f} cyc ({100 * st[:, 9 + 2 * w].mean() / tot:4.1f} %) of {tot:.0f}")
