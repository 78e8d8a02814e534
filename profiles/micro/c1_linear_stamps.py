"""In-kernel phase stamps (knob 20 = 2) of the limb-chained 512 -> 512 Linear launches at C1's size (2 x 1024 points: 64 x 64 tiles):
where the 11 us of such a launch go.      python profiles/micro/c1_linear_stamps.py"""
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
cfg = fa.named_config("c1_dgcnn_global_affine", sample_size=1024)
torch.manual_seed(0)
md = fa.initialize_flow(cfg, device=DEV, mode="test")
g = torch.Generator().manual_seed(1)
B, N = 2, 1024
e0, e1 = torch.rand(B, N, 6, generator=g).to(DEV), torch.rand(B, N, 6, generator=g).to(DEV)
eps = [torch.randn(B, N, cfg["latent_dim"] - cfg["input_dim"], generator=g).to(DEV)] if cfg["latent_dim"] > cfg["input_dim"] else None
for _ in range(2):
    fa.inner_loop((e0, e1, None), md, cfg, eps=eps)
lib.fc_debug_set(20, 2)
fa.inner_loop((e0, e1, None), md, cfg, eps=eps)
torch.cuda.synchronize()
buf = np.zeros(1 << 21, dtype=np.uint64)
n = lib.fc_debug_gemm_stamps(buf.ctypes.data, buf.size)
lib.fc_debug_set(20, 0)
st = buf[:n].reshape(-1, 16).astype(np.int64)
wall = (st[:, 9] - st[:, 8]) / 100.0
span = (st[:, 9].max() - st[:, 8].min()) / 100.0
t = st[:, [0, 1, 2, 3, 6]]
d = np.diff(t, axis=1)
cyc = t[:, -1] - t[:, 0]
ghz = np.median(cyc[wall > 0] / wall[wall > 0]) / 1e3
print(f"{len(st)} workgroups of the last launch, launch span (first entry -> last exit) {span:.2f} us, clock {ghz:.2f} GHz, workgroup life mean {wall.mean():.2f} us  p90 {np.percentile(wall, 90):.2f}")
print(f"    first workgroup entry -> mean entry {((st[:, 8] - st[:, 8].min()) / 100.0).mean():.2f} us, last entry {((st[:, 8] - st[:, 8].min()) / 100.0).max():.2f} us")
for k, nm in enumerate(["entry -> first DMA issued", "first k tile landed", "k loop (rest)", "epilogue"]):
    print(f"    {nm:28s} mean {d[:, k].mean():8.0f} cyc ({d[:, k].mean() / ghz / 1e3:5.2f} us)  p10 {np.percentile(d[:, k], 10):7.0f}  p90 {np.percentile(d[:, k], 90):7.0f}")
