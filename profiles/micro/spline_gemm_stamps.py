"""In-kernel phase stamps of the fused spline GEMM (diagnostic knob 20) on the C2 workload: where a workgroup's time goes.

    python profiles/micro/spline_gemm_stamps.py [knob13 ...]        (default: 2 3)

Per variant: runs two C2 steps with module-init weights (kernel work does not depend on the values), reads the stamps of the last
fused-spline launch and prints per-phase shader cycles (mean over workgroups), the clock, the launch span, and the idle time between
consecutive workgroups of a CU slot."""
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
args = sys.argv[1:]
which = 2 if args and args[0] == "linear" else 1            # "linear": the limb-chained 512 -> 512 Linear launches (VAR 9) instead
if which == 2:
    args = args[1:]
variants = [int(v) for v in args] or ([4] if which == 2 else [2, 3])
cfg = fa.named_config("c2_dgcnn_attn_spline", sample_size=4096)
torch.manual_seed(0)
md = fa.initialize_flow(cfg, device=DEV, mode="test")
g = torch.Generator().manual_seed(1)
B, N = 16, 4096
e0, e1 = torch.rand(B, N, 6, generator=g).to(DEV), torch.rand(B, N, 6, generator=g).to(DEV)
eps = [torch.randn(B, N, cfg["latent_dim"] - cfg["input_dim"], generator=g).to(DEV)]
names = ["prologue issue", "first k tile lands", "main loop (rest)", "operands ready", "spline evaluation", "stores"]
for v in variants:
    lib.fc_debug_set(13, v)
    lib.fc_debug_set(20, 0)
    fa.inner_loop((e0, e1, None), md, cfg, eps=eps)
    lib.fc_debug_set(20, which)
    fa.inner_loop((e0, e1, None), md, cfg, eps=eps)
    torch.cuda.synchronize()
    buf = np.zeros(1 << 21, dtype=np.uint64)
    n = lib.fc_debug_gemm_stamps(buf.ctypes.data, buf.size)
    lib.fc_debug_set(20, 0)
    st = buf[:n].reshape(-1, 16).astype(np.int64)
    t = st[:, 0:7]
    if which == 2:
        t = t.copy(); t[:, 4] = t[:, 3]; t[:, 5] = t[:, 3]            # (Linear epilogue: one phase, reported under 'stores')
    d = np.diff(t, axis=1)
    wall = (st[:, 9] - st[:, 8]) / 100.0                      # us
    cyc = t[:, 6] - t[:, 0]
    ghz = np.median(cyc[wall > 0] / wall[wall > 0]) / 1e3
    span = (st[:, 9].max() - st[:, 8].min()) / 100.0
    print(f"knob 13 = {v}: {len(st)} workgroups, launch span {span:.1f} us, median clock {ghz:.2f} GHz, workgroup life mean {wall.mean():.2f} us")
    for k, nm in enumerate(names):
        print(f"    {nm:22s} mean {d[:, k].mean():9.0f} cyc  ({d[:, k].mean() / ghz / 1e3:6.2f} us)  p10 {np.percentile(d[:, k], 10):8.0f}  p90 {np.percentile(d[:, k], 90):8.0f}")
    if st[:, 10].any():                                       # one k step in detail (persistent kernel)
        dd = np.diff(st[:, 10:15], axis=1)
        for k, nm in enumerate(["wait own DMA", "barrier", "issue 8 DMA pieces", "20 ds_read + 24 MFMA issued"]):
            print(f"    k step 5: {nm:28s} mean {dd[:, k].mean():7.0f} cyc  p10 {np.percentile(dd[:, k], 10):7.0f}  p90 {np.percentile(dd[:, k], 90):7.0f}")
    # CU slots: workgroups of one CU (xcc, se, cu) sorted by start; two are resident at a time
    hw = st[:, 7]
    cu = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 8) & 0xF) * 16 + ((hw >> 12) & 1)
    busy, gaps = [], []
    for c in np.unique(cu):
        m = np.where(cu == c)[0]
        o = m[np.argsort(st[m, 8])]
        busy.append(((st[o, 9] - st[o, 8]).sum()) / 100.0)
        # greedy two-slot assignment: a new workgroup takes the slot that freed first
        free = [None, None]
        for i in o:
            k = 0 if (free[0] is None or (free[1] is not None and free[0] <= free[1])) else 1
            if free[k] is not None:
                gaps.append((st[i, 8] - free[k]) / 100.0)
            free[k] = st[i, 9]
    print(f"    CUs seen {len(np.unique(cu))}, workgroups per CU {len(st) / len(np.unique(cu)):.1f}, resident workgroup-time per CU {np.mean(busy):.1f} us "
          f"(= {np.mean(busy) / span:.2f} slots busy), slot hand-over gap mean {np.mean(gaps):.2f} us p90 {np.percentile(gaps, 90):.2f}")
lib.fc_debug_set(13, 2)
