"""In-kernel slot stamps of the staggered attention kernel (knob 20 = 5): cycles each group spends in each slot of a tile and at the barriers.
    python profiles/micro/attn_stamps.py [N M]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowcompare_amd import engine
N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
L = engine.lib()
g = torch.Generator().manual_seed(0)
q, k, v = (torch.rand(16, n, 64, generator=g).cuda() - 0.5 for n in (N, M, M))
for _ in range(3):
    engine.op_attention(q, k, v, 0.125)
L.fc_debug_set(20, 5)
engine.op_attention(q, k, v, 0.125)
torch.cuda.synchronize()
L.fc_debug_set(20, 0)
buf = (ctypes.c_uint64 * 512)()
L.fc_debug_gemm_stamps.restype = ctypes.c_int64
n = L.fc_debug_gemm_stamps(buf, 512)
st = list(buf)
for grp in (0, 1):
    a = st[grp * 256: grp * 256 + 192]
    arrive, leave = a[0::2], a[1::2]
    print(f"group {grp}: slot work (barrier leave -> next arrive) and wait at the barrier, cycles, slots 1..3 of tiles 1..8 (slot 0 = prologue)")
    for t in range(1, 9):
        row = []
        for sl in range(3):
            i = 1 + 3 * t + sl
            row.append(f"{arrive[i] - leave[i - 1]:6d}+{leave[i] - arrive[i]:5d}")
        print(f"  tile {t}: " + "  ".join(row) + f"   tile total {leave[3 + 3 * t] - leave[3 * t]:6d}")
