"""CPU oracle for the PAConv context embedder (PointNet++ SSG U-Net with PAConv layers).  TEST INFRASTRUCTURE ONLY.

Parity status, per part:
  * PINNED by golden vectors (tests/golden/e2e_paconv_*.npz, emb_paconv.npz): the network wiring, QueryAndGroup, PAConv /
    ScoreNet / weight-bank maths, SharedMLP, feature propagation and the head MLP — the fixtures are produced by running the
    reference's own Python classes (PointNet2SSGSeg.forward etc.) on CPU;
  * restated from the reference's CUDA sources and NOT runnable here (the reference needs a CUDA device for them, SURVEY.md
    F9): the six pointops kernels below (the three with arithmetic and tie rules literally, in C: oracle/pointops_oracle.c).  For the golden run they are substituted INTO the reference by gen_golden.py, so the
    fixtures pin everything around them but the kernels themselves are "parity unpinned" at reference level (no golden
    vectors exist for them in the reference).  Each cites the .cu lines it follows.

Paths are relative to models/scene_seg_PAConv/ in the reference repository.
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ the six pointops kernels (CUDA in the reference)
# Three of them carry arithmetic and tie rules (furthest point sampling, heap k-NN, three nearest neighbours): those are restated LITERALLY in C
# (oracle/pointops_oracle.c: the kernels' loops, their max-heap, the distance as nvcc's default -fmad=true contracts it) and called through
# ctypes; the library is built on first use with gcc (-ffp-contract=off: every fusion is explicit in the source).  Round 3's torch versions
# (separately rounded squares, stable argsort instead of the heap's order) deviated from the kernels in exactly those two points.
_LIB = None


def _pointops_lib():
    global _LIB
    if _LIB is None:
        import ctypes
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        src, out = os.path.join(here, "pointops_oracle.c"), os.path.join(here, "_build", "libpointops_oracle.so")
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(out), exist_ok=True)
            tmp = f"{out}.{os.getpid()}.tmp"
            subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", tmp, src, "-lm"], check=True)
            os.replace(tmp, out)
        _LIB = ctypes.CDLL(out)
    return _LIB


def _suffix(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError(f"pointops oracle: float32 or float64, got {t.dtype}")


def _p(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def opt_n_threads(n):
    """lib/pointops/src/cuda_utils.h:15-18."""
    return max(min(1 << int(math.log(n) / math.log(2.0)), 1024), 1)


def furthest_sampling(xyz, m):
    """lib/pointops/src/sampling/sampling_cuda_kernel.cu:58-168.  xyz [B,n,3] -> idx [B,m] (int64).
    Start at index 0; temp = min(temp, d); arg-max with the kernel's tie rule: thread t scans k = t, t+T, ... keeping the FIRST
    maximum (strict >), the shared-memory tree keeps the LOWER thread on ties (pointops_oracle.c fps_*: the block simulated thread by thread)."""
    B, n, _ = xyz.shape
    idx = torch.zeros(B, max(m, 0), dtype=torch.int32)
    if m <= 0:
        return idx.long()
    x = xyz.detach().contiguous()
    getattr(_pointops_lib(), "fps_" + _suffix(x))(B, n, m, _p(x), _p(idx))
    return idx.long()


def gathering(feat_cf, idx):
    """lib/pointops/src/sampling/sampling_cuda_kernel.cu:6-20: out[b,c,j] = feat[b,c,idx[b,j]]."""
    return torch.gather(feat_cf, 2, idx[:, None, :].expand(-1, feat_cf.shape[1], -1).long())


def knnquery_heap(nsample, xyz, new_xyz):
    """lib/pointops/src/knnquery_heap/knnquery_heap_cuda_kernel.cu:21-89: nsample nearest of xyz [B,n,3] for every new_xyz [B,m,3] through
    the kernel's max-heap (strict `d2 < root` insertion, heap sort): ascending squared distance, equal distances in the HEAP's order; when
    n < nsample the unfilled slots keep (1e10, index 0) and sort to the end."""
    B, n, _ = xyz.shape
    m = new_xyz.shape[1]
    x, q = xyz.detach().contiguous(), new_xyz.detach().contiguous()
    idx = torch.zeros(B, m, nsample, dtype=torch.int32)
    d2 = torch.zeros(B, m, nsample, dtype=x.dtype)
    rc = getattr(_pointops_lib(), "knn_heap_" + _suffix(x))(B, n, m, nsample, _p(x), _p(q), _p(idx), _p(d2))
    if rc != 0:
        raise ValueError("knnquery_heap: nsample must be 1..100 (the kernel's fixed arrays)")
    return idx.long()


def grouping(feat_cf, idx):
    """lib/pointops/src/grouping/grouping_cuda_kernel.cu:60-74: out[b,c,p,s] = feat[b,c,idx[b,p,s]]."""
    B, C, _ = feat_cf.shape
    _, m, k = idx.shape
    return torch.gather(feat_cf, 2, idx.reshape(B, 1, m * k).expand(-1, C, -1).long()).reshape(B, C, m, k)


def nearest_neighbor3(unknown, known):
    """lib/pointops/src/interpolation/interpolation_cuda_kernel.cu:134-176 (+ sqrt in functions/pointops.py:112): the 3 nearest
    known points (strict '<' updates -> earliest index wins ties), best values kept in double, returned as float sqrt; with fewer than three
    known points the unfilled slots stay (1e40 -> +inf as float, index 0)."""
    B, n, _ = unknown.shape
    u, k = unknown.detach().contiguous(), known.detach().contiguous()
    d2 = torch.zeros(B, n, 3, dtype=u.dtype)
    idx = torch.zeros(B, n, 3, dtype=torch.int32)
    getattr(_pointops_lib(), "three_nn_" + _suffix(u))(B, n, known.shape[1], _p(u), _p(k), _p(d2), _p(idx))
    return torch.sqrt(d2), idx.long()


def interpolation(feat_cf, idx, weight):
    """lib/pointops/src/interpolation/interpolation_cuda_kernel.cu:181-195: sum of 3 weighted known features."""
    B, C, _ = feat_cf.shape
    n = idx.shape[1]
    g = torch.gather(feat_cf, 2, idx.reshape(B, 1, n * 3).expand(-1, C, -1).long()).reshape(B, C, n, 3)
    return (g[..., 0] * weight[:, None, :, 0] + g[..., 1] * weight[:, None, :, 1]) + g[..., 2] * weight[:, None, :, 2]


# ------------------------------------------------------------------ network (pure PyTorch in the reference)
def _bn(sd, p, x, dim=1):
    """BatchNorm over channel axis `dim`: running statistics in eval mode, the batch's biased statistics inside
    flow_oracle.train_mode() (module.train())."""
    from . import flow_oracle
    shape = [1] * x.dim()
    shape[dim] = -1
    if flow_oracle.BN_BATCH_STATS:
        red = tuple(d for d in range(x.dim()) if d != dim)
        mean, var = x.mean(red), x.var(red, unbiased=False)
    else:
        mean, var = sd[f"{p}.running_mean"], sd[f"{p}.running_var"]
    inv = torch.rsqrt(var + 1e-5)
    return (x - mean.reshape(shape)) * (inv * sd[f"{p}.weight"]).reshape(shape) + sd[f"{p}.bias"].reshape(shape)


def scorenet(sd, p, xyz_diff):
    """model/pointnet2/paconv.py:31-54, hidden [16], m = 8, last_bn False, softmax over m.  xyz_diff [B,3,N,K] -> [B,N,K,m]."""
    w0 = sd[f"{p}.mlp_convs_hidden.0.weight"][:, :, 0, 0]
    h = F.relu(_bn(sd, f"{p}.mlp_bns_hidden.0", torch.einsum("oc,bcnk->bonk", w0, xyz_diff)))
    w1 = sd[f"{p}.mlp_convs_hidden.1.weight"][:, :, 0, 0]
    s = torch.einsum("oc,bcnk->bonk", w1, h) + sd[f"{p}.mlp_convs_hidden.1.bias"][None, :, None, None]
    return torch.softmax(s, dim=1).permute(0, 2, 3, 1)


def paconv_layer(sd, p, in_feat, grouped_xyz, m=8):
    """model/pointnet2/paconv.py:107-153 (kernel_input 'neighbor', score_input 'identity', BN + ReLU).
    in_feat [B,C,N1,K], grouped_xyz [B,3,N1,K] -> [B,Cout,N1,K].  The centre is neighbour 0 of each group."""
    B, C, N1, K = in_feat.shape
    xyz_diff = grouped_xyz - grouped_xyz[..., :1]
    feat = torch.cat((in_feat - in_feat[..., :1], in_feat), 1)
    scores = scorenet(sd, f"{p}.scorenet", xyz_diff)                                  # [B,N1,K,m]
    bank = sd[f"{p}.weightbank"]                                                      # [2C, m*Cout]
    out = (feat.permute(0, 2, 3, 1) @ bank).reshape(B, N1, K, m, -1)
    out = (scores[..., None, :] @ out)[..., 0, :]                                     # paconv_util.py:52-56
    out = out.permute(0, 3, 1, 2)
    return F.relu(_bn(sd, f"{p}.bn", out))


def sa_module(sd, p, xyz, feat_cf, nsample=32):
    """model/pointnet2/pointnet2_paconv_modules.py:20-61 with QueryAndGroup (lib/pointops/functions/pointops.py:557-594), use_xyz,
    3 PAConv layers, max over the neighbours.  xyz [B,n,3], feat_cf [B,C,n] -> new_xyz [B,n//4,3], [B,Cout,n//4]."""
    npoint = xyz.shape[1] // 4
    idx = furthest_sampling(xyz, npoint)
    new_xyz = gathering(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    nidx = knnquery_heap(nsample, xyz, new_xyz)
    grouped_xyz = grouping(xyz.transpose(1, 2).contiguous(), nidx)
    x = torch.cat((grouped_xyz - new_xyz.transpose(1, 2)[..., None], grouping(feat_cf, nidx)), 1)
    i = 0
    while f"{p}.mlps.0.layer{i}.weightbank" in sd:
        x = paconv_layer(sd, f"{p}.mlps.0.layer{i}", x, grouped_xyz)
        i += 1
    return new_xyz, x.max(dim=-1)[0]


def fp_module(sd, p, unknown, known, unknown_feats, known_feats):
    """model/pointnet2/pointnet2_paconv_modules.py:206-238 + SharedMLP (util/block.py:14-39: conv1x1 no bias, BN, ReLU)."""
    dist, idx = nearest_neighbor3(unknown, known)
    rec = 1.0 / (dist + 1e-8)
    w = rec / rec.sum(dim=2, keepdim=True)
    x = interpolation(known_feats, idx, w)
    if unknown_feats is not None:
        x = torch.cat((x, unknown_feats), 1)
    i = 0
    while f"{p}.mlp.layer{i}.conv.weight" in sd:
        x = torch.einsum("oc,bcn->bon", sd[f"{p}.mlp.layer{i}.conv.weight"][:, :, 0, 0], x)
        x = F.relu(_bn(sd, f"{p}.mlp.layer{i}.bn.bn", x))
        i += 1
    return x


def paconv_embed(sd, pts):
    """model/pointnet2/pointnet2_paconv_seg.py:63-82: pts [B,M,3+c] -> [B,M,E]."""
    from . import flow_oracle
    xyz = pts[..., :3].contiguous()
    feats = pts[..., 3:].transpose(1, 2).contiguous()
    l_xyz, l_f = [xyz], [feats]
    for i in range(4):
        nx, nf = sa_module(sd, f"SA_modules.{i}", l_xyz[i], l_f[i])
        l_xyz.append(nx)
        l_f.append(nf)
    for i in range(-1, -5, -1):
        l_f[i - 1] = fp_module(sd, f"FP_modules.{4 + i}", l_xyz[i - 1], l_xyz[i], l_f[i - 1], l_f[i])
    return flow_oracle.mlp(sd, "out_mlp", l_f[0].permute(0, 2, 1), F.gelu)
