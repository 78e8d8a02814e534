"""ctypes binding of libfcflow.so (include/fcflow.h) + the handles the nn.Module mirrors use.

PyTorch is plumbing here: device memory (tensors), the current HIP stream and torch.distributed.  Every function
below hands raw device pointers to the C ABI; nothing is computed in PyTorch and there is no CPU fallback —
a missing or unloadable library raises immediately.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FCFLOW_LIB", os.path.join(_HERE, "libfcflow.so"))   # FCFLOW_LIB: A/B another build in profiles/kernel_bench.py
ABI_VERSION = 7

FLOW_TYPES = {"AffineCoupling": 0, "RationalQuadraticSplineCoupling": 1, "ExponentialCoupling": 2}
SCALE_FNS = {"exp": 0, "sigmoid": 1}
ACTS = {"GELU": 1, "RELU": 2, "ELU": 3}
PERMUTERS = {"LinearLU": 0, "random_permute": 1, "FullCombiner": 2, "ExponentialCombiner": 3}
EXPM = {"torch": 0, "original": 1}

EXPORTS = [
    "fc_abi_version", "fc_last_error",
    "fc_flow_create", "fc_flow_destroy", "fc_flow_workspace_bytes", "fc_flow_noise_count", "fc_flow_noise_width",
    "fc_flow_logprob_f32", "fc_flow_inverse_f32",
    "fc_dgcnn_create", "fc_dgcnn_destroy", "fc_dgcnn_out_dim", "fc_dgcnn_workspace_bytes", "fc_dgcnn_embed_f32",
    "fc_paconv_create", "fc_paconv_destroy", "fc_paconv_out_dim", "fc_paconv_workspace_bytes", "fc_paconv_embed_f32", "fc_op_fps_f32",
    "fc_range_check_defer", "fc_range_check_resolve", "fc_range_check_pending",
    "fc_profile_enable", "fc_profile_reset", "fc_profile_filter", "fc_profile_stride", "fc_profile_report",
    "fc_op_linear_f32", "fc_op_mlp_hidden_f32", "fc_op_attention_f32", "fc_op_knn_f32", "fc_op_knn_warm_f32", "fc_op_rqspline_f32",
    "fc_stage_fps_f32", "fc_stage_co_unit_sphere_f32", "fc_clamp_infs_f32", "fc_change_map_f32",
    "fc_train_linear_pack_bytes", "fc_train_linear_pack_f32", "fc_train_linear_fwd_f32", "fc_train_linear_act_fwd_f32", "fc_train_linear_dgrad_f32", "fc_train_linear_dgrad_act_f32",
    "fc_train_linear_wgrad_ws_bytes", "fc_train_linear_wgrad_f32", "fc_train_act_fwd_f32", "fc_train_act_bwd_f32",
    "fc_train_attention_ws_bytes", "fc_train_attention_fwd_f32", "fc_train_attention_bwd_f32",
    "fc_train_rqspline_fwd_f32", "fc_train_rqspline_bwd_f32", "fc_train_layernorm_fwd_f32", "fc_train_layernorm_bwd_f32",
    "fc_train_colsum_ws_bytes", "fc_train_colsum_f32",
    "fc_train_affine_fwd_f32", "fc_train_affine_bwd_f32", "fc_train_gauss_fwd_f32", "fc_train_gauss_bwd_f32", "fc_train_base_fwd_f32", "fc_train_base_bwd_f32",
    "fc_train_normlp_fwd_f32", "fc_train_normlp_bwd_f32", "fc_train_expm_fwd_f32", "fc_train_expm_bwd_f32",
    "fc_train_edge_ws_bytes", "fc_train_edge_stats_f32", "fc_train_edge_fwd_f32", "fc_train_edge_bwd_prep_f32", "fc_train_edge_bwd_scatter_f32", "fc_train_edge_bwd_gather_f32", "fc_train_pool_fwd_f32", "fc_train_pool_bwd_f32",
    "fc_op_paconv_knn_f32", "fc_train_paconv_group_f32", "fc_train_softmax_fwd_f32", "fc_train_softmax_bwd_f32", "fc_train_assign_fwd_f32", "fc_train_assign_bwd_f32",
    "fc_train_centerdiff_fwd_f32", "fc_train_centerdiff_bwd_f32", "fc_train_rows_gather_bwd_f32", "fc_train_three_nn_f32", "fc_train_interp_fwd_f32",
    "fc_train_sqnorm_ws_bytes", "fc_train_sqnorm_f32", "fc_train_adam_f32",
]


class FcTensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p), ("ndim", ctypes.c_int32), ("shape", ctypes.c_int64 * 4)]


class FcFlowConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "struct_size", "input_dim", "latent_dim", "cif_latent_dim", "n_flow_layers", "flow_type", "affine_scale_fn",
        "permuter_type", "act_norm", "nonlinearity", "global_context", "extra_context_dim", "input_embedding_dim",
        "num_bins_spline", "expm_algo")] + [(n, ctypes.c_float) for n in ("linear_lu_eps", "eps_expm", "clamp_dist")]


class FcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libfcflow error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Loads libfcflow.so once; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -m flowcompare_amd.build` "
                               "(flowcompare_amd has no PyTorch/CPU fallback for the flow)")
        L = ctypes.CDLL(LIB_PATH)
        L.fc_last_error.restype = ctypes.c_char_p
        L.fc_flow_destroy.restype = None
        L.fc_dgcnn_destroy.restype = None
        L.fc_paconv_destroy.restype = None
        L.fc_train_linear_pack_bytes.restype = ctypes.c_size_t
        L.fc_train_linear_wgrad_ws_bytes.restype = ctypes.c_size_t
        L.fc_train_attention_ws_bytes.restype = ctypes.c_size_t
        L.fc_train_colsum_ws_bytes.restype = ctypes.c_size_t
        L.fc_train_edge_ws_bytes.restype = ctypes.c_size_t
        L.fc_train_sqnorm_ws_bytes.restype = ctypes.c_size_t
        if L.fc_abi_version() != ABI_VERSION:
            raise RuntimeError("libfcflow.so ABI version mismatch: rebuild with `python -m flowcompare_amd.build --force`")
        _lib = L
    return _lib


def _check(code):
    if code != 0:
        raise FcError(code, lib().fc_last_error().decode())


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f32(t, device=None):
    t = t.detach()
    if device is not None:
        t = t.to(device)
    if not t.is_cuda:
        raise RuntimeError("flowcompare_amd: tensors must live on a HIP device (there is no CPU path)")
    return t.to(torch.float32).contiguous()


def params_version(module):
    """Changes whenever a parameter/buffer is modified in place or replaced (engine re-pack trigger)."""
    return tuple((id(t), t._version) for t in list(module.parameters()) + list(module.buffers()))


def _tensor_table(state_dict):
    """state_dict -> (ctypes array of fc_tensor, keep-alive list): host fp32 copies, integer buffers as floats.  Device tensors are
    flattened into ONE buffer on their device and brought over in a single copy (a 115-layer flow has ~3000 tensors / 1.5 GB: one
    synchronising copy per tensor costs more than the bytes)."""
    keep, items = [], []
    names = list(state_dict.keys())
    host = {}
    dev_groups = {}
    for name in names:
        t = state_dict[name].detach()
        if t.dim() > 4:
            raise RuntimeError(f"state_dict entry {name} has more than 4 dims")
        if t.is_cuda:
            dev_groups.setdefault(t.device, []).append(name)
        else:
            host[name] = t.to(torch.float32).contiguous()
    for dev, group in dev_groups.items():
        flat = torch.cat([state_dict[n].detach().to(torch.float32).reshape(-1) for n in group]).cpu()
        off = 0
        for n in group:
            k = state_dict[n].numel()
            host[n] = flat[off:off + k].view(state_dict[n].shape)
            off += k
        keep.append(flat)
    for name in names:
        h = host[name]
        ft = FcTensor()
        nb = name.encode()
        ft.name = nb
        ft.data = h.data_ptr()
        ft.ndim = h.dim()
        for i, sz in enumerate(h.shape):
            ft.shape[i] = sz
        keep += [h, nb]
        items.append(ft)
    arr = (FcTensor * len(items))(*items)
    return arr, keep


# ---------------------------------------------------------------- deferred range check (include/fcflow.h)
_deferred_keep = []          # tensors handed to passes that are still pending (inputs, outputs, workspaces must outlive them)


class deferred_range_check:
    """Context manager: inside it the flow / embedder entry points only enqueue their split-fp16 pass -- no stream synchronisation per
    call, forwards queue back to back -- and the range flags are read when the block is left (or at `resolve()`).  `repeated` holds the
    number of passes that had to be repeated on the bf16-limb loops: when it is not 0, tensors derived from the passes' outputs inside
    the block (losses, ...) are stale and must be recomputed from the outputs, which the repeats rewrote in place."""

    def __init__(self):
        self.repeated = 0

    def __enter__(self):
        _check(lib().fc_range_check_defer(1))
        return self

    def resolve(self):
        n = ctypes.c_int32(0)
        _check(lib().fc_range_check_resolve(ctypes.byref(n)))
        del _deferred_keep[:]
        self.repeated += n.value
        return n.value

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                self.resolve()
        finally:
            lib().fc_range_check_defer(0)
            del _deferred_keep[:]
        return False


def _keep_if_deferred(*tensors):
    if lib().fc_range_check_pending() > 0:
        _deferred_keep.extend(t for t in tensors if t is not None)


class _Workspace:
    """Grow-only device scratch owned by a handle (the C ABI never allocates inside compute calls)."""
    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self.buf


class FlowHandle:
    """fc_flow wrapper: replaces the compute of models.Flow (reference models/transform.py:61-84)."""

    def __init__(self, config, state_dict, version, device):
        self.version = version
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("flowcompare_amd: the flow must be on a HIP device (`.to('cuda')`); there is no CPU path")
        c = FcFlowConfig()
        c.struct_size = ctypes.sizeof(FcFlowConfig)
        c.input_dim, c.latent_dim, c.cif_latent_dim = config["input_dim"], config["latent_dim"], config["cif_latent_dim"]
        c.n_flow_layers = config["n_flow_layers"]
        c.flow_type = FLOW_TYPES[config["flow_type"]]
        c.affine_scale_fn = SCALE_FNS[config["affine_scale_fn"]]
        c.permuter_type = PERMUTERS[config["permuter_type"]]
        c.act_norm = int(bool(config["act_norm"]))
        c.nonlinearity = ACTS[config["coupling_block_nonlinearity"]]
        c.global_context = int(bool(config["global"]))
        c.extra_context_dim = int(config["extra_context_dim"])
        c.input_embedding_dim = config["input_embedding_dim"]
        c.num_bins_spline = config["num_bins_spline"]
        c.expm_algo = EXPM[config["coupling_expm_algo"]]
        c.linear_lu_eps, c.eps_expm = float(config["linear_lu_eps"]), float(config["eps_expm"])
        c.clamp_dist = float(config["clamp_dist"] or 0.0)
        self.latent_dim, self.input_dim, self.X = c.latent_dim, c.input_dim, c.extra_context_dim
        arr, keep = _tensor_table(state_dict)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib().fc_flow_create(ctypes.byref(c), arr, len(arr), ctypes.byref(self._h)))
        del keep
        self._ws = _Workspace()
        self.n_noise = lib().fc_flow_noise_count(self._h)
        self.noise_width = [lib().fc_flow_noise_width(self._h, i) for i in range(self.n_noise)]

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.fc_flow_destroy(self._h)
            self._h = None

    def _prep(self, x, context, extra_context, eps):
        B, N = x.shape[0], x.shape[1]
        x = _dev_f32(x)
        ctx = _dev_f32(context)
        M = ctx.shape[1]
        extra = None
        if self.X:
            if extra_context is None:
                raise RuntimeError("this flow was built with extra context (extra_z_value_context) but extra_context is None")
            if extra_context.dim() == 3:                     # [B,n,X] as produced by inner_loop's repeat: constant over n
                if extra_context.shape[1] != N:
                    raise RuntimeError(f"Sizes of tensors must match: extra_context has {extra_context.shape[1]} points, x has {N} "
                                       "(extra_context is repeated to config['sample_size'], reference model_initialization.py:213)")
                extra_context = extra_context[:, 0, :]
            extra = _dev_f32(extra_context)
        eps = [_dev_f32(e) for e in (eps or [])]
        if len(eps) != self.n_noise:
            raise RuntimeError(f"flow needs {self.n_noise} noise tensors, got {len(eps)}")
        for e, wdt in zip(eps, self.noise_width):
            if tuple(e.shape) != (B, N, wdt):
                raise RuntimeError(f"noise tensor has shape {tuple(e.shape)}, expected {(B, N, wdt)}")
        return x, ctx, extra, eps, B, N, M

    def log_prob(self, x, context, extra_context, eps, return_latent=False):
        x, ctx, extra, eps, B, N, M = self._prep(x, context, extra_context, eps)
        L = lib()
        with torch.cuda.device(self.device):
            need = ctypes.c_size_t()
            _check(L.fc_flow_workspace_bytes(self._h, B, N, M, ctypes.byref(need)))
            ws = self._ws.get(need.value, self.device)
            out = torch.empty(B, N, dtype=torch.float32, device=self.device)
            z = torch.empty(B, N, self.latent_dim, dtype=torch.float32, device=self.device) if return_latent else None
            eps_arr = (ctypes.c_void_p * max(1, len(eps)))(*[e.data_ptr() for e in eps])
            _check(L.fc_flow_logprob_f32(self._h, _ptr(x), _ptr(ctx), _ptr(extra), eps_arr, len(eps), _ptr(out), _ptr(z),
                                         B, N, M, _ptr(ws), ctypes.c_size_t(ws.numel()), _stream()))
            _keep_if_deferred(x, ctx, extra, out, z, ws, *eps)
        return (out, z) if return_latent else out

    def inverse(self, z, context, extra_context, eps):
        """Inverse pass of Flow.sample from a drawn latent z [B,n,latent_dim] -> x [B,n,input_dim]."""
        B, N = z.shape[0], z.shape[1]
        n_inv = self.n_noise - (1 if self.latent_dim > self.input_dim else 0)        # one draw per CIF block (Slice.inverse)
        if eps is None:
            eps = [torch.randn(B, N, self.noise_width[-1], device=self.device) for _ in range(n_inv)]
        if len(eps) != n_inv:
            raise RuntimeError(f"inverse pass needs {n_inv} noise tensors, got {len(eps)}")
        z = _dev_f32(z)
        ctx = _dev_f32(context)
        if ctx.shape[0] != B:
            raise RuntimeError(f"context batch {ctx.shape[0]} != latent batch {B}")
        M = ctx.shape[1]
        extra = None
        if self.X:
            if extra_context is None:
                raise RuntimeError("this flow was built with extra context (extra_z_value_context) but extra_context is None")
            extra = _dev_f32(extra_context[:, 0, :] if extra_context.dim() == 3 else extra_context)
        eps = [_dev_f32(e) for e in eps]
        L = lib()
        with torch.cuda.device(self.device):
            need = ctypes.c_size_t()
            _check(L.fc_flow_workspace_bytes(self._h, B, N, M, ctypes.byref(need)))
            ws = self._ws.get(need.value, self.device)
            out = torch.empty(B, N, self.input_dim, dtype=torch.float32, device=self.device)
            eps_arr = (ctypes.c_void_p * max(1, len(eps)))(*[e.data_ptr() for e in eps])
            _check(L.fc_flow_inverse_f32(self._h, _ptr(z), _ptr(ctx), _ptr(extra), eps_arr, len(eps), _ptr(out), B, N, M,
                                         _ptr(ws), ctypes.c_size_t(ws.numel()), _stream()))
        return out


class DgcnnHandle:
    """fc_dgcnn wrapper: replaces models.DGCNNembedder / DGCNNembedderGlobal forward (reference models/pytorch_gcn.py:81-188)."""

    def __init__(self, n_neighbors, is_global, state_dict, version, device):
        self.version = version
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("flowcompare_amd: the embedder must be on a HIP device (`.to('cuda')`); there is no CPU path")
        self.is_global = bool(is_global)
        arr, keep = _tensor_table(state_dict)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib().fc_dgcnn_create(int(n_neighbors), int(self.is_global), arr, len(arr), ctypes.byref(self._h)))
        del keep
        self.out_dim = lib().fc_dgcnn_out_dim(self._h)
        self._ws = _Workspace()

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.fc_dgcnn_destroy(self._h)
            self._h = None

    def embed(self, pts):
        pts = _dev_f32(pts)
        B, M = pts.shape[0], pts.shape[1]
        L = lib()
        with torch.cuda.device(self.device):
            need = ctypes.c_size_t()
            _check(L.fc_dgcnn_workspace_bytes(self._h, B, M, ctypes.byref(need)))
            ws = self._ws.get(need.value, self.device)
            shape = (B, self.out_dim) if self.is_global else (B, M, self.out_dim)
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
            _check(L.fc_dgcnn_embed_f32(self._h, _ptr(pts), _ptr(out), B, M, _ptr(ws), ctypes.c_size_t(ws.numel()), _stream()))
            _keep_if_deferred(pts, out, ws)
        return out


class PaconvHandle:
    """fc_paconv wrapper: replaces models.PointNet2SSGSeg.forward (reference pointnet2_paconv_seg.py:63-82)."""

    def __init__(self, state_dict, version, device):
        self.version = version
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("flowcompare_amd: the embedder must be on a HIP device (`.to('cuda')`); there is no CPU path")
        arr, keep = _tensor_table(state_dict)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib().fc_paconv_create(arr, len(arr), ctypes.byref(self._h)))
        del keep
        self.out_dim = lib().fc_paconv_out_dim(self._h)
        self._ws = _Workspace()

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.fc_paconv_destroy(self._h)
            self._h = None

    def embed(self, pts):
        pts = _dev_f32(pts)
        B, M = pts.shape[0], pts.shape[1]
        L = lib()
        with torch.cuda.device(self.device):
            need = ctypes.c_size_t()
            _check(L.fc_paconv_workspace_bytes(self._h, B, M, ctypes.byref(need)))
            ws = self._ws.get(need.value, self.device)
            out = torch.empty(B, M, self.out_dim, dtype=torch.float32, device=self.device)
            _check(L.fc_paconv_embed_f32(self._h, _ptr(pts), _ptr(out), B, M, _ptr(ws), ctypes.c_size_t(ws.numel()), _stream()))
            _keep_if_deferred(pts, out, ws)
        return out


# ---------------------------------------------------------------- single operators (unit-level parity tests)
def op_fps(xyz, m):
    xyz = _dev_f32(xyz)
    B, n, _ = xyz.shape
    idx = torch.empty(B, m, dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _check(lib().fc_op_fps_f32(_ptr(xyz), _ptr(idx), B, n, m, _stream()))
    return idx


def stage_fps(pts, m, n_coord=None):
    """pts [B,n,C] -> idx [B,m] int64 (fc_stage_fps_f32: torch_cluster.fps semantics, random_start=False)."""
    pts = _dev_f32(pts)
    B, n, ld = pts.shape
    idx = torch.empty(B, m, dtype=torch.int64, device=pts.device)
    with torch.cuda.device(pts.device):
        _check(lib().fc_stage_fps_f32(_ptr(pts), ld, ld if n_coord is None else n_coord, _ptr(idx), B, n, m, _stream()))
    return idx


def stage_co_unit_sphere(p0, p1):
    """p0 [B,n0,C], p1 [B,n1,C] -> (out0, out1, inverse [B,4] = furthest distance, mean xyz) (fc_stage_co_unit_sphere_f32)."""
    p0, p1 = _dev_f32(p0), _dev_f32(p1)
    B, n0, ld = p0.shape
    if p1.shape[0] != B or p1.shape[2] != ld:
        raise RuntimeError(f"co_unit_sphere: clouds of shape {tuple(p0.shape)} and {tuple(p1.shape)} do not pair up")
    o0, o1 = torch.empty_like(p0), torch.empty_like(p1)
    inv = torch.empty(B, 4, dtype=torch.float32, device=p0.device)
    with torch.cuda.device(p0.device):
        _check(lib().fc_stage_co_unit_sphere_f32(_ptr(p0), n0, _ptr(p1), p1.shape[1], ld, _ptr(o0), _ptr(o1), _ptr(inv), B, _stream()))
    return o0, o1, inv


def clamp_infs(t):
    """fc_clamp_infs_f32 in place on a contiguous fp32 device tensor."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError("clamp_infs: expects a contiguous float32 tensor on the GPU (flowcompare_amd has no CPU fallback)")
    with torch.cuda.device(t.device):
        _check(lib().fc_clamp_infs_f32(_ptr(t), ctypes.c_int64(t.numel()), _stream()))
    return t


def change_map(lp10, lp00, multiple, hard_cutoff=None):
    """fc_change_map_f32 on contiguous fp32 [B,N] / [B,N0] device tensors (clamped in place); returns (out, invalid flag)."""
    for t in (lp10, lp00):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 2):
            raise RuntimeError("change_map: expects contiguous float32 [B, N] tensors on the GPU (flowcompare_amd has no CPU fallback)")
    B, N = lp10.shape
    if lp00.shape[0] != B:
        raise RuntimeError("change_map: batch sizes differ")
    out = torch.empty_like(lp10)
    bad = ctypes.c_int32(0)
    with torch.cuda.device(lp10.device):
        _check(lib().fc_change_map_f32(_ptr(lp10), N, _ptr(lp00), lp00.shape[1], _ptr(out), B, ctypes.c_float(multiple),
                                       ctypes.c_float(0.0 if hard_cutoff is None else hard_cutoff), 0 if hard_cutoff is None else 1,
                                       ctypes.byref(bad), _stream()))
    return out, bool(bad.value)


def op_linear(x, W, bias=None, residual=None, act="none"):
    code = {"none": 0, "gelu": 1, "relu": 2, "elu": 3, "lrelu": 4}[act]
    x, W = _dev_f32(x), _dev_f32(W)
    rows, K = x.shape
    N = W.shape[0]
    y = torch.empty(rows, N, dtype=torch.float32, device=x.device)
    b = _dev_f32(bias) if bias is not None else None
    r = _dev_f32(residual) if residual is not None else None
    with torch.cuda.device(x.device):
        _check(lib().fc_op_linear_f32(_ptr(x), _ptr(W), _ptr(b), _ptr(r), _ptr(y), rows, N, K, code, _stream()))
    return y


def op_mlp_hidden(x0, x1, state_dict, rowscal=None, act="gelu", use_rows=True):
    """Last hidden activation [rows, 512] of a 512-wide reference MLP over cat(x0, x1) (fc_op_mlp_hidden_f32); `state_dict` holds
    net.in_layer / net.layers.<i> / net.out_layer tensors (host or device; copied to the host), optionally net.colvec.
    use_rows: True = the row-resident chain kernel, False = one launch per layer, "wide" = hidden layers on the 256 x 256 one-accumulator kernel."""
    code = {"none": 0, "gelu": 1, "relu": 2, "elu": 3, "lrelu": 4}[act]
    x0 = _dev_f32(x0)
    x1 = _dev_f32(x1) if x1 is not None else None
    rs = _dev_f32(rowscal) if rowscal is not None else None
    rows = x0.shape[0]
    out = torch.empty(rows, 512, dtype=torch.float32, device=x0.device)
    arr, keep = _tensor_table({k: v.detach().cpu() for k, v in state_dict.items()})
    with torch.cuda.device(x0.device):
        _check(lib().fc_op_mlp_hidden_f32(_ptr(x0), x0.shape[1], _ptr(x1), x1.shape[1] if x1 is not None else 0, _ptr(rs), arr, len(arr),
                                          _ptr(out), rows, code, 2 if use_rows == "wide" else int(bool(use_rows)), _stream()))
    del keep
    return out


def op_attention(q, k, v, scale):
    q, k, v = _dev_f32(q), _dev_f32(k), _dev_f32(v)
    B, N, D = q.shape
    M = k.shape[1]
    out = torch.empty_like(q)
    with torch.cuda.device(q.device):
        _check(lib().fc_op_attention_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, N, M, D, ctypes.c_float(scale), _stream()))
    return out


def op_knn(f, k, warm=None):
    """k nearest neighbours in feature space [B, M, C] -> int32 [B, M, k] (unordered sets); `warm`: neighbour sets [B, M, k] of the same cloud from
    another feature space to start the search from (the result is the exact top-k either way)."""
    f = _dev_f32(f)
    B, M, C = f.shape
    idx = torch.empty(B, M, k, dtype=torch.int32, device=f.device)
    with torch.cuda.device(f.device):
        if warm is None:
            _check(lib().fc_op_knn_f32(_ptr(f), _ptr(idx), B, M, C, k, _stream()))
        else:
            warm = warm.to(device=f.device, dtype=torch.int32).contiguous()
            if tuple(warm.shape) != (B, M, k):
                raise RuntimeError(f"op_knn: warm sets have shape {tuple(warm.shape)}, expected {(B, M, k)}")
            _check(lib().fc_op_knn_warm_f32(_ptr(f), _ptr(warm), _ptr(idx), B, M, C, k, _stream()))
    return idx


def op_rqspline(x, params, num_bins, inverse=False):
    x, params = _dev_f32(x), _dev_f32(params)
    n = x.numel()
    assert params.numel() == n * (3 * num_bins + 1)
    y, lad = torch.empty_like(x), torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(lib().fc_op_rqspline_f32(_ptr(x), _ptr(params), _ptr(y), _ptr(lad), ctypes.c_int64(n), num_bins, int(inverse), _stream()))
    return y, lad


# ---------------------------------------------------------------- in-library kernel timing
def profile_enable(on=True):
    _check(lib().fc_profile_enable(int(bool(on))))


def profile_filter(kernel_substr=None):
    """Bracket only launches whose kernel name contains `kernel_substr` (None = all)."""
    _check(lib().fc_profile_filter(kernel_substr.encode() if kernel_substr else None))


def profile_stride(n=1):
    """Of the launches that pass the filter bracket every `n`-th one only (1 = all); the report then counts the bracketed launches."""
    _check(lib().fc_profile_stride(int(n)))


def profile_reset():
    _check(lib().fc_profile_reset())


def profile_report():
    """[{kernel, launches, ms, flops, bytes}] accumulated since the last reset (HIP events on the launch stream)."""
    import json
    buf = ctypes.create_string_buffer(1 << 16)
    _check(lib().fc_profile_report(buf, ctypes.c_size_t(len(buf))))
    return json.loads(buf.value.decode())
