"""Host-side mirror of the reference's model classes for the forward log-prob path.

These nn.Modules are PARAMETER CONTAINERS whose attribute names reproduce the
reference's checkpoint layout, so `state_dict()` /
`load_state_dict()` exchange checkpoints with the reference unchanged.  They do
not compute anything in PyTorch: `Flow.log_prob`, `Flow.sample` and the
embedders' `forward` hand raw device pointers to the HIP engine
(libfcflow.so, include/fcflow.h) and fail loudly if it is missing.

Reference classes mirrored (file:line in the reference repo):
  MLP                              models/nets.py:6-30
  PreNorm / AttentionControlledOut / AttentionMine   models/perceiver.py:18-35, 89-119
  CouplingPreconditionerAttn/Global, CIFblock        models/cif_block.py:6-112
  PreConditionApplier / Flow / IdentityTransform     models/transform.py:39-92
  AffineCoupling                   models/affine_coupling.py:8-28
  RationalQuadraticSplineCoupling  models/spline_coupling.py:172-185
  ExponentialCoupling              models/exponential_coupling.py:20-33
  ActNormBijectionCloud            models/act_norm.py:9-88
  LinearLU / Permuter / Reverse / FullCombiner / ExponentialCombiner   models/permuters.py:15-198
  ConditionalNormal / StandardNormal / Normal        models/distributions.py:120-219
  Augment / AugmentAttentionPreconditioner           models/augmenter.py:7-67
  Slice                            models/slice.py:7-58
  DGCNNembedder / DGCNNembedderGlobal                models/pytorch_gcn.py:50-188
  PointNet2SSGSeg (PAConv embedder)                  models/scene_seg_PAConv/model/pointnet2/pointnet2_paconv_seg.py:14-82
    PointNet2SAModule / PointNet2FPModule            .../pointnet2_paconv_modules.py:64-238
    PAConv / ScoreNet / SharedPAConv                 .../paconv.py:12-260
    SharedMLP / Conv2d                               models/scene_seg_PAConv/util/block.py:14-150
"""
import math

import numpy as np
import torch
from torch import nn

from . import engine as _engine


def _no_torch_forward(self, *a, **k):
    raise RuntimeError(
        f"{type(self).__name__} is a parameter container: the computation runs in the HIP engine through "
        "Flow.log_prob / Flow.sample / the embedder's forward (flowcompare_amd has no PyTorch fallback)")


class _Container(nn.Module):
    forward = _no_torch_forward


# ------------------------------------------------------------------ nets
class MLP(_Container):
    def __init__(self, in_dim, sizes, out_dim, nonlin=None, residual=True):
        super().__init__()
        self.in_dim, self.sizes, self.out_dim = in_dim, list(sizes), out_dim
        self.nonlin, self.residual = nonlin, residual
        self.in_layer = nn.Linear(in_dim, self.sizes[0])
        self.out_layer = nn.Linear(self.sizes[-1], out_dim)
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(self.sizes[:-1], self.sizes[1:]))


# ------------------------------------------------------------------ cross attention
class AttentionMine(_Container):
    def __init__(self, query_dim, context_dim, heads, dim_head):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.scale = self.inner_dim ** -0.5
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=False)
        self.to_kv = nn.Linear(context_dim, self.inner_dim * 2, bias=False)


class AttentionControlledOut(_Container):
    def __init__(self, out_dim, query_dim, context_dim, heads, dim_head, dropout):
        super().__init__()
        self.attention = AttentionMine(query_dim, context_dim, heads, dim_head)
        self.lin = nn.Linear(self.attention.inner_dim, out_dim)


class PreNorm(_Container):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)


def get_cross_attn(out_dim, query_dim, context_dim, heads, dim_head, dropout):
    return PreNorm(query_dim, AttentionControlledOut(out_dim, query_dim, context_dim, heads, dim_head, dropout))


# ------------------------------------------------------------------ couplings
class AffineCoupling(_Container):
    def __init__(self, input_dim, hidden_dims, nonlinearity=None, context_dim=0, scale_fn_type="exp", split_dim=None):
        super().__init__()
        if scale_fn_type not in ("exp", "sigmoid"):
            raise Exception("Invalid scale_fn_type")
        self.input_dim, self.context_dim, self.scale_fn_type = input_dim, context_dim, scale_fn_type
        self.split_dim = input_dim // 2 if split_dim is None else split_dim
        self.nn = MLP(self.split_dim + context_dim, hidden_dims, (input_dim - self.split_dim) * 2, nonlinearity)


class RationalQuadraticSplineCoupling(_Container):
    def __init__(self, input_dim, hidden_dims, nonlinearity, num_bins, context_dim=0):
        super().__init__()
        self.input_dim, self.context_dim, self.num_bins = input_dim, context_dim, num_bins
        self.split_dim = input_dim // 2
        self.nn = MLP(self.split_dim + context_dim, hidden_dims, (3 * num_bins + 1) * self.split_dim, nonlinearity)


class ExponentialCoupling(_Container):
    def __init__(self, input_dim, hidden_dims, nonlinearity, context_dim=0, algo="original", eps_expm=1e-8):
        super().__init__()
        self.input_dim, self.context_dim, self.algo, self.eps_expm = input_dim, context_dim, algo, eps_expm
        self.scale = nn.Parameter(torch.ones(1) / 8)
        self.shift = nn.Parameter(torch.zeros(1))
        self.rescale = nn.Parameter(torch.ones(1))
        self.reshift = nn.Parameter(torch.zeros(1))
        self.split_dim = input_dim // 2
        d2 = input_dim - self.split_dim
        self.nn = MLP(self.split_dim + context_dim, hidden_dims, d2 * d2 + d2, nonlinearity)


class CouplingPreconditionerAttn(_Container):
    def __init__(self, attn, pre_attention_mlp, x1_dim):
        super().__init__()
        self.attn = attn
        self.pre_attention_mlp = pre_attention_mlp
        self.x1_dim = x1_dim


class CouplingPreconditionerGlobal(_Container):
    pass


class PreConditionApplier(_Container):
    def __init__(self, transform, pre_conditioner):
        super().__init__()
        self.pre_conditioner = pre_conditioner
        self.transform = transform


class IdentityTransform(_Container):
    pass


# ------------------------------------------------------------------ ActNorm / permuters
class ActNormBijectionCloud(_Container):
    def __init__(self, num_features, data_dep_init=True, eps=1e-6):
        super().__init__()
        self.num_features, self.data_dep_init, self.eps = num_features, data_dep_init, eps
        self.register_buffer("initialized", torch.zeros(1) if data_dep_init else torch.ones(1))
        self.shift = nn.Parameter(torch.zeros(1, num_features))
        self.log_scale = nn.Parameter(torch.zeros(1, num_features))


class LinearLU(_Container):
    def __init__(self, num_features, eps=1e-3):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        n_tri = (num_features - 1) * num_features // 2
        self.lower_entries = nn.Parameter(torch.zeros(n_tri))
        self.upper_entries = nn.Parameter(torch.zeros(n_tri))
        # identity init: softplus(c) + eps == 1  (permuters.py:136-140)
        self.unconstrained_upper_diag = nn.Parameter(torch.full((num_features,), float(np.log(np.exp(1 - eps) - 1))))


class Permuter(_Container):
    def __init__(self, permutation):
        super().__init__()
        self.register_buffer("permutation", permutation)
        self.register_buffer("inv_permutation", torch.argsort(permutation))


class Reverse(Permuter):
    def __init__(self, dim_size):
        super().__init__(torch.arange(dim_size - 1, -1, -1))


class FullCombiner(_Container):
    def __init__(self, dim):
        super().__init__()
        self.w = nn.Parameter(torch.empty(dim, dim))
        nn.init.orthogonal_(self.w)


class ExponentialCombiner(_Container):
    def __init__(self, dim, eps_expm=1e-8):
        super().__init__()
        self.eps_expm = eps_expm
        self.w = nn.Parameter(torch.randn(dim, dim))
        self.scale = nn.Parameter(torch.ones(1) / 8)
        self.shift = nn.Parameter(torch.zeros(1))
        self.rescale = nn.Parameter(torch.ones(1))
        self.reshift = nn.Parameter(torch.zeros(1))


# ------------------------------------------------------------------ distributions / augment / slice
class StandardNormal(_Container):
    def __init__(self, shape):
        super().__init__()
        self.shape = torch.Size(shape)
        self.register_buffer("buffer", torch.zeros(1))

    def sample(self, num_samples, context=None, n_points=None):
        shp = list(self.shape)
        shp[-2] = n_points
        return torch.randn(num_samples, *shp, device=self.buffer.device, dtype=self.buffer.dtype)


class Normal(_Container):
    def __init__(self, loc, scale, shape):
        super().__init__()
        self.std_normal = StandardNormal(shape)
        self.shape = torch.Size(shape)
        self.register_buffer("loc", loc)
        self.register_buffer("scale", scale)

    def sample(self, num_samples, context=None, n_points=None):
        return self.std_normal.sample(num_samples, n_points=n_points) * self.scale + self.loc


class ConditionalNormal(_Container):
    def __init__(self, net, clamp=False):
        super().__init__()
        self.net = net
        self.clamp = clamp


class Augment(_Container):
    def __init__(self, noise_dist, x_size, use_context=True):
        super().__init__()
        self.noise_dist = noise_dist
        self.x_size, self.use_context = x_size, use_context


class AugmentAttentionPreconditioner(_Container):
    def __init__(self, augment, attn, pre_attn_mlp):
        super().__init__()
        self.augment = augment
        self.attn = attn()
        self.pre_attn_mlp = pre_attn_mlp


class Slice(_Container):
    def __init__(self, noise_dist, num_keep):
        super().__init__()
        self.noise_dist = noise_dist
        self.num_keep = num_keep


class CIFblock(_Container):
    """models/cif_block.py:49-69; augmenter and slicer share ONE ConditionalNormal (its weights appear
    under both prefixes in the state_dict, as in the reference)."""
    def __init__(self, config, flow, attn):
        super().__init__()
        D, Dc = config["latent_dim"], config["cif_latent_dim"]
        net = MLP(D, config["net_cif_dist_hidden_dims"], (Dc - D) * 2)
        dist = ConditionalNormal(net, clamp=config["clamp_dist"])
        self.act_norm = ActNormBijectionCloud(Dc)
        self.augmenter = Augment(dist, D)
        pre = MLP(D // 2, config["pre_attention_mlp_hidden_dims"], config["attn_input_dim"])
        self.affine_cif = AffineCoupling(Dc, config["affine_cif_hidden"], scale_fn_type="sigmoid", split_dim=Dc - D)
        self.flow = PreConditionApplier(flow(D, config["attn_dim"]), CouplingPreconditionerAttn(attn(), pre, D // 2))
        self.slicer = Slice(dist, D)
        self.reverse = Reverse(Dc)


def cif_helper(config, flow, attn, pre_attention_mlp):
    """models/cif_block.py:30-46 (same exceptions)."""
    D, Dc = config["latent_dim"], config["cif_latent_dim"]
    if D < Dc:
        if config["using_extra_context"]:
            raise Exception("Not implemented extra context with cif")
        if config["global"]:
            raise Exception("CIF + global embedding not implemented")
        return CIFblock(config, flow, attn)
    if D == Dc:
        if not config["global"]:
            return PreConditionApplier(flow(D, config["attn_dim"] + config["extra_context_dim"]),
                                       CouplingPreconditionerAttn(attn(), pre_attention_mlp(D // 2), D // 2))
        return PreConditionApplier(flow(D, config["input_embedding_dim"] + config["extra_context_dim"]),
                                   CouplingPreconditionerGlobal())
    raise Exception("Augment dim smaller than main latent!")


# ------------------------------------------------------------------ the flow
class Flow(nn.Module):
    """models/transform.py:61-84.  log_prob / sample run in the HIP engine."""

    def __init__(self, transform_list, base_dist, sample_dist=None, config=None):
        super().__init__()
        self.base_dist = base_dist
        self.sample_dist = sample_dist if sample_dist is not None else base_dist
        self.transforms = nn.ModuleList(transform_list)
        self._config = dict(config or {})
        self._handle = None
        self.last_eps = None
        self._inverse_eps = None      # optional explicit noise for the CIF Slice.inverse draws of the next sample() call

    # -- engine plumbing
    def _engine(self):
        key = _engine.params_version(self)
        if self._handle is None or self._handle.version != key:
            self._handle = _engine.FlowHandle(self._config, self.state_dict(), key, next(self.parameters()).device)
        return self._handle

    def noise_shapes(self, B, N):
        """Noise tensors one forward consumes, in draw order."""
        c = self._config
        shapes = []
        if c["latent_dim"] > c["input_dim"]:
            shapes.append((B, N, c["latent_dim"] - c["input_dim"]))
        if c["latent_dim"] < c["cif_latent_dim"]:
            shapes += [(B, N, c["cif_latent_dim"] - c["latent_dim"])] * c["n_flow_layers"]
        return shapes

    def log_prob(self, x, context=None, extra_context=None, eps=None):
        """x [B,N,input_dim]; context [B,M,E] ([B,N,E] for the global embedder); extra_context [B,N,X]
        as produced by inner_loop (constant over N) or None.  `eps`: optional list of explicit noise
        tensors (shapes: noise_shapes) making the stochastic forward reproducible; drawn with
        torch.randn when omitted, like the reference's rsample()."""
        B, N = x.shape[0], x.shape[1]
        if eps is None:
            eps = [torch.randn(s, device=x.device, dtype=torch.float32) for s in self.noise_shapes(B, N)]
        self.last_eps = eps
        if torch.is_grad_enabled() and (self.training or x.requires_grad or (context is not None and context.requires_grad)):
            # training (train.py:108-112): the differentiable path over the HIP training primitives (train_flow.py); the fused
            # inference engine below has no backward
            from . import train_flow
            return train_flow.flow_log_prob(self, x, context, extra_context, eps)
        from . import library_ops                          # the inference call as a torch.library op (torch.ops.flowcompare_amd.flow_log_prob)
        return torch.ops.flowcompare_amd.flow_log_prob(x, context, extra_context, list(eps), library_ops.register(self._engine()))

    def sample(self, num_samples, n_points, context=None, sample_distrib=None, extra_context=None, eps=None):
        dist = sample_distrib if sample_distrib is not None else self.sample_dist
        z = dist.sample(num_samples, n_points=n_points)
        if eps is None:
            eps, self._inverse_eps = self._inverse_eps, None
        return self._engine().inverse(z, context, extra_context, eps)

    forward = _no_torch_forward


# ------------------------------------------------------------------ embedders
class _DGCNNBase(nn.Module):
    def _build_trunk(self, in_ch):
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm2d(64), nn.BatchNorm2d(64), nn.BatchNorm2d(128)
        self.bn4, self.bn5 = nn.BatchNorm2d(256), nn.BatchNorm1d(512)
        act = lambda: nn.LeakyReLU(negative_slope=0.2)
        self.conv1 = nn.Sequential(nn.Conv2d(in_ch * 2, 64, kernel_size=1, bias=False), self.bn1, act())
        self.conv2 = nn.Sequential(nn.Conv2d(128, 64, kernel_size=1, bias=False), self.bn2, act())
        self.conv3 = nn.Sequential(nn.Conv2d(128, 128, kernel_size=1, bias=False), self.bn3, act())
        self.conv4 = nn.Sequential(nn.Conv2d(256, 256, kernel_size=1, bias=False), self.bn4, act())
        self.conv5 = nn.Sequential(nn.Conv1d(512, 512, kernel_size=1, bias=False), self.bn5, act())
        self._handle = None

    def _engine(self):
        if self.training:
            raise RuntimeError("flowcompare_amd: the HIP DGCNN embedder implements eval-mode BatchNorm only "
                               "(forward log-prob path); call .eval() / initialize_flow(mode='test')")
        key = _engine.params_version(self)
        if self._handle is None or self._handle.version != key:
            self._handle = _engine.DgcnnHandle(self.n_neighbors, self.is_global, self.state_dict(), key,
                                               next(self.parameters()).device)
        return self._handle

    def forward(self, x):
        """x [B,M,6] -> [B,M,E] (per-point) or [B,E] (global).  In train() mode: the differentiable HIP path with BatchNorm batch
        statistics (train_embed.py); in eval() mode the fused inference engine (running statistics)."""
        if self.training:                                  # (also under no_grad: train-mode BatchNorm normalises with batch statistics)
            from . import train_embed
            return train_embed.dgcnn_embed(self, x)
        from . import library_ops
        return torch.ops.flowcompare_amd.context_embed(x, library_ops.register(self._engine()))


class DGCNNembedder(_DGCNNBase):
    is_global = False

    def __init__(self, out_mlp_dims, emb_dim=22, dropout=0, n_neighbors=20):
        super().__init__()
        self.n_neighbors = n_neighbors
        self._build_trunk(6)
        self.out_mlp = MLP(512, out_mlp_dims, emb_dim)


class DGCNNembedderGlobal(_DGCNNBase):
    is_global = True

    def __init__(self, input_dim, out_mlp_dims, emb_dim=22, n_neighbors=20):
        super().__init__()
        self.n_neighbors, self.input_dim = n_neighbors, input_dim
        self._build_trunk(input_dim)
        self.out_mlp = MLP(1024, out_mlp_dims, emb_dim)


# ------------------------------------------------------------------ PAConv embedder (PointNet++ SSG U-Net)
class ScoreNet(_Container):
    def __init__(self, in_channel, out_channel, hidden_unit):
        super().__init__()
        dims = [in_channel] + list(hidden_unit) + [out_channel]
        self.mlp_convs_hidden = nn.ModuleList()
        self.mlp_bns_hidden = nn.ModuleList()
        for i in range(1, len(dims)):
            conv = nn.Conv2d(dims[i - 1], dims[i], 1, bias=i == len(dims) - 1)      # last_bn False -> only the last conv has a bias
            nn.init.xavier_normal_(conv.weight)
            if conv.bias is not None:
                nn.init.constant_(conv.bias, 0)
            self.mlp_convs_hidden.append(conv)
            self.mlp_bns_hidden.append(nn.BatchNorm2d(dims[i]))


class PAConv(_Container):
    def __init__(self, input_dim, output_dim, m=8, hidden=(16,)):
        super().__init__()
        self.input_dim, self.output_dim, self.m = input_dim, output_dim, m
        self.bn = nn.BatchNorm2d(output_dim, momentum=0.1)
        self.scorenet = ScoreNet(3, m, list(hidden))
        bank = nn.init.kaiming_normal_(torch.empty(m, input_dim * 2, output_dim))
        self.weightbank = nn.Parameter(bank.permute(1, 0, 2).reshape(input_dim * 2, m * output_dim).contiguous())


class _Grouper(_Container):
    pass


class PointNet2SAModule(_Container):
    def __init__(self, mlp, nsample=32):
        super().__init__()
        self.nsample = nsample
        self.groupers = nn.ModuleList([_Grouper()])
        spec = list(mlp)
        spec[0] += 3                                                    # use_xyz
        self.mlps = nn.ModuleList([nn.Sequential()])
        for i in range(len(spec) - 1):
            self.mlps[0].add_module(f"layer{i}", PAConv(spec[i], spec[i + 1]))


class _BN2dWrap(nn.Sequential):
    def __init__(self, c):
        super().__init__()
        self.add_module("bn", nn.BatchNorm2d(c))


class _ConvBNReLU(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        conv = nn.Conv2d(cin, cout, kernel_size=(1, 1), bias=False)
        nn.init.kaiming_normal_(conv.weight)
        self.add_module("conv", conv)
        self.add_module("bn", _BN2dWrap(cout))
        self.add_module("activation", nn.ReLU(inplace=True))


class PointNet2FPModule(_Container):
    def __init__(self, mlp):
        super().__init__()
        self.mlp = nn.Sequential()
        for i in range(len(mlp) - 1):
            self.mlp.add_module(f"layer{i}", _ConvBNReLU(mlp[i], mlp[i + 1]))


class PointNet2SSGSeg(nn.Module):
    """PAConv context embedder: 4 set-abstraction levels (FPS to n/4, 32-NN grouping, 3 PAConv layers, max) + 4 feature
    propagation levels (3-NN inverse-distance interpolation, skip concat, shared MLP) + head MLP."""
    TRAINABLE = True       # train() mode runs the differentiable HIP path (train_paconv.py); False would keep it frozen in eval() mode

    def __init__(self, c=3, k=13, use_xyz=True, out_mlp_dims=(512, 512, 512), args=None):
        super().__init__()
        sa = [[c, 32, 32, 64], [64, 64, 64, 128], [128, 128, 128, 256], [256, 256, 256, 512]]
        fp = [[128 + c, 128, 128, 128], [256 + 64, 256, 128], [256 + 128, 256, 256], [512 + 256, 256, 256]]
        self.SA_modules = nn.ModuleList(PointNet2SAModule(m) for m in sa)
        self.FP_modules = nn.ModuleList(PointNet2FPModule(m) for m in fp)
        self.out_mlp = MLP(128, list(out_mlp_dims), k)
        self._handle = None

    def _engine(self):
        if self.training:
            raise RuntimeError("flowcompare_amd: the HIP PAConv embedder implements eval-mode BatchNorm only; call .eval()")
        key = _engine.params_version(self)
        if self._handle is None or self._handle.version != key:
            self._handle = _engine.PaconvHandle(self.state_dict(), key, next(self.parameters()).device)
        return self._handle

    def forward(self, pointcloud):
        """pointcloud [B,M,3+c] (xyz first) -> [B,M,k].  In train() mode: the differentiable HIP path with BatchNorm batch statistics
        (train_paconv.py); in eval() mode the fused inference engine (running statistics)."""
        if self.training:
            from . import train_paconv
            return train_paconv.paconv_embed(self, pointcloud)
        from . import library_ops
        return torch.ops.flowcompare_amd.context_embed(pointcloud, library_ops.register(self._engine()))
