"""Drop-in counterpart of the reference's model_initialization.py for the forward log-prob path.

Same public functions, arguments, return values, derived-config side effects and
exceptions as the reference (model_initialization.py:18-245):

    initialize_flow(config, device='cuda', mode='train') -> {'parameters','flow','input_embedder'}
    inner_loop(batch, models_dict, config)               -> (loss, log_prob[B,N], bpd)
    make_sample(n_points, extract_0, models_dict, config, sample_distrib=None, extra_context=None)
    save_flow(model_dict, config, optimizer, scheduler, save_path) / load_flow(load_dict, models_dict)

The modules are parameter containers with the reference's checkpoint names
(flowcompare_amd/modules.py); all arithmetic happens in the HIP engine.
"""
import math

import torch
from torch import nn

from . import modules as M


def load_flow(load_dict, models_dict):
    """model_initialization.py:18-23."""
    models_dict["input_embedder"].load_state_dict(load_dict["input_embedder"])
    models_dict["flow"].load_state_dict(load_dict["flow"])
    return models_dict


def save_flow(model_dict, config, optimizer, scheduler, save_path):
    """model_initialization.py:25-28 (config may be a wandb Config with ._items or a plain dict)."""
    save_dict = {"config": getattr(config, "_items", config),
                 "optimizer": optimizer.state_dict() if optimizer is not None else None,
                 "flow": model_dict["flow"].state_dict(),
                 "input_embedder": model_dict["input_embedder"].state_dict(),
                 "scheduler": scheduler.state_dict() if scheduler is not None else None}
    torch.save(save_dict, save_path)


def initialize_flow(config, device="cuda", mode="train"):
    """model_initialization.py:30-202."""
    X = 1 if config["extra_z_value_context"] else 0
    config["extra_context_dim"] = X
    config["using_extra_context"] = X > 0
    config["global"] = config["input_embedder"] in ["DGCNNembedderGlobal"]

    if config["coupling_block_nonlinearity"] not in ("ELU", "RELU", "GELU"):
        raise Exception("Invalid coupling_block_nonlinearity")
    act = config["coupling_block_nonlinearity"]

    def attn():
        return M.get_cross_attn(config["attn_dim"], config["attn_input_dim"], config["input_embedding_dim"],
                                config["cross_heads"], config["cross_dim_head"], config["attn_dropout"])

    D, Din = config["latent_dim"], config["input_dim"]
    if D > Din:
        if config["augmenter_dist"] == "ConditionalNormal" and config["use_attn_augment"]:
            net = M.MLP(config["attn_dim"] + Din + X, config["net_augmenter_dist_hidden_dims"], (D - Din) * 2, act)
            aug = M.Augment(M.ConditionalNormal(net), x_size=Din, use_context=True)
            augmenter = M.AugmentAttentionPreconditioner(aug, attn, M.MLP(Din, config["hidden_dims"], config["attn_input_dim"], act))
        elif config["augmenter_dist"] in ("ConditionalNormal", "StandardNormal"):
            # The reference builds a bare Augment here whose forward() cannot take the extra_context
            # keyword Flow.log_prob passes: it is not a working configuration there either.
            raise Exception("augmenter without use_attn_augment is not a working reference configuration (Augment.forward "
                            "rejects extra_context); use use_attn_augment: true or latent_dim == input_dim")
        else:
            raise Exception("Invalid augmenter_dist")
    elif D == Din:
        augmenter = M.IdentityTransform()
    else:
        raise Exception("Latent dim < Input dim")

    if config["flow_type"] == "AffineCoupling":
        def flow_for_cif(input_dim, context_dim):
            return M.AffineCoupling(input_dim, config["hidden_dims"], act, context_dim=context_dim,
                                    scale_fn_type=config["affine_scale_fn"])
    elif config["flow_type"] == "ExponentialCoupling":
        def flow_for_cif(input_dim, context_dim):
            return M.ExponentialCoupling(input_dim, config["hidden_dims"], act, context_dim=context_dim,
                                         eps_expm=config["eps_expm"], algo=config["coupling_expm_algo"])
    elif config["flow_type"] == "RationalQuadraticSplineCoupling":
        def flow_for_cif(input_dim, context_dim):
            return M.RationalQuadraticSplineCoupling(input_dim, config["hidden_dims"], act, config["num_bins_spline"],
                                                     context_dim=context_dim)
    else:
        raise Exception("Invalid flow type")

    def pre_attention_mlp(in_dim):
        return M.MLP(in_dim, config["pre_attention_mlp_hidden_dims"], config["attn_input_dim"], act, residual=True)

    ptype = config["permuter_type"]
    if ptype == "ExponentialCombiner":
        permuter = lambda dim: M.ExponentialCombiner(dim, eps_expm=config["eps_expm"])
    elif ptype == "random_permute":
        permuter = lambda dim: M.Permuter(torch.randperm(dim, dtype=torch.long))
    elif ptype == "LinearLU":
        permuter = lambda dim: M.LinearLU(dim, eps=config["linear_lu_eps"])
    elif ptype == "FullCombiner":
        permuter = lambda dim: M.FullCombiner(dim)
    else:
        raise Exception(f"Invalid permuter type: {ptype}")

    transforms = [augmenter]
    L = config["n_flow_layers"]
    for index in range(L):
        transforms.append(M.cif_helper(config, flow_for_cif, attn, pre_attention_mlp))
        if index != L - 1:                      # no ActNorm / permuter after the last coupling
            if config["act_norm"]:
                transforms.append(M.ActNormBijectionCloud(D, data_dep_init=True))
            transforms.append(permuter(D))

    base_dist = M.StandardNormal(shape=(config["sample_size"], D))
    sample_dist = M.Normal(torch.zeros(1), torch.ones(1) * 0.6, shape=(config["sample_size"], D))
    flow = M.Flow(transforms, base_dist, sample_dist, config=config)

    emb = config["input_embedder"]
    if emb == "DGCNNembedder":
        input_embedder = M.DGCNNembedder(emb_dim=config["input_embedding_dim"], n_neighbors=config["n_neighbors"],
                                         out_mlp_dims=config["hidden_dims_embedder_out"])
    elif emb == "DGCNNembedderGlobal":
        input_embedder = M.DGCNNembedderGlobal(input_dim=Din, out_mlp_dims=config["hidden_dims_embedder_out"],
                                               n_neighbors=config["n_neighbors"], emb_dim=config["input_embedding_dim"])
    elif emb == "PAConv":
        input_embedder = M.PointNet2SSGSeg(c=Din - 3, k=config["input_embedding_dim"], out_mlp_dims=config["hidden_dims_embedder_out"])
    elif emb == "idenity":                      # sic, model_initialization.py:173
        input_embedder = nn.Identity()
    else:
        raise Exception("Invalid input embeder!")

    if mode == "train":
        input_embedder.train()
        flow.train()
        if isinstance(input_embedder, M.PointNet2SSGSeg) and not M.PointNet2SSGSeg.TRAINABLE:
            # the PAConv embedder has inference kernels only: it stays frozen in eval() mode (running-statistics BatchNorm) and
            # receives no gradient; the flow on top of it trains
            input_embedder.eval()
            input_embedder.requires_grad_(False)
    else:
        input_embedder.eval()
        flow.eval()
    sharded = False
    if config["data_parallel"]:
        # model_initialization.py:186-188 wraps both modules in nn.DataParallel (one process, one thread per GPU).  Here the same request means
        # scene sharding with ONE PROCESS PER GPU (flowcompare_amd/shard.py: this rank's scenes on its own device, no data-path collective, RCCL
        # only for the loss scalar and the gradient all-reduce): under `python -m torch.distributed.run --nproc-per-node N` every rank builds the
        # same replica on its LOCAL_RANK's device and inner_loop shards the global batch; outside a process group there is nothing to shard over.
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("config['data_parallel'] is true but no torch.distributed process group is initialised: flowcompare_amd shards scenes with one "
                               "process per GPU -- launch with `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` and "
                               "call torch.distributed.init_process_group('nccl') before initialize_flow (nn.DataParallel's one-process threading, "
                               "model_initialization.py:186-188, has no counterpart here)")
        sharded = True
    input_embedder = input_embedder.to(device)
    flow = flow.to(device)

    parameters = list(input_embedder.parameters()) + list(flow.parameters())
    print(f"Number of trainable parameters: {sum(p.numel() for p in parameters)}")
    md = {"parameters": parameters, "flow": flow, "input_embedder": input_embedder}
    if sharded:
        md["sharded"] = True                                  # inner_loop then takes a GLOBAL batch and runs this rank's scenes (shard.sharded_inner_loop)
    return md


def inner_loop(batch, models_dict, config, eps=None):
    """model_initialization.py:206-228.  `eps` (optional) pins the augmenter noise."""
    if models_dict.get("sharded") and not models_dict.get("_in_shard"):
        # config['data_parallel']: `batch` is the GLOBAL batch, every rank runs its own scenes; returns (global loss, LOCAL log_prob, global bpd)
        from . import shard
        models_dict["_in_shard"] = True
        try:
            return shard.sharded_inner_loop(batch, models_dict, config, eps=eps)
        finally:
            models_dict["_in_shard"] = False
    extract_0, extract_1, extra_context = batch
    Din = config["input_dim"]
    extract_0, extract_1 = extract_0[:, :, :Din], extract_1[:, :, :Din]
    if extra_context is not None:
        # einops.repeat(extra_context, 'b c -> b n c', n=config['sample_size']) in the reference
        extra_context = extra_context[:, None, :].expand(-1, config["sample_size"], -1)
    emb = models_dict["input_embedder"](extract_0)
    if config["global"]:
        emb = emb[:, None, :].expand(-1, extract_1.shape[1], -1)
    log_prob = models_dict["flow"].log_prob(extract_1, context=emb, extra_context=extra_context, eps=eps)
    loss = -log_prob.mean()
    with torch.no_grad():
        bpd = loss * math.log2(math.exp(1)) / Din
    return loss, log_prob, bpd


def make_sample(n_points, extract_0, models_dict, config, sample_distrib=None, extra_context=None):
    """model_initialization.py:231-245."""
    extract_0 = extract_0[:, :, :config["input_dim"]]
    emb = models_dict["input_embedder"](extract_0)
    if extra_context is not None:
        extra_context = extra_context[:, None, :].expand(-1, n_points, -1)
    if config["global"]:
        emb = emb[:, None, :].expand(-1, n_points, -1)
    x = models_dict["flow"].sample(num_samples=1, n_points=n_points, context=emb, sample_distrib=sample_distrib,
                                   extra_context=extra_context).squeeze()
    return x
