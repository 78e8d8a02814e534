#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only).  It never
travels: the committed artefacts are the *.npz fixtures (inputs + expected
outputs, no reference source) plus this script.

How the reference is made importable on a CPU-only box:
IO / visualisation / CUDA-extension modules that the flow maths never calls
are stubbed in sys.modules; nothing in /root/reference is modified.

Weights are NOT stored: every state_dict entry is synthesised from
(seed, name, shape) by tests/golden/synth.py and loaded into the reference
model here, and into the oracle / HIP engine in the tests.

The augmenter noise eps is made an explicit input by replacing
torch.distributions.normal._standard_normal with a FIFO of stored tensors.

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz
"""
import json
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import synth

REF = "/root/reference"


def _import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    stub("laspy")
    sys.modules["laspy"].file = stub("laspy.file", File=object)
    stub("open3d")
    stub("dash_core_components")
    stub("dash_html_components")
    stub("pykeops")
    stub("pykeops.torch", Vi=None, Vj=None)
    stub("pointops_cuda")
    sys.path.insert(0, REF)
    import models  # noqa: F401
    import model_initialization
    import utils
    return models, model_initialization, utils


models, mi, ref_utils = _import_reference()

# ---------------------------------------------------------------- eps FIFO
_EPS_QUEUE = []


def _fifo_standard_normal(shape, dtype, device):
    e = _EPS_QUEUE.pop(0)
    assert tuple(e.shape) == tuple(shape), (e.shape, shape)
    return e.to(dtype=dtype, device=device)


torch.distributions.normal._standard_normal = _fifo_standard_normal


class _FixedZ:
    """sample_distrib stand-in for Flow.sample (transform.py:79-84)."""
    def __init__(self, z):
        self.z = z

    def sample(self, num_samples, n_points=None, context=None):
        return self.z


# ---------------------------------------------------------------- helpers
def load_cfg(name, **over):
    cfg = ref_utils.config_loader(f"{REF}/config/{name}.yaml")
    cfg["load_checkpoint"] = False
    cfg.update(over)
    return cfg


def build(cfg, seed, dtype):
    torch.manual_seed(0)
    md = mi.initialize_flow(cfg, "cpu", "test")
    kept = {}
    for key in ("flow", "input_embedder"):
        sd = md[key].state_dict()
        new = synth.synth_state_dict(sd, seed)
        for k, v in sd.items():
            if k.split(".")[-1] in ("permutation", "inv_permutation"):
                kept[f"sd/{key}/{k}"] = v.numpy().astype(np.int32)     # constructor-drawn, stored in the fixture
        md[key].load_state_dict(new)
        md[key].to(dtype)
        md[key].eval()
    md["_kept"] = kept
    return md


def eps_shapes(cfg, B, N):
    """Noise tensors consumed by one forward, in draw order (F5)."""
    shapes = []
    D, Din = cfg["latent_dim"], cfg["input_dim"]
    if D > Din and cfg["augmenter_dist"] == "ConditionalNormal":
        shapes.append((B, N, D - Din))
    if cfg["latent_dim"] < cfg["cif_latent_dim"]:
        for _ in range(cfg["n_flow_layers"]):
            shapes.append((B, N, cfg["cif_latent_dim"] - D))
    return shapes


def run_forward(cfg, md, batch, eps_list, dtype, keep_pts=8):
    """inner_loop + per-transform records (z on the first keep_pts points, ldj)."""
    recs = []

    def hook(mod, args, kwargs, out):
        z, ldj = out
        if not torch.is_tensor(ldj):
            ldj = torch.zeros(z.shape[:-1], dtype=z.dtype)
        recs.append((z[:, :keep_pts].detach().clone(), ldj.detach().clone().expand(z.shape[:-1]).clone()))
    hs = [t.register_forward_hook(hook, with_kwargs=True) for t in md["flow"].transforms]
    e0, e1, ex = batch
    b = (e0.to(dtype), e1.to(dtype), None if ex is None else ex.to(dtype))
    _EPS_QUEUE[:] = [e.to(dtype) for e in eps_list]
    with torch.no_grad():
        emb = md["input_embedder"](b[0][:, :, :cfg["input_dim"]])
        _EPS_QUEUE[:] = [e.to(dtype) for e in eps_list]
        loss, lp, bpd = mi.inner_loop(b, md, cfg)
    for h in hs:
        h.remove()
    assert not _EPS_QUEUE
    return dict(emb=emb, loss=loss, log_prob=lp, bpd=bpd,
                z_sub=torch.stack([r[0] for r in recs]) if len({r[0].shape for r in recs}) == 1 else None,
                z_last=recs[-1][0], ldj=torch.stack([r[1] for r in recs]))


def npy(t):
    return None if t is None else t.detach().cpu().numpy()


def save(name, cfg, arrays, meta=None):
    arrays = {k: v for k, v in arrays.items() if v is not None}
    clean_cfg = {k: v for k, v in cfg.items()}
    arrays["config_json"] = np.frombuffer(json.dumps(clean_cfg).encode(), dtype=np.uint8)
    arrays["meta_json"] = np.frombuffer(json.dumps(meta or {}).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  {os.path.getsize(path)/1024:.0f} KiB")


def e2e_case(name, cfg_name, over, B, N, M, seed, with_sample=True, z_scale=0.6):
    cfg = load_cfg(cfg_name, sample_size=N, **over)
    print(f"[{name}] {cfg_name} {over} B={B} N={N} M={M}")
    e0 = synth.synth_points(name + "/e0", B, M, seed)
    e1 = synth.synth_points(name + "/e1", B, N, seed)
    ex = None
    if cfg["extra_z_value_context"]:
        ex = torch.from_numpy(synth.uniform(name + "/extra", (B, 1), 0.0, 15.0, seed)).float()
    arrays = dict(extract_0=npy(e0), extract_1=npy(e1), extra=npy(ex))
    out = {}
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        c = dict(cfg)
        md = build(c, seed, dtype)
        eps_list = [torch.from_numpy(synth.normal(f"{name}/eps{i}", s, seed)).float()
                    for i, s in enumerate(eps_shapes(c, B, N))]
        r = run_forward(c, md, (e0, e1, ex), eps_list, dtype)
        out[tag] = r
        arrays[f"log_prob_{tag}"] = npy(r["log_prob"])
        arrays[f"bpd_{tag}"] = npy(r["bpd"])
        arrays[f"loss_{tag}"] = npy(r["loss"])
        arrays[f"ldj_{tag}"] = npy(r["ldj"])
        if tag == "f64":
            arrays.update(md["_kept"])
            keys = {k: {n: list(v.shape) for n, v in md[k].state_dict().items()} for k in ("flow", "input_embedder")}
            arrays["sd_keys_json"] = np.frombuffer(json.dumps(keys).encode(), dtype=np.uint8)
            arrays["emb_f64"] = npy(r["emb"])
            arrays["z_sub_f64"] = npy(r["z_sub"])
            arrays["z_last_f64"] = npy(r["z_last"])
            for i, e in enumerate(eps_list):
                arrays[f"eps{i}"] = npy(e)
            if with_sample:
                # inverse / sampling path (transform.py:79-84) with a fixed latent z
                npts = 24
                z = torch.from_numpy(synth.normal(name + "/zsample", (1, npts, c["latent_dim"]), seed)).double() * z_scale
                n_inv_eps = []
                if c["latent_dim"] < c["cif_latent_dim"]:
                    n_inv_eps = [torch.from_numpy(synth.normal(f"{name}/inveps{i}", (1, npts, c["cif_latent_dim"] - c["latent_dim"]), seed)).double()
                                 for i in range(c["n_flow_layers"])]
                _EPS_QUEUE[:] = list(n_inv_eps)
                with torch.no_grad():
                    xs = mi.make_sample(npts, e0[:1].double(), md, c, sample_distrib=_FixedZ(z),
                                        extra_context=None if ex is None else ex[:1].double())
                assert not _EPS_QUEUE
                arrays["sample_z"] = npy(z)
                arrays["sample_x_f64"] = npy(xs)
                for i, e in enumerate(n_inv_eps):
                    arrays[f"inveps{i}"] = npy(e)
    d = (out["f64"]["log_prob"] - out["f32"]["log_prob"].double()).abs()
    print(f"   log_prob mean {out['f64']['log_prob'].mean():.4f}  f32-vs-f64 max {d.max():.2e} mean {d.mean():.2e}"
          f"  bpd diff {abs(out['f64']['bpd'].item()-out['f32']['bpd'].item()):.2e}")
    save(name, cfg, arrays, meta=dict(B=B, N=N, M=M, seed=seed, cfg_name=cfg_name, over=over))


TINY = dict(latent_dim=12, cif_latent_dim=12, attn_dim=16, attn_input_dim=8, input_embedding_dim=10,
            cross_dim_head=8, hidden_dims=[24, 24, 24], pre_attention_mlp_hidden_dims=[12, 12, 12],
            net_augmenter_dist_hidden_dims=[20, 20], hidden_dims_embedder_out=[32, 32], n_neighbors=8)


def knn_case():
    """knn + get_graph_feature (pytorch_gcn.py:13-47) incl. fp64 margins for near-tie detection."""
    from models.pytorch_gcn import knn, get_graph_feature
    name = "op_knn"
    arrays = {}
    for tag, (B, C, M, k) in dict(xyzrgb=(2, 6, 96, 40), feat64=(2, 64, 80, 40)).items():
        x = torch.from_numpy(synth.uniform(f"{name}/{tag}", (B, C, M), -1.0, 1.0, 1)).float()
        idx32 = knn(x, k)
        xd = x.double()
        inner = -2 * torch.matmul(xd.transpose(2, 1), xd)
        xx = torch.sum(xd ** 2, dim=1, keepdim=True)
        pd = -xx - inner - xx.transpose(2, 1)
        srt = pd.sort(dim=-1, descending=True)
        idx64 = srt.indices[..., :k]
        margin = srt.values[..., k - 1] - srt.values[..., k]
        feat = get_graph_feature(x, k=k)
        arrays[f"{tag}_x"] = npy(x)
        arrays[f"{tag}_idx_f32"] = npy(idx32).astype(np.int16)
        arrays[f"{tag}_idx_f64"] = npy(idx64).astype(np.int16)
        arrays[f"{tag}_margin_f64"] = npy(margin)
        arrays[f"{tag}_feat_sum"] = npy(feat.double().sum(dim=-1))   # [B,2C,M] compact check of the gather
    save(name, {}, arrays)


def spline_case():
    """unconstrained_rational_quadratic_spline fwd + inverse (spline_coupling.py:24-169) incl. tails & knots."""
    from models.spline_coupling import unconstrained_rational_quadratic_spline as urqs
    name = "op_spline"
    P, D, K = 64, 20, 8
    x = torch.from_numpy(synth.uniform(name + "/x", (2, P, D), -4.0, 4.0, 2)).double()
    x[0, 0, :5] = torch.tensor([-3.0, 3.0, 0.0, -3.0000001, 2.9999999], dtype=torch.float64)
    w = torch.from_numpy(synth.uniform(name + "/w", (2, P, D, K), -2.0, 2.0, 2)).double()
    h = torch.from_numpy(synth.uniform(name + "/h", (2, P, D, K), -2.0, 2.0, 2)).double()
    d = torch.from_numpy(synth.uniform(name + "/d", (2, P, D, K + 1), -2.0, 2.0, 2)).double()
    y, ld = urqs(x, w, h, d)
    xi, ldi = urqs(y, w, h, d, inverse=True)
    y32, ld32 = urqs(x.float(), w.float(), h.float(), d.float())
    print(f"[op_spline] roundtrip max {float((xi-x).abs().max()):.2e}")
    save(name, {}, dict(x=npy(x), w=npy(w), h=npy(h), d=npy(d), y=npy(y), logabsdet=npy(ld),
                        x_inv=npy(xi), logabsdet_inv=npy(ldi), y_f32=npy(y32), logabsdet_f32=npy(ld32)))


def patch_pointops():
    """The six CUDA kernels of the PAConv embedder cannot run here: substitute the oracle's CPU restatements
    INTO the reference so that its own Python (PointNet2SSGSeg, QueryAndGroup, PAConv, ScoreNet, FP modules) runs on CPU."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import paconv_oracle as P
    from models.scene_seg_PAConv.lib.pointops.functions import pointops
    pointops.furthestsampling = lambda xyz, m: P.furthest_sampling(xyz, m).int()
    pointops.gathering = lambda feat, idx: P.gathering(feat, idx.long())
    pointops.knnquery_heap = lambda nsample, xyz, new_xyz: P.knnquery_heap(nsample, xyz, new_xyz).int()
    pointops.grouping = lambda feat, idx: P.grouping(feat, idx.long())
    pointops.nearestneighbor = lambda unknown, known: P.nearest_neighbor3(unknown, known)
    pointops.interpolation = lambda feat, idx, w: P.interpolation(feat, idx.long(), w)


def paconv_embedder_case():
    """PAConv embedder alone: point counts 320 -> 80 -> 20 -> 5 -> 1 exercise n < nsample (heap slots left at index 0) and a
    single known point in feature propagation."""
    name = "emb_paconv"
    cfg = load_cfg("summer-terrain", sample_size=64, n_flow_layers=1)
    pts = synth.synth_points(name + "/pts", 2, 320, 31)
    arrays = dict(pts=npy(pts))
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        md = build(dict(cfg), 31, dtype)
        with torch.no_grad():
            emb = md["input_embedder"](pts.to(dtype))
        arrays[f"emb_{tag}"] = npy(emb)
        if tag == "f64":
            keys = {k: {n: list(v.shape) for n, v in md[k].state_dict().items()} for k in ("flow", "input_embedder")}
            arrays["sd_keys_json"] = np.frombuffer(json.dumps(keys).encode(), dtype=np.uint8)
    print(f"[{name}] f32-vs-f64 max {np.abs(arrays['emb_f64'] - arrays['emb_f32']).max():.2e}")
    save(name, cfg, arrays, meta=dict(B=2, M=320, seed=31))


def main():
    torch.set_num_threads(8)
    patch_pointops()
    paconv_embedder_case()
    e2e_case("e2e_paconv_L2", "summer-terrain", dict(n_flow_layers=2), B=2, N=48, M=256, seed=32)
    # ---- real-dims end-to-end slices (few layers)
    e2e_case("e2e_dulcet_L3", "dulcet-universe", dict(n_flow_layers=3), B=2, N=64, M=80, seed=11)
    e2e_case("e2e_c1_global_L2", "helpful-sponge", dict(n_flow_layers=2), B=2, N=64, M=64, seed=12)
    e2e_case("e2e_spline_L2", "swept-energy", dict(n_flow_layers=2, flow_type="RationalQuadraticSplineCoupling"),
             B=2, N=48, M=64, seed=13)
    e2e_case("e2e_affine_exp_L2", "swept-energy", dict(n_flow_layers=2, affine_scale_fn="exp"), B=1, N=40, M=48, seed=14)
    # ---- tiny-dims variants: every code path of cif_block / permuters / couplings
    e2e_case("e2e_tiny_affine", "dulcet-universe", dict(n_flow_layers=4, **TINY), B=3, N=20, M=24, seed=21)
    e2e_case("e2e_tiny_spline_relu", "dulcet-universe",
             dict(n_flow_layers=3, flow_type="RationalQuadraticSplineCoupling", coupling_block_nonlinearity="RELU", **TINY),
             B=2, N=20, M=24, seed=22)
    e2e_case("e2e_tiny_expcoupling", "swept-energy",
             dict(n_flow_layers=2, flow_type="ExponentialCoupling", coupling_expm_algo="torch",
                  coupling_block_nonlinearity="ELU", **TINY), B=2, N=20, M=24, seed=23)
    e2e_case("e2e_tiny_expcoupling_orig", "swept-energy",
             dict(n_flow_layers=2, flow_type="ExponentialCoupling", coupling_expm_algo="original", **TINY),
             B=2, N=20, M=24, seed=24)
    cif = dict(TINY); cif.update(cif_latent_dim=16, net_cif_dist_hidden_dims=[16, 16], affine_cif_hidden=[16, 16, 16])
    e2e_case("e2e_tiny_cif", "swept-energy", dict(n_flow_layers=3, **cif), B=2, N=20, M=24, seed=25, z_scale=0.05)
    for perm in ("random_permute", "FullCombiner", "ExponentialCombiner"):
        e2e_case(f"e2e_tiny_{perm}", "swept-energy", dict(n_flow_layers=3, permuter_type=perm, act_norm=(perm != "FullCombiner"), **TINY),
                 B=2, N=20, M=24, seed=26)
    glob = dict(TINY); glob.update(input_embedder="DGCNNembedderGlobal")
    e2e_case("e2e_tiny_global_extra", "dulcet-universe", dict(n_flow_layers=3, **glob), B=2, N=24, M=24, seed=27)
    ident = dict(TINY); ident.update(latent_dim=6, cif_latent_dim=6)
    e2e_case("e2e_tiny_identity_aug", "swept-energy", dict(n_flow_layers=3, **ident), B=2, N=20, M=24, seed=28)
    # ---- op-level
    knn_case()
    spline_case()


if __name__ == "__main__":
    main()
