"""CPU-side checks: checkpoint-name compatibility of the module mirror, config handling, the C-ABI surface
(library loads and exports every symbol include/fcflow.h declares) and 'no silent fallback' behaviour."""
import ctypes
import os
import re

import pytest
import torch

import flowcompare_amd as fa
from flowcompare_amd import engine
from conftest import E2E_REAL, E2E_TINY, ROOT, Fixture


@pytest.mark.parametrize("name", E2E_REAL + E2E_TINY)
def test_state_dict_names_match_reference(name):
    """Our containers must expose exactly the reference's checkpoint keys and shapes."""
    fx = Fixture(name)
    cfg = dict(fx.cfg)
    md = fa.initialize_flow(cfg, device="cpu", mode="test")
    for part in ("flow", "input_embedder"):
        ours = {k: list(v.shape) for k, v in md[part].state_dict().items()}
        assert ours == fx.sd_keys[part], f"{part}: " + str(set(ours) ^ set(fx.sd_keys[part]))
    assert cfg["extra_context_dim"] == (1 if cfg["extra_z_value_context"] else 0)
    assert cfg["global"] == (cfg["input_embedder"] == "DGCNNembedderGlobal")
    assert len(md["parameters"]) == len(list(md["flow"].parameters())) + len(list(md["input_embedder"].parameters()))


def test_load_save_roundtrip(tmp_path):
    fx = Fixture("e2e_tiny_affine")
    cfg = dict(fx.cfg)
    md = fa.initialize_flow(cfg, device="cpu", mode="test")
    sd_flow, sd_emb = fx.state_dicts()
    fa.load_flow({"flow": sd_flow, "input_embedder": sd_emb}, md)
    p = tmp_path / "ckpt.pt"
    opt = torch.optim.Adam(md["parameters"], lr=1e-4)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt)
    fa.save_flow(md, cfg, opt, sched, str(p))
    ck = torch.load(str(p), weights_only=False)
    assert set(ck) == {"config", "optimizer", "flow", "input_embedder", "scheduler"}
    for k, v in sd_flow.items():
        assert torch.equal(ck["flow"][k].float(), v.float())


def test_named_configs_and_reference_yaml_format(tmp_path):
    c2 = fa.named_config("c2_dgcnn_attn_spline")
    assert c2["flow_type"] == "RationalQuadraticSplineCoupling" and c2["n_flow_layers"] == 115 and c2["latent_dim"] == 300
    c1 = fa.named_config("c1_dgcnn_global_affine")
    assert c1["input_embedder"] == "DGCNNembedderGlobal" and c1["input_embedding_dim"] == 124 and len(c1["hidden_dims"]) == 6
    p = tmp_path / "wandb_style.yaml"
    p.write_text("latent_dim:\n  desc: x\n  value: 12\nflow_type:\n  value: AffineCoupling\n")
    c = fa.config_loader(str(p))
    assert c["latent_dim"] == 12 and c["flow_type"] == "AffineCoupling" and c["n_neighbors"] == 40


def test_invalid_configs_raise_like_reference():
    base = Fixture("e2e_tiny_affine").cfg
    for over, msg in ((dict(flow_type="Nope"), "Invalid flow type"), (dict(permuter_type="Nope"), "Invalid permuter type"),
                      (dict(coupling_block_nonlinearity="TANH"), "Invalid coupling_block_nonlinearity"),
                      (dict(latent_dim=4), "Latent dim < Input dim"), (dict(input_embedder="Nope"), "Invalid input embeder!"),
                      (dict(cif_latent_dim=8), "Augment dim smaller than main latent!")):
        cfg = dict(base); cfg.update(over)
        with pytest.raises(Exception, match=msg):
            fa.initialize_flow(cfg, device="cpu", mode="test")
    cfg = dict(Fixture("e2e_tiny_cif").cfg); cfg["extra_z_value_context"] = True
    with pytest.raises(Exception, match="Not implemented extra context with cif"):
        fa.initialize_flow(cfg, device="cpu", mode="test")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "fcflow.h")).read()
    declared = set(re.findall(r"\b(fc_[a-z0-9_]+)\s*\(", header))
    assert declared == set(engine.EXPORTS), declared ^ set(engine.EXPORTS)
    assert os.path.exists(engine.LIB_PATH), "libfcflow.so missing: __graft_entry__.build() must have produced it"
    lib = ctypes.CDLL(engine.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} not exported"
    assert lib.fc_abi_version() == engine.ABI_VERSION


def test_no_cpu_fallback():
    """The product path must fail loudly without a HIP device; it must never route through PyTorch or the oracle."""
    fx = Fixture("e2e_tiny_affine")
    cfg = dict(fx.cfg)
    md = fa.initialize_flow(cfg, device="cpu", mode="test")
    batch = (fx.t("extract_0"), fx.t("extract_1"), fx.t("extra"))
    with pytest.raises(RuntimeError, match="HIP device|no CPU path"):
        fa.inner_loop(batch, md, cfg)
    with pytest.raises(RuntimeError, match="parameter container"):
        md["flow"].transforms[1](batch[1])
    src = "".join(open(os.path.join(ROOT, "flowcompare_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "flowcompare_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src and "flow_oracle" not in src


def test_spline_parameter_columns_are_a_bijection_in_register_slot_order():
    """csrc/spline.h: the 3K+1 parameters of DPT = 128 // (3K+1) transformed dims share a 128-column GEMM tile.  K = 8 packs them in the
    register-slot order of the transposed MFMA product (dims 0, 1 / 2, 3 in slots 0..49 of the lower / upper half-wave, dim 4 split
    14 + 11, columns 125..127 unused); K = 4, 16 stay dim-major.  Checks, without a GPU: every (dim, parameter) owns its own column inside
    its dim's tile, `spline_tile_pos` inverts the mapping to the dim-major position, and the slot arithmetic the kernel relies on."""
    lib = ctypes.CDLL(engine.LIB_PATH)
    col, pos = lib.fc_debug_spline_col, lib.fc_debug_spline_tile_pos
    for K in (4, 8, 16):
        per, dpt = 3 * K + 1, 128 // (3 * K + 1)
        for d2 in (1, dpt, dpt + 1, 150):
            seen = {}
            for j in range(d2):
                for pp in range(per):
                    c = col(j, pp, K)
                    assert c // 128 == j // dpt and c not in seen, (K, j, pp, c)
                    seen[c] = (j, pp)
                    assert pos(c % 128, K) == (j % dpt) * per + pp                 # where the LDS-tile epilogues put that column
            assert len(seen) == d2 * per
    # K = 8: slot s of half h is column (s // 16) * 32 + ((s % 16) // 4) * 8 + 4 h + s % 4 (accumulator block, register group, lane half, register)
    slot_col = lambda s, h: (s // 16) * 32 + ((s % 16) // 4) * 8 + 4 * h + s % 4
    for dl in range(4):
        assert [col(dl, pp, 8) for pp in range(25)] == [slot_col((dl & 1) * 25 + pp, dl >> 1) for pp in range(25)]
    assert [col(4, pp, 8) for pp in range(25)] == [slot_col(50 + pp, 0) for pp in range(14)] + [slot_col(50 + pp, 1) for pp in range(11)]
    assert sorted(set(range(128)) - {col(j, pp, 8) for j in range(5) for pp in range(25)}) == [125, 126, 127]


def test_inference_entry_points_are_torch_library_ops():
    """the forward / embedder entry points are registered as torch.library custom ops (flowcompare_amd/library_ops.py), HIP devices
    only; their fake implementations give the output shapes (what torch.compile / export trace through); a CPU call fails loudly."""
    from flowcompare_amd import library_ops
    for name in ("flow_log_prob", "context_embed"):
        assert hasattr(torch.ops.flowcompare_amd, name)
    x = torch.empty(3, 7, 6, device="meta")
    lp = torch.ops.flowcompare_amd.flow_log_prob(x, torch.empty(3, 9, 64, device="meta"), None, [torch.empty(3, 7, 294, device="meta")], 0)
    assert lp.shape == (3, 7) and lp.dtype == torch.float32 and lp.device.type == "meta"

    class _H:                                              # stands in for an engine handle: the fake implementation only reads its geometry
        out_dim, is_global = 64, False
    h = _H()
    key = library_ops.register(h)
    emb = torch.ops.flowcompare_amd.context_embed(torch.empty(2, 11, 6, device="meta"), key)
    assert emb.shape == (2, 11, 64)
    with pytest.raises(NotImplementedError):
        torch.ops.flowcompare_amd.flow_log_prob(torch.zeros(1, 2, 6), torch.zeros(1, 2, 64), None, [], key)       # no CPU kernel exists
    del h
    with pytest.raises(RuntimeError, match="no live engine handle"):
        library_ops._get(key)


def test_data_parallel_without_a_process_group_says_how_to_launch():
    """config['data_parallel'] (model_initialization.py:186-188: nn.DataParallel) means scene sharding with one process per GPU here; outside a
    torch.distributed process group initialize_flow says so instead of building something that cannot shard."""
    fx = Fixture("e2e_tiny_affine")
    cfg = dict(fx.cfg)
    cfg["data_parallel"] = True
    with pytest.raises(RuntimeError, match="torch.distributed.run"):
        fa.initialize_flow(cfg, device="cpu", mode="test")
