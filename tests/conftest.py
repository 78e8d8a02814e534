import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Fixture:
    """One tests/golden/*.npz: inputs + expected outputs produced by running the reference (gen_golden.py)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        self.a = {k: z[k] for k in z.files}
        self.cfg = json.loads(bytes(self.a.pop("config_json")).decode())
        self.meta = json.loads(bytes(self.a.pop("meta_json")).decode())
        self.sd_keys = json.loads(bytes(self.a.pop("sd_keys_json")).decode()) if "sd_keys_json" in self.a else None

    def t(self, key, dtype=None):
        if key not in self.a:
            return None
        v = torch.from_numpy(self.a[key])
        return v.to(dtype) if dtype is not None else v

    def eps(self, dtype=torch.float32, prefix="eps"):
        out, i = [], 0
        while f"{prefix}{i}" in self.a:
            out.append(self.t(f"{prefix}{i}", dtype))
            i += 1
        return out

    def derived_cfg(self):
        """config with the keys initialize_flow derives (model_initialization.py:33-45)."""
        c = dict(self.cfg)
        c["extra_context_dim"] = 1 if c["extra_z_value_context"] else 0
        c["using_extra_context"] = c["extra_context_dim"] > 0
        c["global"] = c["input_embedder"] in ["DGCNNembedderGlobal"]
        return c

    def state_dicts(self, dtype=torch.float32):
        """(flow_sd, embedder_sd) synthesised exactly as gen_golden.py loaded them into the reference."""
        import synth
        out = []
        for part in ("flow", "input_embedder"):
            sd = {}
            for name, shape in self.sd_keys[part].items():
                leaf = name.split(".")[-1]
                int_like = leaf in ("num_batches_tracked", "permutation", "inv_permutation")
                ref = torch.zeros(shape, dtype=torch.int64 if int_like else torch.float32)
                key = f"sd/{part}/{name}"
                if key in self.a:
                    sd[name] = torch.from_numpy(self.a[key]).long()
                    continue
                v = synth.synth_tensor(name, ref, self.meta["seed"])
                if v is None:
                    v = _constructor_value(name, shape)
                sd[name] = v if int_like else v.to(dtype)
            # CIFblock: augmenter and slicer share ONE ConditionalNormal; load_state_dict writes the slicer.* entries last,
            # so those are the values the reference actually ran with (a real checkpoint holds identical copies).
            for name in list(sd):
                if ".slicer.noise_dist." in name:
                    sd[name.replace(".slicer.noise_dist.", ".augmenter.noise_dist.")] = sd[name]
            out.append(sd)
        return out


def _constructor_value(name, shape):
    leaf = name.split(".")[-1]
    if leaf == "scale":
        return torch.ones(shape) * 0.6          # sample_dist.scale, model_initialization.py:156-157
    return torch.zeros(shape)                   # base_dist.buffer, sample_dist.loc, std_normal.buffer


E2E_REAL = ["e2e_dulcet_L3", "e2e_c1_global_L2", "e2e_spline_L2", "e2e_affine_exp_L2", "e2e_paconv_L2"]
E2E_TINY = ["e2e_tiny_affine", "e2e_tiny_spline_relu", "e2e_tiny_expcoupling", "e2e_tiny_expcoupling_orig", "e2e_tiny_cif",
            "e2e_tiny_random_permute", "e2e_tiny_FullCombiner", "e2e_tiny_ExponentialCombiner", "e2e_tiny_global_extra",
            "e2e_tiny_identity_aug"]
