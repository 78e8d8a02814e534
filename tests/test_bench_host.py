"""Host-side pieces of bench.py's contract that need no GPU: the two-phase output contract of a multi-rank run (the training leg runs in child
processes AFTER the forward numbers are complete; its outcome becomes the line's `train` object, an error there never loses the forward line) and
the device census that proves N ranks sit on N devices (a world-size-2 gloo run)."""
import json
import os
import socket
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from flowcompare_amd import shard  # noqa: E402


class _Args:
    gpus, config, batch, points, ctx_points, layers, weights, knob, train_steps = 4, "c4_dgcnn_attn_extra_affine", 20, 1024, 1250, None, "conditioned", ["29=1"], 2


def test_training_child_command_and_rendezvous():
    cmd = bench.training_child_command(_Args, 2)
    assert cmd[0] == sys.executable and cmd[1].endswith("bench.py") and "--train-child" in cmd
    for flag, val in (("--gpus", "4"), ("--steps", "2"), ("--config", _Args.config), ("--batch", "20"), ("--points", "1024"), ("--ctx-points", "1250"), ("--knob", "29=1")):
        assert cmd[cmd.index(flag) + 1] == val
    env = bench.training_child_env({"MASTER_PORT": "29500", "MASTER_ADDR": "127.0.0.1", "RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "4", "TORCHELASTIC_USE_AGENT_STORE": "True"})
    assert env["MASTER_PORT"] == "29501" and env["RANK"] == "3" and env["WORLD_SIZE"] == "4"      # same ranks, the next port
    assert "TORCHELASTIC_USE_AGENT_STORE" not in env                                               # ... whose store the children's rank 0 hosts itself
    env1 = bench.training_child_env({})
    assert env1["RANK"] == "0" and env1["WORLD_SIZE"] == "1" and env1["MASTER_ADDR"] == "127.0.0.1"


def test_training_child_outcomes_become_the_train_object():
    ok = json.dumps({"metric": "points/sec (training step ...)", "value": 1.0, "train": {"ms_per_step": 900.0, "steps": 2, "all_reduce_exposed_ms_per_step": 3.5}})
    t = bench.parse_training_child(0, "RCCL banner\n" + ok + "\n")
    assert t["ms_per_step"] == 900.0 and t["all_reduce_exposed_ms_per_step"] == 3.5 and "child" in t["phase"] and "error" not in t
    assert "error" in bench.parse_training_child(134, ok)                     # an abort inside the collective: the line is not trusted
    assert "error" in bench.parse_training_child(0, "no json here")
    assert "time limit" in bench.parse_training_child(-1, "", timed_out=True)["error"]
    assert "error" in bench.parse_training_child(0, json.dumps({"value": 1.0}))    # a line without the train object


def test_kernel_peaks_by_symbol():
    n, peak, _ = bench.kernel_peak("void fc::spline_wide_kernel<0, 2, 3, 3, 0, 2, 3, 3, 0>(fc::SplineWideParams)")
    assert (n, peak) == (3, 2500.0)
    n, peak, _ = bench.kernel_peak("void fc::gemm_f32_kernel<128, 128, 4, 2, 0, 3>(fc::GemmParams)")
    assert (n, peak) == (6, 2500.0)
    n, peak, _ = bench.kernel_peak("void fc::knn_mfma_kernel<64>(float const*, int, int, int*, int, int, int, int const*)")
    assert (n, peak) == (1, 157.3)
    r = bench.roofline_of({"kernel": "void fc::attn16_kernel<64>(fc::Attn16Params)", "launches": 2, "ms": 1.0, "flops": 2.0e9, "bytes": 0.0}, {}, {}, "no pass")
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / 2500.0) < 1e-12 and abs(r["frac_issued"] - 3 * r["frac"]) < 1e-12 and r["traffic"] is None and r["traffic_note"] == "no pass"


def _census_worker(rank, world, port, idents, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, shard.device_census(idents[rank])))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("idents,unique", [(("uuid-a|0:1:0", "uuid-b|0:2:0"), 2), (("uuid-a|0:1:0", "uuid-a|0:1:0"), 1)])
def test_device_census_counts_distinct_devices(idents, unique):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_census_worker, args=(r, 2, port, idents, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(2)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for _, c in res:
        assert c == {"world": 2, "unique_devices": unique}
