"""bench.py's contract with the driver: ONE JSON line on stdout (RCCL's banner, the reference API's prints and the progress log go to
stderr) with the metric / timing keys, the `roofline` object (frac = algorithmic fraction, frac_issued beside it), `cpu_baseline` and the
`train` object -- on a 2-layer, 2 x 256-point workload so that the whole run takes seconds.  The numbers of such a run mean nothing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_cpu_baseline_and_train():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--layers", "2", "--batch", "2", "--points", "256",
           "--train-steps", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {r.stdout[:400]}"
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "train", "weights", "fp16_fallbacks"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["unit"] == "nats/sec" and j["value"] > 0 and abs(j["value"] - 2 * 256 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    assert "workload" in j["config"] and "model" not in j["config"]
    roof = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches", "launches_bracketed"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    if "frac_issued" in roof:
        assert roof["frac_issued"] >= roof["frac"]                  # the issue rate of a limb loop is a multiple of the algorithmic fraction
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "nats/sec" and "sample" in cb
    tr = j["train"]
    assert "error" not in tr, tr
    assert tr["steps"] == 1 and tr["ms_per_step"] > 0 and tr["points_per_sec"] > 0 and tr["peak_mem_GiB"] > 0 and tr["loss"] == tr["loss"]


def test_training_leg_as_guarded_second_phase_and_device_census():
    """What a multi-rank run does, on one rank: the forward line first (kept in memory), then the training leg in a CHILD process with its own RCCL
    process group (bench.py --train-in-child); still ONE line on stdout, with the `train` object from the child (all-reduce exposed time included)
    and the `rccl` census (world 1, 1 distinct device).  Under torch.distributed.run the same path runs with N ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29611",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--layers", "2", "--batch", "2", "--points", "256", "--ctx-points", "320",
           "--train-steps", "1", "--train-in-child", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, f"stdout must carry exactly one JSON line, got {len(lines)}: {r.stdout[:400]}"
    j = json.loads(lines[0])
    assert j["rccl"]["world"] == 1 and j["rccl"]["unique_devices"] == 1
    assert j["config"]["points_per_scene"] == 256 and j["config"]["context_points_per_scene"] == 320
    tr = j["train"]
    assert "error" not in tr, tr
    assert tr["steps"] == 1 and tr["ms_per_step"] > 0 and "child" in tr["phase"] and tr["all_reduce_exposed_ms_per_step"] is not None
    for k in j["kernels"]:
        assert "frac" in k and "traffic" in k and ("traffic_unit" in k or "traffic_note" in k)
