#!/usr/bin/env python3
"""Headline benchmark: forward log-prob throughput ("nats/sec" = per-point log-prob values per second)
of the FlowCompare conditional flow on synthetic 4096-point coloured pairs.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path (`inner_loop`: DGCNN context embedder -> 115-layer conditional flow -> loss/bpd) over one
batch that is already resident in HBM.  Workload at N=1: BASELINE.json configs[1] = C2 (DGCNN + cross-attention, rational-
quadratic spline coupling, batch 16, 4096 target + 4096 context points).  For N > 1 every rank runs the same per-GPU batch on
its own scenes (scenes are independent: no data-path collective; weak scaling); the global loss is one scalar all-reduce.

Prints ONE JSON line on rank 0, with `roofline` (dominant kernel: a sample of its launches -- every n-th, about 32 per step --
is bracketed by HIP events inside the library over the timed region; the other kernels are only bracketed in the last warmup step,
which yields the `kernels` breakdown --
bracketing every launch of the timed region costs 3 % of the throughput) and `cpu_baseline` (the pinned CPU oracle timed on the
host cores on a bounded sample of the same workload).
"""
import argparse
import contextlib
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import flowcompare_amd as fa  # noqa: E402
from flowcompare_amd import engine, shard  # noqa: E402

# SURVEY.md §8(d) / BASELINE.md §4: algorithmic MFLOP per target point (1 MAC = 2 FLOP), by config
ALG_MFLOP_PER_POINT = {("c1_dgcnn_global_affine", 1024): 396.0, ("c2_dgcnn_attn_spline", 4096): 859.0, ("c4_dgcnn_attn_extra_affine", 4096): 453.0,
                       ("c4_dgcnn_attn_extra_affine", 16384): 824.0}      # (config, points per scene): C1, C2, C4, C5
PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MATRIX_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense (not the 2:1-sparse headline)
PEAK_HBM_GBS = 8000.0
SPLIT_MFMA_PER_PRODUCT = {5: 3, 6: 3, 7: 3, 8: 3, 9: 3, 10: 3, 11: 3, 3: 6}   # gemm.hip: the split-fp16 loops (VAR 5; VAR 7 = A operand pre-split) issue 3 fp16 MFMAs per fp32-equivalent product block, the
                                        # split-bf16 loop (VAR 3, its fallback) 6 bf16 MFMAs; both run at the 2500 TFLOP/s dense 16-bit rate


def synth_pairs(B, n_ctx, n_tgt, seed, device):
    """xyz ~ U(-1,1)^3, pair-centred and scaled to the joint unit sphere; rgb ~ U[0,1)."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, n_ctx + n_tgt, 3, generator=g) * 2 - 1
    xyz = xyz - xyz.mean(1, keepdim=True)
    xyz = xyz / xyz.norm(dim=-1).amax(1)[:, None, None]
    rgb = torch.rand(B, n_ctx + n_tgt, 3, generator=g)
    pts = torch.cat((xyz, rgb), -1)
    extra = torch.rand(B, 1, generator=g) * 15.0
    return pts[:, :n_ctx].contiguous().to(device), pts[:, n_ctx:].contiguous().to(device), extra.to(device), g


def host_cores():
    """CPU share this process may use: affinity mask, capped by the cgroup quota and by 16 (the 1-GPU box share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


@contextlib.contextmanager
def stdout_to_stderr():
    """File-descriptor level: RCCL prints its version banner to stdout when a communicator is created; stdout carries only the JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def init_rccl(dist, dev, **kw):
    with stdout_to_stderr():
        dist.init_process_group("nccl", device_id=dev, **kw)
        t = torch.zeros(1, device=dev)
        dist.all_reduce(t)                                # the communicator (and its banner) come with the first collective
        torch.cuda.synchronize()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(cfg, md, points, seed, sd=None):
    """The pinned CPU oracle (oracle/flow_oracle.py, fp32, eager PyTorch on the host cores) on ONE scene of the workload."""
    from oracle import flow_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle on {cores} host threads, 1 scene x {points} points ...")
    c = dict(cfg)
    c["sample_size"] = points
    sd_f, sd_e = sd if sd is not None else ({k: v.detach().cpu() for k, v in md["flow"].state_dict().items()},
                                            {k: v.detach().cpu() for k, v in md["input_embedder"].state_dict().items()})
    e0, e1, ex, g = synth_pairs(1, points, points, seed, "cpu")
    eps = [torch.randn(1, points, c["latent_dim"] - c["input_dim"], generator=g)] if c["latent_dim"] > c["input_dim"] else []
    batch = (e0, e1, ex if c["extra_z_value_context"] else None)
    with torch.no_grad():
        t0 = time.perf_counter()
        _, lp, _ = O.inner_loop(c, sd_f, sd_e, batch, eps)
        dt = time.perf_counter() - t0
    return {"value": points / dt, "unit": "nats/sec", "cores": cores, "kind": "port",
            "sample": f"1 scene of the workload (B=1, {points} target + {points} context points, all {c['n_flow_layers']} layers, fp32), "
                      f"{dt:.1f} s of host time"}, lp


def kernel_peak(name):
    """(limb products issued per fp32-equivalent product, peak TFLOP/s, note) of a matrix-core kernel by the name rocprofv3 prints for it.
    SURVEY.md 8(d): achieved = ALGORITHMIC fp32-equivalent FLOPs of the launches / their time, against the dense 16-bit MFMA peak the loop runs
    on; the split loops issue n limb products per fp32-equivalent product: that issue rate is frac_issued, never frac."""
    var = next((v for v in SPLIT_MFMA_PER_PRODUCT if name.endswith(f", {v}>(fc::GemmParams)")), None)   # ..., VAR>
    if var is None and any(k in name for k in ("attn16_kernel", "mlp_rows_kernel", "premlp_rows_kernel", "premlp_kernel", "spline_wide_kernel")):
        var = 5                                             # split-fp16 attention / row-resident MLP chains / the one-accumulator 256 x 256 kernels: 3 limb products
    if var is not None:
        n = SPLIT_MFMA_PER_PRODUCT[var]
        return n, PEAK_BF16_MATRIX_TFLOPS, (f"split-{'fp16' if var != 3 else 'bf16'} loop on the dense fp16/bf16 MFMA peak 2500 TFLOP/s: achieved = algorithmic fp32-equivalent "
                                            f"multiply-add FLOPs (padding excluded); the loop issues {n} limb products per fp32-equivalent product, so the 16-bit MFMA FLOPs "
                                            "actually issued are issued_tflops (frac_issued of the same peak); the fp32-input MFMA peak would be 157.3 TFLOP/s")
    return 1, PEAK_F32_MATRIX_TFLOPS, "fp32-input MFMA (v_mfma_f32_32x32x2_f32) against its dense peak 157.3 TFLOP/s"


def pmc_tables(workload_key):
    """Committed counter passes of THIS workload (profiles/pmc_traffic.json: FETCH_SIZE / WRITE_SIZE per launch, FETCH doubled per the gfx950
    correction; profiles/pmc_mfma_busy.json: matrix-pipe busy fraction and clock), keyed by the kernel names rocprofv3 prints.  Counters cannot be
    collected inside a timed run; a table measured on another workload does not apply.  Returns (traffic, busy, reason-or-None)."""
    out = [{}, {}]
    why = None
    for i, f in enumerate(("pmc_traffic.json", "pmc_mfma_busy.json")):
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", f)))
            meas = pj.get("measured", {})
            if meas.get("workload", "c2_dgcnn_attn_spline 16 x 4096 + 4096") != workload_key:
                why = f"the committed counter passes are of another workload ({meas.get('workload', 'c2_dgcnn_attn_spline 16 x 4096 + 4096')})"
                continue
            out[i] = {"kernels": pj["kernels"], "measured": meas}
        except (OSError, KeyError, ValueError):
            why = why or f"profiles/{f} missing"
    return out[0], out[1], why


def roofline_of(p, traffic, busy, why):
    """One kernel's roofline entry from its in-library HIP-event record p = {kernel, launches, ms, flops, bytes}."""
    per_launch_ms = p["ms"] / p["launches"]
    if p["flops"] > 0:
        useful = p["flops"] / p["launches"] / (per_launch_ms * 1e-3) / 1e12          # fp32-equivalent multiply-add TFLOP/s
        n, peak, note = kernel_peak(p["kernel"])
        r = {"bound": "mfma", "kernel": p["kernel"], "achieved": useful, "peak": peak, "unit": "TFLOP/s", "frac": useful / peak,
             "frac_issued": useful * n / peak, "issued_tflops": useful * n, "peak_source": "MI355X_MICROARCH.md; " + note}
    elif p["bytes"] > 0:
        achieved = p["bytes"] / p["launches"] / (per_launch_ms * 1e-3) / 1e9
        r = {"bound": "hbm", "kernel": p["kernel"], "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS}
    else:
        r = {"bound": None, "kernel": p["kernel"], "achieved": None, "peak": None, "unit": None, "frac": None}
    t = traffic.get("kernels", {}).get(p["kernel"]) if traffic else None
    r["traffic"] = t["hbm_bytes_per_launch"] if t else None
    if t:
        meas = traffic.get("measured", {})
        r["traffic_unit"] = ("HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                             f"(profiles/pmc_traffic.json, measured {meas.get('date', '?')} on build {meas.get('label', '?')})")
    else:
        r["traffic_note"] = why or "no counter pass committed for this kernel"
    b = busy.get("kernels", {}).get(p["kernel"]) if busy else None
    if b:
        r["matrix_pipe_busy"] = b.get("mfma_busy_frac")
        r["clock_ghz"] = b.get("clock_ghz")
    r["avg_launch_ms"] = per_launch_ms
    r["launches"] = p["launches"]
    return r


def device_identity(dev):
    """What tells two GPUs apart: UUID where torch exposes it, PCI address otherwise (both constant per device)."""
    pr = torch.cuda.get_device_properties(dev)
    uuid = str(getattr(pr, "uuid", ""))
    pci = ":".join(str(getattr(pr, a, "?")) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    return f"{uuid}|{pci}|{pr.name}"


def training_child_command(args, steps):
    """argv of the child that runs the training leg of THIS workload (bench.py --train-child: prints bench.py --train's JSON line)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--train-child", "--gpus", str(args.gpus), "--steps", str(steps), "--warmup", "1",
           "--config", args.config, "--batch", str(args.batch), "--points", str(args.points), "--weights", args.weights]
    if args.ctx_points:
        cmd += ["--ctx-points", str(args.ctx_points)]
    if args.layers:
        cmd += ["--layers", str(args.layers)]
    for kv in args.knob:
        cmd += ["--knob", kv]
    return cmd


def training_child_env(environ):
    """The child's rendezvous: same rank / world / address, the NEXT port (the parent's store may still hold its port)."""
    env = dict(environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 1)
    # under torch.distributed.run the workers are CLIENTS of the agent's store (TORCHELASTIC_USE_AGENT_STORE=True) at MASTER_PORT; nobody serves
    # the next port, so the children's rank 0 must host their store itself (first form of this leg: the child waited for a server forever)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("RANK", "0")
    env.setdefault("LOCAL_RANK", "0")
    env.setdefault("WORLD_SIZE", "1")
    return env


def parse_training_child(rc, stdout, timed_out=False):
    """The `train` object from a child's outcome: its JSON line's `train` field, or {"error": ...} (non-zero exit, no line, timeout)."""
    if timed_out:
        return {"error": "training leg (child process) exceeded its time limit and was killed"}
    line = next((l for l in reversed(stdout.splitlines()) if l.startswith("{")), None)
    if rc != 0 or line is None:
        return {"error": f"training leg (child process) failed: exit code {rc}" + ("" if line else ", no JSON line")}
    try:
        j = json.loads(line)
        return dict(j["train"], phase="second, guarded phase: child processes with their own RCCL process group")
    except (ValueError, KeyError) as e:
        return {"error": f"training leg (child process): unreadable line ({e})"}


def run_training_child(args, rank, world, limit_s=600.0):
    import subprocess
    cmd, env = training_child_command(args, args.train_steps), training_child_env(os.environ)
    log(f"rank {rank}: training leg in a child process (MASTER_PORT {env['MASTER_PORT']})")
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, timeout=limit_s)
        return parse_training_child(r.returncode, r.stdout)
    except subprocess.TimeoutExpired:
        return parse_training_child(-1, "", timed_out=True)


def fp16_fallbacks():
    import ctypes
    L = engine.lib()
    L.fc_debug_fp16_fallbacks.restype = ctypes.c_int64
    return int(L.fc_debug_fp16_fallbacks())


def train_steps(args, cfg, md, batch, eps, dist, world, rank, dev, steps, warmup):
    """W warm-up + K timed training steps per rank on its own scenes (weak scaling), barrier + synchronize on both sides, MAX over ranks;
    the one exchange step is the bucketed RCCL SUM all-reduce of the gradients (flowcompare_amd/shard.py).  Returns the result dict."""
    md["flow"].train()
    md["input_embedder"].train()
    params = [p for p in md["parameters"] if p.requires_grad]
    reducer = shard.GradientReducer(params)
    reducer.time_exposed = True
    opt = shard.FlatAdam(reducer, lr=1e-5)                # clip_grad_norm_ + Adam as HIP kernels on the reducer's flat buffers
    B, N = batch[1].shape[0], batch[1].shape[1]
    n_global = world * B * N
    loss = norm = None
    fb0 = fp16_fallbacks()
    try:
        for i in range(warmup):
            loss, lp, bpd, norm = shard.local_training_step(batch, n_global, md, cfg, reducer, optimizer=opt, eps=eps)
            torch.cuda.synchronize()
            log(f"rank {rank}: training warmup step {i}: loss {float(loss):.4f} |grad| {float(norm):.3e} peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
        dist.barrier()
        torch.cuda.synchronize()
        reducer.exposed_events = []
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, lp, bpd, norm = shard.local_training_step(batch, n_global, md, cfg, reducer, optimizer=opt, eps=eps)
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor(dt, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    finally:
        reducer.remove()
        md["flow"].eval()
        md["input_embedder"].eval()
    n_par = sum(p.numel() for p in reducer.params)
    exposed = reducer.exposed_ms()
    return {"all_reduce_exposed_ms_per_step": exposed,
            "ms_per_step": dt / steps * 1e3, "points_per_sec": n_global * steps / dt, "steps": steps, "warmup": warmup,
            "loss": float(loss), "grad_norm": float(norm), "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2**30,
            "fp16_fallbacks": fp16_fallbacks() - fb0,
            "what": "forward + backward (HIP training kernels) + bucketed RCCL gradient all-reduce + clip_grad_norm_ + Adam, embedder in train() mode "
                    "(train.py:108-120)",
            "gradient_all_reduce": f"{len(reducer.buckets)} buckets over {n_par * 4 / 2**20:.0f} MiB (RCCL), world {world}"}


def train_main(args, cfg, md, batch, eps, dist, world, rank, dev):
    """bench.py --train: the training step as its own JSON line."""
    r = train_steps(args, cfg, md, batch, eps, dist, world, rank, dev, args.steps, args.warmup)
    B, N = batch[1].shape[0], batch[1].shape[1]
    if rank == 0:
        print(json.dumps({
            "metric": "points/sec (training step: forward + backward + RCCL gradient all-reduce + clip + Adam, all HIP kernels)", "value": r["points_per_sec"],
            "unit": "points/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16x2 split (fp32-equivalent operands, f32 accumulate) / f32",
            "data": "synthetic (conditioned random-init weights)", "weights": args.weights, "fp16_fallbacks": r["fp16_fallbacks"],
            "config": {"workload": f"{args.config}: batch {B} scenes/GPU x {N} target + {N} context points, {cfg['n_flow_layers']} flow layers, embedder trained",
                       "global_batch": world * B, "points_per_scene": N,
                       "parallelism": f"scene-sharded x{world}; gradient all-reduce: {r['gradient_all_reduce']}"},
            "loss": r["loss"], "grad_norm": r["grad_norm"], "peak_mem_GiB": r["peak_mem_GiB"], "train": r}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2_dgcnn_attn_spline")
    ap.add_argument("--batch", type=int, default=16, help="scenes per GPU")
    ap.add_argument("--points", type=int, default=4096, help="target points per scene (and context points, unless --ctx-points)")
    ap.add_argument("--ctx-points", type=int, default=None,
                    help="context points per scene when they differ from the target points: the reference's native training shape is 20 scenes x 1024 "
                         "target x 1250 context points (config/dulcet-universe.yaml:1-3,44-46,185-187): --config c4_dgcnn_attn_extra_affine --batch 20 "
                         "--points 1024 --ctx-points 1250")
    ap.add_argument("--fixed-eps", action="store_true",
                    help="diagnostic: draw the augmenter's noise once before the timed region (rounds 1-3).  Default: every step draws its own eps on the "
                         "device inside the timed region, as the reference does in every forward (models/augmenter.py:49-63)")
    ap.add_argument("--train-in-child", action="store_true", help="run the training leg as the guarded second phase also with one rank (what more than one rank always does)")
    ap.add_argument("--train-child", action="store_true", help=argparse.SUPPRESS)      # (internal: the training leg of a multi-rank run, as a child process)
    ap.add_argument("--layers", type=int, default=None, help="override n_flow_layers (INVALID as a headline number)")
    ap.add_argument("--weights", choices=("conditioned", "module"), default="conditioned",
                    help="conditioned (default): module init + flowcompare_amd.conditioning.condition_flow (near-identity coupling output layers, "
                         "LinearLU mixing, ActNorm first-batch statistics) -- the state the full-depth parity tests gate at 1e-4 bpd; "
                         "module: the constructors' init as it is (ill-conditioned at 115 layers: diagnostic only)")
    ap.add_argument("--train", action="store_true",
                    help="time the TRAINING step instead (SURVEY.md 8f N1): forward + backward on the HIP training kernels, bucketed RCCL "
                         "gradient all-reduce overlapped with backward, clip_grad_norm_, Adam; embedder in train() mode.  Prints its own JSON line "
                         "(metric 'points/sec (training step ...)'); the default forward metric is BASELINE.json's")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=None,
                    help="timed training steps of the `train` object that the default (forward) line carries: after the forward region, one warm-up + "
                         "this many timed training steps of the same workload (0 = leave the object out).  Default: 2 on one GPU; 0 under "
                         "torch.distributed.run with more than one rank -- the multi-rank RCCL gradient exchange has only ever run with one rank on "
                         "hardware, and a crash there must not take the forward scaling line with it (`--train` or an explicit --train-steps measure it)")
    ap.add_argument("--sync-range-check", action="store_true",
                    help="diagnostic: the round-2 behaviour -- every forward call reads its fp16 range flag back (one stream synchronisation per call) "
                         "instead of the deferred check (fc_range_check_defer: K forwards queued back to back, flags read once at the end of the timed region)")
    ap.add_argument("--no-profile", action="store_true", help="diagnostic: leave the in-library HIP-event profiler off in the timed region")
    ap.add_argument("--knob", action="append", default=[], help="K=V tuning knob for same-box A/B runs (fc_debug_set); not for headline numbers")
    ap.add_argument("--cpu-points", type=int, default=None, help="points per scene of the CPU sample (default: same as --points)")
    args = ap.parse_args()
    for kv in args.knob:
        k, v = kv.split("=")
        from flowcompare_amd import engine as _eng
        if _eng.lib().fc_debug_set(int(k), int(v)) != 0:
            raise SystemExit(f"unknown --knob {kv}")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.train_steps is None:
        args.train_steps = 2                              # (with more than one rank the leg runs as a second, guarded phase in child processes: below)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:             # under torch.distributed.run (also with one rank: exercises the RCCL path)
        import torch.distributed as dist
        init_rccl(dist, dev)
    elif args.train or args.train_steps > 0:          # the training step's exchange is RCCL's also on one GPU (a one-rank group)
        import socket
        import torch.distributed as dist
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        init_rccl(dist, dev, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)

    census = None
    if dist is not None:
        census = shard.device_census(device_identity(dev))
        if census["unique_devices"] != world:
            log(f"rank {rank}: WARNING {world} ranks on {census['unique_devices']} distinct devices")

    over = {"sample_size": args.points}
    if args.layers:
        over["n_flow_layers"] = args.layers
    cfg = fa.named_config(args.config, **over)
    torch.manual_seed(0)                                  # same random-init weights on every rank
    log(f"rank {rank}: building {args.config} ({cfg['n_flow_layers']} layers) ...")
    with contextlib.redirect_stdout(sys.stderr):          # the reference API prints its parameter count: stdout carries only the JSON line
        md = fa.initialize_flow(cfg, device=dev, mode="test")
    B, N = args.batch, args.points
    M = args.ctx_points or N
    if args.weights == "conditioned":
        # the same two conditioning scenes on every rank -> identical weights on every rank (kernel work does not depend on the values)
        from flowcompare_amd.conditioning import condition_flow
        c0, c1, cx, cg = synth_pairs(2, M, N, 999, dev)
        ceps = [torch.randn(2, N, cfg["latent_dim"] - cfg["input_dim"], generator=cg).to(dev)]
        t_c = time.perf_counter()
        condition_flow(md, cfg, (c0, c1, cx if cfg["extra_z_value_context"] else None), eps=ceps)
        torch.cuda.synchronize()
        log(f"rank {rank}: weights conditioned in {time.perf_counter() - t_c:.1f} s")
        del c0, c1, cx, ceps
    e0, e1, extra, g = synth_pairs(B, M, N, 1000 + rank, dev)   # every rank owns different scenes
    batch = (e0, e1, extra if cfg["extra_z_value_context"] else None)
    eps = [torch.randn(B, N, cfg["latent_dim"] - cfg["input_dim"], generator=g).to(dev)]
    torch.cuda.manual_seed(4000 + rank)

    if args.train or args.train_child:
        return train_main(args, cfg, md, batch, eps, dist, world, rank, dev)

    def step():
        # the augmenter's noise is part of a forward (models/augmenter.py:49-63: drawn in every call): one device randn of B x N x 294 per step
        e = eps if args.fixed_eps else [torch.randn(B, N, cfg["latent_dim"] - cfg["input_dim"], device=dev)]
        loss, lp, bpd = fa.inner_loop(batch, md, cfg, eps=e)
        if dist is not None:                              # global mean over all ranks' scenes: the only exchange of the path
            loss, bpd = shard.global_loss_bpd(lp, cfg["input_dim"])
        return loss, lp, bpd

    t_build = time.perf_counter()
    md["input_embedder"]._engine() if hasattr(md["input_embedder"], "_engine") else None
    md["flow"]._engine()                                  # weight folding / packing / upload (one-time, not timed)
    log(f"rank {rank}: engine packed in {time.perf_counter() - t_build:.1f} s; warmup ...")
    # Every launch of the LAST warmup step is bracketed by HIP events: that gives the per-kernel breakdown and names the dominant
    # kernel.  In the timed region only that kernel's launches are bracketed (two event records cost ~4 us of stream time each:
    # bracketing all ~1000 launches of a step takes 3 % off the throughput being measured).
    warm_prof = []
    for i in range(args.warmup):
        last = i == args.warmup - 1 and not args.no_profile
        if last:
            engine.profile_filter(None); engine.profile_reset(); engine.profile_enable(True)
        loss, lp, bpd = step()
        torch.cuda.synchronize()
        if last:
            engine.profile_enable(False)
            warm_prof = engine.profile_report()
        log(f"rank {rank}: warmup step {i} done")
    dom_warm = max(warm_prof, key=lambda p: p["ms"]) if warm_prof else None
    dominant = dom_warm["kernel"] if dom_warm else None
    # ... and of those at most ~32 per step: a launch-bound forward (C1: 575 launches of the dominant kernel in 13 ms) runs 30 % slower with
    # every one of them bracketed, and a sample gives the same average launch duration
    stride = max(1, -(-dom_warm["launches"] // 32)) if dom_warm else 1

    engine.profile_reset()
    engine.profile_filter(dominant)
    engine.profile_stride(stride)
    engine.profile_enable(not args.no_profile)
    fb0 = fp16_fallbacks()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # Deferred range check (include/fcflow.h fc_range_check_defer): the K forwards are queued back to back on the stream, each followed by a
    # 4-byte copy of its split-fp16 range flag into pinned host memory; leaving the block reads the flags in order (and would repeat an
    # out-of-range pass on the bf16-limb loops) -- INSIDE the timed region, so the K steps are complete and checked when the clock stops.
    with (contextlib.nullcontext() if args.sync_range_check else engine.deferred_range_check()) as drc:
        for _ in range(args.steps):
            loss, lp, bpd = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    repeated = drc.repeated if drc is not None else 0
    if repeated:                                          # outputs were rewritten in place by the repeats: recompute the scalars from them
        loss = -lp.mean()
        bpd = loss * math.log2(math.e) / cfg["input_dim"]
    fallbacks = fp16_fallbacks() - fb0
    engine.profile_enable(False)
    engine.profile_filter(None)
    engine.profile_stride(1)
    prof = engine.profile_report()
    log(f"rank {rank}: {args.steps} timed steps in {dt:.3f} s ({fallbacks} passes repeated on the bf16-limb loops)")
    if dist is not None:
        t = torch.tensor(dt, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    mean_nats, bpd_f = float(-loss), float(bpd)
    sd = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # the weights the forward ran on (the training leg below updates them)
        sd = ({k: v.detach().cpu().clone() for k, v in md["flow"].state_dict().items()},
              {k: v.detach().cpu().clone() for k, v in md["input_embedder"].state_dict().items()})

    out = None
    if rank == 0 and args.no_profile:
        print(json.dumps({"diagnostic": "profiler off", "value": world * B * N * args.steps / dt, "ms_per_step": dt / args.steps * 1e3}), flush=True)
    elif rank == 0:
        total_pts = world * B * N * args.steps
        value = total_pts / dt
        prof.sort(key=lambda p: -p["ms"])
        dom = prof[0]                                       # live HIP-event timing of the dominant kernel over the timed region
        breakdown, bsteps = (warm_prof, 1) if warm_prof else (prof, args.steps)      # all kernels: last warmup step
        breakdown.sort(key=lambda p: -p["ms"])
        tot_ms = sum(p["ms"] for p in breakdown) / bsteps * args.steps or 1.0
        workload_key = f"{args.config} {B} x {N} + {M}" + (f" ({args.layers} layers)" if args.layers else "")
        traffic, busy, why = pmc_tables(workload_key)
        roof = roofline_of(dom, traffic, busy, why)
        roof.update({"frac_note": "frac = algorithmic fp32-equivalent FLOPs (SURVEY.md 8d) / peak; frac_issued = MFMA FLOPs issued / the same peak (issue rate)",
                     "launches_bracketed": f"every {stride}th launch of this kernel in the timed region" if stride > 1 else "all",
                     "share_of_gpu_time": dom_warm["ms"] / sum(p["ms"] for p in warm_prof) if warm_prof else dom["ms"] / tot_ms,
                     "flops_counted": "useful multiply-adds of the launches (padding excluded), HIP events on the launch stream"})
        alg = ALG_MFLOP_PER_POINT.get((args.config, N))
        out = {
            "metric": "nats/sec (forward log-prob) on 4096-pt coloured pairs", "value": value, "unit": "nats/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16x2 split (fp32-equivalent operands, f32 accumulate) / f32", "data": "synthetic" + (" (conditioned random-init weights, flowcompare_amd/conditioning.py)" if args.weights == "conditioned" else " (module-init weights)"),
            "config": {"workload": f"{args.config}: batch {B} scenes/GPU x {N} target + {M} context points, "
                                   f"{cfg['n_flow_layers']} flow layers ({cfg['flow_type']}), embedder {cfg['input_embedder']}",
                       "global_batch": world * B, "points_per_scene": N, "context_points_per_scene": M, "parallelism": f"scene-sharded x{world}, no data-path collective"},
            "mean_nats": mean_nats, "bpd": bpd_f, "weights": args.weights, "fp16_fallbacks": fallbacks,
            "eps": "drawn once before the timed region (--fixed-eps)" if args.fixed_eps else "one device randn per step inside the timed region (models/augmenter.py:49-63)",
            "range_check": "per call (stream synchronisation)" if args.sync_range_check else "deferred: flags of the K queued forwards read at the end of the timed region (fc_range_check_defer)",
            "job_algorithmic_tflops": None if alg is None or args.layers else alg * 1e6 * value / 1e12,
            "rccl": None if census is None else dict(census, backend="nccl (RCCL)", note="all_gather of every rank's device UUID / PCI address"),
            "roofline": roof,
            "kernels_source": "HIP events around every launch of the last warmup step" if warm_prof else "timed region",
            # every kernel of a step with its own roofline entry (frac / frac_issued / traffic / matrix-pipe busy where a counter pass of this workload is committed)
            "kernels": [dict(roofline_of(p, traffic, busy, why), ms_per_step=p["ms"] / bsteps, launches=p["launches"],
                             tflops=(p["flops"] / (p["ms"] * 1e-3) / 1e12) if p["flops"] else None,
                             gbs=(p["bytes"] / (p["ms"] * 1e-3) / 1e9) if p["bytes"] else None) for p in breakdown[:10]],
        }

    # `train` object: the training step of the same workload on the same clock (`bench.py --train` prints it as its own
    # line).  Every rank takes part; the forward numbers are complete before it starts and are printed even if the leg fails or hangs.
    #   one rank: in this process, behind a watchdog;
    #   more than one rank (or --train-in-child): a SECOND, GUARDED PHASE -- the multi-rank RCCL gradient exchange runs in fresh child processes
    #   (one per rank, their own process group on MASTER_PORT + 1), after this process has released its device memory and left its process
    #   group: a crash, an abort inside RCCL or a hang there cannot take the forward line with it, and the line then carries train.error and
    #   the run exits non-zero.  (A child is started with subprocess; nothing is exec'ed from a process that has touched the GPU.)
    exit_code = 0
    if args.train_steps > 0 and not args.no_profile and (world > 1 or args.train_in_child):
        del step
        md = batch = eps = e0 = e1 = extra = lp = loss = bpd = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
            dist = None
        train_obj = run_training_child(args, rank, world)
        if "error" in train_obj:
            exit_code = 5
        if out is not None:
            out["train"] = train_obj
    elif args.train_steps > 0 and not args.no_profile:
        import threading
        done = threading.Event()

        def watchdog():                                   # a leg that hangs (a collective that never completes) must not take the forward line with it
            if not done.wait(240.0):
                log(f"rank {rank}: training leg exceeded 240 s -- abandoned")
                if out is not None:
                    out["train"] = {"error": "training leg abandoned after 240 s"}
                    print(json.dumps(out), flush=True)
                os._exit(4)                               # non-zero on EVERY rank: the forward numbers are in the JSON line, not in the exit status
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            torch.cuda.reset_peak_memory_stats()
            train_obj = train_steps(args, cfg, md, batch, eps, dist, world, rank, dev, args.train_steps, 1)
        except Exception as e:                            # noqa: BLE001 -- reported in the line, the forward numbers stand
            train_obj = {"error": f"{type(e).__name__}: {e}"[:300]}
            log(f"rank {rank}: training leg failed: {e}")
            exit_code = 5
        done.set()
        if out is not None:
            out["train"] = train_obj
    if out is not None:
        if world == 1 and not args.no_cpu_baseline:
            cb, _ = cpu_baseline(cfg, md, args.cpu_points or N, 1000, sd)
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = out["value"] / cb["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
