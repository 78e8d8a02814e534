"""N > 1 path on CPU: world_size-2 gloo processes exercising the scene sharding + the scalar loss reduction + the log-prob
gather.  The per-rank compute is the pinned oracle here (this container has no GPU); on the GPU box the same helpers wrap
the HIP engine (bench.py)."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import Fixture
from flowcompare_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import flow_oracle as O
        fx = Fixture("e2e_tiny_affine")                       # B = 3 scenes: uneven split 2 + 1
        cfg = fx.derived_cfg()
        sd_flow, sd_emb = fx.state_dicts(torch.float64)
        batch = (fx.t("extract_0", torch.float64), fx.t("extract_1", torch.float64), fx.t("extra", torch.float64))
        eps = fx.eps(torch.float64)
        lo, hi = shard.shard_bounds(3, rank, world)
        local = shard.shard_batch(batch, rank, world)
        assert local[0].shape[0] == hi - lo
        with torch.no_grad():
            _, lp, _ = O.inner_loop(cfg, sd_flow, sd_emb, local, [e[lo:hi] for e in eps])
        loss, bpd = shard.global_loss_bpd(lp, cfg["input_dim"])
        full = shard.gather_log_prob(lp, 3)
        q.put((rank, float(loss), float(bpd), full.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_reproduces_single_process_result():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    fx = Fixture("e2e_tiny_affine")
    for rank, loss, bpd, full in res:
        assert abs(loss - float(fx.a["loss_f64"])) < 1e-9 and abs(bpd - float(fx.a["bpd_f64"])) < 1e-10
        assert abs(full - fx.a["log_prob_f64"]).max() < 1e-8


def test_shard_bounds_cover_batch_exactly():
    for B in (1, 2, 3, 16, 17, 64, 128):
        for G in (1, 2, 3, 4, 8):
            spans = [shard.shard_bounds(B, r, G) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


# ---------------------------------------------------------------- training: sharded backward + bucketed gradient all-reduce
def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import flow_oracle as O
        fx = Fixture("e2e_tiny_affine")                       # 3 scenes: uneven split 2 + 1
        cfg = fx.derived_cfg()
        sd_flow, sd_emb = fx.state_dicts(torch.float64)
        params = []
        for sd in (sd_flow, sd_emb):
            for v in sd.values():
                if v.is_floating_point():
                    v.requires_grad_(True)
                    params.append(v)
        reducer = shard.GradientReducer(params, bucket_bytes=64 << 10)          # small buckets: several collectives in flight
        assert len(reducer.buckets) > 3
        batch = (fx.t("extract_0", torch.float64), fx.t("extract_1", torch.float64), fx.t("extra", torch.float64))
        lo, hi = shard.shard_bounds(3, rank, world)
        local = shard.shard_batch(batch, rank, world)
        _, lp, _ = O.inner_loop(cfg, sd_flow, sd_emb, local, [e[lo:hi] for e in fx.eps(torch.float64)])
        shard.local_loss(lp, batch[1].shape[0] * batch[1].shape[1]).backward()
        launched_during_backward = sum(x is not None for x in reducer.inflight)
        reducer.finish()
        grads = {}
        for part, sd in (("flow", sd_flow), ("input_embedder", sd_emb)):
            for n, v in sd.items():
                if v.is_floating_point() and v.grad is not None:
                    grads[f"{part}/{n}"] = v.grad.clone()
        q.put((rank, launched_during_backward, {k: v.numpy() for k, v in grads.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_backward_reproduces_the_reference_full_batch_gradients():
    """Each rank differentiates its scenes (2 + 1), the bucketed SUM all-reduce runs from post-accumulate hooks during backward, and
    every rank ends with the gradient of the GLOBAL mean loss: checked against the fixture the reference's own loss.backward()
    produced on the full batch (tests/golden/grad_tiny_affine.npz)."""
    import json
    import re

    import numpy as np
    import synth
    from conftest import GOLDEN
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    z = np.load(os.path.join(GOLDEN, "grad_tiny_affine.npz"))
    names = json.loads(bytes(z["names_json"]).decode())["eval"]
    for rank, launched, grads in res:
        assert launched > 0                                    # buckets started while backward was still running
        for key in names:
            part, n = key.split("/", 1)
            alias = part + "/" + re.sub(r"^bn(\d)\.", r"conv\1.1.", n)
            g = torch.from_numpy(grads[alias]).double().reshape(-1)
            want = z["eval/" + key]
            r = torch.from_numpy(synth.normal("gradproj/" + key, (g.numel(),), 0))
            got = np.array([g.sum().item(), g.abs().sum().item(), (g * r).sum().item()])
            assert np.abs(got - want[:3]).max() < 1e-8 * max(want[1], 1e-9 * float(z["eval/grad_norm"])) + 1e-12, (rank, key)
    for k in res[0][2]:
        assert np.array_equal(res[0][2][k], res[1][2][k])      # both ranks hold identical reduced gradients


def _bn_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bn = torch.nn.BatchNorm1d(3)
        holder = torch.nn.Module()
        holder.bn1 = bn
        holder.alias = torch.nn.Sequential(bn)               # the embedders register every BatchNorm under two names
        with torch.no_grad():
            bn.running_mean.copy_(torch.tensor([1.0, -2.0, 0.5]) * (rank + 1))
            bn.running_var.copy_(torch.tensor([0.5, 1.0, 2.0]) + rank)
        holder.eval()
        before = bn.running_mean.clone()
        shard.sync_batchnorm_buffers(holder)                 # eval mode: the buffers did not move, nothing is exchanged
        assert torch.equal(bn.running_mean, before)
        holder.train()
        shard.sync_batchnorm_buffers(holder)
        q.put((rank, bn.running_mean.clone().numpy(), bn.running_var.clone().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_batchnorm_running_statistics_combine_as_one_population():
    """sync_batchnorm_buffers: mean of the ranks' means; variance E[var + mean^2] - mean^2, i.e. the spread of the ranks' means counts."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=100) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    m = [torch.tensor([1.0, -2.0, 0.5]) * (r + 1) for r in range(2)]
    v = [torch.tensor([0.5, 1.0, 2.0]) + r for r in range(2)]
    mean = (m[0] + m[1]) / 2
    var = (v[0] + m[0] ** 2 + v[1] + m[1] ** 2) / 2 - mean ** 2
    for _, rm, rv in res:
        assert torch.allclose(torch.from_numpy(rm), mean, atol=1e-6) and torch.allclose(torch.from_numpy(rv), var, atol=1e-6)


def _mask_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b, c = (torch.nn.Parameter(torch.ones(3) * (i + 1)) for i in range(3))
        red = shard.GradientReducer([a, b, c], bucket_bytes=16)
        seen = []
        # step 1: c receives a gradient on rank 1 only; steps 2, 3: on no rank -- rank 0's OWN pattern (a, b) never changes
        for step in range(3):
            red.zero_grad()
            loss = (a * a).sum() + (b * 2.0).sum()
            if rank == 1 and step == 0:
                loss = loss + (c * 3.0).sum()
            loss.backward()
            red.finish()
            seen.append([p.grad is None for p in (a, b, c)])
            if not seen[-1][2]:
                assert torch.allclose(c.grad, torch.full((3,), 3.0))          # the SUM over ranks: 0 + 3
        red.remove()
        q.put((rank, seen))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gradient_presence_mask_follows_the_other_rank():
    """A parameter that loses its gradient on the ONLY rank that had one must become `.grad = None` on every rank in that same step, also on
    the ranks whose own presence pattern did not change (round-3 advisor finding: a cache keyed on the local pattern kept the stale list
    there, and FlatAdam then updated the parameter on some replicas only)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mask_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=100) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, seen in res:
        assert seen == [[False, False, False], [False, False, True], [False, False, True]], (rank, seen)


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import flowcompare_amd as fa
        from flowcompare_amd import model_initialization as MI
        fx = Fixture("e2e_tiny_affine")
        cfg = dict(fx.cfg)
        cfg["data_parallel"] = True
        md = fa.initialize_flow(cfg, device="cpu", mode="test")
        seen = {}

        def fake_inner(batch, models_dict, config, eps=None):          # what the rank would hand its engine: record the shard, return a log-prob
            seen["scenes"] = batch[0].shape[0]
            lp = torch.full((batch[1].shape[0], batch[1].shape[1]), float(rank + 1))
            return -lp.mean(), lp, torch.tensor(0.0)
        real = MI.inner_loop
        batch = (fx.t("extract_0"), fx.t("extract_1"), fx.t("extra"))   # B = 3 scenes: 2 + 1
        import unittest.mock as mock
        with mock.patch.object(MI, "inner_loop", side_effect=lambda *a, **k: fake_inner(*a, **k) if md.get("_in_shard") else real(*a, **k)):
            loss, lp, _ = real(batch, md, cfg)
        q.put((rank, bool(md.get("sharded")), seen["scenes"], float(loss), tuple(lp.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_data_parallel_config_routes_inner_loop_to_the_scene_shards():
    """config['data_parallel']: under a process group initialize_flow marks the model dict as sharded and inner_loop takes the GLOBAL batch,
    runs this rank's scenes (2 + 1 of 3) and returns the global loss: -(2 scenes x 1.0 + 1 scene x 2.0) / 3 per point."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=100) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True] and [r[2] for r in res] == [2, 1]
    for _, _, _, loss, _ in res:
        assert abs(loss - (-(2 * 1.0 + 1 * 2.0) / 3)) < 1e-6
