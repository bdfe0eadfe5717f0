"""CPU, world_size 2 (gloo): clip sharding + result gather used by bench.py / the launcher for N > 1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "oakink2-tamf_amd")]
    from oakink2_tamf_amd import shard
    from oracle import mdm_oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # equal shards: each rank "samples" its clips (stand-in: Philox draw keyed by global clip id)
        per = 3
        base = shard.clip_id_base(rank, per)
        local = torch.from_numpy(O.philox_normal(5, np.arange(base, base + per), 1, 99, 8))
        full = shard.gather_clips(local)
        ref = torch.from_numpy(O.philox_normal(5, np.arange(0, world * per), 1, 99, 8))
        ok1 = torch.equal(full, ref)
        # ragged contiguous split of 7 items (launch/sample.py:198-199)
        n = 7
        counts = [shard.worker_range(n, r, world)[1] - shard.worker_range(n, r, world)[0] for r in range(world)]
        s0, s1 = shard.worker_range(n, rank, world)
        loc = torch.arange(s0, s1, dtype=torch.float32).view(-1, 1) * torch.ones(1, 4)
        got = shard.gather_ragged(loc, counts)
        ok2 = torch.equal(got[:, 0], torch.arange(n, dtype=torch.float32))
        q.put((rank, ok1, ok2))
    finally:
        dist.destroy_process_group()


def test_worker_range_partitions_like_the_reference():
    from oakink2_tamf_amd import shard

    for n in (0, 1, 7, 64, 1001):
        for w in (1, 2, 3, 8):
            rs = [shard.worker_range(n, i, w) for i in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))


def test_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res == [(0, True, True), (1, True, True)]


def test_gather_force_collective_world1_gloo():
    """A group of ONE rank normally short-circuits to a copy; force_collective runs the real all_gather_into_tensor (the GPU
    suite uses this to put the RCCL call on a device with a single GPU, tests/test_hip_rccl.py)."""
    from oakink2_tamf_amd import shard

    assert not dist.is_initialized()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        x = torch.arange(2 * 3 * 4, dtype=torch.float32).view(2, 3, 4)
        out = torch.empty_like(x)
        got = shard.gather_clips(x, out, force_collective=True)
        assert got is out and torch.equal(out, x)
        assert torch.equal(shard.gather_clips(x, force_collective=True), x)
    finally:
        dist.destroy_process_group()
