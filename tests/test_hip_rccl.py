"""GPU: the multi-GPU path's collective on a device.  The driver's 8-GPU box is the first place world_size > 1 runs on
hardware; this puts the SAME calls - init_process_group("nccl", device_id=...), shard.gather_clips ->
dist.all_gather_into_tensor, the barrier and the MAX all_reduce of bench.py - on cuda:0 in a group of one rank, and runs
bench.py's own main() on the config 3 / config 5 presets (reference split: launch/sample.py:198-199,264-292)."""
import io
import json
import os
import socket
import sys
from contextlib import redirect_stdout

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture()
def nccl_world1():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        yield dev
    finally:
        dist.destroy_process_group()


def test_rccl_all_gather_of_the_bench_result_tensor(nccl_world1):
    from oakink2_tamf_amd import shard

    dev = nccl_world1
    g = torch.Generator().manual_seed(3)
    local = torch.randn(64, 99, 1, 196, generator=g).to(dev)  # one rank's sampled poses at the bench shape (5 MB)
    out = torch.empty_like(local)
    got = shard.gather_clips(local, out, force_collective=True)  # dist.all_gather_into_tensor over RCCL
    dist.barrier()
    torch.cuda.synchronize(dev)
    assert got is out and torch.equal(out, local)
    # ragged gather + the timing reduction of bench.py
    assert torch.equal(shard.gather_ragged(local[:7], [7]), local[:7])
    te = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(te, op=dist.ReduceOp.MAX)
    assert float(te.item()) == 1.25


@pytest.mark.parametrize("config,dtype,batch", [(3, "f16x3", 32), (5, "bf16", 64)])
def test_bench_presets_execute_on_the_device(config, dtype, batch):
    """bench.py's config 3 (32 clips per GPU) and config 5 (bf16) presets through its own main(), 40 DDPM steps per loop:
    the JSON line is well-formed, the samples are finite and the in-run oracle check is inside the per-dtype tolerance."""
    sys.path.insert(0, ROOT)
    import bench

    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = bench.main(["--gpus", "1", "--config", str(config), "--steps", "1", "--warmup", "1", "--ddpm-steps", "40",
                         "--no-cpu-baseline", "--also", ""])
    assert rc == 0
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    assert line["dtype"] == dtype and line["config"]["clips_per_gpu"] == batch and line["config"]["preset"] == config
    assert line["finite"] is True and line["check_ok"] is True
    assert line["roofline"]["frac"] > 0 and line["value"] > 0
