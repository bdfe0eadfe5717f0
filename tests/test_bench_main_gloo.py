"""CPU, world_size 2 (gloo): bench.py's own multi-rank path - self-spawned ranks, process-group set-up, per-rank clip
ranges, the result gather inside the timed region, max-over-ranks timing and the single JSON line - driven through
`python bench.py --gpus 2` exactly as on a GPU node, with the HIP sampler replaced by tests/bench_stub.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, sampler="bench_stub:StubSampler"):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if env_extra:
        env.update(env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--sampler", sampler,
           "--arch", "arch_mdm", "--batch", "3", "--frames", "8", "--ddpm-steps", "4", "--steps", "2", "--warmup", "1"] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)


def test_bench_self_spawns_two_ranks_and_gathers():
    r = _run(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly one JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["world_size_seen"] == 2
    assert j["config"]["rank_clip_ranges"] == [[0, 3], [3, 6]]
    assert j["config"]["global_clips"] == 6 and j["finite"] is True
    assert j["value"] > 0 and abs(j["value"] - 6 * 8 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]


def test_bench_world_8_as_the_scaling_run_will_launch_it():
    """The driver's 8-GPU form, `--gpus 8` (SCALE_rNN.json), on CPU: eight self-spawned ranks, eight contiguous clip ranges, one gather
    per loop, one JSON line whose value counts the clips of ALL ranks - the builder never holds more than one GPU, so this is the
    closest rehearsal of the first 8-GPU contact that can run before it."""
    r = _run(["--gpus", "8", "--config", "3", "--batch", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["world_size_seen"] == 8 and j["config"]["preset"] == 3
    assert j["config"]["rank_clip_ranges"] == [[2 * k, 2 * k + 2] for k in range(8)] and j["config"]["global_clips"] == 16
    ranks = j["config"]["ranks"]
    assert [i["rank"] for i in ranks] == list(range(8)) and len({i["pid"] for i in ranks}) == 8 and all(i["finite"] for i in ranks)
    assert abs(j["value"] - 16 * 8 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
    assert "cpu_baseline" not in j and "other_dtypes" not in j  # N > 1: no per-point host measurement, no extra modes


def test_a_dead_rank_ends_its_siblings():
    """spawn_ranks watches its children: rank 1 exits with 7 before the process-group set-up, rank 0 - which would sit in the
    rendezvous until its timeout - is terminated, and the launcher returns rank 1's code within seconds."""
    import time

    t0 = time.monotonic()
    r = _run(["--gpus", "2"], sampler="bench_stub_dies:StubSampler")
    assert r.returncode == 7, (r.returncode, r.stderr[-1500:])
    assert time.monotonic() - t0 < 120
    assert "terminating the other ranks" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no JSON line from a failed run


def test_sampler_hook_needs_the_gloo_backend():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sampler", "bench_stub:StubSampler"], capture_output=True, text=True,
                       env=dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")])), timeout=120)
    assert r.returncode != 0 and "needs --backend gloo" in r.stderr


def test_bench_under_an_external_launcher_env():
    """RANK / WORLD_SIZE / MASTER_* from the environment (the torch.distributed.run contract), world size 1."""
    r = _run(["--gpus", "1"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29511"})
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["config"]["rank_clip_ranges"] == [[0, 3]]


def test_bench_presets():
    sys.path.insert(0, ROOT)
    import bench

    a = bench.parse_args(["--config", "3"])
    assert a.batch == 32 and a.dtype == bench.DEFAULT_DTYPE
    a = bench.parse_args(["--config", "5"])
    assert a.batch == 64 and a.dtype == "bf16"
    a = bench.parse_args([])
    assert a.batch == 64 and a.frames == 196 and a.ddpm_steps == 1000 and a.gpus == 1 and a.steps == 3 and a.warmup == 1
    a = bench.parse_args(["--config", "4"])  # BASELINE configs[3]: the R trunk, one forward per step
    assert a.arch == "arch_refine" and a.batch == 64 and a.frames == 196 and a.steps == 200 and a.warmup == 20
    a = bench.parse_args(["--config", "4", "--steps", "7", "--warmup", "2"])
    assert (a.steps, a.warmup) == (7, 2)
    import pytest

    with pytest.raises(SystemExit):
        bench.parse_args(["--arch", "arch_refine"])
    assert bench.flops_per_clip_step(bench.ARCHS["arch_mdm_l"], 196) == 11127660544  # 11.128 GF per clip-step


def test_check_ok_logic():
    sys.path.insert(0, ROOT)
    import bench

    good = dict(finite_by={"f16x3": True, "f32": True}, range_flags={"f16x3": False}, check={"f16x3": 4.4e-6, "f32": 2.2e-6, "bf16": 1e-2})
    assert bench.checks_ok(**good)
    assert not bench.checks_ok(**dict(good, finite_by={"f16x3": True, "f32": False}))
    assert not bench.checks_ok(**dict(good, range_flags={"f16x3": True}))
    assert not bench.checks_ok(**dict(good, check={"f16x3": 2e-5}))          # outside the 1e-5 gate
    assert not bench.checks_ok(**dict(good, check={"bf16x3": float("nan")}))  # NaN never passes
    assert bench.checks_ok(finite_by={"bf16": True}, range_flags={}, check={})
