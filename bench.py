#!/usr/bin/env python3
"""bench.py - sampled motion frames/s of the MF-MDM 1000-step DDPM sampler on MI355X.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is ONE complete reverse loop (n_ddpm = 1000 DDPM steps, hipGraph-replayed) over one batch of synthetic
clips: the workload of BASELINE.json configs[1] - arch_mdm_l, B = 64 clips per GPU, T = 196 frames, synthetic
CLIP / object conditioning, device-Philox noise keyed by global clip id.  Clips are independent, so ranks shard them
with no data-path collective (weak scaling: 64 clips per GPU); the only exchange is the RCCL all_gather of the
sampled poses at the end of every loop, which is inside the timed region.

value = (clips of all ranks) * T * K / (max-over-ranks wall time of the K timed loops), inputs resident in HBM.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oakink2-tamf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0}  # MI355X_MICROARCH.md: dense MFMA peaks
ARCHS = {
    "arch_mdm": dict(latent_dim=256, ff_size=1024, num_layers=8, num_heads=4),
    "arch_mdm_l": dict(latent_dim=512, ff_size=2048, num_layers=8, num_heads=4),
}


def flops_per_clip_step(arch, T):
    """SURVEY.md section 8(a)/BASELINE.md section 3: algorithmic FLOPs of one denoiser evaluation of one clip."""
    d, ff, L = arch["latent_dim"], arch["ff_size"], arch["num_layers"]
    S = T + 5
    return L * S * (8 * d * d + 4 * d * ff + 4 * S * d) + T * (4 * 99 * d + 6 * d * d) + 4 * d * d


def hbm_traffic(dtype, kernel, B, T):
    """HBM-side bytes per launch of `kernel` from the committed PMC measurement of this exact workload
    (profiles/r01/hbm_traffic_<dtype>.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH doubled
    per the gfx950 correction of MI355X_MICROARCH.md).  None when no measurement matches."""
    path = os.path.join(ROOT, "profiles", "r01", f"hbm_traffic_{dtype}.json")
    try:
        with open(path) as f:
            m = json.load(f)
        if m.get("B") == B and m.get("T") == T and kernel in m["kernels"]:
            return m["kernels"][kernel]["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def usable_cores() -> int:
    """Host cores this process may actually use: CPU affinity capped by the cgroup CPU quota
    (the GPU box exposes 256 logical CPUs but grants a 16-CPU quota; oversubscribing it is 20x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def synthetic_cond(B, T, seed, nobj=2):
    """Synthetic conditioning of SURVEY.md section 8(d): unit-normal CLIP / object embeddings and trajectories,
    alternating hand side, per-clip constant betas."""
    import torch

    g = torch.Generator().manual_seed(seed)
    return {
        "text_embedding": torch.randn(B, 512, generator=g),
        "hand_side": ["rh" if b % 2 == 0 else "lh" for b in range(B)],
        "shape": torch.randn(B, 1, 10, generator=g).repeat(1, T, 1).contiguous(),
        "obj_embedding": torch.randn(B, nobj, 768, generator=g),
        "obj_traj": torch.randn(B, nobj, T, 9, generator=g),
    }


def cpu_baseline(arch_name, sd, T, n_ddpm, sample_B=16, timed=2):
    """The oracle (torch-CPU restatement of the reference, proven equal to it on tests/golden) timed on the host
    cores with the SAME weights as the GPU run: a bounded sample of the same workload - `sample_B` clips x
    (1 warm-up + `timed`) denoiser+DDPM steps - extrapolated to the n_ddpm-step loop (every step does identical work).
    This is the only place bench.py touches oracle/."""
    import torch

    from oracle import mdm_oracle as O

    arch = {"arch_mdm": O.ARCH_MDM, "arch_mdm_l": O.ARCH_MDM_L}[arch_name]
    cores = usable_cores()
    torch.set_num_threads(cores)
    cond = synthetic_cond(sample_B, T, seed=12345)
    tab = O.make_tables(n_ddpm, "cosine")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(sample_B, 99, 1, T, generator=g)
    times = []
    with torch.no_grad():
        for it in range(1 + timed):
            i = n_ddpm - 1 - it
            t0 = time.perf_counter()
            x0 = O.denoiser_forward(sd, arch, x, torch.full((sample_B,), i, dtype=torch.long), cond)
            x = O.ddpm_step(tab, x, x0, i, torch.randn(x.shape, generator=g))
            times.append(time.perf_counter() - t0)
    step_s = sum(times[1:]) / timed
    return {
        "value": sample_B * T / (step_s * n_ddpm),
        "unit": "frames/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{sample_B} clips x T={T}, {timed} timed denoiser+DDPM steps after 1 warm-up ({step_s * 1e3:.0f} ms/step), "
        f"extrapolated to {n_ddpm} steps; conditioning recomputed every step as the reference does",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default=os.environ.get("TAMF_BENCH_DTYPE", "bf16x3"), choices=list(PEAK_TFLOPS))
    ap.add_argument("--arch", default="arch_mdm_l", choices=list(ARCHS))
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=196)
    ap.add_argument("--ddpm-steps", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--profile-out", default=None, help="write the per-kernel HIP-event profile of one step here (json)")
    ap.add_argument("--also", default="bf16", help="comma list of extra dtypes measured with 1 loop each (reported under other_dtypes); '' = none")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)

    from oakink2_tamf_amd import shard
    from oakink2_tamf_amd.hip_backend import TamfContext
    from oakink2_tamf_amd.model.diffusion_util import create_gaussian_diffusion
    from oakink2_tamf_amd.model.interaction_segment_mdm import InterationSegmentMDM

    arch = ARCHS[args.arch]
    B, T, N = args.batch, args.frames, args.ddpm_steps
    # random-init weights of the named architecture (PyTorch default initialisers, fixed seed; identical on all ranks)
    torch.manual_seed(0)
    sd = InterationSegmentMDM(**arch).state_dict()
    ctx = TamfContext(arch, B, T, precision=args.dtype, device=dev)
    ctx.load_state_dict(sd, max_timesteps=max(N, 1000))
    tab = create_gaussian_diffusion(diffusion_steps=N, noise_schedule="cosine")
    ctx.set_schedule(tab.posterior_mean_coef1, tab.posterior_mean_coef2, tab.posterior_log_variance_clipped)
    clip0 = shard.clip_id_base(rank, B)
    cond = synthetic_cond(B, T, seed=1000 + rank)
    cond_dev = {k: (v.to(dev) if hasattr(v, "to") else v) for k, v in cond.items()}
    out = torch.empty(B, 99, 1, T, device=dev)
    gathered = torch.empty(world * B, 99, 1, T, device=dev) if world > 1 else None

    def one_loop(seed):
        # the complete path: step-invariant conditioning precompute + n_ddpm-step reverse loop + result gather
        ctx.set_cond(cond_dev["text_embedding"], cond_dev["hand_side"], cond_dev["shape"], cond_dev["obj_embedding"],
                     cond_dev["obj_traj"])
        ctx.sample_loop(noise=None, seed=seed, clip_id_base=clip0, use_graph=not args.no_graph, out=out)
        if world > 1:
            shard.gather_clips(out, gathered)
        return gathered if world > 1 else out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for w in range(args.warmup):
        one_loop(1000 + w)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        res = one_loop(k)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    finite = bool(torch.isfinite(res).all().item())

    # dominant kernel, measured live with HIP events on the launch stream (rank 0)
    roofline = None
    prof_rows = None
    if rank == 0:
        ctx.set_cond(cond_dev["text_embedding"], cond_dev["hand_side"], cond_dev["shape"], cond_dev["obj_embedding"],
                     cond_dev["obj_traj"])
        agg = {}
        reps = 5
        for r in range(reps + 1):
            rows = ctx.step_profile()
            if r == 0:
                continue  # warm-up
            for name, ms, fl in rows:
                a = agg.setdefault(name, [0.0, 0.0, 0])
                a[0] += ms
                a[1] += fl
                a[2] += 1
        step_ms = sum(a[0] for a in agg.values()) / reps
        prof_rows = [
            {"kernel": n, "launches_per_step": a[2] // reps, "avg_ms": a[0] / a[2], "share": a[0] / reps / step_ms,
             "algorithmic_gflop_per_launch": a[1] / a[2] / 1e9, "tflops": (a[1] / a[2]) / (a[0] / a[2] * 1e-3) / 1e12 if a[0] > 0 else 0.0}
            for n, a in sorted(agg.items(), key=lambda kv: -kv[1][0])
        ]
        dom = prof_rows[0]
        peak = PEAK_TFLOPS[args.dtype]
        roofline = {
            "bound": "mfma",
            "kernel": dom["kernel"],
            "achieved": dom["tflops"],
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": dom["tflops"] / peak,
            "traffic": hbm_traffic(args.dtype, dom["kernel"], B, T),
            "avg_launch_ms": dom["avg_ms"],
            "share_of_step": dom["share"],
        }
        if args.profile_out:
            with open(args.profile_out, "w") as f:
                json.dump({"dtype": args.dtype, "B": B, "T": T, "step_ms_eventsum": step_ms, "kernels": prof_rows}, f, indent=1)

    # secondary dtypes: one warm-up + one timed loop each on rank 0's shard only (context, not the headline)
    other = {}
    if rank == 0 and world == 1:
        for dt in [d for d in args.also.split(",") if d and d != args.dtype]:
            c2 = TamfContext(arch, B, T, precision=dt, device=dev)
            c2.load_state_dict(sd, max_timesteps=max(N, 1000))
            c2.set_schedule(tab.posterior_mean_coef1, tab.posterior_mean_coef2, tab.posterior_log_variance_clipped)
            c2.set_cond(cond_dev["text_embedding"], cond_dev["hand_side"], cond_dev["shape"], cond_dev["obj_embedding"],
                        cond_dev["obj_traj"])
            c2.sample_loop(noise=None, seed=1, clip_id_base=clip0, out=out)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            c2.sample_loop(noise=None, seed=2, clip_id_base=clip0, out=out)
            torch.cuda.synchronize(dev)
            dt_s = time.perf_counter() - t1
            other[dt] = {"value": B * T / dt_s, "unit": "frames/s", "ms_per_ddpm_step": dt_s / N * 1e3,
                         "whole_path_tflops": flops_per_clip_step(arch, T) * B * N / dt_s / 1e12,
                         "note": "same workload, 1 timed loop; max |err| vs reference in DESIGN.md section 2"}
            c2.close()

    if rank == 0:
        frames = world * B * T * args.steps
        value = frames / elapsed
        fl_step = flops_per_clip_step(arch, T) * B * world
        whole_tflops = fl_step * N * args.steps / elapsed / 1e12
        line = {
            "metric": "sampled motion frames/s (1000-step DDPM, arch_mdm_l)" if (args.arch == "arch_mdm_l" and N == 1000)
            else f"sampled motion frames/s ({N}-step DDPM, {args.arch})",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (random-init weights of the named arch, N(0,1) CLIP/object conditioning, device Philox noise)",
            "config": {
                "workload": f"{args.arch} B={B}/GPU T={T} {N}-step DDPM (BASELINE.json configs[1]); step = one full reverse loop",
                "clips_per_gpu": B,
                "frames": T,
                "ddpm_steps": N,
                "global_clips": world * B,
                "parallelism": f"clip-sharded x{world}, RCCL all_gather of results" if world > 1 else "single GPU",
                "hipgraph": not args.no_graph,
                "kernels_per_ddpm_step": ctx.step_kernel_count,
            },
            "ms_per_ddpm_step": elapsed / args.steps / N * 1e3,
            "whole_path_tflops": whole_tflops,
            "whole_path_frac_of_peak": whole_tflops / (PEAK_TFLOPS[args.dtype] * world),
            "finite": finite,
            "roofline": roofline,
        }
        if other:
            line["other_dtypes"] = other
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only (one host measurement, not one per scaling point)
            line["cpu_baseline"] = cpu_baseline(args.arch, sd, T, N)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    ctx.close()
    if world > 1:
        dist.barrier()  # rank 0 was still profiling / printing
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
