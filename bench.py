#!/usr/bin/env python3
"""bench.py - sampled motion frames/s of the MF-MDM 1000-step DDPM sampler on MI355X.

Contract (one JSON line on rank 0's stdout):
  python bench.py --gpus N --steps K --warmup W
  N > 1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
  --master-port P bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or plainly
  `python bench.py --gpus N ...`: with WORLD_SIZE unset the parent spawns its N ranks itself as fresh child processes
  BEFORE anything has touched the GPU (a process that has initialised HIP is never re-executed).

A "step" is ONE complete reverse loop (n_ddpm = 1000 DDPM steps, hipGraph-replayed) over one batch of synthetic
clips.  Workload presets (--config): 2 = BASELINE.json configs[1] (arch_mdm_l, 64 clips per GPU, T = 196; the default),
3 = configs[2] (32 clips per GPU: B = 256 over 8 GPUs), 5 = configs[4] (bf16, 64 clips per GPU: B = 512 over 8 GPUs).
4 = configs[3] (arch_refine trunk, B = 64, T = 196: one forward per step, 1 GPU only).
Clips are independent, so ranks shard them with no data-path collective (weak scaling); the only exchange is the RCCL
all_gather of the sampled poses at the end of every loop, which is inside the timed region.

value = (clips of all ranks) * T * K / (max-over-ranks wall time of the K timed loops), inputs resident in HBM.
The line also carries, measured in the same run: `check` (max abs error of one denoiser evaluation against the oracle
for every dtype reported), `roofline` (dominant kernel of the headline dtype, HIP events on the launch stream; `peak` is the
nominal figure, `mfma_sustained` what register-only MFMA loops reach on this very board under its power management),
`other_dtypes` (the same workload in the other arithmetic modes, among them the reference's own fp32) and
`cpu_baseline` (the oracle on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oakink2-tamf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# MI355X_MICROARCH.md: dense MFMA peaks.  The split modes issue three 16-bit MFMAs per algorithmic product; their
# fraction is still quoted against the plain 16-bit peak (achieved = ALGORITHMIC flops / time).
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0, "bf16x3": 2500.0, "bf16": 2500.0}
MFMA_PER_PRODUCT = {"f32": 1, "f16x3": 3, "bf16x3": 3, "bf16": 1}
ARCHS = {
    "arch_mdm": dict(latent_dim=256, ff_size=1024, num_layers=8, num_heads=4),
    "arch_mdm_l": dict(latent_dim=512, ff_size=2048, num_layers=8, num_heads=4),
    "arch_refine": dict(latent_dim=256, ff_size=1024, num_layers=8, num_heads=4),  # MF-MDM R trunk (--config 4 only)
}
# --config presets: BASELINE.json configs[] (1-based numbering as in the task text: config 2 = configs[1])
CONFIGS = {
    2: dict(batch=64, dtype=None, label="BASELINE.json configs[1]: arch_mdm_l B=64 T=196 1000-step DDPM on 1 GPU"),
    3: dict(batch=32, dtype=None, label="BASELINE.json configs[2]: arch_mdm_l B=256 T=196 1000-step DDPM sharded over 8 GPUs (32 clips per GPU)"),
    4: dict(batch=64, dtype=None, label="BASELINE.json configs[3]: arch_refine (MF-MDM R) trunk B=64 T=196, one forward per step on cached G samples + h2o distances resident in HBM"),
    5: dict(batch=64, dtype="bf16", label="BASELINE.json configs[4]: arch_mdm_l bf16 B=512 T=196 hipGraph 1000-step loop over 8 GPUs (64 clips per GPU)"),
}
DEFAULT_DTYPE = "f16x3"  # = oakink2_tamf_amd.hip_backend.DEFAULT_PRECISION (asserted in main): one default everywhere
# in-run parity gates (max abs error of one denoiser evaluation vs the oracle, outputs O(1)): the gates of tests/test_hip_forward.py
CHECK_TOL = {"f32": 1e-5, "f16x3": 1e-5, "bf16x3": 6e-5, "bf16": 3e-2}


def flops_per_clip_step(arch, T):
    """SURVEY.md section 8(a)/BASELINE.md section 3: algorithmic FLOPs of one denoiser evaluation of one clip."""
    d, ff, L = arch["latent_dim"], arch["ff_size"], arch["num_layers"]
    S = T + 5
    return L * S * (8 * d * d + 4 * d * ff + 4 * S * d) + T * (4 * 99 * d + 6 * d * d) + 4 * d * d


def csrc_digest():
    """sha256 (first 16 hex digits) over the kernel sources (csrc/*.h, *.hip, sorted by name): what a committed counter file is
    stamped with (tools/collect_round_profiles.sh) and compared against - the GPU box has no .git, so a commit id cannot be
    checked there, the sources can."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "oakink2-tamf_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def hbm_traffic(dtype, kernel, B, T):
    """(bytes per launch, source, stale) of `kernel` from the committed PMC measurement of this exact workload
    (profiles/rNN/hbm_traffic_<dtype>.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH doubled
    per the gfx950 correction of MI355X_MICROARCH.md).  The counters cannot be collected inside a timed run (they need
    the profiler's own passes), so the value is NOT measured by this process: `source` names the file and the commit the
    counters were taken at; `stale` is True when the file's `csrc_sha16` stamp is missing or differs from the kernel sources of this
    tree (the counters then describe OTHER kernels and the line says so).  Newest round first; (None, None, None) when no
    measurement matches."""
    pdir = os.path.join(ROOT, "profiles")
    for rnd in sorted((d for d in os.listdir(pdir) if d.startswith("r")), reverse=True) if os.path.isdir(pdir) else []:
        try:
            rel = os.path.join("profiles", rnd, f"hbm_traffic_{dtype}.json")
            with open(os.path.join(ROOT, rel)) as f:
                m = json.load(f)
            if m.get("B") == B and m.get("T") == T and kernel in m["kernels"]:
                return (m["kernels"][kernel]["traffic_bytes_per_launch"], f"{rel}@{m.get('commit', 'unstamped')}",
                        m.get("csrc_sha16") != csrc_digest())
        except (OSError, ValueError, KeyError):
            continue
    return None, None, None


class PowerTrace:
    """Package power / shader clock of the GPU during the timed loops, sampled by a CHILD process (tools/power_sampler.py: sysfs hwmon
    of every amdgpu card, ~4 Hz; it touches no GPU API).  The child is started before this process initialises the GPU (a process that
    has must not fork + exec on the pool).  The card of this job = the one whose power rises most from the seconds before the loops to
    the timed window."""

    def __init__(self):
        import tempfile

        self.dir = tempfile.mkdtemp(prefix="tamf_power_")
        self.out, self.stop = os.path.join(self.dir, "samples.txt"), os.path.join(self.dir, "stop")
        self.t_start = time.time()
        try:
            self.proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "power_sampler.py"), self.out, self.stop],
                                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except OSError:
            self.proc = None
        import atexit

        atexit.register(self.cleanup)  # (a bench that dies early must not leave its sampler behind; the sampler also ends by itself after 30 min)

    def cleanup(self):
        import shutil

        try:
            open(self.stop, "w").close()
        except OSError:
            pass
        if self.proc is not None and self.proc.poll() is None:
            try:
                self.proc.wait(timeout=3)
            except subprocess.TimeoutExpired:
                self.proc.kill()
        shutil.rmtree(self.dir, ignore_errors=True)

    def finish(self, t0, t1, pci=None):
        """-> dict for the bench line (or None): mean watts / sclk of this job's card over [t0, t1].  pci = the PCI address of this
        process's device ("0000:bb:dd.f"): matched against the sampler's header; without a match the card is guessed from the power rise."""
        if self.proc is None:
            return None
        open(self.stop, "w").close()
        try:
            self.proc.wait(timeout=5)
        except subprocess.TimeoutExpired:
            self.proc.kill()
        column = None
        try:
            lines = open(self.out).read().split("\n")
            rows = [list(map(float, l.split())) for l in lines if l.strip() and l[0] != "#"]
            for l in lines:
                if l.startswith("# pci:") and pci:
                    addrs = [a.lower() for a in l.split()[2:]]
                    if pci.lower() in addrs:
                        column = addrs.index(pci.lower())
        except (OSError, ValueError):
            return None
        return self.summarise(rows, self.t_start, t0, t1, column)

    @staticmethod
    def summarise(rows, t_start, t0, t1, column=None):
        """rows: [t, watts card 0, sclk card 0, watts card 1, ...] (tools/power_sampler.py).  The card of this job = `column` when the
        caller could match its device's PCI address; else the one whose mean power rises most from the first 3 s after t_start (this
        process was still importing) to the timed window [t0, t1] - a guess that another job starting on a neighbouring card can fool."""
        rows = [r for r in rows if len(r) >= 3 and len(r) % 2 == 1]
        if not rows:
            return None
        ncard = (len(rows[0]) - 1) // 2

        def mean(k, a, b, col):
            xs = [r[1 + 2 * k + col] for r in rows if a <= r[0] <= b and len(r) == 1 + 2 * ncard and r[1 + 2 * k + col] == r[1 + 2 * k + col]]
            return (sum(xs) / len(xs) if xs else None), len(xs)

        best = None
        for k in (range(ncard) if column is None or column >= ncard else [column]):
            w, n = mean(k, t0 + 0.5, t1 - 0.2, 0)
            pre, _ = mean(k, t_start, t_start + 3.0, 0)
            if w is None:
                continue
            rise = w - (pre if pre is not None else 0.0)
            if best is None or rise > best[0]:
                best = (rise, k, w, n, pre)
        if best is None:
            return None
        _, k, w, n, pre = best
        mhz, _ = mean(k, t0 + 0.5, t1 - 0.2, 1)
        return {"watts": w, "sclk_mhz": mhz, "samples": n, "card_column": k, "card_matched_by": "pci address" if column is not None and column < ncard else "power rise (guess)",
                "cards_sampled": ncard, "watts_before_the_run": pre,
                "what": "mean package power / shader clock of this job's GPU over the timed loops (sysfs hwmon, sampled by tools/power_sampler.py)"}


def power_model(dtype, power, ms_per_ddpm_step):
    """roofline.power_model: the committed energy model of one DDPM step (profiles/rNN/energy_model.json: joules per MFMA / byte /
    VALU instruction from single-resource microbenchmarks x the step's resource counts, tools/energy_model.py) beside the power this
    run drew.  predicted_ms = modelled dynamic joules / (measured watts - idle watts).  None when there is no model for the dtype."""
    pdir = os.path.join(ROOT, "profiles")
    for rnd in sorted((d for d in os.listdir(pdir) if d.startswith("r")), reverse=True) if os.path.isdir(pdir) else []:
        rel = os.path.join("profiles", rnd, "energy_model.json")
        try:
            with open(os.path.join(ROOT, rel)) as f:
                m = json.load(f)
            mm = m["modes"][dtype]
        except (OSError, ValueError, KeyError):
            continue
        e = mm["joules_model"]
        jb = e.get("fabric_read", 0.0) + e.get("fabric_write", 0.0) + e.get("l2_to_lds", 0.0) + e.get("lds_fragment_reads", 0.0)
        out = {"source": rel, "workload": "B=64 T=196 arch_mdm_l", "joules_mfma": e.get("mfma", 0.0) + e.get("mfma_operand_refresh", 0.0), "joules_bytes": jb,
               "joules_valu": e.get("valu", 0.0) + e.get("salu", 0.0),
               "joules_dynamic_modelled": mm["joules_dynamic_model"], "idle_watts": m["idle_watts"], "cap_watts": m["cap_watts"],
               "model_run": {"measured_ms": mm["measured_ms"], "measured_watts": mm["measured_watts"], "predicted_ms": mm["predicted_ms_at_measured_power"]}}
        if power and power.get("watts"):
            w = power["watts"]
            out.update({"measured_ms": ms_per_ddpm_step, "measured_watts": w, "measured_joules": w * ms_per_ddpm_step * 1e-3,
                        "joules_idle": m["idle_watts"] * ms_per_ddpm_step * 1e-3,
                        "predicted_ms": mm["joules_dynamic_model"] / max(1.0, w - m["idle_watts"]) * 1e3})
        return out
    return None


def checks_ok(finite_by, range_flags, check):
    """The line's `check_ok`: every reported dtype sampled finite values, no fp16 range flag was raised during the timed loops,
    and every in-run oracle comparison is inside its per-dtype tolerance (NaN fails).  bench.py exits 3 when this is False."""
    ok = all(finite_by.values()) and not any(range_flags.values())
    return ok and all(e == e and e < CHECK_TOL[d] for d, e in check.items())


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


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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


def cpu_baseline(arch_name, sd, B, T, n_ddpm, timed=5):
    """The oracle (torch-CPU restatement of the reference, proven equal to it on tests/golden) timed on the host
    cores with the SAME weights as the GPU run, at the workload's own (B, T): 1 warm-up + `timed` denoiser+DDPM steps
    (BASELINE.md section 4), extrapolated to the n_ddpm-step loop (every step does identical work).
    This and `oracle_check` are the only places bench.py touches oracle/."""
    import torch

    from oracle import mdm_oracle as O

    arch = {"arch_mdm": O.ARCH_MDM, "arch_mdm_l": O.ARCH_MDM_L}[arch_name]
    cores = usable_cores()
    torch.set_num_threads(cores)
    cond = synthetic_cond(B, T, seed=12345)
    tab = O.make_tables(n_ddpm, "cosine")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 99, 1, T, generator=g)
    times = []
    with torch.no_grad():
        for it in range(1 + timed):
            i = n_ddpm - 1 - it
            t0 = time.perf_counter()
            x0 = O.denoiser_forward(sd, arch, x, torch.full((B,), i, dtype=torch.long), cond)
            x = O.ddpm_step(tab, x, x0, i, torch.randn(x.shape, generator=g))
            times.append(time.perf_counter() - t0)
    step_s = sum(times[1:]) / timed
    return {
        "value": B * T / (step_s * n_ddpm),
        "unit": "frames/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "cpu_model": cpu_model(),
        "sample": f"{B} clips x T={T}, {timed} timed denoiser+DDPM steps after 1 warm-up ({step_s * 1e3:.0f} ms/step), "
        f"extrapolated to {n_ddpm} steps; conditioning recomputed every step as the reference does",
    }


def torch_rocm_baseline(arch_name, sd, cond, B, T, n_ddpm, dev, check_in=None, timed=10):
    """Same-box LIBRARY yardstick (outside every timed region of the product path, never in it): what the reference's own code path
    - launch/sample.py:213-229 on PyTorch-ROCm, i.e. hipBLASLt / rocBLAS GEMMs + torch's attention - delivers on THIS board under the
    same power cap, on the same weights, batch and clip length as the headline.  Two forms of the denoiser x two arithmetic types:
      "oracle_algebra"       oracle.denoiser_forward (plain tensor algebra: matmul / softmax / matmul) with its tensors on the device;
      "nn_transformer_encoder"  the same forward with torch's own nn.TransformerEncoder - the module the reference instantiates
                             (model/interaction_segment_mdm.py:63-70: seq-first, gelu, eval) - as the encoder: F.multi_head_attention_forward
                             -> scaled_dot_product_attention;
      fp32, and bf16 under torch.autocast.
    2 warm-up + `timed` denoiser + DDPM steps (conditioning recomputed every step, as the reference does, CLIP excluded), HIP-synchronised,
    extrapolated to the n_ddpm-step loop exactly like cpu_baseline.  A reported baseline, not the product: nothing here is shipped."""
    import torch

    from oracle import mdm_oracle as O

    arch = {"arch_mdm": O.ARCH_MDM, "arch_mdm_l": O.ARCH_MDM_L}[arch_name]
    sd_dev = {k: v.to(dev) for k, v in sd.items()}
    cond_dev = {k: (v.to(dev) if hasattr(v, "to") else v) for k, v in cond.items()}
    tab = O.make_tables(n_ddpm, "cosine")
    d = arch.latent_dim
    layer = torch.nn.TransformerEncoderLayer(d_model=d, nhead=arch.num_heads, dim_feedforward=arch.ff_size, dropout=0.1, activation="gelu")
    enc = torch.nn.TransformerEncoder(layer, num_layers=arch.num_layers)
    enc.load_state_dict({k[len("seqTransEncoder."):]: v for k, v in sd.items() if k.startswith("seqTransEncoder.")})
    enc = enc.to(dev).eval()

    def nn_encoder(seq):  # batch-first in / out; the module runs seq-first as in the reference
        return enc(seq.transpose(0, 1)).transpose(0, 1)

    out = {"what": "the oracle / torch's nn.TransformerEncoder on the device through PyTorch-ROCm's libraries: same box, same weights, same (B, T); "
                   "a reported yardstick outside the product path", "torch": torch.__version__, "hip": torch.version.hip, "variants": {}}
    g = torch.Generator().manual_seed(0)
    x_init = torch.randn(B, 99, 1, T, generator=g).to(dev)
    noise = torch.randn(B, 99, 1, T, generator=g).to(dev)
    for form, encoder in (("oracle_algebra", None), ("nn_transformer_encoder", nn_encoder)):
        for dt_name in ("f32", "bf16_autocast"):
            try:
                x = x_init.clone()
                times = []
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=(dt_name != "f32")):
                    for it in range(2 + timed):
                        i = n_ddpm - 1 - it
                        torch.cuda.synchronize(dev)
                        t0 = time.perf_counter()
                        x0 = O.denoiser_forward(sd_dev, arch, x, torch.full((B,), i, dtype=torch.long, device=dev), cond_dev, encoder=encoder)
                        x = O.ddpm_step(tab, x, x0.float(), i, noise)
                        torch.cuda.synchronize(dev)
                        times.append(time.perf_counter() - t0)
                    err = None
                    if check_in is not None:
                        xc, tc, ref, nc = check_in
                        sub = {k: (v[:nc] if hasattr(v, "to") else list(v[:nc])) for k, v in cond_dev.items()}
                        got = O.denoiser_forward(sd_dev, arch, xc[:nc].to(dev), tc[:nc].to(dev), sub, encoder=encoder)
                        err = float((got.float().cpu() - ref).abs().max())
                step_s = sum(times[2:]) / timed
                tf = flops_per_clip_step(ARCHS[arch_name], T) * B / step_s / 1e12
                out["variants"][f"{form}/{dt_name}"] = {
                    "value": B * T / (step_s * n_ddpm), "unit": "frames/s", "ms_per_ddpm_step": step_s * 1e3, "whole_path_tflops": tf,
                    "max_abs_err_vs_cpu_oracle": err, "sample": f"{timed} timed denoiser+DDPM steps after 2 warm-up, extrapolated to {n_ddpm}"}
            except Exception as e:  # noqa: BLE001  (a yardstick that cannot run must not take the bench line with it)
                out["variants"][f"{form}/{dt_name}"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    ok = {k: v for k, v in out["variants"].items() if "value" in v}
    if ok:
        for cls, pick in (("f32", [k for k in ok if k.endswith("/f32")]), ("bf16", [k for k in ok if k.endswith("/bf16_autocast")])):
            if pick:
                best = max(pick, key=lambda k: ok[k]["value"])
                out[f"best_{cls}"] = {"variant": best, "value": ok[best]["value"], "ms_per_ddpm_step": ok[best]["ms_per_ddpm_step"]}
    return out


def oracle_reference(arch_name, sd, cond, x, t, n_check):
    """oracle.denoiser_forward on the first n_check clips of the bench batch (outside every timed region)."""
    import torch

    from oracle import mdm_oracle as O

    arch = {"arch_mdm": O.ARCH_MDM, "arch_mdm_l": O.ARCH_MDM_L}[arch_name]
    sub = {k: (v[:n_check].cpu() if hasattr(v, "cpu") else list(v[:n_check])) for k, v in cond.items()}
    with torch.no_grad():
        return O.denoiser_forward(sd, arch, x[:n_check].cpu(), t[:n_check].cpu(), sub)


class HipSampler:
    """The product path: one library context of one arithmetic mode, conditioning resident in HBM."""

    def __init__(self, arch, sd, B, T, N, dtype, dev, tab, use_graph=True):
        from oakink2_tamf_amd.hip_backend import TamfContext

        self.ctx = TamfContext(arch, B, T, precision=dtype, device=dev)
        self.ctx.load_state_dict(sd, max_timesteps=max(N, 1000))
        self.ctx.set_schedule(tab.posterior_mean_coef1, tab.posterior_mean_coef2, tab.posterior_log_variance_clipped)
        self.use_graph = use_graph

    def set_cond(self, cond_dev):
        self.ctx.set_cond(cond_dev["text_embedding"], cond_dev["hand_side"], cond_dev["shape"], cond_dev["obj_embedding"],
                          cond_dev["obj_traj"])

    def sample(self, seed, clip0, out):
        self.ctx.sample_loop(noise=None, seed=seed, clip_id_base=clip0, use_graph=self.use_graph, out=out)

    def denoise(self, x, t):
        return self.ctx.denoise(x, t)

    def step_profile(self):
        return self.ctx.step_profile()

    def status_flags(self):
        return self.ctx.status_flags(clear=True)

    @property
    def kernels_per_step(self):
        return self.ctx.step_kernel_count

    def close(self):
        self.ctx.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv, poll_s=0.2, grace_s=10.0):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent has not initialised the
    GPU and never will) and watch them.  Rank 0's stdout is the JSON line.  When a rank exits non-zero its siblings - which
    would otherwise sit in the process-group set-up or a collective until the RCCL timeout - are terminated (SIGTERM, then
    SIGKILL after `grace_s`) and that rank's code is returned; otherwise 0 once all have finished."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            return 0
        time.sleep(poll_s)
    failed = [r for r, c in enumerate(codes) if c not in (None, 0)]
    print(f"bench.py: rank(s) {failed} exited with {rc}; terminating the other ranks", file=sys.stderr, flush=True)
    for p in procs:
        if p.poll() is None:
            p.terminate()
    t_end = time.monotonic() + grace_s
    for p in procs:
        try:
            p.wait(timeout=max(0.0, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return rc


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 3 reverse loops; --config 4: 200 forwards)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 1; --config 4: 20)")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json workload preset")
    ap.add_argument("--dtype", default=os.environ.get("TAMF_BENCH_DTYPE"), choices=list(PEAK_TFLOPS))
    ap.add_argument("--arch", default="arch_mdm_l", choices=list(ARCHS))
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: the preset's)")
    ap.add_argument("--frames", type=int, default=196)
    ap.add_argument("--ddpm-steps", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-torch-baseline", action="store_true", help="skip the same-box PyTorch-ROCm library yardstick (torch_rocm_baseline)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="do not start the power / clock sampler child")
    ap.add_argument("--check-clips", type=int, default=8, help="clips of the in-run oracle check (0 = off)")
    ap.add_argument("--profile-out", default=None, help="write the per-kernel HIP-event profile of one step here (json)")
    ap.add_argument("--also", default=None,
                    help="comma list of extra dtypes measured with 1 loop each on 1 GPU (other_dtypes); default: all other modes; '' = none")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: CPU test of the launch/gather path (needs --sampler)")
    ap.add_argument("--sampler", default=None, help="test hook (only with --backend gloo): module:callable replacing the HIP sampler (tests/bench_stub.py)")
    ap.add_argument("--fp32-loops", type=int, default=3, help="timed loops of the strict-fp32 entry of the line (N = 1 only; 0 = off)")
    args = ap.parse_args(argv)
    if args.sampler and args.backend != "gloo":
        ap.error("--sampler is the CPU test hook of the launch / gather path: it needs --backend gloo")
    preset = CONFIGS[args.config]
    if args.steps is None:
        args.steps = 200 if args.config == 4 else 3
    if args.warmup is None:
        args.warmup = 20 if args.config == 4 else 1
    if args.config == 4:
        args.arch = "arch_refine"
    elif args.arch == "arch_refine":
        ap.error("arch_refine is the R trunk: use --config 4")
    if args.batch is None:
        args.batch = preset["batch"]
    if args.dtype is None:
        args.dtype = preset["dtype"] or DEFAULT_DTYPE
    if args.also is None:
        args.also = ",".join(d for d in PEAK_TFLOPS if d != args.dtype)
    return args


def main_refine(args, dev, ptrace):
    """--config 4 = BASELINE.json configs[3]: the arch_refine (MF-MDM R) trunk, B = 64, T = 196, ONE forward per step
    (reference model/segment_refine_model.py:175-217 as launch/sample_refine.py:236-238 calls it), inputs - the cached G samples, the
    hand->object distances, the conditioning - resident in HBM.  The conditioning precompute (tamf_set_cond) is inside the step, as in
    the G bench.  value = refined frames/s; ms_per_step = ms per batch (tiny and latency-dominated, reported as such)."""
    import torch

    from oakink2_tamf_amd.hip_backend import TamfContext
    from oakink2_tamf_amd.model.segment_refine_model import SegmentRefineModel

    arch = ARCHS["arch_refine"]
    B, T = args.batch, args.frames
    # random-init weights of the named architecture (PyTorch default initialisers, fixed seed), synthetic inputs of the config's shape
    torch.manual_seed(0)
    sd = {k: v for k, v in SegmentRefineModel(None, **arch).state_dict().items()}
    cond = {k: v for k, v in synthetic_cond(B, T, seed=1000, nobj=2).items() if k != "text_embedding"}
    g = torch.Generator().manual_seed(77)
    x_in = torch.randn(B, T, 99, generator=g).to(dev)
    h2o = (torch.randn(B, T, 778, generator=g).abs() * 0.05).to(dev)
    cond_dev = {k: (v.to(dev) if hasattr(v, "to") else v) for k, v in cond.items()}
    n_check = max(0, min(args.check_clips, B))
    ref = None
    if n_check or not args.no_cpu_baseline:  # the checker / CPU baseline: the only use of oracle/ here
        from oracle import mdm_oracle as O

        oarch = O.ARCH_REFINE
    if n_check:
        sub = {k: (v[:n_check] if not isinstance(v, list) else v[:n_check]) for k, v in cond.items()}
        with torch.no_grad():
            ref = O.refine_forward(sd, oarch, x_in[:n_check].cpu(), h2o[:n_check].cpu(), sub)

    def run_mode(dt, steps, warmup):
        ctx = TamfContext(arch, B, T, precision=dt, device=dev, kind="R")
        ctx.load_state_dict(sd)

        def fwd():
            ctx.set_cond(None, cond_dev["hand_side"], cond_dev["shape"], cond_dev["obj_embedding"], cond_dev["obj_traj"])
            return ctx.refine(x_in, h2o)

        for _ in range(warmup):
            out = fwd()
        torch.cuda.synchronize(dev)
        t0, w0 = time.perf_counter(), time.time()
        for _ in range(steps):
            out = fwd()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        w1 = time.time()
        flag = bool(ctx.status_flags(clear=True) & 1) if dt == "f16x3" else False
        err = float((out[:n_check].cpu() - ref).abs().max()) if ref is not None else None
        agg, reps = {}, 5
        for r in range(reps + 1):
            rows = ctx.refine_profile(x_in, h2o)
            if r:
                for name, ms, fl in rows:
                    a = agg.setdefault(name, [0.0, 0.0, 0])
                    a[0], a[1], a[2] = a[0] + ms, a[1] + fl, a[2] + 1
        fwd_flops = sum(a[1] for a in agg.values()) / reps
        prof = [{"kernel": n, "launches_per_forward": a[2] // reps, "avg_ms": a[0] / a[2], "algorithmic_gflop_per_launch": a[1] / a[2] / 1e9,
                 "tflops": (a[1] / a[2]) / (a[0] / a[2] * 1e-3) / 1e12 if a[0] > 0 else 0.0, "share": a[0] / sum(x[0] for x in agg.values())}
                for n, a in sorted(agg.items(), key=lambda kv: -kv[1][0])]
        finite = bool(torch.isfinite(out).all().item())
        ctx.close()
        return {"elapsed": el, "wall": (w0, w1), "err": err, "flag": flag, "finite": finite, "profile": prof, "forward_gflop": fwd_flops / 1e9}

    head = run_mode(args.dtype, args.steps, args.warmup)
    power = None
    if ptrace is not None:
        pr = torch.cuda.get_device_properties(dev)
        pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id) if hasattr(pr, "pci_bus_id") else None
        power = ptrace.finish(head["wall"][0], head["wall"][1], pci)
    peak = PEAK_TFLOPS[args.dtype]
    dom = head["profile"][0]
    ms_batch = head["elapsed"] / args.steps * 1e3
    whole_tf = head["forward_gflop"] * 1e9 / (ms_batch * 1e-3) / 1e12
    roofline = {"bound": "mfma", "kernel": dom["kernel"], "dtype": args.dtype, "mfma_per_product": MFMA_PER_PRODUCT[args.dtype],
                "achieved": dom["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": dom["tflops"] / peak, "traffic": None,
                "avg_launch_ms": dom["avg_ms"], "share_of_step": dom["share"], "forward_algorithmic_gflop": head["forward_gflop"],
                "launches_per_forward": sum(r["launches_per_forward"] for r in head["profile"]),
                "kernels": [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()} for r in head["profile"]],
                "note": "a 1.2-ms forward of 50 launches over 12 736 token rows of width 256: launch ramps and epilogues, not a roofline, bound it (SURVEY.md 8d)"}
    check = {args.dtype: head["err"]} if head["err"] is not None else {}
    finite_by, range_flags = {args.dtype: head["finite"]}, ({args.dtype: head["flag"]} if args.dtype == "f16x3" else {})
    other = {}
    for dt in [d for d in args.also.split(",") if d and d != args.dtype]:
        r = run_mode(dt, max(20, args.steps // 4), 5)
        ms = r["elapsed"] / max(20, args.steps // 4) * 1e3
        other[dt] = {"value": B * T / ms * 1e3, "unit": "frames/s", "ms_per_batch": ms, "finite": r["finite"],
                     "whole_path_tflops": r["forward_gflop"] / ms, "whole_path_frac_of_peak": r["forward_gflop"] / ms / PEAK_TFLOPS[dt],  # (GFLOP per ms = TFLOP/s)
                     "dominant_kernel": r["profile"][0]["kernel"], "dominant_kernel_frac": r["profile"][0]["tflops"] / PEAK_TFLOPS[dt]}
        finite_by[dt] = r["finite"]
        if r["err"] is not None:
            check[dt] = r["err"]
        if dt == "f16x3":
            range_flags[dt] = r["flag"]
    if "f32" in other:
        roofline["f32_value"], roofline["f32_ms_per_batch"] = other["f32"]["value"], other["f32"]["ms_per_batch"]
        roofline["f32_whole_path_frac"] = other["f32"]["whole_path_frac_of_peak"]
    line = {
        "metric": "refined motion frames/s (arch_refine trunk, one forward per batch)", "value": B * T * args.steps / head["elapsed"],
        "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_batch, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic (random-init weights of arch_refine, N(0,1) G samples, |N(0, 5 cm)| hand->object distances, N(0,1) conditioning)",
        "config": {"workload": f"arch_refine trunk B={B} T={T}, one forward per step ({CONFIGS[4]['label']})", "preset": 4, "clips_per_gpu": B,
                   "frames": T, "parallelism": "single GPU"},
        "ms_per_batch": ms_batch, "whole_path_tflops": whole_tf, "whole_path_frac_of_peak": whole_tf / peak, "finite": head["finite"],
        "roofline": roofline, "kernels": head["profile"][:8],
    }
    if power is not None:
        line["power"] = power
    if check:
        line["check"] = {"max_abs_err_vs_oracle": check, "tolerance": {d: CHECK_TOL[d] for d in check},
                         "what": f"refine_pose_repr of the first {n_check} clips of the bench batch vs oracle.refine_forward (fp32 torch-CPU restatement of the reference)"}
    if range_flags:
        line["f16_range_flag"] = range_flags
    line["finite_by_dtype"] = finite_by
    line["check_ok"] = checks_ok(finite_by, range_flags, check)
    if other:
        line["other_dtypes"] = other
    if not args.no_cpu_baseline:
        cores = usable_cores()
        torch.set_num_threads(cores)
        times = []
        with torch.no_grad():
            for it in range(4):
                t0 = time.perf_counter()
                O.refine_forward(sd, oarch, x_in.cpu(), h2o.cpu(), cond)
                times.append(time.perf_counter() - t0)
        fs = sum(times[1:]) / 3
        line["cpu_baseline"] = {"value": B * T / fs, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": cpu_model(),
                                "sample": f"{B} clips x T={T}: 3 timed oracle.refine_forward calls after 1 warm-up ({fs * 1e3:.0f} ms per batch)"}
        line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
    print(json.dumps(line), flush=True)
    if not line["check_ok"]:
        print("bench.py: in-run check FAILED (see check / finite_by_dtype / f16_range_flag in the line)", file=sys.stderr, flush=True)
        return 3
    return 0


def main(argv=None, sampler_factory=None):
    """sampler_factory(arch, sd, B, T, N, dtype, dev, tab) -> object with the HipSampler interface; the default is the
    HIP path.  The CPU test of the multi-rank plumbing passes a stub (tests/test_bench_main_gloo.py)."""
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become the launcher (before any GPU call in this process)
        raise SystemExit(spawn_ranks(args.gpus, argv))

    # power / clock sampler child: started BEFORE this process touches the GPU (rank 0 of a real run only)
    ptrace = None
    # (never under a profiler: rocprofv3's preloaded library has initialised the GPU before this line, and a process that has must
    #  not fork + exec on the pool - tools/*.sh pass --no-power as well)
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    import numpy as np  # noqa: F401
    import torch  # (importing torch does not initialise the GPU; is_available() / any HIP call does)
    import torch.distributed as dist

    # ... nor from a process that has ALREADY initialised the GPU (bench.main called from a test or a notebook that used the GPU before)
    if int(os.environ.get("RANK", "0")) == 0 and not args.sampler and not args.no_power and not under_profiler and not torch.cuda.is_initialized():
        ptrace = PowerTrace()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if sampler_factory is None and args.sampler:
        import importlib

        mod, attr = args.sampler.split(":")
        sampler_factory = getattr(importlib.import_module(mod), attr)
    stub = sampler_factory is not None
    if stub and args.backend == "gloo":
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dev.type == "cuda":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from oakink2_tamf_amd import shard
    from oakink2_tamf_amd.model.diffusion_util import create_gaussian_diffusion
    from oakink2_tamf_amd.model.interaction_segment_mdm import InterationSegmentMDM

    if args.config == 4:
        if world != 1 or stub:
            raise SystemExit("--config 4 (the R trunk, one forward per step) is a single-GPU measurement")
        return main_refine(args, dev, ptrace)
    factory = sampler_factory or HipSampler
    rc = 0
    if not stub:
        from oakink2_tamf_amd.hip_backend import DEFAULT_PRECISION

        assert DEFAULT_DTYPE == DEFAULT_PRECISION, "bench.py and the package must share one default precision"
    arch = ARCHS[args.arch]
    B, T, N = args.batch, args.frames, args.ddpm_steps
    # random-init weights of the named architecture (PyTorch default initialisers, fixed seed; identical on all ranks)
    torch.manual_seed(0)
    sd = InterationSegmentMDM(**arch).state_dict()
    tab = create_gaussian_diffusion(diffusion_steps=N, noise_schedule="cosine")
    sampler = factory(arch, sd, B, T, N, args.dtype, dev, tab, use_graph=not args.no_graph)
    clip0 = shard.clip_id_base(rank, B)
    cond = synthetic_cond(B, T, seed=1000 + rank)
    cond_dev = {k: (v.to(dev) if hasattr(v, "to") else v) for k, v in cond.items()}
    out = torch.empty(B, 99, 1, T, device=dev)
    gathered = torch.empty(world * B, 99, 1, T, device=dev) if world > 1 else None

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def one_loop(seed):
        # the complete path: step-invariant conditioning precompute + n_ddpm-step reverse loop + result gather
        sampler.set_cond(cond_dev)
        sampler.sample(seed, clip0, out)
        if world > 1:
            shard.gather_clips(out, gathered)
        return gathered if world > 1 else out

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    for w in range(args.warmup):
        one_loop(1000 + w)
    barrier()
    t0 = time.perf_counter()
    wall0 = time.time()
    for k in range(args.steps):
        res = one_loop(k)
    barrier()
    elapsed = time.perf_counter() - t0
    power = None
    if ptrace is not None:
        pci = None
        if dev.type == "cuda":
            try:
                pr = torch.cuda.get_device_properties(dev)
                pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            except (AttributeError, RuntimeError):
                pci = None
        power = ptrace.finish(wall0, time.time(), pci)
        if power is not None:
            power["pci"] = pci
    if world > 1:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    finite = bool(torch.isfinite(res).all().item())
    # range guard of the split-fp16 mode: sticky device flag raised by any operand store beyond +-65504 during the loops above
    range_flags = {}
    my_flag = False
    if not stub and args.dtype == "f16x3":
        my_flag = bool(sampler.status_flags() & 1)
        flag_any = my_flag
        if world > 1:  # every rank's flag counts: MAX over ranks, like `elapsed`
            tf_ = torch.tensor([1.0 if my_flag else 0.0], device=dev, dtype=torch.float64)
            dist.all_reduce(tf_, op=dist.ReduceOp.MAX)
            flag_any = bool(tf_.item() > 0)
        range_flags[args.dtype] = flag_any
    # what every rank sampled, as seen by the collective (printed by rank 0)
    rank_info = {"rank": rank, "clips": [clip0, clip0 + B], "device": str(dev), "f16_range_flag": my_flag,
                 "finite": bool(torch.isfinite(out).all().item()), "pid": os.getpid(), "local_rank": local_rank}
    if dev.type == "cuda":  # which physical GPU this rank really drove, and the collective library it spoke through
        try:
            pr = torch.cuda.get_device_properties(dev)
            rank_info["pci"] = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            rank_info["gpu"] = pr.name
        except (AttributeError, RuntimeError):
            rank_info["pci"] = None
        try:
            rank_info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001  (a torch build without the binding: the line simply lacks the field)
            rank_info["rccl_version"] = None
    if world > 1:
        infos = [None] * world
        dist.all_gather_object(infos, rank_info)
        world_seen = dist.get_world_size()
    else:
        infos, world_seen = [rank_info], 1

    # in-run parity: one denoiser evaluation of the first clips of this batch against the oracle, per dtype reported
    check = {}
    check_in = None
    if rank == 0 and args.check_clips > 0 and not stub:
        nc = min(args.check_clips, B)
        g = torch.Generator().manual_seed(4242)
        xc = torch.randn(B, 99, 1, T, generator=g)
        # the checked clips sit at different points of the schedule (t is per clip in the denoiser's contract, respace.py:114-119)
        tc = torch.tensor([(N // 2, 0, N - 1, N // 4, 3 * N // 4, 1, N - 2, N // 3)[b % 8] for b in range(B)], dtype=torch.long)
        ref = oracle_reference(args.arch, sd, cond, xc, tc, nc)
        check_in = (xc, tc, ref, nc)
        sampler.set_cond(cond_dev)
        got = sampler.denoise(xc.to(dev), tc.to(dev))[:nc].cpu()
        check[args.dtype] = float((got - ref).abs().max())

    # what the matrix pipe of THIS board sustains by itself (register-only MFMA loops with random operands, 2 s per instruction
    # kind, measured once): with real operand bits MI355X throttles well below the nominal peak - DESIGN.md section 6, "power"
    sustained_cache = {}

    def mfma_sustained(dtype):
        kind = {"f32": "f32", "bf16": "bf16", "bf16x3": "bf16", "f16x3": "f16x3"}[dtype]
        if kind not in sustained_cache:
            from oakink2_tamf_amd.hip_backend import mfma_sustained_rate

            tf, mhz = mfma_sustained_rate(kind, 2000, dev)
            sustained_cache[kind] = {"value": tf, "unit": "TFLOP/s", "implied_sclk_mhz": mhz,
                                     "what": "register-only " + {"f32": "v_mfma_f32_16x16x4_f32", "bf16": "v_mfma_f32_16x16x32_bf16", "f16x3": "v_mfma_f32_16x16x32_f16"}[kind]
                                             + " loops with random operands on every SIMD of this GPU, 2 s, last two thirds timed (tamf_bench_mfma_rate)"}
        return sustained_cache[kind]

    # dominant kernel, measured live with HIP events on the launch stream (rank 0)
    def roofline_of(smp, dtype, profile_out=None):
        agg = {}
        reps = 5
        for r in range(reps + 1):
            rows = smp.step_profile()
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
        peak = PEAK_TFLOPS[dtype]
        if profile_out:
            with open(profile_out, "w") as f:
                json.dump({"dtype": dtype, "B": B, "T": T, "step_ms_eventsum": step_ms, "kernels": prof_rows}, f, indent=1)
        sus = mfma_sustained(dtype)
        return {
            "bound": "mfma",
            "kernel": dom["kernel"],
            "dtype": dtype,
            "mfma_per_product": MFMA_PER_PRODUCT[dtype],
            "achieved": dom["tflops"],
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": dom["tflops"] / peak,
            # the mode's own ceiling: three MFMAs per product in the split modes, i.e. at most a third of the 16-bit peak (= frac elsewhere)
            "frac_of_split_ceiling": dom["tflops"] * MFMA_PER_PRODUCT[dtype] / peak,
            "mfma_sustained": sus,  # informational: `peak` above stays the nominal figure of MI355X_MICROARCH.md
            "mfma_issue_frac_of_sustained": dom["tflops"] * MFMA_PER_PRODUCT[dtype] / sus["value"] if sus["value"] > 0 else None,
            "traffic": hbm_traffic(dtype, dom["kernel"], B, T)[0],
            "traffic_source": hbm_traffic(dtype, dom["kernel"], B, T)[1],
            "traffic_stale": hbm_traffic(dtype, dom["kernel"], B, T)[2],  # True: the counter file was taken with other kernel sources
            "avg_launch_ms": dom["avg_ms"],
            "share_of_step": dom["share"],
            "attention": next(({"avg_launch_ms": r["avg_ms"], "tflops": r["tflops"], "frac": r["tflops"] / peak,
                               "traffic": hbm_traffic(dtype, "attention", B, T)[0]}
                              for r in prof_rows if r["kernel"].startswith("attention")), None),
        }

    roofline = None
    if rank == 0 and not stub:
        sampler.set_cond(cond_dev)
        roofline = roofline_of(sampler, args.dtype, args.profile_out)

    # the other arithmetic modes on rank 0's shard (context, not the headline): one warm-up loop, then `fp32-loops` timed loops
    # for f32 - the reference's own arithmetic, reported as the top-level "fp32" entry - and one for the rest; every mode
    # reported gets its own finiteness test, oracle check and dominant-kernel roofline
    other = {}
    finite_by = {args.dtype: finite}
    if rank == 0 and world == 1 and not stub:
        extra = [d for d in args.also.split(",") if d and d != args.dtype]
        if args.fp32_loops > 0 and args.dtype != "f32" and "f32" not in extra:
            extra.append("f32")
        for dt in extra:
            loops = max(1, args.fp32_loops) if dt == "f32" else 1
            s2 = HipSampler(arch, sd, B, T, N, dt, dev, tab)
            s2.set_cond(cond_dev)
            s2.sample(1, clip0, out)
            sync()
            t1 = time.perf_counter()
            for k in range(loops):
                s2.sample(2 + k, clip0, out)
            sync()
            dt_s = (time.perf_counter() - t1) / loops
            finite_by[dt] = bool(torch.isfinite(out).all().item())
            if dt == "f16x3":
                range_flags[dt] = bool(s2.status_flags() & 1)
            tf = flops_per_clip_step(arch, T) * B * N / dt_s / 1e12
            other[dt] = {"value": B * T / dt_s, "unit": "frames/s", "ms_per_ddpm_step": dt_s / N * 1e3,
                         "whole_path_tflops": tf, "whole_path_frac_of_peak": tf / PEAK_TFLOPS[dt], "finite": finite_by[dt],
                         "timed_loops": loops, "note": f"same workload, {loops} timed loop(s) after 1 warm-up loop"}
            if check_in is not None:
                xc, tc, ref, nc = check_in
                check[dt] = float((s2.denoise(xc.to(dev), tc.to(dev))[:nc].cpu() - ref).abs().max())
            other[dt]["roofline"] = roofline_of(s2, dt)  # this mode's own dominant kernel, HIP events, counted traffic
            s2.close()

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
                "workload": f"{args.arch} B={B}/GPU T={T} {N}-step DDPM ({CONFIGS[args.config]['label']}); step = one full reverse loop"
                + ("; headline dtype f16x3 = split-fp16 operands, fp32-TOLERANCE mode (1e-5 vs the reference, checked in-run); the same workload in the "
                   "reference's own fp32 arithmetic is under roofline.reference_arithmetic (and the top-level fp32 entry)" if args.dtype == "f16x3" else ""),
                "preset": args.config,
                "clips_per_gpu": B,
                "frames": T,
                "ddpm_steps": N,
                "global_clips": world * B,
                "parallelism": f"clip-sharded x{world}, RCCL all_gather of results" if world > 1 else "single GPU",
                "world_size_seen": world_seen,
                "rank_clip_ranges": [i["clips"] for i in sorted(infos, key=lambda i: i["rank"])],
                "ranks": sorted(infos, key=lambda i: i["rank"]),
                "hipgraph": not args.no_graph,
                "kernels_per_ddpm_step": sampler.kernels_per_step,
            },
            "ms_per_ddpm_step": elapsed / args.steps / N * 1e3,
            "whole_path_tflops": whole_tflops,
            "whole_path_frac_of_peak": whole_tflops / (PEAK_TFLOPS[args.dtype] * world),
            "finite": finite,
            "roofline": roofline,
        }
        # strict fp32 (the reference's arithmetic, launch/sample.py:173) as a first-class entry of the line
        if args.dtype == "f32":
            line["fp32"] = {"value": value, "unit": "frames/s", "ms_per_ddpm_step": line["ms_per_ddpm_step"], "timed_loops": args.steps,
                            "whole_path_frac_of_peak": line["whole_path_frac_of_peak"], "roofline": roofline, "finite": finite}
        elif "f32" in other:
            o = other["f32"]
            line["fp32"] = {k: o[k] for k in ("value", "unit", "ms_per_ddpm_step", "timed_loops", "whole_path_tflops",
                                              "whole_path_frac_of_peak", "finite", "roofline")}
            line["fp32"]["what"] = ("the same workload in the reference's own arithmetic (v_mfma_f32_16x16x4_f32: exact fp32 products, "
                                    "fp32 accumulate), 1 GPU")
        # ... and INSIDE `roofline` (drivers that keep only the contract's top-level keys keep this one): the credited number at the
        # reference's precision beside the fp32-tolerance headline
        if roofline is not None and "fp32" in line:
            fr = line["fp32"].get("roofline") or {}
            roofline["reference_arithmetic"] = {
                "dtype": "f32", "value": line["fp32"]["value"], "unit": "frames/s", "ms_per_ddpm_step": line["fp32"]["ms_per_ddpm_step"],
                "timed_loops": line["fp32"].get("timed_loops"), "whole_path_frac_of_peak": line["fp32"].get("whole_path_frac_of_peak"),
                "kernel": fr.get("kernel"), "achieved": fr.get("achieved"), "peak": fr.get("peak"), "frac": fr.get("frac"),
                "attention_frac": (fr.get("attention") or {}).get("frac"),
                "what": "the same workload in the reference's own arithmetic (exact fp32 MFMA products, fp32 accumulate)"}
        # ... and once more as FLAT SCALARS of `roofline` (a parser that keeps only scalar members of the contract's objects - the round-4
        # driver did - still carries the credited strict-fp32 number beside the fp32-tolerance headline)
        if roofline is not None:
            roofline["attention_frac"] = (roofline.get("attention") or {}).get("frac")
            roofline["mfma_sustained_tflops"] = (roofline.get("mfma_sustained") or {}).get("value")
            if "fp32" in line:
                fr = line["fp32"].get("roofline") or {}
                roofline["f32_value"] = line["fp32"]["value"]
                roofline["f32_ms_per_ddpm_step"] = line["fp32"]["ms_per_ddpm_step"]
                roofline["f32_frac"] = fr.get("frac")
                roofline["f32_whole_path_frac"] = line["fp32"].get("whole_path_frac_of_peak")
                roofline["f32_attention_frac"] = (fr.get("attention") or {}).get("frac")
                roofline["f32_kernel"] = fr.get("kernel")
        if power is not None:
            # (rank 0's GPU; every rank runs its own clips through the same number of DDPM steps in ms_per_ddpm_step)
            power["joules_per_ddpm_step_per_gpu"] = power["watts"] * line["ms_per_ddpm_step"] * 1e-3 if power.get("watts") else None
            line["power"] = power
        if roofline is not None and world == 1 and B == 64 and T == 196 and args.arch == "arch_mdm_l":
            pmod = power_model(args.dtype, power, line["ms_per_ddpm_step"])
            if pmod is not None:
                roofline["power_model"] = pmod
                # the clock was pulled below its maximum by the power management during the timed loops: the step is energy-bound
                if power and power.get("sclk_mhz") and power["sclk_mhz"] < 2350.0:
                    roofline["bound"] = "power"
                    roofline["bound_note"] = ("shader clock %.0f MHz < 2400 at %.0f W during the timed loops: the power management, not the MFMA issue rate, "
                                              "sets the step time; frac stays achieved / nominal MFMA peak" % (power["sclk_mhz"], power["watts"]))
        if check:
            line["check"] = {"max_abs_err_vs_oracle": check, "tolerance": {d: CHECK_TOL[d] for d in check},
                             "what": f"one denoiser evaluation (t = N/2, 0, N-1, N/4, 3N/4, 1, N-2, N/3 by clip) of the first "
                             f"{min(args.check_clips, B)} clips of the bench batch vs oracle.denoiser_forward (fp32 torch-CPU restatement of the reference), outputs O(1)"}
        ok = checks_ok(finite_by, range_flags, check)
        if range_flags:
            line["f16_range_flag"] = range_flags  # True = an operand left the fp16 range during the timed loops (tamf_get_status_flags)
        line["finite_by_dtype"] = finite_by
        line["check_ok"] = ok  # every reported dtype: finite samples, oracle check inside its tolerance, no range flag
        if other:
            line["other_dtypes"] = other
        if not args.no_torch_baseline and world == 1 and not stub:
            # same-box library yardstick: the reference's own op sequence through PyTorch-ROCm (hipBLASLt / SDPA), after every timed region
            try:
                tb = torch_rocm_baseline(args.arch, sd, cond, B, T, N, dev, check_in)
            except Exception as e:  # noqa: BLE001  (a yardstick that cannot run must not take the bench line with it)
                tb = {"error": f"{type(e).__name__}: {str(e)[:300]}", "variants": {}}
            line["torch_rocm_baseline"] = tb
            mine = {"f32": (line.get("fp32") or {}).get("value"), "bf16": (other.get("bf16") or {}).get("value") if args.dtype != "bf16" else value,
                    args.dtype: value}
            tb["product_over_library"] = {
                "headline_%s_over_library_f32" % args.dtype: value / tb["best_f32"]["value"] if "best_f32" in tb else None,
                "f32_over_library_f32": mine["f32"] / tb["best_f32"]["value"] if mine.get("f32") and "best_f32" in tb else None,
                "bf16_over_library_bf16": mine["bf16"] / tb["best_bf16"]["value"] if mine.get("bf16") and "best_bf16" in tb else None}
        if not args.no_cpu_baseline and world == 1 and not stub:  # at N = 1 only (one host measurement, not one per scaling point)
            line["cpu_baseline"] = cpu_baseline(args.arch, sd, B, T, N)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
        if not line["check_ok"]:
            print("bench.py: in-run check FAILED (see check / finite_by_dtype / f16_range_flag in the line)", file=sys.stderr, flush=True)
            rc = 3
    sampler.close()
    if world > 1:
        dist.barrier()  # rank 0 was still profiling / printing
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
