"""GPU: print max|err| of every golden case for each arithmetic mode (used to state tolerances in DESIGN.md)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from conftest import golden_cond, load_golden
from oracle import det, mdm_oracle as O
from test_hip_forward import _make_ctx, _set_cond

import subprocess
print("commit", subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or os.environ.get("TAMF_COMMIT", "(snapshot without .git: see the file name / commit message)"))
for prec in ("f32", "f16x3", "bf16x3", "bf16"):
    for name, arch in (("tiny", O.ARCH_TINY), ("arch_mdm", O.ARCH_MDM), ("arch_mdm_l", O.ARCH_MDM_L), ("arch_mdm_l_t196", O.ARCH_MDM_L)):
        fix = load_golden(f"forward_{name}.npz"); sd = O.det_state_dict(arch, tag=f"{name}/w")
        x = torch.from_numpy(fix["x"]); B, _, _, T = x.shape
        ctx = _make_ctx(arch, sd, B, T, prec); _set_cond(ctx, golden_cond(fix))
        errs = [np.abs(ctx.denoise(x, torch.full((B,), int(t), dtype=torch.long)).cpu().numpy() - fix[f"out/t{int(t)}"]).max() for t in fix["ts"]]
        print(f"{prec:7s} forward {name:16s} max|err| = {max(errs):.3e}"); ctx.close()
    for name, arch, B, T, N in (("arch_mdm_b4_t64_50", O.ARCH_MDM, 4, 64, 50), ("tiny_1000", O.ARCH_TINY, 2, 16, 1000)):
        fix = load_golden(f"loop_{name}.npz"); sd = O.det_state_dict(arch, tag=f"{name}/w")
        shape = (B, 99, 1, T)
        draws = torch.from_numpy(np.stack([det.det_normal(det.step_noise_tag(f"{name}/eps", k), shape) for k in range(N + 1)]))
        ctx = _make_ctx(arch, sd, B, T, prec, n_steps=N); _set_cond(ctx, golden_cond(fix))
        out = ctx.sample_loop(noise=draws).cpu().numpy()
        print(f"{prec:7s} loop    {name:20s} max|err| = {np.abs(out - fix['final']).max():.3e}  mean|err| = {np.abs(out - fix['final']).mean():.3e}"); ctx.close()
