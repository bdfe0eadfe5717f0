"""GPU: BASELINE.json configs[3] - arch_refine (MF-MDM R) trunk, B=64, T=196, single forward on cached G samples
(synthetic here) + the h2o distance feature and the pose decode that surround it.  Reports ms per batch."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
import torch
from oakink2_tamf_amd.hip_backend import TamfContext
from oakink2_tamf_amd import geometry
from oracle import mdm_oracle as O, det
B, T = 64, 196
arch = O.ARCH_REFINE
sd = O.det_state_dict(arch, tag="bench_r/w")
cond = O.det_cond(B, T, tag="bench_r/c", arch=arch)
x_in = torch.from_numpy(det.det_normal("bench_r/x", (B, T, 99))).cuda()
h2o = (torch.from_numpy(det.det_normal("bench_r/h", (B, T, 778))) * 0.05).cuda()
res = {}
for prec in ("f16x3", "bf16x3", "bf16", "f32"):
    ctx = TamfContext(dict(latent_dim=256, ff_size=1024, num_layers=8, num_heads=4), B, T, precision=prec, kind="R")
    ctx.load_state_dict(sd)
    ctx.set_cond(None, cond["hand_side"], cond["shape"].cuda(), cond["obj_embedding"].cuda(), cond["obj_traj"].cuda())
    for _ in range(3): out = ctx.refine(x_in, h2o)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 20
    for _ in range(n): out = ctx.refine(x_in, h2o)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / n * 1e3
    ref = O.refine_forward(sd, arch, x_in[:2].cpu(), h2o[:2].cpu(), {k: (v[:2] if not isinstance(v, list) else v[:2]) for k, v in cond.items()})
    err = float((out[:2].cpu() - ref).abs().max())
    res[prec] = {"ms_per_batch": ms, "frames_per_s": B * T / ms * 1e3, "max_abs_err_vs_oracle_2clips": err}
    ctx.close()
print(json.dumps({"config": "arch_refine trunk B=64 T=196 single forward (BASELINE configs[3])", "results": res}))
