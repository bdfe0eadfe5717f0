"""sample.sh-compatible launcher of the G stage on MI355X (reference launch/sample.py:54-295, script/sample.sh:33-41).

    python -m oakink2_tamf_amd.launch.sample --cfg config/arch_mdm_l.yml --debug.model_weight_filepath CKPT \
        --debug.sample_save_offset test/arch_mdm_l__0399 --runtime.device_id 0,1,2,3 --commit \
        [--data.cond_npz clips.npz | --synthetic B,T]

Same flags, same yml schema (`model: {input_dim, ..., activation}`, repeated --cfg merged in order, dotted
overrides such as --model.latent_dim 512), same output tree
`<cwd>/common/sample/<exp_id>/sample/<offset>/<sample_id:06d>.npy` float32 (T, 99), nothing written without
--commit.  Differences: clips are sampled in batches (--runtime.batch_size, default 64) instead of one by one;
workers default to ONE process per visible GPU (the reference's default is 8 workers on devices 0-3, launch/sample.py:114-127:
with batched sampling a second context on the same GPU only contends for it - two B <= 64 contexts on one MI355X take as long
as one after the other, DESIGN.md section 6 - and GPUs 4-7 of an 8-GPU node would idle); --runtime.num_worker / --runtime.device_id
still override, at most two workers per listed device; the dataset toolkit (thirdparty/OakInk2, absent) is replaced by either a
pre-collated conditioning file (--data.cond_npz: arrays text_embedding, hand_side, shape, obj_embedding, obj_traj
with a leading clip axis - the tensors InteractionSegmentData + interaction_segment_collate produce)
or synthetic conditioning (--synthetic B,T).
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
from typing import Dict, List

import numpy as np

from ..hip_backend import DEFAULT_PRECISION, hand_side_code
from .formats import write_sample_npy
from .upkeep import ckpt_opt, ckpt_setup, decode_file_macro

_logger = logging.getLogger("oakink2_tamf_amd.launch.sample")
PROG = "sample"

MODEL_DEFAULTS = dict(input_dim=99, obj_input_dim=9, hand_shape_dim=10, obj_embed_dim=768, latent_dim=256, ff_size=1024,
                      num_layers=8, num_heads=4, dropout=0.1, activation="gelu")


def _set_dotted(cfg: Dict, key: str, value):
    cur = cfg
    parts = key.split(".")
    for p in parts[:-1]:
        cur = cur.setdefault(p, {})
    cur[parts[-1]] = value


def _merge(dst: Dict, src: Dict):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def parse_args(argv: List[str]):
    ap = argparse.ArgumentParser(prog="oakink2_tamf_amd.launch.sample", allow_abbrev=False)
    ap.add_argument("--cfg", action="append", default=[], help="yml preset; may be repeated, merged in order")
    ap.add_argument("--exp_id", default="main")
    ap.add_argument("--commit", action="store_true", help="write outputs (dry run otherwise)")
    ap.add_argument("--synthetic", default=None, help="B,T : synthetic conditioning for B clips of T frames")
    ap.add_argument("--precision", default=DEFAULT_PRECISION, choices=["f32", "f16x3", "bf16x3", "bf16"],
                    help="MFMA operand format (default: the package-wide default, fp32-equivalent split fp16 with range fallback to f32)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--diffusion_steps", type=int, default=1000)
    known, rest = ap.parse_known_args(argv)
    dotted = {}
    i = 0
    while i < len(rest):
        tok = rest[i]
        if not tok.startswith("--") or "." not in tok:
            ap.error(f"unrecognised argument {tok}")
        if i + 1 >= len(rest):
            ap.error(f"{tok} needs a value")
        dotted[tok[2:]] = rest[i + 1]
        i += 2
    return known, dotted


def build_config(known, dotted) -> Dict:
    import yaml

    # runtime defaults: one worker per visible device (None = decided in main(); the reference's 8 workers over devices 0-3,
    # launch/sample.py:114-127, would put two contexts on each of four GPUs and leave the other four idle)
    cfg: Dict = {"model": dict(MODEL_DEFAULTS), "data": {}, "debug": {}, "runtime": {"num_worker": None, "device_id": None, "batch_size": 64}}
    for path in known.cfg:
        with open(path) as f:
            _merge(cfg, yaml.safe_load(f) or {})
    for k, v in dotted.items():
        if k == "runtime.device_id":
            v = [int(x) for x in str(v).split(",") if x != ""]
        elif k.startswith("model.") and k.split(".")[1] in MODEL_DEFAULTS:
            typ = type(MODEL_DEFAULTS[k.split(".")[1]])
            v = typ(v)
        elif k in ("runtime.num_worker", "runtime.batch_size"):
            v = int(v)
        elif k == "data.process_range":  # comma list; ?(file:<path>) entries expand to the file's lines (upkeep.decode_file_macro)
            v = decode_file_macro([x for x in str(v).split(",") if x != ""])
        _set_dotted(cfg, k, v)
    cfg["exp_id"] = known.exp_id
    cfg["commit"] = known.commit
    cfg["ckpt_path"] = os.path.join(os.getcwd(), "common", PROG, known.exp_id)
    return cfg


def load_conditioning(cfg, known):
    """-> dict of numpy arrays with a leading clip axis."""
    if known.synthetic:
        B, T = (int(x) for x in known.synthetic.split(","))
        rng = np.random.default_rng(known.seed)
        shape = np.repeat(rng.standard_normal((B, 1, 10)).astype(np.float32), T, axis=1)
        return {
            "text_embedding": rng.standard_normal((B, 512)).astype(np.float32),
            "hand_side": np.array([b % 2 for b in range(B)], dtype=np.uint8),
            "shape": shape,
            "obj_embedding": rng.standard_normal((B, 2, cfg["model"]["obj_embed_dim"])).astype(np.float32),
            "obj_traj": rng.standard_normal((B, 2, T, cfg["model"]["obj_input_dim"])).astype(np.float32),
        }
    path = cfg["data"].get("cond_npz")
    if not path:
        raise SystemExit(
            "no clips to sample: the OakInk2 dataset toolkit (thirdparty/OakInk2) is not available in this build; pass "
            "--data.cond_npz <file> with pre-collated conditioning tensors or --synthetic B,T")
    with np.load(path, allow_pickle=False) as z:
        cond = {k: z[k] for k in ("text_embedding", "hand_side", "shape", "obj_embedding", "obj_traj")}
        keys = z["process_key"] if "process_key" in z.files else None
    pr = cfg["data"].get("process_range")
    if pr is not None:
        # the reference walks the dataset's clips of the listed process keys (launch/sample.py:161-166); here the clips
        # are the rows of the .npz, selected by its `process_key` column
        if keys is None:
            raise SystemExit("--data.process_range needs a `process_key` array in --data.cond_npz")
        want = set(pr)
        sel = np.array([i for i, k in enumerate(keys) if str(k.decode() if isinstance(k, bytes) else k) in want], dtype=np.int64)
        cond = {k: v[sel] for k, v in cond.items()}
    return cond


def sample_worker(worker_id: int, num_worker: int, device_id: int, cfg: Dict, cond: Dict, known_seed: int, precision: str,
                  diffusion_steps: int):
    import torch

    from ..model.diffusion_util import create_gaussian_diffusion
    from ..model.interaction_segment_mdm import InterationSegmentMDM
    from ..shard import worker_range

    logging.basicConfig(level=logging.INFO, format=f"worker {worker_id:02d} | %(message)s")
    device = torch.device(f"cuda:{device_id}")
    torch.cuda.set_device(device)
    mc = cfg["model"]
    n = int(cond["shape"].shape[0])
    start, stop = worker_range(n, worker_id, num_worker)
    bs = int(cfg["runtime"].get("batch_size", 64))
    T = int(cond["shape"].shape[1])
    torch.manual_seed(known_seed)  # without a checkpoint every worker draws the SAME random weights (the split of the clips must not change a sample)
    model = InterationSegmentMDM(**mc, precision=precision, max_batch=min(bs, max(stop - start, 1)), max_frames=T).to(device)
    diffusion = create_gaussian_diffusion(diffusion_steps=diffusion_steps, noise_schedule="cosine")
    wpath = cfg["debug"].get("model_weight_filepath")
    if wpath:
        state_dict = torch.load(wpath, map_location="cpu")
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        missing = [k for k in missing if not k.startswith("clip_model")]
        if worker_id == 0:
            _logger.info("missing_keys: %s", missing)
            _logger.info("unexpected_keys: %s", unexpected)
    else:
        _logger.warning("no --debug.model_weight_filepath: sampling from randomly initialised weights")
    _logger.info("%06d %06d", start, stop)
    for b0 in range(start, stop, bs):
        b1 = min(b0 + bs, stop)
        batch = {
            "text_embedding": torch.from_numpy(cond["text_embedding"][b0:b1]).to(device),
            "hand_side": ["rh" if hand_side_code(v) == 0 else "lh" for v in cond["hand_side"][b0:b1]],
            "shape": torch.from_numpy(cond["shape"][b0:b1]).to(device),
            "obj_embedding": torch.from_numpy(cond["obj_embedding"][b0:b1]).to(device),
            "obj_traj": torch.from_numpy(cond["obj_traj"][b0:b1]).to(device),
        }
        shape = (b1 - b0, mc["input_dim"], 1, T)
        sample = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"batch": batch}, skip_timesteps=0,
                                         init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False,
                                         seed=known_seed, clip_id_base=b0)
        sample_np = sample.permute((0, 3, 1, 2)).detach().cpu().numpy().squeeze(3)  # (b, T, 99)
        for j, sample_id in enumerate(range(b0, b1)):
            if cfg["commit"]:
                write_sample_npy(cfg["ckpt_path"], cfg["debug"].get("sample_save_offset", ""), sample_id, sample_np[j])
            _logger.info("sample %06d", sample_id)


def main(argv=None):
    known, dotted = parse_args(sys.argv[1:] if argv is None else argv)
    cfg = build_config(known, dotted)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    ckpt_setup(cfg, argv=sys.argv[1:] if argv is None else argv)
    ckpt_opt(cfg)
    cond = load_conditioning(cfg, known)
    import torch
    import torch.multiprocessing as mp

    # (device_count() may call hipGetDeviceCount in THIS process; harmless here: the workers below are fresh `spawn` processes, this
    #  parent is never re-executed, and with one worker the parent is the worker)
    n_dev = max(torch.cuda.device_count(), 1)
    want = cfg["runtime"].get("device_id")
    device_ids = ([d for d in want if d < n_dev] or [0]) if want else list(range(n_dev))
    n_clips = int(cond["shape"].shape[0])
    asked = cfg["runtime"].get("num_worker")
    num_worker = int(asked) if asked else len(device_ids)
    if num_worker > 2 * len(device_ids):
        _logger.warning("runtime.num_worker=%d on %d device(s): clamped to two per device", num_worker, len(device_ids))
        num_worker = 2 * len(device_ids)
    num_worker = max(1, min(num_worker, n_clips))

    if num_worker == 1:
        sample_worker(0, 1, device_ids[0], cfg, cond, known.seed, known.precision, known.diffusion_steps)
        return 0
    mp.set_start_method("spawn", force=True)
    procs = []
    for w in range(num_worker):
        p = mp.Process(target=sample_worker, args=(w, num_worker, device_ids[w % len(device_ids)], cfg, cond, known.seed,
                                                   known.precision, known.diffusion_steps))
        p.start()
        procs.append(p)
    rc = 0
    for p in procs:
        p.join()
        rc = rc or p.exitcode  # the reference ignores worker exit codes (launch/sample.py:291-292); we do not
    return rc


if __name__ == "__main__":
    sys.exit(main())
