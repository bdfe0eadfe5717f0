"""sample.sh-compatible launcher of the G stage on MI355X (reference launch/sample.py:54-295, script/sample.sh:33-41).

    python -m oakink2_tamf_amd.launch.sample --cfg config/obj_embedding.yml --data.process_range "?(file:./asset/split/test.txt)" \
        --data.cache_dict_filepath common/save_cache_dict/main/cache/test.pkl --cfg config/arch_mdm_l.yml \
        --debug.model_weight_filepath CKPT --debug.sample_save_offset test/arch_mdm_l__0399 --runtime.device_id 0,1,2,3 --commit

i.e. script/sample.sh's own argument list.  Same flags, same yml schema (`model: {input_dim, ..., activation}`,
`data: {obj_embedding_prefix}`, repeated --cfg merged in order, dotted overrides such as --model.latent_dim 512), same output tree
`<cwd>/common/sample/<exp_id>/sample/<offset>/<sample_id:06d>.npy` float32 (T, 99) with sample_id = index into the cache dict
(what GeneratedPoseReprSampleAdaptor joins on), nothing written without --commit; a dotted flag the launcher does not know is an
error (config_reg rejects unregistered keys).

Where the clips come from, in this order:
  --synthetic B,T              synthetic conditioning (BASELINE configs);
  --data.cond_npz FILE         pre-collated conditioning tensors (text_embedding, hand_side, shape, obj_embedding, obj_traj with a
                               leading clip axis, optional process_key filtered by --data.process_range);
  --data.cache_dict_filepath   the reference's segment cache (default common/save_cache_dict/main/cache/test.pkl, launch/sample.py:88-93)
                               through dataset.interaction_segment.InteractionSegmentData + interaction_segment_collate, object
                               embeddings from --data.obj_embedding_prefix/<obj_id>.pt.  As in the reference the cache decides the
                               clips; --data.process_range and --data.data_prefix are accepted and not consulted (:312-324).
The CLIP prompt: the cache carries `text` strings and the CLIP tower is not part of this build (its output is an input of the path,
SURVEY.md 8c), so --data.text_embedding_filepath names a pickle {text: (512,) float32} (or an .npz with `text` / `embedding`)
holding `clip_model.encode_text(...).float()` of every distinct prompt; without it the module's own CLIP branch is used if the
`clip` package is importable.

Differences: clips are sampled in batches (--runtime.batch_size, default 64) instead of one by one; workers default to ONE process
per visible GPU (the reference's default is 8 workers on devices 0-3, launch/sample.py:114-127: with batched sampling a second
context on the same GPU only contends for it, and GPUs 4-7 of an 8-GPU node would idle); --runtime.num_worker / --runtime.device_id
still override, at most two workers per listed device; step noise is device Philox keyed by (seed, sample_id) - a clip's sample does
not depend on the batch or worker it landed in; and because the reference calls the model one clip at a time, so that a clip's object
means never include another clip's zero padding, the batched module is built with per_clip_object_mean=True (C-ABI:
tamf_set_cond_ragged with the collate's obj_num).
"""
from __future__ import annotations

import argparse
import logging
import os
import pickle
import sys
from typing import Callable, Dict, List, Optional

import numpy as np

from ..hip_backend import DEFAULT_PRECISION, hand_side_code
from .formats import write_sample_npy
from .upkeep import ckpt_opt, ckpt_setup, decode_file_macro

_logger = logging.getLogger("oakink2_tamf_amd.launch.sample")
PROG = "sample"

MODEL_DEFAULTS = dict(input_dim=99, obj_input_dim=9, hand_shape_dim=10, obj_embed_dim=768, latent_dim=256, ff_size=1024,
                      num_layers=8, num_heads=4, dropout=0.1, activation="gelu")
DEFAULT_CACHE_DICT = os.path.join("common", "save_cache_dict", "main", "cache", "test.pkl")  # launch/sample.py:88-93


def split_outside_macros(value: str, seps: str = ":,") -> List[str]:
    """'a:?(file:x.txt):b' -> ['a', '?(file:x.txt)', 'b']: the reference's COLON_SEP list pattern (launch/sample.py:83), with a
    `?(...)` macro kept whole although it contains the separator; commas separate as well (process keys hold neither)"""
    out, cur, depth = [], "", 0
    for i, ch in enumerate(value):
        if ch == "(" and i > 0 and value[i - 1] == "?":
            depth += 1
        elif ch == ")" and depth:
            depth -= 1
        if ch in seps and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [x for x in out if x != ""]


def _abspath(v: str) -> str:
    return os.path.normpath(os.path.abspath(str(v)))  # config_reg.callback.abspath_callback: relative to the working directory


def _int_list(v) -> List[int]:
    return [int(x) for x in str(v).split(",") if x != ""]


def _str_list(v) -> List[str]:
    return decode_file_macro(split_outside_macros(str(v)))


# every dotted option a launcher registers: name -> converter of its command-line string (launch/sample.py:57-128,
# launch/sample_refine.py:49-116, launch/param/model.py, launch/param/mano.py; the last block is this build's own)
DOTTED_OPTIONS: Dict[str, Callable] = {
    "data.data_prefix": _abspath, "data.obj_embedding_prefix": _abspath, "data.obj_pointcloud_prefix": _abspath,
    "data.process_range": _str_list, "data.cache_dict_filepath": _abspath,
    "debug.model_weight_filepath": _abspath, "debug.sample_save_offset": str,
    "runtime.num_worker": int, "runtime.device_id": _int_list,
    "mano.mano_path": _abspath,
    **{f"model.{k}": type(v) for k, v in MODEL_DEFAULTS.items()},
    "runtime.batch_size": int, "data.cond_npz": _abspath, "data.text_embedding_filepath": _abspath, "data.clips_pkl": _abspath,
    "data.pose_repr_sample_dir_list": lambda v: [_abspath(x) for x in split_outside_macros(str(v))], "mano.factory": str,
}


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


def parse_args(argv: List[str], prog: str = "oakink2_tamf_amd.launch.sample"):
    ap = argparse.ArgumentParser(prog=prog, allow_abbrev=False)
    ap.add_argument("--cfg", action="append", default=[], help="yml preset; may be repeated, merged in order")
    ap.add_argument("--exp_id", default="main")
    ap.add_argument("--commit", action="store_true", help="write outputs (dry run otherwise)")
    ap.add_argument("--synthetic", default=None, help="B,T : synthetic conditioning for B clips of T frames")
    ap.add_argument("--precision", default=DEFAULT_PRECISION, choices=["f32", "f16x3", "bf16x3", "bf16"],
                    help="MFMA operand format (default: the package-wide default, fp32-equivalent split fp16 with range fallback to f32)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--diffusion_steps", type=int, default=1000)
    ap.add_argument("--timestep_respacing", default="", help="this build's own flag: sample over a subset of the trained timesteps "
                    "(section counts of the reference's space_timesteps, e.g. 100 or ddim50; default: every step, as the reference)")
    known, rest = ap.parse_known_args(argv)
    dotted = {}
    i = 0
    while i < len(rest):
        tok = rest[i]
        if not tok.startswith("--") or "." not in tok:
            ap.error(f"unrecognised argument {tok}")
        key = tok[2:]
        if "=" in key:
            key, val = key.split("=", 1)
        elif i + 1 < len(rest):
            val = rest[i + 1]
            i += 1
        else:
            ap.error(f"{tok} needs a value")
        if key not in DOTTED_OPTIONS:
            import difflib

            near = difflib.get_close_matches(key, DOTTED_OPTIONS, n=1)
            ap.error(f"unknown option --{key}" + (f" (did you mean --{near[0]}?)" if near else ""))
        try:
            dotted[key] = DOTTED_OPTIONS[key](val)
        except ValueError as e:
            ap.error(f"--{key}: {e}")
        i += 1
    return known, dotted


def build_config(known, dotted) -> Dict:
    import yaml

    # runtime defaults: one worker per visible device (None = decided in main(); the reference's 8 workers over devices 0-3,
    # launch/sample.py:114-127, would put two contexts on each of four GPUs and leave the other four idle)
    cfg: Dict = {"model": dict(MODEL_DEFAULTS), "data": {}, "debug": {}, "runtime": {"num_worker": None, "device_id": None, "batch_size": 64}}
    for path in known.cfg:
        with open(path) as f:
            _merge(cfg, yaml.safe_load(f) or {})
    for key in ("obj_embedding_prefix", "obj_pointcloud_prefix", "cache_dict_filepath", "data_prefix"):  # abspath also for yml values
        if cfg["data"].get(key):
            cfg["data"][key] = _abspath(cfg["data"][key])
    if isinstance(cfg["data"].get("process_range"), list):
        cfg["data"]["process_range"] = decode_file_macro(cfg["data"]["process_range"])
    for k, v in dotted.items():
        _set_dotted(cfg, k, v)
    cfg["exp_id"] = known.exp_id
    cfg["commit"] = known.commit
    cfg["ckpt_path"] = os.path.join(os.getcwd(), "common", PROG, known.exp_id)
    return cfg


# ---- clip sources ---------------------------------------------------------------------------------------------------------
class ArrayClips:
    """clips given as pre-collated arrays with a leading clip axis (--synthetic, --data.cond_npz)"""

    ragged = False  # pre-collated arrays: the object axis is what the caller made it

    def __init__(self, cond: Dict[str, np.ndarray]):
        self.cond = cond
        self.n = int(cond["shape"].shape[0])
        self.frames = int(cond["shape"].shape[1])

    def batch(self, b0: int, b1: int, device):
        import torch

        c = self.cond
        return {
            "text_embedding": torch.from_numpy(c["text_embedding"][b0:b1]).to(device),
            "hand_side": ["rh" if hand_side_code(v) == 0 else "lh" for v in c["hand_side"][b0:b1]],
            "shape": torch.from_numpy(c["shape"][b0:b1]).to(device),
            "obj_embedding": torch.from_numpy(c["obj_embedding"][b0:b1]).to(device),
            "obj_traj": torch.from_numpy(c["obj_traj"][b0:b1]).to(device),
        }


def load_text_embeddings(path: Optional[str]) -> Optional[Dict[str, np.ndarray]]:
    if not path:
        return None
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {str(t): np.asarray(e, np.float32) for t, e in zip(z["text"], z["embedding"])}
    with open(path, "rb") as f:
        table = pickle.load(f)
    return {str(t): np.asarray(e, np.float32).reshape(-1) for t, e in table.items()}


class CacheDictClips:
    """clips of the reference's segment cache: item -> interaction_segment_collate -> device (launch/sample.py:158-215)"""

    DEVICE_FIELDS = ("mask", "pose_repr", "shape", "obj_num", "obj_traj", "obj_embedding")  # the `select` of :208-213
    ragged = True  # clips of one batch have different object counts: the module averages over each clip's own (batch["obj_num"])

    def __init__(self, cfg: Dict):
        from ..dataset.interaction_segment import InteractionSegmentData, load_cache_dict

        d = cfg["data"]
        if not d.get("obj_embedding_prefix"):
            raise SystemExit("the cache-dict clips need --data.obj_embedding_prefix (config/obj_embedding.yml): the denoiser reads "
                             "batch['obj_embedding'] (interaction_segment_mdm.py:155)")
        self.dataset = InteractionSegmentData(process_range_list=d.get("process_range"), data_prefix=d.get("data_prefix"),
                                              obj_embedding_prefix=d["obj_embedding_prefix"], enable_obj_model=True,
                                              cache_dict=load_cache_dict(d["cache_dict_filepath"]))
        self.n = len(self.dataset)
        self.frames = int(self.dataset.slice_max_len)
        self.text_table = load_text_embeddings(d.get("text_embedding_filepath"))
        if self.text_table is not None:
            missing = sorted(set(self.dataset.texts()) - set(self.text_table))
            if missing:
                raise SystemExit(f"--data.text_embedding_filepath lacks {len(missing)} of the cache's prompts, e.g. {missing[0]!r}")

    @property
    def needs_clip(self) -> bool:
        return self.text_table is None

    def batch(self, b0: int, b1: int, device):
        import torch

        from ..dataset.batching import interaction_segment_collate

        out = interaction_segment_collate([self.dataset[i] for i in range(b0, b1)])
        for k in self.DEVICE_FIELDS:
            if isinstance(out.get(k), torch.Tensor):
                t = out[k]
                out[k] = t.to(device=device, dtype=torch.float32) if t.is_floating_point() else t.to(device)
        if self.text_table is not None:
            out["text_embedding"] = torch.from_numpy(np.stack([self.text_table[t] for t in out["text"]])).to(device)
        return out


def synthetic_conditioning(cfg, B: int, T: int, seed: int) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    shape = np.repeat(rng.standard_normal((B, 1, 10)).astype(np.float32), T, axis=1)
    return {
        "text_embedding": rng.standard_normal((B, 512)).astype(np.float32),
        "hand_side": np.array([b % 2 for b in range(B)], dtype=np.uint8),
        "shape": shape,
        "obj_embedding": rng.standard_normal((B, 2, cfg["model"]["obj_embed_dim"])).astype(np.float32),
        "obj_traj": rng.standard_normal((B, 2, T, cfg["model"]["obj_input_dim"])).astype(np.float32),
    }


def load_clips(cfg, known):
    """-> a clip source (`n`, `frames`, `batch(b0, b1, device)`)"""
    if known.synthetic:
        B, T = (int(x) for x in known.synthetic.split(","))
        return ArrayClips(synthetic_conditioning(cfg, B, T, known.seed))
    path = cfg["data"].get("cond_npz")
    if path:
        with np.load(path, allow_pickle=False) as z:
            cond = {k: z[k] for k in ("text_embedding", "hand_side", "shape", "obj_embedding", "obj_traj")}
            keys = z["process_key"] if "process_key" in z.files else None
        pr = cfg["data"].get("process_range")
        if pr is not None:
            # the clips are the rows of the .npz, selected by its `process_key` column
            if keys is None:
                raise SystemExit("--data.process_range needs a `process_key` array in --data.cond_npz")
            want = set(pr)
            sel = np.array([i for i, k in enumerate(keys) if str(k.decode() if isinstance(k, bytes) else k) in want], dtype=np.int64)
            cond = {k: v[sel] for k, v in cond.items()}
        return ArrayClips(cond)
    if not cfg["data"].get("cache_dict_filepath"):
        cfg["data"]["cache_dict_filepath"] = _abspath(DEFAULT_CACHE_DICT)
    if not os.path.exists(cfg["data"]["cache_dict_filepath"]):
        raise SystemExit(f"no clips to sample: segment cache {cfg['data']['cache_dict_filepath']} not found; pass "
                         "--data.cache_dict_filepath <pkl> (+ --data.obj_embedding_prefix), --data.cond_npz <file> or --synthetic B,T")
    return CacheDictClips(cfg)


def count_clips(cfg, known) -> int:
    """number of clips of the source, for the parent's worker split - without building the clip source where that is expensive (ADVICE
    r5: the parent used to construct CacheDictClips, i.e. torch.load every <obj_id>.pt, only to read `.n`; every worker builds its own)"""
    d = cfg["data"]
    if known.synthetic or d.get("cond_npz"):
        return load_clips(cfg, known).n
    if not d.get("cache_dict_filepath"):
        d["cache_dict_filepath"] = _abspath(DEFAULT_CACHE_DICT)
    if not os.path.exists(d["cache_dict_filepath"]):
        return load_clips(cfg, known).n  # (raises the explanatory SystemExit)
    if not d.get("obj_embedding_prefix") or not os.path.isdir(d["obj_embedding_prefix"]):
        return load_clips(cfg, known).n  # (raises, or fails on the missing directory, as a worker would)
    from ..dataset.interaction_segment import check_cache_dict, load_cache_dict

    cache = load_cache_dict(d["cache_dict_filepath"])
    check_cache_dict(cache, d["cache_dict_filepath"])
    return len(cache["interaction_segment_len_list"])


def sample_worker(worker_id: int, num_worker: int, device_id: int, cfg: Dict, known):
    import torch

    from ..model.diffusion_util import create_gaussian_diffusion
    from ..model.interaction_segment_mdm import InterationSegmentMDM
    from ..shard import worker_range

    logging.basicConfig(level=logging.INFO, format=f"worker {worker_id:02d} | %(message)s")
    device = torch.device(f"cuda:{device_id}")
    torch.cuda.set_device(device)
    mc = cfg["model"]
    clips = load_clips(cfg, known)  # every worker reads the clip source itself, as the reference's workers do (:158-166)
    start, stop = worker_range(clips.n, worker_id, num_worker)
    bs = int(cfg["runtime"].get("batch_size", 64))
    T = clips.frames
    torch.manual_seed(known.seed)  # without a checkpoint every worker draws the SAME random weights (the split of the clips must not change a sample)
    model = InterationSegmentMDM(**mc, precision=known.precision, max_batch=min(bs, max(stop - start, 1)), max_frames=T,
                                 load_clip=bool(getattr(clips, "needs_clip", False)), per_clip_object_mean=clips.ragged).to(device)
    diffusion = create_gaussian_diffusion(diffusion_steps=known.diffusion_steps, noise_schedule="cosine",
                                          timestep_respacing=known.timestep_respacing)
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
        batch = clips.batch(b0, b1, device)
        shape = (b1 - b0, mc["input_dim"], 1, T)
        sample = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"batch": batch}, skip_timesteps=0,
                                         init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False,
                                         seed=known.seed, clip_id_base=b0)
        sample_np = sample.permute((0, 3, 1, 2)).detach().cpu().numpy().squeeze(3)  # (b, T, 99)
        for j, sample_id in enumerate(range(b0, b1)):
            if cfg["commit"]:
                write_sample_npy(cfg["ckpt_path"], cfg["debug"].get("sample_save_offset") or "", sample_id, sample_np[j])
            _logger.info("sample %06d", sample_id)


def main(argv=None):
    known, dotted = parse_args(sys.argv[1:] if argv is None else argv)
    cfg = build_config(known, dotted)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    ckpt_setup(cfg, argv=sys.argv[1:] if argv is None else argv)
    ckpt_opt(cfg)
    n_clips = count_clips(cfg, known)  # (also fails early, in the parent, on a missing cache / embedding directory)
    import torch
    import torch.multiprocessing as mp

    # (device_count() may call hipGetDeviceCount in THIS process; harmless here: the workers below are fresh `spawn` processes, this
    #  parent is never re-executed, and with one worker the parent is the worker)
    n_dev = max(torch.cuda.device_count(), 1)
    want = cfg["runtime"].get("device_id")
    device_ids = ([d for d in want if d < n_dev] or [0]) if want else list(range(n_dev))
    asked = cfg["runtime"].get("num_worker")
    num_worker = int(asked) if asked else len(device_ids)
    if num_worker > 2 * len(device_ids):
        _logger.warning("runtime.num_worker=%d on %d device(s): clamped to two per device", num_worker, len(device_ids))
        num_worker = 2 * len(device_ids)
    num_worker = max(1, min(num_worker, n_clips))

    if num_worker == 1:
        sample_worker(0, 1, device_ids[0], cfg, known)
        return 0
    mp.set_start_method("spawn", force=True)
    procs = []
    for w in range(num_worker):
        p = mp.Process(target=sample_worker, args=(w, num_worker, device_ids[w % len(device_ids)], cfg, known))
        p.start()
        procs.append(p)
    rc = 0
    for p in procs:
        p.join()
        rc = rc or p.exitcode  # the reference ignores worker exit codes (launch/sample.py:291-292); we do not
    return rc


if __name__ == "__main__":
    sys.exit(main())
