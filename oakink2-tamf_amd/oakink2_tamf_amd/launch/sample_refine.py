"""sample_refine.sh-compatible launcher of the R stage on MI355X (reference launch/sample_refine.py:52-296,
script/sample_refine.sh).

    python -m oakink2_tamf_amd.launch.sample_refine --data.process_range "?(file:./asset/split/test.txt)" \
        --data.cache_dict_filepath common/save_cache_dict/main/cache/test.pkl --debug.model_weight_filepath CKPT \
        --debug.sample_save_offset test/arch_mdm_l__0399 --commit            [--cfg config/arch_refine.yml --mano.factory pkg.mod:make_mano]

i.e. script/sample_refine.sh's own argument list (+ the MANO factory, below).  Same flags, same defaults for the data files
(`common/save_cache_dict/main/cache/test.pkl`, `common/retrieve_obj_embedding/main/embedding`,
`common/retrieve_obj_pointcloud/main/pointcloud`, reference :49-100), same output:
`<cwd>/common/sample_refine/<exp_id>/sample/<offset>/<process_key '/'->'++'>/<info[1]>/<info[2]>/save_dict.pkl` with the keys of
`launch/formats.py:REFINE_KEYS` (reference :274-296); nothing is written without --commit; unknown dotted flags are an error.

The clips: the segment cache through `dataset.interaction_segment.InteractionSegmentData` (object embeddings + point clouds) joined
with the G stage's `.npy` tree by `dataset.pose_repr_sample.GeneratedPoseReprSampleAdaptor` (reference :156-171).  The reference
hard-codes the tree as `common/sample/main/sample/test/arch_mdm_l__0399` (:170); here it is `--data.pose_repr_sample_dir_list`
(colon / comma separated, the key of config/refine_sample_param.yml) and defaults to `common/sample/main/sample/<offset>` - the
directory `launch.sample` wrote for the same `--debug.sample_save_offset`.  `--data.clips_pkl` (a pickled list of ready item dicts)
is kept as the toolkit-free alternative.  Clips whose `info` was seen before are skipped (:217-222).  Clips are refined
`--runtime.batch_size` at a time (default 64; the reference: one per forward).

MANO (licence-gated assets, manotorch): `--mano.factory module:function`; the function receives `cfg["mano"]` and the torch device and
returns `(layer_rh, layer_lh, faces_closed_rh, faces_closed_lh)` - the two `ManoLayer(rot_mode="quat", center_idx=0, use_pca=False,
flat_hand_mean=True)` objects of the reference (:175-194) and their `get_mano_closed_faces()` arrays.
The model runs the reference's forward on the GPU (HIP pose decode -> MANO -> HIP hand->object distance -> HIP trunk), then the
refined pose is pushed through pose decode + MANO once more for the joints / vertices of the save dict (:254-272).
"""
from __future__ import annotations

import importlib
import logging
import os
import pickle
import sys
from typing import Dict, Iterable, Iterator, List

import numpy as np

from . import formats
from .sample import DEFAULT_CACHE_DICT, _abspath, _merge, _set_dotted, parse_args

_logger = logging.getLogger("oakink2_tamf_amd.launch.sample_refine")
PROG = "sample_refine"

MODEL_DEFAULTS = dict(input_dim=99, obj_input_dim=9, hand_shape_dim=10, obj_embed_dim=768, latent_dim=256, ff_size=1024,
                      num_layers=8, num_heads=4, dropout=0.1, activation="gelu")
DATA_DEFAULTS = dict(obj_embedding_prefix=os.path.join("common", "retrieve_obj_embedding", "main", "embedding"),   # :74-81
                     obj_pointcloud_prefix=os.path.join("common", "retrieve_obj_pointcloud", "main", "pointcloud"),  # :82-89
                     cache_dict_filepath=DEFAULT_CACHE_DICT)                                                        # :92-99
DEVICE_FIELDS = ("mask", "pose_repr", "shape", "obj_num", "obj_traj", "obj_embedding", "sample_pose_repr")  # the `select` of :229-234


def build_config(known, dotted) -> Dict:
    import yaml

    cfg: Dict = {"model": dict(MODEL_DEFAULTS), "data": {}, "debug": {}, "mano": {}, "runtime": {"device_id": [0], "batch_size": 64}}
    for path in known.cfg:
        with open(path) as f:
            _merge(cfg, yaml.safe_load(f) or {})
    for k, v in dotted.items():
        _set_dotted(cfg, k, v)
    for k, v in DATA_DEFAULTS.items():
        cfg["data"][k] = _abspath(cfg["data"].get(k) or v)
    cfg["exp_id"] = known.exp_id
    cfg["commit"] = known.commit
    cfg["ckpt_path"] = formats.ckpt_path(PROG, known.exp_id)
    return cfg


def load_mano(cfg, device):
    spec = cfg["mano"].get("factory")
    if not spec or ":" not in str(spec):
        raise SystemExit("the refine stage needs MANO: pass --mano.factory module:function returning "
                         "(layer_rh, layer_lh, faces_closed_rh, faces_closed_lh); the MANO assets are licence-gated and not shipped")
    mod, fn = str(spec).split(":", 1)
    return getattr(importlib.import_module(mod), fn)(cfg["mano"], device)


def sample_dir_list(cfg) -> List[str]:
    dirs = cfg["data"].get("pose_repr_sample_dir_list")
    if dirs:
        return [_abspath(d) for d in dirs]
    return [_abspath(os.path.join("common", "sample", "main", "sample", cfg["debug"].get("sample_save_offset") or ""))]


def load_clips(cfg):
    """-> a sequence of item dicts (dataset fields + sample_info + sample_pose_repr + obj_pointcloud)"""
    path = cfg["data"].get("clips_pkl")
    if path:
        with open(path, "rb") as f:
            return list(pickle.load(f))
    from ..dataset.interaction_segment import InteractionSegmentData, load_cache_dict
    from ..dataset.pose_repr_sample import GeneratedPoseReprSampleAdaptor

    d = cfg["data"]
    if not os.path.exists(d["cache_dict_filepath"]):
        raise SystemExit(f"no clips to refine: segment cache {d['cache_dict_filepath']} not found; pass --data.cache_dict_filepath <pkl> "
                         "(+ --data.obj_embedding_prefix, --data.obj_pointcloud_prefix, --data.pose_repr_sample_dir_list) or --data.clips_pkl <file>")
    dataset = InteractionSegmentData(process_range_list=d.get("process_range"), data_prefix=d.get("data_prefix"),
                                     obj_embedding_prefix=d["obj_embedding_prefix"], enable_obj_model=True,
                                     obj_pointcloud_prefix=d["obj_pointcloud_prefix"], append_reverse_segment=False,
                                     cache_dict=load_cache_dict(d["cache_dict_filepath"]))
    dirs = sample_dir_list(cfg)
    missing = [p for p in dirs if not os.path.isdir(p)]
    if missing:
        raise SystemExit(f"G-stage sample directory not found: {missing[0]} (run launch.sample with the same --debug.sample_save_offset "
                         "and --commit first, or pass --data.pose_repr_sample_dir_list)")
    return GeneratedPoseReprSampleAdaptor(dataset, dirs)


def max_sample_frames(clips) -> int:
    """longest G sample of the clip source WITHOUT materialising its items (item i of the adaptor converts poses to rot6d and
    stacks trajectories, embeddings and point clouds - the old max() over clips[i] built every item once here and once more in
    unique_clips()): the adaptor holds the loaded .npy arrays, a --data.clips_pkl list is scanned as it is"""
    arrays = getattr(clips, "pose_repr_map", None)
    if arrays:
        return max(int(np.asarray(a).shape[0]) for a in arrays.values())
    return max(int(np.asarray(clips[i]["sample_pose_repr"]).shape[0]) for i in range(len(clips)))


def unique_clips(clips) -> Iterator:
    """(sample_id, item) of the first clip of every `info`: the reverse segments repeat their forward twin's (:217-222)"""
    seen = set()
    for sample_id in range(len(clips)):
        clip = clips[sample_id]
        info = clip["info"]
        key = tuple(info) if isinstance(info, (list, tuple)) else info
        if key in seen:
            continue
        seen.add(key)
        yield sample_id, clip


def batches(pairs: Iterable, batch_size: int) -> Iterator[List]:
    """consecutive clips of equal length, at most batch_size of them"""
    cur: List = []
    for sample_id, clip in pairs:
        T = int(np.asarray(clip["sample_pose_repr"]).shape[0])
        if cur and (len(cur) >= batch_size or int(np.asarray(cur[0][1]["sample_pose_repr"]).shape[0]) != T):
            yield cur
            cur = []
        cur.append((sample_id, clip))
    if cur:
        yield cur


def refine_clips(model, mano, clips: List[Dict], device) -> List[Dict]:
    """B clips through R in one forward, then each through MANO -> the reference's save_dicts (launch/sample_refine.py:224-285)"""
    import torch

    from ..dataset.batching import interaction_segment_collate
    from ..geometry import pose_repr_to_quat

    layer_rh, layer_lh, faces_rh, faces_lh = mano
    dev_batch = interaction_segment_collate(clips)
    for k in DEVICE_FIELDS:
        if isinstance(dev_batch.get(k), torch.Tensor):
            t = dev_batch[k]
            dev_batch[k] = t.to(device=device, dtype=torch.float32) if t.is_floating_point() else t.to(device)
    out = model(dev_batch)
    res = []
    for b, clip in enumerate(clips):
        refined = out["refine_pose_repr"][b]  # (T, 99)
        hand_side = clip["hand_side"]
        if hand_side not in ("rh", "lh"):
            raise ValueError(f"unexpected hand_side: {hand_side}")
        tsl, quat = pose_repr_to_quat(refined)
        shape = torch.as_tensor(clip["shape"]).to(device=device, dtype=torch.float32)
        mo = (layer_rh if hand_side == "rh" else layer_lh)(pose_coeffs=quat, betas=shape)
        joints = (mo.joints + tsl.unsqueeze(1)).detach().cpu().numpy()
        verts = (mo.verts + tsl.unsqueeze(1)).detach().cpu().numpy()
        res.append(formats.build_refine_save_dict(clip["info"], hand_side, joints, verts, faces_rh if hand_side == "rh" else faces_lh,
                                                  clip["obj_list"], clip["len"], clip["frame_id"], refined.detach().cpu().numpy()))
    return res


def refine_clip(model, mano, clip: Dict, device) -> Dict:
    return refine_clips(model, mano, [clip], device)[0]


def main(argv=None):
    import torch

    from ..model.segment_refine_model import SegmentRefineModel

    known, dotted = parse_args(sys.argv[1:] if argv is None else argv, prog="oakink2_tamf_amd.launch.sample_refine")
    cfg = build_config(known, dotted)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    from .upkeep import ckpt_opt, ckpt_setup

    ckpt_setup(cfg, argv=sys.argv[1:] if argv is None else argv)
    device = torch.device(f"cuda:{cfg['runtime']['device_id'][0]}")
    torch.cuda.set_device(device)
    mano = load_mano(cfg, device)
    clips = load_clips(cfg)
    mc = cfg["model"]
    bs = max(1, int(cfg["runtime"].get("batch_size", 64)))
    T_max = max_sample_frames(clips)
    model = SegmentRefineModel(cfg["mano"].get("mano_path"), **mc, use_pc=True, precision=known.precision, max_batch=min(bs, len(clips)),
                               max_frames=T_max, mano_layer_rh=mano[0], mano_layer_lh=mano[1], per_clip_object_mean=True).to(device)
    wpath = cfg["debug"].get("model_weight_filepath")
    if wpath:
        missing, unexpected = model.load_state_dict(torch.load(wpath, map_location="cpu"), strict=False)
        _logger.info("missing_keys: %s", [k for k in missing if not k.startswith("clip_model")])
        _logger.info("unexpected_keys: %s", unexpected)
    else:
        _logger.warning("no --debug.model_weight_filepath: refining with randomly initialised weights")
    ckpt_opt(cfg)
    n_refined = n_written = 0
    for group in batches(unique_clips(clips), bs):
        for sample_id, _ in group:
            _logger.info("sample_id: %d", sample_id)
        for save_dict in refine_clips(model, mano, [c for _, c in group], device):
            n_refined += 1
            if cfg["commit"]:
                formats.write_refine_sample(cfg["ckpt_path"], cfg["debug"].get("sample_save_offset") or "", save_dict)
                n_written += 1
    _logger.info("refined %d clips, wrote %d", n_refined, n_written)
    return 0


if __name__ == "__main__":
    sys.exit(main())
