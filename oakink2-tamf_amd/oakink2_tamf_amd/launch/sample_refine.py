"""sample_refine.sh-compatible launcher of the R stage on MI355X (reference launch/sample_refine.py:52-296,
script/sample_refine.sh).

    python -m oakink2_tamf_amd.launch.sample_refine --cfg config/arch_refine.yml --debug.model_weight_filepath CKPT \
        --debug.sample_save_offset test/arch_mdm_l__0399 --data.clips_pkl clips.pkl --mano.factory pkg.mod:make_mano --commit

Same flags as the reference where they apply (`--cfg` repeated, `--debug.model_weight_filepath`,
`--debug.sample_save_offset`, `--runtime.device_id`, dotted `--model.*` overrides, `--exp_id`, `--commit`), same output:
`<cwd>/common/sample_refine/<exp_id>/sample/<offset>/<process_key '/'->'++'>/<info[1]>/<info[2]>/save_dict.pkl` with the keys of
`launch/formats.py:REFINE_KEYS` (reference :274-296); nothing is written without --commit.

What replaces the parts that cannot ship:
  * the OakInk2 dataset toolkit + GeneratedPoseReprSampleAdaptor (reference :156-171): `--data.clips_pkl` names a pickle holding the
    list of per-clip dicts those two produce (`InteractionSegmentData.__getitem__` fields + "sample_pose_repr" (T, 99) read from
    the G stage's .npy tree, + "obj_pointcloud" (nobj, P, 3)); duplicates of `info` are skipped as in the reference (:217-222);
  * MANO (licence-gated assets, manotorch): `--mano.factory module:function`; the function receives `cfg["mano"]` and the torch
    device and returns `(layer_rh, layer_lh, faces_closed_rh, faces_closed_lh)` - the two `ManoLayer(rot_mode="quat", center_idx=0,
    use_pca=False, flat_hand_mean=True)` objects of the reference (:175-194) and their `get_mano_closed_faces()` arrays.
The model runs the reference's forward on the GPU (HIP pose decode -> MANO -> HIP hand->object distance -> HIP trunk), then the
refined pose is pushed through pose decode + MANO once more for the joints / vertices of the save dict (:254-272).
"""
from __future__ import annotations

import importlib
import logging
import os
import pickle
import sys
from typing import Dict, List

import numpy as np

from . import formats
from .sample import _merge, _set_dotted, parse_args

_logger = logging.getLogger("oakink2_tamf_amd.launch.sample_refine")
PROG = "sample_refine"

MODEL_DEFAULTS = dict(input_dim=99, obj_input_dim=9, hand_shape_dim=10, obj_embed_dim=768, latent_dim=256, ff_size=1024,
                      num_layers=8, num_heads=4, dropout=0.1, activation="gelu")


def build_config(known, dotted) -> Dict:
    import yaml

    cfg: Dict = {"model": dict(MODEL_DEFAULTS), "data": {}, "debug": {}, "mano": {}, "runtime": {"device_id": [0]}}
    for path in known.cfg:
        with open(path) as f:
            _merge(cfg, yaml.safe_load(f) or {})
    for k, v in dotted.items():
        if k == "runtime.device_id":
            v = [int(x) for x in str(v).split(",") if x != ""]
        elif k.startswith("model.") and k.split(".")[1] in MODEL_DEFAULTS:
            v = type(MODEL_DEFAULTS[k.split(".")[1]])(v)
        _set_dotted(cfg, k, v)
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


def load_clips(cfg) -> List[Dict]:
    path = cfg["data"].get("clips_pkl")
    if not path:
        raise SystemExit("no clips to refine: the OakInk2 dataset toolkit is not available in this build; pass --data.clips_pkl <file> "
                         "with the list of per-clip dicts (dataset fields + sample_pose_repr + obj_pointcloud)")
    with open(path, "rb") as f:
        clips = pickle.load(f)
    return list(clips)


def refine_clip(model, mano, clip: Dict, device, precision_dtype=None) -> Dict:
    """one clip through R and MANO -> the reference's save_dict (launch/sample_refine.py:224-285)"""
    import torch

    from ..dataset.batching import interaction_segment_collate
    from ..geometry import pose_repr_to_quat

    layer_rh, layer_lh, faces_rh, faces_lh = mano
    batch = interaction_segment_collate([clip])
    dev_batch = dict(batch)
    for k in ("mask", "pose_repr", "shape", "obj_num", "obj_traj", "obj_embedding", "sample_pose_repr"):
        if k in dev_batch and isinstance(dev_batch[k], torch.Tensor):
            t = dev_batch[k]
            dev_batch[k] = t.to(device=device, dtype=torch.float32) if t.is_floating_point() else t.to(device)
    out = model(dev_batch)
    refined = out["refine_pose_repr"][0]  # (T, 99)
    hand_side = clip["hand_side"]
    tsl, quat = pose_repr_to_quat(refined)
    shape = torch.as_tensor(clip["shape"]).to(device=device, dtype=torch.float32)
    layer = layer_rh if hand_side == "rh" else layer_lh
    mo = layer(pose_coeffs=quat, betas=shape)
    joints = (mo.joints + tsl.unsqueeze(1)).detach().cpu().numpy()
    verts = (mo.verts + tsl.unsqueeze(1)).detach().cpu().numpy()
    return formats.build_refine_save_dict(clip["info"], hand_side, joints, verts, faces_rh if hand_side == "rh" else faces_lh,
                                          clip["obj_list"], clip["len"], clip["frame_id"], refined.detach().cpu().numpy())


def main(argv=None):
    import torch

    from ..model.segment_refine_model import SegmentRefineModel

    known, dotted = parse_args(sys.argv[1:] if argv is None else argv)
    cfg = build_config(known, dotted)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    from .upkeep import ckpt_opt, ckpt_setup

    ckpt_setup(cfg, argv=sys.argv[1:] if argv is None else argv)
    device = torch.device(f"cuda:{cfg['runtime']['device_id'][0]}")
    torch.cuda.set_device(device)
    mano = load_mano(cfg, device)
    clips = load_clips(cfg)
    mc = cfg["model"]
    T_max = max(int(np.asarray(c["sample_pose_repr"]).shape[0]) for c in clips)
    model = SegmentRefineModel(cfg["mano"].get("mano_path"), **mc, use_pc=True, precision=known.precision, max_batch=1, max_frames=T_max,
                               mano_layer_rh=mano[0], mano_layer_lh=mano[1]).to(device)
    wpath = cfg["debug"].get("model_weight_filepath")
    if wpath:
        missing, unexpected = model.load_state_dict(torch.load(wpath, map_location="cpu"), strict=False)
        _logger.info("missing_keys: %s", [k for k in missing if not k.startswith("clip_model")])
        _logger.info("unexpected_keys: %s", unexpected)
    else:
        _logger.warning("no --debug.model_weight_filepath: refining with randomly initialised weights")
    ckpt_opt(cfg)
    seen = set()
    n_written = 0
    for sample_id, clip in enumerate(clips):
        info = clip["info"]
        key = tuple(info) if isinstance(info, (list, tuple)) else info
        if key in seen:  # the reverse segments repeat their forward twin's info (:217-222)
            continue
        seen.add(key)
        _logger.info("sample_id: %d", sample_id)
        save_dict = refine_clip(model, mano, clip, device)
        if cfg["commit"]:
            formats.write_refine_sample(cfg["ckpt_path"], cfg["debug"].get("sample_save_offset", ""), save_dict)
            n_written += 1
    _logger.info("refined %d clips, wrote %d", len(seen), n_written)
    return 0


if __name__ == "__main__":
    sys.exit(main())
