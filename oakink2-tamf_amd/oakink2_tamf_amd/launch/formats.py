"""On-disk formats either side of the two stages, so that the reference's downstream scripts
(script/sample_refine.sh, script/compute_score/*) read what this package writes and vice versa.

G stage  (launch/sample.py:234-237):          <ckpt_path>/sample/<offset>/<sample_id:06d>.npy   float32 (T, 99)
R stage  (launch/sample_refine.py:274-296):   <ckpt_path>/sample/<offset>/<process_key with '/' -> '++'>/<info[1]>/<info[2]>/save_dict.pkl
         a pickled dict with exactly the keys REFINE_KEYS.
<ckpt_path> is <cwd>/common/<prog>/<exp_id> (dev_fn/upkeep/ckpt.py:67-72)."""
from __future__ import annotations

import os
import pickle
from typing import Dict, Sequence

import numpy as np

REFINE_KEYS = ("process_key", "info", "hand_side", "joints", "verts", "faces", "obj_list", "len", "frame_id", "refine_pose_repr")


def ckpt_path(prog: str, exp_id: str, cwd: str | None = None) -> str:
    return os.path.join(cwd if cwd is not None else os.getcwd(), "common", prog, exp_id)


def sample_npy_path(ckpt: str, offset: str, sample_id: int) -> str:
    return os.path.join(ckpt, "sample", offset, f"{int(sample_id):06d}.npy")


def write_sample_npy(ckpt: str, offset: str, sample_id: int, pose_repr: np.ndarray) -> str:
    pose_repr = np.asarray(pose_repr, dtype=np.float32)
    if pose_repr.ndim != 2:
        raise ValueError(f"expected (T, 99) pose representation, got shape {pose_repr.shape}")
    path = sample_npy_path(ckpt, offset, sample_id)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.save(path, pose_repr)
    return path


def refine_sample_path(ckpt: str, offset: str, info: Sequence) -> str:
    return os.path.join(ckpt, "sample", offset, str(info[0]).replace("/", "++"), str(info[1]), str(info[2]), "save_dict.pkl")


def build_refine_save_dict(info: Sequence, hand_side: str, joints, verts, faces, obj_list, avai_len, frame_id,
                           refine_pose_repr) -> Dict:
    """info = (process_key, segment ids ...) as produced by the dataset; joints (T, 21, 3) and verts (T, 778, 3) are the MANO
    outputs already translated by the wrist translation (sample_refine.py:268-271); faces are the closed-hand faces of the
    respective side."""
    if hand_side not in ("rh", "lh"):
        raise ValueError(f"unexpected hand_side: {hand_side}")
    joints, verts = np.asarray(joints), np.asarray(verts)
    refine_pose_repr = np.asarray(refine_pose_repr)
    if joints.shape[0] != verts.shape[0] or verts.shape[0] != refine_pose_repr.shape[0]:
        raise ValueError("joints, verts and refine_pose_repr must cover the same frames")
    return {"process_key": info[0], "info": info, "hand_side": hand_side, "joints": joints, "verts": verts, "faces": faces,
            "obj_list": obj_list, "len": avai_len, "frame_id": frame_id, "refine_pose_repr": refine_pose_repr}


def write_refine_sample(ckpt: str, offset: str, save_dict: Dict) -> str:
    missing = [k for k in REFINE_KEYS if k not in save_dict]
    if missing or len(save_dict) != len(REFINE_KEYS):
        raise KeyError(f"save_dict must hold exactly {REFINE_KEYS}; missing {missing}, extra {sorted(set(save_dict) - set(REFINE_KEYS))}")
    path = refine_sample_path(ckpt, offset, save_dict["info"])
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(save_dict, f)
    return path


def read_refine_sample(path: str) -> Dict:
    with open(path, "rb") as f:
        d = pickle.load(f)
    missing = [k for k in REFINE_KEYS if k not in d]
    if missing:
        raise KeyError(f"{path}: not a refine save_dict, missing {missing}")
    return d
