"""Clip assembly from the reference's segment cache: the data format on the INPUT side of the hot path.

The reference's launchers never walk the OakInk2 recordings at sampling time: `launch/sample.py:146-160` and
`launch/sample_refine.py:155-167` unpickle a *cache dict* (written once by `InteractionSegmentData.get_cache`,
dataset/interaction_segment.py:454-466) and hand it to `InteractionSegmentData(cache_dict=...)`, whose `__getitem__`
(:389-449) turns segment i into the per-clip dict that `interaction_segment_collate` batches.  This module reads that
pickle and reproduces that item dict bit for bit (fixture: tests/golden/cache_dict_items.pkl, captured from the reference's
class by oracle/capture_golden.py:capture_cache_dict), so `script/sample.sh`'s literal command line runs here.

The cache dict holds ten parallel columns, one entry per segment (CACHE_KEYS):

    info      (process_key, primitive_identifier, hand_side)        len       valid frames (16..slice_max_len)
    pose      (slice_max_len, 16, 3, 3) float32 rotation matrices   tsl       (slice_max_len, 3) float32
    shape     (slice_max_len, 10) float32 MANO betas                hand_side "rh" | "lh"
    text      task description (the CLIP prompt)                    obj_traj  {obj_id: (slice_max_len, 4, 4) float32}
    frame_id  list of `len` mocap frame ids                         + interaction_object_list: sorted ids of all objects

all zero-padded past `len` (dataset/setment_slice.py:27-31).  An item is

    pose_repr (T, 99) = [tsl | 16 x rot6d]   rot6d = the first two ROWS of the rotation matrix, row-major
                                             (dev_fn/transform/rotation_np.py:502-518)
    obj_traj  (nobj, T, 9) = [tsl | rot6d] of each object's 4 x 4 pose, objects in sorted-id order
                                             (dev_fn/transform/transform_np.py:159-166)
    mask (slice_max_len,) 1 / 0, shape, len, info, hand_side, text, obj_list, obj_num, frame_id,
    obj_embedding (nobj, 768) from <obj_embedding_prefix>/<obj_id>.pt (:267-274),
    obj_pointcloud (nobj, P, 3) from <obj_pointcloud_prefix>/<obj_id>.npz["point"] (:276-283),
    obj_verts / obj_faces lists when object meshes are available (:424-429).

What cannot come from the pickle: walking the raw recordings (`load_dataset`, :60-160) and the object meshes of
`enable_obj_model=True` (:355-362) need the OakInk2 toolkit, which does not ship.  Without a cache dict the constructor
raises; object meshes are taken from `obj_model_loader(obj_id) -> (verts, faces)` when one is given and left out otherwise
(neither G nor the point-cloud R forward reads them: interaction_segment_mdm.py:145-162, segment_refine_model.py:183-186).
"""
from __future__ import annotations

import logging
import os
import pickle
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np

from .segment_slice import segment_slice_from_gap  # noqa: F401  (re-exported: the slicer that produced the cache's padding)

_logger = logging.getLogger(__name__)

COLUMNS = ("info", "len", "pose", "tsl", "shape", "hand_side", "text", "obj_traj", "frame_id")
CACHE_KEYS = tuple(f"interaction_segment_{c}_list" for c in COLUMNS) + ("interaction_object_list",)
FPS_MOCAP = 120.0  # oakink2_toolkit.meta.FPS_MOCAP (the toolkit is absent; the OakInk2 mocap rate, only used for target_gap)


def load_cache_dict(path: str) -> Dict:
    """unpickle a segment cache and check that it is one (launch/sample.py:158-159)"""
    with open(path, "rb") as f:
        cache = pickle.load(f)
    check_cache_dict(cache, path)
    return cache


def check_cache_dict(cache, where: str = "cache_dict") -> None:
    if not isinstance(cache, dict):
        raise TypeError(f"{where}: expected the dict written by InteractionSegmentData.get_cache, got {type(cache).__name__}")
    missing = [k for k in CACHE_KEYS if k not in cache]
    if missing:
        raise KeyError(f"{where}: not a segment cache, missing {missing}")
    n = len(cache[CACHE_KEYS[1]])
    ragged = [k for k in CACHE_KEYS[:-1] if len(cache[k]) != n]
    if ragged:
        raise ValueError(f"{where}: columns of different length: {ragged} (expected {n} segments)")


def rotmat_to_rot6d(rotmat: np.ndarray) -> np.ndarray:
    """(..., 3, 3) -> (..., 6): rows 0 and 1, row-major (rotation_np.py:502-518)"""
    rotmat = np.asarray(rotmat)
    return np.ascontiguousarray(rotmat[..., 0:2, :]).reshape(rotmat.shape[:-2] + (6,))


def transf_to_tslrot6d(transf: np.ndarray) -> np.ndarray:
    """(..., 4, 4) rigid pose -> (..., 9) = [translation | rot6d] (transform_np.py:159-166)"""
    transf = np.asarray(transf)
    return np.concatenate((transf[..., 0:3, 3], rotmat_to_rot6d(transf[..., 0:3, 0:3])), axis=-1)


def _reverse_valid_prefix(arr: np.ndarray, n: int) -> np.ndarray:
    out = np.array(arr, copy=True)
    out[:n] = np.asarray(arr)[:n][::-1]
    return out


class InteractionSegmentData:
    """Drop-in for the reference's dataset class on its cache-dict path (same constructor keywords, same item dicts).
    A plain sequence (`len()`, integer indexing); usable as a torch Dataset without inheriting from it."""

    def __init__(self, process_range_list: Optional[Sequence[str]] = None, data_prefix: Optional[str] = None,
                 target_fps: float = 10.0, slice_min_len: int = 16, slice_max_len: int = 160, rank: Optional[int] = None,
                 enable_obj_model: bool = False, obj_embedding_prefix: Optional[str] = None,
                 obj_pointcloud_prefix: Optional[str] = None, cache_dict: Optional[Dict] = None,
                 append_reverse_segment: bool = False,
                 obj_model_loader: Optional[Callable[[str], tuple]] = None):
        # with a cache dict the reference ignores process_range_list and data_prefix as well (:312-324); kept as attributes
        self.process_range_list = list(process_range_list) if process_range_list is not None else None
        self.data_prefix = data_prefix
        self.origin_fps, self.target_fps = FPS_MOCAP, target_fps
        self.target_gap = int(self.origin_fps // self.target_fps)
        self.slice_min_len, self.slice_max_len = slice_min_len, slice_max_len
        if cache_dict is None:
            raise NotImplementedError(
                "InteractionSegmentData without cache_dict walks the OakInk2 recordings through oakink2_toolkit "
                "(dataset/interaction_segment.py:60-160), which is not part of this build: pass the cache dict that "
                "the reference's save_cache_dict step wrote (--data.cache_dict_filepath)")
        check_cache_dict(cache_dict)
        for key in CACHE_KEYS:
            setattr(self, key, cache_dict[key])
        self.append_reverse_segment = append_reverse_segment
        if append_reverse_segment:
            self._append_reversed()
            _logger.info("load reverse segment")
        self.len = len(self.interaction_segment_len_list)
        if not rank:
            _logger.info("collect %d segments", self.len)

        self.enable_obj_model = bool(enable_obj_model and obj_model_loader is not None)
        if enable_obj_model and obj_model_loader is None:
            _logger.info("enable_obj_model: no obj_model_loader (the OakInk2 toolkit is absent) - items carry no obj_verts / obj_faces")
        self.obj_store = {o: obj_model_loader(o) for o in self.interaction_object_list} if self.enable_obj_model else None

        self.enable_obj_embedding = obj_embedding_prefix is not None
        self.obj_embedding_prefix = obj_embedding_prefix
        self.obj_embedding_store = self.load_object_embedding() if self.enable_obj_embedding else None
        self.enable_obj_pointcloud = obj_pointcloud_prefix is not None
        self.obj_pointcloud_prefix = obj_pointcloud_prefix
        self.obj_pointcloud_store = self.load_object_pointcloud() if self.enable_obj_pointcloud else None

    # ---- per-object side files (:267-283) -------------------------------------------------------------------------------
    def load_object_embedding(self) -> Dict[str, np.ndarray]:
        import torch

        store = {}
        for obj_id in self.interaction_object_list:
            emb = torch.load(os.path.join(self.obj_embedding_prefix, f"{obj_id}.pt"), map_location="cpu")
            store[obj_id] = np.array(emb.numpy(), dtype=np.float32)
        return store

    def load_object_pointcloud(self) -> Dict[str, np.ndarray]:
        store = {}
        for obj_id in self.interaction_object_list:
            with np.load(os.path.join(self.obj_pointcloud_prefix, f"{obj_id}.npz")) as z:
                store[obj_id] = np.array(z["point"], dtype=np.float32)
        return store

    # ---- time-reversed twins (:162-265) ---------------------------------------------------------------------------------
    def _append_reversed(self) -> None:
        """every segment once more with its valid frames in reverse order (same info: launch/sample_refine.py:217-222 skips
        the twins by that)"""
        n = len(self.interaction_segment_len_list)
        col = {c: list(getattr(self, f"interaction_segment_{c}_list")) for c in COLUMNS}
        for i in range(n):
            L = col["len"][i]
            for c in ("info", "len", "hand_side", "text"):
                col[c].append(col[c][i])
            for c in ("pose", "tsl", "shape"):
                col[c].append(_reverse_valid_prefix(col[c][i], L))
            col["obj_traj"].append({o: _reverse_valid_prefix(t, L) for o, t in col["obj_traj"][i].items()})
            col["frame_id"].append(col["frame_id"][i][::-1])
        for c in COLUMNS:
            setattr(self, f"interaction_segment_{c}_list", col[c])

    # ---- one clip (:389-449) --------------------------------------------------------------------------------------------
    def __getitem__(self, index: int) -> Dict:
        n_valid = self.interaction_segment_len_list[index]
        pose = np.asarray(self.interaction_segment_pose_list[index])  # (T, 16, 3, 3)
        rot6d = rotmat_to_rot6d(pose)
        pose_repr = np.concatenate((self.interaction_segment_tsl_list[index], rot6d.reshape(rot6d.shape[0], 16 * 6)), axis=-1)
        traj = self.interaction_segment_obj_traj_list[index]
        obj_list = sorted(traj.keys())
        mask = np.zeros((self.slice_max_len,), dtype=np.float32)
        mask[:n_valid] = 1.0
        item = {
            "info": self.interaction_segment_info_list[index],
            "len": n_valid,
            "mask": mask,
            "pose_repr": pose_repr,
            "shape": self.interaction_segment_shape_list[index],
            "hand_side": self.interaction_segment_hand_side_list[index],
            "text": self.interaction_segment_text_list[index],
            "obj_list": obj_list,
            "obj_num": len(obj_list),
            "obj_traj": np.stack([transf_to_tslrot6d(traj[o]) for o in obj_list], axis=0),
            "frame_id": self.interaction_segment_frame_id_list[index],
        }
        if self.enable_obj_model:
            item["obj_verts"] = [np.array(self.obj_store[o][0]) for o in obj_list]
            item["obj_faces"] = [np.array(self.obj_store[o][1]) for o in obj_list]
        if self.enable_obj_embedding:
            item["obj_embedding"] = np.stack([self.obj_embedding_store[o] for o in obj_list], axis=0)
        if self.enable_obj_pointcloud:
            item["obj_pointcloud"] = np.stack([self.obj_pointcloud_store[o] for o in obj_list], axis=0)
        return item

    def __len__(self) -> int:
        return self.len

    def __iter__(self) -> Iterable[Dict]:
        return (self[i] for i in range(self.len))

    def get_cache(self) -> Dict:
        """the dict `load_cache_dict` reads (:454-466)"""
        return {k: getattr(self, k) for k in CACHE_KEYS}

    def texts(self) -> List[str]:
        return list(self.interaction_segment_text_list)
