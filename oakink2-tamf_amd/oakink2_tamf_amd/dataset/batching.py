"""Batch assembly of interaction segments: the input contract of the hot path.

`interaction_segment_collate` reproduces the reference collate (oakink2_tamf/dataset/collate.py:6-58): per-clip dicts from
InteractionSegmentData.__getitem__ become one batch dict in which
  * array-like fields are stacked along a new leading clip axis (torch tensors),
  * per-clip Python objects (strings, lists, meshes, ids) stay lists of length B,
  * the per-object fields `obj_traj` (nobj, T, 9) and `obj_embedding` (nobj, 768) are zero-padded on the object axis to the
    largest object count of the batch before stacking - which is why the denoiser's object means run over PADDED objects
    (interaction_segment_mdm.py:233-263) and why `obj_num` travels with the batch.
A field outside the three groups is an error, as in the reference."""
from __future__ import annotations

from typing import Dict, Iterable, List

import numpy as np
import torch

STACKED_FIELDS = ("pose_repr", "pose_repr_lh", "pose_repr_rh", "shape", "shape_lh", "shape_rh", "len", "mask", "obj_num",
                  "sample_pose_repr")
LISTED_FIELDS = ("hand_side", "text", "obj_list", "info", "obj_verts", "obj_faces", "obj_pointcloud", "sample_info",
                 "frame_id")
OBJECT_PADDED_FIELDS = ("obj_traj", "obj_embedding")


def _stack(values: List) -> torch.Tensor:
    """torch.utils.data.default_collate semantics for the value kinds the dataset produces: numpy arrays / tensors are
    stacked keeping their dtype, Python ints -> int64, Python floats -> float64, numpy scalars keep their dtype."""
    first = values[0]
    if isinstance(first, torch.Tensor):
        return torch.stack(list(values), dim=0)
    if isinstance(first, np.ndarray):
        return torch.stack([torch.as_tensor(v) for v in values], dim=0)
    if isinstance(first, (bool, np.bool_)):
        return torch.tensor([bool(v) for v in values])
    if isinstance(first, (int, np.integer)):
        return torch.as_tensor(np.asarray(values)) if isinstance(first, np.integer) else torch.tensor(list(values), dtype=torch.int64)
    if isinstance(first, (float, np.floating)):
        return torch.as_tensor(np.asarray(values)) if isinstance(first, np.floating) else torch.tensor(list(values), dtype=torch.float64)
    raise TypeError(f"cannot stack values of type {type(first)}")


def pad_object_axis(items: Iterable[np.ndarray]) -> List[np.ndarray]:
    """zero-pad every array along axis 0 to the longest one"""
    items = [np.asarray(a) for a in items]
    n_max = max(a.shape[0] for a in items)
    out = []
    for a in items:
        if a.shape[0] < n_max:
            padded = np.zeros((n_max,) + a.shape[1:], dtype=a.dtype)
            padded[: a.shape[0]] = a
            a = padded
        out.append(a)
    return out


def interaction_segment_collate(batch: List[Dict]) -> Dict:
    if not batch:
        raise ValueError("empty batch")
    out: Dict = {}
    for field in batch[0].keys():
        if field not in STACKED_FIELDS and field not in LISTED_FIELDS and field not in OBJECT_PADDED_FIELDS:
            raise KeyError(f"unexpected key in batch! got {field}")
        column = [clip[field] for clip in batch]
        if field in STACKED_FIELDS:
            out[field] = _stack(column)
        elif field in LISTED_FIELDS:
            out[field] = column
        else:
            out[field] = _stack(pad_object_axis(column))
    return out
