"""Strided sub-sampling of a mocap segment to the target frame rate (reference dataset/setment_slice.py:10-36).

A segment of n frames at 120 fps becomes `gap` interleaved clips traj[o::gap], o = 0..gap-1, each between min_len and max_len
frames and zero-padded to max_len - the layout every column of the segment cache carries (dataset/interaction_segment.py).
gap is the nominal one (origin_fps // target_fps) unless that would give clips shorter than min_len (then n // min_len) or
longer than max_len (then ceil(n / max_len))."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def slice_gap(n_frames: int, gap: int, max_len: int, min_len: int) -> int:
    if n_frames < min_len * gap:
        return n_frames // min_len
    if n_frames > max_len * gap:
        return -(-n_frames // max_len)
    return gap


def segment_slice_from_gap(traj: np.ndarray, gap: int, max_len: int, min_len: int) -> Tuple[List[np.ndarray], List[int]]:
    """-> (clips zero-padded to max_len along axis 0, their valid lengths)"""
    traj = np.asarray(traj)
    g = slice_gap(int(traj.shape[0]), gap, max_len, min_len)
    clips, lens = [], []
    for offset in range(g):
        part = traj[offset::g]
        n = int(part.shape[0])
        if not min_len <= n <= max_len:
            raise AssertionError(f"slice of {n} frames outside [{min_len}, {max_len}]")
        padded = np.zeros((max_len,) + part.shape[1:], dtype=part.dtype)
        padded[:n] = part
        clips.append(padded)
        lens.append(n)
    return clips, lens


class SegmentSlice:
    from_gap = staticmethod(segment_slice_from_gap)
