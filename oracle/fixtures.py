"""TEST INFRASTRUCTURE ONLY: deterministic inputs shared by oracle/capture_golden.py and tests/."""
import numpy as np

from . import det


def ragged_clips():
    """three per-clip dicts shaped like InteractionSegmentData.__getitem__ output (dataset/interaction_segment.py:389-449)
    with 1, 3 and 2 objects"""
    clips = []
    for i, nobj in enumerate([1, 3, 2]):
        T = 8
        tag = f"collate/{i}"
        clips.append({
            "pose_repr": det.det_normal(tag + "/pose", (T, 99)).astype(np.float32),
            "shape": det.det_normal(tag + "/shape", (T, 10)).astype(np.float32),
            "len": 5 + i,
            "mask": (np.arange(T) < 5 + i),
            "obj_num": nobj,
            "hand_side": "rh" if i % 2 == 0 else "lh",
            "text": f"clip {i}",
            "obj_list": [f"O{i}_{k}" for k in range(nobj)],
            "info": (f"scene/seq{i}", i, "rh" if i % 2 == 0 else "lh"),
            "frame_id": list(range(10 * i, 10 * i + T)),
            "obj_traj": det.det_normal(tag + "/traj", (nobj, T, 9)).astype(np.float32),
            "obj_embedding": det.det_normal(tag + "/emb", (nobj, 768)).astype(np.float32),
        })
    return clips
