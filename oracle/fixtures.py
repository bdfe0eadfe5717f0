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


def icosphere(subdiv: int = 3):
    """closed triangle mesh of the unit sphere (12 * 4^subdiv ... vertices shared), float64 vertices, int64 faces"""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1),
         (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(x, np.float64) / np.linalg.norm(x) for x in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = (v[a] + v[b]) / 2.0
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v, np.float64), np.asarray(f, np.int64)


def torus(nu: int = 48, nv: int = 24, R: float = 1.0, r: float = 0.35):
    """closed genus-1 mesh (a vertical ray can cross it four times)"""
    u = np.arange(nu) * (2 * np.pi / nu)
    w = np.arange(nv) * (2 * np.pi / nv)
    uu, ww = np.meshgrid(u, w, indexing="ij")
    v = np.stack([(R + r * np.cos(ww)) * np.cos(uu), (R + r * np.cos(ww)) * np.sin(uu), r * np.sin(ww)], axis=-1).reshape(-1, 3)
    f = []
    for i in range(nu):
        for j in range(nv):
            a, b = i * nv + j, ((i + 1) % nu) * nv + j
            c, d = ((i + 1) % nu) * nv + (j + 1) % nv, i * nv + (j + 1) % nv
            f += [(a, b, c), (a, c, d)]
    return v.astype(np.float64), np.asarray(f, np.int64)


def siv_cases():
    """(name, verts, faces, points): a hand-sized bumpy closed blob and a tilted torus, queried with points in and around
    them, points far outside the bounding box, and the mesh's own vertices (rays through vertices / edges)"""
    out = []
    v, f = icosphere(3)
    bump = 1.0 + 0.25 * np.sin(5.0 * v[:, 0]) * np.cos(4.0 * v[:, 1]) + 0.15 * np.sin(7.0 * v[:, 2])
    v = v * bump[:, None] * np.array([0.05, 0.09, 0.03]) + np.array([0.01, -0.02, 0.4])
    out.append(("blob", v, f))
    v, f = torus()
    c, s = np.cos(0.6), np.sin(0.6)
    v = (v @ np.array([[1, 0, 0], [0, c, -s], [0, s, c]]).T) * 0.06 + np.array([-0.1, 0.05, 0.2])
    out.append(("torus", v, f))
    cases = []
    for name, v, f in out:
        lo, hi = v.min(axis=0), v.max(axis=0)
        u = det.det_uniform(f"siv/{name}/u", (6000, 3))
        pts = lo - 0.15 * (hi - lo) + u * 1.3 * (hi - lo)
        pts = np.concatenate([pts, v[::7], (v[f[::11, 0]] + v[f[::11, 1]] + v[f[::11, 2]]) / 3.0, np.array([[10.0, 10.0, 10.0], lo, hi])])
        cases.append((name, v, f, pts.astype(np.float64)))
    return cases
