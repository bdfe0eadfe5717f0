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


def _det_rotations(tag: str, shape):
    """float32 rotation matrices (*shape, 3, 3) from hashed unit quaternions"""
    q = det.det_normal(tag, tuple(shape) + (4,)).astype(np.float64)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=-1)
    return R.reshape(tuple(shape) + (3, 3)).astype(np.float32)


def synthetic_cache_dict(n_segments: int = 5, max_len: int = 160, tag: str = "cache"):
    """A segment cache shaped like InteractionSegmentData.get_cache() (dataset/interaction_segment.py:454-466): ten parallel
    columns, arrays zero-padded past each segment's length (dataset/setment_slice.py:27-31), 1-3 objects per segment out of
    four ids (so segments share objects and the sorted-id order differs from insertion order), two segments sharing one info
    tuple (the reference's loaders skip such twins)."""
    obj_ids = ["O02@0042@00003", "C10001", "O02@0007@00001", "S11005"]
    texts = ["pour water from the bottle into the cup", "open the laptop lid", "pick up the knife and cut the apple",
             "screw the cap onto the bottle"]
    cols = {k: [] for k in ("info", "len", "pose", "tsl", "shape", "hand_side", "text", "obj_traj", "frame_id")}
    used = set()
    for i in range(n_segments):
        t = f"{tag}/{i}"
        L = [max_len, 16, 97, 131, 40][i % 5] if max_len >= 131 else max(1, max_len - (i % 3))
        side = "lh" if i % 3 == 1 else "rh"
        key = i - 1 if i == 3 else i  # segments 2 and 3: the same info
        pose = _det_rotations(t + "/pose", (max_len, 16))
        tsl = (det.det_normal(t + "/tsl", (max_len, 3)) * 0.2).astype(np.float32)
        shape = np.repeat(det.det_normal(t + "/shape", (1, 10)), max_len, axis=0).astype(np.float32)
        pose[L:], tsl[L:], shape[L:] = 0.0, 0.0, 0.0
        nobj = 1 + i % 3
        mine = [obj_ids[(i + k + 1) % 4] for k in range(nobj)]  # deliberately not sorted
        traj = {}
        for o in mine:
            T4 = np.zeros((max_len, 4, 4), np.float32)
            T4[:, :3, :3] = _det_rotations(f"{t}/obj/{o}/R", (max_len,))
            T4[:, :3, 3] = det.det_normal(f"{t}/obj/{o}/t", (max_len, 3)) * 0.3
            T4[:, 3, 3] = 1.0
            T4[L:] = 0.0
            traj[o] = T4
        used.update(mine)
        cols["info"].append((f"scene_0{key % 4}__A00{key}++seq__{key:04x}", f"{key:02d}_primitive", side))
        cols["len"].append(L)
        cols["pose"].append(pose)
        cols["tsl"].append(tsl)
        cols["shape"].append(shape)
        cols["hand_side"].append(side)
        cols["text"].append(texts[i % len(texts)])
        cols["obj_traj"].append(traj)
        cols["frame_id"].append(list(range(100 * i, 100 * i + 12 * L, 12)))
    cache = {f"interaction_segment_{k}_list": v for k, v in cols.items()}
    cache["interaction_object_list"] = sorted(used)
    return cache


def synthetic_object_embedding(obj_id: str, dim: int = 768):
    return det.det_normal(f"objemb/{obj_id}", (dim,)).astype(np.float32)


def synthetic_object_pointcloud(obj_id: str, n_points: int = 64):
    return (det.det_normal(f"objpc/{obj_id}", (n_points, 3)) * 0.05).astype(np.float32)


def synthetic_text_embedding(text: str, dim: int = 512):
    return det.det_normal(f"clip/{text}", (dim,)).astype(np.float32)


def write_synthetic_dataset(root: str, n_segments: int = 5, max_len: int = 160, tag: str = "cache"):
    """the files script/sample.sh / sample_refine.sh name, under `root`: the cache pickle, <obj_id>.pt embeddings,
    <obj_id>.npz point clouds, and the text-embedding table that stands in for the CLIP tower -> dict of paths"""
    import os
    import pickle

    import torch

    cache = synthetic_cache_dict(n_segments, max_len, tag)
    paths = {"cache": os.path.join(root, "common", "save_cache_dict", "main", "cache", "test.pkl"),
             "emb": os.path.join(root, "common", "retrieve_obj_embedding", "main", "embedding"),
             "pc": os.path.join(root, "common", "retrieve_obj_pointcloud", "main", "pointcloud"),
             "text": os.path.join(root, "common", "text_embedding", "test.pkl"),
             "split": os.path.join(root, "asset", "split", "test.txt")}
    for p in (os.path.dirname(paths["cache"]), paths["emb"], paths["pc"], os.path.dirname(paths["text"]), os.path.dirname(paths["split"])):
        os.makedirs(p, exist_ok=True)
    with open(paths["cache"], "wb") as f:
        pickle.dump(cache, f)
    for o in cache["interaction_object_list"]:
        torch.save(torch.from_numpy(synthetic_object_embedding(o)), os.path.join(paths["emb"], f"{o}.pt"))
        np.savez(os.path.join(paths["pc"], f"{o}.npz"), point=synthetic_object_pointcloud(o))
    with open(paths["text"], "wb") as f:
        pickle.dump({t: synthetic_text_embedding(t) for t in sorted(set(cache["interaction_segment_text_list"]))}, f)
    with open(paths["split"], "w") as f:
        f.write("\n".join(dict.fromkeys(i[0] for i in cache["interaction_segment_info_list"])) + "\n")
    return paths, cache


def synthetic_object_mesh(obj_id: str):
    """(verts (V, 3) float64, faces (F, 3) int64): a small scaled icosphere per object id"""
    v, f = icosphere(1)
    s = 0.03 + 0.01 * (sum(obj_id.encode()) % 5)
    return v * s, f


def synthetic_sample_pose_repr(dir_name: str, sample_id: int, T: int = 160):
    return det.det_normal(f"gsample/{dir_name}/{sample_id}", (T, 99)).astype(np.float32)


def guidance_fn(x, t, **kwargs):
    """the cond_fn of the guided fixture (capture_golden.capture_respaced) and of the tests that replay it: a pull towards a t-dependent
    target - it uses BOTH arguments, so a wrong timestep map shows"""
    return -0.5 * x + 0.2 * (t.float() / 1000.0).view(-1, 1, 1, 1)
