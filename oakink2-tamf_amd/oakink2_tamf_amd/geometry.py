"""HIP versions of the geometry steps around the trunks.

pose_repr_to_quat      reference: rot6d_to_rotmat + rotmat_to_quat (dev_fn/transform/rotation.py:446-467,167-213) as used by
                       launch/sample_refine.py:254-260 and model/segment_refine_model.py:117-124 before the MANO layer
multi_object_h2o_dist  reference: SegmentRefineModel.multi_object_h2o_dist (model/segment_refine_model.py:142-168) ->
                       point2point_signed (model/loss/chamfer_distance.py:4-64) -> external chamfer_distance CUDA extension
contact_min_dist       reference: transf_merge_obj_pointcloud + contact_min_cdist (script/compute_score/compute_score_cr.py:122-149),
contact_ratio          the Contact-Ratio score built on it (:282-283, threshold 5 mm)
transform_points       reference: tslrot6d_to_transf_np + transf_point_array_np (dev_fn/transform/transform_np.py:169-175,36-53)
vertex_normals         reference: Meshes(verts, faces).verts_normals_packed() of the MANO hand (model/segment_refine_model.py:131-133;
                       pytorch3d 0.7.2 _compute_vertex_normals)
mesh_contains          reference: check_mesh_contains (dev_fn/external/libmesh/inside_mesh.py:8-149 + Cython TriangleHash),
solid_intersection_volume  the SIV score built on it (script/compute_score/compute_score_siv.py:128-153)
All return torch tensors on the inputs' device; no CPU fallback."""
from __future__ import annotations

from ctypes import c_void_p
from typing import Optional, Sequence

import torch

from .hip_backend import _check, _dev_f32, _stream_ptr, lib, require_gpu


def _bind():
    from ctypes import c_int32, c_int64

    L = lib()
    L.tamf_pose_decode.argtypes = [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]
    L.tamf_h2o_dist.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p, c_void_p]
    L.tamf_contact_min_dist.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p, c_void_p]
    L.tamf_transform_points.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]
    L.tamf_vertex_normals.argtypes = [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]
    L.tamf_mesh_contains.argtypes = [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_void_p,
                                     c_void_p, c_void_p]
    return L


def pose_repr_to_quat(pose_repr: torch.Tensor):
    """(..., 3 + 6J) -> tsl (..., 3), quat (..., J, 4) (w, x, y, z), w >= 0."""
    dev = require_gpu(pose_repr.device)
    p = _dev_f32(pose_repr, dev)
    F = p.shape[-1]
    J = (F - 3) // 6
    assert 3 + 6 * J == F, "pose representation must be 3 + 6*J wide"
    n = p.numel() // F
    tsl = torch.empty(p.shape[:-1] + (3,), device=dev, dtype=torch.float32)
    quat = torch.empty(p.shape[:-1] + (J, 4), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _check(_bind().tamf_pose_decode(c_void_p(p.data_ptr()), n, J, c_void_p(tsl.data_ptr()), c_void_p(quat.data_ptr()),
                                        c_void_p(_stream_ptr(dev))))
    return tsl, quat


def _h2o_call(entry: str, hand_verts, obj_traj, obj_points, obj_num, per_vertex: bool) -> torch.Tensor:
    dev = require_gpu(hand_verts.device)
    hv, tr, pts = _dev_f32(hand_verts, dev), _dev_f32(obj_traj, dev), _dev_f32(obj_points, dev)
    B, T, V, _ = hv.shape
    nobj, P = pts.shape[1], pts.shape[2]
    assert tuple(tr.shape) == (B, nobj, T, 9) and pts.shape[0] == B and pts.shape[3] == 3
    on = None
    if obj_num is not None:
        on = torch.as_tensor(list(obj_num), dtype=torch.int32, device=dev)
        assert on.numel() == B
    out = torch.empty((B, T, V) if per_vertex else (B, T), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _check(getattr(_bind(), entry)(c_void_p(hv.data_ptr()), c_void_p(tr.data_ptr()), c_void_p(pts.data_ptr()),
                                       c_void_p(on.data_ptr() if on is not None else 0), B, T, V, nobj, P,
                                       c_void_p(out.data_ptr()), c_void_p(_stream_ptr(dev))))
    return out


def multi_object_h2o_dist(hand_verts: torch.Tensor, obj_traj: torch.Tensor, obj_points: torch.Tensor,
                          obj_num: Optional[Sequence[int]] = None) -> torch.Tensor:
    """hand_verts (B,T,V,3), obj_traj (B,nobj,T,9), obj_points (B,nobj,P,3), obj_num per clip -> (B,T,V)."""
    return _h2o_call("tamf_h2o_dist", hand_verts, obj_traj, obj_points, obj_num, True)


def contact_min_dist(hand_verts: torch.Tensor, obj_traj: torch.Tensor, obj_points: torch.Tensor,
                     obj_num: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Per-frame hand-object contact distance, (B,T): min over hand vertices and (transformed) object points."""
    return _h2o_call("tamf_contact_min_dist", hand_verts, obj_traj, obj_points, obj_num, False)


def contact_ratio(min_dist: torch.Tensor, valid_len: Optional[Sequence[int]] = None, threshold: float = 0.005) -> float:
    """Fraction of frames in contact (compute_score_cr.py:282-283); valid_len[b] frames of clip b count (`avai_len`)."""
    d = min_dist
    if valid_len is not None:
        keep = torch.arange(d.shape[1], device=d.device)[None, :] < torch.as_tensor(list(valid_len), device=d.device)[:, None]
        d = d[keep]
    return float((d < threshold).double().mean())


_CSR_CACHE = {}


def vertex_incidence_csr(faces, n_verts: int):
    """faces (F,3) int -> (off (V+1,), ent (3F,2)) int32 numpy: for every vertex the (next, prev) vertex pairs of its face
    corners, in the order index_add_ accumulates them on the CPU: corner 1 of all faces, then corner 2, then corner 0."""
    import numpy as np

    f = np.asarray(faces.detach().cpu().numpy() if isinstance(faces, torch.Tensor) else faces).astype(np.int64)
    F = f.shape[0]
    owner = np.concatenate([f[:, 1], f[:, 2], f[:, 0]])
    nxt = np.concatenate([f[:, 2], f[:, 0], f[:, 1]])
    prv = np.concatenate([f[:, 0], f[:, 1], f[:, 2]])
    order = np.argsort(owner, kind="stable")  # stable: keeps the pass / face order inside a vertex
    off = np.zeros(n_verts + 1, np.int32)
    np.cumsum(np.bincount(owner, minlength=n_verts), out=off[1:])
    ent = np.stack([nxt[order], prv[order]], axis=1).astype(np.int32)
    assert ent.shape[0] == 3 * F
    return off, np.ascontiguousarray(ent)


def vertex_normals(verts: torch.Tensor, faces) -> torch.Tensor:
    """verts (..., V, 3), faces (F, 3) -> unit vertex normals (..., V, 3): area-weighted face normals summed per vertex
    (pytorch3d's verts_normals_packed for every mesh of the sequence)."""
    dev = require_gpu(verts.device)
    v = _dev_f32(verts, dev)
    V = int(v.shape[-2])
    is_t = isinstance(faces, torch.Tensor)
    f_t = faces if is_t else torch.as_tensor(faces)
    # one CSR per topology, keyed without touching the device (storage address, shape, in-place version counter): a batch that
    # alternates rh / lh hands (two MANO face lists) keeps both; the keyed tensor is kept alive with its entry, so a recycled
    # address cannot be mistaken for it.  A handful of entries at most (LRU of 8).
    # A numpy array / list has no version counter (torch.as_tensor gives a fresh tensor with _version 0 every call, and an in-place
    # edit of the array would go unseen): such faces are keyed by CONTENT.
    if is_t:
        key = (f_t.data_ptr(), tuple(f_t.shape), f_t._version, str(f_t.device), str(f_t.dtype), V, str(dev))
    else:
        import hashlib

        fc = f_t.contiguous()
        key = ("content", hashlib.sha1(fc.numpy().tobytes()).hexdigest(), tuple(fc.shape), str(fc.dtype), V, str(dev))
    hit = _CSR_CACHE.get(key)
    if hit is None:
        off, ent = vertex_incidence_csr(f_t, V)
        hit = (torch.from_numpy(off).to(dev), torch.from_numpy(ent).to(dev), f_t)  # (f_t kept alive with its key)
        while len(_CSR_CACHE) >= 8:
            _CSR_CACHE.pop(next(iter(_CSR_CACHE)))
    else:
        _CSR_CACHE.pop(key)  # re-inserted below: most recently used last
    _CSR_CACHE[key] = hit
    n = v.numel() // (V * 3)
    out = torch.empty_like(v)
    with torch.cuda.device(dev):
        _check(_bind().tamf_vertex_normals(c_void_p(v.data_ptr()), n, V, c_void_p(hit[0].data_ptr()), c_void_p(hit[1].data_ptr()),
                                           c_void_p(out.data_ptr()), c_void_p(_stream_ptr(dev))))
    return out


def mesh_contains(verts, faces, points: torch.Tensor, resolution: int = 512) -> torch.Tensor:
    """verts (V,3), faces (F,3) of a closed triangle mesh, points (N,3) -> bool (N,): point inside the mesh, with the
    reference's float64 arithmetic (bit-identical booleans).  The mesh may live on the host (numpy) or the device; the
    bounding-box rescaling of inside_mesh.py:21-26 is computed on the host in float64 exactly as the reference does."""
    import numpy as np

    dev = require_gpu(points.device)
    v_np = verts.detach().cpu().numpy() if isinstance(verts, torch.Tensor) else np.asarray(verts)
    f_np = faces.detach().cpu().numpy() if isinstance(faces, torch.Tensor) else np.asarray(faces)
    v_np = v_np.astype(np.float64)
    tri = v_np[f_np].reshape(-1, 3)
    bmin, bmax = tri.min(axis=0), tri.max(axis=0)
    scale = np.ascontiguousarray((resolution - 1) / (bmax - bmin), dtype=np.float64)
    translate = np.ascontiguousarray(0.5 - scale * bmin, dtype=np.float64)
    v = torch.from_numpy(np.ascontiguousarray(v_np)).to(dev)
    f = torch.from_numpy(np.ascontiguousarray(f_np.astype(np.int32))).to(dev)
    p = points.to(device=dev, dtype=torch.float64).contiguous()
    n = p.shape[0]
    out = torch.empty(n, device=dev, dtype=torch.uint8)
    if n == 0:
        return out.bool()
    ws = torch.empty(f.shape[0] * 16, device=dev, dtype=torch.float64)
    with torch.cuda.device(dev):
        _check(_bind().tamf_mesh_contains(c_void_p(v.data_ptr()), c_void_p(f.data_ptr()), int(f.shape[0]), c_void_p(p.data_ptr()), n,
                                          scale.ctypes.data_as(c_void_p), translate.ctypes.data_as(c_void_p), int(resolution),
                                          c_void_p(ws.data_ptr()), c_void_p(out.data_ptr()), c_void_p(_stream_ptr(dev))))
    return out.bool()


def solid_intersection_volume(hand_verts, hand_faces, obj_points_list, el_vols) -> float:
    """SIV of one frame in cm^3: per object the interior voxel centres (already in the frame's pose) inside the hand mesh
    times the voxel volume (compute_score_siv.py:128-153)."""
    siv = 0.0
    for pts, el_vol in zip(obj_points_list, el_vols):
        siv += float(mesh_contains(hand_verts, hand_faces, pts).sum().item()) * float(el_vol) * (10 ** 6)
    return siv


def transform_points(obj_traj: torch.Tensor, obj_points: torch.Tensor) -> torch.Tensor:
    """obj_traj (..., T, 9) = [tsl | rot6d], obj_points (..., P, 3) in the object frame -> (..., T, P, 3) in the frame's pose;
    float32 or float64 (taken from obj_traj)."""
    dev = require_gpu(obj_traj.device)
    dt = torch.float64 if obj_traj.dtype == torch.float64 else torch.float32
    tr = obj_traj.to(device=dev, dtype=dt).contiguous()
    pts = obj_points.to(device=dev, dtype=dt).contiguous()
    lead = tr.shape[:-2]
    assert pts.shape[:-2] == lead and tr.shape[-1] == 9 and pts.shape[-1] == 3
    T, P = tr.shape[-2], pts.shape[-2]
    n = 1
    for v in lead:
        n *= int(v)
    out = torch.empty(tuple(lead) + (T, P, 3), device=dev, dtype=dt)
    with torch.cuda.device(dev):
        _check(_bind().tamf_transform_points(c_void_p(tr.data_ptr()), c_void_p(pts.data_ptr()), n, T, P, int(dt == torch.float64),
                                             c_void_p(out.data_ptr()), c_void_p(_stream_ptr(dev))))
    return out
