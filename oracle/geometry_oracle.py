"""CPU restatement of the geometry steps either side of the trunks.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations relative to /root/reference/src/.

  pose_decode          99-d pose repr -> translation + 16 unit quaternions
                       (oakink2_tamf/launch/sample_refine.py:254-260, model/segment_refine_model.py:117-124;
                        dev_fn/transform/rotation.py:446-467 rot6d_to_rotmat, :167-213 rotmat_to_quat, :24-35, :156-164)
  vertex_normals       hand-mesh vertex normals (model/segment_refine_model.py:131-133 -> pytorch3d 0.7.2, absent: restated
                       from its published algorithm, pinned on analytic cases)
  h2o_dist             hand-vertex -> nearest object point distance feature of R
                       (oakink2_tamf/model/segment_refine_model.py:142-168 -> model/loss/chamfer_distance.py:4-64 with
                        y_normals=None: x2y_signed = || x - y[nearest] ||; dev_fn/transform/transform.py:148-154,36-52)

Third-party piece: the nearest-neighbour search itself is `chamfer_distance.ChamferDistance` (otaheri/chamfer_distance,
un-pinned submodule, source absent): its published algorithm is a brute-force argmin of squared euclidean distances in both
directions; the golden vectors were captured from the reference's own point2point_signed with a stand-in module implementing
exactly that (oracle/capture_golden.py:capture_geometry), so the in-tree arithmetic around it is pinned.
"""
from __future__ import annotations

import numpy as np
import torch


def rot6d_to_rotmat(d6: torch.Tensor) -> torch.Tensor:
    """Gram-Schmidt of the two 3-vectors, rows of the result = b1, b2, b1 x b2 (rotation.py:446-467).
    F.normalize semantics: v / max(||v||, 1e-12)."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = a1 / a1.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = b2 / b2.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def rotmat_to_quat(m: torch.Tensor) -> torch.Tensor:
    """(w, x, y, z) with w >= 0; the best-conditioned of the four candidate formulas (rotation.py:167-213)."""
    m00, m01, m02 = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    m10, m11, m12 = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    m20, m21, m22 = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    sq = torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1)
    q_abs = torch.where(sq > 0, torch.sqrt(sq.clamp_min(0)), torch.zeros_like(sq))
    cand = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    cand = cand / (2.0 * q_abs[..., None].clamp_min(0.1))
    best = q_abs.argmax(dim=-1)
    out = torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)
    return torch.where(out[..., 0:1] < 0, -out, out)


def pose_decode(pose_repr: torch.Tensor):
    """(..., 99) -> tsl (..., 3), quat (..., 16, 4)."""
    tsl = pose_repr[..., 0:3]
    rot6d = pose_repr[..., 3:99].reshape(pose_repr.shape[:-1] + (16, 6))
    return tsl, rotmat_to_quat(rot6d_to_rotmat(rot6d))


def h2o_dist(hand_verts: torch.Tensor, obj_traj: torch.Tensor, obj_points: torch.Tensor, obj_num=None) -> torch.Tensor:
    """hand_verts (B, T, V, 3); obj_traj (B, nobj, T, 9) = [tsl | rot6d]; obj_points (B, nobj, P, 3) in object frame;
    obj_num[b] objects of clip b are real.  -> (B, T, V): distance of every hand vertex to its nearest object point
    after the per-frame rigid transform p -> R p + t."""
    B, T, V, _ = hand_verts.shape
    nobj = obj_traj.shape[1]
    out = torch.empty(B, T, V, dtype=hand_verts.dtype)
    for b in range(B):
        n = nobj if obj_num is None else int(obj_num[b])
        R = rot6d_to_rotmat(obj_traj[b, :n, :, 3:9])  # (n, T, 3, 3)
        t = obj_traj[b, :n, :, 0:3]  # (n, T, 3)
        pts = torch.einsum("otij,opj->otpi", R, obj_points[b, :n]) + t[:, :, None, :]  # (n, T, P, 3)
        pts = pts.permute(1, 0, 2, 3).reshape(T, -1, 3)
        d2 = ((hand_verts[b][:, :, None, :] - pts[:, None, :, :]) ** 2).sum(-1)  # (T, V, n*P)
        idx = d2.argmin(dim=-1)
        near = torch.gather(pts, 1, idx[..., None].expand(T, V, 3))
        out[b] = (hand_verts[b] - near).norm(dim=-1)
    return out


def contact_min_dist(hand_verts: torch.Tensor, obj_traj: torch.Tensor, obj_points: torch.Tensor, obj_num=None) -> torch.Tensor:
    """Per-frame contact distance of the Contact-Ratio score (script/compute_score/compute_score_cr.py:122-149):
    object points of all (real) objects are moved to the frame's pose and merged, then the smallest hand-vertex to
    object-point distance of the frame is taken.  -> (B, T)"""
    return h2o_dist(hand_verts, obj_traj, obj_points, obj_num).min(dim=-1).values


def contact_ratio(min_dist: torch.Tensor, threshold: float = 0.005) -> float:
    """compute_score_cr.py:282-283: share of frames whose contact distance is below 5 mm."""
    return float((min_dist < threshold).double().mean())


def mesh_contains(verts, faces, points, resolution: int = 512):
    """Point-in-closed-mesh test of the SIV score (dev_fn/external/libmesh/inside_mesh.py:8-149, used by
    script/compute_score/compute_score_siv.py:128-153), float64 numpy, brute force over all triangles.

    The reference rescales mesh and points to [0.5, resolution - 0.5]^3 (:21-26,140-142), finds for every point the triangles
    whose xy projection STRICTLY contains it (:144-149 TriangleIntersector2d.check_triangles; the Cython TriangleHash in front of
    it is only an acceleration structure: a triangle that strictly contains the point covers the point's grid cell, but the hash
    drops points whose cell index reaches `resolution`, triangle_hash.pyx:63-67), compares the intersection depth of the
    vertical ray with the point's z, both scaled by |n_z| (:82-110), and calls the point inside when the numbers of
    intersections above-or-at and below are both odd (:60-77).  Returns a bool array."""
    import numpy as np

    verts = np.asarray(verts, dtype=np.float64)
    faces = np.asarray(faces)
    pts = np.asarray(points, dtype=np.float64)
    tri = verts[faces]  # (F, 3, 3)
    flat = tri.reshape(-1, 3)
    bmin, bmax = flat.min(axis=0), flat.max(axis=0)
    scale = (resolution - 1) / (bmax - bmin)
    translate = 0.5 - scale * bmin
    tri = scale * tri + translate
    p = scale * pts + translate
    contains = np.zeros(len(p), dtype=bool)
    ok = np.all((0 <= p) & (p <= resolution), axis=1)
    ok &= (p[:, 0].astype(np.int64) < resolution) & (p[:, 1].astype(np.int64) < resolution)  # hash cell range
    idx = np.nonzero(ok)[0]
    t1, t2, t3 = tri[:, 0], tri[:, 1], tri[:, 2]
    # 2D containment terms per triangle (check_triangles): A = [t1 - t3, t2 - t3]^T in xy
    a00, a01 = t1[:, 0] - t3[:, 0], t2[:, 0] - t3[:, 0]
    a10, a11 = t1[:, 1] - t3[:, 1], t2[:, 1] - t3[:, 1]
    det = a00 * a11 - a01 * a10
    sdet, adet = np.sign(det), np.abs(det)
    # depth terms (compute_intersection_depth)
    v1, v2 = t3 - t1, t2 - t1
    nrm = np.cross(v1, v2)
    n2 = nrm[:, 2]
    sn2, an2 = np.sign(n2), np.abs(n2)
    for lo in range(0, len(idx), 4096):
        ii = idx[lo:lo + 4096]
        q = p[ii]
        y0 = q[:, None, 0] - t3[None, :, 0]
        y1 = q[:, None, 1] - t3[None, :, 1]
        u = (a11[None] * y0 - a01[None] * y1) * sdet[None]
        v = (-a10[None] * y0 + a00[None] * y1) * sdet[None]
        s = u + v
        hit = (adet[None] != 0.0) & (0 < u) & (u < adet[None]) & (0 < v) & (v < adet[None]) & (0 < s) & (s < adet[None])
        alpha = nrm[None, :, 0] * (t1[None, :, 0] - q[:, None, 0]) + nrm[None, :, 1] * (t1[None, :, 1] - q[:, None, 1])
        depth = t1[None, :, 2] * an2[None] + alpha * sn2[None]
        zq = q[:, None, 2] * an2[None]
        live = hit & (an2[None] != 0)  # depth is NaN for triangles with n_z == 0 (:97-100): neither comparison holds
        above = live & (depth >= zq)
        below = live & (depth < zq)
        contains[ii] = (above.sum(axis=1) % 2 == 1) & (below.sum(axis=1) % 2 == 1)
    return contains


def solid_intersection_volume(hand_verts, hand_faces, obj_points_list, el_vols) -> float:
    """compute_score_siv.py:128-153: per object, voxel centres of the object's interior (already moved to the frame's pose) that
    fall inside the hand mesh, times the voxel volume, in cm^3 (x 1e6)."""
    siv = 0.0
    for pts, el_vol in zip(obj_points_list, el_vols):
        siv += float(mesh_contains(hand_verts, hand_faces, pts).sum()) * float(el_vol) * (10 ** 6)
    return siv


def transform_points(obj_traj: torch.Tensor, obj_points: torch.Tensor) -> torch.Tensor:
    """tslrot6d_to_transf_np + transf_point_array_np (dev_fn/transform/transform_np.py:169-175,36-53): obj_traj (..., T, 9),
    obj_points (..., P, 3) -> (..., T, P, 3) = R p + t per frame."""
    R = rot6d_to_rotmat(obj_traj[..., 3:9])  # (..., T, 3, 3), rows b1, b2, b3
    return torch.einsum("...tij,...pj->...tpi", R, obj_points) + obj_traj[..., None, 0:3]


def vertex_normals(verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Vertex normals as the reference obtains them (model/segment_refine_model.py:131-133):
    pytorch3d.structures.Meshes(verts, faces).verts_normals_packed().  pytorch3d (pinned 0.7.2, requirements.dist.txt:331) is
    absent here; this restates its published algorithm (structures/meshes.py, Meshes._compute_vertex_normals):

        vertices_faces = verts[faces]                                    # (F, 3, 3)
        n.index_add(0, faces[:, 1], cross(vf[:, 2] - vf[:, 1], vf[:, 0] - vf[:, 1]))
        n.index_add(0, faces[:, 2], cross(vf[:, 0] - vf[:, 2], vf[:, 1] - vf[:, 2]))
        n.index_add(0, faces[:, 0], cross(vf[:, 1] - vf[:, 0], vf[:, 2] - vf[:, 0]))
        n = torch.nn.functional.normalize(n, eps=1e-6, dim=1)            # x / max(|x|, eps)

    verts (..., V, 3) float32, faces (F, 3) -> (..., V, 3) float32, accumulated sequentially in float32 (np.add.at = the CPU
    index_add_ order).  Parity of this row is pinned on the definition (analytic cases in tests), not on pytorch3d itself."""
    v = np.asarray(verts, dtype=np.float32)
    f = np.asarray(faces, dtype=np.int64)
    lead = v.shape[:-2]
    vv = v.reshape((-1,) + v.shape[-2:])
    out = np.zeros_like(vv)
    for m in range(vv.shape[0]):
        x = vv[m]
        vf = x[f]
        n = np.zeros_like(x)
        np.add.at(n, f[:, 1], np.cross(vf[:, 2] - vf[:, 1], vf[:, 0] - vf[:, 1]).astype(np.float32))
        np.add.at(n, f[:, 2], np.cross(vf[:, 0] - vf[:, 2], vf[:, 1] - vf[:, 2]).astype(np.float32))
        np.add.at(n, f[:, 0], np.cross(vf[:, 1] - vf[:, 0], vf[:, 2] - vf[:, 0]).astype(np.float32))
        ln = np.sqrt((n.astype(np.float32) ** 2).sum(axis=1, dtype=np.float32))
        out[m] = n / np.maximum(ln, np.float32(1e-6))[:, None]
    return out.reshape(lead + v.shape[-2:])
