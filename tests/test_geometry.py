"""Rows 8(f)-1/-2/-4: pose decode, hand->object distance, Contact-Ratio frame distance, SIV point-in-mesh.  CPU: oracle vs the reference's golden outputs.
GPU: HIP kernels vs the same fixtures (tolerance 2e-6 abs on unit quaternions / 1e-6 abs on distances of ~0.05)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import geometry_oracle as G


def test_oracle_pose_decode_matches_reference():
    fix = load_golden("geometry.npz")
    tsl, quat = G.pose_decode(torch.from_numpy(fix["pose"]))
    np.testing.assert_array_equal(tsl.numpy(), fix["pose"][:, :3])
    np.testing.assert_allclose(quat.numpy(), fix["quat"], rtol=0, atol=1e-7)
    assert np.allclose(np.linalg.norm(fix["quat"][5:], axis=-1), 1.0, atol=1e-5) and (fix["quat"][..., 0] >= 0).all()


def test_oracle_h2o_matches_reference():
    fix = load_golden("geometry.npz")
    out = G.h2o_dist(torch.from_numpy(fix["hand_verts"]), torch.from_numpy(fix["obj_traj"]), torch.from_numpy(fix["obj_points"]),
                     fix["obj_num"])
    np.testing.assert_allclose(out.numpy(), fix["h2o"], rtol=0, atol=1e-7)


def test_oracle_contact_matches_reference():
    """fixture = the reference's transform helpers + torch.cdist(...).min exactly as compute_score_cr.py:122-149 calls them;
    cdist's matmul formulation is good to ~1e-6 here, hence the tolerance"""
    fix = load_golden("contact.npz")
    d = G.contact_min_dist(torch.from_numpy(fix["hand_verts"]), torch.from_numpy(fix["obj_traj"]), torch.from_numpy(fix["obj_points"]))
    np.testing.assert_allclose(d.numpy(), fix["min_dist"], rtol=0, atol=3e-6)
    assert G.contact_ratio(d) == pytest.approx(float(fix["contact_ratio"]), abs=1e-12)
    assert 0.1 < float(fix["contact_ratio"]) < 0.9  # the fixture has frames on both sides of the 5 mm threshold


@pytest.mark.gpu
def test_hip_contact_min_dist():
    from oakink2_tamf_amd import geometry

    fix = load_golden("contact.npz")
    hv, tr, pts = (torch.from_numpy(fix[k]).cuda() for k in ("hand_verts", "obj_traj", "obj_points"))
    d = geometry.contact_min_dist(hv, tr, pts)
    assert d.shape == fix["min_dist"].shape
    np.testing.assert_allclose(d.cpu().numpy(), fix["min_dist"], rtol=0, atol=3e-6)
    # bit-identical to the minimum of the per-vertex feature the same kernel produces
    assert torch.equal(d, geometry.multi_object_h2o_dist(hv, tr, pts).min(dim=-1).values)
    assert geometry.contact_ratio(d) == pytest.approx(float(fix["contact_ratio"]), abs=1e-12)
    # ragged: per-clip object counts and clip lengths (`avai_len`)
    ref = G.contact_min_dist(hv.cpu(), tr.cpu(), pts.cpu(), [2, 1, 2])
    got = geometry.contact_min_dist(hv, tr, pts, [2, 1, 2])
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-6)
    lens = [12, 7, 3]
    keep = torch.arange(12)[None, :] < torch.tensor(lens)[:, None]
    assert geometry.contact_ratio(got, lens) == pytest.approx(float((ref[keep] < 0.005).double().mean()), abs=1e-12)


@pytest.mark.gpu
def test_hip_pose_decode():
    from oakink2_tamf_amd import geometry

    fix = load_golden("geometry.npz")
    tsl, quat = geometry.pose_repr_to_quat(torch.from_numpy(fix["pose"]).cuda())
    np.testing.assert_array_equal(tsl.cpu().numpy(), fix["pose"][:, :3])
    q, r = quat.cpu().numpy(), fix["quat"]
    # rows 1-4 are exact 180-degree rotations / degenerate inputs: q and -q describe the same rotation when w == 0
    err = np.minimum(np.abs(q - r).max(-1), np.abs(q + r).max(-1) + (np.abs(r[..., 0]) > 1e-6) * 1e9)
    assert err.max() < 2e-6, err.max()
    # batched shape (B, T, 99)
    p3 = torch.from_numpy(fix["pose"]).reshape(4, 16, 99).cuda()
    _, q3 = geometry.pose_repr_to_quat(p3)
    assert q3.shape == (4, 16, 16, 4) and torch.equal(q3.reshape(64, 16, 4), quat)


@pytest.mark.gpu
def test_hip_h2o_dist():
    from oakink2_tamf_amd import geometry

    fix = load_golden("geometry.npz")
    out = geometry.multi_object_h2o_dist(torch.from_numpy(fix["hand_verts"]).cuda(), torch.from_numpy(fix["obj_traj"]).cuda(),
                                         torch.from_numpy(fix["obj_points"]).cuda(), fix["obj_num"].tolist())
    np.testing.assert_allclose(out.cpu().numpy(), fix["h2o"], rtol=0, atol=1e-6)
    # ragged point count (P not a multiple of the 256-point tile) and full object list against the oracle
    g = torch.Generator().manual_seed(0)
    hv = torch.randn(1, 3, 778, 3, generator=g) * 0.1
    tr = torch.randn(1, 3, 3, 9, generator=g)
    pts = torch.randn(1, 3, 1000, 3, generator=g) * 0.1
    ref = G.h2o_dist(hv, tr, pts)
    got = geometry.multi_object_h2o_dist(hv.cuda(), tr.cuda(), pts.cuda())
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-6)


def _siv_cases():
    fix = load_golden("siv.npz")
    return [(n, fix[n + "/verts"], fix[n + "/faces"], fix[n + "/points"], fix[n + "/contains"]) for n in ("blob", "torus")]


def test_oracle_mesh_contains_matches_reference():
    """fixture = the reference's check_mesh_contains with its own Cython TriangleHash (oracle/build_ref.sh); booleans, exact"""
    for name, v, f, p, ref in _siv_cases():
        got = G.mesh_contains(v, f, p)
        assert got.dtype == bool and np.array_equal(got, ref), name
        assert 50 < ref.sum() < len(ref) // 2  # the fixture has points on both sides
    # points outside the bounding box and the empty query
    assert not G.mesh_contains(v, f, np.array([[10.0, 10.0, 10.0]])).any()
    assert G.mesh_contains(v, f, np.zeros((0, 3))).shape == (0,)
    # SIV in cm^3: count x voxel volume x 1e6
    name, v, f, p, ref = _siv_cases()[0]
    assert G.solid_intersection_volume(v, f, [p, p[:100]], [8e-9, 1e-9]) == pytest.approx((ref.sum() * 8e-9 + ref[:100].sum() * 1e-9) * 1e6)


@pytest.mark.gpu
def test_hip_mesh_contains():
    from oakink2_tamf_amd import geometry

    for name, v, f, p, ref in _siv_cases():
        got = geometry.mesh_contains(v, f, torch.from_numpy(p).cuda())
        assert got.dtype == torch.bool and np.array_equal(got.cpu().numpy(), ref), name  # bit-exact against the reference
    # a larger random query against the oracle, mesh given as device tensors
    name, v, f, _, _ = _siv_cases()[1]
    g = np.random.default_rng(5)
    q = v.min(0) + g.random((50000, 3)) * (v.max(0) - v.min(0))
    got = geometry.mesh_contains(torch.from_numpy(v).cuda(), torch.from_numpy(f).cuda(), torch.from_numpy(q).cuda())
    assert np.array_equal(got.cpu().numpy(), G.mesh_contains(v, f, q))
    assert geometry.mesh_contains(v, f, torch.zeros(0, 3).cuda()).shape == (0,)
    vol = geometry.solid_intersection_volume(v, f, [torch.from_numpy(q).cuda()], [1e-9])
    assert vol == pytest.approx(float(G.mesh_contains(v, f, q).sum()) * 1e-9 * 1e6)


def test_oracle_transform_points_matches_reference():
    """fixture: tslrot6d_to_transf_np + transf_point_array_np of the reference in float64 (every 25th point of clip 0)"""
    fix = load_golden("contact.npz")
    tr, pts = torch.from_numpy(fix["obj_traj"][0]).double(), torch.from_numpy(fix["obj_points"][0]).double()
    got = G.transform_points(tr, pts)
    assert got.shape == (2, 12, 700, 3)
    np.testing.assert_allclose(got[:, :, ::25].numpy(), fix["moved_clip0_f64"], rtol=0, atol=1e-15)


@pytest.mark.gpu
def test_hip_transform_points():
    from oakink2_tamf_amd import geometry

    fix = load_golden("contact.npz")
    tr, pts = torch.from_numpy(fix["obj_traj"]), torch.from_numpy(fix["obj_points"])
    got64 = geometry.transform_points(tr[0].double().cuda(), pts[0].double().cuda())
    assert got64.dtype == torch.float64 and got64.shape == (2, 12, 700, 3)
    np.testing.assert_allclose(got64[:, :, ::25].cpu().numpy(), fix["moved_clip0_f64"], rtol=0, atol=1e-14)
    got32 = geometry.transform_points(tr.cuda(), pts.cuda())  # batched (B, nobj, ...) float32
    assert got32.dtype == torch.float32 and got32.shape == (3, 2, 12, 700, 3)
    ref = G.transform_points(tr.double(), pts.double())
    np.testing.assert_allclose(got32.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6)


# ---- vertex normals (segment_refine_model.py:131-133 -> pytorch3d verts_normals_packed) ---------------------------------

def test_oracle_vertex_normals_analytic_cases():
    """pytorch3d is absent: the restatement of its published algorithm is pinned on cases with known answers."""
    from oracle import fixtures as FX

    # unit cube, outward-wound: every corner is met by 3 faces' worth of triangles; by symmetry the area-weighted sum of the
    # incident triangle normals points along a (+-1, +-1, +-1) direction only when the triangle areas balance - check the
    # exact float64 evaluation of the definition instead, plus unit length and outwardness
    v = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4],
                  [1, 5, 7], [1, 7, 3]], np.int64)
    n = G.vertex_normals(v, f)
    ref = np.zeros((8, 3))
    for a, b, c in f:
        fn = np.cross(v[b].astype(np.float64) - v[a], v[c].astype(np.float64) - v[a])
        for k in (a, b, c):
            ref[k] += fn
    ref /= np.linalg.norm(ref, axis=1, keepdims=True)
    np.testing.assert_allclose(n, ref, atol=1e-6)
    assert (np.einsum("ij,ij->i", n, v - 0.5) > 0).all()  # outward
    # icosphere: normals of a sphere mesh are radial
    sv, sf = FX.icosphere(3)
    sn = G.vertex_normals(sv.astype(np.float32), sf)
    assert np.einsum("ij,ij->i", sn, sv / np.linalg.norm(sv, axis=1, keepdims=True)).min() > 0.9995
    # degenerate vertex (no incident face) -> zero vector (x / max(|x|, eps)); batch axes are preserved
    v2 = np.concatenate([v, [[5, 5, 5]]]).astype(np.float32)
    n2 = G.vertex_normals(np.stack([v2, v2 * 2]), f)
    assert n2.shape == (2, 9, 3) and (n2[:, 8] == 0).all()
    np.testing.assert_allclose(n2[1, :8], n, atol=1e-6)  # scale invariance


def test_oracle_vertex_normals_golden():
    fix = load_golden("vertex_normals.npz")
    np.testing.assert_array_equal(G.vertex_normals(fix["verts"], fix["faces"]), fix["normals"])


def test_vertex_incidence_csr_order():
    """the incidence list follows the CPU index_add_ order: corner 1 of all faces, corner 2, corner 0; faces ascending"""
    from oakink2_tamf_amd import geometry

    f = np.array([[0, 1, 2], [2, 1, 3], [0, 2, 3]])
    off, ent = geometry.vertex_incidence_csr(f, 5)
    assert off.tolist() == [0, 2, 4, 7, 9, 9]
    # vertex 2: corner 1 of face 2 -> (next, prev) = (3, 0); corner 2 of face 0 -> (0, 1); corner 0 of face 1 -> (1, 3)
    assert ent[off[2]:off[3]].tolist() == [[3, 0], [0, 1], [1, 3]]


@pytest.mark.gpu
def test_hip_vertex_normals():
    from oakink2_tamf_amd import geometry
    from oracle import fixtures as FX

    fix = load_golden("vertex_normals.npz")
    got = geometry.vertex_normals(torch.from_numpy(fix["verts"]).cuda(), fix["faces"])
    np.testing.assert_allclose(got.cpu().numpy(), fix["normals"], rtol=0, atol=1e-6)
    # the MANO-sized case: 778 vertices, a sequence of 196 frames, batch axis in front; bit-for-bit against the oracle
    # (same summation order, no FMA contraction)
    sv, sf = FX.icosphere(3)  # 642 vertices -> pad to 778 with unreferenced vertices
    v = np.zeros((778, 3), np.float32)
    v[:642] = sv
    g = torch.Generator().manual_seed(4)
    seq = (torch.from_numpy(v)[None, None] * (1 + 0.05 * torch.randn(2, 196, 1, 1, generator=g)) + 0.01 * torch.randn(2, 196, 778, 3, generator=g))
    got = geometry.vertex_normals(seq.cuda(), sf).cpu().numpy()
    ref = G.vertex_normals(seq.numpy(), sf)
    assert got.shape == (2, 196, 778, 3)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-7)
    assert (got[:, :, 642:] == 0).all()


@pytest.mark.gpu
def test_hip_vertex_normals_numpy_faces_are_keyed_by_content():
    """a numpy face array edited in place must not hit the CSR cached for its old content (torch.as_tensor of a numpy
    array always has _version 0), and equal lists hit the cache."""
    from oakink2_tamf_amd import geometry
    from oracle import fixtures as FX

    sv, sf = FX.icosphere(1)
    v = torch.from_numpy(sv.astype(np.float32)).cuda()
    faces = np.array(sf, dtype=np.int64)
    n1 = geometry.vertex_normals(v, faces).cpu().numpy()
    np.testing.assert_allclose(n1, G.vertex_normals(sv.astype(np.float32), faces), rtol=0, atol=2e-7)
    faces[:, [1, 2]] = faces[:, [2, 1]]  # flip every triangle IN PLACE: the normals must flip too
    n2 = geometry.vertex_normals(v, faces).cpu().numpy()
    np.testing.assert_allclose(n2, -n1, rtol=0, atol=2e-7)
    n_before = len(geometry._CSR_CACHE)
    geometry.vertex_normals(v, faces.tolist())  # same content as a list: same key
    assert len(geometry._CSR_CACHE) == n_before
