"""Generate tests/golden/*.npz by running the REFERENCE implementation in the build container.

Run here only:  python -m oracle.capture_golden   (needs /root/reference; never runs on the GPU box)

The reference (pure Python/PyTorch) is imported from /root/reference/src with a stub `clip` module
(CLIP's package and weights are absent and its output is an *input* of the path).
Weights come from oracle.det (seed-free recipe) and are loaded into the reference module through its own
load_state_dict; inputs / noise come from the same recipe, so fixtures hold only small inputs and the
reference's outputs.  Nothing from /root/reference is copied: fixtures are data.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

from . import det
from . import mdm_oracle as O
from .fixtures import guidance_fn

REF_SRC = "/root/reference/src"
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _install_clip_stub():
    clip = types.ModuleType("clip")
    clip.model = types.ModuleType("clip.model")

    class _FakeClip(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self._emb = None

        def encode_text(self, tokens):
            return self._emb

    clip.load = lambda *a, **k: (_FakeClip(), None)
    clip.model.convert_weights = lambda m: None
    clip.tokenize = lambda texts, context_length=77, truncate=True: torch.zeros(
        (len(texts), context_length), dtype=torch.long
    )
    sys.modules["clip"] = clip
    sys.modules["clip.model"] = clip.model


def _ref_model(arch: O.Arch, sd):
    from oakink2_tamf.model.interaction_segment_mdm import InterationSegmentMDM

    m = InterationSegmentMDM(
        input_dim=arch.input_dim,
        obj_input_dim=arch.obj_input_dim,
        hand_shape_dim=arch.hand_shape_dim,
        obj_embed_dim=arch.obj_embed_dim,
        latent_dim=arch.latent_dim,
        ff_size=arch.ff_size,
        num_layers=arch.num_layers,
        num_heads=arch.num_heads,
        dropout=0.1,
        activation="gelu",
    )
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("clip_model") for k in missing), missing
    m.eval()  # returns None in the reference (interaction_segment_mdm.py:176-178)
    return m


def _ref_batch(cond, m):
    m.clip_model._emb = cond["text_embedding"]
    B = cond["text_embedding"].shape[0]
    return {
        "text": ["x"] * B,
        "hand_side": cond["hand_side"],
        "shape": cond["shape"],
        "obj_embedding": cond["obj_embedding"],
        "obj_traj": cond["obj_traj"],
    }


def _cond_np(cond):
    return {
        "text_embedding": cond["text_embedding"].numpy(),
        "hand_side": np.array([0 if h == "rh" else 1 for h in cond["hand_side"]], dtype=np.uint8),
        "shape": cond["shape"].numpy(),
        "obj_embedding": cond["obj_embedding"].numpy(),
        "obj_traj": cond["obj_traj"].numpy(),
    }


def capture_schedule():
    from oakink2_tamf.model.diffusion_util import create_gaussian_diffusion

    out = {}
    for n in (1000, 50):
        dif = create_gaussian_diffusion(diffusion_steps=n, noise_schedule="cosine")
        tab = O.make_tables(n, "cosine")
        for k in (
            "betas",
            "alphas_cumprod",
            "alphas_cumprod_prev",
            "posterior_variance",
            "posterior_log_variance_clipped",
            "posterior_mean_coef1",
            "posterior_mean_coef2",
            "sqrt_alphas_cumprod",
            "sqrt_one_minus_alphas_cumprod",
        ):
            ref = np.asarray(getattr(dif, k), dtype=np.float64)
            mine = getattr(tab, k)
            err = np.max(np.abs(ref - mine))
            print(f"schedule N={n} {k}: oracle-vs-reference max abs err {err:.3e}")
            out[f"n{n}/{k}"] = ref
        assert dif.timestep_map == list(range(n))
    np.savez_compressed(os.path.join(OUT_DIR, "schedule.npz"), **out)


def capture_forward(name: str, arch: O.Arch, B: int, T: int, ts, nobj=2, nonfinite=False, sd_fn=O.det_state_dict, cond_fn=O.det_cond):
    sd = sd_fn(arch, tag=f"{name}/w")
    m = _ref_model(arch, sd)
    cond = cond_fn(B, T, nobj=nobj, tag=f"{name}/c", arch=arch)
    x = torch.from_numpy(det.det_normal(f"{name}/x", (B, arch.input_dim, 1, T)))
    if nonfinite:
        # exercise the three nan_to_num sites (interaction_segment_mdm.py:158,166,173)
        cond["text_embedding"][0, 3] = float("nan")
        x[1, 5, 0, 2] = float("inf")
    batch = _ref_batch(cond, m)
    fix = {"x": x.numpy(), "B": B, "T": T, "nobj": nobj, "ts": np.array(ts, dtype=np.int64)}
    fix.update({f"cond/{k}": v for k, v in _cond_np(cond).items()})
    with torch.no_grad():
        for t in ts:
            tt = torch.full((B,), t, dtype=torch.long)
            ref = m(x, tt, batch)
            mine = O.denoiser_forward(sd, arch, x, tt, cond)
            mine64 = O.denoiser_forward(sd, arch, x, tt, cond, dtype=torch.float64)
            e32 = (ref - mine).abs().max().item()
            e64 = (ref.double() - mine64).abs().max().item()
            print(f"forward {name} t={t}: |ref-oracle32|={e32:.3e} |ref-oracle64|={e64:.3e} |ref|max={ref.abs().max():.3f}")
            fix[f"out/t{t}"] = ref.numpy()
        # per-sample distinct timesteps in one call (training-style call pattern, launch/train.py:518-524)
        tt = torch.tensor([ts[i % len(ts)] for i in range(B)], dtype=torch.long)
        fix["ts_mixed"] = tt.numpy()
        fix["out/mixed"] = m(x, tt, batch).numpy()
    np.savez_compressed(os.path.join(OUT_DIR, f"forward_{name}.npz"), **fix)


def capture_loop(name: str, arch: O.Arch, B: int, T: int, steps: int, store_noise: bool, dump_steps=None, sd_fn=O.det_state_dict,
                 cond_fn=O.det_cond, respacing=None, base_steps=None, guide=None):
    """respacing (round 6): the reference's SpacedDiffusion over a SUBSET of `base_steps` timesteps (respace.py:60-119; its factory
    hard-codes the full set, so the class is constructed directly with the factory's other arguments); `steps` is then the number of
    kept steps."""
    from oakink2_tamf.model.diffusion_util import create_gaussian_diffusion
    from oakink2_tamf.model.diffusion import gaussian_diffusion as gd

    sd = sd_fn(arch, tag=f"{name}/w")
    m = _ref_model(arch, sd)
    cond = cond_fn(B, T, tag=f"{name}/c", arch=arch)
    batch = _ref_batch(cond, m)
    shape = (B, arch.input_dim, 1, T)
    if respacing is None:
        dif = create_gaussian_diffusion(diffusion_steps=steps, noise_schedule="cosine")
    else:
        from oakink2_tamf.model.diffusion.respace import SpacedDiffusion, space_timesteps

        dif = SpacedDiffusion(use_timesteps=space_timesteps(base_steps, respacing), betas=gd.get_named_beta_schedule("cosine", base_steps, 1.0),
                              model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
                              loss_type=gd.LossType.MSE, rescale_timesteps=False)
        assert dif.num_timesteps == steps, (dif.num_timesteps, steps)

    calls = {"k": 0}
    draws = []

    def draw(k):
        z = torch.from_numpy(det.det_normal(det.step_noise_tag(f"{name}/eps", k), shape))
        return z

    class _ThProxy:
        """torch look-alike handed to the reference module so that its th.randn / th.randn_like calls
        (gaussian_diffusion.py:604,448) return the recipe's draws in call order."""

        def __getattr__(self, a):
            return getattr(torch, a)

        def randn(self, *s, **kw):
            k = calls["k"]
            calls["k"] += 1
            z = draw(k)
            draws.append(z)
            return z

        def randn_like(self, x):
            return self.randn(*x.shape)

    real_th = gd.th
    gd.th = _ThProxy()
    try:
        with torch.no_grad():
            res = dif.p_sample_loop(
                m, shape, clip_denoised=False, model_kwargs={"batch": batch}, dump_steps=dump_steps, cond_fn=guide
            )
    finally:
        gd.th = real_th
    assert calls["k"] == steps + 1
    if respacing is None:
        tab = O.make_tables(steps, "cosine")
        fix = {"B": B, "T": T, "steps": steps}
    else:
        tab = O.make_tables(base_steps, "cosine", O.space_timesteps(base_steps, respacing))
        assert tab.timestep_map == list(dif.timestep_map)
        fix = {"B": B, "T": T, "steps": steps, "base_steps": base_steps, "respacing": np.array(respacing),
               "timestep_map": np.array(dif.timestep_map, dtype=np.int64)}
        for k in ("betas", "posterior_mean_coef1", "posterior_mean_coef2", "posterior_log_variance_clipped"):
            ref_t = np.asarray(getattr(dif, k), dtype=np.float64)
            print(f"respaced schedule {name} {k}: oracle-vs-reference max abs err {np.max(np.abs(ref_t - getattr(tab, k))):.3e}")
            fix[f"tab/{k}"] = ref_t
    fix.update({f"cond/{k}": v for k, v in _cond_np(cond).items()})
    if dump_steps is not None:
        dump_ref = res
        dump_mine: list = []
        O.sample_loop(sd, arch, tab, cond, shape, draw, dump=dump_mine, cond_fn=guide)
        for j, s in enumerate(dump_steps):
            e = (dump_ref[j] - dump_mine[s]).abs().max().item()
            print(f"loop {name}: dump step {s}: |ref-oracle32| = {e:.3e}")
            fix[f"dump/{s}"] = dump_ref[j].numpy()
        fix["dump_steps"] = np.array(dump_steps)
        final = dump_ref[-1]
    else:
        final = res
        mine = O.sample_loop(sd, arch, tab, cond, shape, draw)
        mine64 = O.sample_loop(sd, arch, tab, cond, shape, draw, dtype=torch.float64)
        print(
            f"loop {name} ({steps} steps): |ref-oracle32|={(final - mine).abs().max():.3e} "
            f"|ref-oracle64|={(final.double() - mine64).abs().max():.3e} |ref|mean={final.abs().mean():.3f}"
        )
    fix["final"] = final.numpy()
    if store_noise:
        fix["draws"] = np.stack([d.numpy() for d in draws], axis=0)
    np.savez_compressed(os.path.join(OUT_DIR, f"loop_{name}.npz"), **fix)


def capture_refine(name: str, arch: O.Arch, B: int, T: int, nobj=2):
    """R trunk (segment_refine_model.py:175-217) with the external pieces stubbed as SURVEY.md 8c prescribes:
    manotorch.ManoLayer / pytorch3d.Meshes are stand-in modules, MANO recovery returns zeros and the
    hand->object distance is the supplied tensor, so refine_pose_repr pins exactly the trunk."""
    mt = types.ModuleType("manotorch")
    mtl = types.ModuleType("manotorch.manolayer")

    class ManoLayer(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.register_buffer("th_faces", torch.zeros(1538, 3, dtype=torch.long))

    mtl.ManoLayer = ManoLayer
    mt.manolayer = mtl
    p3 = types.ModuleType("pytorch3d")
    p3s = types.ModuleType("pytorch3d.structures")
    p3s.Meshes = object
    p3.structures = p3s
    for k, v in (("manotorch", mt), ("manotorch.manolayer", mtl), ("pytorch3d", p3), ("pytorch3d.structures", p3s)):
        sys.modules.setdefault(k, v)
    from oakink2_tamf.model.segment_refine_model import SegmentRefineModel

    h2o = torch.from_numpy(det.det_normal(f"{name}/h2o", (B, T, arch.h2o_dim))) * 0.05

    class Trunk(SegmentRefineModel):
        def batch_recover_mano_from_pose_repr(self, pose_repr, shape, hand_side):
            b, t = pose_repr.shape[:2]
            return torch.zeros(b, t, 778, 3), torch.zeros(b, t, 21, 3), torch.zeros(b, t, 778, 3)

        def multi_object_h2o_dist(self, *a, **k):
            return h2o

    sd = O.det_state_dict(arch, tag=f"{name}/w")
    m = Trunk(None, input_dim=arch.input_dim, obj_input_dim=arch.obj_input_dim, hand_shape_dim=arch.hand_shape_dim,
              obj_embed_dim=arch.obj_embed_dim, latent_dim=arch.latent_dim, ff_size=arch.ff_size,
              num_layers=arch.num_layers, num_heads=arch.num_heads)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("mano_layer") for k in missing), (missing, unexpected)
    m.eval()
    cond = O.det_cond(B, T, nobj=nobj, tag=f"{name}/c", arch=arch)
    x_in = torch.from_numpy(det.det_normal(f"{name}/x", (B, T, arch.input_dim)))
    batch = {"sample_pose_repr": x_in, "pose_repr": x_in, "hand_side": cond["hand_side"], "shape": cond["shape"],
             "obj_embedding": cond["obj_embedding"], "obj_traj": cond["obj_traj"], "obj_list": None, "obj_verts": None}
    with torch.no_grad():
        ref = m(batch)["refine_pose_repr"]
    mine = O.refine_forward(sd, arch, x_in, h2o, cond)
    print(f"refine {name}: |ref-oracle32|={(ref - mine).abs().max():.3e} |ref|max={ref.abs().max():.3f}")
    fix = {"x_in": x_in.numpy(), "h2o": h2o.numpy(), "out": ref.numpy(), "B": B, "T": T}
    fix.update({f"cond/{k}": v for k, v in _cond_np(cond).items() if k != "text_embedding"})
    np.savez_compressed(os.path.join(OUT_DIR, f"refine_{name}.npz"), **fix)


def capture_vertex_normals():
    """Row 8(f)-1, vertex normals: pytorch3d is absent, so this fixture comes from the restatement of its published
    algorithm (geometry_oracle.vertex_normals) - it pins the KERNEL on the oracle and the oracle on a file; the oracle
    itself is pinned on analytic cases (tests/test_geometry.py)."""
    from . import fixtures as FX
    from . import geometry_oracle as G

    v, f = FX.icosphere(2)
    frames = []
    for k in range(3):  # a breathing, sheared sphere: three "frames" of one topology
        d = det.det_normal(f"vnormal/{k}", v.shape).astype(np.float32) * 0.03
        frames.append((v * (1.0 + 0.1 * k) + d + np.array([0.1 * k, 0.0, 0.2], np.float32)).astype(np.float32))
    verts = np.stack(frames)
    normals = G.vertex_normals(verts, f)
    np.savez_compressed(os.path.join(OUT_DIR, "vertex_normals.npz"), verts=verts, faces=f.astype(np.int32), normals=normals)
    print("vertex_normals", verts.shape, f.shape)


def capture_geometry():
    """Rows 8(f)-1/-2: the reference's own rot6d/quaternion helpers (dev_fn.transform) and its hand->object distance
    (SegmentRefineModel.multi_object_h2o_dist -> point2point_signed) with a stand-in `chamfer_distance` module that
    implements the published brute-force nearest-neighbour search (source of the real CUDA extension is absent)."""
    from dev_fn.transform.rotation import rot6d_to_rotmat, rotmat_to_quat
    from . import geometry_oracle as G

    chd = types.ModuleType("chamfer_distance")

    class ChamferDistance(torch.nn.Module):
        def forward(self, x, y):
            d2 = ((x[:, :, None, :] - y[:, None, :, :]) ** 2).sum(-1)
            d1, i1 = d2.min(dim=2)
            dd2, i2 = d2.min(dim=1)
            return d1, dd2, i1.int(), i2.int()

    chd.ChamferDistance = ChamferDistance
    sys.modules["chamfer_distance"] = chd
    from oakink2_tamf.model.segment_refine_model import SegmentRefineModel

    # pose decode: random poses plus degenerate ones (zero vectors, near-180-degree rotations -> every quaternion branch)
    N = 64
    pose = torch.from_numpy(det.det_normal("geom/pose", (N, 99)))
    pose[0, 3:9] = 0.0
    pose[1, 3:9] = torch.tensor([1.0, 0, 0, 0, -1.0, 0])   # 180 deg about x
    pose[2, 3:9] = torch.tensor([-1.0, 0, 0, 0, 1.0, 0])   # 180 deg about y
    pose[3, 3:9] = torch.tensor([-1.0, 0, 0, 0, -1.0, 0])  # 180 deg about z
    pose[4, 3:9] = torch.tensor([1.0, 0, 0, 1.0, 0, 0])    # collinear
    rot6d = pose[:, 3:99].reshape(N, 16, 6)
    quat = rotmat_to_quat(rot6d_to_rotmat(rot6d))
    tsl_m, quat_m = G.pose_decode(pose)
    print(f"pose decode: |ref-oracle| = {(quat - quat_m).abs().max():.3e}")

    # h2o distance: 2 clips, 2 (padded) objects, the second clip has one real object
    B, T, V, nobj, P = 2, 6, 778, 2, 500
    hv = torch.from_numpy(det.det_normal("geom/hv", (B, T, V, 3))) * 0.1
    traj = torch.from_numpy(det.det_normal("geom/traj", (B, nobj, T, 9)))
    traj[..., 0:3] *= 0.1
    pts = torch.from_numpy(det.det_normal("geom/pts", (B, nobj, P, 3))) * 0.1
    obj_num = [2, 1]
    obj_list = [["a", "b"], ["a"]]
    ref = SegmentRefineModel.multi_object_h2o_dist(None, hv, torch.zeros_like(hv), obj_list, traj,
                                                   [pts[b].numpy() for b in range(B)])
    mine = G.h2o_dist(hv, traj, pts, obj_num)
    print(f"h2o dist: |ref-oracle| = {(ref - mine).abs().max():.3e}  mean dist {ref.mean():.4f}")
    np.savez_compressed(os.path.join(OUT_DIR, "geometry.npz"), pose=pose.numpy(), quat=quat.numpy(), hand_verts=hv.numpy(),
                        obj_traj=traj.numpy(), obj_points=pts.numpy(), obj_num=np.array(obj_num, dtype=np.int32),
                        h2o=ref.numpy())


def capture_contact():
    """Row 8(f)-4 (Contact Ratio): script/compute_score/compute_score_cr.py cannot be imported (config_reg, the dataset
    toolkit and manotorch are absent), so its two functions are replayed call by call on the reference's own library:
    transf_merge_obj_pointcloud (:122-137) = tslrot6d_to_transf_np + transf_point_array_np (dev_fn/transform/transform_np)
    and contact_min_cdist (:140-149) = torch.cdist(hv, pc, p=2) -> min over (vertex, point) per frame."""
    from dev_fn.transform.transform_np import tslrot6d_to_transf_np, transf_point_array_np
    from . import geometry_oracle as G

    B, T, V, nobj, P = 3, 12, 778, 2, 700
    hv = det.det_normal("contact/hv", (B, T, V, 3)).astype(np.float32) * 0.08
    traj = det.det_normal("contact/traj", (B, nobj, T, 9)).astype(np.float32)
    traj[..., 0:3] *= 0.22  # object centres wander in and out of the hand's vertex cloud: a mix of contact / no contact
    pts = det.det_normal("contact/pts", (B, nobj, P, 3)).astype(np.float32) * 0.02
    # clip 2: object far away -> no contact in any frame
    traj[2, :, :, 0:3] += 1.0
    ref = np.zeros((B, T), np.float32)
    for b in range(B):
        obj_traj, obj_pointcloud = traj[b], pts[b]                                  # (nobj, T, 9), (nobj, P, 3)
        transf = tslrot6d_to_transf_np(obj_traj)                                    # (nobj, T, 4, 4)
        pc = np.broadcast_to(np.expand_dims(obj_pointcloud, 1), (nobj, T, P, 3))
        pc = transf_point_array_np(transf, pc)                                      # (nobj, T, P, 3)
        pc = np.swapaxes(pc, 0, 1).reshape((T, -1, 3))
        dist = torch.cdist(torch.from_numpy(hv[b]).float(), torch.from_numpy(np.ascontiguousarray(pc)).float(), p=2)
        ref[b] = dist.reshape(T, -1).min(dim=1).values.numpy()
    # the transformed clouds themselves (row 8f-2, second half), float64 through the reference's helpers
    t64, p64 = traj[0].astype(np.float64), pts[0].astype(np.float64)
    moved = transf_point_array_np(tslrot6d_to_transf_np(t64), np.broadcast_to(np.expand_dims(p64, 1), (nobj, T, P, 3)))
    moved_mine = G.transform_points(torch.from_numpy(t64), torch.from_numpy(p64)).numpy()
    print(f"transform_points: |ref-oracle| = {np.abs(moved - moved_mine).max():.3e}")
    mine = G.contact_min_dist(torch.from_numpy(hv), torch.from_numpy(traj), torch.from_numpy(pts)).numpy()
    ratio = float(np.mean(ref < 0.005))
    print(f"contact min dist: |ref-oracle| = {np.abs(ref - mine).max():.3e}  min {ref.min():.5f} contact ratio {ratio:.4f}"
          f" (oracle {G.contact_ratio(torch.from_numpy(mine)):.4f})")
    np.savez_compressed(os.path.join(OUT_DIR, "contact.npz"), hand_verts=hv, obj_traj=traj, obj_points=pts, min_dist=ref,
                        contact_ratio=np.float64(ratio), moved_clip0_f64=moved[:, :, ::25, :])


def capture_collate():
    """Row 8(f)-3: the reference's interaction_segment_collate (dataset/collate.py:33-58) on a ragged batch."""
    from oakink2_tamf.dataset.collate import interaction_segment_collate

    from .fixtures import ragged_clips

    out = interaction_segment_collate(ragged_clips())
    arrays = {}
    for k, v in out.items():
        if isinstance(v, torch.Tensor):
            arrays["t__" + k] = v.numpy()
            arrays["dtype__" + k] = np.array(str(v.dtype))
    arrays["listed_keys"] = np.array(sorted(k for k, v in out.items() if not isinstance(v, torch.Tensor)))
    np.savez_compressed(os.path.join(OUT_DIR, "collate.npz"), **arrays)
    print("collate:", {k: (tuple(v.shape), str(v.dtype)) if isinstance(v, torch.Tensor) else type(v).__name__ for k, v in out.items()})


def _install_toolkit_stubs():
    """oakink2_toolkit (dataset walker + constants) and manotorch are absent; dataset/interaction_segment.py imports both at
    module level.  With a cache dict its constructor only instantiates OakInk2__Dataset and, for enable_obj_model, asks it
    for the object meshes: the stub serves the synthetic meshes of oracle.fixtures."""
    from types import SimpleNamespace

    from . import fixtures

    tk = types.ModuleType("oakink2_toolkit")
    tk.dataset = types.ModuleType("oakink2_toolkit.dataset")
    tk.meta = types.ModuleType("oakink2_toolkit.meta")

    class OakInk2__Dataset:
        def __init__(self, dataset_prefix=None, return_instantiated=True):
            self.dataset_prefix = dataset_prefix

        def load_affordance(self, obj_id):
            v, f = fixtures.synthetic_object_mesh(obj_id)
            return SimpleNamespace(obj_mesh=SimpleNamespace(vertices=v, faces=f))

    tk.dataset.OakInk2__Dataset = OakInk2__Dataset
    tk.meta.FPS_MOCAP, tk.meta.HAND_SIDE, tk.meta.HAND_SIDE_MAP = 120.0, ["rh", "lh"], {"rh": "right", "lh": "left"}
    for name, mod in (("oakink2_toolkit", tk), ("oakink2_toolkit.dataset", tk.dataset), ("oakink2_toolkit.meta", tk.meta)):
        sys.modules[name] = mod
    if "manotorch" not in sys.modules:
        sys.modules["manotorch"] = types.ModuleType("manotorch")


def _items_to_arrays(prefix, items, arrays):
    """item dicts -> npz entries: arrays as they are (dtype kept), everything else as one JSON document per item"""
    import json

    for i, it in enumerate(items):
        meta = {}
        for k, v in it.items():
            if isinstance(v, np.ndarray):
                arrays[f"{prefix}/{i}/{k}"] = v
            elif isinstance(v, list) and v and isinstance(v[0], np.ndarray):
                for j, a in enumerate(v):
                    arrays[f"{prefix}/{i}/{k}/{j}"] = a
                meta[k] = {"__arrays__": len(v)}
            else:
                meta[k] = v
        arrays[f"{prefix}/{i}/__meta__"] = np.array(json.dumps(meta))
        arrays[f"{prefix}/{i}/__keys__"] = np.array(list(it.keys()))


def capture_cache_dict():
    """Row 8(f)-3, input side: the reference's InteractionSegmentData on its cache-dict path (constructor :285-387 with the
    toolkit stubbed, __getitem__ :389-449, reverse twins :162-265), GeneratedPoseReprSampleAdaptor (pose_repr_sample.py:18-52)
    over a two-directory .npy tree, and the slicer (setment_slice.py:10-36) - on oracle.fixtures' synthetic segment cache."""
    import tempfile

    _install_toolkit_stubs()
    from oakink2_tamf.dataset.interaction_segment import InteractionSegmentData
    from oakink2_tamf.dataset.pose_repr_sample import GeneratedPoseReprSampleAdaptor
    from oakink2_tamf.dataset.setment_slice import segment_slice_from_gap

    from . import fixtures

    arrays = {}
    with tempfile.TemporaryDirectory() as root:
        paths, cache = fixtures.write_synthetic_dataset(root)
        ds = InteractionSegmentData(process_range_list=["ignored"], data_prefix="/nonexistent", obj_embedding_prefix=paths["emb"],
                                    enable_obj_model=True, obj_pointcloud_prefix=paths["pc"], cache_dict=cache)
        _items_to_arrays("fwd", [ds[i] for i in range(len(ds))], arrays)
        ds_rev = InteractionSegmentData(process_range_list=[], data_prefix="/nonexistent", obj_embedding_prefix=paths["emb"],
                                        cache_dict=cache, append_reverse_segment=True)
        assert len(ds_rev) == 2 * len(ds)
        _items_to_arrays("rev", [ds_rev[i] for i in range(len(ds), len(ds_rev))], arrays)
        # the G stage's output tree: two directories, names out of numeric order on purpose
        dirs = [os.path.join(root, "sample", "b_part"), os.path.join(root, "sample", "a_part")]
        split = [[3, 0, 10], [2, 1]]
        for d, ids in zip(dirs, split):
            os.makedirs(d)
            for sid in ids:
                np.save(os.path.join(d, f"{sid:06d}.npy"), fixtures.synthetic_sample_pose_repr(os.path.basename(d), sid))
        ds_plain = InteractionSegmentData(process_range_list=[], data_prefix="/nonexistent", obj_embedding_prefix=paths["emb"], cache_dict=cache)
        ad = GeneratedPoseReprSampleAdaptor(ds_plain, dirs)
        _items_to_arrays("adaptor", [ad[i] for i in range(len(ad))], arrays)
    for n, gap, mx, mn in ((500, 12, 160, 16), (100, 12, 160, 16), (5000, 12, 160, 16), (1920, 12, 160, 16), (192, 12, 160, 16)):
        traj = np.arange(n * 2, dtype=np.float32).reshape(n, 2)
        clips, lens = segment_slice_from_gap(traj, gap, mx, mn)
        arrays[f"slice/{n}/clips"], arrays[f"slice/{n}/lens"] = np.stack(clips), np.asarray(lens)
    np.savez_compressed(os.path.join(OUT_DIR, "cache_dict_items.npz"), **arrays)
    print("cache_dict:", len(ds), "items,", len(arrays), "arrays,", os.path.getsize(os.path.join(OUT_DIR, "cache_dict_items.npz")) >> 10, "KiB")


def capture_siv():
    """Row 8(f)-4 (SIV): the reference's check_mesh_contains (dev_fn/external/libmesh/inside_mesh.py) with its Cython
    TriangleHash built from the reference source by oracle/build_ref.sh into oracle/_ref/libmesh.  The package is assembled at
    import time from the two directories (python files from the reference, the compiled module from oracle/_ref)."""
    import importlib.util
    from types import SimpleNamespace

    from . import geometry_oracle as G
    from .fixtures import siv_cases

    ref_pkg = os.path.join(REF_SRC, "dev_fn", "external", "libmesh")
    built = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libmesh")
    assert any(n.startswith("triangle_hash") for n in os.listdir(built)), "run oracle/build_ref.sh first"
    pkg = types.ModuleType("tamf_ref_libmesh")
    pkg.__path__ = [built, ref_pkg]
    sys.modules["tamf_ref_libmesh"] = pkg
    spec = importlib.util.spec_from_file_location("tamf_ref_libmesh.inside_mesh", os.path.join(ref_pkg, "inside_mesh.py"))
    inside_mesh = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = inside_mesh
    spec.loader.exec_module(inside_mesh)

    arrays = {}
    for name, v, f, pts in siv_cases():
        ref = inside_mesh.check_mesh_contains(SimpleNamespace(vertices=v, faces=f), pts)
        mine = G.mesh_contains(v, f, pts)
        print(f"siv {name}: {len(f)} faces, {len(pts)} points, inside {int(ref.sum())}, oracle mismatches {int((ref != mine).sum())}")
        arrays[f"{name}/verts"], arrays[f"{name}/faces"], arrays[f"{name}/points"] = v, f.astype(np.int32), pts
        arrays[f"{name}/contains"] = ref
    np.savez_compressed(os.path.join(OUT_DIR, "siv.npz"), **arrays)


def capture_loop_arch_mdm_l_1000():
    """The path the metric is quoted on: the reference's p_sample_loop (gaussian_diffusion.py:506-640) run 1000 x over
    arch_mdm_l at T = 196 (B = 2), det-recipe noise in reference call order; final + the states after steps 0, 499, 998."""
    capture_loop("arch_mdm_l_b2_t196_1000", O.ARCH_MDM_L, B=2, T=196, steps=1000, store_noise=False, dump_steps=[0, 499, 998, 999])


def capture_loop_arch_mdm_1000():
    """BASELINE.json configs[0]'s model (arch_mdm, B = 4, T = 64) over the full 1000-step schedule."""
    capture_loop("arch_mdm_b4_t64_1000", O.ARCH_MDM, B=4, T=64, steps=1000, store_noise=False, dump_steps=[0, 499, 998, 999])


def capture_stress():
    """the fp32-tolerance gate on trained-like dynamic range, at the dataset's clip length (T = 160), arch_mdm_l, B = 2:
    (i) real conditioning magnitudes on the default weights, (ii) those plus LayerNorm gains in [0.2, 5] and x30 outlier rows;
    each: one evaluation at t in {0, 500, 999} + a 50-step loop (states after steps 0, 24, 48 and the final sample)."""
    capture_forward("stress_cond_t160", O.ARCH_MDM_L, B=2, T=160, ts=[0, 500, 999], cond_fn=O.det_cond_stress)
    capture_loop("stress_cond_b2_t160_50", O.ARCH_MDM_L, B=2, T=160, steps=50, store_noise=False, dump_steps=[0, 24, 48, 49], cond_fn=O.det_cond_stress)
    capture_forward("stress_weights_t160", O.ARCH_MDM_L, B=2, T=160, ts=[0, 500, 999], sd_fn=O.det_state_dict_stress, cond_fn=O.det_cond_stress)
    capture_loop("stress_weights_b2_t160_50", O.ARCH_MDM_L, B=2, T=160, steps=50, store_noise=False, dump_steps=[0, 24, 48, 49],
                 sd_fn=O.det_state_dict_stress, cond_fn=O.det_cond_stress)


def capture_stress_dc():
    """Round 5: (iii) the stress weights + LayerNorm biases of 2 +- 0.5 on every feature (a DC component in every LayerNorm output, i.e.
    residual rows whose mean is several times their spread) - the input the HIP path's deferred LayerNorm is most exposed to."""
    capture_forward("stress_dc_t160", O.ARCH_MDM_L, B=2, T=160, ts=[0, 500, 999], sd_fn=O.det_state_dict_stress_dc, cond_fn=O.det_cond_stress)
    capture_loop("stress_dc_b2_t160_50", O.ARCH_MDM_L, B=2, T=160, steps=50, store_noise=False, dump_steps=[0, 24, 48, 49],
                 sd_fn=O.det_state_dict_stress_dc, cond_fn=O.det_cond_stress)


# --------------------------------------------------------------------------------------
# Weights that have been through the reference's own optimiser step
# --------------------------------------------------------------------------------------
TRAINED = {
    # name: (arch, training steps, batch, frames)
    "trained_tiny": (O.ARCH_TINY, 4000, 32, 40),
    "trained_hd128": (O.Arch(latent_dim=128, ff_size=256, num_layers=2, num_heads=1), 4000, 32, 40),  # head dim 128, as arch_mdm_l
}


def trained_weights_path(name: str) -> str:
    return os.path.join(OUT_DIR, f"{name}_weights.npz")


def trained_state_dict(arch: O.Arch, tag: str = "") -> dict:
    """sd_fn of capture_forward / capture_loop: the stored weights of the fixture the tag names (tag = '<name>.../w')."""
    name = next(n for n in TRAINED if tag.startswith(n))
    with np.load(trained_weights_path(name)) as z:
        sd = {k: torch.from_numpy(z[k].copy()) for k in z.files if not k.startswith("meta/")}
    pe = O.positional_table(arch.latent_dim).unsqueeze(1).contiguous()  # (buffers, not parameters: regenerated, not stored)
    for k in O.state_dict_spec(arch):
        if k.endswith(".pe"):
            sd[k] = pe
    return sd


def _smooth_motion_batch(gen: torch.Generator, arch: O.Arch, B: int, T: int):
    """Synthetic 'smooth motion' clips and their conditioning, with the magnitudes of the real data: a pose trajectory
    x0 = offset(cond) + a few low harmonics over the clip, CLIP features of norm 10, object trajectories [metres | unit rot6d].
    The pose offset depends LINEARLY on the conditioning (text, shape, object embedding, hand side) through fixed mixing matrices, so
    the denoiser has something to learn from every prefix token."""
    F = arch.input_dim
    te = torch.randn(B, arch.clip_dim, generator=gen)
    te = te / te.norm(dim=-1, keepdim=True) * 10.0
    shape = torch.randn(B, 1, arch.hand_shape_dim, generator=gen).repeat(1, T, 1).contiguous()
    oe = torch.randn(B, 2, arch.obj_embed_dim, generator=gen)
    raw = torch.randn(B, 2, T, 12, generator=gen)
    # smooth object translation: a random walk scaled to decimetres
    tsl = 0.02 * torch.cumsum(raw[..., :3], dim=2) + 0.3 * torch.randn(B, 2, 1, 3, generator=gen)
    a1, a2 = raw[:, :, :1, 3:6].expand(-1, -1, T, -1), raw[:, :, :1, 6:9].expand(-1, -1, T, -1)
    b1 = a1 / a1.norm(dim=-1, keepdim=True)
    a2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = a2 / a2.norm(dim=-1, keepdim=True)
    rot6d = torch.stack([b1, b2], dim=-1).reshape(B, 2, T, 6)
    traj = torch.cat([tsl, rot6d], dim=-1).contiguous()
    side = ["rh" if int(v) == 0 else "lh" for v in torch.randint(0, 2, (B,), generator=gen)]
    mix = torch.Generator().manual_seed(777)  # the fixed "world": how conditioning maps to motion
    M_t = torch.randn(arch.clip_dim, F, generator=mix) * (0.6 / 10.0 / arch.clip_dim ** 0.5) * 10.0
    M_s = torch.randn(arch.hand_shape_dim, F, generator=mix) * (0.4 / arch.hand_shape_dim ** 0.5)
    M_o = torch.randn(arch.obj_embed_dim, F, generator=mix) * (0.4 / arch.obj_embed_dim ** 0.5)
    v_side = torch.randn(F, generator=mix) * 0.3
    off = te @ M_t / 10.0 + shape[:, 0] @ M_s + oe.mean(dim=1) @ M_o
    off = off + torch.tensor([1.0 if h == "lh" else -1.0 for h in side])[:, None] * v_side
    tt = torch.arange(T, dtype=torch.float32) / T
    x = off[:, None, :].expand(B, T, F).clone()
    for k in (1, 2, 3):
        amp = torch.randn(B, 1, F, generator=gen) * (0.35 / k)
        ph = torch.rand(B, 1, F, generator=gen) * 6.2831853
        x = x + amp * torch.sin(6.2831853 * k * tt[None, :, None] + ph)
    x = x + 0.5 * tsl.mean(dim=1).repeat(1, 1, F // 3)[..., :F]  # the hand follows the objects
    cond = {"text_embedding": te, "hand_side": side, "shape": shape, "obj_embedding": oe, "obj_traj": traj}
    return x.contiguous(), cond


def capture_trained(name: str):
    """Train the REFERENCE module with the REFERENCE's own training step - GaussianDiffusion.training_losses
    (model/diffusion/gaussian_diffusion.py:1106-1188: q_sample + masked MSE on the x0 prediction), the uniform schedule sampler, AdamW
    lr 1e-4 / weight decay 0 and per-parameter gradient clipping clip_gradient(optimizer, 0.1, 2.0) exactly as launch/train.py:462-533
    runs it (model.train(): dropout 0.1 active) - on synthetic smooth motions, and store the resulting state dict.  Not reproduced:
    the MANO-based extra loss (InteractionSegmentExtraLoss needs manotorch + licence-gated assets: loss_callback=None) and DDP.
    The point is not motion quality: these are weights an optimiser produced - weight / LayerNorm statistics no hand-made recipe of
    oracle/mdm_oracle.py imitates - for the fp32-tolerance gates of the HIP path to be measured on."""
    from oakink2_tamf.model.diffusion_util import create_gaussian_diffusion
    from oakink2_tamf.model.diffusion.resample import create_named_schedule_sampler
    from oakink2_tamf.util.net_util import clip_gradient

    arch, steps, B, T = TRAINED[name]
    torch.manual_seed(20261004)
    sd0 = O.det_state_dict(arch, tag=f"{name}/init")
    m = _ref_model(arch, sd0)
    for p_ in m.clip_model.parameters():
        p_.requires_grad_(False)
    diffusion = create_gaussian_diffusion(diffusion_steps=1000, noise_schedule="cosine")
    sampler = create_named_schedule_sampler(name="uniform", diffusion=diffusion)
    params = [p_ for n_, p_ in m.named_parameters() if not n_.startswith("clip_model")]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.0)
    gen = torch.Generator().manual_seed(1)
    dev = torch.device("cpu")
    losses = []
    import time as _t

    t0 = _t.time()
    for it in range(steps):
        opt.zero_grad()
        m.train()
        x, cond = _smooth_motion_batch(gen, arch, B, T)
        batch = _ref_batch(cond, m)
        batch["mask"] = torch.ones(B, T)
        t, weights = sampler.sample(B, dev)
        x_start = x.unsqueeze(3).permute(0, 2, 3, 1)  # (bs, in_dim, 1, seqlen), launch/train.py:519-521
        loss_store, _ = diffusion.training_losses(m, x_start, t, model_kwargs={"batch": batch}, loss_callback=None)
        loss = (loss_store["loss"] * weights).mean()
        loss.backward()
        clip_gradient(opt, 0.1, 2.0)
        opt.step()
        losses.append(float(loss))
        if it % 250 == 0 or it == steps - 1:
            print(f"train {name}: step {it:5d} loss {sum(losses[-50:]) / len(losses[-50:]):.4f}  ({_t.time() - t0:.0f} s)", flush=True)
    m.eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.startswith("clip_model")}
    spec = O.state_dict_spec(arch)
    assert set(sd) >= set(spec), set(spec) - set(sd)
    out = {k: sd[k].numpy() for k in spec if not k.endswith(".pe")}
    moved = max(float((sd[k] - sd0[k]).abs().max()) for k in spec if not k.endswith(".pe"))
    out["meta/steps"] = np.int64(steps)
    out["meta/loss_first_last"] = np.array([sum(losses[:50]) / 50, sum(losses[-50:]) / 50])
    out["meta/max_weight_change"] = np.float64(moved)
    print(f"train {name}: loss {out['meta/loss_first_last'][0]:.4f} -> {out['meta/loss_first_last'][1]:.4f}, largest weight change {moved:.3f}")
    np.savez_compressed(trained_weights_path(name), **out)


def trained_cond(B: int, T: int, nobj: int = 2, tag: str = "c0", arch: O.Arch = O.ARCH_TINY):
    """cond_fn of the trained fixtures: conditioning drawn like the training data's (seeded by the tag)."""
    seed = int.from_bytes(tag.encode()[:8].ljust(8, b"\0"), "little") % (2 ** 31)
    _, cond = _smooth_motion_batch(torch.Generator().manual_seed(seed), arch, B, T)
    assert nobj == 2
    return cond


def capture_trained_all():
    for name, (arch, steps, B, T) in TRAINED.items():
        if not os.path.exists(trained_weights_path(name)) or os.environ.get("TAMF_RETRAIN"):
            capture_trained(name)
        capture_forward(name, arch, B=4, T=40, ts=[0, 1, 500, 999], sd_fn=trained_state_dict, cond_fn=trained_cond)
        capture_loop(f"{name}_b2_t40_1000", arch, B=2, T=40, steps=1000, store_noise=False, dump_steps=[0, 499, 998, 999],
                     sd_fn=trained_state_dict, cond_fn=trained_cond)


def capture_respaced():
    """The reference's SpacedDiffusion over 50 of the 1000 timesteps (space_timesteps(1000, "50")) and over the DDIM stride of 100
    ("ddim100"), trained weights: ancestral sampling on a strided subset - the denoiser is evaluated at timestep_map[t]."""
    arch = TRAINED["trained_hd128"][0]
    capture_loop("trained_hd128_respaced50_b2_t40", arch, B=2, T=40, steps=50, store_noise=False, dump_steps=[0, 24, 48, 49],
                 sd_fn=trained_state_dict, cond_fn=trained_cond, respacing="50", base_steps=1000)
    capture_loop("trained_hd128_respaced_ddim100_b2_t40", arch, B=2, T=40, steps=100, store_noise=False, dump_steps=[0, 49, 98, 99],
                 sd_fn=trained_state_dict, cond_fn=trained_cond, respacing="ddim100", base_steps=1000)
    # classifier-style guidance (cond_fn, gaussian_diffusion.py:346-357,453-454) on the respaced process: p_sample adds variance * gradient
    capture_loop("trained_tiny_guided_respaced20_b2_t40", TRAINED["trained_tiny"][0], B=2, T=40, steps=20, store_noise=False,
                 dump_steps=[0, 10, 18, 19], sd_fn=trained_state_dict, cond_fn=trained_cond, respacing="20", base_steps=1000, guide=guidance_fn)


def main():
    assert os.path.isdir(REF_SRC), "the reference is only present in the build container"
    _install_clip_stub()
    sys.path.insert(0, REF_SRC)
    os.makedirs(OUT_DIR, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1:  # e.g. `python -m oracle.capture_golden contact geometry`: only the named captures
        for name in sys.argv[1:]:
            globals()["capture_" + name]()
        return
    capture_schedule()
    capture_forward("tiny", O.ARCH_TINY, B=2, T=16, ts=[0, 1, 500, 999])
    capture_forward("tiny_ragged", O.ARCH_TINY, B=3, T=21, ts=[7], nobj=3)
    capture_forward("tiny_nonfinite", O.ARCH_TINY, B=2, T=16, ts=[10], nonfinite=True)
    capture_forward("arch_mdm", O.ARCH_MDM, B=2, T=16, ts=[0, 999])
    capture_forward("arch_mdm_l", O.ARCH_MDM_L, B=2, T=16, ts=[0, 500])
    capture_forward("arch_mdm_l_t196", O.ARCH_MDM_L, B=1, T=196, ts=[250])
    # one p_sample step + short loops with the draws stored (torch-RNG independent)
    capture_loop("tiny_10", O.ARCH_TINY, B=2, T=16, steps=10, store_noise=True, dump_steps=list(range(10)))
    # BASELINE.json configs[0]: arch_mdm, B=4, T=64, 50 DDPM steps (noise from the det recipe)
    capture_loop("arch_mdm_b4_t64_50", O.ARCH_MDM, B=4, T=64, steps=50, store_noise=False)
    # full-length loop on the tiny arch
    capture_loop("tiny_1000", O.ARCH_TINY, B=2, T=16, steps=1000, store_noise=False)
    capture_loop_arch_mdm_l_1000()
    capture_loop_arch_mdm_1000()
    capture_stress()
    capture_stress_dc()
    capture_trained_all()
    capture_respaced()
    capture_refine("tiny_r", O.ARCH_TINY_R, B=2, T=16)
    capture_refine("arch_refine", O.ARCH_REFINE, B=2, T=24)
    capture_geometry()
    capture_vertex_normals()
    capture_contact()
    capture_collate()
    capture_cache_dict()
    capture_siv()


if __name__ == "__main__":
    main()
