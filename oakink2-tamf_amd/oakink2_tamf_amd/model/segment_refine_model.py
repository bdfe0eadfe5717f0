"""SegmentRefineModel trunk - MF-MDM "R" (reference model/segment_refine_model.py:21-250) on the HIP library.

Only the transformer trunk is on the MI355X path: three prefix tokens
(hand side, shape, object), per-frame tokens from [pose | object trajectory | hand->object distance], the same
8-layer post-LN encoder as G, and the residual output x_in + head(...).  The MANO forward kinematics, vertex normals
and the brute-force hand->object signed distance that produce `h2o_dist` in the reference (:107-168, external
manotorch / pytorch3d / chamfer_distance) are "next" rows (section 8f): either the caller supplies
batch["h2o_dist"] (B, T, 778), or it hands the module its MANO layers (`mano_layer_rh` / `mano_layer_lh`, any callable with
manotorch's `layer(pose_coeffs=quat (T,16,4), betas=(T,10)) -> .verts (T,778,3), .joints (T,21,3)` contract; the MANO assets
are licence-gated and not part of this package) and the module runs the reference's whole forward on the GPU: HIP pose decode
-> MANO -> HIP vertex normals -> HIP hand->object distance -> HIP trunk.  The vertex normals (:131-133, pytorch3d's
verts_normals_packed) are computed when the MANO layer exposes its faces (`th_faces`, as manotorch's does) and returned in
the result dict like the reference's; they do not enter the distance feature (:165 keeps x2y, unsigned because y_normals
is None).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .interaction_segment_mdm import HandsideProcess, PositionalEncoding, _default_precision, _HipDenoiserBase, _Linear


class SegmentRefineModel(_HipDenoiserBase):
    kind = "R"

    def __init__(self, mano_path=None, input_dim=99, obj_input_dim=9, hand_shape_dim=10, obj_embed_dim=768,
                 latent_dim=256, ff_size=1024, num_layers=8, num_heads=4, dropout=0.1, activation="gelu", use_pc=False,
                 h2o_dim=778, precision=None, max_batch=None, max_frames=None, mano_layer_rh=None,
                 mano_layer_lh=None, range_check: str = "fallback", per_clip_object_mean: bool = False):
        super().__init__()
        if activation != "gelu":
            raise NotImplementedError("the HIP FFN kernel fuses the exact erf-GELU (activation='gelu') only")
        self.latent_dim, self.ff_size, self.num_layers, self.num_heads = latent_dim, ff_size, num_layers, num_heads
        self.input_feats, self.obj_input_feats = input_dim, obj_input_dim
        self.use_pc = use_pc
        self.mano_path = mano_path  # kept for signature compatibility; the MANO layers themselves are handed in
        # not registered as submodules: their buffers are not part of this model's state dict (load_state_dict drops mano_layer_*)
        self.__dict__["mano_layer_rh"], self.__dict__["mano_layer_lh"] = mano_layer_rh, mano_layer_lh
        self.hand_side_process = HandsideProcess(latent_dim)
        self.hand_shape_process = _Linear("shape_embed", hand_shape_dim, latent_dim)
        self.obj_embed_process = _Linear("embedding", obj_embed_dim, latent_dim)
        self.input_process = _Linear("poseEmbedding", input_dim, latent_dim)
        self.obj_input_process = _Linear("poseEmbedding", obj_input_dim, latent_dim)
        self.h2o_dist_input_process = _Linear("poseEmbedding", h2o_dim, latent_dim)
        self.input_merge = nn.Sequential(nn.Linear(latent_dim * 3, latent_dim), nn.SiLU(), nn.Linear(latent_dim, latent_dim))
        self.sequence_pos_encoder = PositionalEncoding(latent_dim, dropout)
        layer = nn.TransformerEncoderLayer(d_model=latent_dim, nhead=num_heads, dim_feedforward=ff_size, dropout=dropout,
                                           activation=activation)
        self.seqTransEncoder = nn.TransformerEncoder(layer, num_layers=num_layers, enable_nested_tensor=False)
        self.output_process = _Linear("poseFinal", latent_dim, input_dim)
        for p in self.parameters():
            p.requires_grad_(False)
        self.eval()
        self._init_hip(dict(input_dim=input_dim, obj_input_dim=obj_input_dim, hand_shape_dim=hand_shape_dim,
                            obj_embed_dim=obj_embed_dim, latent_dim=latent_dim, ff_size=ff_size, num_layers=num_layers,
                            num_heads=num_heads, h2o_dim=h2o_dim), precision or _default_precision(), max_batch, max_frames,
                       range_check, per_clip_object_mean)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        # checkpoints of the reference also carry the manotorch buffers (mano_layer_rh.*, mano_layer_lh.*)
        sd = {k: v for k, v in state_dict.items() if not k.startswith("mano_layer_")}
        return super().load_state_dict(sd, strict=strict, **kw)

    def retrieve_hand_faces(self, hand_side):
        """reference :99-105: the triangle list of the side's MANO layer (None when the stand-in layer has no `th_faces`)"""
        if hand_side not in ("rh", "lh"):
            raise ValueError(f"unexpected hand_side: {hand_side}")
        return getattr(self.mano_layer_rh if hand_side == "rh" else self.mano_layer_lh, "th_faces", None)

    # ---- reference :107-140 ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def batch_recover_mano_from_pose_repr(self, batch_pose_repr, batch_shape, batch_hand_side):
        """(B,T,99), (B,T,10), list of "rh"/"lh" -> hand_verts (B,T,778,3), hand_joints (B,T,21,3) (wrist translation added),
        hand_normals (B,T,778,3) or None when a layer does not expose its faces"""
        from ..geometry import pose_repr_to_quat, vertex_normals

        tsl, quat = pose_repr_to_quat(batch_pose_repr)  # HIP: rot6d -> rotmat -> quaternion
        verts, joints, normals = [], [], []
        for b, side in enumerate(batch_hand_side):
            if side not in ("rh", "lh"):
                raise ValueError(f"unexpected hand_side: {side}")
            layer = self.mano_layer_rh if side == "rh" else self.mano_layer_lh
            if layer is None:
                raise RuntimeError(f"no MANO layer for hand side {side!r}: pass mano_layer_rh / mano_layer_lh or supply batch['h2o_dist']")
            out = layer(pose_coeffs=quat[b], betas=batch_shape[b].to(quat))
            verts.append(out.verts + tsl[b].unsqueeze(1))
            joints.append(out.joints + tsl[b].unsqueeze(1))
            faces = self.retrieve_hand_faces(side)
            normals.append(vertex_normals(verts[-1], faces) if faces is not None else None)  # HIP: :131-133
        hand_normals = torch.stack(normals, dim=0) if all(n is not None for n in normals) else None
        return torch.stack(verts, dim=0), torch.stack(joints, dim=0), hand_normals

    @staticmethod
    def _pad_object_points(obj_points_list, device):
        """list over clips of (nobj_b, P, 3) arrays -> (B, nobj_max, P, 3) float32 tensor, zero padded, + per-clip counts"""
        import numpy as np

        arrs = [np.asarray(a, dtype=np.float32) for a in obj_points_list]
        n_max, P = max(a.shape[0] for a in arrs), arrs[0].shape[1]
        out = np.zeros((len(arrs), n_max, P, 3), np.float32)
        for b, a in enumerate(arrs):
            if a.shape[1] != P:
                raise ValueError("object point clouds of one batch must have the same number of points")
            out[b, : a.shape[0]] = a
        return torch.from_numpy(out).to(device), [a.shape[0] for a in arrs]

    @torch.no_grad()
    def multi_object_h2o_dist(self, batch_hand_verts, batch_obj_list, batch_obj_traj, batch_obj_verts_list):
        """reference :142-168 (hand normals dropped, see module docstring) on the HIP kernel"""
        from ..geometry import multi_object_h2o_dist

        pts, _ = self._pad_object_points(batch_obj_verts_list, batch_hand_verts.device)
        n_traj = batch_obj_traj.shape[1]
        if pts.shape[1] < n_traj:  # trajectories are padded to the batch maximum by the collate, point lists are not
            pts = torch.cat([pts, pts.new_zeros(pts.shape[0], n_traj - pts.shape[1], pts.shape[2], 3)], dim=1)
        return multi_object_h2o_dist(batch_hand_verts, batch_obj_traj[:, : pts.shape[1]], pts, [len(o) for o in batch_obj_list])

    @torch.no_grad()
    def forward(self, batch, with_refined_geometry: bool = False):
        """batch: "sample_pose_repr" (B, T, 99), "hand_side", "shape", "obj_embedding", "obj_traj" and either "h2o_dist"
        (B, T, 778) or - with MANO layers - "obj_list" + "obj_pointcloud"/"obj_verts"  ->  {"refine_pose_repr": (B, T, 99),
        "sample_h2o_dist", ["sample_hand_verts", "sample_hand_joints"], [refine_* when with_refined_geometry]}."""
        x_in = batch["sample_pose_repr"]
        res = {}
        if "h2o_dist" in batch:
            h2o = batch["h2o_dist"]
        elif self.mano_layer_rh is not None or self.mano_layer_lh is not None:
            obj_pts = batch["obj_pointcloud"] if self.use_pc else batch["obj_verts"]
            hv, hj, hn = self.batch_recover_mano_from_pose_repr(x_in, batch["shape"], batch["hand_side"])
            h2o = self.multi_object_h2o_dist(hv, batch["obj_list"], batch["obj_traj"], obj_pts)
            res["sample_hand_verts"], res["sample_hand_joints"] = hv, hj
            if hn is not None:
                res["sample_hand_normals"] = hn
        else:
            raise KeyError("batch['h2o_dist'] (B, T, 778) must be supplied, or the module built with mano_layer_rh / mano_layer_lh: "
                           "MANO FK is outside this package (SURVEY.md section 8f, row 1)")
        B, T, _ = x_in.shape
        while True:  # (repeated once in f32 when an activation left the fp16 range of the default f16x3 mode)
            ctx = self._context(B, T)
            self._set_cond(ctx, batch, None)
            out = ctx.refine(x_in, h2o)
            if not self._range_tripped(ctx):
                break
        res["refine_pose_repr"], res["sample_h2o_dist"] = out, h2o
        if with_refined_geometry and "h2o_dist" not in batch:
            rv, rj, rn = self.batch_recover_mano_from_pose_repr(out, batch["shape"], batch["hand_side"])
            res["refine_hand_verts"], res["refine_hand_joints"] = rv, rj
            if rn is not None:
                res["refine_hand_normals"] = rn
            res["refine_h2o_dist"] = self.multi_object_h2o_dist(rv, batch["obj_list"], batch["obj_traj"], obj_pts)
        return res
