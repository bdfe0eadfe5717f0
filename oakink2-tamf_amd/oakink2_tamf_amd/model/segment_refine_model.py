"""SegmentRefineModel trunk - MF-MDM "R" (reference model/segment_refine_model.py:21-250) on the HIP library.

Only the transformer trunk is on the MI355X path: three prefix tokens
(hand side, shape, object), per-frame tokens from [pose | object trajectory | hand->object distance], the same
8-layer post-LN encoder as G, and the residual output x_in + head(...).  The MANO forward kinematics, vertex normals
and the brute-force hand->object signed distance that produce `h2o_dist` in the reference (:107-168, external
manotorch / pytorch3d / chamfer_distance) are "next" rows (section 8f) and are supplied by the caller as
batch["h2o_dist"] (B, T, 778).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .interaction_segment_mdm import HandsideProcess, PositionalEncoding, _HipDenoiserBase, _Linear


class SegmentRefineModel(_HipDenoiserBase):
    kind = "R"

    def __init__(self, mano_path=None, input_dim=99, obj_input_dim=9, hand_shape_dim=10, obj_embed_dim=768,
                 latent_dim=256, ff_size=1024, num_layers=8, num_heads=4, dropout=0.1, activation="gelu", use_pc=False,
                 h2o_dim=778, precision: str = "bf16x3", max_batch=None, max_frames=None):
        super().__init__()
        if activation != "gelu":
            raise NotImplementedError("the HIP FFN kernel fuses the exact erf-GELU (activation='gelu') only")
        self.latent_dim, self.ff_size, self.num_layers, self.num_heads = latent_dim, ff_size, num_layers, num_heads
        self.input_feats, self.obj_input_feats = input_dim, obj_input_dim
        self.use_pc = use_pc
        self.mano_path = mano_path  # kept for signature compatibility; MANO runs on the host side of the caller
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
                            num_heads=num_heads, h2o_dim=h2o_dim), precision, max_batch, max_frames)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        # checkpoints of the reference also carry the manotorch buffers (mano_layer_rh.*, mano_layer_lh.*)
        sd = {k: v for k, v in state_dict.items() if not k.startswith("mano_layer_")}
        return super().load_state_dict(sd, strict=strict, **kw)

    @torch.no_grad()
    def forward(self, batch):
        """batch: "sample_pose_repr" (B, T, 99), "h2o_dist" (B, T, 778), "hand_side", "shape", "obj_embedding",
        "obj_traj"  ->  {"refine_pose_repr": (B, T, 99), "sample_h2o_dist": h2o_dist}."""
        x_in = batch["sample_pose_repr"]
        if "h2o_dist" not in batch:
            raise KeyError("batch['h2o_dist'] (B, T, 778) must be supplied: MANO FK + hand->object distance are outside "
                           "the MI355X hot path (SURVEY.md section 8f, row 1)")
        B, T, _ = x_in.shape
        ctx = self._context(B, T)
        self._set_cond(ctx, batch, None)
        out = ctx.refine(x_in, batch["h2o_dist"])
        return {"refine_pose_repr": out, "sample_h2o_dist": batch["h2o_dist"]}
