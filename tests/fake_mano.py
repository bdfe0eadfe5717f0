"""Stand-in for the (licence-gated) MANO layers in tests: a smooth deterministic map with manotorch's call contract."""
from types import SimpleNamespace

import numpy as np
import torch


class FakeManoLayer:
    def __init__(self, sign: float, device):
        g = torch.Generator().manual_seed(3)
        self.Wq = (torch.randn(64, 778 * 3, generator=g) * 0.02 * sign).to(device)
        self.Wb = (torch.randn(10, 778 * 3, generator=g) * 0.01).to(device)

    def __call__(self, pose_coeffs, betas):
        v = (pose_coeffs.reshape(pose_coeffs.shape[0], 64) @ self.Wq.to(pose_coeffs) + betas @ self.Wb.to(pose_coeffs)).reshape(-1, 778, 3)
        return SimpleNamespace(verts=v, joints=v[:, :21])


def make(mano_cfg, device):
    faces = np.arange(1554 * 3, dtype=np.int64).reshape(1554, 3) % 778
    return FakeManoLayer(1.0, device), FakeManoLayer(-1.0, device), faces, faces[:, ::-1].copy()
