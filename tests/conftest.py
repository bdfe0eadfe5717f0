"""pytest configuration: markers, import paths, shared fixture loaders."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_PARENT = os.path.join(ROOT, "oakink2-tamf_amd")
for p in (ROOT, PKG_PARENT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
collect_ignore = ["scripts"]  # measurement / report scripts that use the oracle as checker, not tests


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def golden_cond(fix):
    """Rebuild the oracle/host 'cond' dict from a fixture."""
    import torch

    hs = ["rh" if int(v) == 0 else "lh" for v in fix["cond/hand_side"]]
    return {
        "text_embedding": torch.from_numpy(fix["cond/text_embedding"]),
        "hand_side": hs,
        "shape": torch.from_numpy(fix["cond/shape"]),
        "obj_embedding": torch.from_numpy(fix["cond/obj_embedding"]),
        "obj_traj": torch.from_numpy(fix["cond/obj_traj"]),
    }


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
