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


TRAINED_ARCHS = {  # fixtures whose weights came out of the reference's own training step (oracle/capture_golden.py:capture_trained)
    "trained_tiny": dict(latent_dim=128, ff_size=256, num_layers=2, num_heads=2),
    "trained_hd128": dict(latent_dim=128, ff_size=256, num_layers=2, num_heads=1),  # head dim 128, as arch_mdm_l
}


def trained_arch(name):
    from oracle import mdm_oracle as O

    return O.Arch(**TRAINED_ARCHS[name])


def load_trained_sd(name):
    """state dict of a trained fixture: tests/golden/<name>_weights.npz holds the parameters; the positional tables are buffers
    (never trained) and are regenerated"""
    import torch

    from oracle import mdm_oracle as O

    arch = trained_arch(name)
    with np.load(os.path.join(GOLDEN, f"{name}_weights.npz")) as z:
        sd = {k: torch.from_numpy(z[k].copy()) for k in z.files if not k.startswith("meta/")}
        meta = {k[5:]: z[k] for k in z.files if k.startswith("meta/")}
    pe = O.positional_table(arch.latent_dim).unsqueeze(1).contiguous()
    for k in O.state_dict_spec(arch):
        if k.endswith(".pe"):
            sd[k] = pe
    assert set(sd) == set(O.state_dict_spec(arch))
    return sd, meta


@pytest.fixture
def test_hooks():
    """GPU tests that need include/tamf_hip_test.h (kernel-selection overrides, guard bands, failure injection): for the duration of the
    test, contexts are created through libtamf_hip_hooks.so - the -DTAMF_TEST_HOOKS build of the same sources - and
    hip_backend.lib() is that library.  Every other GPU test runs on libtamf_hip.so, which exports the drop-in surface only."""
    from oakink2_tamf_amd import hip_backend as hb

    hb.use_test_hooks(True)
    try:
        yield hb.hooks()
    finally:
        hb.use_test_hooks(False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
