"""CPU: host-side mirror of the reference interface - schedule factory, checkpoint key set, CLI/config surface,
C-ABI symbol export - no GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import mdm_oracle as O


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # (comments name functions too)
    return set(re.findall(r"\b(tamf_[a-z0-9_]+)\s*\(", hdr))


def test_cabi_exports_every_declared_symbol():
    """libtamf_hip.so exports exactly what include/tamf_hip.h declares - the drop-in surface and nothing else (VERDICT r5 #7e: the test
    hooks, kernel benchmarks and the process-global tuning word used to be exported from the product library)"""
    from oakink2_tamf_amd import _lib

    lib = _lib.load()
    declared = _declared("tamf_hip.h")
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS)
    for sym in declared:
        assert isinstance(getattr(lib, sym), ctypes._CFuncPtr)
    for sym in _lib.HOOK_EXPORTS:
        assert not hasattr(lib, sym), f"{sym} is a test hook: it must not be exported by libtamf_hip.so"
    import subprocess

    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        exported = {l.split()[-1] for l in nm.stdout.splitlines() if " T " in l and l.split()[-1].startswith("tamf_")}
        assert exported == declared, exported ^ declared


def test_hooks_library_exports_the_test_header_too():
    """libtamf_hip_hooks.so = the same sources with -DTAMF_TEST_HOOKS: everything of tamf_hip.h plus everything of tamf_hip_test.h"""
    from oakink2_tamf_amd import _lib

    hooks = _lib.load_hooks()
    declared = _declared("tamf_hip_test.h")
    assert declared == set(_lib.HOOK_EXPORTS), declared ^ set(_lib.HOOK_EXPORTS)
    for sym in list(declared) + _lib.EXPORTS:
        assert isinstance(getattr(hooks, sym), ctypes._CFuncPtr)


def test_no_gpu_fails_loudly():
    from oakink2_tamf_amd.hip_backend import TamfContext, TamfError

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(TamfError, match="no CPU fallback"):
        TamfContext(dict(latent_dim=128, ff_size=256, num_layers=2, num_heads=2), 1, 8)


@pytest.mark.parametrize("n", [1000, 50])
def test_schedule_factory_matches_reference_tables(n):
    from oakink2_tamf_amd.model.diffusion_util import create_gaussian_diffusion

    fix = load_golden("schedule.npz")
    dif = create_gaussian_diffusion(diffusion_steps=n, noise_schedule="cosine")
    assert dif.num_timesteps == n and dif.timestep_map == list(range(n))
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "posterior_variance", "posterior_log_variance_clipped",
              "posterior_mean_coef1", "posterior_mean_coef2", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
        np.testing.assert_allclose(getattr(dif, k), fix[f"n{n}/{k}"], rtol=0, atol=1e-15)


def test_unsupported_sampler_features_raise():
    from oakink2_tamf_amd.model.diffusion_util import create_gaussian_diffusion

    dif = create_gaussian_diffusion(10, "cosine")
    with pytest.raises(NotImplementedError):
        dif.ddim_sample_loop()
    with pytest.raises(NotImplementedError):
        dif.training_losses()
    with pytest.raises(NotImplementedError):
        create_gaussian_diffusion(10, "cosine", sigma_small=False)
    with pytest.raises(NotImplementedError):
        create_gaussian_diffusion(10, "sigmoid")


def test_generic_sampler_path_with_plain_torch_model():
    """The per-step path (clip_denoised etc.) is plain torch and must reproduce the oracle's DDPM arithmetic."""
    from oakink2_tamf_amd.model.diffusion_util import create_gaussian_diffusion

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(0.5))

        def forward(self, x, t, **kw):
            return self.w * x + t.view(-1, 1, 1, 1).float() * 1e-3

    torch.manual_seed(0)
    dif = create_gaussian_diffusion(12, "cosine")
    shape = (2, 99, 1, 8)
    out = dif.p_sample_loop(Toy(), shape, clip_denoised=False, device=torch.device("cpu"))
    torch.manual_seed(0)
    tab = O.make_tables(12, "cosine")
    x = torch.randn(*shape)
    m = Toy()
    with torch.no_grad():
        for i in range(11, -1, -1):
            x0 = m(x, torch.full((2,), i))
            x = O.ddpm_step(tab, x, x0, i, torch.randn_like(x))
    np.testing.assert_allclose(out.detach().numpy(), x.numpy(), rtol=0, atol=1e-6)
    dumps = dif.p_sample_loop(Toy(), shape, clip_denoised=True, device=torch.device("cpu"), dump_steps=[0, 11])
    assert len(dumps) == 2 and dumps[0].shape == shape


@pytest.mark.parametrize("arch", [O.ARCH_MDM, O.ARCH_MDM_L, O.ARCH_TINY])
def test_module_state_dict_keys_match_checkpoint_format(arch):
    from oakink2_tamf_amd.model.interaction_segment_mdm import InterationSegmentMDM

    m = InterationSegmentMDM(latent_dim=arch.latent_dim, ff_size=arch.ff_size, num_layers=arch.num_layers, num_heads=arch.num_heads)
    spec = O.state_dict_spec(arch)
    sd = m.state_dict()
    assert set(sd) == set(spec)
    for k, shp in spec.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    # the reference checkpoint (no clip_model.*) loads strictly; the positional table equals the oracle's
    m.load_state_dict(O.det_state_dict(arch))
    np.testing.assert_array_equal(sd["sequence_pos_encoder.pe"][:, 0].numpy(), O.positional_table(arch.latent_dim).numpy())
    assert m.eval() is m


def test_refine_module_state_dict_keys():
    from oakink2_tamf_amd.model.segment_refine_model import SegmentRefineModel

    m = SegmentRefineModel(None)
    spec = O.state_dict_spec(O.ARCH_REFINE)
    assert set(m.state_dict()) == set(spec)
    sd = O.det_state_dict(O.ARCH_REFINE)
    sd["mano_layer_rh.th_faces"] = torch.zeros(3)
    m.load_state_dict(sd)  # manotorch buffers of reference checkpoints are dropped


def test_refine_module_mano_layers_are_not_state():
    """MANO layers handed to the module stay outside its state dict (reference checkpoints carry mano_layer_* buffers that are
    dropped on load); object point lists are zero padded on the object axis like the collate pads the trajectories"""
    from oakink2_tamf_amd.model.segment_refine_model import SegmentRefineModel

    class FakeMano(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.register_buffer("th_faces", torch.zeros(4, 3))

    m = SegmentRefineModel(None, mano_layer_rh=FakeMano(), mano_layer_lh=FakeMano())
    assert set(m.state_dict()) == set(O.state_dict_spec(O.ARCH_REFINE)) and m.mano_layer_rh is not None
    pts, counts = SegmentRefineModel._pad_object_points([np.ones((1, 5, 3)), np.ones((3, 5, 3), np.float64)], "cpu")
    assert pts.shape == (2, 3, 5, 3) and pts.dtype == torch.float32 and counts == [1, 3]
    assert float(pts[0, 1:].abs().sum()) == 0.0 and float(pts[1].sum()) == 45.0
    with pytest.raises(ValueError):
        SegmentRefineModel._pad_object_points([np.ones((1, 5, 3)), np.ones((1, 6, 3))], "cpu")


def test_cli_config_surface(tmp_path):
    from oakink2_tamf_amd.launch import sample as S

    argv = ["--cfg", os.path.join(ROOT, "config", "arch_mdm_l.yml"), "--debug.model_weight_filepath", "w.pt",
            "--debug.sample_save_offset", "test/arch_mdm_l__0399", "--runtime.device_id", "0,1,2,3", "--model.num_layers", "4",
            "--commit"]
    known, dotted = S.parse_args(argv)
    cfg = S.build_config(known, dotted)
    assert cfg["model"]["latent_dim"] == 512 and cfg["model"]["ff_size"] == 2048 and cfg["model"]["num_layers"] == 4
    assert cfg["runtime"]["device_id"] == [0, 1, 2, 3] and cfg["commit"] is True
    assert cfg["ckpt_path"].endswith(os.path.join("common", "sample", "main"))
    assert cfg["debug"]["sample_save_offset"] == "test/arch_mdm_l__0399"
    known, dotted = S.parse_args(["--synthetic", "3,16"])
    clips = S.load_clips(S.build_config(known, dotted), known)
    assert clips.cond["obj_traj"].shape == (3, 2, 16, 9) and clips.cond["shape"].shape == (3, 16, 10) and (clips.n, clips.frames) == (3, 16)
    monkey_cwd = os.getcwd()
    os.chdir(tmp_path)  # no segment cache at the reference's default place under the working directory
    try:
        with pytest.raises(SystemExit, match="segment cache"):
            S.load_clips(S.build_config(*S.parse_args([])), S.parse_args([])[0])
    finally:
        os.chdir(monkey_cwd)
    # a dotted flag the launcher does not register is an error, as with the reference's config_reg (not silently dropped)
    for bad in (["--runtime.devce_id", "0"], ["--data.cache_dict", "x.pkl"], ["--model.latent_dim", "many"], ["--data.cond_npz"]):
        with pytest.raises(SystemExit):
            S.parse_args(bad)
    known, dotted = S.parse_args(["--runtime.batch_size=8", "--data.process_range", "?(file:./asset/split/test.txt):scene_a"])
    assert dotted["runtime.batch_size"] == 8 and dotted["data.process_range"] == ["scene_a"]  # (the split file does not exist: no entries)
    assert S.split_outside_macros("a:?(file:x:y.txt),b") == ["a", "?(file:x:y.txt)", "b"]


def test_refine_oracle_matches_reference():
    for name, arch in (("tiny_r", O.ARCH_TINY_R), ("arch_refine", O.ARCH_REFINE)):
        fix = load_golden(f"refine_{name}.npz")
        sd = O.det_state_dict(arch, tag=f"{name}/w")
        cond = {"hand_side": ["rh" if int(v) == 0 else "lh" for v in fix["cond/hand_side"]],
                "shape": torch.from_numpy(fix["cond/shape"]), "obj_embedding": torch.from_numpy(fix["cond/obj_embedding"]),
                "obj_traj": torch.from_numpy(fix["cond/obj_traj"])}
        out = O.refine_forward(sd, arch, torch.from_numpy(fix["x_in"]), torch.from_numpy(fix["h2o"]), cond)
        np.testing.assert_allclose(out.numpy(), fix["out"], rtol=0, atol=1e-5)


def test_file_macro_and_process_range(tmp_path):
    """`?(file:<path>)` list entries expand to the file's stripped lines, missing files to nothing, duplicates are dropped
    keeping first occurrences (reference dev_fn/upkeep/config.py:26-72); --data.process_range selects the clips of a
    conditioning file by their process keys (reference launch/sample.py:161-166 walks the dataset by them)."""
    from oakink2_tamf_amd.launch import sample as S
    from oakink2_tamf_amd.launch.upkeep import decode_file_macro

    lst = tmp_path / "keys.txt"
    lst.write_text("  scene_b  \nscene_c\nscene_a\n")
    assert decode_file_macro(["scene_a", f"?(file:{lst})", "?(file:/nonexistent/x.txt)", "scene_d", "?(other:cmd)"]) == [
        "scene_a", "scene_b", "scene_c", "scene_d"]
    assert decode_file_macro(None) is None
    B, T = 5, 8
    rng = np.random.default_rng(0)
    npz = tmp_path / "cond.npz"
    np.savez(npz, text_embedding=rng.standard_normal((B, 512)).astype(np.float32), hand_side=np.array(["rh", "lh", "rh", "lh", "rh"]),
             shape=rng.standard_normal((B, T, 10)).astype(np.float32), obj_embedding=rng.standard_normal((B, 2, 768)).astype(np.float32),
             obj_traj=rng.standard_normal((B, 2, T, 9)).astype(np.float32),
             process_key=np.array(["scene_a", "scene_x", "scene_c", "scene_y", "scene_b"]))
    known, dotted = S.parse_args(["--data.cond_npz", str(npz), "--data.process_range", f"scene_c,?(file:{lst})"])
    cfg = S.build_config(known, dotted)
    assert cfg["data"]["process_range"] == ["scene_c", "scene_b", "scene_a"]
    cond = S.load_clips(cfg, known).cond
    assert cond["shape"].shape[0] == 3 and list(cond["hand_side"]) == ["rh", "rh", "rh"]  # clips 0, 2, 4 in file order
    # defaults: one worker per visible device, decided in main() (not the reference's 8 workers on devices 0-3)
    assert cfg["runtime"]["num_worker"] is None and cfg["runtime"]["device_id"] is None
    assert known.precision == "f16x3"  # the one package-wide default precision


def test_ckpt_setup_log_file(tmp_path, monkeypatch):
    """commit mode creates common/<prog>/<exp_id>/ with log.txt and opt.yml; a dry run writes nothing (upkeep/ckpt.py:110-149)"""
    import logging

    from oakink2_tamf_amd.launch import sample as S
    from oakink2_tamf_amd.launch.upkeep import ckpt_opt, ckpt_setup

    monkeypatch.chdir(tmp_path)
    cfg = S.build_config(*S.parse_args(["--synthetic", "2,8", "--exp_id", "e1"]))
    ckpt_setup(cfg, argv=["--synthetic", "2,8"])
    ckpt_opt(cfg)
    assert not os.path.exists(tmp_path / "common")
    cfg = S.build_config(*S.parse_args(["--synthetic", "2,8", "--exp_id", "e1", "--commit"]))
    logging.getLogger().setLevel(logging.INFO)
    ckpt_setup(cfg, argv=["--synthetic", "2,8", "--commit"])
    ckpt_opt(cfg)
    for h in list(logging.getLogger().handlers):
        if isinstance(h, logging.FileHandler):
            h.flush()
            logging.getLogger().removeHandler(h)
    d = tmp_path / "common" / "sample" / "e1"
    text = (d / "log.txt").read_text()
    assert "commit mode: setup ckpt" in text and "cmd:" in text and "--synthetic 2,8" in text
    assert (d / "opt.yml").exists()
