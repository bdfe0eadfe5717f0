"""Row 8(f)-3: batch assembly and on-disk formats (host logic, CPU).  The collate fixture is the output of the reference's
own interaction_segment_collate on oracle.fixtures.ragged_clips() (oracle/capture_golden.py: capture_collate)."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.fixtures import ragged_clips
from oakink2_tamf_amd.dataset.batching import interaction_segment_collate, pad_object_axis
from oakink2_tamf_amd.launch import formats


def test_collate_matches_reference():
    fix = load_golden("collate.npz")
    out = interaction_segment_collate(ragged_clips())
    tensors = {k[3:] for k in fix if k.startswith("t__")}
    assert {k for k, v in out.items() if isinstance(v, torch.Tensor)} == tensors
    assert sorted(k for k, v in out.items() if not isinstance(v, torch.Tensor)) == fix["listed_keys"].tolist()
    for k in tensors:
        assert str(out[k].dtype) == str(fix["dtype__" + k]), k
        np.testing.assert_array_equal(out[k].numpy(), fix["t__" + k])
    # object axis zero-padded to the batch maximum (3), clip 0 has one real object
    assert out["obj_traj"].shape == (3, 3, 8, 9) and float(out["obj_traj"][0, 1:].abs().max()) == 0.0
    assert out["hand_side"] == ["rh", "lh", "rh"] and out["obj_list"][1] == ["O1_0", "O1_1", "O1_2"]
    assert list(out.keys()) == list(ragged_clips()[0].keys())


def test_collate_errors_and_padding():
    clips = ragged_clips()
    clips[0]["surprise"] = 1
    with pytest.raises(KeyError, match="unexpected key in batch"):
        interaction_segment_collate(clips)
    with pytest.raises(ValueError):
        interaction_segment_collate([])
    a, b = pad_object_axis([np.ones((1, 2), np.float32), np.ones((3, 2), np.float32)])
    assert a.shape == (3, 2) and a[1:].sum() == 0 and b.sum() == 6 and a.dtype == np.float32
    # a batch of equal object counts is stacked unchanged
    same = interaction_segment_collate([ragged_clips()[2], ragged_clips()[2]])
    assert same["obj_embedding"].shape == (2, 2, 768)


def test_collated_batch_feeds_the_module_contract():
    """the collated dict has the keys the denoiser forward reads (interaction_segment_mdm.py:145-162)"""
    out = interaction_segment_collate(ragged_clips())
    for k in ("hand_side", "shape", "obj_embedding", "obj_traj", "text"):
        assert k in out
    assert out["shape"].shape[0] == out["obj_traj"].shape[0] == len(out["hand_side"])


def test_g_stage_npy_tree(tmp_path):
    ck = formats.ckpt_path("sample", "main", cwd=str(tmp_path))
    assert ck == os.path.join(str(tmp_path), "common", "sample", "main")
    x = np.random.default_rng(0).standard_normal((160, 99))
    p = formats.write_sample_npy(ck, "test/arch_mdm_l__0399", 7, x)
    assert p == os.path.join(ck, "sample", "test/arch_mdm_l__0399", "000007.npy")
    back = np.load(p)
    assert back.dtype == np.float32 and back.shape == (160, 99) and np.array_equal(back, x.astype(np.float32))
    with pytest.raises(ValueError):
        formats.write_sample_npy(ck, "o", 0, np.zeros((2, 3, 4)))


def test_r_stage_save_dict(tmp_path):
    ck = formats.ckpt_path("sample_refine", "main", cwd=str(tmp_path))
    T = 6
    info = ("scene_01/seq__a/b", 12, "rh")
    d = formats.build_refine_save_dict(info, "rh", np.zeros((T, 21, 3), np.float32), np.ones((T, 778, 3), np.float32),
                                       np.zeros((1554, 3), np.int64), ["O02@0001"], 5, list(range(T)), np.zeros((T, 99), np.float32))
    assert tuple(d.keys()) == formats.REFINE_KEYS and d["process_key"] == info[0]
    p = formats.write_refine_sample(ck, "test/arch_mdm_l__0399", d)
    assert p == os.path.join(ck, "sample", "test/arch_mdm_l__0399", "scene_01++seq__a++b", "12", "rh", "save_dict.pkl")
    with open(p, "rb") as f:
        raw = pickle.load(f)  # plain pickle, as the reference's readers expect
    assert set(raw) == set(formats.REFINE_KEYS) and raw["len"] == 5 and raw["verts"].shape == (T, 778, 3)
    assert formats.read_refine_sample(p)["obj_list"] == ["O02@0001"]
    with pytest.raises(ValueError):
        formats.build_refine_save_dict(info, "both", d["joints"], d["verts"], d["faces"], [], 5, [], d["refine_pose_repr"])
    with pytest.raises(ValueError):
        formats.build_refine_save_dict(info, "lh", d["joints"][:2], d["verts"], d["faces"], [], 5, [], d["refine_pose_repr"])
    bad = dict(d); bad.pop("faces")
    with pytest.raises(KeyError):
        formats.write_refine_sample(ck, "o", bad)
