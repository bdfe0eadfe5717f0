"""Row 8(f)-3, input side (host logic, CPU): the segment-cache reader, clip assembly and the G -> R join against item dicts
captured from the reference's own InteractionSegmentData / GeneratedPoseReprSampleAdaptor / segment_slice_from_gap
(oracle/capture_golden.py:capture_cache_dict -> tests/golden/cache_dict_items.npz), bit for bit."""
import json
import os
import pickle

import numpy as np
import pytest

from conftest import load_golden
from oracle import fixtures
from oakink2_tamf_amd.dataset.batching import interaction_segment_collate
from oakink2_tamf_amd.dataset.interaction_segment import (CACHE_KEYS, InteractionSegmentData, check_cache_dict, load_cache_dict,
                                                          rotmat_to_rot6d, transf_to_tslrot6d)
from oakink2_tamf_amd.dataset.pose_repr_sample import GeneratedPoseReprSampleAdaptor, IdentitySampleAdaptor
from oakink2_tamf_amd.dataset.segment_slice import SegmentSlice, segment_slice_from_gap


def _assert_item_equals_fixture(item, fix, prefix):
    assert list(item.keys()) == fix[f"{prefix}/__keys__"].tolist(), prefix  # same keys in the same order
    meta = json.loads(str(fix[f"{prefix}/__meta__"]))
    for k, v in item.items():
        if isinstance(v, np.ndarray):
            ref = fix[f"{prefix}/{k}"]
            assert v.dtype == ref.dtype and v.shape == ref.shape, (prefix, k, v.dtype, ref.dtype, v.shape, ref.shape)
            np.testing.assert_array_equal(v, ref, err_msg=f"{prefix}/{k}")
        elif isinstance(v, list) and v and isinstance(v[0], np.ndarray):
            assert meta[k] == {"__arrays__": len(v)}
            for j, a in enumerate(v):
                ref = fix[f"{prefix}/{k}/{j}"]
                assert a.dtype == ref.dtype
                np.testing.assert_array_equal(a, ref)
        else:
            got = json.loads(json.dumps(v))  # tuples -> lists, as the fixture stored them
            assert got == meta[k], (prefix, k, got, meta[k])


@pytest.fixture(scope="module")
def dataset_files(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("synthetic_dataset"))
    paths, cache = fixtures.write_synthetic_dataset(root)
    return root, paths, cache


def test_items_match_the_reference_dataset_class(dataset_files):
    root, paths, cache = dataset_files
    fix = load_golden("cache_dict_items.npz")
    ds = InteractionSegmentData(process_range_list=["ignored"], data_prefix="/nonexistent", obj_embedding_prefix=paths["emb"],
                                enable_obj_model=True, obj_pointcloud_prefix=paths["pc"], cache_dict=load_cache_dict(paths["cache"]),
                                obj_model_loader=fixtures.synthetic_object_mesh)
    assert len(ds) == 5
    for i in range(len(ds)):
        _assert_item_equals_fixture(ds[i], fix, f"fwd/{i}")
    it = ds[1]
    assert it["pose_repr"].shape == (160, 99) and it["pose_repr"].dtype == np.float32 and it["mask"].sum() == it["len"] == 16
    assert it["obj_list"] == sorted(it["obj_list"]) and it["obj_traj"].shape == (len(it["obj_list"]), 160, 9)
    assert float(np.abs(it["pose_repr"][16:]).max()) == 0.0  # the cache's zero padding survives the conversion


def test_reverse_twins_match_the_reference(dataset_files):
    _, paths, cache = dataset_files
    fix = load_golden("cache_dict_items.npz")
    ds = InteractionSegmentData(obj_embedding_prefix=paths["emb"], cache_dict=cache, append_reverse_segment=True)
    assert len(ds) == 10
    for i in range(5):
        _assert_item_equals_fixture(ds[5 + i], fix, f"rev/{i}")
        fwd, rev = ds[i], ds[5 + i]
        L = fwd["len"]
        assert rev["info"] == fwd["info"] and rev["frame_id"] == fwd["frame_id"][::-1]
        np.testing.assert_array_equal(rev["pose_repr"][:L], fwd["pose_repr"][:L][::-1])
    assert len(cache["interaction_segment_len_list"]) == 5  # the caller's cache dict is not grown in place
    assert set(ds.get_cache()) == set(CACHE_KEYS) and len(ds.get_cache()["interaction_segment_len_list"]) == 10


def test_g_to_r_adaptor_matches_the_reference(dataset_files, tmp_path):
    _, paths, cache = dataset_files
    fix = load_golden("cache_dict_items.npz")
    dirs = [str(tmp_path / "sample" / "b_part"), str(tmp_path / "sample" / "a_part")]
    for d, ids in zip(dirs, [[3, 0, 10], [2, 1]]):
        os.makedirs(d)
        for sid in ids:
            np.save(os.path.join(d, f"{sid:06d}.npy"), fixtures.synthetic_sample_pose_repr(os.path.basename(d), sid))
        open(os.path.join(d, "log.txt"), "w").write("not a sample")  # other files are skipped
    ds = InteractionSegmentData(obj_embedding_prefix=paths["emb"], cache_dict=cache)
    ad = GeneratedPoseReprSampleAdaptor(ds, dirs)
    assert len(ad) == 5 and [ad[i]["sample_info"] for i in range(5)] == [("b_part", 0), ("b_part", 3), ("b_part", 10), ("a_part", 1), ("a_part", 2)]
    for i in range(5):
        _assert_item_equals_fixture(ad[i], fix, f"adaptor/{i}")
    with pytest.raises(AssertionError, match="generated samples"):
        GeneratedPoseReprSampleAdaptor(ds, dirs[:1])
    ident = IdentitySampleAdaptor(ds)[2]
    assert ident["sample_info"] is None and ident["sample_pose_repr"] is ident["pose_repr"]


def test_slicer_matches_the_reference():
    fix = load_golden("cache_dict_items.npz")
    for n in (500, 100, 5000, 1920, 192):
        traj = np.arange(n * 2, dtype=np.float32).reshape(n, 2)
        clips, lens = segment_slice_from_gap(traj, 12, 160, 16)
        np.testing.assert_array_equal(np.stack(clips), fix[f"slice/{n}/clips"])
        assert lens == fix[f"slice/{n}/lens"].tolist()
        assert all(c.shape == (160, 2) and c.dtype == np.float32 for c in clips)
    assert SegmentSlice.from_gap(np.zeros((192, 1)), 12, 160, 16)[1] == [16] * 12
    assert segment_slice_from_gap(np.zeros((15, 1)), 12, 160, 16) == ([], [])  # shorter than one minimum-length clip: gap 0, no clips (as the reference)


def test_cache_dict_errors(tmp_path, dataset_files):
    _, paths, cache = dataset_files
    with pytest.raises(NotImplementedError, match="cache dict"):
        InteractionSegmentData(process_range_list=["scene"], data_prefix="/data")
    broken = {k: v for k, v in cache.items() if k != "interaction_segment_tsl_list"}
    with pytest.raises(KeyError, match="interaction_segment_tsl_list"):
        check_cache_dict(broken)
    ragged = dict(cache, interaction_segment_text_list=cache["interaction_segment_text_list"][:-1])
    with pytest.raises(ValueError, match="different length"):
        InteractionSegmentData(cache_dict=ragged)
    p = tmp_path / "not_a_cache.pkl"
    p.write_bytes(pickle.dumps([1, 2, 3]))
    with pytest.raises(TypeError):
        load_cache_dict(str(p))
    with pytest.raises(FileNotFoundError):  # an object without its embedding file, as in the reference (torch.load raises)
        InteractionSegmentData(obj_embedding_prefix=str(tmp_path), cache_dict=cache)
    # without the prefixes the optional fields are simply absent (:431-442)
    it = InteractionSegmentData(cache_dict=cache)[0]
    assert "obj_embedding" not in it and "obj_pointcloud" not in it and "obj_verts" not in it


def test_rot6d_conventions():
    R = fixtures._det_rotations("t/rot", (7,))
    r6 = rotmat_to_rot6d(R)
    assert r6.shape == (7, 6) and np.array_equal(r6[:, :3], R[:, 0, :]) and np.array_equal(r6[:, 3:], R[:, 1, :])
    T4 = np.zeros((7, 4, 4), np.float32)
    T4[:, :3, :3], T4[:, :3, 3], T4[:, 3, 3] = R, np.arange(21, dtype=np.float32).reshape(7, 3), 1
    t9 = transf_to_tslrot6d(T4)
    assert t9.dtype == np.float32 and np.array_equal(t9[:, :3], T4[:, :3, 3]) and np.array_equal(t9[:, 3:], r6)


def test_items_collate_into_the_module_contract(dataset_files):
    """dataset items -> the reference's collate -> the keys and shapes the denoiser forward reads"""
    _, paths, cache = dataset_files
    ds = InteractionSegmentData(obj_embedding_prefix=paths["emb"], cache_dict=cache)
    b = interaction_segment_collate([ds[i] for i in range(len(ds))])
    assert b["pose_repr"].shape == (5, 160, 99) and b["shape"].shape == (5, 160, 10) and b["mask"].shape == (5, 160)
    assert b["obj_traj"].shape == (5, 3, 160, 9) and b["obj_embedding"].shape == (5, 3, 768) and b["obj_num"].tolist() == [1, 2, 3, 1, 2]
    assert float(b["obj_embedding"][0, 1:].abs().max()) == 0.0 and len(b["text"]) == 5


def test_sample_launcher_reads_the_cache_dict_with_sample_sh_arguments(dataset_files, monkeypatch):
    """script/sample.sh:33-41's literal argument list resolves to the cache-dict clip source (CPU part: parsing, defaults, paths,
    batch assembly; the GPU part is tests/test_hip_pipeline.py)"""
    import shutil

    import torch
    from conftest import ROOT
    from oakink2_tamf_amd.launch import sample as S
    from oakink2_tamf_amd.launch import sample_refine as R

    root, paths, cache = dataset_files
    monkeypatch.chdir(root)
    shutil.copytree(os.path.join(ROOT, "config"), os.path.join(root, "config"), dirs_exist_ok=True)
    argv = ["--cfg", "config/obj_embedding.yml", "--data.process_range", "?(file:./asset/split/test.txt)",
            "--data.cache_dict_filepath", "common/save_cache_dict/main/cache/test.pkl", "--cfg", "config/arch_mdm_l.yml",
            "--debug.model_weight_filepath", "model.pt", "--debug.sample_save_offset", "test/arch_mdm_l__0399",
            "--runtime.device_id", "0,1,2,3", "--commit"]
    known, dotted = S.parse_args(argv)
    cfg = S.build_config(known, dotted)
    assert cfg["data"]["obj_embedding_prefix"] == paths["emb"] and cfg["data"]["cache_dict_filepath"] == paths["cache"]
    assert cfg["data"]["process_range"] == list(dict.fromkeys(i[0] for i in cache["interaction_segment_info_list"]))
    assert cfg["model"]["latent_dim"] == 512 and cfg["debug"]["model_weight_filepath"] == os.path.join(root, "model.pt")
    clips = S.load_clips(cfg, known)
    assert isinstance(clips, S.CacheDictClips) and (clips.n, clips.frames) == (5, 160) and clips.needs_clip
    b = clips.batch(1, 4, "cpu")
    assert b["shape"].shape == (3, 160, 10) and b["obj_traj"].shape == (3, 3, 160, 9) and b["obj_embedding"].dtype == torch.float32
    assert b["hand_side"] == ["lh", "rh", "rh"] and "text_embedding" not in b and len(b["text"]) == 3
    # with the prompt table the batch carries what the module needs without a CLIP tower
    known2, dotted2 = S.parse_args(argv + ["--data.text_embedding_filepath", paths["text"]])
    clips2 = S.load_clips(S.build_config(known2, dotted2), known2)
    b2 = clips2.batch(0, 5, "cpu")
    assert not clips2.needs_clip and b2["text_embedding"].shape == (5, 512)
    np.testing.assert_array_equal(b2["text_embedding"][3].numpy(), fixtures.synthetic_text_embedding(cache["interaction_segment_text_list"][3]))
    # the default cache location is the reference's, relative to the working directory
    known3, dotted3 = S.parse_args(["--cfg", "config/obj_embedding.yml"])
    assert S.load_clips(S.build_config(known3, dotted3), known3).n == 5
    # sample_refine.sh's argument list: data defaults + the G tree of the same offset
    kr, dr = S.parse_args(["--data.process_range", "?(file:./asset/split/test.txt)", "--data.cache_dict_filepath",
                           "common/save_cache_dict/main/cache/test.pkl", "--debug.model_weight_filepath", "r.pt",
                           "--debug.sample_save_offset", "test/arch_mdm_l__0399", "--commit"])
    cr = R.build_config(kr, dr)
    assert cr["data"]["obj_pointcloud_prefix"] == paths["pc"] and cr["data"]["obj_embedding_prefix"] == paths["emb"]
    assert R.sample_dir_list(cr) == [os.path.join(root, "common", "sample", "main", "sample", "test", "arch_mdm_l__0399")]
    with pytest.raises(SystemExit, match="G-stage sample directory"):
        R.load_clips(cr)
    d = R.sample_dir_list(cr)[0]
    os.makedirs(d)
    for i in range(5):
        np.save(os.path.join(d, f"{i:06d}.npy"), fixtures.synthetic_sample_pose_repr("g", i))
    items = R.load_clips(cr)
    assert len(items) == 5 and items[4]["sample_info"] == ("arch_mdm_l__0399", 4) and items[0]["obj_pointcloud"].shape[1:] == (64, 3)
    uniq = list(R.unique_clips(items))
    assert [i for i, _ in uniq] == [0, 1, 2, 4]  # segment 3 repeats segment 2's info
    assert [len(g) for g in R.batches(uniq, 3)] == [3, 1]
    # sizing the context and splitting the workers does not materialise the items: the adaptor's arrays / the cache's length
    class _NoItems(type(items)):
        def __getitem__(self, i):
            raise AssertionError("max_sample_frames must not build items")
    probe = object.__new__(_NoItems)
    probe.__dict__.update(items.__dict__)
    assert R.max_sample_frames(probe) == 160
    assert R.max_sample_frames([{"sample_pose_repr": np.zeros((7, 99))}, {"sample_pose_repr": np.zeros((12, 99))}]) == 12  # (--data.clips_pkl lists)
    built = []
    monkeypatch.setattr(S, "CacheDictClips", lambda cfg_: built.append(cfg_) or (_ for _ in ()).throw(AssertionError("count_clips built the clip source")))
    assert S.count_clips(cfg, known) == 5 and not built
