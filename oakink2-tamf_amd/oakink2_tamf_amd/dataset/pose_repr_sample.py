"""G-stage samples joined to their dataset items: the input of the R stage (reference dataset/pose_repr_sample.py:18-52,
97-108).

`launch/sample.py:234-237` leaves one `<sample_id:06d>.npy` (T, 99) float32 per dataset index under
`<cwd>/common/sample/<exp_id>/sample/<offset>/`; `launch/sample_refine.py:168-171` wraps the dataset in
`GeneratedPoseReprSampleAdaptor(dataset, [<that directory>, ...])`, whose item i is dataset item i plus

    sample_info       (basename of the directory, sample_id)
    sample_pose_repr  the array of the i-th .npy file, files taken directory by directory in sorted name order

The join is positional: the number of .npy files must equal the number of dataset items (the reference asserts it)."""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple

import numpy as np


def list_sample_files(dir_list: Sequence[str]) -> List[Tuple[Tuple[str, int], str]]:
    """[((dir basename, sample id), path)] in the adaptor's order"""
    out = []
    for d in dir_list:
        base = os.path.basename(d)
        for name in sorted(n for n in os.listdir(d) if os.path.splitext(n)[-1] == ".npy"):
            out.append(((base, int(os.path.splitext(name)[0])), os.path.join(d, name)))
    return out


class GeneratedPoseReprSampleAdaptor:
    def __init__(self, interaction_segment_dataset, dir_list: Sequence[str]):
        self.interaction_segment_dataset = interaction_segment_dataset
        self.dir_list = list(dir_list)
        files = list_sample_files(self.dir_list)
        if len(files) != len(interaction_segment_dataset):
            raise AssertionError(f"{len(files)} generated samples under {self.dir_list} for {len(interaction_segment_dataset)} "
                                 "dataset items: the G stage must have sampled every clip of the cache dict")
        self.pose_repr_info_list = [info for info, _ in files]
        self.pose_repr_map = {info: np.load(path) for info, path in files}
        self.len = len(files)

    def __getitem__(self, index: int) -> Dict:
        item = self.interaction_segment_dataset[index]
        info = self.pose_repr_info_list[index]
        item["sample_info"] = info
        item["sample_pose_repr"] = self.pose_repr_map[info]
        return item

    def __len__(self) -> int:
        return self.len


class IdentitySampleAdaptor:
    """item i with its own ground-truth pose as the "sample" (:97-108; the R stage's sanity input)"""

    def __init__(self, interaction_segment_dataset):
        self.interaction_segment_dataset = interaction_segment_dataset
        self.len = len(interaction_segment_dataset)

    def __getitem__(self, index: int) -> Dict:
        item = self.interaction_segment_dataset[index]
        item["sample_info"] = None
        item["sample_pose_repr"] = item["pose_repr"]
        return item

    def __len__(self) -> int:
        return self.len
