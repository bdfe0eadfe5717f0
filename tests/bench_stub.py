"""Stand-in for bench.HipSampler in the CPU test of bench.py's multi-rank plumbing (tests/test_bench_main_gloo.py):
"samples" a clip as the Philox draw 1 of its GLOBAL clip id, so the gathered result is checkable on every rank."""
import numpy as np
import torch


class StubSampler:
    def __init__(self, arch, sd, B, T, N, dtype, dev, tab, use_graph=True):
        self.B, self.T = B, T
        self.cond_calls = 0

    def set_cond(self, cond_dev):
        self.cond_calls += 1

    def sample(self, seed, clip0, out):
        from oracle import mdm_oracle as O

        out.copy_(torch.from_numpy(O.philox_normal(int(seed), np.arange(clip0, clip0 + self.B), 1, 99, self.T)))

    kernels_per_step = 0

    def close(self):
        pass
