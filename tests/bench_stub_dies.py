"""Test hook of tests/test_bench_main_gloo.py: rank 1 dies while it imports its sampler - before the process-group set-up, where
rank 0 would otherwise wait for it until the rendezvous times out."""
import os
import sys

if os.environ.get("RANK") == "1":
    sys.exit(7)

from bench_stub import StubSampler  # noqa: E402,F401
