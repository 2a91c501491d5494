import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# everything the tests load; built in-tree by __graft_entry__.build() (the files are not kept in the history)
ARTIFACTS = [
    "feature_tracker_amd/_lib/libft_device.so", "feature_tracker_amd/_lib/libft_tracker.so", "feature_tracker_amd/_lib/libft_synth.so",
    "feature_tracker_amd/_lib/test_tracking", "oracle/libft_oracle.so", "oracle/libft_oracle_tracker.so", "oracle/test_tracking_checker",
]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_sessionstart(session):
    # a fresh checkout has no built libraries: build them once (a no-op when they are all there, as on the GPU box,
    # which receives the built tree)
    if all(os.path.exists(os.path.join(ROOT, a)) for a in ARTIFACTS):
        return
    import __graft_entry__
    __graft_entry__.build()
