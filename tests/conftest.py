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
    # Build whatever is missing or older than its sources (__graft_entry__.build() compares modification times and
    # make does the rest): a no-op when everything is up to date.  On the GPU box, which receives the built tree and has
    # no reason to rebuild, a tree whose artifacts are all present is used as it is.
    have_all = all(os.path.exists(os.path.join(ROOT, a)) for a in ARTIFACTS)
    if have_all and os.environ.get("GRAFT_REPO_ROOT"):
        return
    import __graft_entry__
    __graft_entry__.build()
