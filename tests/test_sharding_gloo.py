"""The N > 1 path on CPU: two processes (gloo) each track their share of the streams, the totals are reduced the way
bench.py reduces them, and the result equals one process tracking every stream."""
import json
import os
import socket
import subprocess
import sys

import pytest

import oracle_lib as O
from feature_tracker_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "helpers", "shard_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_stream_assignment_is_a_partition():
    for total in (1, 5, 8, 64):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += sharding.streams_of_rank(total, world, r)
            assert sorted(seen) == list(range(total))
    ids = [sharding.weak_scaling_stream_ids(64, r) for r in range(8)]
    flat = [i for row in ids for i in row]
    assert len(set(flat)) == len(flat) == 512
    with pytest.raises(ValueError):
        sharding.streams_of_rank(4, 2, 2)


def test_reduce_without_process_group_is_identity():
    dt, counts = sharding.reduce_run_totals(1.5, [3, 4])
    assert dt == 1.5 and counts == [3.0, 4.0]


def test_two_ranks_equal_one_process(tmp_path):
    O.lib()
    total_streams, n_frames, w, h = 4, 6, 188, 120
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), WORKER, str(tmp_path), str(total_streams), str(n_frames), str(w), str(h)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-4000:]
    ranks = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]

    # every stream ran on exactly one rank
    assert sorted(ranks[0]["streams"] + ranks[1]["streams"]) == list(range(total_streams))
    # both ranks hold the same whole-job totals: counters summed, time = the slowest rank
    for r in ranks:
        assert r["tracked_all"] == ranks[0]["tracked"] + ranks[1]["tracked"]
        assert r["attempted_all"] == ranks[0]["attempted"] + ranks[1]["attempted"]
        assert r["dt_max"] == pytest.approx(max(ranks[0]["dt"], ranks[1]["dt"]))
    assert ranks[0]["tracked_all"] > 0

    # one process tracking all streams gives the same counters and the same per-stream state
    sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
    import shard_worker
    tracked, attempted, per_stream = shard_worker.track_streams(list(range(total_streams)), n_frames, w, h)
    assert tracked == ranks[0]["tracked_all"] and attempted == ranks[0]["attempted_all"]
    merged = dict(ranks[0]["per_stream"], **ranks[1]["per_stream"])
    assert merged == {k: list(v) for k, v in per_stream.items()}
