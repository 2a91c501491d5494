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


def test_bench_stream_assignment_for_both_scaling_modes():
    """bench.py's own assignment: strong scaling deals the workload's streams round-robin (a partition of the global
    ids, stream s on rank s mod N); weak scaling and per-GPU workloads give every rank the full count with ids that no
    other rank uses."""
    sys.path.insert(0, ROOT)
    import bench
    wl = bench.WORKLOADS["cfg4"]
    for world in (1, 2, 4, 8):
        strong = [bench.stream_ids_of_rank(wl, "strong", r, world) for r in range(world)]
        assert sorted(i for ids in strong for i in ids) == list(range(wl["streams"]))
        assert all(i % world == r for r, ids in enumerate(strong) for i in ids)
        weak = [bench.stream_ids_of_rank(wl, "weak", r, world) for r in range(world)]
        assert all(len(ids) == wl["streams"] for ids in weak)
        flat = [i for ids in weak for i in ids]
        assert len(set(flat)) == len(flat)
    per_gpu = bench.WORKLOADS["cfg5"]
    assert per_gpu.get("per_gpu") and all(len(bench.stream_ids_of_rank(per_gpu, "strong", r, 8)) == per_gpu["streams"] for r in range(8))


def _bench_line(argv, env=None, timeout=300):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_flag_launches_the_ranks_itself():
    """Plain `bench.py --gpus 2` (no launcher around it) starts two ranks as a child torch.distributed.run and relays
    rank 0's line; --dry-run rehearses launch, sharding and the reduction without device work."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res, out = _bench_line(["--gpus", "2", "--dry-run"], env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    assert out["n_gpus"] == 2 and out["dry_run"] and out["value"] is None
    assert out["config"]["streams_total"] == 64 and out["config"]["streams_this_gpu"] == 32
    assert out["process_group"] == {"backend": "gloo", "world_size": 2}
    res, out = _bench_line(["--gpus", "4", "--dry-run", "--workload", "cfg5"], env=env)   # per-GPU workload: weak by definition
    assert res.returncode == 0 and out["n_gpus"] == 4 and out["scaling"] == "weak"
    assert out["config"]["streams_total"] == 4 and out["config"]["streams_this_gpu"] == 1


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """Under a launcher WORLD_SIZE must equal --gpus: a mismatch (the flag ignored, or a wrong launch) exits non-zero
    before any work."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res, out = _bench_line(["--gpus", "2", "--dry-run"], env=env)
    assert res.returncode == 2 and out is None and "WORLD_SIZE=3" in res.stderr
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res, out = _bench_line(["--dry-run"], env=env)     # --gpus defaults to 1
    assert res.returncode == 2 and out is None


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_two_ranks_on_one_gpu(scaling, tmp_path):
    """bench.py --gpus 2, started plain: it launches its two ranks itself (gloo for the reduction of the totals, both ranks
    on GPU 0).  The JSON line names the scaling mode and the total stream count, and the job's rate is the sum over
    ranks divided by the slowest rank's time."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FT_BENCH_BACKEND"] = "gloo"
    res, out = _bench_line(["--gpus", "2", "--steps", "6", "--warmup", "2", "--repeats", "1",
                            "--workload", "cfg2" if scaling == "weak" else "cfg4", "--scaling", scaling], env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["steps"] == 6 and out["warmup"] == 2
    assert out["process_group"] == {"backend": "gloo", "world_size": 2}
    want_total = 64 if scaling == "strong" else 2
    assert out["config"]["streams_total"] == want_total and out["config"]["streams_this_gpu"] == want_total // 2
    assert out["value"] > 0 and abs(out["frames_per_sec"] - want_total * 6 / (out["ms_per_step"] * 6e-3)) < 1e-6 * out["frames_per_sec"]
