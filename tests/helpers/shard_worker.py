"""Worker of the world_size-2 sharding test: tracks this rank's streams on the CPU checker backend, reduces the run
totals with the same helper bench.py uses, and writes what it saw to a JSON file."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch.distributed as dist

from feature_tracker_amd import sharding, synth, tracker

sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

ORC = os.path.join(ROOT, "oracle", "libft_oracle_tracker.so")


def track_streams(stream_ids, n_frames, w, h):
    seqs = [synth.Sequence(w, h, n_frames, stream_id=s) for s in stream_ids]
    b = tracker.BatchTracker(len(seqs), w, h, lib=O.checker_tracker_lib(), n_groups=1, host_threads=1, num_pts=60, win_size=15, pyr_levels=2, use_gyro_seed=1)
    for s, q in enumerate(seqs):
        b.set_calibration(s, q.calib8, q.T_cam_imu)
        for t, g, a in zip(q.imu_t, q.gyro, q.accel):
            b.feed_imu(s, t, a, g)
    tracked = attempted = 0
    per_stream = {}
    for i in range(n_frames):
        tr, at = b.feed_host([q.timestamps[i] for q in seqs], [q.frames[i] for q in seqs])
        tracked += tr
        attempted += at
    for s, sid in enumerate(stream_ids):
        xy, ids = b.points(s)
        per_stream[str(sid)] = [int(ids.sum()), float(xy.sum())]
    return tracked, attempted, per_stream


def main():
    out_dir, total_streams, n_frames, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = sharding.streams_of_rank(total_streams, world, rank)
    t0 = time.perf_counter()
    tracked, attempted, per_stream = track_streams(mine, n_frames, w, h)
    dt = time.perf_counter() - t0 + rank      # ranks report different times: the reduction must take the largest
    dist.barrier()
    dt_max, (tracked_all, attempted_all) = sharding.reduce_run_totals(dt, [tracked, attempted])
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(dict(rank=rank, world=world, streams=mine, dt=dt, dt_max=dt_max, tracked=tracked, attempted=attempted,
                       tracked_all=tracked_all, attempted_all=attempted_all, per_stream=per_stream), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
