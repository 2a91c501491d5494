"""Longer GPU-vs-CPU-checker runs than the test suite affords: several option sets, many frames, every frame compared
(tracked counts, ids, positions).  Needs a GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth as tsynth
from feature_tracker_amd import synth, tracker
import oracle_lib as O
ORC = os.path.join(ROOT, "oracle", "libft_oracle_tracker.so")
N_FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 80
CASES = [
    ("cfg4-like", 752, 480, dict(win_size=21, use_gyro_seed=1), False),
    ("win15 no gyro", 752, 480, dict(win_size=15), False),
    ("fb + reject, moving patches", 752, 480, dict(win_size=21, use_gyro_seed=1, forward_backward=1, fb_thresh=1.0, reject_outliers=1), True),
    ("720p 2000 pts L4", 1280, 720, dict(win_size=21, pyr_levels=4, num_pts=2000, grid_x=9, grid_y=5, use_gyro_seed=1), False),
    ("small frames", 188, 120, dict(win_size=9, pyr_levels=2, num_pts=60, grid_x=3, grid_y=2, min_px_dist=5), False),
]
bad = 0
for name, w, h, opts, patches in CASES:
    seqs = [synth.Sequence(w, h, N_FRAMES, stream_id=60 + s) for s in range(3)]
    if patches:
        for s, q in enumerate(seqs):
            q.frames = tsynth.with_moving_patches(q.frames, seed=s, size=64)
    runs = []
    for lib in (None, O.checker_tracker_lib()):
        b = tracker.BatchTracker(3, w, h, lib=lib, **opts)
        for s, q in enumerate(seqs):
            b.set_calibration(s, q.calib8, q.T_cam_imu)
            for t, g, a in zip(q.imu_t, q.gyro, q.accel):
                b.feed_imu(s, t, a, g)
        hist = []
        t0 = time.perf_counter()
        for i in range(N_FRAMES):
            tr = b.feed_host([q.timestamps[i] for q in seqs], [q.frames[i] for q in seqs])
            hist.append((tr, [b.points(s) for s in range(3)]))
        runs.append((hist, time.perf_counter() - t0))
    first_diff, max_d, tracked = None, 0.0, 0
    for i, ((tr_a, pts_a), (tr_b, pts_b)) in enumerate(zip(runs[0][0], runs[1][0])):
        same = tr_a == tr_b
        for (xy_a, id_a), (xy_b, id_b) in zip(pts_a, pts_b):
            if not np.array_equal(id_a, id_b):
                same = False
            else:
                max_d = max(max_d, float(np.abs(xy_a - xy_b).max(initial=0)))
        tracked += tr_a[0]
        if not same and first_diff is None:
            first_diff = i
    print(f"{name:32s} frames {N_FRAMES} tracked {tracked:7d} max |dxy| {max_d:.2e} first differing frame {first_diff} "
          f"(gpu {runs[0][1]:.2f} s, cpu {runs[1][1]:.2f} s)", flush=True)
    bad += first_diff is not None or max_d > 0.01
sys.exit(1 if bad else 0)
