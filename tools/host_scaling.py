"""Host bookkeeping scaling over stream groups and helper threads on the CPU checker backend (no GPU needed)."""
import sys, time, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_tracker_amd import synth, tracker
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC = os.path.join(ROOT, 'oracle', 'libft_oracle_tracker.so')
lib = tracker.bind(ctypes.CDLL(ORC)); lib.fto_set_threads(1)
S = 32
seqs = [synth.Sequence(188, 120, 44, stream_id=s) for s in range(S)]
for g, ht in ((1, 1), (1, 2), (1, 4), (1, 8), (2, 2), (4, 4)):
    b = tracker.BatchTracker(S, 188, 120, lib=lib, win_size=15, n_groups=g, host_threads=ht, num_pts=150, grid_x=3, grid_y=2, min_px_dist=4)
    for s, q in enumerate(seqs): b.set_calibration(s, q.calib8, q.T_cam_imu)
    t0 = None
    for i in range(44):
        if i == 4:
            b.stage_times(); t0 = time.perf_counter()
        b.feed_host([q.timestamps[i] for q in seqs], [q.frames[i] for q in seqs])
    dt = (time.perf_counter() - t0) / 40
    st = b.stage_times()
    print(g, ht, round(dt * 1e3, 2), {k: round(v / g, 3) for k, v in st.items() if k in ('klt_prepare', 'db_update', 'klt_enqueue', 'total')}, flush=True)
