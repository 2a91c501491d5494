"""Host bookkeeping cost of the batch tracker (prepare / finish stages), measured on the CPU checker backend so it
runs without a GPU: the stages are backend independent."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_tracker_amd import synth, tracker
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC = os.path.join(ROOT, "oracle", "libft_oracle_tracker.so")
lib = tracker.bind(ctypes.CDLL(ORC)); lib.fto_set_threads(8)
S, n_frames = 16, 40
seqs = [synth.Sequence(376, 240, n_frames, stream_id=s) for s in range(S)]
b = tracker.BatchTracker(S, 376, 240, lib=lib, win_size=15, pyr_levels=2, n_groups=1, host_threads=1, num_pts=150, use_gyro_seed=1)
for s, q in enumerate(seqs):
    b.set_calibration(s, q.calib8, q.T_cam_imu)
    for t, g, a in zip(q.imu_t, q.gyro, q.accel):
        b.feed_imu(s, t, a, g)
for i in range(5):
    b.feed_host([q.timestamps[i] for q in seqs], [q.frames[i] for q in seqs])
b.stage_times(reset=True)
tr = 0
for i in range(5, n_frames):
    tr += b.feed_host([q.timestamps[i] for q in seqs], [q.frames[i] for q in seqs])[0]
st = b.stage_times()
per = 1e3 / S   # ms per feed -> us per stream-frame
print("tracked per stream-frame", tr / (S * (n_frames - 5)))
print({k: round(v * per, 2) for k, v in st.items()}, "us per stream-frame")
