"""Single-stream latency (one 752x480 frame handed over as a host pointer, results back on the host) with the host
stage timers of the batch driver.  Needs a GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_tracker_amd import synth, tracker
n = 300
q = synth.Sequence(752, 480, n + 10, stream_id=99)
b = tracker.BatchTracker(1, 752, 480, num_pts=150, win_size=21, pyr_levels=3, use_gyro_seed=1)
b.set_calibration(0, q.calib8, q.T_cam_imu)
for t, g, a in zip(q.imu_t, q.gyro, q.accel):
    b.feed_imu(0, t, a, g)
lat = []
for i in range(n + 10):
    if i == 10:
        b.stage_times(reset=True)
    t0 = time.perf_counter()
    b.feed_host([q.timestamps[i]], [q.frames[i]])
    lat.append(time.perf_counter() - t0)
lat = np.array(lat[10:]) * 1e3
print("p50 %.4f ms  p95 %.4f ms  min %.4f" % (np.percentile(lat, 50), np.percentile(lat, 95), lat.min()))
print({k: round(v, 4) for k, v in b.stage_times().items()})
