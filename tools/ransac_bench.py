"""Outlier-rejection kernel micro-benchmark (needs a GPU): 64 point sets of 150 pairs, few outliers (the benchmark's
regime: the search stops within the first round) and many outliers (several rounds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import synth as tsynth
from feature_tracker_amd import device as D
if os.environ.get("FT_EXP_LIB"):   # an experimental build of the device library
    D.LIB_PATH = os.path.abspath(os.environ["FT_EXP_LIB"])
calib = np.array([458.7, 457.3, 367.2, 248.4, 0, 0, 0, 0.0])
ctx = D.DeviceContext(64, 64, max_levels=1, n_streams=1, n_slots=1, max_points=64 * 200, max_grid_cells=4, max_per_cell=2)
for label, n_out, cal in (("1% outliers", 2, calib), ("30% outliers", 45, calib), ("1% outliers, distorted", 2, calib + np.array([0, 0, 0, 0, -0.28, 0.07, 1e-4, 1e-5]))):
    sets = []
    for s in range(64):
        prev, nxt, _ = tsynth.two_view_matches(150, n_out, cal, seed=s)
        sets.append(dict(prev_xy=prev, next_xy=nxt, calib_prev=cal))
    ctx.reject_outliers(sets, hypotheses=256, seed=1)
    ctx.profile(1); ctx.kernel_times()
    for r in range(20):
        st = ctx.reject_outliers(sets, hypotheses=256, seed=r)
    t = ctx.kernel_times(); ctx.profile(0)
    print(f"{label:26s} {1e3 * t['reject_ms'] / t['reject_launches']:7.1f} us per launch, kept {np.mean([x.mean() for x in st]):.3f}", flush=True)
