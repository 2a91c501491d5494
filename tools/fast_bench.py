"""Detector / pyramid kernel micro-benchmark: 64 images of 752x480; prints the mean device time per call (needs a GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_tracker_amd import device as D, synth
if os.environ.get("FT_EXP_LIB"):   # an experimental build of the device library
    D.LIB_PATH = os.path.abspath(os.environ["FT_EXP_LIB"])
S = 64
w, h = 752, 480
seqs = [synth.Sequence(w, h, 1, stream_id=s) for s in range(S)]
ctx = D.DeviceContext(w, h, max_levels=4, n_streams=S, n_slots=2, max_points=S * 200, max_grid_cells=49, max_per_cell=8)
for s, q in enumerate(seqs):
    ctx.upload(s, 0, q.frames[0])
streams, slots = list(range(S)), [0] * S
ctx.build_pyramid(streams, slots, 3, 21, with_gradients=False)
kp, n, cells = ctx.fast_grid(streams, slots, 20, 5, 5, 7)
print("keypoints per image", n.mean(), "NMS survivors per image", cells.reshape(S, -1)[:, :25].sum(1).mean(), flush=True)
ctx.profile(True); ctx.kernel_times()
for _ in range(20):
    ctx.fast_grid(streams, slots, 20, 5, 5, 7)
t = ctx.kernel_times()
print(f"fast_grid   {1e3 * t['fast_ms'] / t['fast_launches']:8.1f} us", flush=True)
for _ in range(20):
    ctx.build_pyramid(streams, slots, 3, 21, with_gradients=False)
t = ctx.kernel_times()
print(f"pyramid     {1e3 * t['pyramid_ms'] / t['pyramid_launches']:8.1f} us", flush=True)
for _ in range(20):
    ctx.build_pyramid(streams, slots, 3, 21, with_gradients=True)
t = ctx.kernel_times()
print(f"pyramid+grad{1e3 * t['pyramid_ms'] / t['pyramid_launches']:8.1f} us", flush=True)
ctx.profile(False)
