import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_tracker_amd import device as D, synth
S = 64
w, h = 752, 480
seqs = [synth.Sequence(w, h, 1, stream_id=s) for s in range(S)]
ctx = D.DeviceContext(w, h, max_levels=4, n_streams=S, n_slots=2, max_points=S * 200, max_grid_cells=49, max_per_cell=8)
for s, q in enumerate(seqs):
    ctx.upload(s, 0, q.frames[0])
streams, slots = list(range(S)), [0] * S
ctx.build_pyramid(streams, slots, 3, 21, with_gradients=False)
for thr in (20, 40, 80, 255, 5):
    kp, n, cells = ctx.fast_grid(streams, slots, thr, 5, 5, 7)
    ctx.profile(True); ctx.kernel_times()
    for _ in range(20):
        ctx.fast_grid(streams, slots, thr, 5, 5, 7)
    t = ctx.kernel_times(); ctx.profile(False)
    print(f"thr {thr:3d}: {1e3 * t['fast_ms'] / t['fast_launches']:7.1f} us  survivors/img {cells.reshape(S, -1)[:, :25].sum(1).mean():8.1f}", flush=True)
