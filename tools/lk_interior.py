"""How much of the tracking kernel's time goes into border handling: the same number of points, once spread over the
whole frame and once confined to the region whose windows stay inside the image on every pyramid level (needs a GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_tracker_amd import device as D, synth
S, w, h = 64, 752, 480
seqs = [synth.Sequence(w, h, 2, stream_id=s) for s in range(S)]
ctx = D.DeviceContext(w, h, max_levels=4, n_streams=S, n_slots=2, max_points=S * 200, max_grid_cells=49, max_per_cell=8)
for s, q in enumerate(seqs):
    ctx.upload(s, 0, q.frames[0]); ctx.upload(s, 1, q.frames[1])
ctx.build_pyramid(list(range(S)) * 2, [0] * S + [1] * S, 3, 21)
rng = np.random.default_rng(0)
def jobs_in(x0, x1, y0, y1):
    return [dict(prev=(s, 0), next=(s, 1), prev_xy=np.stack([rng.uniform(x0, x1, 150), rng.uniform(y0, y1, 150)], 1).astype(np.float32)) for s in range(S)]
def run(label, jobs):
    prm = D.LKParams(win=21, max_level=3)
    ctx.lk_track(jobs, prm); ctx.sync()
    ctx.profile(True); ctx.kernel_times()
    for _ in range(20):
        ctx.lk_track(jobs, prm)
    t = ctx.kernel_times(); ctx.profile(False)
    print(f"{label:34s} {1e3 * t['lk_ms'] / t['lk_launches']:8.1f} us  iterations/level {t['lk_iterations'] / max(1, t['lk_level_visits']):.2f}", flush=True)
run("whole frame", jobs_in(12, w - 12, 12, h - 12))
run("interior on every level", jobs_in(200, 550, 130, 330))
run("whole frame", jobs_in(12, w - 12, 12, h - 12))
