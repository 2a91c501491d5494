"""LK kernel time with one wavefront per compute unit or fewer (needs a GPU): the latency of one wavefront's chain,
split by parameter set (levels, Newton iterations)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_tracker_amd import device as D, synth
if os.environ.get("FT_EXP_LIB"):   # an experimental build of the device library
    D.LIB_PATH = os.path.abspath(os.environ["FT_EXP_LIB"])
S = 64; w, h = 752, 480
seqs = [synth.Sequence(w, h, 2, stream_id=s) for s in range(S)]
ctx = D.DeviceContext(w, h, max_levels=4, n_streams=S, n_slots=2, max_points=S * 400, max_grid_cells=49, max_per_cell=16)
for s, q in enumerate(seqs):
    ctx.upload(s, 0, q.frames[0]); ctx.upload(s, 1, q.frames[1])
ctx.build_pyramid(list(range(S)) * 2, [0] * S + [1] * S, 3, 21, with_gradients=False)
kp, n, _ = ctx.fast_grid(list(range(S)), [0] * S, 20, 5, 5, 13)
PER = [int(x) for x in sys.argv[1:]] or [1, 4]
for per in PER:
    jobs = [dict(prev=(s, 0), next=(s, 1), prev_xy=kp[s, :min(n[s], per), :2].copy()) for s in range(S)]
    npts = sum(len(j["prev_xy"]) for j in jobs)
    for label, kw in (("full", {}), ("max_iter=1", dict(max_iter=1)), ("max_iter=0", dict(max_iter=0)),
                      ("max_level=0", dict(max_level=0)), ("max_level=0 max_iter=0", dict(max_level=0, max_iter=0)),
                      ("max_level=1", dict(max_level=1))):
        kw = dict(kw)
        prm = D.LKParams(win=21, max_level=kw.pop("max_level", 3), **kw)
        ctx.lk_track(jobs, prm); ctx.profile(True); ctx.kernel_times()
        for _ in range(20): ctx.lk_track(jobs, prm)
        t = ctx.kernel_times(); ctx.profile(False)
        us = 1e3 * t['lk_ms'] / t['lk_launches']
        print(f"{npts:6d} points  {label:24s} {us:7.1f} us", flush=True)
