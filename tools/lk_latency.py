"""Latency chain of one LK wavefront (needs a GPU): with at most one wavefront per SIMD the kernel time is the time one
wavefront needs, so the parameter sets below separate the prologue, the per-level staging + template and the iterations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_tracker_amd import device as D, synth
S = 64; w, h = 752, 480
seqs = [synth.Sequence(w, h, 2, stream_id=s) for s in range(S)]
ctx = D.DeviceContext(w, h, max_levels=4, n_streams=S, n_slots=2, max_points=S * 400, max_grid_cells=49, max_per_cell=16)
for s, q in enumerate(seqs):
    ctx.upload(s, 0, q.frames[0]); ctx.upload(s, 1, q.frames[1])
ctx.build_pyramid(list(range(S)) * 2, [0] * S + [1] * S, 3, 21, with_gradients=False)
kp, n, _ = ctx.fast_grid(list(range(S)), [0] * S, 20, 5, 5, 13)
for per in (1, 16):
    jobs = [dict(prev=(s, 0), next=(s, 1), prev_xy=kp[s, :min(n[s], per), :2].copy()) for s in range(S)]
    npts = sum(len(j["prev_xy"]) for j in jobs)
    for label, kw in (("full", {}), ("max_iter=1", dict(max_iter=1)), ("max_iter=0", dict(max_iter=0)), ("L2 only, no iter", dict(max_level=2, max_iter=0)),
                      ("L1 only, no iter", dict(max_level=1, max_iter=0)), ("L0 only, no iter", dict(max_level=0, max_iter=0)), ("L0 only", dict(max_level=0)),
                      ("no error", dict(flags=D.FT_LK_USE_INITIAL_FLOW | D.FT_LK_NO_ERROR))):
        prm = D.LKParams(win=21, **dict(dict(max_level=3), **kw))
        ctx.lk_track(jobs, prm); ctx.profile(1); ctx.kernel_times()
        for _ in range(20): ctx.lk_track(jobs, prm)
        t = ctx.kernel_times(); ctx.profile(0)
        print(f"{npts:6d} points  {label:18s} {1e3 * t['lk_ms'] / t['lk_launches']:7.1f} us", flush=True)
# points far outside the image: every level is skipped at once (launch + prologue + epilogue only)
for per in (1, 16):
    jobs = [dict(prev=(s, 0), next=(s, 1), prev_xy=np.full((per, 2), -900.0, np.float32)) for s in range(S)]
    prm = D.LKParams(win=21, max_level=3)
    ctx.lk_track(jobs, prm); ctx.profile(1); ctx.kernel_times()
    for _ in range(20): ctx.lk_track(jobs, prm)
    t = ctx.kernel_times(); ctx.profile(0)
    print(f"{per * S:6d} points  all outside        {1e3 * t['lk_ms'] / t['lk_launches']:7.1f} us", flush=True)
ctx.queue_time_begin(); print("empty event pair", 1e3 * ctx.queue_time_end(), "us")
