"""LK kernel micro-benchmark: 64 images, 150 detected features each; prints the mean kernel time per launch for
several parameter sets (needs a GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feature_tracker_amd import device as D, synth
if os.environ.get("FT_EXP_LIB"):   # an experimental build of the device library
    D.LIB_PATH = os.path.abspath(os.environ["FT_EXP_LIB"])
S = 64
w, h = 752, 480
seqs = [synth.Sequence(w, h, 2, stream_id=s) for s in range(S)]
ctx = D.DeviceContext(w, h, max_levels=4, n_streams=S, n_slots=2, max_points=S * 200, max_grid_cells=49, max_per_cell=8)
for s, q in enumerate(seqs):
    ctx.upload(s, 0, q.frames[0]); ctx.upload(s, 1, q.frames[1])
ctx.build_pyramid(list(range(S)) * 2, [0] * S + [1] * S, 3, 21)
kp, n, _ = ctx.fast_grid(list(range(S)), [0] * S, 20, 5, 5, 7)
jobs = [dict(prev=(s, 0), next=(s, 1), prev_xy=kp[s, :min(n[s], 150), :2].copy()) for s in range(S)]
npts = sum(len(j["prev_xy"]) for j in jobs)
def run(label, **kw):
    prm = D.LKParams(win=kw.pop("win", 21), max_level=kw.pop("max_level", 3), **kw)
    ctx.lk_track(jobs, prm); ctx.sync()
    ctx.profile(True); ctx.kernel_times()
    for _ in range(20):
        ctx.lk_track(jobs, prm)
    t = ctx.kernel_times()
    ctx.profile(False)
    print(f"{label:28s} {1e3 * t['lk_ms'] / t['lk_launches']:8.1f} us  ({npts} points)", flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if mode == "iter0":      # one parameter set only, for counter passes
    run("max_iter=0", max_iter=0); sys.exit(0)
if mode == "iter1":
    run("max_iter=1", max_iter=1); sys.exit(0)
run("full win21 L3")
if mode == "full":
    sys.exit(0)
run("max_iter=1", max_iter=1)
run("max_iter=0", max_iter=0)
run("max_level=0", max_level=0)
run("max_level=0 max_iter=0", max_level=0, max_iter=0)
run("win15 L3", win=15)
run("fb win21 L3", flags=D.FT_LK_USE_INITIAL_FLOW | D.FT_LK_FORWARD_BACKWARD)
