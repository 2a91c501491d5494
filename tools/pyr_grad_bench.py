"""Pyramid + Scharr gradient image build against the HBM roofline (needs a GPU): ft_build_pyramid with gradient
images (one pass per level: read the level once, write its gradient image and the next level) on batches of frames
already resident in the context's slots.  Prints, per shape, the mean device time of the launch group, the
algorithmic bytes (pyramid: read every level with a successor + write every level above 0; gradients: read every level
+ write 4 B/px) and the achieved GB/s against 8 TB/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import pyramid_kernel_bytes, HBM_PEAK_GBS
from feature_tracker_amd import device as D

rng = np.random.default_rng(5)
out = []
for w, h, lv, n in [(752, 480, 3, 64), (752, 480, 3, 256), (1280, 720, 4, 64), (3840, 2160, 5, 8), (3840, 2160, 5, 32)]:
    ctx = D.DeviceContext(w, h, max_levels=lv + 1, n_streams=n, n_slots=1, max_points=64, max_grid_cells=25, max_per_cell=2)
    img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    for s in range(n):
        ctx.upload(s, 0, img)
    streams, slots = list(range(n)), [0] * n
    for grads in (True, False):
        ctx.build_pyramid(streams, slots, lv, 21, with_gradients=grads)
        ctx.sync()
        ctx.profile(True); ctx.kernel_times()
        for _ in range(20):
            ctx.build_pyramid(streams, slots, lv, 21, with_gradients=grads)
        t = ctx.kernel_times()
        ctx.profile(False)
        ms = t["pyramid_ms"] / t["pyramid_launches"]
        by = pyramid_kernel_bytes(w, h, lv, with_gradients=grads, with_ingest=False) * n
        e = dict(shape=f"{n} x {w}x{h}, maxLevel {lv}", gradients=grads, avg_ms=ms, algorithmic_bytes=by, achieved_gbs=by / ms / 1e6,
                 frac_hbm_peak=by / ms / 1e6 / HBM_PEAK_GBS)
        out.append(e)
        print(json.dumps(e), flush=True)
    ctx.close()
