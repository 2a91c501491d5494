"""Per-call durations of the pipelined feed on the bench workload (needs a GPU): shows whether slow steps are uniform
or intermittent."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from feature_tracker_amd import tracker
wl = bench.WORKLOADS["cfg4"]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seqs = bench.make_sequences(wl, K + 10, 0, 16)
dev = torch.device("cuda", 0)
frames = [torch.from_numpy(q.frames).to(dev) for q in seqs]
torch.cuda.synchronize()
S, w, h = wl["streams"], wl["w"], wl["h"]
ptrs = [[frames[s][i].data_ptr() for s in range(S)] for i in range(K + 10)]
stamps = [[q.timestamps[i] for q in seqs] for i in range(K + 10)]
for rep in range(3):
    b = bench.setup_tracker(tracker, wl, seqs, device=0)
    for i in range(9):
        b.feed_device(stamps[i], ptrs[i], w, h, w)
    b.stage_times(reset=True)
    dts = []
    t0 = time.perf_counter()
    for i in range(9, K + 9):
        t1 = time.perf_counter()
        b.feed_device(stamps[i], ptrs[i], w, h, w, pipelined=True)
        dts.append(time.perf_counter() - t1)
    t_loop = time.perf_counter() - t0
    b.flush()
    total = time.perf_counter() - t0
    d = np.array(dts) * 1e6
    st = b.stage_times()
    print(f"rep {rep}: loop {1e6 * t_loop / K:.0f} us/step, flush {1e3 * (total - t_loop):.2f} ms;", end=" ")
    print(f"{1e6 * total / K:.0f} us/step; per-call p10 {np.percentile(d, 10):.0f} p50 {np.percentile(d, 50):.0f} p90 {np.percentile(d, 90):.0f} max {d.max():.0f}")
    print("   first 40:", " ".join(f"{x:.0f}" for x in d[:40]))
    print("   stages(us):", {k: round(v * 1e3) for k, v in st.items()})
    del b
