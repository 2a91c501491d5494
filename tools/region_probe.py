"""Timeline of a short timed region of the benchmark workload: wall time at which every pipelined feed call returns,
then the flush; three regions of K steps after W warm-up steps (needs a GPU).  usage: region_probe.py [K] [W]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from feature_tracker_amd import tracker
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
wl = bench.WORKLOADS["cfg4"]
ids = list(range(wl["streams"]))
seqs = bench.make_sequences(wl, W + 1 + 3 * K, ids, 16)
dev = torch.device("cuda", 0)
dev_frames = [torch.from_numpy(q.frames).to(dev) for q in seqs]
torch.cuda.synchronize()
f = bench.Feeder(tracker, wl, seqs, dev_frames, 0, 0, 0, False)
f.run(0, W + 1)
for r in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for i in range(W + 1 + r * K, W + 1 + (r + 1) * K):
        f.feed(i)
        marks.append(time.perf_counter() - t0)
    f.b.flush()
    t_end = time.perf_counter() - t0
    spin = float(os.environ.get("PROBE_SPIN_US", "0")) * 1e-6   # busy-wait on the host before the synchronize
    while time.perf_counter() - t0 - t_end < spin:
        pass
    torch.cuda.synchronize()
    t_sync = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_sync2 = time.perf_counter() - t0
    print(f"region {r}: device synchronize after the flush returns at {1e6 * t_sync:.0f}, a second one at {1e6 * t_sync2:.0f}")
    print(f"region {r}: starts at {1e6 * t0:.0f} on the monotonic clock (FT_GROUP_TRACE=1 prints the groups' instants on it); feeds return at (us)", " ".join(f"{1e6 * m:.0f}" for m in marks), f"| flush done {1e6 * t_end:.0f}", flush=True)
