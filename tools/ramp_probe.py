"""Step time of the benchmark workload as a function of the step index from a cold start (needs a GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from feature_tracker_amd import tracker
wl = bench.WORKLOADS["cfg4"]
K=int(os.environ.get('RAMP_STEPS', '240'))
seqs = bench.make_sequences(wl, K + 2, list(range(wl["streams"])), 16)
dev = torch.device("cuda", 0)
dev_frames = [torch.from_numpy(q.frames).to(dev) for q in seqs]
torch.cuda.synchronize()
time.sleep(0.5)
f = bench.Feeder(tracker, wl, seqs, dev_frames, 0, 0, 0, False)
f.run(0, 1)
torch.cuda.synchronize()
if os.environ.get('RAMP_PREHEAT'):   # keep the device busy for a while right before the first step: clock ramp or host-side warm-up?
    a = torch.randn(4096, 4096, device=dev)
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < float(os.environ['RAMP_PREHEAT']):
        for _ in range(10):
            a = (a @ a).clamp_(-1, 1)
        torch.cuda.synchronize()
t0 = time.perf_counter(); marks=[]
for i in range(1, K+1):
    f.feed(i); marks.append(time.perf_counter()-t0)
f.b.flush()
m=np.array(marks)
for a in range(0, K-20, 20):
    print(f"steps {a:3d}-{a+20:3d}: {(m[a+20-1]-m[a-1 if a else 0])/ (20 if a else 19)*1e6:7.1f} us/step")

if os.environ.get('RAMP_ONLY'): sys.exit(0)
# the same from a fresh tracker with synchronous feeds: host stage times of every early step (summed over the groups)
del f
f = bench.Feeder(tracker, wl, seqs, dev_frames, 0, 0, 0, False)
for i in range(0, 40):
    t = time.perf_counter()
    f.feed(i, pipelined=False)
    dt = time.perf_counter() - t
    st = f.b.stage_times(reset=True)
    if i < 12 or i % 4 == 0:
        print(f"sync step {i:3d}: {dt*1e6:7.1f} us  " + " ".join(f"{k}={v*1e3:.0f}" for k, v in st.items()))

# a third tracker, pipelined again: everything that is paid once per process has been paid by now
del f
f = bench.Feeder(tracker, wl, seqs, dev_frames, 0, 0, 0, False)
f.run(0, 1)
torch.cuda.synchronize()
t0 = time.perf_counter(); marks = []
n3 = min(120, K)   # the sequences hold K + 1 frames
for i in range(1, n3 + 1):
    f.feed(i); marks.append(time.perf_counter() - t0)
f.b.flush()
m = np.array(marks)
print("warm process, new tracker, pipelined: us per step over steps", " ".join(f"{a}-{a+10}:{(m[a+9]-(m[a-1] if a else 0))/10*1e6:.0f}" for a in range(0, n3 - 9, 10)))
