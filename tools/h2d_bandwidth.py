import torch, time
n=64*752*480
h=torch.empty(n,dtype=torch.uint8).pin_memory()
d=torch.empty(n,dtype=torch.uint8,device='cuda')
for size in (n, n//64):
    hs=h[:size]; ds=d[:size]
    for _ in range(5): ds.copy_(hs,non_blocking=True)
    torch.cuda.synchronize()
    t=time.perf_counter(); R=50
    for _ in range(R): ds.copy_(hs,non_blocking=True)
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t)/R
    print(f"H2D {size/1e6:.2f} MB: {dt*1e6:.1f} us  {size/dt/1e9:.1f} GB/s")
# 64 separate copies of one frame each
fs=752*480
for _ in range(3):
    for i in range(64): d[i*fs:(i+1)*fs].copy_(h[i*fs:(i+1)*fs],non_blocking=True)
torch.cuda.synchronize()
t=time.perf_counter(); R=20
for _ in range(R):
    for i in range(64): d[i*fs:(i+1)*fs].copy_(h[i*fs:(i+1)*fs],non_blocking=True)
torch.cuda.synchronize()
dt=(time.perf_counter()-t)/R
print(f"64 x frame copies: {dt*1e6:.1f} us  {n/dt/1e9:.1f} GB/s")
