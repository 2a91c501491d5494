"""What a plain streaming kernel reaches on this GPU (needs a GPU): device-to-device copy, fill and read-reduce of a
buffer far larger than the caches, through PyTorch's own kernels.  The pyramid / gradient passes are judged against the
8 TB/s peak; these figures say how much of that peak any kernel gets in practice."""
import time, torch
dev = torch.device("cuda", 0)
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device=dev)
b = torch.empty(n, dtype=torch.uint8, device=dev)
a.fill_(1); b.fill_(2)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps
t = timed(lambda: b.copy_(a)); print(f"copy  1 GiB: {2 * n / t / 1e12:.2f} TB/s (read + write)")
t = timed(lambda: b.fill_(3)); print(f"fill  1 GiB: {n / t / 1e12:.2f} TB/s (write)")
a32 = a.view(torch.int32)
t = timed(lambda: a32.sum()); print(f"sum   1 GiB: {n / t / 1e12:.2f} TB/s (read)")
# the benchmark's own footprint: 64 frames of 752x480 (23 MB) read, 4 x as much written
src = torch.empty(64 * 752 * 480, dtype=torch.uint8, device=dev)
dst = torch.empty(4 * 64 * 752 * 480, dtype=torch.uint8, device=dev)
t = timed(lambda: dst.fill_(5), reps=200); print(f"fill 92 MB : {dst.numel() / t / 1e12:.2f} TB/s (write, {t * 1e6:.1f} us)")
t = timed(lambda: dst[:src.numel()].copy_(src), reps=200); print(f"copy 23 MB : {2 * src.numel() / t / 1e12:.2f} TB/s ({t * 1e6:.1f} us)")
