"""What HBM gives plain streaming kernels on this box (torch fill / copy / sum on 1 GiB), for reading the step's
per-kernel bandwidths against: python scripts/hbm_rates.py   (through gpurun)"""
import torch

n = 1 << 28  # floats: 1 GiB
a = torch.empty(n, dtype=torch.float32, device="cuda:0")
b = torch.empty(n, dtype=torch.float32, device="cuda:0")


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for name, fn, nbytes in (("fill (write only)", lambda: a.fill_(1.0), 4 * n),
                         ("copy (read + write)", lambda: b.copy_(a), 8 * n),
                         ("sum (read only)", lambda: a.sum(), 4 * n),
                         ("a += 1 (read + write in place)", lambda: a.add_(1.0), 8 * n)):
    t = timed(fn)
    print(f"{name:32s} {nbytes / t / 1e12:6.2f} TB/s  ({t * 1e6:8.1f} us)")
for mb in (64, 128):  # buffers the size of one layer's activations
    m = mb * (1 << 18)
    t = timed(lambda: a[:m].fill_(1.0), 50)
    print(f"fill {mb:4d} MB                     {4 * m / t / 1e12:6.2f} TB/s  ({t * 1e6:8.1f} us)")
    t = timed(lambda: b[:m].copy_(a[:m]), 50)
    print(f"copy {mb:4d} MB -> {mb:4d} MB          {8 * m / t / 1e12:6.2f} TB/s  ({t * 1e6:8.1f} us)")
