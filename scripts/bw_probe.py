"""Calibration: achievable HBM write / copy bandwidth with plain torch kernels."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (64, 256, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    us = t(lambda: a.fill_(1.0)); print(f"fill  {mb:5d} MB {us:8.1f} us {mb*1.048576/us*1e3/1e3:6.2f} TB/s")
    us = t(lambda: b.copy_(a)); print(f"copy  {mb:5d} MB {us:8.1f} us {2*mb*1.048576/us*1e3/1e3:6.2f} TB/s (r+w)")
    us = t(lambda: a.sum()); print(f"sum   {mb:5d} MB {us:8.1f} us {mb*1.048576/us*1e3/1e3:6.2f} TB/s")
