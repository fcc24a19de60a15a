#!/usr/bin/env python3
"""HBM bandwidth calibration with stock torch kernels (fill / copy / read-reduce) at the sizes of this model's activations."""
import torch
dev = "cuda"
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for mb in (64, 128, 256, 512, 1024):
    n = mb * (1 << 20) // 4
    bufs = [torch.empty(n, device=dev) for _ in range(4)]
    src = [torch.randn(n, device=dev) for _ in range(4)]
    i = [0]
    def fill(): i[0] += 1; bufs[i[0] % 4].fill_(1.0)
    def copy(): i[0] += 1; bufs[i[0] % 4].copy_(src[i[0] % 4])
    def read(): i[0] += 1; src[i[0] % 4].sum()
    tf, tc, tr = t(fill), t(copy), t(read)
    print(f"{mb:5d} MiB: fill {tf:7.1f} us = {mb * 1.048576e6 / tf / 1e6:5.2f} TB/s write | copy {tc:7.1f} us = {2 * mb * 1.048576e6 / tc / 1e6:5.2f} TB/s r+w | sum {tr:7.1f} us = {mb * 1.048576e6 / tr / 1e6:5.2f} TB/s read")
