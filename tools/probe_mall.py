#!/usr/bin/env python3
"""Does the loss kernel read a just-written reconstruction from the Infinity Cache?  Times the MSE part of mmvae_vae_loss on a
[65536][782] fp32 tensor (205 MB) right after that tensor was written vs after 1 GiB of other traffic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops
B, A = 65536, 782
dev = "cuda"
x = torch.rand(B, A, device=dev); t = torch.rand(B, A, device=dev)
g = torch.empty(B, 784, dtype=torch.bfloat16, device=dev)
big = torch.empty(1 << 28, device=dev)           # 1 GiB
sums = torch.zeros(4, dtype=torch.float64, device=dev)
def run(prep):
    ts = []
    for _ in range(5):
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.vae_loss(B, recon_a=x, a=t, sums=sums, g_a=g)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[2]
print("recon just written (x.mul_(1.0)):        %.1f us" % run(lambda: x.mul_(1.0)))
print("recon AND target just touched:           %.1f us" % run(lambda: (t.mul_(1.0), x.mul_(1.0))))
print("after 1 GiB of unrelated writes:          %.1f us" % run(lambda: big.fill_(1.0)))
