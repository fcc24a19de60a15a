#!/usr/bin/env python3
"""Cost of one TN m-step / NT k-step at full occupancy (512 workgroups, 2 per CU): slope of time vs depth."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops
from mmvae.ops import PREC_BF16
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
res = []
for steps in (4, 8, 16, 32, 64):
    M = 512 * 64 * steps
    P = torch.randn(M, 128, device=dev).bfloat16(); Q = torch.randn(M, 128, device=dev).bfloat16()
    dw = torch.zeros(128, 128, device=dev); db = torch.zeros(128, device=dev)
    res.append((steps, round(timeit(lambda: ops.gemm_tn(PREC_BF16, P, Q, dw, db, 128, 128, nsplit=512)), 1)))
print("TN 128x128 tile, 512 blocks, m-steps/block sweep (us):", res)
res = []
for K in (256, 512, 1024, 2048, 4096):
    M = 65536
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(128, K, device=dev) / K ** 0.5; b = torch.zeros(128, device=dev)
    pl = ops.PreparedLinear([W], [b], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    out = torch.empty(M, 128, dtype=torch.bfloat16, device=dev)
    res.append((K // 64, round(timeit(lambda: ops.gemm_nt(PREC_BF16, A, pl.w, 128, K, out, bias=pl.bias)), 1)))
print("NT 128x128 tile, 512 blocks, k-steps/block sweep (us):", res)
