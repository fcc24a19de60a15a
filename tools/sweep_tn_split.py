#!/usr/bin/env python3
"""Batch-split sweep of the plain bf16 x bf16 dW GEMM (gemm_tn DMA form + its slab reduce): us per call for each nsplit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "vae-los-angeles_amd")]
os.environ.setdefault("M", "65536")
import torch
from bench_gemm_lib import timeit, M, dev
from mmvae import ops
from mmvae.ops import PREC_BF16
slab = torch.empty(1 << 25, device=dev)

def tn(N, K, nsplit):
    P = [torch.randn(M, ops.ceil_to(N, 8), device=dev).bfloat16() for _ in range(3)]
    Q = [torch.randn(M, ops.ceil_to(K, 8), device=dev).bfloat16() for _ in range(3)]
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    i = [0]
    def f():
        i[0] += 1
        ops.gemm_tn(PREC_BF16, P[i[0] % 3], Q[i[0] % 3], dw, db, N, K, slab=slab, nsplit=nsplit)
    return timeit(f)

for N, K in ((512, 256), (256, 512), (572, 512), (782, 128)):
    print(N, K, {s: round(tn(N, K, s), 1) for s in (0, 8, 16, 24, 32, 40, 48, 64)})
