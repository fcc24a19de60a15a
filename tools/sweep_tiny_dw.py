#!/usr/bin/env python3
"""Split-count sweep of the dW GEMMs with a tiny output (latent / class widths): us per launch incl. the slab reduce,
for the slab form and the atomic form.  Shapes = (N, K, q dtype) of the step's six tiny dW GEMMs at B=65536."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops
from mmvae.ops import PREC_BF16

dev, M = "cuda", 65536
slab = torch.empty(1 << 25, dtype=torch.float32, device=dev)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def tn(N, K, nsplit, use_slab):
    P = [torch.randn(M, ops.ceil_to(N, 8), device=dev).bfloat16() for _ in range(4)]
    Q = [torch.randn(M, ops.ceil_to(K, 8), device=dev).bfloat16() for _ in range(4)]
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    it = [0]
    def f():
        it[0] += 1
        ops.gemm_tn(PREC_BF16, P[it[0] & 3], Q[it[0] & 3], dw, db, N, K, nsplit=nsplit, slab=slab if use_slab else None)
    return timeit(f)


shapes = [("EncB.heads", 40, 256), ("EncA.heads", 40, 128), ("DecB.L0", 256, 20), ("DecA.L0", 128, 20), ("DecC.L0", 64, 20), ("DecC.L1", 24, 64)]
for name, N, K in shapes:
    row = []
    for ns in (0, 32, 64, 128, 256, 512):
        row.append((ns, round(tn(N, K, ns, True), 1), round(tn(N, K, ns, False), 1)))
    print(f"{name:12s} N={N:3d} K={K:3d}  (nsplit, slab us, atomic us):", row, flush=True)
