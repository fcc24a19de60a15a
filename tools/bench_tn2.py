#!/usr/bin/env python3
"""A/B of the hot TN (dW) shapes with plain bf16 operands; run under different MMVAE_* settings."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "vae-los-angeles_amd")]
os.environ.setdefault("M", "65536")
import torch
from bench_gemm_lib import timeit, M, dev
from mmvae import ops
from mmvae.ops import PREC_BF16
slab = torch.empty(1 << 25, device=dev)

def tn(N, K):
    P = [torch.randn(M, ops.ceil_to(N, 8), device=dev).bfloat16() for _ in range(3)]
    Q = [torch.randn(M, ops.ceil_to(K, 8), device=dev).bfloat16() for _ in range(3)]
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    i = [0]
    def f():
        i[0] += 1
        ops.gemm_tn(PREC_BF16, P[i[0] % 3], Q[i[0] % 3], dw, db, N, K, slab=slab)
    return timeit(f)

print(os.environ.get("TAG", ""), {k: v for k, v in os.environ.items() if k.startswith("MMVAE_")})
print("  DecB.L2.dW N=572 K=512:", round(tn(572, 512), 1))
print("  DecB.L1.dW N=512 K=256:", round(tn(512, 256), 1))
print("  DecA.L1.dW N=782 K=128:", round(tn(782, 128), 1))
print("  EncB.L1.dW-like N=256 K=512:", round(tn(256, 512), 1))
print("  EncB.L0.dW-like N=512 K=576 (bf16 shadow of b):", round(tn(512, 572), 1))
print("  EncA.L0.dW-like N=128 K=782 (bf16 shadow of a):", round(tn(128, 782), 1))
