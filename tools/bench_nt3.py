#!/usr/bin/env python3
"""Interleaved A/B of the NT kernel generations on the hot bf16-A shapes (one process, mmvae_set_tuning switches, rounds interleaved)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops, _lib as L
from mmvae.ops import PREC_BF16
lib = L.load()
dev, M = "cuda", int(os.environ.get("M", 65536))


def make(N, K, kind):
    A = [torch.randn(M, ops.ceil_to(K, 8), device=dev).bfloat16() for _ in range(3)]
    W = torch.randn(N, K, device=dev) / K ** 0.5
    pl = ops.PreparedLinear([W], [torch.zeros(N, device=dev)], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    H = torch.randn(M, ops.ceil_to(N, 8), device=dev).bfloat16()
    i = [0]
    if kind == "f32":
        out = torch.empty(M, N, device=dev)
        def f(): i[0] += 1; ops.gemm_nt(PREC_BF16, A[i[0] % 3], pl.w, N, K, out, bias=pl.bias, act=2)
    elif kind == "bf16":
        out = torch.empty(M, ops.ceil_to(N, 8), dtype=torch.bfloat16, device=dev)
        def f(): i[0] += 1; ops.gemm_nt(PREC_BF16, A[i[0] % 3], pl.w, N, K, out, bias=pl.bias, act=1)
    elif kind == "relu":
        out = torch.empty(M, ops.ceil_to(N, 8), dtype=torch.bfloat16, device=dev)
        def f(): i[0] += 1; ops.gemm_nt(PREC_BF16, A[i[0] % 3], pl.w, N, K, out, epilogue=ops.EPI_RELU_MASK, h=H)
    else:
        mask = (torch.rand(M, N, device=dev) > 0.1).to(torch.uint8)
        g = lambda: torch.rand(N, device=dev) + 0.5
        bn = (g(), g() - 1.0, g() - 1.0, g(), mask, 1.0 / 0.9)
        st = torch.zeros(2, N, dtype=torch.float64, device=dev)
        out = torch.empty(M, ops.ceil_to(N, 8), dtype=torch.bfloat16, device=dev)
        def f(): i[0] += 1; ops.gemm_nt(PREC_BF16, A[i[0] % 3], pl.w, N, K, out, epilogue=ops.EPI_BN_BWD, h=H, bn=bn, bn_phase=2, stats=st)
    return f


def t(f, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


shapes = [("DecB.L2.fwd 572x512 f32", 572, 512, "f32"), ("DecB.L1.fwd 512x256 bf16", 512, 256, "bf16"), ("DecB.L2.dX 512x572 relu", 512, 572, "relu"),
          ("DecB.L1.dX 256x512 relu", 256, 512, "relu"), ("EncB.L0.dX 512x256 bnbwd", 512, 256, "bn"), ("512x1024 bf16", 512, 1024, "bf16")]
modes = [("gen1", 0, 0), ("gen2", 0, 1), ("gen3", 2, 1)]
for name, N, K, kind in shapes:
    f = make(N, K, kind)
    res = {m[0]: [] for m in modes}
    for rnd in range(5):
        for mname, nt3, nt2 in modes:
            lib.mmvae_set_tuning(1, nt3); lib.mmvae_set_tuning(2, nt2)
            if rnd == 0:
                t(f, 3)
            res[mname].append(t(f))
    print(f"{name:28s}", "  ".join(f"{m}: med {sorted(v)[len(v) // 2]:6.1f} min {min(v):6.1f}" for m, v in res.items()))
lib.mmvae_set_tuning(1, 0); lib.mmvae_set_tuning(2, 1)
