#!/usr/bin/env python3
"""Micro-benchmark of single GEMM launches (through the C ABI) for kernel tuning:
time vs K (slope = cost of one K step, intercept = prologue + epilogue) and the hot-path shapes.

"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops
from mmvae.ops import PREC_BF16

dev = "cuda"
M = int(os.environ.get("M", 65536))


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3      # us


def nt(N, K, a_dtype=torch.bfloat16, out_dtype=torch.bfloat16, epi="store"):
    A = torch.randn(M, ops.ceil_to(K, 8), device=dev).to(a_dtype)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.zeros(N, device=dev)
    pl = ops.PreparedLinear([W], [b], PREC_BF16, dev)
    ops.WeightPrep([pl], dev).run()
    out = torch.empty(M, ops.ceil_to(N, 8) if out_dtype == torch.bfloat16 else N, dtype=out_dtype, device=dev)
    if epi == "store":
        return timeit(lambda: ops.gemm_nt(PREC_BF16, A, pl.w, N, K, out, bias=pl.bias, act=1))
    H = torch.randn(M, ops.ceil_to(N, 8), device=dev).bfloat16()
    return timeit(lambda: ops.gemm_nt(PREC_BF16, A, pl.w, N, K, out, epilogue=ops.EPI_RELU_MASK, h=H))


def tn(N, K, q_dtype=torch.bfloat16, nsplit=0):
    P = torch.randn(M, ops.ceil_to(N, 8), device=dev).bfloat16()
    Q = torch.randn(M, ops.ceil_to(K, 8) if q_dtype == torch.bfloat16 else K, device=dev).to(q_dtype)
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    return timeit(lambda: ops.gemm_tn(PREC_BF16, P, Q, dw, db, N, K, nsplit=nsplit))



def nt_bn_bwd(N, K, a_dtype=torch.bfloat16, with_stats=True):
    """dX GEMM with the BatchNorm-backward epilogue (phase 2: store d + column statistics; phase 1: apply, no statistics)."""
    A = torch.randn(M, ops.ceil_to(K, 8), device=dev).to(a_dtype)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    pl = ops.PreparedLinear([W], [torch.zeros(N, device=dev)], PREC_BF16, dev)
    ops.WeightPrep([pl], dev).run()
    y = torch.randn(M, N, device=dev).bfloat16()
    mask = (torch.rand(M, N, device=dev) > 0.1).to(torch.uint8)
    f = lambda: torch.rand(N, device=dev) + 0.5
    bn = (f(), f() - 1.0, f() - 1.0, f(), mask, 1.0 / 0.9)
    d = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
    coef = torch.rand(3, N, device=dev)
    if with_stats:
        return timeit(lambda: ops.gemm_nt(PREC_BF16, A, pl.w, N, K, d, epilogue=ops.EPI_BN_BWD, h=y, bn=bn, bn_phase=2, stats=stats))
    return timeit(lambda: ops.gemm_nt(PREC_BF16, A, pl.w, N, K, d, epilogue=ops.EPI_BN_BWD, h=y, bn=bn, bn_coef=coef))


def nt_stats(N, K, a_dtype=torch.float32, with_stats=True):
    A = torch.randn(M, ops.ceil_to(K, 8) if a_dtype == torch.bfloat16 else K, device=dev).to(a_dtype)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    pl = ops.PreparedLinear([W], [torch.zeros(N, device=dev)], PREC_BF16, dev)
    ops.WeightPrep([pl], dev).run()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
    return timeit(lambda: ops.gemm_nt(PREC_BF16, A, pl.w, N, K, out, bias=pl.bias, stats=stats if with_stats else None))


