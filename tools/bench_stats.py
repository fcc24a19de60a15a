#!/usr/bin/env python3
"""What the f64 column-statistics atomics at the end of a GEMM cost (512 row tiles add onto the same N addresses):
    python tools/bench_stats.py            (product library)
    MMVAE_LIB_PATH=.../libmmvae_nostat.so python tools/bench_stats.py     (make VARIANT=nostat VARIANT_FLAGS=-DMM_NO_STAT_ATOMICS)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops
from mmvae.ops import PREC_BF16
dev, M = "cuda", 65536


def t(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print(os.environ.get("MMVAE_LIB_PATH", "product library"))
for name, N, K, a32 in (("EncoderA.L0.fwd 128<-782", 128, 782, True), ("EncoderB.L0.fwd 512<-572", 512, 572, True), ("EncoderB.L1-like 256<-512 bf16 A", 256, 512, False)):
    A = [torch.rand(M, K, device=dev) if a32 else torch.rand(M, K, device=dev).bfloat16() for _ in range(3)]
    W = torch.randn(N, K, device=dev) / K ** 0.5
    pl = ops.PreparedLinear([W], [torch.zeros(N, device=dev)], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    st = torch.zeros(2, N, dtype=torch.float64, device=dev)
    i = [0]
    def f(): i[0] += 1; ops.gemm_nt(PREC_BF16, A[i[0] % 3], pl.w, N, K, out, bias=pl.bias, stats=st)
    def g(): i[0] += 1; ops.gemm_nt(PREC_BF16, A[i[0] % 3], pl.w, N, K, out, bias=pl.bias)
    print(f"  {name:36s} with statistics {t(f):6.1f} us   without {t(g):6.1f} us")
# BN-backward dX epilogue (statistics of d and d*xhat)
for name, N, K in (("EncoderA.L0.dX 128<-40", 128, 40), ("EncoderB.L1.dX 256<-40", 256, 40), ("EncoderB.L0.dX 512<-256", 512, 256)):
    A = [torch.randn(M, K, device=dev) if K == 40 else torch.randn(M, K, device=dev).bfloat16() for _ in range(3)]
    W = torch.randn(N, K, device=dev) / K ** 0.5
    pl = ops.PreparedLinear([W], [torch.zeros(N, device=dev)], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    Y = [torch.randn(M, N, device=dev).bfloat16() for _ in range(3)]
    mask = [(torch.rand(M, N, device=dev) > 0.1).to(torch.uint8) for _ in range(3)]
    fn = lambda: torch.rand(N, device=dev) + 0.5
    sc, sh, mu, rs = fn(), fn() - 1.0, fn() - 1.0, fn()
    d = torch.empty(M, N, dtype=torch.bfloat16, device=dev); st = torch.zeros(2, N, dtype=torch.float64, device=dev)
    i = [0]
    def f(): i[0] += 1; j = i[0] % 3; ops.gemm_nt(PREC_BF16, A[j], pl.w, N, K, d, epilogue=ops.EPI_BN_BWD, h=Y[j], bn=(sc, sh, mu, rs, mask[j], 1.0 / 0.9), bn_phase=2, stats=st)
    print(f"  {name:36s} BatchNorm-backward epilogue {t(f):6.1f} us")
