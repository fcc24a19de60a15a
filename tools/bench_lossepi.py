#!/usr/bin/env python3
"""Decoder-final GEMM with the reconstruction loss in its epilogue (EPI_LOSS_*) against the store epilogue + mmvae_vae_loss pair it
replaces: results and time.   python tools/bench_lossepi.py [M]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops
from mmvae.ops import PREC_BF16
dev, M = "cuda", int(sys.argv[1]) if len(sys.argv) > 1 else 65536


def t(f, n=10):
    f(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, N, K, bce in (("DecoderA.L1 782<-128 MSE", 782, 128, False), ("DecoderB.L2 572<-512 BCE", 572, 512, True)):
    torch.manual_seed(1)
    A = torch.randn(M, K, device=dev).bfloat16()
    W = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev) * 0.1
    pl = ops.PreparedLinear([W], [bias], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    T = torch.rand(M, N, device=dev) if bce else torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev)
    Np = ops.ceil_to(N, 8)
    g_ref = torch.empty(M, Np, dtype=torch.bfloat16, device=dev); g_new = torch.full((M, Np), 7.0, dtype=torch.bfloat16, device=dev)
    sums, _ = ops.loss_workspace(dev)
    sums2 = torch.zeros(5, dtype=torch.float64, device=dev)

    def old():
        ops.gemm_nt(PREC_BF16, A, pl.w, N, K, out, bias=pl.bias, act=ops.ACT_SIGMOID if bce else ops.ACT_NONE)
        if bce:
            ops.vae_loss(M, recon_b=out, b=T, sums=sums, g_b=g_ref, grad_b_wrt_logit=True)
        else:
            ops.vae_loss(M, recon_a=out, a=T, sums=sums, g_a=g_ref)

    def new():
        ops.gemm_nt(PREC_BF16, A, pl.w, N, K, g_new, bias=pl.bias, epilogue=ops.EPI_LOSS_BCE_LOGIT if bce else ops.EPI_LOSS_MSE, h=T,
                    loss_sum=sums2[1:2] if bce else sums2[0:1])
    sums.zero_(); old(); torch.cuda.synchronize()
    new(); torch.cuda.synchronize()
    ref = sums[1 if bce else 0].item(); got = sums2[1 if bce else 0].item()
    print(f"{name}: loss {got:.9e} vs {ref:.9e} (rel {abs(got / ref - 1):.1e}); gradient identical: {torch.equal(g_ref, g_new)} "
          f"(max |diff| {(g_ref.float() - g_new.float()).abs().max().item():.2e})")
    print(f"    GEMM(store) + loss kernel {t(old):7.1f} us     GEMM with the loss epilogue {t(new):7.1f} us", flush=True)
