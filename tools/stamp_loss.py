#!/usr/bin/env python3
"""Cycle stamps of the decoder-final GEMMs WITH their loss epilogue (gemm_nt2.h, EpiLoss): main-loop phases per K step and the
epilogue per tile.  Diagnostic library (make STAMP=1).   python tools/stamp_loss.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MMVAE_LIB_PATH"] = os.path.join(ROOT, "vae-los-angeles_amd", "mmvae", "libmmvae_stamp.so")
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import _lib as L, ops
from mmvae.ops import PREC_BF16
dev, M = "cuda", 65536
lib = L.load()
lib.mmvae_debug_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
buf = (C.c_uint64 * 12)()
for name, N, K, bce in (("DecoderB.L2 572<-512 BCE", 572, 512, True), ("DecoderA.L1 782<-128 MSE", 782, 128, False)):
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(3)]
    W = torch.randn(N, K, device=dev) / K ** 0.5
    pl = ops.PreparedLinear([W], [torch.zeros(N, device=dev)], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    T = [torch.rand(M, N, device=dev) for _ in range(3)]
    g = torch.empty(M, ops.ceil_to(N, 8), dtype=torch.bfloat16, device=dev)
    sums = torch.zeros(5, dtype=torch.float64, device=dev)

    def run(reps):
        for i in range(reps):
            ops.gemm_nt(PREC_BF16, A[i % 3], pl.w, N, K, g, bias=pl.bias, epilogue=ops.EPI_LOSS_BCE_LOGIT if bce else ops.EPI_LOSS_MSE,
                        h=T[i % 3], loss_sum=sums[1:2] if bce else sums[0:1])
        torch.cuda.synchronize()
    run(2); lib.mmvae_debug_stamps(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record(); run(reps); e1.record(); torch.cuda.synchronize()
    lib.mmvae_debug_stamps(buf, 1)
    s = [int(x) for x in buf]
    steps, waves = s[4], s[5]
    nk = (K + 63) // 64
    tiles = steps / nk
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / reps:.1f} us/launch (stamped); {waves // reps} sampled waves, {steps / waves:.0f} K steps, {tiles / waves:.1f} tiles per wave")
    for nm, v in zip(["wait for the own DMA", "barrier wait", "DMA issue of the next step", "fragment reads + 32 MFMA"], s[:4]):
        print(f"   {nm:32s} {v / steps:8.0f} cycles/K-step")
    print(f"   whole kernel {s[6] / waves:9.0f} cycles/wave: main loop {sum(s[:4]) / waves:9.0f}, epilogues {s[9] / waves:9.0f} = {s[9] / tiles:7.0f} per tile")
