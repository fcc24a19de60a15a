#!/usr/bin/env python3
"""fp32-A NT GEMM (EncoderB.L0.fwd: y = b W^T + bias, bf16 out, BatchNorm statistics): LDS-DMA form (gemm_nt2.h, AT = float) against the
first generation's register staging (mmvae_set_tuning key 5), interleaved; outputs must be bit-identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops, _lib as L
from mmvae.ops import PREC_BF16
lib = L.load()
dev, M = "cuda", int(os.environ.get("M", 65536))


def t(f, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, N, K, stats in (("EncoderB.L0.fwd 512<-572 +stats", 512, 572, True), ("512<-572 plain", 512, 572, False), ("256<-1024 +stats", 256, 1024, True), ("768<-300", 768, 300, False)):
    A = [torch.rand(M, K, device=dev) for _ in range(3)]
    W = torch.randn(N, K, device=dev) / K ** 0.5
    pl = ops.PreparedLinear([W], [torch.randn(N, device=dev) * 0.1], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    outs = []
    i = [0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    st = torch.zeros(2, N, dtype=torch.float64, device=dev) if stats else None
    def f(): i[0] += 1; ops.gemm_nt(PREC_BF16, A[i[0] % 3], pl.w, N, K, out, bias=pl.bias, stats=st)
    for on in (0, 1):
        lib.mmvae_set_tuning(5, on); i[0] = 0
        if st is not None: st.zero_()
        f(); torch.cuda.synchronize()
        outs.append((out.clone(), None if st is None else st.clone()))
    same = torch.equal(outs[0][0], outs[1][0])
    sdev = 0.0 if st is None else ((outs[0][1] - outs[1][1]).abs().max() / outs[0][1].abs().max()).item()
    res = {0: [], 1: []}
    for rnd in range(5):
        for on in (0, 1):
            lib.mmvae_set_tuning(5, on)
            if rnd == 0: t(f, 3)
            res[on].append(t(f))
    print(f"{name:34s} registers: med {sorted(res[0])[2]:6.1f}   LDS-DMA: med {sorted(res[1])[2]:6.1f} min {min(res[1]):6.1f}   outputs identical: {same}, stats dev {sdev:.1e}", flush=True)
lib.mmvae_set_tuning(5, 1)
