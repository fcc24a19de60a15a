#!/usr/bin/env python3
"""EncoderB's second Linear forward (bf16 A through the BatchNorm + ReLU + Dropout prologue, N = 256, K = 512, B = 65 536) in isolation;
ABL=<name> selects an experimental library built by tools/abl_ntp.sh.  SHAPE=plain: DecoderB.L1.fwd (plain bf16 A, ReLU, N = 512, K = 256)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("ABL"):
    os.environ["MMVAE_LIB_PATH"] = os.path.join(ROOT, "vae-los-angeles_amd", "mmvae", f"libmmvae_{os.environ['ABL']}.so")
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops
from mmvae.ops import PREC_BF16
PLAIN = os.environ.get("SHAPE") == "plain"
dev, M, N, K = ("cuda", 65536, 512, 256) if PLAIN else ("cuda", 65536, 256, 512)
A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(3)]
pro = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3, (torch.rand(M, K, device=dev) > 0.1).to(torch.uint8), 1.0 / 0.9)
W = torch.randn(N, K, device=dev) / K ** 0.5
pl = ops.PreparedLinear([W], [torch.zeros(N, device=dev)], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); st = torch.zeros(2, N, dtype=torch.float64, device=dev)
fn = (lambda i: ops.gemm_nt(PREC_BF16, A[i % 3], pl.w, N, K, out, bias=pl.bias, act=ops.ACT_RELU)) if PLAIN else \
     (lambda i: ops.gemm_nt(PREC_BF16, A[i % 3], pl.w, N, K, out, bias=pl.bias, stats=st, prologue=pro))
for r in range(3):
    fn(0); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(12): fn(i)
    e.record(); torch.cuda.synchronize()
    print(f"{'DecoderB.L1.fwd' if PLAIN else 'EncoderB.L1.fwd'} {os.environ.get('ABL', 'product')}: {s.elapsed_time(e) / 12 * 1e3:.1f} us", flush=True)
