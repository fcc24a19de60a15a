#!/usr/bin/env python3
"""A/B of the wave-specialised NT kernel (gemm_ntp.h, mmvae_set_tuning key 8) against the tile kernels on the forward first layers at
B = 65 536, interleaved rounds in one process, inputs rotated over buffers that exceed the Infinity Cache."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("ABL"):        # timing-only ablation library (tools/abl_ntp.sh): wrong results by construction
    os.environ["MMVAE_LIB_PATH"] = os.path.join(ROOT, "vae-los-angeles_amd", "mmvae", f"libmmvae_{os.environ['ABL']}.so")
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops, _lib
from mmvae.ops import PREC_BF16

dev, M = "cuda", int(os.environ.get("M", 65536))
lib = _lib.load()


def case(N, K, nbuf=3):
    A = [torch.rand(M, K, device=dev) for _ in range(nbuf)]
    W = torch.randn(N, K, device=dev) / K ** 0.5
    pl = ops.PreparedLinear([W], [torch.zeros(N, device=dev)], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    st = torch.zeros(2, N, dtype=torch.float64, device=dev)
    return lambda i: ops.gemm_nt(PREC_BF16, A[i % nbuf], pl.w, N, K, out, bias=pl.bias, stats=st)


def timeit(fn, n=12):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, N, K in (("EncoderB.L0.fwd", 512, 572), ("EncoderA.L0.fwd", 128, 782)):
    fn = case(N, K)
    byts = M * K * 4 + M * N * 2 + N * K * 2
    res = {0: [], 1: []}
    for r in range(int(os.environ.get("ROUNDS", 5))):
        for on in ((1,) if os.environ.get("ABL") else (0, 1)):
            lib.mmvae_set_tuning(8, on)
            fn(0); torch.cuda.synchronize()
            res[on].append(timeit(fn))
    lib.mmvae_set_tuning(8, 1)
    for on in ((1,) if os.environ.get("ABL") else (0, 1)):
        t = sorted(res[on])
        print(f"{name} ntp={on}: median {t[len(t) // 2]:.1f} us  min {t[0]:.1f} us  -> {byts / t[len(t) // 2] / 1e6:.2f} TB/s of {byts / 1e6:.0f} MB", flush=True)
