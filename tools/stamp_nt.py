#!/usr/bin/env python3
"""Where does a K step of the NT GEMM go?  Runs one bf16 shape through the DIAGNOSTIC library
(make -C vae-los-angeles_amd/csrc STAMP=1 -> libmmvae_stamp.so, selected via MMVAE_LIB_PATH) whose kernels
stamp s_memtime at the drain points of every K step, and prints cycles per K step per wave:

    python tools/stamp_nt.py [N] [K] [M]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MMVAE_LIB_PATH"] = os.path.join(ROOT, "vae-los-angeles_amd", "mmvae", "libmmvae_stamp.so")
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch  # noqa: E402
from mmvae import _lib as L, ops  # noqa: E402
from mmvae.ops import PREC_BF16  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
M = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
dev = "cuda"
lib = L.load()
lib.mmvae_debug_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
SRC_F32 = os.environ.get("SRC") == "f32"
STATS = os.environ.get("STATS") == "1"
A = [torch.randn(M, K, device=dev) if SRC_F32 else torch.randn(M, K, device=dev).bfloat16() for _ in range(3)]
stats = torch.zeros(2, N, dtype=torch.float64, device=dev) if STATS else None
W = torch.randn(N, K, device=dev) / 30
bias = torch.zeros(N, device=dev)
pl = ops.PreparedLinear([W], [bias], PREC_BF16, dev)
ops.WeightPrep([pl], dev).run()
out = torch.empty(M, ops.ceil_to(N, 8), dtype=torch.bfloat16, device=dev)
buf = (C.c_uint64 * 12)()


def run(reps):
    for i in range(reps):
        ops.gemm_nt(PREC_BF16, A[i % 3], pl.w, N, K, out, bias=pl.bias, stats=stats)
    torch.cuda.synchronize()


for wide in (1, 0):
    lib.mmvae_set_tuning(0, 0 if wide else 1 << 30)
    run(2)
    lib.mmvae_debug_stamps(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record(); run(reps); e1.record(); torch.cuda.synchronize()
    lib.mmvae_debug_stamps(buf, 1)
    s = [int(x) for x in buf]
    steps, waves = s[4], s[5]
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"N={N} K={K} M={M} {'wide 128x256' if wide else 'narrow 128x128'}: {us:.1f} us/launch (stamped build), {waves // reps} waves, {steps // waves} K steps/wave")
    names = ["reads(s1) + mma(s0) issue", "stage: vmcnt wait + ds_write + drain", "barrier wait", "fetch + reads(s0) + mma(s1) issue"]
    if os.environ.get("NT2") == "1":          # second-generation kernel (gemm_nt2.h): plain bf16 A, K > 64
        names = ["wait for the own DMA (vmcnt)", "barrier wait", "DMA issue of the next step", "fragment reads + 32 MFMA"]
    tot = sum(s[:4])
    for nm, v in zip(names, s[:4]):
        print(f"   {nm:40s} {v / steps:8.1f} cycles/K-step  {100.0 * v / tot:5.1f} %")
    print(f"   {'  of stage: wait for the global loads':40s} {s[7] / steps:8.1f} cycles/K-step")
    print(f"   {'sum':40s} {tot / steps:8.1f} cycles/K-step;  whole kernel {s[6] / waves:9.0f} cycles/wave, main loop {tot / waves:9.0f}, before it {s[8] / waves:7.0f}, epilogue {s[9] / waves:7.0f}")
