#!/usr/bin/env python3
"""Where do the waves of the wave-specialised NT kernel (gemm_ntp.h) spend their cycles?  Diagnostic library only
(make -C vae-los-angeles_amd/csrc STAMP=1 -> libmmvae_stamp.so): s_memtime stamps per role, summed over a sample of workgroups.

    python tools/stamp_ntp.py [N] [K] [M]
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
K = int(sys.argv[2]) if len(sys.argv) > 2 else 572
M = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
dev = "cuda"
lib = L.load()
lib.mmvae_debug_ntp_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
PRO = os.environ.get("PRO") == "1"      # bf16 A through the producers' BatchNorm + ReLU + Dropout prologue (EncoderB's second Linear: N=256 K=512)
A = [torch.randn(M, K, device=dev).bfloat16() if PRO else torch.rand(M, K, device=dev) for _ in range(3)]
pro = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3, (torch.rand(M, K, device=dev) > 0.1).to(torch.uint8), 1.0 / 0.9) if PRO else None
stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
W = torch.randn(N, K, device=dev) / 30
pl = ops.PreparedLinear([W], [torch.zeros(N, device=dev)], PREC_BF16, dev)
ops.WeightPrep([pl], dev).run()
out = torch.empty(M, ops.ceil_to(N, 8), dtype=torch.bfloat16, device=dev)
buf = (C.c_uint64 * 24)()


def run(reps):
    for i in range(reps):
        ops.gemm_nt(PREC_BF16, A[i % 3], pl.w, N, K, out, bias=pl.bias, stats=stats, prologue=pro)
    torch.cuda.synchronize()


run(2)
lib.mmvae_debug_ntp_stamps(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record(); run(reps); e1.record(); torch.cuda.synchronize()
lib.mmvae_debug_ntp_stamps(buf, 1)
s = [int(x) for x in buf]
wgs = max(s[6], 1)
steps, tiles = s[3] / wgs, s[4] / wgs
print(f"N={N} K={K} M={M}: {e0.elapsed_time(e1) * 1e3 / reps:.1f} us/launch (stamped build); per sampled workgroup: {steps:.0f} K steps, {tiles:.0f} tiles, "
      f"{s[5] / wgs:.0f} cycles in the kernel")
print(f"  consumer   : barrier wait {s[0] / wgs / steps:7.0f} /step   reads+MFMA {s[1] / wgs / steps:7.0f} /step   epilogue {s[2] / wgs / max(tiles, 1):7.0f} /tile")
names = ["barrier wait", "wait for the set's loads", "v_cvt_pk (fp32 -> bf16)", "ds_write issue", "W DMA issue", "A load issue", "wait for W of the next step"]
print("  producer   : " + "   ".join(f"{n} {s[8 + i] / wgs / steps:6.0f}" for i, n in enumerate(names)) + "   (cycles per K step)")
