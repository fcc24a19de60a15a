#!/usr/bin/env python3
"""Where does a batch step of the wide-tile dW kernel go?  Diagnostic library (make -C vae-los-angeles_amd/csrc STAMP=1):
    python tools/stamp_tnw.py [bn|plain] [N] [K] [M]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MMVAE_LIB_PATH"] = os.path.join(ROOT, "vae-los-angeles_amd", "mmvae", "libmmvae_stamp.so")
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import _lib as L, ops
from mmvae.ops import PREC_BF16
kind = sys.argv[1] if len(sys.argv) > 1 else "plain"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 572
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
M = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
dev = "cuda"
lib = L.load()
lib.mmvae_debug_stamps_tnw.argtypes = [C.POINTER(C.c_uint64), C.c_int]
slab = torch.empty(1 << 25, device=dev)
Np, Kp = ops.ceil_to(N, 8), ops.ceil_to(K, 8)
P = [torch.randn(M, Np, device=dev).bfloat16() for _ in range(3)]
dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
if kind == "bn":
    Y = [torch.randn(M, Np, device=dev).bfloat16() for _ in range(3)]
    Q = [torch.randn(M, K, device=dev) for _ in range(3)]
    mean, rstd = torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5
    coef = torch.stack([torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.01, torch.randn(N, device=dev) * 0.01]).contiguous()
    def f(j): ops.gemm_tn(PREC_BF16, P[j], Q[j], dw, db, N, K, p_prologue=(Y[j], mean, rstd, coef), slab=slab)
else:
    Q = [torch.randn(M, Kp, device=dev).bfloat16() for _ in range(3)]
    def f(j): ops.gemm_tn(PREC_BF16, P[j], Q[j], dw, db, N, K, slab=slab)
buf = (C.c_uint64 * 12)()
for i in range(3):
    f(i % 3)
torch.cuda.synchronize()
lib.mmvae_debug_stamps_tnw(buf, 1)
reps = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(reps):
    f(i % 3)
e1.record(); torch.cuda.synchronize()
lib.mmvae_debug_stamps_tnw(buf, 1)
s = [int(x) for x in buf]
steps, wgs = s[4], s[5]
print(f"{kind} N={N} K={K} M={M}: {e0.elapsed_time(e1) * 1e3 / reps:.1f} us/call incl. reduce (stamped build); {wgs // reps} stamped workgroups, {steps // max(wgs, 1)} steps each")
print(f"   whole kernel {s[6] / max(wgs, 1):9.0f} cycles per workgroup, epilogue {s[7] / max(wgs, 1):8.0f}")
tot = sum(s[:4])
for nm, v in zip(["wait for the own DMA (vmcnt)", "barrier", "DMA issue", "fragments + MFMA"], s[:4]):
    print(f"   {nm:32s} {v / max(steps, 1):8.1f} cycles/step {100.0 * v / max(tot, 1):5.1f} %")
