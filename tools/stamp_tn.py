#!/usr/bin/env python3
"""Where does a batch step of the TN (dW) GEMM go?  Diagnostic library only (make -C vae-los-angeles_amd/csrc STAMP=1).
    python tools/stamp_tn.py [N] [K] [M]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MMVAE_LIB_PATH"] = os.path.join(ROOT, "vae-los-angeles_amd", "mmvae", "libmmvae_stamp.so")
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch  # noqa: E402
from mmvae import _lib as L, ops  # noqa: E402
from mmvae.ops import PREC_BF16  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 572
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
M = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
dev = "cuda"
lib = L.load()
lib.mmvae_debug_stamps_tn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
P = [torch.randn(M, ops.ceil_to(N, 8), device=dev).bfloat16() for _ in range(3)]
Q = [torch.randn(M, ops.ceil_to(K, 8), device=dev).bfloat16() for _ in range(3)]
dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
slab = torch.empty(1 << 25, device=dev)            # as in the engine: partial tiles of the batch splits go to a slab workspace
buf = (C.c_uint64 * 12)()


def run(reps):
    for i in range(reps):
        ops.gemm_tn(PREC_BF16, P[i % 3], Q[i % 3], dw, db, N, K, slab=slab)
    torch.cuda.synchronize()


run(2)
lib.mmvae_debug_stamps_tn(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record(); run(reps); e1.record(); torch.cuda.synchronize()
lib.mmvae_debug_stamps_tn(buf, 1)
s = [int(x) for x in buf]
steps, waves = s[4], s[5]
print(f"TN N={N} K={K} M={M}: {e0.elapsed_time(e1) * 1e3 / reps:.1f} us/launch (stamped build), {waves // reps} sampled waves, {steps // waves} batch steps/wave")
names = ["fragment step 0 (tr reads + 16 MFMA)", "stage (vmcnt wait + ds_write)", "fragment step 1 + fetch issue + drain", "barrier wait"]
tot = sum(s[:4])
for nm, v in zip(names, s[:4]):
    print(f"   {nm:42s} {v / steps:8.1f} cycles/step  {100.0 * v / tot:5.1f} %")
print(f"   {'  of stage: wait for the global loads':42s} {s[7] / steps:8.1f} cycles/step")
print(f"   sum {tot / steps:8.1f} cycles/step; whole kernel {s[6] / waves:9.0f} cycles/wave, loop {tot / waves:9.0f}, before {s[8] / waves:7.0f}, after {s[9] / waves:7.0f}")
