#!/usr/bin/env python3
"""A/B of the wide-tile dW kernel (gemm_tn_wide.hip, mmvae_set_tuning key 4) against the 128 x 128 kernel on the large weight
gradients of the step, interleaved in one process; also prints the largest deviation between the two results."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops, _lib as L
from mmvae.ops import PREC_BF16
lib = L.load()
dev, M = "cuda", int(os.environ.get("M", 65536))
slab = torch.empty(1 << 25, device=dev)


def make(N, K, kind):
    Np, Kp = ops.ceil_to(N, 8), ops.ceil_to(K, 8)
    P = [torch.randn(M, Np, device=dev).bfloat16() for _ in range(3)]
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    i = [0]
    if kind == "bn":
        Y = [torch.randn(M, Np, device=dev).bfloat16() for _ in range(3)]
        Q = [torch.randn(M, K, device=dev) for _ in range(3)]
        mean, rstd = torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5
        coef = torch.stack([torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.01, torch.randn(N, device=dev) * 0.01]).contiguous()
        def f(): i[0] += 1; j = i[0] % 3; ops.gemm_tn(PREC_BF16, P[j], Q[j], dw, db, N, K, p_prologue=(Y[j], mean, rstd, coef), slab=slab)
    else:
        Q = [torch.randn(M, Kp, device=dev).bfloat16() for _ in range(3)]
        def f(): i[0] += 1; j = i[0] % 3; ops.gemm_tn(PREC_BF16, P[j], Q[j], dw, db, N, K, slab=slab)
    return f, dw, db, i


def t(f, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


shapes = [("EncB.L0.dW 512x572 bn x f32", 512, 572, "bn"), ("EncA.L0.dW 128x782 bn x f32", 128, 782, "bn"), ("DecB.L2.dW 572x512 bf16", 572, 512, "plain"),
          ("256x1024 bf16", 256, 1024, "plain")]
for name, N, K, kind in shapes:
    f, dw, db, i = make(N, K, kind)
    outs = []
    for on in (0, 1):
        lib.mmvae_set_tuning(4, on); dw.zero_(); db.zero_(); i[0] = 0; f(); torch.cuda.synchronize()
        outs.append((dw.clone(), db.clone()))
    dev_w = ((outs[0][0] - outs[1][0]).abs().max() / outs[0][0].abs().max()).item()
    dev_b = ((outs[0][1] - outs[1][1]).abs().max() / outs[0][1].abs().max().clamp_min(1e-30)).item()
    res = {0: [], 1: []}
    for rnd in range(5):
        for on in (0, 1):
            lib.mmvae_set_tuning(4, on)
            if rnd == 0:
                t(f, 3)
            res[on].append(t(f))
    print(f"{name:30s} 128x128: med {sorted(res[0])[2]:6.1f}  wide: med {sorted(res[1])[2]:6.1f} min {min(res[1]):6.1f}   dW dev {dev_w:.1e} db dev {dev_b:.1e}", flush=True)
lib.mmvae_set_tuning(4, 1)
