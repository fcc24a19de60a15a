#!/usr/bin/env python3
"""Launch one hot-path GEMM shape in isolation (same shapes/dtypes as in bench.py at B=65536) so that
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE rows all belong to that launch:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/run_dominant.py DecoderB.L2.dW
Inputs are > 256 MiB in total or rotated over several buffers so the Infinity Cache cannot hide re-reads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops
from mmvae.ops import PREC_BF16

dev, M = "cuda", 65536
tag = sys.argv[1] if len(sys.argv) > 1 else "DecoderB.L2.dW"
reps, nbuf = 6, 4


def bf(rows, cols):
    return [torch.randn(rows, ops.ceil_to(cols, 8), device=dev).bfloat16() for _ in range(nbuf)]


if tag == "DecoderB.L2.dW":            # dW[572,512] += g_b^T H2
    P, Q = bf(M, 572), bf(M, 512)
    dw, db = torch.zeros(572, 512, device=dev), torch.zeros(572, device=dev)
    slab = torch.empty(1 << 24, device=dev)             # as in the engine: partial tiles of the batch splits go to a slab workspace
    run = lambda i: ops.gemm_tn(PREC_BF16, P[i % nbuf], Q[i % nbuf], dw, db, 572, 512, slab=slab)
elif tag == "EncoderB.L0.dW":          # dW[512,572] += dy^T b with dy = BatchNorm-backward correction of d applied on the load (b fp32)
    P = bf(M, 512); Y = bf(M, 512); Q = [torch.rand(M, 572, device=dev) for _ in range(nbuf)]
    f = lambda: torch.rand(512, device=dev) + 0.5
    mean, rstd, coef = f() - 1.0, f(), torch.stack([f(), (f() - 1.0) * 0.1, (f() - 1.0) * 0.1]).contiguous()
    dw, db = torch.zeros(512, 572, device=dev), torch.zeros(512, device=dev)
    slab = torch.empty(1 << 24, device=dev)
    run = lambda i: ops.gemm_tn(PREC_BF16, P[i % nbuf], Q[i % nbuf], dw, db, 512, 572, p_prologue=(Y[i % nbuf], mean, rstd, coef), slab=slab)
elif tag in ("EncoderB.L0.fwd", "EncoderA.L0.fwd"):   # y[B,512] = b W^T / y[B,128] = a W^T (+ BN statistics): the north_star's named GEMMs
    N, K = (512, 572) if tag == "EncoderB.L0.fwd" else (128, 782)
    A = [torch.rand(M, K, device=dev) for _ in range(nbuf)]
    W = torch.randn(N, K, device=dev) / K ** 0.5; bias = torch.zeros(N, device=dev)
    pl = ops.PreparedLinear([W], [bias], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); st = torch.zeros(2, N, dtype=torch.float64, device=dev)
    run = lambda i: ops.gemm_nt(PREC_BF16, A[i % nbuf], pl.w, N, K, out, bias=pl.bias, stats=st)
elif tag in ("DecoderB.L2.fwd", "DecoderA.L1.fwd"):   # the decoders' last layers as the training step runs them: reconstruction loss in the epilogue
    N, K, bce = (572, 512, True) if tag == "DecoderB.L2.fwd" else (782, 128, False)
    A = bf(M, K)
    W = torch.randn(N, K, device=dev) / K ** 0.5; bias = torch.zeros(N, device=dev)
    pl = ops.PreparedLinear([W], [bias], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    T = [torch.rand(M, N, device=dev) for _ in range(2)]
    g = torch.empty(M, ops.ceil_to(N, 8), dtype=torch.bfloat16, device=dev)
    sums = torch.zeros(5, dtype=torch.float64, device=dev)
    run = lambda i: ops.gemm_nt(PREC_BF16, A[i % nbuf], pl.w, N, K, g, bias=pl.bias, epilogue=ops.EPI_LOSS_BCE_LOGIT if bce else ops.EPI_LOSS_MSE,
                                h=T[i % 2], loss_sum=sums[1:2] if bce else sums[0:1])
elif tag == "DecoderB.L2.fwd.store":   # recon_b[B,572] = sigmoid(H2 W^T + b), fp32 out (the public forward)
    A = bf(M, 512)
    W = torch.randn(572, 512, device=dev) / 22; bias = torch.zeros(572, device=dev)
    pl = ops.PreparedLinear([W], [bias], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    out = torch.empty(M, 572, device=dev)
    run = lambda i: ops.gemm_nt(PREC_BF16, A[i % nbuf], pl.w, 572, 512, out, bias=pl.bias, act=ops.ACT_SIGMOID)
elif tag in ("EncoderB.L0.dX", "EncoderB.L1.dX"):   # dX GEMM with the BatchNorm-backward epilogue (store d + column statistics)
    N, K = (512, 256) if tag == "EncoderB.L0.dX" else (256, 40)
    A = bf(M, K) if tag == "EncoderB.L0.dX" else [torch.randn(M, K, device=dev) for _ in range(nbuf)]
    W = torch.randn(N, K, device=dev) / K ** 0.5
    pl = ops.PreparedLinear([W], [torch.zeros(N, device=dev)], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    Y = bf(M, N)
    mask = [(torch.rand(M, N, device=dev) > 0.1).to(torch.uint8) for _ in range(nbuf)]
    f = lambda: torch.rand(N, device=dev) + 0.5
    sc, sh, mu, rs = f(), f() - 1.0, f() - 1.0, f()
    d = torch.empty(M, N, dtype=torch.bfloat16, device=dev); st = torch.zeros(2, N, dtype=torch.float64, device=dev)
    run = lambda i: ops.gemm_nt(PREC_BF16, A[i % nbuf], pl.w, N, K, d, epilogue=ops.EPI_BN_BWD, h=Y[i % nbuf],
                                bn=(sc, sh, mu, rs, mask[i % nbuf], 1.0 / 0.9), bn_phase=2, stats=st)
elif tag == "vae_loss":                # the fused loss pass of the step: MSE(782) + BCE(572) + CE(24) + KL(20), bf16 gradients
    ra = [torch.randn(M, 782, device=dev) for _ in range(nbuf)]; a = [torch.randn(M, 782, device=dev).abs() for _ in range(2)]
    rb = [torch.rand(M, 572, device=dev) for _ in range(nbuf)]; b = [torch.rand(M, 572, device=dev) for _ in range(2)]
    rc, site = torch.randn(M, 24, device=dev), torch.randint(0, 24, (M,), device=dev)
    mu, lv = torch.randn(M, 20, device=dev), torch.randn(M, 20, device=dev)
    ga = torch.empty(M, 784, dtype=torch.bfloat16, device=dev); gb = torch.empty(M, 576, dtype=torch.bfloat16, device=dev)
    gc, gm, gl = torch.empty_like(rc), torch.empty_like(mu), torch.empty_like(lv)
    sums = torch.zeros(5, dtype=torch.float64, device=dev)
    run = lambda i: ops.vae_loss(M, recon_a=ra[i % nbuf], a=a[i % 2], recon_b=rb[i % nbuf], b=b[i % 2], logits=rc, site=site, mu=mu, logvar=lv,
                                 sums=sums, g_a=ga, g_b=gb, grad_b_wrt_logit=True, g_c=gc, g_mu=gm, g_lv=gl)
else:
    raise SystemExit(f"unknown tag {tag}")
for i in range(reps):
    run(i)
torch.cuda.synchronize()
print("done", tag)
