#!/usr/bin/env python3
"""The small / medium non-GEMM launches of the step in isolation (B = 65 536): where their time goes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops
dev, B, Ld, S = "cuda", 65536, 20, 24


def t(f, n=20):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


g_mu, g_lv = torch.randn(B, Ld, device=dev), torch.randn(B, Ld, device=dev)
dzs = [torch.randn(B, 24, device=dev)[:, :Ld] for _ in range(3)]
eps, lv = torch.randn(B, Ld, device=dev), torch.randn(B, Ld, device=dev)
d_heads = torch.empty(B, 2 * Ld, device=dev)
d_table = torch.zeros(S, 2 * Ld, device=dev)
site = torch.randint(0, S, (B,), device=dev)
print("fuse_reparam_bwd  3 dz, with table scatter :", round(t(lambda: ops.fuse_reparam_bwd(B, Ld, 3, g_mu, g_lv, dzs, eps, lv, d_heads, d_table, site)), 1))
d_table8 = torch.zeros(8, S, 2 * Ld, device=dev)
print("fuse_reparam_bwd  3 dz, table in 8 copies  :", round(t(lambda: ops.fuse_reparam_bwd(B, Ld, 3, g_mu, g_lv, dzs, eps, lv, d_heads, d_table8, site)), 1))
print("fuse_reparam_bwd  3 dz, no table           :", round(t(lambda: ops.fuse_reparam_bwd(B, Ld, 2, g_mu, g_lv, dzs, eps, lv, d_heads, None, None)), 1))
print("fuse_reparam_bwd  1 dz, no table           :", round(t(lambda: ops.fuse_reparam_bwd(B, Ld, 2, g_mu, g_lv, dzs[:1], eps, lv, d_heads, None, None)), 1))
widths = [128, 512, 256]
total = sum(ops.ceil_to(B * w, 16) for w in widths)
buf = torch.empty(total, dtype=torch.uint8, device=dev); e2 = torch.empty(B, Ld, device=dev)
off = torch.zeros(16384, dtype=torch.int64, device=dev)
print("noise  58.7 M mask bytes + 1.3 M normals    :", round(t(lambda: ops.noise(buf, e2, 0.9, 1234, 0, off, advance=True)), 1))
print("noise  masks only                            :", round(t(lambda: ops.noise(buf, None, 0.9, 1234, 0, off, advance=True)), 1))
rc, gc = torch.randn(B, S, device=dev), torch.empty(B, S, device=dev)
mu_, lv_ = torch.randn(B, Ld, device=dev), torch.randn(B, Ld, device=dev)
gm, gl = torch.empty_like(mu_), torch.empty_like(lv_)
sums = torch.zeros(5, dtype=torch.float64, device=dev)
print("vae_loss  class + KL terms                  :", round(t(lambda: ops.vae_loss(B, logits=rc, site=site, mu=mu_, logvar=lv_, sums=sums, g_c=gc, g_mu=gm, g_lv=gl)), 1))
print("vae_loss  class term only                    :", round(t(lambda: ops.vae_loss(B, logits=rc, site=site, sums=sums, g_c=gc)), 1))
print("vae_loss  KL term only                       :", round(t(lambda: ops.vae_loss(B, mu=mu_, logvar=lv_, sums=sums, g_mu=gm, g_lv=gl)), 1))
