#!/usr/bin/env python3
"""Time the parts of mmvae_vae_loss alone (B=65536, workload widths, cold: 1 GiB of unrelated writes before each launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae import ops
B, A, D, S, Lz = 65536, 782, 572, 24, 20
dev = "cuda"
xa, ta = torch.rand(B, A, device=dev), torch.rand(B, A, device=dev)
xb, tb = torch.rand(B, D, device=dev), (torch.rand(B, D, device=dev) > 0.5).float()
ga = torch.empty(B, 784, dtype=torch.bfloat16, device=dev); gb = torch.empty(B, 576, dtype=torch.bfloat16, device=dev)
lg = torch.randn(B, S, device=dev); site = torch.randint(0, S, (B,), device=dev); gc = torch.empty(B, S, device=dev)
mu, lv = torch.randn(B, Lz, device=dev), torch.randn(B, Lz, device=dev); gmu, glv = torch.empty_like(mu), torch.empty_like(lv)
big = torch.empty(1 << 28, device=dev)
sums = torch.zeros(5, dtype=torch.float64, device=dev)
parts = {
    "mse": dict(recon_a=xa, a=ta, g_a=ga),
    "bce": dict(recon_b=xb, b=tb, g_b=gb, grad_b_wrt_logit=True),
    "ce": dict(logits=lg, site=site, g_c=gc),
    "kl": dict(mu=mu, logvar=lv, g_mu=gmu, g_lv=glv),
}
parts["mse+bce"] = {**parts["mse"], **parts["bce"]}
parts["ce+kl"] = {**parts["ce"], **parts["kl"]}
parts["all"] = {k: v for p in (parts["mse"], parts["bce"], parts["ce"], parts["kl"]) for k, v in p.items()}
bytes_ = {"mse": B * A * 10, "bce": B * D * 10, "ce": B * S * 8, "kl": B * Lz * 16}
bytes_["mse+bce"] = bytes_["mse"] + bytes_["bce"]
bytes_["ce+kl"] = bytes_["ce"] + bytes_["kl"]
bytes_["all"] = bytes_["mse+bce"] + bytes_["ce"] + bytes_["kl"]
for name, kw in parts.items():
    ts = []
    for _ in range(7):
        big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.vae_loss(B, sums=sums, **kw); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[3]
    print(f"{name:4s} {t:7.1f} us   {bytes_[name] / t / 1e6:5.2f} TB/s")
