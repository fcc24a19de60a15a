#!/usr/bin/env python3
"""The scaled omics widths of BASELINE.json (RNA 20 000 / DNA 27 000 features, latent 128) at the bench batch of 65 536 rows on ONE GPU:
captured training step (fp32 inputs of 5.2 / 7.1 GB resident in HBM), ms per step.  A data point for DESIGN.md, not the bench metric."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae.graphs import GraphedTrainStep
from mmvae.optim import FusedAdamW
from src.models import MultiModalVAE
A, D, S, L, B = 20000, 27000, 24, int(os.environ.get("LATENT", 128)), int(os.environ.get("B", 65536))
dev = "cuda"
torch.manual_seed(0)
model = MultiModalVAE(A, D, S, L).to(dev).set_precision("bf16").train()
opt = FusedAdamW(model.parameters(), lr=5e-4, weight_decay=1e-5)
a = torch.randn(B, A, device=dev); b = torch.rand(B, D, device=dev); site = torch.randint(0, S, (B,), device=dev)
gs = GraphedTrainStep(model, opt, a, b, site, beta=1e-3, gamma=1.0, warmup=2)
for _ in range(3):
    gs()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    gs()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
flops = 6.0 * B * (A * 128 + D * 512 + 512 * 256 + 2 * L * (128 + 256) + L * (128 + 256 + 64) + 128 * A + 256 * 512 + 512 * D + 64 * S)
print(f"scaled widths A={A} D={D} latent={L} B={B}: {ms:.1f} ms/step = {B / ms * 1e3 / 1e6:.2f} M samples/s, ~{flops / ms / 1e9:.0f} TFLOP/s; losses {gs.losses()}; "
      f"peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
