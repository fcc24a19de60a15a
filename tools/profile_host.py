#!/usr/bin/env python3
"""cProfile of the eager (Python-issued) training step: where the host time goes."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae.optim import FusedAdamW
from src.models import MultiModalVAE
from src.utils import vae_loss
torch.manual_seed(0)
B = int(os.environ.get("B", 65536))
m = MultiModalVAE(782, 572, 24, 20).cuda().train()
opt = FusedAdamW(m.parameters(), lr=5e-4, weight_decay=1e-5)
a = torch.randn(B, 782).abs().cuda(); b = torch.rand(B, 572).cuda(); s = torch.randint(0, 24, (B,)).cuda()
def step():
    ra, rb, rc, mu, lv = m(a=a, b=b, site=s)
    loss, *_ = vae_loss(ra, a, rb, b, rc, s, mu, lv)
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
