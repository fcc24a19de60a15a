#!/usr/bin/env python3
"""Can the data-parallel step (RCCL all-reduce inside backward) be captured as a hipGraph?  Run on ONE GPU with a
world of 1 (exercises RCCL + capture; correctness of the N>1 path itself is covered by the gloo tests)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from mmvae import parallel
from mmvae.graphs import GraphedTrainStep
from mmvae.optim import FusedAdamW
from src.models import MultiModalVAE
torch.manual_seed(0)
B = 8192
m = MultiModalVAE(782, 572, 24, 20).cuda()
parallel.broadcast_parameters(m); parallel.attach(m, overlap=True)
opt = FusedAdamW(m.parameters(), lr=5e-4, weight_decay=1e-5)
a = torch.randn(B, 782).abs().cuda(); b = torch.rand(B, 572).cuda(); s = torch.randint(0, 24, (B,)).cuda()
gs = GraphedTrainStep(m, opt, a, b, s, warmup=2)
for _ in range(5):
    gs()
print("losses", gs.losses())
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL capture OK")
