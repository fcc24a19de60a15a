#!/usr/bin/env python3
"""Which Python call sites zero-fill device memory during ONE captured-form training step (launch diet: every fill is a launch)?
Patches torch.zeros / zeros_like / Tensor.zero_ / Tensor.fill_ while GraphedTrainStep.run_eager() runs and prints size + caller."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vae-los-angeles_amd")]
import torch
from mmvae.graphs import GraphedTrainStep
from mmvae.optim import FusedAdamW
from src.models import MultiModalVAE
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MultiModalVAE(bench.A, bench.D, bench.S, bench.L).to(dev).set_precision("bf16").train()
opt = FusedAdamW(model.parameters(), lr=5e-4, weight_decay=1e-5)
a, b, site = bench.synth_batch(65536, 0, dev)
gs = GraphedTrainStep(model, opt, a, b, site, warmup=2)
log = []


def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "mmvae" in fr.filename or "src/" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "?"


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*args, **kw):
        out = orig(*args, **kw)
        t = out if isinstance(out, torch.Tensor) else args[0]
        if t.is_cuda:
            log.append((name, t.numel() * t.element_size(), str(t.dtype), where()))
        return out
    setattr(mod, name, f)
    return orig


saved = [(torch, "zeros", wrap(torch, "zeros")), (torch, "zeros_like", wrap(torch, "zeros_like")),
         (torch.Tensor, "zero_", wrap(torch.Tensor, "zero_")), (torch.Tensor, "fill_", wrap(torch.Tensor, "fill_"))]
gs.run_eager()
torch.cuda.synchronize()
for m, n, o in saved:
    setattr(m, n, o)
for e in log:
    print(e)
