#!/usr/bin/env python3
"""Time the store-epilogue NT shapes under the NT2 ablation bits (1 = no DMA, 2 = no MFMA, 4 = no epilogue): where does a tile's time go?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools")]
os.environ.setdefault("M", "65536")
import torch
from bench_gemm_lib import nt
print("ablate", os.environ.get("MMVAE_NT2_ABLATE", "0"),
      "| DecB.L2.fwd 572x512 f32:", round(nt(572, 512, out_dtype=torch.float32), 1),
      "| DecA.L1.fwd 782x128 f32:", round(nt(782, 128, out_dtype=torch.float32), 1),
      "| DecB.L1.fwd 512x256 bf16:", round(nt(512, 256), 1),
      "| 512x1024 bf16:", round(nt(512, 1024), 1))
