#!/usr/bin/env python3
"""A/B of the hot NT shapes (bf16 A): run under different MMVAE_* env settings and compare."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools")]
os.environ.setdefault("M", "65536")
import torch
from bench_gemm_lib import nt, nt_bn_bwd
print(os.environ.get("TAG", ""), {k: v for k, v in os.environ.items() if k.startswith("MMVAE_")})
print("  DecB.L2.fwd  N=572 K=512 f32 out :", round(nt(572, 512, out_dtype=torch.float32), 1))
print("  DecA.L1.fwd  N=782 K=128 f32 out :", round(nt(782, 128, out_dtype=torch.float32), 1))
print("  DecB.L1.fwd  N=512 K=256 bf16 out:", round(nt(512, 256), 1))
print("  DecB.L2.dX   N=512 K=572 relumask:", round(nt(512, 572, epi="relu"), 1))
print("  DecB.L1.dX   N=256 K=512 relumask:", round(nt(256, 512, epi="relu"), 1))
print("  DecA.L1.dX   N=128 K=782 relumask:", round(nt(128, 782, epi="relu"), 1))
print("  EncB.L0.dX   N=512 K=256 bn-bwd   :", round(nt_bn_bwd(512, 256), 1))
