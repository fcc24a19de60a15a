"""FusedAdamW: torch.optim.AdamW semantics (the optimiser the reference's trainers construct,
optimize_hyperparameters.py:93-97, train_dna2rna.py:185-189) with every parameter tensor updated
by ONE HIP launch (mmvae_adamw_step) driven by a device-resident pointer table.

State layout and hyper-parameter names follow torch.optim.AdamW (`exp_avg`, `exp_avg_sq`,
`step`; `lr`, `betas`, `eps`, `weight_decay`, `maximize`) so `state_dict()` round-trips and LR
schedulers such as ReduceLROnPlateau (train_dna2rna.py:190-195) work unchanged."""
import ctypes as C

import torch

from . import _lib as L
from . import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, maximize=False):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, maximize=maximize))
        self._tables = {}

    def _table(self, gi, entries):
        key = tuple((p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr()) for p, s in entries)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached[1], cached[2]
        if len(self._tables) > 64:
            self._tables.clear()
        items = [L.AdamWItem(p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(), p.numel())
                 for p, s in entries]
        arr = (L.AdamWItem * len(items))(*items)
        dev = entries[0][0].device
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        max_numel = max(p.numel() for p, _ in entries)
        self._tables[gi] = (key, table, max_numel)
        return table, max_numel

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            buckets = {}            # step number -> [(param, state)]; normally a single bucket
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdamW runs on the MI355X only; there is no CPU fallback")
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdamW expects contiguous fp32 parameters and gradients")
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                buckets.setdefault(int(st["step"].item()), []).append((p, st))
            b1, b2 = group["betas"]
            for step_no, entries in buckets.items():
                table, max_numel = self._table((gi, step_no if len(buckets) > 1 else 0), entries)
                ops.adamw_step(table, len(entries), max_numel, float(group["lr"]), b1, b2, group["eps"],
                               group["weight_decay"], 1.0 - b1 ** step_no, 1.0 - b2 ** step_no, group["maximize"])
        return loss
