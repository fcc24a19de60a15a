"""FusedAdamW: torch.optim.AdamW semantics (the optimiser the reference's trainers construct,
optimize_hyperparameters.py:93-97, train_dna2rna.py:185-189) with every parameter tensor updated
by ONE HIP launch (mmvae_adamw_step); the (p, g, m, v) pointer records travel in the kernel arguments.

State layout and hyper-parameter names follow torch.optim.AdamW (`exp_avg`, `exp_avg_sq`,
`step`; `lr`, `betas`, `eps`, `weight_decay`, `maximize`) so `state_dict()` round-trips and LR
schedulers such as ReduceLROnPlateau (train_dna2rna.py:190-195) work unchanged.

The step count used for the bias corrections lives ON THE DEVICE (one uint64 per step bucket,
advanced by mmvae_counter_add after the launch), so `step()` is hipGraph-capturable: a replayed
graph keeps counting.  `lr` is passed by value: re-capture when a scheduler changes it."""
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
        self._step_dev = {}           # (group, host step at creation) -> int64[1] device counter
        self._fast = {}               # group -> [params, grad ptrs, records, pending host steps, device counter]

    def _table(self, key_id, entries):
        """Host-side array of (p, g, m, v, n) records; the launch copies it into its kernel arguments."""
        key = tuple((p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr()) for p, s in entries)
        cached = self._tables.get(key_id)
        if cached is not None and cached[0] == key:
            return cached[1]
        items = (L.AdamWItem * len(entries))(*[L.AdamWItem(p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(),
                                                            s["exp_avg_sq"].data_ptr(), p.numel()) for p, s in entries])
        self._tables[key_id] = (key, items)
        return items

    def _device_step(self, gi, bucket_step, device):
        """Device counter holding (steps already applied) for the parameters of this bucket."""
        k = (gi, str(device))
        ent = self._step_dev.get(k)
        if ent is None or ent[0] != bucket_step - 1:
            t = torch.full((1,), bucket_step - 1, dtype=torch.int64, device=device)
            ent = [bucket_step - 1, t]
            self._step_dev[k] = ent
        ent[0] = bucket_step
        return ent[1]

    def note_replayed_step(self):
        """Bookkeeping after a hipGraph replay that contained step(): the device counters advanced, mirror it on the host."""
        self._sync_fast_steps()
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] += 1
        for ent in self._step_dev.values():
            ent[0] += 1

    def note_captured_step(self):
        """A step() that ran under hipGraph CAPTURE only recorded launches: take its host-side counting back."""
        self._sync_fast_steps()
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] -= 1
        for ent in self._step_dev.values():
            ent[0] -= 1

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            buckets = {}            # step number -> [(param, state)]; normally a single bucket
            fast = self._fast.get(gi)
            if fast is not None and all(p.grad is not None and p.grad.data_ptr() == g for p, g in zip(fast[0], fast[1])):
                # steady state: same parameters, gradients at the same addresses as last step -> reuse records,
                # one shared host step counter (the per-parameter `step` tensors are refreshed in state_dict())
                fast[3] += 1
                b1, b2 = group["betas"]
                with ops.pinned_stream():
                    ops.adamw_step(fast[2], float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], 1.0, 1.0,
                                   group["maximize"], step_dev=fast[4])
                    ops.counter_add(fast[4], 1)
                for ent in self._step_dev.values():
                    if ent[1] is fast[4]:
                        ent[0] += 1
                continue
            self._sync_fast_steps(gi)
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
                single = len(buckets) == 1
                items = self._table((gi, 0 if single else step_no), entries)
                step_dev = self._device_step(gi if single else (gi, step_no), step_no, entries[0][0].device)
                ops.adamw_step(items, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], 1.0, 1.0,
                               group["maximize"], step_dev=step_dev)
                ops.counter_add(step_dev, 1)
                if single:
                    ps = [p for p, _ in entries]
                    self._fast[gi] = [ps, [p.grad.data_ptr() for p in ps], items, 0, step_dev]
        return loss

    def _sync_fast_steps(self, gi=None):
        """Fold the steps taken on the fast path back into the per-parameter `step` tensors."""
        for g, fast in list(self._fast.items()):
            if gi is not None and g != gi:
                continue
            if fast[3]:
                for p in fast[0]:
                    self.state[p]["step"] += fast[3]
                fast[3] = 0
            if gi is not None:
                del self._fast[g]

    def state_dict(self):
        self._sync_fast_steps()
        return super().state_dict()
