"""FusedAdamW: torch.optim.AdamW semantics (the optimiser the reference's trainers construct,
optimize_hyperparameters.py:93-97, train_dna2rna.py:185-189) with every parameter tensor updated
by ONE HIP launch (mmvae_adamw_step); the (p, g, m, v) pointer records travel in the kernel arguments.

State layout and hyper-parameter names follow torch.optim.AdamW (`exp_avg`, `exp_avg_sq`,
`step`; `lr`, `betas`, `eps`, `weight_decay`, `maximize`) so `state_dict()` / `load_state_dict()`
round-trip (also with a stock torch.optim.AdamW) and LR schedulers such as ReduceLROnPlateau
(train_dna2rna.py:190-195) work unchanged.

The step count used for the bias corrections lives ON THE DEVICE (one int64 per step bucket, incremented
by the AdamW launch itself once all its blocks have read it), so `step()` is hipGraph-capturable: a
replayed graph keeps counting.  `lr` is passed by value: re-capture when a scheduler changes it.

Caches (pointer tables, the steady-state fast path, device counters) are keyed on the identity of
everything the kernel dereferences -- parameter, gradient and moment addresses and the set of parameters
that have a gradient -- and are dropped by `load_state_dict()` / `add_param_group()`."""
import torch

from . import _lib as L
from . import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, maximize=False):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, maximize=maximize))
        self._lr_dev = {}             # group index -> [host value mirrored, float32[1] device]: see device_lr()
        self.lr_on_device = False
        self._drop_caches()

    # ------------------------------------------------------------------------------------------------------------
    # learning rate in device memory (hipGraph replays follow an LR scheduler without re-capture)
    # ------------------------------------------------------------------------------------------------------------
    def device_lr(self, enable=True):
        """From now on every launch reads the learning rate from a device scalar per parameter group; `sync_lr()` copies
        `param_groups[i]["lr"]` there when a scheduler changed it (GraphedTrainStep calls it before every replay)."""
        self.lr_on_device = bool(enable)
        return self

    def _lr_tensor(self, gi, group, device):
        ent = self._lr_dev.get(gi)
        if ent is None:
            ent = self._lr_dev[gi] = [float(group["lr"]), torch.tensor([float(group["lr"])], dtype=torch.float32, device=device)]
        return ent

    def sync_lr(self):
        """Host -> device copy of the learning rates that changed since the last call (no-op otherwise)."""
        for gi, group in enumerate(self.param_groups):
            ent = self._lr_dev.get(gi)
            if ent is not None and ent[0] != float(group["lr"]):
                ent[0] = float(group["lr"])
                ent[1].fill_(ent[0])

    # ------------------------------------------------------------------------------------------------------------
    # caches
    # ------------------------------------------------------------------------------------------------------------
    def _drop_caches(self):
        self._tables = {}             # key id -> (pointer key, ctypes records)
        self._step_dev = {}           # (group / bucket, device) -> [host mirror of the count, int64[CTR_COPIES] device copies]
        self._fast = {}               # group index -> dict(params, key, items, pending, step_dev)

    @staticmethod
    def _ptr_key(entries):
        return tuple((p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr()) for p, s in entries)

    def _table(self, key_id, entries):
        """Host-side array of (p, g, m, v, n) records; the launch copies it into its kernel arguments."""
        key = self._ptr_key(entries)
        cached = self._tables.get(key_id)
        if cached is not None and cached[0] == key:
            return cached[1], key
        items = (L.AdamWItem * len(entries))(*[L.AdamWItem(p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(),
                                                            s["exp_avg_sq"].data_ptr(), p.numel()) for p, s in entries])
        self._tables[key_id] = (key, items)
        return items, key

    def _device_step(self, key, bucket_step, device):
        """Device counter holding (steps already applied) for the parameters of this bucket: CTR_COPIES identical int64
        copies (mmvae_adamw_step advances them itself, one copy per block)."""
        k = (key, str(device))
        ent = self._step_dev.get(k)
        if ent is None or ent[0] != bucket_step - 1:
            t = torch.full((L.CTR_COPIES,), bucket_step - 1, dtype=torch.int64, device=device)
            ent = [bucket_step - 1, t]
            self._step_dev[k] = ent
        ent[0] = bucket_step
        return ent[1]

    def _fast_valid(self, group, fast):
        """The cached launch is still THE launch: same parameters have gradients, and parameters, gradients and moments
        still live where the cached records point (a parameter that gained a gradient, a re-allocated gradient arena or
        reloaded moments all fail this and take the general path, which rebuilds everything)."""
        ps = fast["params"]
        n = 0
        for p in group["params"]:
            if p.grad is not None:
                n += 1
        if n != len(ps):
            return False
        state = self.state
        for p, (pp, gp, mp, vp) in zip(ps, fast["key"]):
            g = p.grad
            if g is None or g.data_ptr() != gp or p.data_ptr() != pp:
                return False
            st = state.get(p)
            if not st or st["exp_avg"].data_ptr() != mp or st["exp_avg_sq"].data_ptr() != vp:
                return False
        return True

    # ------------------------------------------------------------------------------------------------------------
    # hipGraph bookkeeping
    # ------------------------------------------------------------------------------------------------------------
    def note_replayed_step(self):
        """Bookkeeping after a hipGraph replay that contained step(): the device counters advanced, mirror it on the host."""
        self._bump_host_steps(+1)

    def note_captured_step(self):
        """A step() that ran under hipGraph CAPTURE only recorded launches: take its host-side counting back."""
        self._bump_host_steps(-1)

    def _bump_host_steps(self, inc):
        seen = set()
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st and id(st["step"]) not in seen:             # aliased counters (fast path) move once
                    seen.add(id(st["step"]))
                    st["step"] += inc
        for ent in self._step_dev.values():
            ent[0] += inc

    # ------------------------------------------------------------------------------------------------------------
    # step
    # ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            fast = self._fast.get(gi)
            if fast is not None and self._fast_valid(group, fast):
                # steady state: the parameters of the bucket share ONE host `step` tensor (aliased in their state entries),
                # so `optimizer.state[p]["step"]` stays current at the price of a single host increment
                fast["step_t"] += 1
                with ops.pinned_stream():
                    ops.adamw_step(fast["items"], float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], 1.0, 1.0,
                                   group["maximize"], step_dev=fast["step_dev"], lr_dev=self._lr_arg(gi, group, fast["params"][0].device))
                for ent in self._step_dev.values():
                    if ent[1] is fast["step_dev"]:
                        ent[0] += 1
                continue
            self._sync_fast_steps(gi)
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
                else:
                    self._adopt_loaded_state(p, st)
                st["step"] += 1
                buckets.setdefault(int(st["step"].item()), []).append((p, st))
            for step_no, entries in buckets.items():
                single = len(buckets) == 1
                items, key = self._table((gi, 0 if single else step_no), entries)
                step_dev = self._device_step(gi if single else (gi, step_no), step_no, entries[0][0].device)
                with ops.pinned_stream():
                    ops.adamw_step(items, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], 1.0, 1.0,
                                   group["maximize"], step_dev=step_dev, lr_dev=self._lr_arg(gi, group, entries[0][0].device))
                if single:
                    shared = entries[0][1]["step"]
                    for _, st in entries:
                        st["step"] = shared                       # un-aliased again by _sync_fast_steps()
                    self._fast[gi] = dict(params=[p for p, _ in entries], key=key, items=items, step_t=shared, step_dev=step_dev)
        return loss

    def _lr_arg(self, gi, group, device):
        if not self.lr_on_device:
            return None
        if not torch.cuda.is_current_stream_capturing():
            ent = self._lr_tensor(gi, group, device)
            if ent[0] != float(group["lr"]):              # eager steps keep the device value current themselves
                ent[0] = float(group["lr"])
                ent[1].fill_(ent[0])
            return ent[1]
        return self._lr_tensor(gi, group, device)[1]      # under capture the tensor must already exist (warm-up created it)

    @staticmethod
    def _adopt_loaded_state(p, st):
        """State that came in through load_state_dict(): torch casts it to the parameter's dtype / device already; make
        sure of the layout the kernel assumes (dense fp32 next to the parameter, `step` a host scalar tensor)."""
        for k in ("exp_avg", "exp_avg_sq"):
            t = st[k]
            if t.device != p.device or t.dtype != torch.float32 or not t.is_contiguous():
                st[k] = t.to(device=p.device, dtype=torch.float32).contiguous()
        # a tensor of its own per parameter: torch.load() keeps the `step` tensors of a checkpoint ALIASED when they were
        # aliased at save time (our fast path shares one counter) and Optimizer.load_state_dict passes `step` through as it
        # is -- the per-parameter `+= 1` below would then count one shared tensor once per parameter
        st["step"] = torch.tensor(float(st["step"]))

    def _sync_fast_steps(self, gi=None):
        """Leave the fast path of group gi (all groups when None): every parameter gets a `step` tensor of its own again
        (they were aliased to one shared counter) and the cached launch is dropped."""
        for g, fast in list(self._fast.items()):
            if gi is not None and g != gi:
                continue
            for p in fast["params"]:
                st = self.state.get(p)
                if st is not None and st.get("step") is fast["step_t"]:
                    st["step"] = fast["step_t"].clone()
            del self._fast[g]

    # ------------------------------------------------------------------------------------------------------------
    # (de)serialisation: torch.optim.AdamW's layout
    # ------------------------------------------------------------------------------------------------------------
    def state_dict(self):
        return super().state_dict()               # aliased `step` tensors serialise as equal values, one per parameter

    def load_state_dict(self, state_dict):
        """Resume: moments and step counts are replaced, so every cached pointer record and device counter is stale."""
        self._sync_fast_steps()
        super().load_state_dict(state_dict)
        self._drop_caches()

    # ------------------------------------------------------------------------------------------------------------
    # in-place snapshot / restore (hipGraph capture needs eager warm-up steps that must not advance the run)
    # ------------------------------------------------------------------------------------------------------------
    def snapshot(self):
        """Copies of every state tensor (moments, step counts), keyed by parameter."""
        return {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in self.state.items()}

    @torch.no_grad()
    def restore(self, snap):
        """Write a snapshot() back IN PLACE: moments keep their addresses, so pointer tables, the fast path and a captured
        graph stay valid; the device step counters are set to the restored count.  State created after the snapshot
        (first warm-up step) is reset to zero moments / step 0, which is what a missing state means."""
        for p, st in self.state.items():
            old = snap.get(p)
            for k, v in st.items():
                if torch.is_tensor(v):
                    if old is not None:
                        v.copy_(old[k])           # aliased `step` counters: every alias restores the same value
                    else:
                        v.zero_()
        for gi, fast in list(self._fast.items()):
            steps = {int(self.state[p]["step"].item()) for p in fast["params"]}
            if len(steps) != 1:
                self._drop_caches()
                return
            n = steps.pop()
            fast["step_dev"].fill_(n)
            for ent in self._step_dev.values():
                if ent[1] is fast["step_dev"]:
                    ent[0] = n

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_fast"):
            self._drop_caches()
