"""hipGraph capture of the whole training step (MI355X: "HIP graphs instead of a tracing compiler").

One step of the reference loop (optimize_hyperparameters.py:104-113; train_dna2rna.py:86-96 for the directional models) is
~60 kernel launches; issued from Python they leave the GPU idle between the loss read-back and the first backward kernels.
`GraphedTrainStep` captures [minibatch gather ->] forward -> fused loss -> backward -> AdamW ONCE (through
torch.cuda.CUDAGraph, i.e. hipStreamBeginCapture) and replays it with a single launch per step.  What makes the step capturable:
  * no host synchronisation inside (the loss floats are read from a 20-byte device buffer AFTER the replay);
  * the Philox noise offset and the Adam step count live on the device and are advanced by the kernels that read them;
  * beta / gamma (loss) and the learning rate live in device scalars: the beta warm-up (optimize_hyperparameters.py:103) and
    ReduceLROnPlateau (train_dna2rna.py:190-195,216) change them between replays without re-capture;
  * all buffers come from the graph's private pool; inputs are static tensors -- either the caller copies batches into them, or
    (`dataset=`) the step begins with mmvae_gather_rows from a device-resident dataset through a static index vector.

Data parallel (`reduce=` given): nothing of RCCL is captured (no graph support needed from the collective library); the
step is cut into graphs at the points where a collective is issued eagerly on the flat gradient arena:
  * overlap=True (default): THREE graphs -- [forward, loss, decoder backward] | [fusion + encoder backward] | [AdamW].  The cut
    sits where engine.VAEGraph.backward calls grad_sync.early(): the decoder gradients (tail of the arena) are final there and
    their SUM all-reduce is launched asynchronously, then graph 2 replays while RCCL runs on its own stream; the encoder half
    is reduced after graph 2 and both are waited for before graph 3 ("overlapped with encoder backward").
  * overlap=False: TWO graphs -- [forward, loss, backward] | [AdamW] -- with ONE all-reduce of the whole arena in between."""
import os

import torch

from . import functional as F_
from . import ops


class GraphedTrainStep:
    KINDS = ("multimodal", "dna2rna", "rna2dna")

    def __init__(self, model, optimizer, a=None, b=None, site=None, beta=1e-3, gamma=1.0, class_weights=None, warmup=3, reduce=None,
                 preserve_state=False, kind="multimodal", dataset=None, batch_size=None, overlap=True):
        """kind: which reference loop is captured --
             "multimodal": model(a=, b=, site=) + vae_loss                      (optimize_hyperparameters.py:106-110)
             "dna2rna":    model(dna=b, site=) + dna2rna_loss(recon_rna, a, ..)  (train_dna2rna.py:86-92)
             "rna2dna":    model(rna=a, site=) + rna2dna_loss(recon_dna, b, ..)  (train_rna2dna.py, same lines)
           dataset=(A, B, SITE) device-resident full tensors + batch_size: the static batch buffers are owned here and every step
           begins with ONE gather launch through `self.index` (int64 (batch_size,)); call `set_indices(idx)` before a step."""
        if kind not in self.KINDS:
            raise ValueError(f"kind must be one of {self.KINDS}")
        self.kind = kind
        self.dataset = dataset
        if dataset is not None:
            if batch_size is None:
                raise ValueError("dataset= needs batch_size=")
            dA, dB, dS = dataset
            if not dA.is_cuda:
                raise RuntimeError("GraphedTrainStep needs CUDA/HIP tensors; there is no CPU fallback")
            dev = dA.device
            a = torch.empty((batch_size,) + tuple(dA.shape[1:]), dtype=dA.dtype, device=dev)
            b = torch.empty((batch_size,) + tuple(dB.shape[1:]), dtype=dB.dtype, device=dev)
            site = torch.empty(batch_size, dtype=torch.int64, device=dev)
            self.index = torch.arange(batch_size, dtype=torch.int64, device=dev)
        if not a.is_cuda:
            raise RuntimeError("GraphedTrainStep needs CUDA/HIP tensors; there is no CPU fallback")
        self.model, self.optimizer = model, optimizer
        self.a, self.b, self.site = a, b, site                  # static input buffers: copy_ new batches into them (or set_indices)
        self.beta, self.gamma, self.class_weights = float(beta), float(gamma), class_weights
        self.hyper = torch.tensor([self.beta, self.gamma], dtype=torch.float32, device=a.device)     # device-resident {beta, gamma}
        self.optimizer.device_lr(True)                          # ... and learning rate: no re-capture when they change
        self.warmup = warmup
        self.reduce = reduce                                    # callable(flat_grad_slice, async_op=False) -> work handle or None
        self.overlap = bool(overlap) and reduce is not None
        # preserve_state: the eager warm-up steps (they build weight / optimiser tables and allocator pools, and capture
        # cannot run without them) are UNDONE before the capture -- parameters, BatchNorm buffers, Adam moments and step
        # counts, Philox offset -- so that constructing / re-capturing the step does not advance training (resume from a
        # checkpoint).  Default False: warm-up steps are ordinary training steps.
        self.preserve_state = preserve_state
        self.graph = self.graph_opt = None
        self._one = torch.ones((), dtype=torch.float32, device=a.device)
        self.recapture()

    # --- hyper-parameters that may change between replays ---------------------------------------------------------------
    def set_beta(self, beta, gamma=None):
        """KL weight (and optionally gamma) of the following steps: one 8-byte host->device copy when the value changed."""
        gamma = self.gamma if gamma is None else float(gamma)
        if float(beta) != self.beta or gamma != self.gamma:
            self.beta, self.gamma = float(beta), gamma
            self.hyper.copy_(torch.tensor([self.beta, self.gamma], dtype=torch.float32), non_blocking=False)

    def set_indices(self, idx):
        """Rows of the device-resident dataset that make up the next minibatch (int64 device tensor, batch_size entries)."""
        self.index.copy_(idx, non_blocking=True)

    # --- the captured work ------------------------------------------------------------------------------------------------
    def _forward_loss(self):
        if self.dataset is not None:
            dA, dB, dS = self.dataset
            with ops.pinned_stream():
                ops.gather_rows([(dA, self.a), (dB, self.b), (dS, self.site)], self.index, dA.shape[0])
        # The step never looks at the reconstructions themselves: their loss terms and gradients are computed inside the decoders'
        # last GEMMs (engine.VAEGraph.fused_recon) instead of writing them as fp32 and reading them back with their targets.
        g = self.model._graph()
        fuse = os.environ.get("MMVAE_NO_LOSS_EPILOGUE") is None
        try:
            if self.kind == "multimodal":
                g.fused_recon = [self.a, self.b, None] if fuse else None
                ra, rb, rc, mu, lv = self.model(a=self.a, b=self.b, site=self.site)
                terms = {"a": (ra, self.a), "b": (rb, self.b), "c": (rc, self.site), "kl": (mu, lv)}
            elif self.kind == "dna2rna":
                g.fused_recon = [self.a] if fuse else None
                rec, mu, lv = self.model(dna=self.b, site=self.site)
                terms = {"a": (rec, self.a), "kl": (mu, lv)}
            else:
                g.fused_recon = [self.b] if fuse else None
                rec, mu, lv = self.model(rna=self.a, site=self.site)
                terms = {"b": (rec, self.b), "kl": (mu, lv)}
        finally:
            g.fused_recon = None
        return F_.fused_loss(terms, self.beta, self.gamma, self.class_weights, unit_grad=True, beta_gamma_dev=self.hyper)

    def run_eager(self):
        """One training step of exactly the captured work, issued from Python (profiling passes: events need eager launches)."""
        return self._step()

    def _step(self):
        total, out4 = self._forward_loss()
        self.optimizer.zero_grad(set_to_none=True)
        total.backward(self._one)                               # static ones: no fill launch, no gradient-scaling launch
        self.optimizer.step()
        return out4

    def _fwd_bwd(self):
        total, out4 = self._forward_loss()
        self.optimizer.zero_grad(set_to_none=True)
        total.backward(self._one)
        return out4

    def _flat_grads(self):
        """The flat fp32 gradient arena behind the parameters' .grad views (engine.VAEGraph.backward)."""
        ps = [p for p in self.model._graph().param_list() if p.grad is not None]
        lo = min(p.grad.data_ptr() for p in ps)
        hi = max(p.grad.data_ptr() + p.grad.numel() * 4 for p in ps)
        first = next(p for p in ps if p.grad.data_ptr() == lo)
        n = (hi - lo) // 4
        if n != sum(p.grad.numel() for p in ps):
            raise RuntimeError("gradients are not one contiguous arena")
        return torch.as_strided(first.grad, (n,), (1,))

    def recapture(self):
        self.model.train()
        g = self.model._graph()
        sync, g.grad_sync = g.grad_sync, (None if self.reduce is not None else g.grad_sync)   # two-graph form: the reduce runs between the replays
        try:
            snap = None
            if self.preserve_state:
                if self.warmup < 1:
                    raise ValueError("preserve_state needs at least one (undone) warm-up step")
                dev = self.a.device
                snap = ({k: v.clone() for k, v in self.model.state_dict().items()}, self.optimizer.snapshot(),
                        F_.engine_noise().state_dict(dev))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                        # eager warm-up: builds weight / optimiser tables, allocator pools
                for _ in range(self.warmup):
                    if self.reduce is None:
                        self._step()
                    else:
                        self._fwd_bwd()
                        self.reduce(self._flat_grads())
                        self.optimizer.step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if snap is not None:                                 # in place: every address the capture will record stays valid
                with torch.no_grad():
                    cur = self.model.state_dict()
                    for k, v in snap[0].items():
                        cur[k].copy_(v)
                self.optimizer.restore(snap[1])
                F_.engine_noise().load_state_dict(snap[2], self.a.device)
                torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            self.graph_mid = None
            if self.reduce is None:
                with torch.cuda.graph(self.graph):
                    self.out4 = self._step()
            else:
                if self.overlap:
                    self._capture_split(g)
                else:
                    with torch.cuda.graph(self.graph):
                        self.out4 = self._fwd_bwd()
                self.flat = self._flat_grads()                   # static address: the arena lives in the graph's private pool
                self.graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_opt, pool=self.graph.pool()):
                    self.optimizer.step()
            self.optimizer.note_captured_step()                  # capture recorded the launches, it did not run them
        finally:
            g.grad_sync = sync
        return self

    def _capture_split(self, g):
        """[forward, loss, decoder backward] into self.graph and [fusion, encoder backward] into self.graph_mid: the capture is
        ended and re-begun INSIDE backward, at the point where engine.VAEGraph.backward reports the decoder gradients final."""
        outer = self

        class _Cut:                                              # stands in for mmvae.parallel.GradAllReduce during the capture
            def early(self_, flat, lo):
                outer.graph.capture_end()
                outer.cut = int(lo)
                outer.graph_mid = torch.cuda.CUDAGraph()
                outer.graph_mid.capture_begin(pool=outer.graph.pool(), capture_error_mode="relaxed")

            def final(self_, flat):
                pass

        prev, g.grad_sync = g.grad_sync, _Cut()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(stream):
                # "relaxed": the cut happens inside loss.backward(), i.e. on autograd's device thread -- only a relaxed capture may be
                # ended from a thread other than the one that began it
                self.graph.capture_begin(capture_error_mode="relaxed")
                try:
                    self.out4 = self._fwd_bwd()
                finally:
                    (self.graph_mid if self.graph_mid is not None else self.graph).capture_end()
        finally:
            g.grad_sync = prev
        torch.cuda.current_stream().wait_stream(stream)
        if self.graph_mid is None:
            raise RuntimeError("the backward pass never reported its decoder gradients: nothing to overlap")

    def __call__(self):
        """Run one training step; returns the device tensor [total, recon, class, kld, labels out of range] (fp32) of that step."""
        self.optimizer.sync_lr()                                 # a scheduler may have changed the learning rate
        self.graph.replay()
        if self.reduce is not None:
            if self.graph_mid is not None:
                work = self.reduce(self.flat[self.cut:], async_op=True)      # decoder half: RCCL runs beside the encoder backward
                self.graph_mid.replay()
                self.reduce(self.flat[:self.cut])
                if work is not None:
                    work.wait()
            else:
                self.reduce(self.flat)
            self.graph_opt.replay()
        self.optimizer.note_replayed_step()
        return self.out4

    def losses(self):
        """(total, recon, class, kld) as floats: the ONE host read of the step (vae_loss does the same)."""
        return tuple(F_.read_losses(self.out4))

    # --- pipelined logging -------------------------------------------------------------------------------------------
    # `losses()` right after a replay makes the host wait for the step, and the GPU then waits for the host to wake up,
    # run Python and launch the next graph: ~55 us of idle GPU per 1.9 ms step (rocprofv3 timeline).  The reference loop
    # reads the loss only to log it (optimize_hyperparameters.py:113-117), so the read can trail the launches by one step:
    # every replay is followed, on the same stream, by a 20-byte copy into one of two pinned host slots and an event;
    # `step_logged()` launches step i and THEN waits for the event of step i-1.  Every step's loss still reaches the host.
    def step_logged(self):
        """Launch one step; returns the (total, recon, class, kld) floats of the PREVIOUS step (None on the first call)."""
        if getattr(self, "_pin", None) is None:
            self._pin = [torch.empty(5, dtype=torch.float32).pin_memory() for _ in range(2)]
            self._pin_ev = [torch.cuda.Event(), torch.cuda.Event()]
            self._pin_n = 0                                      # steps launched through this method
        self()
        k = self._pin_n & 1
        self._pin[k].copy_(self.out4, non_blocking=True)         # stream-ordered: runs before the next replay overwrites out4
        self._pin_ev[k].record()
        self._pin_n += 1
        if self._pin_n < 2:
            return None
        self._pin_ev[k ^ 1].synchronize()
        return tuple(F_.read_losses(self._pin[k ^ 1]))

    def reset_logged(self):
        """Start a fresh logging pipeline (e.g. at an epoch boundary, after flush_logged())."""
        if getattr(self, "_pin_n", 0):
            self._pin_n = 0

    def flush_logged(self):
        """Losses of the last step launched by `step_logged()` (waits for it)."""
        if not getattr(self, "_pin_n", 0):
            return None
        k = (self._pin_n - 1) & 1
        self._pin_ev[k].synchronize()
        return tuple(F_.read_losses(self._pin[k]))
