"""hipGraph capture of the whole training step (MI355X: "HIP graphs instead of a tracing compiler").

One step of the reference loop (optimize_hyperparameters.py:104-113) is ~75 kernel launches; issued from
Python they leave the GPU idle between the loss read-back and the first backward kernels.  `GraphedTrainStep`
captures forward -> fused loss -> backward -> AdamW ONCE (through torch.cuda.CUDAGraph, i.e. hipStreamBeginCapture)
and replays it with a single launch per step.  What makes the step capturable:
  * no host synchronisation inside (the loss floats are read from a 16-byte device buffer AFTER the replay);
  * the Philox noise offset and the Adam step count live on the device and are advanced by kernels;
  * all buffers come from the graph's private pool; inputs are static tensors the caller copies batches into.
Host scalars baked into the graph (lr, beta, gamma): call `recapture()` after changing them."""
import torch

from . import functional as F_


class GraphedTrainStep:
    def __init__(self, model, optimizer, a, b, site, beta=1e-3, gamma=1.0, class_weights=None, warmup=3):
        if not a.is_cuda:
            raise RuntimeError("GraphedTrainStep needs CUDA/HIP tensors; there is no CPU fallback")
        self.model, self.optimizer = model, optimizer
        self.a, self.b, self.site = a, b, site                  # static input buffers: copy_ new batches into them
        self.beta, self.gamma, self.class_weights = float(beta), float(gamma), class_weights
        self.warmup = warmup
        self.graph = None
        self.recapture()

    def _step(self):
        ra, rb, rc, mu, lv = self.model(a=self.a, b=self.b, site=self.site)
        terms = {"a": (ra, self.a), "b": (rb, self.b), "c": (rc, self.site), "kl": (mu, lv)}
        total, out4 = F_.fused_loss(terms, self.beta, self.gamma, self.class_weights)
        self.optimizer.zero_grad(set_to_none=True)
        total.backward()
        self.optimizer.step()
        return out4

    def recapture(self):
        self.model.train()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                            # eager warm-up: builds weight / optimiser tables, allocator pools
            for _ in range(self.warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out4 = self._step()
        self.optimizer.note_captured_step()                      # capture recorded the launches, it did not run them
        return self

    def __call__(self):
        """Run one training step; returns the device tensor [total, recon, class, kld] (fp32) of that step."""
        self.graph.replay()
        self.optimizer.note_replayed_step()
        return self.out4

    def losses(self):
        """(total, recon, class, kld) as floats: the ONE host read of the step (vae_loss does the same)."""
        return tuple(self.out4.tolist())
