"""Data parallelism for the MultiModalVAE training path: one process per GPU, the minibatch
sharded by rows, and exactly ONE collective per step -- a SUM all-reduce (RCCL over xGMI when
the backend is "nccl") of the flat fp32 gradient arena that `engine.VAEGraph.backward` fills.

SUM, not mean: every loss term of the reference is reduction='sum' (src/utils/losses.py:31,34,
39,42), so the global-batch gradient is the sum of the shard gradients.  BatchNorm statistics
stay per shard (the north_star's "all-reduce and nothing else"), i.e. N ranks reproduce N
independent reference shards whose gradients are summed.

The arena is laid out encoders first, decoders last; backward produces the decoder part first,
so with `overlap=True` that tail is reduced asynchronously while the encoder backward runs.
"""
import torch
import torch.distributed as dist


class GradAllReduce:
    def __init__(self, group=None, overlap=True):
        self.group, self.overlap = group, overlap
        self._pending = None

    def early(self, flat, lo):
        """Called once the gradients in flat[lo:] (decoders) are final."""
        if self.overlap and lo < flat.numel():
            self._pending = (dist.all_reduce(flat[lo:], op=dist.ReduceOp.SUM, group=self.group, async_op=True), lo)

    def final(self, flat):
        if self._pending is not None:
            work, lo = self._pending
            self._pending = None
            if lo > 0:
                dist.all_reduce(flat[:lo], op=dist.ReduceOp.SUM, group=self.group)
            work.wait()
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)


def shard_rows(n_rows, rank, world, equal=False):
    """Contiguous row range [lo, hi) of `rank` (rank r gets rows r*B..(r+1)*B).
    equal=True: every rank gets floor(n_rows / world) rows (the remainder is dropped, like drop_last): a training loop
    whose ranks run different numbers of steps would leave the others waiting in the gradient all-reduce forever."""
    if equal:
        per = n_rows // world
        return rank * per, (rank + 1) * per
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)


def average_bn_buffers(model, group=None):
    """BatchNorm running statistics are per shard during training (no collective in the step); averaging them before
    validation gives every rank the SAME eval-mode model, hence the same validation loss and the same scheduler /
    early-stopping decisions.  2 x 896 floats at the default widths."""
    world = dist.get_world_size(group)
    with torch.no_grad():
        for name, buf in model.named_buffers():
            if name.endswith("running_mean") or name.endswith("running_var"):
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
                buf.div_(world)


def all_ranks_mean(value, device, group=None):
    """Mean of a host float over the ranks (control-flow decisions must be taken on identical numbers everywhere)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()) / dist.get_world_size(group)


def attach(model, group=None, overlap=True):
    """Make `loss.backward()` of this model all-reduce its gradient arena.  Returns the hook."""
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("torch.distributed is not initialised")
    hook = GradAllReduce(group, overlap)
    model._graph().grad_sync = hook
    return hook


def broadcast_parameters(model, src=0, group=None):
    """Replicate parameters and BatchNorm buffers from `src` (same start on every rank)."""
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t, src=src, group=group)
