"""One rank of the two-process data-parallel test (tests/test_dp2_gpu.py starts two of these on cuda:0 over gloo).

Runs STEPS training steps of the MultiModalVAE on this rank's contiguous row shard through the HIP engine
(libmmvae_hip.so), with the gradient SUM all-reduce of mmvae.parallel, and writes the reduced gradient arena of the
last step and the parameters after it to <out>/<mode>_<prec>_rank<r>.pt.  Not a test module (no test_ prefix)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "vae-los-angeles_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

A, D, S, L = 782, 572, 24, 20
B_GLOBAL, STEPS, SEED = 1024, 4, 321


def make_global_batch():
    g = torch.Generator().manual_seed(5)
    a = torch.randn(B_GLOBAL, A, generator=g).abs()
    b = torch.rand(B_GLOBAL, D, generator=g)
    site = torch.randint(0, S, (B_GLOBAL,), generator=g)
    return a, b, site


def flat_grads(model):
    """Copy of the parameters' gradients in arena order."""
    return torch.cat([p.grad.reshape(-1) for p in model._graph().param_list()]).clone()


def main():
    mode, prec, out = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    from mmvae import parallel
    from mmvae.graphs import GraphedTrainStep
    from mmvae.optim import FusedAdamW
    from src.models import MultiModalVAE
    from src.utils import vae_loss

    torch.manual_seed(SEED + 17 * rank)                  # different initial weights per rank: broadcast_parameters must replace them
    model = MultiModalVAE(A, D, S, L).to(dev).set_precision(prec).train()
    torch.manual_seed(SEED)                              # the Philox key is torch.initial_seed() + rank
    parallel.broadcast_parameters(model)
    opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-5)
    a, b, site = make_global_batch()
    lo, hi = parallel.shard_rows(B_GLOBAL, rank, world)
    a, b, site = a[lo:hi].to(dev), b[lo:hi].to(dev), site[lo:hi].to(dev)
    losses = []
    if mode == "eager":
        parallel.attach(model, overlap=True)
        for _ in range(STEPS):
            ra, rb, rc, mu, lv = model(a=a, b=b, site=site)
            loss, rec, cls, kld = vae_loss(ra, a, rb, b, rc, site, mu, lv, beta=1e-3, gamma=1.0)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.item()))
        grads = flat_grads(model)
    else:
        reduce = lambda flat, async_op=False: dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)
        gs = GraphedTrainStep(model, opt, a, b, site, beta=1e-3, gamma=1.0, warmup=2, reduce=reduce, overlap=(mode == "graph_overlap"))
        for _ in range(STEPS - 2):                           # the two warm-up steps are ordinary training steps
            gs()
            losses.append(gs.losses()[0])
        assert (gs.graph_mid is not None) == (mode == "graph_overlap")
        grads = gs.flat.clone()
    torch.cuda.synchronize()
    torch.save({"grads": grads.cpu(), "params": {k: v.detach().cpu() for k, v in model.named_parameters()},
                "buffers": {k: v.detach().cpu() for k, v in model.named_buffers()}, "losses": losses},
               os.path.join(out, f"{mode}_{prec}_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
