"""world_size-2 gloo tests (CPU) of the data-parallel path: the gradient all-reduce hook that
VAEGraph.backward drives (early decoder bucket + final encoder bucket), row sharding, and the
semantics the multi-GPU run relies on: SUM (not mean) of per-shard gradients with per-shard
BatchNorm statistics == gradients of independent reference shards added up."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "vae-los-angeles_amd"), os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import np_oracle as O
        import torch_ref as T
        from mmvae import parallel

        # (1) the hook: overlap and non-overlap forms give the plain SUM over ranks
        for overlap in (True, False):
            flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
            hook = parallel.GradAllReduce(overlap=overlap)
            hook.early(flat, 600)            # decoder bucket = flat[600:]
            flat[:600] += 0.5                # "encoder backward" keeps writing the head meanwhile
            hook.final(flat)
            want = torch.arange(1000, dtype=torch.float32) * sum(r + 1 for r in range(world))
            want[:600] += 0.5 * world
            assert torch.equal(flat, want), overlap

        # (2) row sharding covers [0, n) exactly once
        lo, hi = parallel.shard_rows(1001, rank, world)
        cover = torch.zeros(1001)
        cover[lo:hi] = 1
        dist.all_reduce(cover)
        assert torch.all(cover == 1)
        # equal=True: every rank the SAME number of rows (same number of training steps, or the all-reduce counts mismatch)
        lo, hi = parallel.shard_rows(1001, rank, world, equal=True)
        assert hi - lo == 1001 // world and lo == rank * (1001 // world)

        # (2b) control-flow agreement: per-rank validation losses / BatchNorm running statistics become ONE value everywhere
        v = parallel.all_ranks_mean(10.0 + rank, "cpu")
        assert v == 10.0 + (world - 1) / 2
        bn = torch.nn.Sequential(torch.nn.BatchNorm1d(4))
        bn[0].running_mean.fill_(float(rank)); bn[0].running_var.fill_(1.0 + rank)
        parallel.average_bn_buffers(bn)
        assert torch.all(bn[0].running_mean == (world - 1) / 2) and torch.all(bn[0].running_var == 1.0 + (world - 1) / 2)

        # (3) DP semantics with the stock-torch restatement: each rank trains on its shard (per-shard BN),
        #     gradients are SUMMED through the hook; rank 0 compares with both shards computed locally.
        A, D, S, L, E, B = 48, 36, 5, 6, 8, 64
        P, Bf = O.make_params(5, A, D, S, L, E)
        a, b, site = O.make_batch(6, B, A, D, S)
        masks, eps = O.make_noise(7, B, L)

        def shard_grads(r):
            lo, hi = parallel.shard_rows(B, r, world)
            p, bufs = T.to_torch(P, Bf)
            m = {k: torch.from_numpy(v[lo:hi].astype(np.float32)) for k, v in masks.items()}
            ra, rb, rc, mu, lv = T.forward(p, bufs, torch.from_numpy(a[lo:hi]), torch.from_numpy(b[lo:hi]),
                                           torch.from_numpy(site[lo:hi]), True, m, torch.from_numpy(eps[lo:hi]))
            loss, *_ = T.loss_fn(ra, torch.from_numpy(a[lo:hi]), rb, torch.from_numpy(b[lo:hi]), rc, torch.from_numpy(site[lo:hi]), mu, lv)
            loss.backward()
            return torch.cat([t.grad.reshape(-1) for t in p.values()])

        mine = shard_grads(rank)
        hook = parallel.GradAllReduce(overlap=True)
        hook.early(mine, mine.numel() // 2)
        hook.final(mine)
        if rank == 0:
            want = sum(shard_grads(r) for r in range(world))
            np.testing.assert_allclose(mine.numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
        out.put((rank, "ok"))
    except Exception as e:        # surface the failure in the parent
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_grad_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_attach_requires_process_group():
    from mmvae import parallel
    from src.models import MultiModalVAE
    with pytest.raises(RuntimeError, match="not initialised"):
        parallel.attach(MultiModalVAE(8, 8, 3, 2))
