"""Shared training harness behind train.py (MultiModalVAE), train_dna2rna.py and train_rna2dna.py.

Loop shape = the reference's trainers (train_dna2rna.py:72-252; optimize_hyperparameters.py:163-211 for MultiModalVAE):
AdamW(lr, weight_decay) + ReduceLROnPlateau(min, factor, patience) + beta warm-up + per-epoch validation + best-checkpoint
`torch.save(model.state_dict())` + early stopping + `latest_*_run_id.txt`.  What is different is HOW a step runs on the MI355X:

  * the dataset lives in HBM; an epoch is a device permutation, a minibatch is ONE gather launch (mmvae_gather_rows) through a
    static index vector -- the reference builds every sample with `torch.tensor(row)` in Python (src/data/dataset.py:28-39);
  * the whole step [gather, forward, loss, backward, AdamW] is one hipGraph replay (mmvae.graphs.GraphedTrainStep); beta and the
    learning rate live in device scalars, so the warm-up and the LR scheduler need no re-capture;
  * the loss floats of every step reach the host one step late (pinned 20-byte copies), so the host never stalls the GPU;
  * under torchrun every rank owns an equal row shard, gradients are SUM-all-reduced (RCCL) between the two graph replays,
    BatchNorm running statistics and the validation loss are averaged over ranks before any control-flow decision.
  * `--resume state.pt` continues a run (model + optimiser + scheduler + Philox position + epoch counters): the reference
    cannot resume (it only saves model weights).
"""
import argparse
import os
import time
from datetime import datetime

import numpy as np
import torch
import torch.distributed as dist

from mmvae import parallel, checkpoint
from mmvae.graphs import GraphedTrainStep
from mmvae.optim import FusedAdamW
from src.config import Config

KINDS = {
    "multimodal": dict(tag="multivae", title="MultiModalVAE"),
    "dna2rna": dict(tag="dna2rna", title="DNA2RNAVAE"),
    "rna2dna": dict(tag="rna2dna", title="RNA2DNAVAE"),
}


def balanced_class_weights(site, n_sites):
    """compute_class_weights of the reference (optimize_hyperparameters.py:33-44): sklearn's 'balanced' weights over the classes
    PRESENT in the training labels, n / (k_present * count_c); classes that do not occur keep weight 1."""
    counts = torch.bincount(site.reshape(-1), minlength=n_sites).double()
    present = counts > 0
    w = torch.ones(n_sites, dtype=torch.float64)
    w[present] = site.numel() / (present.sum().double() * counts[present])
    return w.float()


def synthetic_dataset(n, a_dim, d_dim, n_sites, seed):
    g = torch.Generator().manual_seed(seed)
    tpm = torch.log1p(torch.exp(1.5 * torch.randn(n, a_dim, generator=g)))          # log1p(TPM)-like, >= 0 (prepare_data.py:123-125)
    beta = torch.rand(n, d_dim, generator=g)                                          # methylation beta values in [0, 1]
    site = torch.randint(0, n_sites, (n,), generator=g, dtype=torch.int64)
    return tpm, beta, site


def load_pickled_dataset(path):
    """Same columns as the reference's processed_data.pkl (src/data/dataset.py:20-30)."""
    import pandas as pd
    df = pd.read_pickle(path)          # a file the USER produced with the reference's own scripts -- never one from the reference tree
    tpm = torch.tensor(np.stack(df["tpm_unstranded"].values), dtype=torch.float32)
    beta = torch.tensor(np.stack(df["beta_value"].values), dtype=torch.float32)
    site = torch.tensor(df["primary_site_encoded"].values, dtype=torch.int64)
    return tpm, beta, site


def build_parser(kind):
    ap = argparse.ArgumentParser(description=f"{KINDS[kind]['title']} trainer on MI355X")
    ap.add_argument("--data", default=None, help="processed_data.pkl produced by the reference's prepare scripts (default: synthetic)")
    ap.add_argument("--samples", type=int, default=262144, help="synthetic dataset size")
    ap.add_argument("--input-dim-a", type=int, default=int(os.getenv("INPUT_DIM_A", 782)))      # env overrides as train_dna2rna.py:172-174
    ap.add_argument("--input-dim-b", type=int, default=int(os.getenv("INPUT_DIM_B", 572)))
    ap.add_argument("--n-sites", type=int, default=24)
    ap.add_argument("--latent-dim", type=int, default=int(os.getenv("LATENT_DIM", Config.LATENT_DIM)))
    ap.add_argument("--batch-size", type=int, default=4096, help="rows per GPU")
    ap.add_argument("--epochs", type=int, default=Config.NUM_EPOCHS)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--checkpoint-dir", default=Config.CHECKPOINT_DIR)
    ap.add_argument("--resume", default=None, help="training state written by --save-state (model + optimiser + scheduler + noise)")
    ap.add_argument("--save-state", default=None, help="write a resumable training state here after every epoch")
    ap.add_argument("--eager", action="store_true", help="issue every launch from Python (reference-shaped loop) instead of the captured step")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend under torchrun (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank on cuda:0 (needs --backend gloo)")
    ap.add_argument("--log-json", default=None, help="append one JSON object per epoch (steps, train/val loss, lr, early-stop counter) to "
                                                     "<path>.rank<r>: every rank writes its own file, so that a run can be checked for identical decisions")
    return ap


def make_model(kind, args):
    from src.models import MultiModalVAE, DNA2RNAVAE, RNA2DNAVAE
    cls = {"multimodal": MultiModalVAE, "dna2rna": DNA2RNAVAE, "rna2dna": RNA2DNAVAE}[kind]
    return cls(args.input_dim_a, args.input_dim_b, args.n_sites, args.latent_dim)


def forward_loss(kind, model, a, b, s, beta, class_weights):
    """One reference-shaped forward + loss; returns the total loss tensor."""
    from src.utils import vae_loss
    from src.utils.directional_losses import dna2rna_loss, rna2dna_loss
    if kind == "multimodal":
        ra, rb, rc, mu, lv = model(a=a, b=b, site=s)
        return vae_loss(ra, a, rb, b, rc, s, mu, lv, beta=beta, gamma=Config.GAMMA, class_weights=class_weights)[0]
    if kind == "dna2rna":
        rec, mu, lv = model(dna=b, site=s)
        return dna2rna_loss(rec, a, mu, lv, beta=beta)[0]
    rec, mu, lv = model(rna=a, site=s)
    return rna2dna_loss(rec, b, mu, lv, beta=beta)[0]


def run(kind, argv=None):
    args = build_parser(kind).parse_args(argv)
    tag = KINDS[kind]["tag"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("the trainers need an MI355X: the product path has no CPU fallback")
    if args.single_device:
        if args.backend == "nccl":
            raise SystemExit("--single-device puts every rank on cuda:0, which RCCL refuses: use --backend gloo")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    if args.data:
        tpm, beta_v, site = load_pickled_dataset(args.data)
        args.input_dim_a, args.input_dim_b = tpm.shape[1], beta_v.shape[1]
        args.n_sites = int(site.max()) + 1
    else:
        tpm, beta_v, site = synthetic_dataset(args.samples, args.input_dim_a, args.input_dim_b, args.n_sites, Config.RANDOM_SEED)
    # class indices are validated ONCE, on the host, before any rank trains: inside the loop a bad label only shows up in the loss
    # read of the rank that holds it (read_losses raises there) and the other ranks would wait in the all-reduce.  -100 is
    # F.cross_entropy's ignore_index (losses.py:39): legal for the loss, but the site encoder's Embedding has no row for it.
    lo_, hi_ = int(site.min()), int(site.max())
    if lo_ < 0 or hi_ >= args.n_sites:
        raise SystemExit(f"site labels must lie in [0, {args.n_sites}); found [{lo_}, {hi_}]")
    n = tpm.shape[0]
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(Config.RANDOM_SEED))       # train_test_split(random_state=42) stand-in
    n_val = int(n * Config.TRAIN_TEST_SPLIT)
    val_idx, train_idx = perm[:n_val], perm[n_val:]
    lo, hi = parallel.shard_rows(train_idx.numel(), rank, world, equal=True)   # equal shards: every rank runs the same number of steps
    tr = [t[train_idx[lo:hi]].to(dev).contiguous() for t in (tpm, beta_v, site)]
    va = [t[val_idx].to(dev).contiguous() for t in (tpm, beta_v, site)]
    class_weights = balanced_class_weights(site[train_idx], args.n_sites).to(dev) if kind == "multimodal" else None
    B = args.batch_size
    n_train = tr[0].shape[0]
    steps_per_epoch = n_train // B                                              # drop_last=True
    if steps_per_epoch < 1:
        raise SystemExit(f"batch size {B} exceeds the {n_train} training rows of this rank")

    torch.manual_seed(Config.RANDOM_SEED)
    model = make_model(kind, args).to(dev).set_precision(args.precision)
    if world > 1:
        parallel.broadcast_parameters(model)
    optimizer = FusedAdamW(model.parameters(), lr=Config.LEARNING_RATE, weight_decay=Config.WEIGHT_DECAY)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=Config.LR_SCHEDULER_FACTOR,
                                                           patience=Config.LR_SCHEDULER_PATIENCE)
    run_id = datetime.now().strftime("%Y%m%d_%H%M%S")
    start_epoch, best_val, trigger = 0, float("inf"), 0
    if args.resume:
        extra = checkpoint.load_training_state(args.resume, model, optimizer, scheduler)
        start_epoch, best_val, trigger = int(extra.get("epoch", 0)), float(extra.get("best_val", float("inf"))), int(extra.get("trigger", 0))
        run_id = extra.get("run_id", run_id)
        if rank == 0:
            print(f"Resumed from {args.resume}: epoch {start_epoch}, best validation loss {best_val:.2f}")
    os.makedirs(args.checkpoint_dir, exist_ok=True)
    if rank == 0:
        print(f"Starting {KINDS[kind]['title']} training run: {run_id}  ({n_train} rows/rank x {world} rank(s), batch {B}, {args.precision})")

    graphed = None
    if args.eager:
        if world > 1:
            parallel.attach(model)
    else:
        reduce = (lambda flat, async_op=False: dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)) if world > 1 else None
        beta0 = min(1.0, start_epoch / Config.BETA_WARMUP_EPOCHS) * Config.BETA_START
        graphed = GraphedTrainStep(model, optimizer, beta=beta0, gamma=Config.GAMMA, class_weights=class_weights, warmup=1,
                                   preserve_state=True, reduce=reduce, kind=kind, dataset=tuple(tr), batch_size=B)

    perm_gen = torch.Generator(device=dev).manual_seed(Config.RANDOM_SEED + 1000 * rank)
    for epoch in range(start_epoch, args.epochs):
        model.train()
        beta = min(1.0, epoch / Config.BETA_WARMUP_EPOCHS) * Config.BETA_START          # train_dna2rna.py:80
        perm_gen.manual_seed(Config.RANDOM_SEED + 1000 * rank + epoch)                  # the epoch's shuffle is a function of (seed, rank, epoch): resumable
        order = torch.randperm(n_train, device=dev, generator=perm_gen)
        t0, running, steps = time.time(), 0.0, 0
        if graphed is not None:
            graphed.set_beta(beta)
            for i in range(steps_per_epoch):
                graphed.set_indices(order[i * B:(i + 1) * B])
                prev = graphed.step_logged()
                if prev is not None:
                    running += prev[0]
            running += graphed.flush_logged()[0]
            graphed.reset_logged()                                                # next epoch starts a fresh logging pipeline
            steps = steps_per_epoch
        else:
            for i in range(steps_per_epoch):
                idx = order[i * B:(i + 1) * B]
                a, b, s = tr[0][idx], tr[1][idx], tr[2][idx]
                loss = forward_loss(kind, model, a, b, s, beta, class_weights)
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
                running += loss.item()
                steps += 1
        torch.cuda.synchronize()
        dt = time.time() - t0
        if world > 1:
            parallel.average_bn_buffers(model)                                    # same eval-mode model on every rank
        model.eval()
        val_loss, vsteps = 0.0, 0
        with torch.no_grad():
            for i in range(0, va[0].shape[0], B):
                a, b, s = va[0][i:i + B], va[1][i:i + B], va[2][i:i + B]
                val_loss += forward_loss(kind, model, a, b, s, beta, class_weights).item()
                vsteps += 1
        val_loss /= max(vsteps, 1)
        if world > 1:
            # eps is sampled in eval mode too (vae.py:73) and the Philox streams differ per rank: the scheduler / checkpoint /
            # early-stop decisions below must see ONE number on every rank or the ranks part ways
            val_loss = parallel.all_ranks_mean(val_loss, dev)
        scheduler.step(val_loss)                                                  # a new LR reaches the captured step through its device scalar
        if rank == 0:
            print(f"Epoch [{epoch + 1}/{args.epochs}] | Train Loss: {running / max(steps, 1):.2f} | Val Loss: {val_loss:.2f} | "
                  f"beta={beta:.5f} | lr={optimizer.param_groups[0]['lr']:.2e} | {world * steps * B / dt:,.0f} samples/s")
        if args.log_json:
            import json
            with open(f"{args.log_json}.rank{rank}", "a") as f:
                f.write(json.dumps(dict(epoch=epoch + 1, steps=steps, train_loss=running / max(steps, 1), val_loss=val_loss,
                                        lr=optimizer.param_groups[0]["lr"], best_val=min(best_val, val_loss), trigger=trigger)) + "\n")
        stop = False
        if val_loss < best_val:
            best_val, trigger = val_loss, 0
            if rank == 0:
                torch.save(model.state_dict(), os.path.join(args.checkpoint_dir, f"best_{tag}_{run_id}.pt"))
        else:
            trigger += 1
            stop = trigger >= Config.PATIENCE
        if args.save_state and rank == 0:
            checkpoint.save_training_state(args.save_state, model, optimizer, scheduler, epoch=epoch + 1, best_val=best_val,
                                           trigger=trigger, run_id=run_id)
        if stop:
            if rank == 0:
                print(f"Early stopping triggered at epoch {epoch + 1}!")
            break
    if rank == 0:
        with open(f"latest_{tag}_run_id.txt", "w") as f:                            # train_dna2rna.py:244-245
            f.write(run_id)
        print(f"Training complete. Run ID: {run_id}. Best validation loss: {best_val:.2f}")
    if world > 1:
        dist.destroy_process_group()
