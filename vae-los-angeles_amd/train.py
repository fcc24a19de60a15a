#!/usr/bin/env python3
"""Reference-shaped trainer for MultiModalVAE on MI355X.

The reference's README and run_pipeline.sh call a `train.py` that is absent from its tree
(SURVEY.md D1); this is that harness, with the loop shape of the one in-tree MultiModalVAE
trainer (reference optimize_hyperparameters.py:163-211: AdamW + beta warm-up, per-epoch
validation, best-checkpoint `torch.save(state_dict)`) plus the ReduceLROnPlateau / early-stop /
run-id conventions of train_dna2rna.py:150-252.  The model/loss calls are the reference's:

    recon_a, recon_b, recon_c, mu, logvar = model(a=tpm, b=beta_data, site=site)
    loss, _, _, _ = vae_loss(recon_a, tpm, recon_b, beta_data, recon_c, site, mu, logvar, beta=..., gamma=..., class_weights=...)
    optimizer.zero_grad(); loss.backward(); optimizer.step()

Data: `--data data/processed_data.pkl`-style files cannot be produced offline (the reference's
prepare scripts download from Kaggle); without `--data` a synthetic set with the same layout is
generated (tpm >= 0 float32 (N,A), beta in [0,1] float32 (N,D), site int64 (N,)).  The whole set
lives on the device; each epoch is a device-side permutation + row gather (no per-item
`torch.tensor`, reference src/data/dataset.py:28-39).

    python train.py --epochs 3 --batch-size 4096
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...
"""
import argparse
import os
import sys
import time
from datetime import datetime

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mmvae import parallel  # noqa: E402
from mmvae.optim import FusedAdamW  # noqa: E402
from src.config import Config  # noqa: E402
from src.models import MultiModalVAE  # noqa: E402
from src.utils import vae_loss  # noqa: E402


def balanced_class_weights(site, n_sites):
    """sklearn's compute_class_weight('balanced') (optimize_hyperparameters.py:33-44): n / (k * count)."""
    counts = torch.bincount(site, minlength=n_sites).clamp_min(1).float()
    return site.numel() / (n_sites * counts)


def synthetic_dataset(n, a_dim, d_dim, n_sites, seed):
    g = torch.Generator().manual_seed(seed)
    tpm = torch.log1p(torch.exp(1.5 * torch.randn(n, a_dim, generator=g)))          # log1p(TPM)-like, >= 0
    beta = torch.rand(n, d_dim, generator=g)
    site = torch.randint(0, n_sites, (n,), generator=g, dtype=torch.int64)
    return tpm, beta, site


def load_pickled_dataset(path):
    """Same columns as the reference's processed_data.pkl (src/data/dataset.py:20-30)."""
    import pandas as pd
    df = pd.read_pickle(path)          # a file the USER produced with the reference's own scripts
    tpm = torch.tensor(np.stack(df["tpm_unstranded"].values), dtype=torch.float32)
    beta = torch.tensor(np.stack(df["beta_value"].values), dtype=torch.float32)
    site = torch.tensor(df["primary_site_encoded"].values, dtype=torch.int64)
    return tpm, beta, site


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None)
    ap.add_argument("--samples", type=int, default=262144)
    ap.add_argument("--input-dim-a", type=int, default=int(os.getenv("INPUT_DIM_A", 782)))
    ap.add_argument("--input-dim-b", type=int, default=int(os.getenv("INPUT_DIM_B", 572)))
    ap.add_argument("--n-sites", type=int, default=24)
    ap.add_argument("--latent-dim", type=int, default=int(os.getenv("LATENT_DIM", Config.LATENT_DIM)))
    ap.add_argument("--batch-size", type=int, default=4096, help="rows per GPU")
    ap.add_argument("--epochs", type=int, default=Config.NUM_EPOCHS)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--checkpoint-dir", default=Config.CHECKPOINT_DIR)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    if args.data:
        tpm, beta_v, site = load_pickled_dataset(args.data)
        args.input_dim_a, args.input_dim_b = tpm.shape[1], beta_v.shape[1]
        args.n_sites = int(site.max()) + 1
    else:
        tpm, beta_v, site = synthetic_dataset(args.samples, args.input_dim_a, args.input_dim_b, args.n_sites, Config.RANDOM_SEED)
    n = tpm.shape[0]
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(Config.RANDOM_SEED))
    n_val = int(n * Config.TRAIN_TEST_SPLIT)
    val_idx, train_idx = perm[:n_val], perm[n_val:]
    lo, hi = parallel.shard_rows(train_idx.numel(), rank, world, equal=True)  # each rank owns a row shard of the SAME size: same step count
    tr = [t[train_idx[lo:hi]].to(dev) for t in (tpm, beta_v, site)]
    va = [t[val_idx].to(dev) for t in (tpm, beta_v, site)]
    class_weights = balanced_class_weights(site[train_idx], args.n_sites).to(dev)

    torch.manual_seed(Config.RANDOM_SEED)
    model = MultiModalVAE(args.input_dim_a, args.input_dim_b, args.n_sites, args.latent_dim).to(dev).set_precision(args.precision)
    if world > 1:
        parallel.broadcast_parameters(model)
        parallel.attach(model)
    optimizer = FusedAdamW(model.parameters(), lr=Config.LEARNING_RATE, weight_decay=Config.WEIGHT_DECAY)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=Config.LR_SCHEDULER_FACTOR,
                                                           patience=Config.LR_SCHEDULER_PATIENCE)
    run_id = datetime.now().strftime("%Y%m%d_%H%M%S")
    os.makedirs(args.checkpoint_dir, exist_ok=True)
    best_val, trigger = float("inf"), 0
    B = args.batch_size
    for epoch in range(args.epochs):
        model.train()
        beta = min(1.0, epoch / Config.BETA_WARMUP_EPOCHS) * Config.BETA_START
        order = torch.randperm(tr[0].shape[0], device=dev)
        t0, running, steps = time.time(), 0.0, 0
        for i in range(0, order.numel() - B + 1, B):                         # drop_last=True
            idx = order[i:i + B]
            a, b, s = tr[0][idx], tr[1][idx], tr[2][idx]
            recon_a, recon_b, recon_c, mu, logvar = model(a=a, b=b, site=s)
            loss, _, _, _ = vae_loss(recon_a, a, recon_b, b, recon_c, s, mu, logvar, beta=beta, gamma=Config.GAMMA,
                                     class_weights=class_weights)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            running += loss.item()
            steps += 1
        dt = time.time() - t0
        if world > 1:
            parallel.average_bn_buffers(model)                                # same eval-mode model on every rank
        model.eval()
        val_loss, vsteps = 0.0, 0
        with torch.no_grad():
            for i in range(0, va[0].shape[0], B):
                a, b, s = va[0][i:i + B], va[1][i:i + B], va[2][i:i + B]
                recon_a, recon_b, recon_c, mu, logvar = model(a=a, b=b, site=s)
                loss, _, _, _ = vae_loss(recon_a, a, recon_b, b, recon_c, s, mu, logvar, beta=beta, gamma=Config.GAMMA,
                                         class_weights=class_weights)
                val_loss += loss.item()
                vsteps += 1
        val_loss /= max(vsteps, 1)
        if world > 1:
            # eps is sampled in eval mode too (vae.py:73) and the Philox streams differ per rank: the scheduler / checkpoint /
            # early-stop decisions below must see ONE number on every rank or the ranks part ways (different LR, or a rank
            # leaving the loop while the others wait in the all-reduce)
            val_loss = parallel.all_ranks_mean(val_loss, dev)
        scheduler.step(val_loss)
        if rank == 0:
            print(f"Epoch [{epoch + 1}/{args.epochs}] | Train Loss: {running / max(steps, 1):.2f} | Val Loss: {val_loss:.2f} | "
                  f"beta={beta:.5f} | {world * steps * B / dt:,.0f} samples/s")
        if val_loss < best_val:
            best_val, trigger = val_loss, 0
            if rank == 0:
                torch.save(model.state_dict(), os.path.join(args.checkpoint_dir, f"best_multivae_{run_id}.pt"))
        else:
            trigger += 1
            if trigger >= Config.PATIENCE:
                break
    if rank == 0:
        print(f"Training complete. Run ID: {run_id}. Best validation loss: {best_val:.2f}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
