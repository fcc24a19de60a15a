#!/usr/bin/env python3
"""Training script for DNA2RNAVAE (reference train_dna2rna.py:72-252): model(dna=, site=) + dna2rna_loss (MSE + beta*KL).

    python train_dna2rna.py --epochs 3 --batch-size 4096
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_dna2rna.py ...
Options, loop shape and what runs differently on the MI355X: trainer.py."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from trainer import run, balanced_class_weights, synthetic_dataset  # noqa: E402,F401

if __name__ == "__main__":
    run("dna2rna")
