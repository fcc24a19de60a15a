#!/usr/bin/env python3
"""Training script for RNA2DNAVAE (reference train_rna2dna.py, the same loop): model(rna=, site=) + rna2dna_loss (BCE + beta*KL).

    python train_rna2dna.py --epochs 3 --batch-size 4096
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_rna2dna.py ...
Options, loop shape and what runs differently on the MI355X: trainer.py."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from trainer import run, balanced_class_weights, synthetic_dataset  # noqa: E402,F401

if __name__ == "__main__":
    run("rna2dna")
