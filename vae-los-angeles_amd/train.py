#!/usr/bin/env python3
"""Reference-shaped trainer for MultiModalVAE on MI355X.

The reference's README and run_pipeline.sh call a `train.py` that is absent from its tree; this is that harness,
with the loop of the one in-tree MultiModalVAE trainer (reference optimize_hyperparameters.py:163-211: AdamW + beta warm-up,
per-epoch validation, balanced class weights, best-checkpoint `torch.save(state_dict)`) plus the ReduceLROnPlateau / early-stop /
run-id conventions of train_dna2rna.py:150-252.

    python train.py --epochs 3 --batch-size 4096
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...
Options, loop shape and what runs differently on the MI355X: trainer.py."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from trainer import run, balanced_class_weights, synthetic_dataset  # noqa: E402,F401

if __name__ == "__main__":
    run("multimodal")
