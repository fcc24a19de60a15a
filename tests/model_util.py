"""Shared helpers for the GPU parity tests (model construction from oracle parameters, error
metrics).  Test infrastructure only."""
import numpy as np
import torch

import np_oracle as O

CHAOTIC_BIASES = ("encoder_a.fc.0.bias", "encoder_b.fc.0.bias", "encoder_b.fc.4.bias")


def load_state(model, P, Bf, rename=None):
    """Load oracle-named parameters/buffers into a drop-in module (optionally through the
    directional models' sub-module renaming)."""
    sd = {}
    for k, v in list(P.items()) + list(Bf.items()):
        kk = k
        if rename is not None:
            top, rest = k.split(".", 1)
            if top not in rename:
                continue
            kk = rename[top] + "." + rest
        sd[kk] = torch.from_numpy(np.array(v, dtype=np.int64 if np.asarray(v).dtype == np.int64 else np.float32))
    model.load_state_dict(sd, strict=True)
    return model


def masks_list(masks, which=("encoder_a.fc.3", "encoder_b.fc.3", "encoder_b.fc.7")):
    return [torch.from_numpy(masks[k]) for k in which]


def scaled_err(got, ref):
    """max |got - ref| / max |ref|  (error relative to the tensor's scale)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    s = float(np.max(np.abs(ref))) if ref.size else 1.0
    return float(np.max(np.abs(got - ref))) / (s if s > 0 else 1.0)


def named_grads(model, rename=None):
    inv = {v: k for k, v in rename.items()} if rename else None
    out = {}
    for k, p in model.named_parameters():
        if inv is not None:
            top, rest = k.split(".", 1)
            k = inv[top] + "." + rest
        out[k] = None if p.grad is None else p.grad.detach().float().cpu().numpy()
    return out


def f64(d):
    return O.cast_tree(d, np.float64)
