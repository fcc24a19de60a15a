"""Shared plumbing of the drop-in modules: precision selection and parameter containers."""
import torch.nn as nn

from mmvae import engine


def hidden_stack(widths):
    """nn.Sequential of [Linear, BatchNorm1d, ReLU, Dropout(0.1)] per (in, out) pair.

    Used ONLY as a parameter/buffer container: it fixes the state_dict keys (fc.0.*, fc.1.*,
    fc.4.*, fc.5.*: the reference's checkpoint ABI) and the
    initialisation order; forward never calls it."""
    layers = []
    for fan_in, fan_out in widths:
        layers += [nn.Linear(fan_in, fan_out), nn.BatchNorm1d(fan_out), nn.ReLU(), nn.Dropout(0.1)]
    return nn.Sequential(*layers)


def relu_chain(widths, final=None):
    """nn.Sequential container of Linear(+ReLU) pairs (keys fc.0, fc.2, fc.4)."""
    layers = []
    for i, (fan_in, fan_out) in enumerate(widths):
        layers.append(nn.Linear(fan_in, fan_out))
        if i < len(widths) - 1:
            layers.append(nn.ReLU())
    if final is not None:
        layers.append(final)
    return nn.Sequential(*layers)


class HipModule(nn.Module):
    """nn.Module whose forward runs on the HIP kernels.  `precision` is 'bf16' (default; bf16
    MFMA operands and stored activations, f32 accumulation/statistics) or 'fp32'."""

    precision = None          # None -> mmvae.engine.default_precision()

    def _prec(self):
        if self.precision is None:
            return engine.default_precision()
        return engine._PRECISIONS[self.precision.lower()]

    def set_precision(self, name):
        self.precision = name
        return self
