"""DecoderA / DecoderB / DecoderC (reference src/models/decoders.py:8-50) on the MI355X
kernels: every Linear is one MFMA GEMM with bias + ReLU / Sigmoid in the epilogue."""
import torch.nn as nn

from mmvae import engine, functional as F_
from ._common import HipModule, relu_chain


class _MLPDecoder(HipModule):
    _LIN, _SIGMOID = (), False

    def _block(self):
        blk = getattr(self, "_blk", None)
        if blk is None:
            blk = engine.DecoderMLP([self.fc[i] for i in self._LIN], self._SIGMOID, name=type(self).__name__)
            object.__setattr__(self, "_blk", blk)
            object.__setattr__(self, "_rt", F_.BlockRuntime(blk))
        return blk

    def forward(self, z):
        blk = self._block()
        return F_.DecoderFn.apply(self._rt, self._prec(), z, *blk.params())


class DecoderA(_MLPDecoder):
    """Decoder for RNA expression data (modality A): latent -> 128 -> out."""
    _LIN = (0, 2)

    def __init__(self, latent_dim, output_dim):
        super().__init__()
        self.fc = relu_chain([(latent_dim, 128), (128, output_dim)])


class DecoderB(_MLPDecoder):
    """Decoder for DNA methylation data (modality B): latent -> 256 -> 512 -> out, Sigmoid."""
    _LIN, _SIGMOID = (0, 2, 4), True

    def __init__(self, latent_dim, output_dim):
        super().__init__()
        self.fc = relu_chain([(latent_dim, 256), (256, 512), (512, output_dim)], final=nn.Sigmoid())


class DecoderC(_MLPDecoder):
    """Decoder for primary site classification (modality C): latent -> 64 -> n_sites logits."""
    _LIN = (0, 2)

    def __init__(self, latent_dim, n_sites):
        super().__init__()
        self.fc = relu_chain([(latent_dim, 64), (64, n_sites)])
