"""EncoderA / EncoderB / EncoderC with the reference's constructor signatures, attribute names
and state_dict keys (reference src/models/encoders.py:8-61), computed by the MI355X kernels:
each Linear -> BatchNorm1d -> ReLU -> Dropout stage is one MFMA GEMM that also emits the batch
statistics, with normalise/ReLU/dropout folded into the NEXT GEMM's operand load; fc_mu and
fc_logvar are one GEMM."""
import torch.nn as nn

from mmvae import engine, functional as F_
from ._common import HipModule, hidden_stack


class _MLPEncoder(HipModule):
    _LIN, _BN = (), ()

    def _block(self):
        blk = getattr(self, "_blk", None)
        if blk is None:
            blk = engine.EncoderMLP([self.fc[i] for i in self._LIN], [self.fc[i] for i in self._BN],
                                    self.fc_mu, self.fc_logvar, name=type(self).__name__)
            object.__setattr__(self, "_blk", blk)
            object.__setattr__(self, "_rt", F_.BlockRuntime(blk))
        return blk

    def forward(self, x):
        blk = self._block()
        return F_.EncoderMLPFn.apply(self._rt, self._prec(), self.training, engine.GLOBAL_NOISE, x, *blk.params())


class EncoderA(_MLPEncoder):
    """Encoder for RNA expression data (modality A): in -> 128 -> (mu, logvar)."""
    _LIN, _BN = (0,), (1,)

    def __init__(self, input_dim, latent_dim):
        super().__init__()
        self.fc = hidden_stack([(input_dim, 128)])
        self.fc_mu = nn.Linear(128, latent_dim)
        self.fc_logvar = nn.Linear(128, latent_dim)


class EncoderB(_MLPEncoder):
    """Encoder for DNA methylation data (modality B): in -> 512 -> 256 -> (mu, logvar)."""
    _LIN, _BN = (0, 4), (1, 5)

    def __init__(self, input_dim, latent_dim):
        super().__init__()
        self.fc = hidden_stack([(input_dim, 512), (512, 256)])
        self.fc_mu = nn.Linear(256, latent_dim)
        self.fc_logvar = nn.Linear(256, latent_dim)


class EncoderC(HipModule):
    """Encoder for primary site labels (modality C): Embedding -> (mu, logvar)."""

    def __init__(self, n_sites, latent_dim, embed_dim=32):
        super().__init__()
        self.embedding = nn.Embedding(n_sites, embed_dim)
        self.fc_mu = nn.Linear(embed_dim, latent_dim)
        self.fc_logvar = nn.Linear(embed_dim, latent_dim)

    def _block(self):
        blk = getattr(self, "_blk", None)
        if blk is None:
            blk = engine.EmbedEncoder(self.embedding, self.fc_mu, self.fc_logvar)
            object.__setattr__(self, "_blk", blk)
        return blk

    def forward(self, x):
        blk = self._block()
        return F_.EmbedEncoderFn.apply(blk, x, *blk.params())
