"""MultiModalVAE with the reference's surface (reference src/models/vae.py:11-79): same
constructor, same sub-module names (state_dict ABI), same forward kwargs and return tuple.
One forward = one autograd node whose forward/backward are launch sequences into
libmmvae_hip.so (mmvae.engine.VAEGraph)."""
import torch

from mmvae import engine, functional as F_
from ._common import HipModule
from .encoders import EncoderA, EncoderB, EncoderC
from .decoders import DecoderA, DecoderB, DecoderC


def reparameterize(mu, logvar):
    """z = mu + eps * exp(0.5*logvar), eps ~ N(0,1) sampled in train AND eval (vae.py:11-15).
    Stand-alone helper: eps comes from the device Philox stream and the arithmetic runs in
    mmvae_fuse_reparam_fwd / _bwd with one modality (inside the models the same launch also takes
    the modality mean).  CPU tensors raise: there is no CPU fallback."""
    return F_.ReparamFn.apply(mu, logvar)


class MultiModalVAE(HipModule):
    """Multi-Modal VAE over RNA expression (a), DNA methylation (b) and primary site labels."""

    def __init__(self, input_dim_a, input_dim_b, n_sites, latent_dim, embed_dim=32):
        super().__init__()
        self.encoder_a = EncoderA(input_dim_a, latent_dim)
        self.encoder_b = EncoderB(input_dim_b, latent_dim)
        self.encoder_c = EncoderC(n_sites, latent_dim, embed_dim=embed_dim)

        self.decoder_a = DecoderA(latent_dim, input_dim_a)
        self.decoder_b = DecoderB(latent_dim, input_dim_b)
        self.decoder_c = DecoderC(latent_dim, n_sites)

    def _graph(self):
        g = getattr(self, "_g", None)
        if g is None:
            g = engine.VAEGraph(self.encoder_a._block(), self.encoder_b._block(), self.encoder_c._block(),
                                [self.decoder_a._block(), self.decoder_b._block(), self.decoder_c._block()])
            object.__setattr__(self, "_g", g)
        return g

    def forward(self, a=None, b=None, site=None):
        """Returns (out_a, out_b, out_c, mu, logvar); five Nones when no modality is given."""
        if a is None and b is None and site is None:
            return None, None, None, None, None
        outs, mu, logvar = F_.run_graph(self._graph(), self._prec(), self.training, a, b, site)
        return outs[0], outs[1], outs[2], mu, logvar
