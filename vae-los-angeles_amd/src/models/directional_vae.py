"""RNA2DNAVAE / DNA2RNAVAE (reference src/models/directional_vae.py:12-111): two encoders, mean
fusion, reparameterisation, one decoder -- strict sub-graphs of the MultiModalVAE launch
sequence."""
from mmvae import engine, functional as F_
from ._common import HipModule
from .encoders import EncoderA, EncoderB, EncoderC
from .decoders import DecoderA, DecoderB


class RNA2DNAVAE(HipModule):
    """Encodes RNA + primary site, decodes DNA methylation."""

    def __init__(self, rna_dim, dna_dim, n_sites, latent_dim, embed_dim=32):
        super().__init__()
        self.encoder_rna = EncoderA(rna_dim, latent_dim)
        self.encoder_site = EncoderC(n_sites, latent_dim, embed_dim=embed_dim)
        self.decoder_dna = DecoderB(latent_dim, dna_dim)

    def _graph(self):
        g = getattr(self, "_g", None)
        if g is None:
            g = engine.VAEGraph(enc_a=self.encoder_rna._block(), enc_c=self.encoder_site._block(),
                                decoders=[self.decoder_dna._block()])
            object.__setattr__(self, "_g", g)
        return g

    def forward(self, rna=None, site=None):
        """Returns (reconstructed_dna, mu, logvar); three Nones without inputs."""
        if rna is None and site is None:
            return None, None, None
        outs, mu, logvar = F_.run_graph(self._graph(), self._prec(), self.training, rna, None, site)
        return outs[0], mu, logvar


class DNA2RNAVAE(HipModule):
    """Encodes DNA methylation + primary site, decodes RNA expression."""

    def __init__(self, rna_dim, dna_dim, n_sites, latent_dim, embed_dim=32):
        super().__init__()
        self.encoder_dna = EncoderB(dna_dim, latent_dim)
        self.encoder_site = EncoderC(n_sites, latent_dim, embed_dim=embed_dim)
        self.decoder_rna = DecoderA(latent_dim, rna_dim)

    def _graph(self):
        g = getattr(self, "_g", None)
        if g is None:
            g = engine.VAEGraph(enc_b=self.encoder_dna._block(), enc_c=self.encoder_site._block(),
                                decoders=[self.decoder_rna._block()])
            object.__setattr__(self, "_g", g)
        return g

    def forward(self, dna=None, site=None):
        """Returns (reconstructed_rna, mu, logvar); three Nones without inputs."""
        if dna is None and site is None:
            return None, None, None
        outs, mu, logvar = F_.run_graph(self._graph(), self._prec(), self.training, None, dna, site)
        return outs[0], mu, logvar
