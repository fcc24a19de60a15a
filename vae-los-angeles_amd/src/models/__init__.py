"""Models package (hot-path subset of reference src/models/__init__.py:4-17; the
non-variational AE baselines and the kNN comparator are out of scope)."""
from .vae import MultiModalVAE, reparameterize
from .encoders import EncoderA, EncoderB, EncoderC
from .decoders import DecoderA, DecoderB, DecoderC
from .directional_vae import RNA2DNAVAE, DNA2RNAVAE

__all__ = [
    'MultiModalVAE',
    'reparameterize',
    'EncoderA', 'EncoderB', 'EncoderC',
    'DecoderA', 'DecoderB', 'DecoderC',
    'RNA2DNAVAE', 'DNA2RNAVAE',
]
