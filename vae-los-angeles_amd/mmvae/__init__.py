"""Host-side runtime of the MI355X-native MultiModalVAE training path.

Everything numerical runs in libmmvae_hip.so (hand-written HIP for gfx950, C ABI in
include/mmvae_hip.h); this package only owns device buffers (through torch), orders the
launches, and exposes them to the reference-shaped modules in `src/`.
"""
from . import _lib                                    # noqa: F401
from ._lib import PREC_BF16, PREC_F32, MMVAELibraryError  # noqa: F401
