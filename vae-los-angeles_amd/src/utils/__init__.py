"""Utils package (reference src/utils/__init__.py:4-6)."""
from .losses import vae_loss

__all__ = ['vae_loss']
