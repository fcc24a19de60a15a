"""rna2dna_loss / dna2rna_loss (reference src/utils/directional_losses.py:8-55): the BCE+KL and
MSE+KL subsets of the fused loss pass."""
from mmvae import functional as F_


def rna2dna_loss(recon_dna, dna, mu, logvar, beta=1e-3):
    """Returns (total_loss Tensor, reconstruction_loss float, kl_divergence float)."""
    total, out4 = F_.fused_loss({"b": (recon_dna, dna), "kl": (mu, logvar)}, float(beta), 1.0)
    vals = F_.read_losses(out4)
    return total, vals[1], vals[3]


def dna2rna_loss(recon_rna, rna, mu, logvar, beta=1e-3):
    """Returns (total_loss Tensor, reconstruction_loss float, kl_divergence float)."""
    total, out4 = F_.fused_loss({"a": (recon_rna, rna), "kl": (mu, logvar)}, float(beta), 1.0)
    vals = F_.read_losses(out4)
    return total, vals[1], vals[3]
