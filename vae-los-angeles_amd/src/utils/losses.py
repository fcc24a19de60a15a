"""vae_loss with the reference's signature and return tuple (reference
src/utils/losses.py:8-46), computed by ONE fused HIP pass (mmvae_vae_loss): sum-MSE + clamped
sum-BCE + gamma * weighted sum-CE + beta * KL, plus the gradients w.r.t. its inputs."""
from mmvae import functional as F_


def vae_loss(recon_a, a, recon_b, b, recon_c, site, mu, logvar, beta=1e-3, gamma=1.0, class_weights=None):
    """Returns (total_loss Tensor, reconstruction_loss float, classification_loss float,
    kl_divergence float).  As in the reference, the three floats cost a device->host sync;
    here it is ONE sync for all three (they are read from one 16-byte buffer)."""
    terms = {"kl": (mu, logvar)}
    if recon_a is not None and a is not None:
        terms["a"] = (recon_a, a)
    if recon_b is not None and b is not None:
        terms["b"] = (recon_b, b)
    if recon_c is not None and site is not None:
        terms["c"] = (recon_c, site)
    if "a" not in terms and "b" not in terms or "c" not in terms:
        # the reference ends in `recon.item()` / `class_loss.item()` on a Python int here
        raise AttributeError("'int' object has no attribute 'item'")
    total, out4 = F_.fused_loss(terms, float(beta), float(gamma), class_weights)
    vals = F_.read_losses(out4)
    return total, vals[1], vals[2], vals[3]
