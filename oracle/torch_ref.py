"""Stock-PyTorch CPU restatement of the MultiModalVAE training step -- TEST INFRASTRUCTURE ONLY.

Used (a) as the timed `cpu_baseline` ("port") of bench.py on the GPU node's host cores and
(b) as a second, autograd-based checker next to np_oracle.py.  It is a FUNCTIONAL restatement
over a flat dict of tensors keyed by the reference's state_dict names (no nn.Module mirror of
the reference classes); every op is a stock torch op, as in the reference:
  encoders  src/models/encoders.py:12-23,30-46,53-61   (F.linear, F.batch_norm, relu, F.dropout)
  fusion    src/models/vae.py:65-73                     (stack/mean, randn_like, exp)
  decoders  src/models/decoders.py:12-16,26-33,43-47
  loss      src/utils/losses.py:31,34,39,42,44
  optimiser torch.optim.AdamW as constructed at optimize_hyperparameters.py:93-97
Pinned against tests/golden/*.npz by tests/test_oracle_vs_golden.py::test_torch_ref_*.
"""
import numpy as np
import torch
import torch.nn.functional as F

import np_oracle as O


def to_torch(P, Bf, requires_grad=True):
    params = {k: torch.tensor(np.asarray(v), dtype=torch.float32, requires_grad=requires_grad) for k, v in P.items()}
    bufs = {k: torch.tensor(np.asarray(v)) for k, v in Bf.items()}
    return params, bufs


def _stage(h, p, bufs, pre, li, bi, train, mask):
    y = F.linear(h, p[f"{pre}.fc.{li}.weight"], p[f"{pre}.fc.{li}.bias"])
    base = f"{pre}.fc.{bi}"
    y = F.batch_norm(y, bufs[base + ".running_mean"], bufs[base + ".running_var"], p[base + ".weight"], p[base + ".bias"],
                     training=train, momentum=0.1, eps=1e-5)
    if train:
        bufs[base + ".num_batches_tracked"] += 1
    y = torch.relu(y)
    if train:
        y = y * mask / (1.0 - O.DROP_P) if mask is not None else F.dropout(y, O.DROP_P, True)
    return y


def _encoder(x, p, bufs, pre, idx, train, masks):
    h = x
    for (li, bi, di) in idx:
        m = None if masks is None else masks[f"{pre}.fc.{di}"]
        h = _stage(h, p, bufs, pre, li, bi, train, m)
    return F.linear(h, p[f"{pre}.fc_mu.weight"], p[f"{pre}.fc_mu.bias"]), F.linear(h, p[f"{pre}.fc_logvar.weight"], p[f"{pre}.fc_logvar.bias"])


def _decoder(z, p, pre, idxs, sigmoid):
    h = z
    for j, li in enumerate(idxs):
        h = F.linear(h, p[f"{pre}.fc.{li}.weight"], p[f"{pre}.fc.{li}.bias"])
        if j < len(idxs) - 1:
            h = torch.relu(h)
    return torch.sigmoid(h) if sigmoid else h


def forward(p, bufs, a=None, b=None, site=None, train=True, masks=None, eps=None):
    """masks: dict name -> float tensor (1 = kept) or None for torch's own dropout RNG."""
    mus, lvs = [], []
    if a is not None:
        m, l = _encoder(a, p, bufs, "encoder_a", O.ENC_A_IDX, train, masks)
        mus.append(m); lvs.append(l)
    if b is not None:
        m, l = _encoder(b.view(b.size(0), -1), p, bufs, "encoder_b", O.ENC_B_IDX, train, masks)
        mus.append(m); lvs.append(l)
    if site is not None:
        h = F.embedding(site, p["encoder_c.embedding.weight"])
        mus.append(F.linear(h, p["encoder_c.fc_mu.weight"], p["encoder_c.fc_mu.bias"]))
        lvs.append(F.linear(h, p["encoder_c.fc_logvar.weight"], p["encoder_c.fc_logvar.bias"]))
    if not mus:
        return None, None, None, None, None
    mu = mus[0] if len(mus) == 1 else torch.stack(mus).mean(0)
    lv = lvs[0] if len(lvs) == 1 else torch.stack(lvs).mean(0)
    std = torch.exp(0.5 * lv)
    z = mu + (torch.randn_like(std) if eps is None else eps) * std
    return (_decoder(z, p, "decoder_a", [0, 2], False), _decoder(z, p, "decoder_b", [0, 2, 4], True),
            _decoder(z, p, "decoder_c", [0, 2], False), mu, lv)


def loss_fn(ra, a, rb, b, rc, site, mu, lv, beta=1e-3, gamma=1.0, class_weights=None):
    recon = F.mse_loss(ra, a, reduction="sum") + F.binary_cross_entropy(rb, b, reduction="sum")
    cls = F.cross_entropy(rc, site, weight=class_weights, reduction="sum")
    kld = -0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp())
    return recon + gamma * cls + beta * kld, recon, cls, kld


class CpuTrainer:
    """The reference-shaped step (optimize_hyperparameters.py:104-113) on the host cores."""

    def __init__(self, A, D, S, L, E=32, seed=0, lr=5e-4, wd=1e-5):
        P, Bf = O.make_params(seed, A, D, S, L, E)
        self.p, self.bufs = to_torch(P, Bf)
        self.opt = torch.optim.AdamW(list(self.p.values()), lr=lr, weight_decay=wd)

    def step(self, a, b, site, beta=1e-3, gamma=1.0):
        ra, rb, rc, mu, lv = forward(self.p, self.bufs, a, b, site, True)
        loss, recon, cls, kld = loss_fn(ra, a, rb, b, rc, site, mu, lv, beta, gamma)
        floats = (recon.item(), cls.item(), kld.item())       # the reference's three host reads
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.item(), floats
