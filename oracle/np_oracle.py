"""CPU oracle for the MultiModalVAE training hot path -- TEST INFRASTRUCTURE ONLY.

This is a numpy restatement (explicit forward, hand-derived backward, AdamW) of the
algorithm of marcin119a/vae-los-angeles for the path named in BASELINE.json.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import it;
the product package (`vae-los-angeles_amd/`) never does.

Parity pinning: the reference ships no golden vectors, so this oracle
is pinned by fixtures generated in the build container from the imported reference
(`oracle/make_fixtures.py` -> `tests/golden/*.npz`, checked by
`tests/test_oracle_vs_golden.py`).

Every function cites the reference file:line it follows (paths relative to the
reference repo root).  Parameters are passed as a dict keyed by the reference's
state_dict names, e.g. ``encoder_a.fc.0.weight``.

Noise is an INPUT here: dropout keep-masks (1 = kept) and the reparameterisation eps
are passed in explicitly, in the order the reference consumes its RNG
(EncoderA mask (B,128) -> EncoderB masks (B,512),(B,256) -> eps (B,L)).
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5          # torch.nn.BatchNorm1d default, src/models/encoders.py:14
BN_MOMENTUM = 0.1      # torch.nn.BatchNorm1d default
DROP_P = 0.1           # src/models/encoders.py:16,34,38
BCE_LOG_CLAMP = -100.0  # F.binary_cross_entropy clamps log terms, src/utils/losses.py:34

HID_A = 128            # src/models/encoders.py:13
HID_B1, HID_B2 = 512, 256   # src/models/encoders.py:31,35
DEC_A_H = 128          # src/models/decoders.py:13
DEC_B_H1, DEC_B_H2 = 256, 512  # src/models/decoders.py:27,29
DEC_C_H = 64           # src/models/decoders.py:44


# --------------------------------------------------------------------------- #
# storage-rounding model of the bf16 engine mode                               #
# --------------------------------------------------------------------------- #
def bf16_round(x):
    """Round-to-nearest-even to bfloat16, returned in x's own dtype (what `(bf16)v` / v_cvt_pk_bf16_f32 does to an
    fp32 value on the device)."""
    x = np.asarray(x)
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(x.dtype)


def _id(x):
    return x


BF16 = bf16_round
"""Pass `q=BF16` to the forward / loss / backward functions below to get the *bf16-aware* oracle: the same algorithm
with exactly the tensors rounded that the MI355X path stores or multiplies in bf16 (DESIGN.md section 3) --
MFMA operands (first-layer inputs, prepared weights W and W^T, recomputed BN->ReLU->Dropout activations), the stored
pre-BatchNorm outputs y, hidden decoder activations, z, the loss gradients w.r.t. recon_a and the DecoderB logits, and
every dL/dy that is written back between two backward GEMMs.  Everything else (accumulation, BatchNorm statistics,
heads, mu/logvar, reconstructions, loss sums, parameter gradients, AdamW) stays in the oracle's working precision,
as it stays fp32 on the device.  `q=None` (default) is the reference's arithmetic, pinned by tests/golden."""


# --------------------------------------------------------------------------- #
# deterministic parameter construction (shared by fixtures, tests and bench)   #
# --------------------------------------------------------------------------- #
def param_shapes(A, D, S, L, E=32, prefix_map=None):
    """state_dict layout of MultiModalVAE (src/models/vae.py:27-35), creation order."""
    sh = [
        ("encoder_a.fc.0.weight", (HID_A, A)), ("encoder_a.fc.0.bias", (HID_A,)),
        ("encoder_a.fc.1.weight", (HID_A,)), ("encoder_a.fc.1.bias", (HID_A,)),
        ("encoder_a.fc_mu.weight", (L, HID_A)), ("encoder_a.fc_mu.bias", (L,)),
        ("encoder_a.fc_logvar.weight", (L, HID_A)), ("encoder_a.fc_logvar.bias", (L,)),
        ("encoder_b.fc.0.weight", (HID_B1, D)), ("encoder_b.fc.0.bias", (HID_B1,)),
        ("encoder_b.fc.1.weight", (HID_B1,)), ("encoder_b.fc.1.bias", (HID_B1,)),
        ("encoder_b.fc.4.weight", (HID_B2, HID_B1)), ("encoder_b.fc.4.bias", (HID_B2,)),
        ("encoder_b.fc.5.weight", (HID_B2,)), ("encoder_b.fc.5.bias", (HID_B2,)),
        ("encoder_b.fc_mu.weight", (L, HID_B2)), ("encoder_b.fc_mu.bias", (L,)),
        ("encoder_b.fc_logvar.weight", (L, HID_B2)), ("encoder_b.fc_logvar.bias", (L,)),
        ("encoder_c.embedding.weight", (S, E)),
        ("encoder_c.fc_mu.weight", (L, E)), ("encoder_c.fc_mu.bias", (L,)),
        ("encoder_c.fc_logvar.weight", (L, E)), ("encoder_c.fc_logvar.bias", (L,)),
        ("decoder_a.fc.0.weight", (DEC_A_H, L)), ("decoder_a.fc.0.bias", (DEC_A_H,)),
        ("decoder_a.fc.2.weight", (A, DEC_A_H)), ("decoder_a.fc.2.bias", (A,)),
        ("decoder_b.fc.0.weight", (DEC_B_H1, L)), ("decoder_b.fc.0.bias", (DEC_B_H1,)),
        ("decoder_b.fc.2.weight", (DEC_B_H2, DEC_B_H1)), ("decoder_b.fc.2.bias", (DEC_B_H2,)),
        ("decoder_b.fc.4.weight", (D, DEC_B_H2)), ("decoder_b.fc.4.bias", (D,)),
        ("decoder_c.fc.0.weight", (DEC_C_H, L)), ("decoder_c.fc.0.bias", (DEC_C_H,)),
        ("decoder_c.fc.2.weight", (S, DEC_C_H)), ("decoder_c.fc.2.bias", (S,)),
    ]
    return sh


BN_LAYERS = {"encoder_a.fc.1": HID_A, "encoder_b.fc.1": HID_B1, "encoder_b.fc.5": HID_B2}


def make_params(seed, A, D, S, L, E=32, dtype=np.float32):
    """Deterministic numpy initialisation with the same *distributions* as torch's
    defaults (Linear: U(+-1/sqrt(fan_in)) for W and b; Embedding: N(0,1); BN gamma=1,
    beta=0 -- SURVEY.md section 8b).  BN gamma/beta are perturbed away from (1,0) so that
    fixtures exercise them.  Returns (params, buffers)."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, shape in param_shapes(A, D, S, L, E):
        base = name.rsplit(".", 1)[0]
        if base in BN_LAYERS:
            if name.endswith("weight"):
                params[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(dtype)
            else:
                params[name] = (0.1 * rng.standard_normal(shape)).astype(dtype)
        elif name == "encoder_c.embedding.weight":
            params[name] = rng.standard_normal(shape).astype(dtype)
        else:
            fan_in = shape[1] if len(shape) == 2 else params[base + ".weight"].shape[1]
            bound = 1.0 / np.sqrt(fan_in)
            params[name] = rng.uniform(-bound, bound, size=shape).astype(dtype)
    buffers = {}
    for base, n in BN_LAYERS.items():
        buffers[base + ".running_mean"] = np.zeros(n, dtype)
        buffers[base + ".running_var"] = np.ones(n, dtype)
        buffers[base + ".num_batches_tracked"] = np.zeros((), np.int64)
    return params, buffers


def make_batch(seed, B, A, D, S, dtype=np.float32):
    """Synthetic batch as SURVEY.md section 8(d): a = |N(0,1)|, b ~ U(0,1), site ~ U{0..S-1}."""
    rng = np.random.default_rng(seed)
    a = np.abs(rng.standard_normal((B, A))).astype(dtype)
    b = rng.uniform(0.0, 1.0, size=(B, D)).astype(dtype)
    site = rng.integers(0, S, size=(B,), dtype=np.int64)
    return a, b, site


def make_noise(seed, B, L, dtype=np.float32):
    """Dropout keep-masks (uint8, 1 = kept, P(keep)=0.9) and eps ~ N(0,1)."""
    rng = np.random.default_rng(seed)
    masks = {
        "encoder_a.fc.3": (rng.uniform(size=(B, HID_A)) < 1.0 - DROP_P).astype(np.uint8),
        "encoder_b.fc.3": (rng.uniform(size=(B, HID_B1)) < 1.0 - DROP_P).astype(np.uint8),
        "encoder_b.fc.7": (rng.uniform(size=(B, HID_B2)) < 1.0 - DROP_P).astype(np.uint8),
    }
    eps = rng.standard_normal((B, L)).astype(dtype)
    return masks, eps


# --------------------------------------------------------------------------- #
# layer primitives                                                             #
# --------------------------------------------------------------------------- #
def _linear(x, W, b):
    """nn.Linear: y = x W^T + b  (e.g. src/models/encoders.py:13)."""
    return x @ W.T + b


def _linear_bwd(dy, x, W):
    """Returns (dx, dW, db) of y = x W^T + b."""
    return dy @ W, dy.T @ x, dy.sum(0)


def _bn_relu_drop_fwd(y, gamma, beta, mask, train, rm, rv):
    """Linear -> BatchNorm1d -> ReLU -> Dropout(0.1)  (src/models/encoders.py:12-17).

    train: batch mean / biased variance, eps 1e-5; eval: running stats, no dropout.
    Returns (h, cache)."""
    if train:
        if y.shape[0] < 2:
            # torch raises for B == 1 in training mode (Errors)
            raise ValueError("Expected more than 1 value per channel when training")
        mean = y.mean(0)
        var = y.var(0)            # biased
    else:
        mean, var = rm, rv
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    xhat = (y - mean) * rstd
    yh = xhat * gamma + beta
    r = np.maximum(yh, 0.0)
    if train:
        keep = mask.astype(y.dtype) / (1.0 - DROP_P)
        h = r * keep
    else:
        keep = None
        h = r
    return h, dict(xhat=xhat, rstd=rstd, gamma=gamma, yh=yh, keep=keep, mean=mean, var=var)


def _bn_relu_drop_bwd(dh, c, q=None):
    """Backward of _bn_relu_drop_fwd (train mode, or eval mode when the cache has no keep mask). Returns (dy, dgamma, dbeta).
    q: the device reduces (sum d, sum d*xhat) from the fp32 accumulators, stores d rounded, applies the
    correction to the stored d and rounds the result once more (EpiBnBwd phase 2 + mmvae_bn_bwd_apply)."""
    q = q or _id
    B = dh.shape[0]
    if c["keep"] is None:
        # eval-mode forward (running statistics, no dropout): torch's native_batch_norm_backward with train=False --
        # the statistics are constants, so dy = gamma * rstd * d and dgamma / dbeta are the plain sums
        dyh = dh * (c["yh"] > 0)
        return q(c["gamma"] * c["rstd"] * q(dyh)), (dyh * c["xhat"]).sum(0), dyh.sum(0)
    dyh = dh * c["keep"] * (c["yh"] > 0)
    dgamma = (dyh * c["xhat"]).sum(0)
    dbeta = dyh.sum(0)
    dy = q(c["gamma"] * c["rstd"] * (q(dyh) - dbeta / B - c["xhat"] * dgamma / B))
    return dy, dgamma, dbeta


def _bn_running_update(buffers, base, mean, var, B):
    """BatchNorm1d running-stat update: momentum 0.1, UNBIASED variance."""
    dt = buffers[base + ".running_mean"].dtype
    buffers[base + ".running_mean"] = ((1 - BN_MOMENTUM) * buffers[base + ".running_mean"]
                                       + BN_MOMENTUM * mean).astype(dt)
    buffers[base + ".running_var"] = ((1 - BN_MOMENTUM) * buffers[base + ".running_var"]
                                      + BN_MOMENTUM * var * (B / (B - 1.0))).astype(dt)
    buffers[base + ".num_batches_tracked"] = buffers[base + ".num_batches_tracked"] + 1


# --------------------------------------------------------------------------- #
# encoders / decoders                                                          #
# --------------------------------------------------------------------------- #
def encoder_mlp_fwd(P, Bf, pre, bn_idx, x, masks, train, update_running=True, q=None):
    """EncoderA (bn_idx=[(0,1,3)]) / EncoderB (bn_idx=[(0,1,3),(4,5,7)]).

    src/models/encoders.py:21-23 and :43-46.  `pre` is e.g. 'encoder_a'."""
    q = q or _id
    caches = []
    h = q(x)
    for (li, bi, di) in bn_idx:
        W, b = q(P[f"{pre}.fc.{li}.weight"]), P[f"{pre}.fc.{li}.bias"]
        y = q(_linear(h, W, b))               # BatchNorm sees the stored (rounded) pre-activation
        base = f"{pre}.fc.{bi}"
        hn, c = _bn_relu_drop_fwd(y, P[base + ".weight"], P[base + ".bias"],
                                  masks.get(f"{pre}.fc.{di}") if masks else None, train,
                                  Bf[base + ".running_mean"], Bf[base + ".running_var"])
        if train and update_running:
            _bn_running_update(Bf, base, c["mean"], c["var"], x.shape[0])
        c.update(x=h, W=W, li=li, bi=bi)
        caches.append(c)
        h = q(hn)                             # MFMA operand of the consumer GEMM
    mu = _linear(h, q(P[f"{pre}.fc_mu.weight"]), P[f"{pre}.fc_mu.bias"])
    lv = _linear(h, q(P[f"{pre}.fc_logvar.weight"]), P[f"{pre}.fc_logvar.bias"])
    return mu, lv, dict(layers=caches, h=h, pre=pre)


def encoder_mlp_bwd(P, cache, dmu, dlv, G, q=None):
    """Backward of encoder_mlp_fwd (train mode); accumulates into grad dict G."""
    pre, h = cache["pre"], cache["h"]
    qq = q or _id
    dmu, dlv = qq(dmu), qq(dlv)               # the fp32 d_heads are MFMA operands of the heads' dW and dX GEMMs
    dh_mu, G[f"{pre}.fc_mu.weight"], G[f"{pre}.fc_mu.bias"] = _linear_bwd(dmu, h, qq(P[f"{pre}.fc_mu.weight"]))
    dh_lv, G[f"{pre}.fc_logvar.weight"], G[f"{pre}.fc_logvar.bias"] = _linear_bwd(dlv, h, qq(P[f"{pre}.fc_logvar.weight"]))
    dh = dh_mu + dh_lv
    for c in reversed(cache["layers"]):
        dy, dg, db = _bn_relu_drop_bwd(dh, c, q)
        base = f"{pre}.fc.{c['bi']}"
        G[base + ".weight"], G[base + ".bias"] = dg, db
        dh, G[f"{pre}.fc.{c['li']}.weight"], G[f"{pre}.fc.{c['li']}.bias"] = _linear_bwd(dy, c["x"], c["W"])
    return dh


def encoder_c_fwd(P, site, pre="encoder_c"):
    """EncoderC: Embedding -> two Linear heads (src/models/encoders.py:57-61)."""
    h = P[f"{pre}.embedding.weight"][site]
    mu = _linear(h, P[f"{pre}.fc_mu.weight"], P[f"{pre}.fc_mu.bias"])
    lv = _linear(h, P[f"{pre}.fc_logvar.weight"], P[f"{pre}.fc_logvar.bias"])
    return mu, lv, dict(h=h, site=site, pre=pre)


def encoder_c_bwd(P, cache, dmu, dlv, G):
    pre, h = cache["pre"], cache["h"]
    dh_mu, G[f"{pre}.fc_mu.weight"], G[f"{pre}.fc_mu.bias"] = _linear_bwd(dmu, h, P[f"{pre}.fc_mu.weight"])
    dh_lv, G[f"{pre}.fc_logvar.weight"], G[f"{pre}.fc_logvar.bias"] = _linear_bwd(dlv, h, P[f"{pre}.fc_logvar.weight"])
    dE = np.zeros_like(P[f"{pre}.embedding.weight"])
    np.add.at(dE, cache["site"], dh_mu + dh_lv)     # embedding_dense_backward scatter-add
    G[f"{pre}.embedding.weight"] = dE


def decoder_fwd(P, pre, idxs, z, final_sigmoid, q=None):
    """DecoderA/B/C: Linear(+ReLU) chain (src/models/decoders.py:12-16, 26-33, 43-47).

    idxs: Linear positions inside nn.Sequential, e.g. [0,2] or [0,2,4]."""
    q = q or _id
    z = q(z)
    acts = [z]
    h = z
    for j, li in enumerate(idxs):
        y = _linear(h, q(P[f"{pre}.fc.{li}.weight"]), P[f"{pre}.fc.{li}.bias"])
        if j < len(idxs) - 1:
            h = q(np.maximum(y, 0.0))         # hidden activations are stored in the activation type
        elif final_sigmoid:
            h = 1.0 / (1.0 + np.exp(-y))
        else:
            h = y
        acts.append(h)
    return h, dict(acts=acts, pre=pre, idxs=idxs, sig=final_sigmoid)


def decoder_bwd(P, cache, dout, G, q=None, dout_is_logit_grad=False):
    """dout_is_logit_grad: for a sigmoid decoder, `dout` is already w.r.t. the pre-sigmoid logits (the device's fused
    loss hand-off, vae-los-angeles_amd/mmvae/functional.py)."""
    q = q or _id
    pre, idxs, acts = cache["pre"], cache["idxs"], cache["acts"]
    d = dout
    for j in reversed(range(len(idxs))):
        li = idxs[j]
        out = acts[j + 1]
        if j < len(idxs) - 1:
            d = d * (out > 0)
        elif cache["sig"] and not dout_is_logit_grad:
            d = d * out * (1.0 - out)
        d = q(d)                              # stored / loaded as an MFMA operand
        d, G[f"{pre}.fc.{li}.weight"], G[f"{pre}.fc.{li}.bias"] = _linear_bwd(d, acts[j], q(P[f"{pre}.fc.{li}.weight"]))
    return d


# --------------------------------------------------------------------------- #
# MultiModalVAE forward / backward                                             #
# --------------------------------------------------------------------------- #
ENC_A_IDX = [(0, 1, 3)]
ENC_B_IDX = [(0, 1, 3), (4, 5, 7)]


def reparameterize(mu, logvar, eps):
    """src/models/vae.py:11-15 with eps injected."""
    return mu + eps * np.exp(0.5 * logvar)


def vae_forward(P, Bf, a=None, b=None, site=None, masks=None, eps=None, train=True,
                update_running=True, q=None):
    """MultiModalVAE.forward (src/models/vae.py:37-79).  Returns (out_a,out_b,out_c,mu,logvar,cache)."""
    mus, lvs, cache = [], [], {}
    if a is not None:
        m, l, cache["enc_a"] = encoder_mlp_fwd(P, Bf, "encoder_a", ENC_A_IDX, a, masks, train, update_running, q)
        mus.append(m); lvs.append(l)
    if b is not None:
        m, l, cache["enc_b"] = encoder_mlp_fwd(P, Bf, "encoder_b", ENC_B_IDX, b, masks, train, update_running, q)
        mus.append(m); lvs.append(l)
    if site is not None:
        m, l, cache["enc_c"] = encoder_c_fwd(P, site)
        mus.append(m); lvs.append(l)
    if not mus:
        return None, None, None, None, None, None          # vae.py:65-66
    n = len(mus)
    mu = mus[0] if n == 1 else np.stack(mus).mean(0)        # vae.py:67-71
    lv = lvs[0] if n == 1 else np.stack(lvs).mean(0)
    z = reparameterize(mu, lv, eps)
    out_a, cache["dec_a"] = decoder_fwd(P, "decoder_a", [0, 2], z, False, q)
    out_b, cache["dec_b"] = decoder_fwd(P, "decoder_b", [0, 2, 4], z, True, q)
    out_c, cache["dec_c"] = decoder_fwd(P, "decoder_c", [0, 2], z, False, q)
    cache.update(n=n, eps=eps, lv=lv)
    return out_a, out_b, out_c, mu, lv, cache


def vae_backward(P, cache, d_out_a, d_out_b, d_out_c, d_mu, d_lv, q=None, b_is_logit_grad=False):
    """Autograd of vae_forward (reference caller: optimize_hyperparameters.py:112)."""
    G = {}
    dz = decoder_bwd(P, cache["dec_a"], d_out_a, G, q)
    dz = dz + decoder_bwd(P, cache["dec_b"], d_out_b, G, q, b_is_logit_grad)
    dz = dz + decoder_bwd(P, cache["dec_c"], d_out_c, G, q)
    std = np.exp(0.5 * cache["lv"])
    dmu = d_mu + dz
    dlv = d_lv + dz * cache["eps"] * std * 0.5
    n = cache["n"]
    dmu_m, dlv_m = dmu / n, dlv / n
    if "enc_a" in cache:
        encoder_mlp_bwd(P, cache["enc_a"], dmu_m, dlv_m, G, q)
    if "enc_b" in cache:
        encoder_mlp_bwd(P, cache["enc_b"], dmu_m, dlv_m, G, q)
    if "enc_c" in cache:
        encoder_c_bwd(P, cache["enc_c"], dmu_m, dlv_m, G)
    return G


# --------------------------------------------------------------------------- #
# losses                                                                       #
# --------------------------------------------------------------------------- #
def _kld(mu, lv):
    """src/utils/losses.py:42."""
    return -0.5 * np.sum(1.0 + lv - mu ** 2 - np.exp(lv))


def _bce_sum(p, t):
    """F.binary_cross_entropy(reduction='sum') with torch's log clamp at -100."""
    lp = np.maximum(np.log(p), BCE_LOG_CLAMP)
    l1p = np.maximum(np.log1p(-p), BCE_LOG_CLAMP)
    return -np.sum(t * lp + (1.0 - t) * l1p)


def _bce_grad(p, t):
    """d/dp of the clamped BCE sum: torch's binary_cross_entropy_backward,
    (p - t) / max((1-p) p, 1e-12)."""
    return (p - t) / np.maximum((1.0 - p) * p, 1e-12)


def _ce_sum(logits, site, w):
    """F.cross_entropy(weight=w, reduction='sum'): sum_i w[y_i] * nll_i (losses.py:39)."""
    m = logits.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logits - m).sum(1))
    nll = lse - logits[np.arange(len(site)), site]
    wi = np.ones_like(nll) if w is None else w[site]
    sm = np.exp(logits - lse[:, None])
    g = sm.copy()
    g[np.arange(len(site)), site] -= 1.0
    return np.sum(wi * nll), g * wi[:, None]


def _bce_logit_grad(p, t):
    """d/d(logit) of the clamped BCE sum for p = sigmoid(logit): torch's (p - t) / max(p (1-p), 1e-12) times p (1-p)."""
    pq = (1.0 - p) * p
    return np.where(pq >= 1e-12, p - t, (p - t) * pq * 1e12)


def vae_loss(recon_a, a, recon_b, b, recon_c, site, mu, logvar, beta=1e-3, gamma=1.0,
             class_weights=None, q=None):
    """src/utils/losses.py:8-46.  Returns (total, recon, class, kld, grads) where grads
    holds d total / d {recon_a, recon_b, recon_c, mu, logvar} (+ `recon_b_logit`, the gradient w.r.t. DecoderB's
    pre-sigmoid logits; with q the two reconstruction gradients are rounded as the device stores them)."""
    q = q or _id
    with np.errstate(divide="ignore"):
        mse = np.sum((recon_a - a) ** 2)
        bce = _bce_sum(recon_b, b)
    recon = mse + bce
    cls, g_c = _ce_sum(recon_c, site, class_weights)
    kld = _kld(mu, logvar)
    total = recon + gamma * cls + beta * kld
    grads = dict(
        recon_a=q(2.0 * (recon_a - a)),
        recon_b=_bce_grad(recon_b, b),
        recon_b_logit=q(_bce_logit_grad(recon_b, b)),
        recon_c=gamma * g_c,
        mu=beta * mu,
        logvar=beta * (-0.5) * (1.0 - np.exp(logvar)),
    )
    return total, recon, cls, kld, grads


def rna2dna_loss(recon_dna, dna, mu, logvar, beta=1e-3):
    """src/utils/directional_losses.py:8-30."""
    rec = _bce_sum(recon_dna, dna)
    kld = _kld(mu, logvar)
    grads = dict(recon=_bce_grad(recon_dna, dna), mu=beta * mu,
                 logvar=beta * (-0.5) * (1.0 - np.exp(logvar)))
    return rec + beta * kld, rec, kld, grads


def dna2rna_loss(recon_rna, rna, mu, logvar, beta=1e-3):
    """src/utils/directional_losses.py:33-55."""
    rec = np.sum((recon_rna - rna) ** 2)
    kld = _kld(mu, logvar)
    grads = dict(recon=2.0 * (recon_rna - rna), mu=beta * mu,
                 logvar=beta * (-0.5) * (1.0 - np.exp(logvar)))
    return rec + beta * kld, rec, kld, grads


# --------------------------------------------------------------------------- #
# directional VAEs (src/models/directional_vae.py)                             #
# --------------------------------------------------------------------------- #
def directional_param_names(kind, A, D, S, L, E=32):
    """state_dict of RNA2DNAVAE (kind='rna2dna', directional_vae.py:19-23) or
    DNA2RNAVAE (kind='dna2rna', :70-74), expressed through the MultiModalVAE names."""
    if kind == "rna2dna":
        ren = {"encoder_a": "encoder_rna", "encoder_c": "encoder_site", "decoder_b": "decoder_dna"}
    else:
        ren = {"encoder_b": "encoder_dna", "encoder_c": "encoder_site", "decoder_a": "decoder_rna"}
    return ren


def directional_forward(kind, P, Bf, x=None, site=None, masks=None, eps=None, train=True,
                        update_running=True, q=None):
    """RNA2DNAVAE.forward (directional_vae.py:25-60) / DNA2RNAVAE.forward (:76-111).
    P uses the MultiModalVAE key names of the sub-modules involved."""
    mus, lvs, cache = [], [], {"kind": kind}
    if x is not None:
        if kind == "rna2dna":
            m, l, cache["enc_x"] = encoder_mlp_fwd(P, Bf, "encoder_a", ENC_A_IDX, x, masks, train, update_running, q)
        else:
            m, l, cache["enc_x"] = encoder_mlp_fwd(P, Bf, "encoder_b", ENC_B_IDX, x, masks, train, update_running, q)
        mus.append(m); lvs.append(l)
    if site is not None:
        m, l, cache["enc_c"] = encoder_c_fwd(P, site)
        mus.append(m); lvs.append(l)
    if not mus:
        return None, None, None, None
    n = len(mus)
    mu = mus[0] if n == 1 else np.stack(mus).mean(0)
    lv = lvs[0] if n == 1 else np.stack(lvs).mean(0)
    z = reparameterize(mu, lv, eps)
    if kind == "rna2dna":
        out, cache["dec"] = decoder_fwd(P, "decoder_b", [0, 2, 4], z, True, q)
    else:
        out, cache["dec"] = decoder_fwd(P, "decoder_a", [0, 2], z, False, q)
    cache.update(n=n, eps=eps, lv=lv)
    return out, mu, lv, cache


def directional_backward(P, cache, d_out, d_mu, d_lv, q=None, out_is_logit_grad=False):
    G = {}
    dz = decoder_bwd(P, cache["dec"], d_out, G, q, out_is_logit_grad)
    std = np.exp(0.5 * cache["lv"])
    dmu = (d_mu + dz) / cache["n"]
    dlv = (d_lv + dz * cache["eps"] * std * 0.5) / cache["n"]
    if "enc_x" in cache:
        encoder_mlp_bwd(P, cache["enc_x"], dmu, dlv, G, q)
    if "enc_c" in cache:
        encoder_c_bwd(P, cache["enc_c"], dmu, dlv, G)
    return G


# --------------------------------------------------------------------------- #
# AdamW (torch.optim.AdamW semantics; caller: optimize_hyperparameters.py:93-97) #
# --------------------------------------------------------------------------- #
def adamw_init(P):
    return {k: dict(m=np.zeros_like(v), v=np.zeros_like(v)) for k, v in P.items()}, 0


def adamw_step(P, G, state, step, lr=5e-4, wd=1e-5, b1=0.9, b2=0.999, eps=1e-8):
    """Decoupled weight decay, bias-corrected moments (torch.optim.AdamW, non-amsgrad):
       p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
       p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    step += 1
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for k in P:
        if k not in G:
            continue
        g = G[k].astype(P[k].dtype)
        st = state[k]
        P[k] = P[k] * (1.0 - lr * wd)
        st["m"] = b1 * st["m"] + (1.0 - b1) * g
        st["v"] = b2 * st["v"] + (1.0 - b2) * g * g
        denom = np.sqrt(st["v"]) / np.sqrt(bc2) + eps
        P[k] = P[k] - (lr / bc1) * st["m"] / denom
    return step


def train_step(P, Bf, state, step, a, b, site, masks, eps, beta=1e-3, gamma=1.0,
               class_weights=None, lr=5e-4, wd=1e-5, q=None):
    """One full reference-shaped step: forward -> vae_loss -> backward -> AdamW
    (optimize_hyperparameters.py:104-113).  Mutates P, Bf, state; returns dict."""
    out_a, out_b, out_c, mu, lv, cache = vae_forward(P, Bf, a, b, site, masks, eps, True, q=q)
    total, rec, cls, kld, g = vae_loss(out_a, a, out_b, b, out_c, site, mu, lv, beta, gamma, class_weights, q=q)
    if q is None:
        G = vae_backward(P, cache, g["recon_a"], g["recon_b"], g["recon_c"], g["mu"], g["logvar"])
    else:
        G = vae_backward(P, cache, g["recon_a"], g["recon_b_logit"], g["recon_c"], g["mu"], g["logvar"], q, True)
    step = adamw_step(P, G, state, step, lr, wd)
    return dict(out_a=out_a, out_b=out_b, out_c=out_c, mu=mu, logvar=lv, total=total, recon=rec,
                cls=cls, kld=kld, grads=G, step=step)


def cast_tree(d, dtype):
    return {k: (v.astype(dtype) if np.issubdtype(np.asarray(v).dtype, np.floating) else v)
            for k, v in d.items()}
