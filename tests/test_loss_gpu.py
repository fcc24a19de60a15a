"""GPU parity of the fused loss kernel (mmvae_vae_loss, through the C ABI) on the reference's own edge cases:

  * tests/golden/loss_edges.npz -- generated from the imported reference (oracle/make_fixtures.py::case_loss_edges):
    saturated DecoderB outputs p in {0, 1} against targets {0, 1} (BCE log clamp at -100, losses.py:34; backward divides
    by max(p (1-p), 1e-12)), with and without class weights (losses.py:39), beta / gamma different from their defaults.
    Values AND all five gradients, in both gradient modes of the kernel (w.r.t. the DecoderB output p, as autograd would
    deliver it, and w.r.t. the pre-sigmoid logit, as the fused hand-off to the decoder backward uses it) and for both
    gradient storage types (fp32, bf16).
  * a default-width batch (782 / 572 columns: the vectorised streaming loops) with saturated entries sprinkled in,
    against oracle/np_oracle.py.
  * argument validation of the loss entry (device / shape / length mismatches raise before any launch).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import np_oracle as O  # noqa: E402
from golden_util import load  # noqa: E402
from mmvae import ops  # noqa: E402
from src.utils import vae_loss  # noqa: E402

DEV = "cuda"


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _kernel_call(ra, a, rb, b, rc, site, mu, lv, beta, gamma, cw, logit, gdt):
    """Direct C-ABI call: -> (out4 floats, dict of gradients as float64 numpy)."""
    B = ra.shape[0]
    pad8 = lambda n: (n + 7) // 8 * 8
    ga = torch.full((B, pad8(ra.shape[1])), 7.0, dtype=gdt, device=DEV)
    gb = torch.full((B, pad8(rb.shape[1])), 7.0, dtype=gdt, device=DEV)
    gc = torch.empty_like(rc)
    gm, gl = torch.empty_like(mu), torch.empty_like(lv)
    sums = torch.zeros(5, dtype=torch.float64, device=DEV)
    out4 = torch.empty(5, dtype=torch.float32, device=DEV)
    ops.vae_loss(B, recon_a=ra, a=a, recon_b=rb, b=b, logits=rc, site=site, class_weights=cw, mu=mu, logvar=lv, beta=beta,
                 gamma=gamma, sums=sums, g_a=ga, g_b=gb, grad_b_wrt_logit=logit, g_c=gc, g_mu=gm, g_lv=gl)
    ops.loss_finalize(sums, beta, gamma, out4)
    A, D = ra.shape[1], rb.shape[1]
    assert float(ga[:, A:].abs().max() if ga.shape[1] > A else 0) == 0.0      # pad columns of the GEMM operand are zeroed
    assert float(gb[:, D:].abs().max() if gb.shape[1] > D else 0) == 0.0
    g = dict(recon_a=ga[:, :A], recon_b=gb[:, :D], recon_c=gc, mu=gm, logvar=gl)
    vals = out4.tolist()
    assert vals[4] == 0.0                       # no label outside [0, S)
    return vals[:4], {k: v.double().cpu().numpy() for k, v in g.items()}


@pytest.mark.parametrize("tag", ["now", "w"])
def test_loss_edges_fixture_on_the_kernel(tag):
    fx = load("loss_edges")
    ra, a, rb, b, rc, mu, lv = (t(fx[k]) for k in ("recon_a", "a", "recon_b", "b", "recon_c", "mu", "logvar"))
    site = t(fx["site"])
    cw = t(fx["w"]) if tag == "w" else None
    want = fx[tag + ".loss"]
    p = fx["recon_b"].astype(np.float64)
    for logit in (False, True):
        for gdt in (torch.float32, torch.bfloat16):
            vals, g = _kernel_call(ra, a, rb, b, rc, site, mu, lv, 0.25, 0.7, cw, logit, gdt)
            np.testing.assert_allclose(vals, want, rtol=2e-6)
            rt, at = (1e-5, 1e-6) if gdt == torch.float32 else (4e-3, 1e-6)         # bf16 storage: 2^-9 relative
            for nm in ("recon_a", "recon_c", "mu", "logvar"):
                np.testing.assert_allclose(g[nm], fx[f"{tag}.grad.{nm}"], rtol=rt if nm == "recon_a" else 1e-5, atol=at, err_msg=nm)
            gp = fx[f"{tag}.grad.recon_b"].astype(np.float64)                       # reference: d/dp, 1e12-scale at p in {0, 1}
            if not logit:
                np.testing.assert_allclose(g["recon_b"], gp, rtol=rt, atol=at)
                assert np.abs(gp).max() >= 1e11                                      # the clamped denominator was exercised
            else:
                np.testing.assert_allclose(g["recon_b"], gp * p * (1.0 - p), rtol=rt, atol=at)


@pytest.mark.parametrize("tag", ["now", "w"])
def test_loss_edges_fixture_through_vae_loss(tag):
    """Same fixture through the drop-in `vae_loss` + autograd (the general hand-off: plain leaf tensors)."""
    fx = load("loss_edges")
    names = ("recon_a", "recon_b", "recon_c", "mu", "logvar")
    ts = {k: t(fx[k]).requires_grad_(True) for k in names}
    cw = t(fx["w"]) if tag == "w" else None
    loss, rec, cls, kld = vae_loss(ts["recon_a"], t(fx["a"]), ts["recon_b"], t(fx["b"]), ts["recon_c"], t(fx["site"]),
                                   ts["mu"], ts["logvar"], beta=0.25, gamma=0.7, class_weights=cw)
    loss.backward()
    np.testing.assert_allclose([loss.item(), rec, cls, kld], fx[tag + ".loss"], rtol=2e-6)
    for nm in names:
        np.testing.assert_allclose(ts[nm].grad.cpu().numpy(), fx[f"{tag}.grad.{nm}"], rtol=1e-5, atol=1e-6, err_msg=nm)


@pytest.mark.parametrize("B", [257, 4096])
def test_loss_default_widths_with_saturation_vs_oracle(B):
    A, D, S, L = 782, 572, 24, 20
    rng = np.random.default_rng(B)
    ra = rng.standard_normal((B, A)).astype(np.float32); a = np.abs(rng.standard_normal((B, A))).astype(np.float32)
    rb = rng.uniform(1e-4, 1 - 1e-4, (B, D)).astype(np.float32); b = rng.uniform(0, 1, (B, D)).astype(np.float32)
    sat = rng.uniform(size=(B, D))
    rb[sat < 0.01] = 0.0; rb[sat > 0.99] = 1.0                       # sigmoid saturates in fp32 for |logit| > ~17
    rb[0, :4] = [0.0, 1.0, 0.0, 1.0]; b[0, :4] = [1.0, 0.0, 0.0, 1.0]
    rc = (3 * rng.standard_normal((B, S))).astype(np.float32); site = rng.integers(0, S, B)
    mu = rng.standard_normal((B, L)).astype(np.float32); lv = rng.standard_normal((B, L)).astype(np.float32)
    cw = rng.uniform(0.3, 3.0, S).astype(np.float32)
    f = np.float64
    tot, rec, cls, kld, g = O.vae_loss(ra.astype(f), a.astype(f), rb.astype(f), b.astype(f), rc.astype(f), site, mu.astype(f),
                                       lv.astype(f), 0.3, 1.7, cw.astype(f))
    for logit in (False, True):
        vals, gg = _kernel_call(t(ra), t(a), t(rb), t(b), t(rc), t(site), t(mu), t(lv), 0.3, 1.7, t(cw), logit, torch.float32)
        np.testing.assert_allclose(vals, [tot, rec, cls, kld], rtol=5e-6)
        np.testing.assert_allclose(gg["recon_a"], g["recon_a"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(gg["recon_b"], g["recon_b_logit" if logit else "recon_b"], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(gg["recon_c"], g["recon_c"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(gg["mu"], g["mu"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(gg["logvar"], g["logvar"], rtol=1e-5, atol=1e-7)
    # bf16 gradient storage: exactly the bf16-aware oracle's rounding of the same values
    _, g16 = _kernel_call(t(ra), t(a), t(rb), t(b), t(rc), t(site), t(mu), t(lv), 0.3, 1.7, t(cw), True, torch.bfloat16)
    qa, qb = O.bf16_round(g["recon_a"]), O.bf16_round(g["recon_b_logit"])
    assert np.mean(g16["recon_a"] != qa) <= 1e-3 and np.abs(g16["recon_a"] - qa).max() <= 2 ** -7 * np.abs(qa).max()
    assert np.mean(g16["recon_b"] != qb) <= 1e-3 and np.abs(g16["recon_b"] - qb).max() <= 2 ** -7


def test_loss_rejects_mismatched_arguments():
    """torch raises for device / shape mismatches; the HIP path must not launch with them either."""
    A, D, S, L, B = 16, 12, 5, 4, 8
    g = torch.Generator().manual_seed(0)
    ra, rb = torch.randn(B, A, generator=g).to(DEV), torch.rand(B, D, generator=g).to(DEV)
    rc, mu, lv = torch.randn(B, S, generator=g).to(DEV), torch.randn(B, L, generator=g).to(DEV), torch.randn(B, L, generator=g).to(DEV)
    a, b, site = torch.randn(B, A, generator=g).to(DEV), torch.rand(B, D, generator=g).to(DEV), torch.randint(0, S, (B,), generator=g).to(DEV)
    vae_loss(ra, a, rb, b, rc, site, mu, lv)                                   # baseline: fine
    with pytest.raises(RuntimeError, match="device"):
        vae_loss(ra, a.cpu(), rb, b, rc, site, mu, lv)
    with pytest.raises(RuntimeError, match="shape"):
        vae_loss(ra, a[:, :-1], rb, b, rc, site, mu, lv)
    with pytest.raises(RuntimeError, match="shape"):
        vae_loss(ra, a, rb, b[:-1], rc, site, mu, lv)
    with pytest.raises(RuntimeError, match="site"):
        vae_loss(ra, a, rb, b, rc, site[:-1], mu, lv)
    with pytest.raises(RuntimeError, match="class_weights"):
        vae_loss(ra, a, rb, b, rc, site, mu, lv, class_weights=torch.ones(S - 1, device=DEV))
    with pytest.raises(RuntimeError, match="shape"):
        vae_loss(ra, a, rb, b, rc, site, mu, lv[:, :-1])
    bad = site.clone(); bad[3] = S
    with pytest.raises(RuntimeError, match="class index"):
        vae_loss(ra, a, rb, b, rc, bad, mu, lv)
    bad[3] = -1
    with pytest.raises(RuntimeError, match="class index"):
        vae_loss(ra, a, rb, b, rc, bad, mu, lv)
