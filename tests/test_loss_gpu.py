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


@pytest.mark.parametrize("B", [1000, 4096])
@pytest.mark.parametrize("weighted", [False, True])
def test_reconstruction_loss_inside_the_decoder_gemms(B, weighted):
    """Training-step fusion (mmvae.graphs -> engine.VAEGraph.fused_recon): sum-MSE / sum-BCE and their gradients computed in the
    epilogue of the decoders' last GEMMs (EPI_LOSS_MSE / EPI_LOSS_BCE_LOGIT) instead of mmvae_vae_loss on stored reconstructions
    (losses.py:31,34).  Same arithmetic on the same fp32 values: the loss sums differ by their summation order only, the parameter
    gradients by the order of the atomic accumulations (the bf16 gradient buffers themselves are bit-identical, next test)."""
    import torch
    from mmvae import engine, functional as F_
    from src.models import MultiModalVAE
    A, D, S, L = 782, 572, 24, 20
    torch.manual_seed(3)
    model = MultiModalVAE(A, D, S, L).to("cuda").set_precision("bf16").train()
    g = torch.Generator(device="cuda").manual_seed(4)
    a = torch.randn(B, A, device="cuda", generator=g)
    b = torch.rand(B, D, device="cuda", generator=g)
    b[:7, :5] = 0.0; b[7:13, :5] = 1.0                                       # hard targets next to saturating logits
    site = torch.randint(0, S, (B,), device="cuda", generator=g)
    cw = (torch.rand(S, device="cuda", generator=g) + 0.5) if weighted else None
    dev = torch.device("cuda", torch.cuda.current_device())

    def step(fuse, tb=None):
        engine.GLOBAL_NOISE.offset_tensor(dev).zero_()
        gr = model._graph()
        gr.fused_recon = [a, b, None] if fuse else None
        try:
            ra, rb, rc, mu, lv = model(a=a, b=b, site=site)
        finally:
            gr.fused_recon = None
        if fuse:
            assert ra.stride(0) == 0 and rb.stride(0) == 0               # placeholders: the reconstructions were never stored
        total, out5 = F_.fused_loss({"a": (ra, a), "b": (rb, b if tb is None else tb), "c": (rc, site), "kl": (mu, lv)}, 1e-3, 1.0, cw)
        for p in model.parameters():
            p.grad = None
        total.backward()
        return out5.clone(), {k: p.grad.clone() for k, p in model.named_parameters()}

    o0, g0 = step(False)
    o1, g1 = step(True)
    np.testing.assert_allclose(o1.cpu().numpy()[:4], o0.cpu().numpy()[:4], rtol=2e-6)
    for k in g0:            # db sums and the split dW tiles of small layers are accumulated with f32 atomics: order noise only
        assert float((g0[k] - g1[k]).abs().max()) <= 1e-4 * float(g0[k].abs().max()) + 1e-12, k
    with pytest.raises(RuntimeError, match="target"):
        step(True, tb=b.clone())


@pytest.mark.parametrize("M,N,K,bce", [(1000, 782, 128, False), (777, 572, 512, True), (300, 333, 256, True), (4096, 128, 192, False),
                                       # >= 16 384 rows, >= 8 K steps, 16-byte target rows: the software-pipelined kernel (gemm_nt2x.h) -- ragged last row
                                       # tile (16 640 = 65 x 256), ragged last column tile (572, 332), 8 / 10 / 9 K steps
                                       (16640, 572, 512, True), (16384, 332, 600, False), (32768, 128, 576, True)])
def test_loss_epilogue_against_store_epilogue_plus_loss_kernel(M, N, K, bce):
    """mmvae_gemm_nt with MMVAE_EPI_LOSS_MSE / MMVAE_EPI_LOSS_BCE_LOGIT against the pair it replaces (store epilogue -> fp32
    output -> mmvae_vae_loss): bit-identical bf16 gradient rows incl. zeroed pad columns, loss sum to 1e-7; target rows 16-, 8- and
    4-byte aligned (N = 572, 782, 333), ragged row and column tiles."""
    from mmvae.ops import PREC_BF16
    dev = "cuda"
    torch.manual_seed(M + N)
    A = torch.randn(M, K, device=dev).bfloat16()
    W = torch.randn(N, K, device=dev) / K ** 0.5 * (6.0 if bce else 1.0)           # some logits saturate the sigmoid
    bias = torch.randn(N, device=dev) * 0.1
    pl = ops.PreparedLinear([W], [bias], PREC_BF16, dev); ops.WeightPrep([pl], dev).run()
    T = (torch.rand(M, N, device=dev) > 0.5).float() if bce else torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev)
    Np = ops.ceil_to(N, 8)
    g_ref = torch.empty(M, Np, dtype=torch.bfloat16, device=dev)
    g_new = torch.full((M, Np), 7.0, dtype=torch.bfloat16, device=dev)
    sums, _ = ops.loss_workspace(dev)
    sums2 = torch.zeros(5, dtype=torch.float64, device=dev)
    ops.gemm_nt(PREC_BF16, A, pl.w, N, K, out, bias=pl.bias, act=ops.ACT_SIGMOID if bce else ops.ACT_NONE)
    if bce:
        ops.vae_loss(M, recon_b=out, b=T, sums=sums, g_b=g_ref, grad_b_wrt_logit=True)
    else:
        ops.vae_loss(M, recon_a=out, a=T, sums=sums, g_a=g_ref)
    k = 1 if bce else 0
    ops.gemm_nt(PREC_BF16, A, pl.w, N, K, g_new, bias=pl.bias, epilogue=ops.EPI_LOSS_BCE_LOGIT if bce else ops.EPI_LOSS_MSE, h=T, loss_sum=sums2[k:k + 1])
    assert torch.equal(g_ref.view(torch.int16), g_new.view(torch.int16))
    np.testing.assert_allclose(sums2[k].item(), sums[k].item(), rtol=1e-7)       # same terms; the per-thread fp32 partial sums group them differently
    assert sums2[1 - k].item() == 0.0
    with pytest.raises(RuntimeError):                                             # one K step only: not this kernel's case
        ops.gemm_nt(PREC_BF16, A[:, :64].contiguous(), pl.w, N, 64, g_new, bias=pl.bias, epilogue=ops.EPI_LOSS_MSE, h=T, loss_sum=sums2[0:1])


@pytest.mark.parametrize("S,weighted", [(24, False), (24, True), (22, True), (8, False), (32, True), (40, True)])
def test_class_term_ignore_index(S, weighted):
    """F.cross_entropy's default ignore_index = -100 (losses.py:39 passes none): such a row adds no loss and gets a zero gradient in
    the reference; here it must not count as an out-of-range label either.  Against torch's own cross_entropy on the same logits
    (all three class-term code paths: S % 4 == 0 and <= 32 one row per thread with 16-byte loads, other S <= 32 one row per half
    wave, S > 32 one row per thread, scalar loads)."""
    B = 777
    g = torch.Generator().manual_seed(S)
    logits = torch.randn(B, S, generator=g)
    site = torch.randint(0, S, (B,), generator=g)
    site[::7] = -100
    cw = (torch.rand(S, generator=g) + 0.5) if weighted else None
    lr = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, site, weight=cw, reduction="sum")
    ref.backward()
    ld, sd = logits.to(DEV), site.to(DEV)
    gc = torch.full_like(ld, 7.0)
    sums = torch.zeros(5, dtype=torch.float64, device=DEV)
    ops.vae_loss(B, logits=ld, site=sd, class_weights=None if cw is None else cw.to(DEV), gamma=1.0, sums=sums, g_c=gc)
    assert sums[4].item() == 0.0                                             # -100 is not "outside [0, S)"
    np.testing.assert_allclose(sums[2].item(), ref.item(), rtol=2e-6)
    np.testing.assert_allclose(gc.cpu().numpy(), lr.grad.numpy(), atol=2e-6)
    assert float(gc[::7].abs().max()) == 0.0


@pytest.mark.parametrize("S", [24, 22])
def test_class_term_out_of_range_labels_are_counted(S):
    """A label outside [0, S) (torch device-asserts on it) is counted in sums[4] and treated as class 0, the same way on the
    row-per-thread (S = 24) and the row-per-half-wave (S = 22) forms of the class term."""
    B = 515
    g = torch.Generator().manual_seed(3 * S)
    logits = torch.randn(B, S, generator=g)
    site = torch.randint(0, S, (B,), generator=g)
    bad = torch.arange(5, B, 37)
    site_bad = site.clone(); site_bad[bad] = S + 3
    site_bad[bad[::2]] = -1
    site_ref = site.clone(); site_ref[bad] = 0
    lr = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, site_ref, reduction="sum")
    ref.backward()
    ld = logits.to(DEV)
    gc = torch.full_like(ld, 7.0)
    sums = torch.zeros(5, dtype=torch.float64, device=DEV)
    ops.vae_loss(B, logits=ld, site=site_bad.to(DEV), gamma=0.5, sums=sums, g_c=gc)
    assert sums[4].item() == float(len(bad))
    np.testing.assert_allclose(sums[2].item(), ref.item(), rtol=2e-6)
    np.testing.assert_allclose(gc.cpu().numpy(), 0.5 * lr.grad.numpy(), atol=2e-6)
