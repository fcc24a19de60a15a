"""GPU parity of the drop-in modules (through the C ABI) against
  (1) the golden fixtures generated from the imported reference, and
  (2) the numpy oracle (fp64) on seeded inputs at sizes the oracle finishes in seconds.

Stated tolerances (max abs error relative to the tensor's max abs, "scaled error"):
  fp32 mode  (f32 MFMA, fp32 activations):  outputs 5e-5, losses 2e-5 rel, gradients 1e-3 Frobenius-relative
             (5e-3 scaled max: at B=4096 a single ReLU flip against the fp64 oracle shows up as one outlier)
  bf16 mode  (bf16 MFMA operands + bf16 stored activations, f32 accumulate):
             mu/logvar/outputs 3e-2, losses 3e-3 rel; gradients: Frobenius-relative error <= 0.12
             and scaled max error <= 0.25 per tensor.  The max-norm figure is dominated by ReLU
             mask flips: a pre-BatchNorm activation stored in bf16 whose normalised value lies
             within one bf16 ulp of 0 can land on the other side of the ReLU than in the fp64
             oracle, which switches that sample's whole gradient contribution on or off
             (reproduced on the CPU by rounding ONLY that tensor to bf16: 0.12 scaled / 0.037
             Frobenius on encoder_a.fc.0.weight).  It is zero-mean noise, not a bias.
That deviation from the fp64 reference arithmetic is therefore only REPORTED and loosely bounded; what pins the bf16 kernels
is the bf16-aware oracle (np_oracle with q=BF16: the engine's rounding points restated on the CPU): against it bf16
gradients hold <= 1e-2 Frobenius-relative / <= 2e-2 scaled max per tensor (TOL_Q), as SURVEY 8(d) asks.
The measured values are written to gpurun_out/parity_report.txt and quoted in DESIGN.md.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import np_oracle as O  # noqa: E402
from golden_util import load, expect  # noqa: E402
from model_util import load_state, masks_list, scaled_err, named_grads, f64, CHAOTIC_BIASES  # noqa: E402
from mmvae import engine  # noqa: E402
from mmvae.optim import FusedAdamW  # noqa: E402
from src.models import MultiModalVAE, RNA2DNAVAE, DNA2RNAVAE, EncoderA, EncoderB, EncoderC, DecoderA, DecoderB, DecoderC  # noqa: E402
from src.utils import vae_loss  # noqa: E402
from src.utils.directional_losses import rna2dna_loss, dna2rna_loss  # noqa: E402

DEV = "cuda"
TOL = {"fp32": dict(out=5e-5, loss=2e-5, grad=5e-3, fro=1e-3), "bf16": dict(out=3e-2, loss=3e-3, grad=0.25, fro=0.12)}
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.txt")


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")


def t(x):
    return torch.from_numpy(np.asarray(x)).to(DEV)


# ----------------------------------------------------------------------------------------------
# (1) golden fixtures: full training steps, fp32 mode, tight tolerances
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mm_tiny_b16", "mm_default_b32", "mm_default_b77_w"])
def test_train_steps_vs_golden_fp32(name):
    fx = load(name)
    A, D, S, L, E = [int(x) for x in fx["dims"]]
    B, seed, n_steps = int(fx["B"]), int(fx["seed"]), int(fx["n_steps"])
    beta, gamma = float(fx["beta"]), float(fx["gamma"])
    cw = t(fx["class_weights"]) if "class_weights" in fx.files else None
    P, Bf = O.make_params(seed, A, D, S, L, E)
    model = load_state(MultiModalVAE(A, D, S, L, embed_dim=E), P, Bf).to(DEV).set_precision("fp32")
    opt = FusedAdamW(model.parameters(), lr=float(fx["lr"]), weight_decay=float(fx["wd"]))
    a, b, site = t(fx["a"]), t(fx["b"]), t(fx["site"])
    model.train()
    for s in range(n_steps):
        masks, eps = O.make_noise(seed + 100 + s, B, L)
        engine.GLOBAL_NOISE.inject(masks_list(masks), torch.from_numpy(eps))
        ra, rb, rc, mu, lv = model(a=a, b=b, site=site)
        loss, rec, cls, kld = vae_loss(ra, a, rb, b, rc, site, mu, lv, beta=beta, gamma=gamma, class_weights=cw)
        engine.GLOBAL_NOISE.clear()
        assert isinstance(rec, float) and isinstance(cls, float) and isinstance(kld, float)
        opt.zero_grad()
        loss.backward()
        pre = f"s{s}."
        if s == 0:
            for k, v in (("out_a", ra), ("out_b", rb), ("out_c", rc), ("mu", mu), ("logvar", lv)):
                assert v.dtype == torch.float32
                expect(fx, pre + k, v.detach().cpu().numpy(), 3e-4, 3e-5)
        np.testing.assert_allclose([loss.item(), rec, cls, kld], fx[pre + "loss"], rtol=2e-5)
        if s in (0, n_steps - 1):
            for k, g in named_grads(model).items():
                expect(fx, pre + "grad." + k, g, 2e-3, 2e-4, scale_atol=3e-4)
        opt.step()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        if k in CHAOTIC_BIASES:
            continue
        if k.endswith("num_batches_tracked"):
            assert int(v) == n_steps
            continue
        expect(fx, "final." + k, v, 1e-4, 3e-4 if k.endswith("running_mean") else 2e-6,
               outlier_frac=2e-3, outlier_atol=1.1 * 5e-4 * n_steps)
    # eval-mode and single-modality forwards (downstream_task.py:32,48), with the chaotic tensors adopted
    with torch.no_grad():
        st = model.state_dict()
        for k in list(CHAOTIC_BIASES) + [k for k in st if k.endswith("running_mean")]:
            st[k].copy_(t(fx["final." + k]))
    model.eval()
    _, eps = O.make_noise(seed + 900, B, L)
    with torch.no_grad():
        for tag, kw in (("", dict(a=a, b=b, site=site)), ("only_a.", dict(a=a)), ("only_b.", dict(b=b)),
                        ("only_site.", dict(site=site)), ("a_site.", dict(a=a, site=site))):
            engine.GLOBAL_NOISE.inject([], torch.from_numpy(eps))
            outs = model(**kw)
            engine.GLOBAL_NOISE.clear()
            for nm, v in zip(["out_a", "out_b", "out_c", "mu", "logvar"], outs):
                expect(fx, f"eval.{tag}{nm}", v.cpu().numpy(), 5e-4, 1e-4)


@pytest.mark.parametrize("kind", ["rna2dna", "dna2rna"])
def test_directional_vs_golden_fp32(kind):
    fx = load(kind + "_b32")
    A, D, S, L, E = [int(x) for x in fx["dims"]]
    B, seed, beta = int(fx["B"]), int(fx["seed"]), float(fx["beta"])
    P, Bf = O.make_params(seed, A, D, S, L, E)
    ren = O.directional_param_names(kind, A, D, S, L, E)
    cls = RNA2DNAVAE if kind == "rna2dna" else DNA2RNAVAE
    model = load_state(cls(A, D, S, L, embed_dim=E), P, Bf, ren).to(DEV).set_precision("fp32")
    opt = FusedAdamW(model.parameters(), lr=5e-4, weight_decay=1e-5)
    a, b, site = t(fx["a"]), t(fx["b"]), t(fx["site"])
    masks, eps = O.make_noise(seed + 100, B, L)
    model.train()
    if kind == "rna2dna":
        engine.GLOBAL_NOISE.inject(masks_list(masks, ("encoder_a.fc.3",)), torch.from_numpy(eps))
        rec, mu, lv = model(rna=a, site=site)
        loss, r, k = rna2dna_loss(rec, b, mu, lv, beta=beta)
    else:
        engine.GLOBAL_NOISE.inject(masks_list(masks, ("encoder_b.fc.3", "encoder_b.fc.7")), torch.from_numpy(eps))
        rec, mu, lv = model(dna=b, site=site)
        loss, r, k = dna2rna_loss(rec, a, mu, lv, beta=beta)
    engine.GLOBAL_NOISE.clear()
    opt.zero_grad()
    loss.backward()
    expect(fx, "s0.out", rec.detach().cpu().numpy(), 3e-4, 3e-5)
    expect(fx, "s0.mu", mu.detach().cpu().numpy(), 3e-4, 3e-5)
    np.testing.assert_allclose([loss.item(), r, k], fx["s0.loss"], rtol=2e-5)
    for kname, p in model.named_parameters():
        expect(fx, "s0.grad." + kname, p.grad.cpu().numpy(), 2e-3, 2e-4, scale_atol=3e-4)
    opt.step()
    inv = {v: k for k, v in ren.items()}
    for kname, v in model.state_dict().items():
        top, rest = kname.split(".", 1)
        if inv[top] + "." + rest in CHAOTIC_BIASES or kname.endswith("num_batches_tracked"):
            continue
        expect(fx, "final." + kname, v.cpu().numpy(), 1e-4, 3e-4 if kname.endswith("running_mean") else 2e-6,
               outlier_frac=2e-3, outlier_atol=1.1 * 5e-4)
    # site=None inference (reconstruct_unmatched.py:193) on the HIP path, eval mode, with the chaotic tensors adopted
    with torch.no_grad():
        st = model.state_dict()
        for kname in st:
            top, rest = kname.split(".", 1)
            if inv[top] + "." + rest in CHAOTIC_BIASES or kname.endswith("running_mean"):
                st[kname].copy_(t(fx["final." + kname]))
    model.eval()
    with torch.no_grad():
        engine.GLOBAL_NOISE.inject([], torch.from_numpy(eps))
        rec, mu, lv = model(rna=a) if kind == "rna2dna" else model(dna=b)
        engine.GLOBAL_NOISE.clear()
    expect(fx, "eval.nosite.out", rec.cpu().numpy(), 5e-4, 1e-4)
    expect(fx, "eval.nosite.mu", mu.cpu().numpy(), 5e-4, 1e-4)
    assert model(None, None) == (None, None, None)


# ----------------------------------------------------------------------------------------------
# (2) numpy oracle at larger batches, both precisions, measured error report
# ----------------------------------------------------------------------------------------------
# Tolerances against the bf16-AWARE oracle (np_oracle with q=BF16: the same roundings the engine applies, fp64 in between).
# What is left is fp32-vs-fp64 accumulation: a stored bf16 value lands on the neighbouring bf16 number when the fp32 sum and the
# fp64 sum straddle a rounding boundary (~5e-4 of all elements: |sum error| ~ 1e-6 against a bf16 spacing of 4..8e-3).  One ulp
# of a hidden activation is <= 2.8e-3 of an output's scale (`out`, max-norm); ~0.5 % of those elements sit close enough to zero
# to flip the ReLU behind them, and ONE flip toggles one sample's whole contribution to a gradient row, i.e. ~1/sqrt(B) of that
# row's scale in max-norm (`grad`: 2e-2 as SURVEY 8(d) asks from B = 4096 up, 4/sqrt(B) below) while the Frobenius error stays
# under 1e-2 at every size.
TOL_Q = dict(out=5e-3, loss=1e-4, fro=1e-2, grad=2e-2)


def tol_q(B):
    return dict(TOL_Q, grad=max(TOL_Q["grad"], 4.0 / np.sqrt(B)))


def oracle_step(P64, Bf64, a, b, site, masks, eps, beta, gamma, cw, q):
    """One forward + loss + backward of the numpy oracle (q=None: reference arithmetic; q=O.BF16: bf16-aware)."""
    f = np.float64
    Bf_run = dict(Bf64)
    oa, ob, oc, mu, lv, cache = O.vae_forward(P64, Bf_run, a.astype(f), b.astype(f), site, masks, eps.astype(f), True, q=q)
    tot, rec, cls, kld, g = O.vae_loss(oa, a.astype(f), ob, b.astype(f), oc, site, mu, lv, beta, gamma,
                                       None if cw is None else cw.astype(f), q=q)
    if q is None:
        G = O.vae_backward(P64, cache, g["recon_a"], g["recon_b"], g["recon_c"], g["mu"], g["logvar"])
    else:
        G = O.vae_backward(P64, cache, g["recon_a"], g["recon_b_logit"], g["recon_c"], g["mu"], g["logvar"], q, True)
    return dict(out_a=oa, out_b=ob, out_c=oc, mu=mu, logvar=lv, losses=(tot, rec, cls, kld), G=G, Bf=Bf_run)


def compare_step(model, outs, losses, ref, tol, zero_scale_key="encoder_b.fc.0.weight"):
    """-> dict of measured errors; asserts them against `tol` (keys out, loss, fro, grad) after measuring ALL of them (the
    measured values go to the parity report even when an assertion fires)."""
    errs = {}
    for nm, got in zip(("out_a", "out_b", "out_c", "mu", "logvar"), outs):
        errs[nm] = scaled_err(got.detach().cpu().numpy(), ref[nm])
    lerr = max(abs(g_ - r_) / abs(r_) for g_, r_ in zip(losses, ref["losses"]))
    G = ref["G"]
    gerr, gfro, zero_bias = {}, {}, {}
    for k, gv in named_grads(model).items():
        if k in CHAOTIC_BIASES:          # analytically zero (a bias in front of BatchNorm): rounding noise on both sides,
            # bounded against the scale of the same layer's weight gradient
            zero_bias[k] = float(np.max(np.abs(gv - G[k]))) / float(np.max(np.abs(G[k.replace(".bias", ".weight")])))
            continue
        gerr[k] = scaled_err(gv, G[k])
        gfro[k] = float(np.linalg.norm(gv - G[k]) / np.linalg.norm(G[k]))
    sd = model.state_dict()
    berr = {k: scaled_err(sd[k].cpu().numpy(), v) for k, v in ref["Bf"].items() if k.endswith("running_mean") or k.endswith("running_var")}
    e = dict(out=max(errs.values()), loss=lerr, grad=max(gerr.values()), grad_worst=max(gerr, key=gerr.get),
             fro=max(gfro.values()), fro_worst=max(gfro, key=gfro.get), bn=max(berr.values()))
    report(f"    measured against tol {tol}: {e}")
    for nm, v in errs.items():
        assert v <= tol["out"], (nm, v)
    assert lerr <= tol["loss"], (lerr, losses, ref["losses"])
    for k, v in zero_bias.items():
        assert v <= 3e-2, (k, v)
    for k in gerr:
        assert gerr[k] <= tol["grad"] and gfro[k] <= tol["fro"], (k, gerr[k], gfro[k])
    for k, v in berr.items():
        assert v <= tol["out"], (k, v)
    return e


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("B", [1000, 4096])
def test_step_vs_oracle(prec, B):
    """fp32 mode against the reference arithmetic (fp64 oracle); bf16 mode against the bf16-aware oracle at the TIGHT
    tolerance (TOL_Q) and, as a reported + loosely bounded number, against the fp64 oracle."""
    A, D, S, L, E = 782, 572, 24, 20, 32
    seed = 100 + B
    P, Bf = O.make_params(seed, A, D, S, L, E)
    a, b, site = O.make_batch(seed + 1, B, A, D, S)
    masks, eps = O.make_noise(seed + 2, B, L)
    cw = np.random.default_rng(5).uniform(0.5, 2.0, S).astype(np.float32)
    beta, gamma = 1e-3, 1.0
    P64, Bf64 = f64(P), f64(Bf)

    model = load_state(MultiModalVAE(A, D, S, L, embed_dim=E), P, Bf).to(DEV).set_precision(prec)
    model.train()
    engine.GLOBAL_NOISE.inject(masks_list(masks), torch.from_numpy(eps))
    ra, rb, rc, m_, l_ = model(a=t(a), b=t(b), site=t(site))
    loss, r_, c_, k_ = vae_loss(ra, t(a), rb, t(b), rc, t(site), m_, l_, beta=beta, gamma=gamma, class_weights=t(cw))
    engine.GLOBAL_NOISE.clear()
    loss.backward()
    outs, losses = (ra, rb, rc, m_, l_), (loss.item(), r_, c_, k_)
    ref = oracle_step(P64, Bf64, a, b, site, masks, eps, beta, gamma, cw, None)
    e = compare_step(model, outs, losses, ref, TOL[prec])
    report(f"step_vs_oracle prec={prec} B={B} vs fp64 reference arithmetic: out {e['out']:.3e}; loss rel {e['loss']:.3e}; "
           f"grad max scaled {e['grad']:.3e} ({e['grad_worst']}); grad max Frobenius-rel {e['fro']:.3e} ({e['fro_worst']})")
    if prec == "bf16":
        refq = oracle_step(P64, Bf64, a, b, site, masks, eps, beta, gamma, cw, O.BF16)
        e = compare_step(model, outs, losses, refq, tol_q(B))
        report(f"step_vs_oracle prec=bf16 B={B} vs bf16-aware oracle:        out {e['out']:.3e}; loss rel {e['loss']:.3e}; "
               f"grad max scaled {e['grad']:.3e} ({e['grad_worst']}); grad max Frobenius-rel {e['fro']:.3e} ({e['fro_worst']})")


# ----------------------------------------------------------------------------------------------
# hand-off equivalence, stand-alone blocks, edge cases, ABI of the checkpoint
# ----------------------------------------------------------------------------------------------
def _small_model(prec, seed=3, dims=(50, 36, 5, 6, 8)):
    A, D, S, L, E = dims
    P, Bf = O.make_params(seed, A, D, S, L, E)
    return load_state(MultiModalVAE(A, D, S, L, embed_dim=E), P, Bf).to(DEV).set_precision(prec), P, Bf, dims


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_general_loss_path_matches_fused_handoff(prec):
    """Same forward, gradients delivered (a) by the fused loss stash and (b) through plain autograd
    from torch's own loss functions: identical math, so fp32 agrees to rounding."""
    import torch.nn.functional as F
    model, P, Bf, (A, D, S, L, E) = _small_model(prec)
    B = 200
    a, b, site = O.make_batch(9, B, A, D, S)
    masks, eps = O.make_noise(10, B, L)
    a_, b_, s_ = t(a), t(b), t(site)
    grads = []
    for mode in ("fused", "torch"):
        model.zero_grad(set_to_none=True)
        model.train()
        engine.GLOBAL_NOISE.inject(masks_list(masks), torch.from_numpy(eps))
        ra, rb, rc, mu, lv = model(a=a_, b=b_, site=s_)
        engine.GLOBAL_NOISE.clear()
        if mode == "fused":
            loss, _, _, _ = vae_loss(ra, a_, rb, b_, rc, s_, mu, lv, beta=0.5, gamma=0.7)
        else:
            loss = (F.mse_loss(ra, a_, reduction="sum") + F.binary_cross_entropy(rb, b_, reduction="sum")
                    + 0.7 * F.cross_entropy(rc, s_, reduction="sum")
                    + 0.5 * (-0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp())))
        (2.0 * loss).backward()                   # grad_output != 1 exercises mmvae_scale_if_needed
        grads.append(named_grads(model))
    tol = 2e-4 if prec == "fp32" else 0.25
    for k in grads[0]:
        if k in CHAOTIC_BIASES:
            continue
        assert scaled_err(grads[0][k], grads[1][k]) <= tol, k


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_standalone_blocks_vs_oracle(prec):
    A, D, S, L, E = 40, 28, 6, 4, 8
    B = 130
    P, Bf = O.make_params(12, A, D, S, L, E)
    P64, Bf64 = f64(P), f64(Bf)
    a, b, site = O.make_batch(13, B, A, D, S)
    masks, eps = O.make_noise(14, B, L)
    tol = TOL[prec]

    def sub(prefix):
        return {k.split(".", 1)[1]: torch.from_numpy(np.array(v)) for k, v in list(P.items()) + list(Bf.items()) if k.startswith(prefix + ".")}

    for cls_, pre, x, idx, mk in ((EncoderA, "encoder_a", a, O.ENC_A_IDX, ("encoder_a.fc.3",)),
                                   (EncoderB, "encoder_b", b, O.ENC_B_IDX, ("encoder_b.fc.3", "encoder_b.fc.7"))):
        enc = cls_(x.shape[1], L)
        enc.load_state_dict(sub(pre)); enc.to(DEV).set_precision(prec).train()
        engine.GLOBAL_NOISE.inject(masks_list(masks, mk), None)
        mu, lv = enc(t(x))
        engine.GLOBAL_NOISE.clear()
        rmu, rlv, cache = O.encoder_mlp_fwd(P64, dict(Bf64), pre, idx, x.astype(np.float64), masks, True)
        assert scaled_err(mu.detach().cpu().numpy(), rmu) <= tol["out"]
        assert scaled_err(lv.detach().cpu().numpy(), rlv) <= tol["out"]
        w = torch.randn(B, L, generator=torch.Generator().manual_seed(1))
        ((mu * w.to(DEV)).sum() + (lv * 0.5).sum()).backward()
        G = {}
        O.encoder_mlp_bwd(P64, cache, w.double().numpy(), np.full((B, L), 0.5), G)
        for k, p in enc.named_parameters():
            if f"{pre}.{k}" in CHAOTIC_BIASES:
                continue
            gref = G[f"{pre}.{k}"]                                   # small batch: one ReLU flip is visible in max-norm, use Frobenius
            assert float(np.linalg.norm(p.grad.cpu().numpy() - gref) / np.linalg.norm(gref)) <= (tol["fro"] if prec == "fp32" else 0.2), k

    encc = EncoderC(S, L, embed_dim=E)
    encc.load_state_dict(sub("encoder_c")); encc.to(DEV)
    mu, lv = encc(t(site))
    rmu, rlv, cache = O.encoder_c_fwd(P64, site)
    assert scaled_err(mu.detach().cpu().numpy(), rmu) <= 5e-5
    (mu.sum() + (lv * lv).sum()).backward()
    G = {}
    O.encoder_c_bwd(P64, cache, np.ones((B, L)), 2 * rlv, G)
    for k, p in encc.named_parameters():
        assert scaled_err(p.grad.cpu().numpy(), G["encoder_c." + k]) <= 3e-4, k

    z = np.random.default_rng(3).standard_normal((B, L)).astype(np.float32)
    for cls_, pre, idxs, sig, odim in ((DecoderA, "decoder_a", [0, 2], False, A), (DecoderB, "decoder_b", [0, 2, 4], True, D),
                                        (DecoderC, "decoder_c", [0, 2], False, S)):
        dec = cls_(L, odim)
        dec.load_state_dict(sub(pre)); dec.to(DEV).set_precision(prec)
        zt = t(z).requires_grad_(True)
        out = dec(zt)
        zq = z.astype(np.float64) if prec == "fp32" else torch.from_numpy(z).bfloat16().double().numpy()
        ref, cache = O.decoder_fwd(P64, pre, idxs, zq, sig)
        assert scaled_err(out.detach().cpu().numpy(), ref) <= tol["out"]
        out.sum().backward()
        G = {}
        dz = O.decoder_bwd(P64, cache, np.ones_like(ref), G)
        assert scaled_err(zt.grad.cpu().numpy(), dz) <= tol["grad"]
        for k, p in dec.named_parameters():
            assert scaled_err(p.grad.cpu().numpy(), G[f"{pre}.{k}"]) <= tol["grad"], k


def test_edge_cases():
    model, P, Bf, (A, D, S, L, E) = _small_model("fp32")
    assert model() == (None, None, None, None, None)                       # vae.py:65-66
    a, b, site = O.make_batch(1, 1, A, D, S)
    model.train()
    with pytest.raises(ValueError, match="more than 1 value per channel"):  # BatchNorm1d with B == 1
        model(a=t(a), b=t(b), site=t(site))
    model.eval()
    with torch.no_grad():
        outs = model(a=t(a), b=t(b), site=t(site))                          # fine in eval mode
    assert outs[0].shape == (1, A) and outs[3].shape == (1, L)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(a=torch.from_numpy(a))
    # eps is sampled in eval mode too (vae.py:73): recon stochastic, mu deterministic
    a2, b2, s2 = O.make_batch(2, 64, A, D, S)
    with torch.no_grad():
        o1 = model(a=t(a2), b=t(b2), site=t(s2))
        o2 = model(a=t(a2), b=t(b2), site=t(s2))
    assert torch.equal(o1[3], o2[3]) and not torch.equal(o1[0], o2[0])
    # loss with a missing modality behaves like the reference (losses.py:46 on a Python int)
    with pytest.raises(AttributeError):
        vae_loss(o1[0], t(a2), o1[1], t(b2), None, None, o1[3], o1[4])


def test_state_dict_keys_match_reference_abi():
    A, D, S, L, E = 782, 572, 24, 20, 32
    model = MultiModalVAE(A, D, S, L)
    P, Bf = O.make_params(0, A, D, S, L, E)
    assert set(model.state_dict().keys()) == set(P) | set(Bf)
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(np.asarray(P.get(k, Bf.get(k))).shape), k
    assert [k for k, _ in model.named_parameters()] == [k for k, _ in O.param_shapes(A, D, S, L, E)]


def test_philox_noise_statistics():
    m = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    from mmvae import ops
    ops.dropout_mask(m, 0.9, 1234, 0)
    assert set(m.unique().tolist()) <= {0, 1}
    assert abs(m.float().mean().item() - 0.9) < 2e-3
    m2 = torch.empty_like(m)
    ops.dropout_mask(m2, 0.9, 1234, 0)
    assert torch.equal(m, m2)                      # counter-based: same (seed, offset) -> same stream
    ops.dropout_mask(m2, 0.9, 1234, 1 << 18)
    assert not torch.equal(m, m2)
    e = torch.empty(1 << 20, device=DEV)
    ops.randn(e, 7, 0)
    assert abs(e.mean().item()) < 5e-3 and abs(e.std().item() - 1.0) < 5e-3
    assert abs((e ** 4).mean().item() - 3.0) < 0.1
