"""Optimiser / resume behaviour on the GPU (the reference itself only saves model weights,
train_dna2rna.py:230-231, so the expectation is torch.optim.AdamW's own semantics):

  * FusedAdamW == torch.optim.AdamW step for step when the SET of parameters with a gradient changes between steps
    (model(a=.., site=..) first, all modalities next: the gradient arena keeps its addresses, the set does not);
  * FusedAdamW.load_state_dict() mid-run (moments + step counts replaced) keeps matching torch.optim.AdamW, and state
    dicts travel both ways between the two optimisers;
  * N steps == K steps -> save model + optimiser + scheduler + noise position -> fresh objects -> load -> N-K steps,
    eager and hipGraph-captured;
  * backward through an EVAL-mode forward (BatchNorm on running statistics) against the oracle.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import np_oracle as O  # noqa: E402
from model_util import load_state, masks_list, named_grads, f64, scaled_err, CHAOTIC_BIASES  # noqa: E402
from mmvae import engine, checkpoint  # noqa: E402
from mmvae.optim import FusedAdamW  # noqa: E402
from src.models import MultiModalVAE  # noqa: E402
from src.utils import vae_loss  # noqa: E402

DEV = "cuda"
DIMS = (50, 36, 5, 6, 8)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _model(seed=3, prec="fp32", dims=DIMS):
    A, D, S, L, E = dims
    P, Bf = O.make_params(seed, A, D, S, L, E)
    return load_state(MultiModalVAE(A, D, S, L, embed_dim=E), P, Bf).to(DEV).set_precision(prec).train()


def _step(model, opt, a, b, site, seed, subset=False):
    A, D, S, L, E = DIMS
    masks, eps = O.make_noise(seed, a.shape[0], L)
    if subset:
        engine.GLOBAL_NOISE.inject(masks_list(masks, ("encoder_a.fc.3",)), torch.from_numpy(eps))
        ra, rb, rc, mu, lv = model(a=a, site=site)
    else:
        engine.GLOBAL_NOISE.inject(masks_list(masks), torch.from_numpy(eps))
        ra, rb, rc, mu, lv = model(a=a, b=b, site=site)
    loss, *_ = vae_loss(ra, a, rb, b, rc, site, mu, lv, beta=0.01, gamma=0.5)
    engine.GLOBAL_NOISE.clear()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    return loss


def _mirror_torch_step(model, topt):
    """Apply stock torch.optim.AdamW to a deep copy of (parameters, gradients): the expected update."""
    topt.step()


def test_param_set_change_and_load_state_dict_match_torch_adamw():
    A, D, S, L, E = DIMS
    B = 96
    a, b, site = (t(x) for x in O.make_batch(9, B, A, D, S))
    model = _model()
    ref = copy.deepcopy(model)                      # same values; updated by stock torch AdamW from OUR gradients
    fused = FusedAdamW(model.parameters(), lr=3e-3, weight_decay=1e-2)
    stock = torch.optim.AdamW(ref.parameters(), lr=3e-3, weight_decay=1e-2)
    plan = [True, True, False, False, True, False]          # True: only (a, site) -> encoder_b has no gradient
    for i, subset in enumerate(plan):
        _step(model, fused, a, b, site, 100 + i, subset)
        with torch.no_grad():
            for p, q in zip(model.parameters(), ref.parameters()):
                q.grad = None if p.grad is None else p.grad.clone()
        fused.step(); stock.step()
        for (k, p), q in zip(model.named_parameters(), ref.parameters()):
            assert torch.allclose(p, q, rtol=1e-6, atol=1e-7), (i, k, float((p - q).abs().max()))
        if i == 3:
            # mid-run: round-trip the state through the OTHER optimiser's load_state_dict (moments + steps replaced)
            sd_f, sd_s = copy.deepcopy(fused.state_dict()), copy.deepcopy(stock.state_dict())
            assert sd_f["param_groups"][0]["betas"] == (0.9, 0.999) and set(sd_f) == {"state", "param_groups"}
            fused.load_state_dict(sd_s); stock.load_state_dict(sd_f)
    # encoder_b parameters were stepped 3 times, the others 6: per-parameter step counts, as torch keeps them
    steps = {k: int(fused.state[p]["step"].item()) for k, p in model.named_parameters() if p in fused.state}
    assert steps["encoder_b.fc.0.weight"] == 3 and steps["encoder_a.fc.0.weight"] == 6 and steps["decoder_b.fc.4.weight"] == 6
    sd = fused.state_dict()
    assert all(int(v["step"]) in (3, 6) for v in sd["state"].values())
    assert {int(v["step"]) for v in stock.state_dict()["state"].values()} == {3, 6}


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_resume_equals_straight_run(prec, tmp_path):
    """6 steps straight == 3 steps, checkpoint, fresh model / optimiser / scheduler, load, 3 steps (injected noise)."""
    A, D, S, L, E = DIMS
    B, N, K = 64, 6, 3
    a, b, site = (t(x) for x in O.make_batch(21, B, A, D, S))

    def fresh():
        m = _model(seed=5, prec=prec)
        o = FusedAdamW(m.parameters(), lr=2e-3, weight_decay=1e-5)
        s = torch.optim.lr_scheduler.ReduceLROnPlateau(o, mode="min", factor=0.5, patience=0)
        return m, o, s

    def run(m, o, s, lo, hi, losses):
        for i in range(lo, hi):
            loss = _step(m, o, a, b, site, 500 + i)
            o.step()
            losses.append(loss.item())
            s.step(1.0 if i % 2 else 2.0)                        # plateaus: the LR really changes along the way

    m1, o1, s1 = fresh(); l1 = []
    run(m1, o1, s1, 0, N, l1)
    m2, o2, s2 = fresh(); l2 = []
    run(m2, o2, s2, 0, K, l2)
    path = tmp_path / "state.pt"
    checkpoint.save_training_state(path, m2, o2, s2, epoch=K)
    del m2, o2, s2
    m3, o3, s3 = fresh()
    extra = checkpoint.load_training_state(path, m3, o3, s3)
    assert extra == {"epoch": K}
    run(m3, o3, s3, K, N, l2)
    tol = dict(rtol=1e-6, atol=1e-7) if prec == "fp32" else dict(rtol=2e-3, atol=2e-5)      # atomics order only / bf16 flips
    np.testing.assert_allclose(l2, l1, rtol=1e-6 if prec == "fp32" else 2e-3)
    assert o3.param_groups[0]["lr"] == o1.param_groups[0]["lr"] and o1.param_groups[0]["lr"] < 2e-3
    for (k, p), q in zip(m1.named_parameters(), m3.parameters()):
        if k in CHAOTIC_BIASES:          # zero-gradient biases in front of BatchNorm: Adam turns atomics-order noise into +-lr steps
            continue
        assert torch.allclose(p.detach(), q.detach(), **tol), (k, float((p - q).abs().max()))
    for (k, p), q in zip(m1.named_buffers(), m3.buffers()):
        if k.endswith("running_mean"):   # absorbs the chaotic biases
            continue
        assert torch.allclose(p.float(), q.float(), rtol=1e-5, atol=1e-6), k
    assert int(o3.state[next(iter(m3.parameters()))]["step"].item()) == N


def test_resume_graphed_step_with_device_counters(tmp_path):
    """hipGraph-captured step: Adam step count and Philox offset live on the device.  A checkpoint taken after K replays
    restores both; the resumed (re-captured) run reproduces the straight run's losses."""
    from mmvae.graphs import GraphedTrainStep
    A, D, S, L, B = 782, 572, 24, 20, 512
    g = torch.Generator().manual_seed(5)
    a = torch.randn(B, A, generator=g).abs().to(DEV); b = torch.rand(B, D, generator=g).to(DEV)
    site = torch.randint(0, S, (B,), generator=g).to(DEV)
    dev = torch.device(DEV, torch.cuda.current_device())

    def fresh():
        torch.manual_seed(77)
        m = MultiModalVAE(A, D, S, L).to(DEV).set_precision("fp32").train()
        return m, FusedAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)

    engine.GLOBAL_NOISE.offset_tensor(dev).zero_()
    m1, o1 = fresh()
    g1 = GraphedTrainStep(m1, o1, a, b, site, warmup=2)
    straight = []
    for _ in range(6):
        g1(); straight.append(g1.losses()[0])
    engine.GLOBAL_NOISE.offset_tensor(dev).zero_()
    m2, o2 = fresh()
    g2 = GraphedTrainStep(m2, o2, a, b, site, warmup=2)
    got = []
    for _ in range(3):
        g2(); got.append(g2.losses()[0])
    st = checkpoint.training_state(m2, o2)
    assert st["noise"]["offset"] > 0
    assert int(next(iter(st["optimizer"]["state"].values()))["step"]) == 2 + 3          # warm-up + replays
    path = tmp_path / "g.pt"
    torch.save(st, path)
    del g2, m2, o2
    engine.GLOBAL_NOISE.offset_tensor(dev).zero_()
    m3, o3 = fresh()
    checkpoint.load_training_state(path, m3, o3)
    g3 = GraphedTrainStep(m3, o3, a, b, site, warmup=1, preserve_state=True)     # the warm-up step is undone before the capture
    for _ in range(3):
        g3(); got.append(g3.losses()[0])
    np.testing.assert_allclose(got, straight, rtol=2e-5)
    assert int(o3.state[next(iter(m3.parameters()))]["step"].item()) == 2 + 6


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_backward_through_eval_mode_forward(prec):
    """model.eval() then loss.backward(): BatchNorm uses the running statistics and its backward is dy = gamma*rstd*d
    (torch batch_norm(training=False)); round 1 silently used uninitialised batch statistics here."""
    A, D, S, L, E = 782, 572, 24, 20, 32
    B = 300
    P, Bf = O.make_params(31, A, D, S, L, E)
    rng = np.random.default_rng(1)
    for k in Bf:
        if k.endswith("running_mean"):
            Bf[k] = (0.2 * rng.standard_normal(Bf[k].shape)).astype(np.float32)
        elif k.endswith("running_var"):
            Bf[k] = rng.uniform(0.5, 1.5, Bf[k].shape).astype(np.float32)
    a, b, site = O.make_batch(32, B, A, D, S)
    _, eps = O.make_noise(33, B, L)
    model = load_state(MultiModalVAE(A, D, S, L, embed_dim=E), P, Bf).to(DEV).set_precision(prec).eval()
    engine.GLOBAL_NOISE.inject([], torch.from_numpy(eps))
    ra, rb, rc, mu, lv = model(a=t(a), b=t(b), site=t(site))
    loss, *_ = vae_loss(ra, t(a), rb, t(b), rc, t(site), mu, lv, beta=0.01, gamma=0.5)
    engine.GLOBAL_NOISE.clear()
    loss.backward()
    q = None if prec == "fp32" else O.BF16
    P64, Bf64 = f64(P), f64(Bf)
    f = np.float64
    oa, ob, oc, m, l, cache = O.vae_forward(P64, dict(Bf64), a.astype(f), b.astype(f), site, None, eps.astype(f), False, q=q)
    tot, rec, cls, kld, g = O.vae_loss(oa, a.astype(f), ob, b.astype(f), oc, site, m, l, 0.01, 0.5, q=q)
    if q is None:
        G = O.vae_backward(P64, cache, g["recon_a"], g["recon_b"], g["recon_c"], g["mu"], g["logvar"])
    else:
        G = O.vae_backward(P64, cache, g["recon_a"], g["recon_b_logit"], g["recon_c"], g["mu"], g["logvar"], q, True)
    assert abs(loss.item() - tot) <= (2e-5 if prec == "fp32" else 1e-4) * abs(tot)
    sd = model.state_dict()
    for k in Bf:                                                     # eval mode leaves the running statistics alone
        assert torch.equal(sd[k].cpu(), torch.from_numpy(np.asarray(Bf[k]))), k
    for k, gv in named_grads(model).items():                         # in eval mode the biases in front of BN have real gradients
        fro = float(np.linalg.norm(gv - G[k]) / np.linalg.norm(G[k]))
        assert fro <= (1e-3 if prec == "fp32" else 1e-2) and scaled_err(gv, G[k]) <= (5e-3 if prec == "fp32" else 2e-2), (k, fro)
