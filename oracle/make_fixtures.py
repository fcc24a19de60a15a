#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference on CPU (build container only).

    cd /root/repo && python oracle/make_fixtures.py            # needs /root/reference

TEST INFRASTRUCTURE ONLY.  The reference's Python never travels to the GPU box; what
travels is the data this script writes (inputs, injected noise, expected outputs).

How noise is pinned: the reference draws its dropout masks through
`torch.nn.functional.dropout` (nn.Dropout in src/models/encoders.py:16,34,38) and its
eps through `torch.randn_like` (src/models/vae.py:14).  Both are patched for the
duration of a reference call so that they consume the arrays produced by
`np_oracle.make_noise` instead of torch's RNG; everything else (Linear, BatchNorm1d,
losses, autograd, torch.optim.AdamW) is the reference's own code and stock torch.

Large tensors (numel > FULL_LIMIT) are stored as a fixed pseudo-random sample of
SAMPLE elements plus their sum and sum of squares, to keep fixtures small.
"""
import os
import sys
import contextlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("MMVAE_REFERENCE", "/root/reference")
sys.path.insert(0, REF)          # `src` resolves to the REFERENCE package in this process
sys.path.insert(0, HERE)

import np_oracle as O                                     # noqa: E402
from src.models import MultiModalVAE, RNA2DNAVAE, DNA2RNAVAE   # noqa: E402  (reference)
from src.utils.losses import vae_loss                      # noqa: E402  (reference)
from src.utils.directional_losses import rna2dna_loss, dna2rna_loss  # noqa: E402

FULL_LIMIT = 30000
SAMPLE = 4096


def pack(name, arr, out):
    arr = np.asarray(arr)
    if arr.size <= FULL_LIMIT:
        out[name] = arr
    else:
        idx = np.random.default_rng(sum(map(ord, name))).choice(arr.size, SAMPLE, replace=False)
        idx.sort()
        flat = arr.reshape(-1)
        out[name + "@idx"] = idx.astype(np.int64)
        out[name + "@val"] = flat[idx]
        out[name + "@sum"] = np.float64(flat.astype(np.float64).sum())
        out[name + "@sumsq"] = np.float64((flat.astype(np.float64) ** 2).sum())
        out[name + "@shape"] = np.asarray(arr.shape, np.int64)


@contextlib.contextmanager
def injected_noise(mask_list, eps):
    """Patch F.dropout / torch.randn_like to replay the given noise in call order."""
    import torch.nn.functional as F
    masks = [torch.from_numpy(m.astype(np.float32)) for m in mask_list]
    state = {"i": 0}
    orig_drop, orig_randn = F.dropout, torch.randn_like

    def drop(input, p=0.5, training=True, inplace=False):
        if not training:
            return input
        m = masks[state["i"]]
        state["i"] += 1
        assert m.shape == input.shape, (m.shape, input.shape)
        return input * m / (1.0 - p)

    def randn_like(t, **kw):
        assert tuple(t.shape) == eps.shape
        return torch.from_numpy(eps.astype(np.float32))

    F.dropout = drop
    torch.randn_like = randn_like
    try:
        yield
    finally:
        F.dropout = orig_drop
        torch.randn_like = orig_randn


def load_into(model, P, Bf, rename=None):
    sd = {}
    for k, v in list(P.items()) + list(Bf.items()):
        kk = k
        if rename:
            top = k.split(".", 1)[0]
            if top not in rename:
                continue
            kk = rename[top] + "." + k.split(".", 1)[1]
        sd[kk] = torch.from_numpy(np.array(v))
    missing = model.load_state_dict(sd, strict=True)
    return missing


def t2n(t):
    return t.detach().cpu().numpy().copy()


def case_multimodal(name, dims, B, seed, n_steps, class_weights, beta, gamma):
    A, D, S, L, E = dims
    P, Bf = O.make_params(seed, A, D, S, L, E)
    a, b, site = O.make_batch(seed + 1, B, A, D, S)
    torch.manual_seed(0)
    model = MultiModalVAE(A, D, S, L, embed_dim=E)
    load_into(model, P, Bf)
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=1e-5)
    cw = None if class_weights is None else torch.from_numpy(class_weights)
    ta, tb, ts = torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(site)
    out = dict(dims=np.asarray(dims, np.int64), B=np.int64(B), seed=np.int64(seed),
               n_steps=np.int64(n_steps), beta=np.float64(beta), gamma=np.float64(gamma),
               lr=np.float64(5e-4), wd=np.float64(1e-5))
    if class_weights is not None:
        out["class_weights"] = class_weights
    out["a"], out["b"], out["site"] = a, b, site

    model.train()
    for step in range(n_steps):
        masks, eps = O.make_noise(seed + 100 + step, B, L)
        ml = [masks["encoder_a.fc.3"], masks["encoder_b.fc.3"], masks["encoder_b.fc.7"]]
        with injected_noise(ml, eps):
            ra, rb, rc, mu, lv = model(a=ta, b=tb, site=ts)
            loss, rec, cls, kld = vae_loss(ra, ta, rb, tb, rc, ts, mu, lv, beta=beta, gamma=gamma,
                                           class_weights=cw)
        opt.zero_grad()
        loss.backward()
        pre = f"s{step}."
        if step == 0:
            pack(pre + "out_a", t2n(ra), out); pack(pre + "out_b", t2n(rb), out)
            pack(pre + "out_c", t2n(rc), out)
            pack(pre + "mu", t2n(mu), out); pack(pre + "logvar", t2n(lv), out)
        out[pre + "loss"] = np.asarray([loss.item(), rec, cls, kld], np.float64)
        if step in (0, n_steps - 1):
            for k, p in model.named_parameters():
                pack(pre + "grad." + k, t2n(p.grad), out)
        opt.step()
    for k, v in model.state_dict().items():
        pack("final." + k, t2n(v), out)

    # eval-mode forward after training (running stats, no dropout, eps still sampled: vae.py:73)
    model.eval()
    _, eps = O.make_noise(seed + 900, B, L)
    with torch.no_grad(), injected_noise([], eps):
        ra, rb, rc, mu, lv = model(a=ta, b=tb, site=ts)
        for nm, t in zip(["out_a", "out_b", "out_c", "mu", "logvar"], [ra, rb, rc, mu, lv]):
            pack("eval." + nm, t2n(t), out)
        # single-modality inference shapes used by downstream_task.py:32,48
        for tag, kw in (("only_a", dict(a=ta)), ("only_b", dict(b=tb)), ("only_site", dict(site=ts)),
                        ("a_site", dict(a=ta, site=ts))):
            ra, rb, rc, mu, lv = model(**kw)
            for nm, t in zip(["out_a", "out_b", "out_c", "mu", "logvar"], [ra, rb, rc, mu, lv]):
                pack(f"eval.{tag}.{nm}", t2n(t), out)
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, f"{os.path.getsize(path)/1e6:.2f} MB", "loss0", out["s0.loss"])


def case_directional(name, kind, dims, B, seed, beta):
    A, D, S, L, E = dims
    P, Bf = O.make_params(seed, A, D, S, L, E)
    a, b, site = O.make_batch(seed + 1, B, A, D, S)
    ren = O.directional_param_names(kind, A, D, S, L, E)
    torch.manual_seed(0)
    model = (RNA2DNAVAE if kind == "rna2dna" else DNA2RNAVAE)(A, D, S, L, embed_dim=E)
    load_into(model, P, Bf, ren)
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=1e-5)
    ta, tb, ts = torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(site)
    masks, eps = O.make_noise(seed + 100, B, L)
    out = dict(dims=np.asarray(dims, np.int64), B=np.int64(B), seed=np.int64(seed), beta=np.float64(beta),
               a=a, b=b, site=site)
    model.train()
    if kind == "rna2dna":
        ml = [masks["encoder_a.fc.3"]]
        with injected_noise(ml, eps):
            rec, mu, lv = model(rna=ta, site=ts)
            loss, r, k = rna2dna_loss(rec, tb, mu, lv, beta=beta)
    else:
        ml = [masks["encoder_b.fc.3"], masks["encoder_b.fc.7"]]
        with injected_noise(ml, eps):
            rec, mu, lv = model(dna=tb, site=ts)
            loss, r, k = dna2rna_loss(rec, ta, mu, lv, beta=beta)
    opt.zero_grad()
    loss.backward()
    pack("s0.out", t2n(rec), out); pack("s0.mu", t2n(mu), out); pack("s0.logvar", t2n(lv), out)
    out["s0.loss"] = np.asarray([loss.item(), r, k], np.float64)
    for kname, p in model.named_parameters():
        pack("s0.grad." + kname, t2n(p.grad), out)
    opt.step()
    for kname, v in model.state_dict().items():
        pack("final." + kname, t2n(v), out)
    # site=None inference path (reconstruct_unmatched.py:193)
    model.eval()
    with torch.no_grad(), injected_noise([], eps):
        rec, mu, lv = model(rna=ta) if kind == "rna2dna" else model(dna=tb)
        pack("eval.nosite.out", t2n(rec), out); pack("eval.nosite.mu", t2n(mu), out)
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, f"{os.path.getsize(path)/1e6:.2f} MB", "loss0", out["s0.loss"])


def case_loss_edges():
    """Known-answer values of the reference loss at its edges (BCE clamp, weights)."""
    out = {}
    B, A, D, S, L = 4, 3, 5, 4, 2
    rng = np.random.default_rng(7)
    ra = rng.standard_normal((B, A)).astype(np.float32)
    a = rng.standard_normal((B, A)).astype(np.float32)
    rb = rng.uniform(0.05, 0.95, (B, D)).astype(np.float32)
    rb[0, 0], rb[0, 1], rb[1, 0], rb[1, 1] = 0.0, 1.0, 0.0, 1.0      # saturated predictions
    b = rng.uniform(0, 1, (B, D)).astype(np.float32)
    b[0, 0], b[0, 1], b[1, 0], b[1, 1] = 1.0, 0.0, 0.0, 1.0          # worst / best case targets
    rc = (3 * rng.standard_normal((B, S))).astype(np.float32)
    site = np.array([0, 3, 1, 1], np.int64)
    mu = rng.standard_normal((B, L)).astype(np.float32)
    lv = rng.standard_normal((B, L)).astype(np.float32)
    w = np.array([0.5, 2.0, 1.0, 4.0], np.float32)
    for tag, cw in (("now", None), ("w", w)):
        ts = [torch.from_numpy(x).requires_grad_(x.dtype != np.int64) for x in (ra, rb, rc, mu, lv)]
        loss, r, c, k = vae_loss(ts[0], torch.from_numpy(a), ts[1], torch.from_numpy(b), ts[2],
                                 torch.from_numpy(site), ts[3], ts[4], beta=0.25, gamma=0.7,
                                 class_weights=None if cw is None else torch.from_numpy(cw))
        loss.backward()
        out[tag + ".loss"] = np.asarray([loss.item(), r, c, k], np.float64)
        for nm, t in zip(["recon_a", "recon_b", "recon_c", "mu", "logvar"], ts):
            out[tag + ".grad." + nm] = t2n(t.grad)
    out.update(recon_a=ra, a=a, recon_b=rb, b=b, recon_c=rc, site=site, mu=mu, logvar=lv, w=w)
    path = os.path.join(ROOT, "tests", "golden", "loss_edges.npz")
    np.savez_compressed(path, **out)
    print("loss_edges ->", path, out["now.loss"], out["w.loss"])


if __name__ == "__main__":
    torch.set_num_threads(8)
    w24 = np.random.default_rng(3).uniform(0.3, 3.0, 24).astype(np.float32)
    w5 = np.random.default_rng(4).uniform(0.3, 3.0, 5).astype(np.float32)
    # tiny dims (hidden widths are fixed by the reference), everything stored in full where small
    case_multimodal("mm_tiny_b16", (24, 40, 5, 4, 8), 16, 11, 3, w5, 1e-3, 1.0)
    # BASELINE configs[0]: default dims, batch 32
    case_multimodal("mm_default_b32", (782, 572, 24, 20, 32), 32, 21, 3, None, 1e-3, 1.0)
    # ragged batch (not a multiple of any tile), class weights, non-default beta/gamma
    case_multimodal("mm_default_b77_w", (782, 572, 24, 20, 32), 77, 31, 2, w24, 0.5, 0.3)
    case_directional("rna2dna_b32", "rna2dna", (782, 572, 24, 20, 32), 32, 41, 1e-3)
    case_directional("dna2rna_b32", "dna2rna", (782, 572, 24, 20, 32), 32, 51, 1e-3)
    case_loss_edges()
