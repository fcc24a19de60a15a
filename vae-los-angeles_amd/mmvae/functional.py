"""torch.autograd bridges: the reference's callers drive training through
`loss.backward()` and `optimizer.step()` (optimize_hyperparameters.py:111-113), so the HIP
launch sequences of `engine.VAEGraph` sit behind `torch.autograd.Function`s.

Two gradient hand-offs exist between the loss and the model:
  * general: any loss may be applied to the model outputs; fp32 gradients arrive through
    autograd and are consumed directly by the backward GEMMs (converted on the fly).
  * fused (what `src.utils.losses.vae_loss` uses when it is given the outputs of one forward
    of our own model): the loss kernel already writes activation-typed gradients
    (w.r.t. the pre-sigmoid logits for DecoderB) into buffers owned by that forward's saved
    state; autograd then only carries `None`s plus the scalar grad_output, and no fp32
    gradient of the size of the reconstructions is ever materialised.
"""
import torch
from torch.autograd.function import once_differentiable

from . import ops
from .ops import ceil_to, act_dtype


class VAEGraphFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, prec, train, xa, xb, site, *params):
        ctx.set_materialize_grads(False)
        outs, mu, logvar, saved = graph.forward(prec, xa, xb, site, train)
        ctx.graph, ctx.saved, ctx.n_out = graph, saved, len(outs)
        ctx.param_ids = [id(p) for p in graph.param_list()]
        graph._last_saved = saved
        return (*outs, mu, logvar)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        graph, saved, n_out = ctx.graph, ctx.saved, ctx.n_out
        if saved.get("consumed"):
            raise RuntimeError("backward through this forward was already run (buffers are not retained)")
        g_outs = [None if g is None else _as_rows(g) for g in gs[:n_out]]
        g_mu = None if gs[n_out] is None else gs[n_out].contiguous().float()
        g_lv = None if gs[n_out + 1] is None else gs[n_out + 1].contiguous().float()
        flags = [False] * n_out
        stash = saved.get("loss_grads")
        if stash is not None and stash.get("armed"):                 # the loss handle's backward ran: its gradients are the stash
            if stash.get("scale") is not None:                       # None: unit_grad promised by the caller (fused_loss)
                ops.scale_many(list(stash["g_outs"]) + [stash["g_mu"], stash["g_lv"]], stash["scale"])       # one launch, not five
            for i in range(n_out):
                sg = stash["g_outs"][i]
                if sg is None:
                    continue
                if g_outs[i] is not None:            # rare: another loss term also touched this output
                    sg[:, :g_outs[i].shape[1]] += _to_stash_space(g_outs[i], saved["dec"][i][1], graph.decoders[i], sg.dtype)
                g_outs[i], flags[i] = sg, graph.decoders[i].final_sigmoid
            for key, cur in (("g_mu", g_mu), ("g_lv", g_lv)):
                sg = stash[key]
                if cur is not None:
                    sg += cur
            g_mu, g_lv = stash["g_mu"], stash["g_lv"]
        flat, grads = graph.backward(saved, g_outs, flags, g_mu, g_lv)
        saved["consumed"] = True
        used = _participating(graph, saved, g_outs)
        out = []
        for p in graph.param_list():
            out.append(grads[p] if id(p) in used else None)
        del grads                                     # keep the views unique so autograd can adopt them
        return (None, None, None, None, None, None, *out)


def _as_rows(g):
    if g.dtype not in (torch.float32, torch.bfloat16):
        g = g.float()
    if g.dim() != 2 or g.stride(1) != 1 or (g.shape[0] > 1 and g.stride(0) < g.shape[1]):
        g = g.contiguous()
    return g


def _to_stash_space(g, out, dec, dtype):
    if dec.final_sigmoid:
        g = g * out * (1.0 - out)
    return g.to(dtype)


def _participating(graph, saved, g_outs):
    used = set()
    blocks = []
    if "enc_a" in saved:
        blocks.append(graph.enc_a)
    if "enc_b" in saved:
        blocks.append(graph.enc_b)
    if "site" in saved:
        blocks.append(graph.enc_c)
    for dec, g in zip(graph.decoders, g_outs):
        if g is not None:
            blocks.append(dec)
    for b in blocks:
        used.update(id(p) for p in b.params())
    return used


def run_graph(graph, prec, train, xa, xb, site):
    """Forward of a VAEGraph with autograd wiring.  Returns (outs, mu, logvar)."""
    params = graph.param_list()
    # a backward will follow: the forward's ONE memset then also zeroes the gradient arena, the backward's BatchNorm sums, the table
    # gradient and the loss accumulators (3 fill launches per step -> 1)
    graph._want_bwd = train and torch.is_grad_enabled() and any(p.requires_grad for p in params)
    try:
        res = VAEGraphFn.apply(graph, prec, train, xa, xb, site, *params)
    finally:
        graph._want_bwd = False
    n = len(graph.decoders)
    outs, mu, logvar = list(res[:n]), res[n], res[n + 1]
    saved = graph._last_saved
    graph._last_saved = None
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        tag = {"saved": saved, "inputs": (xa, xb, site)}
        for i, o in enumerate(outs):
            o._mmvae = (tag, "out", i)
        mu._mmvae = (tag, "mu", 0)
        logvar._mmvae = (tag, "logvar", 0)
    return outs, mu, logvar


# --------------------------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------------------------
class _LossHandleFn(torch.autograd.Function):
    """Connects the fused loss value to the model's autograd node; gradients travel through the
    stash (see module docstring), autograd only delivers grad_output."""

    @staticmethod
    def forward(ctx, total, stash, *model_outputs):
        ctx.stash = stash
        return total.view(())            # a view of the kernel's output buffer (not of an input that requires grad): no copy launch

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if not ctx.stash.get("unit_grad"):
            ctx.stash["scale"] = g.reshape(1).float().contiguous()
        ctx.stash["armed"] = True
        return (None, None) + (None,) * 16


class _LossGeneralFn(torch.autograd.Function):
    """Fused loss for arbitrary inputs: fp32 gradients are produced by the same kernel pass and
    handed to autograd."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        ctx.set_materialize_grads(False)
        total, grads = spec(tensors)
        ctx.grads = grads
        return total

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        scale = g.reshape(1).float().contiguous()
        out = []
        for t in ctx.grads:
            if t is not None:
                ops.scale_if_needed(t, scale)
            out.append(t)
        return (None, *out)


def engine_noise():
    from . import engine
    return engine.GLOBAL_NOISE


def _tag_of(t):
    return getattr(t, "_mmvae", None) if t is not None else None


def fused_loss(terms, beta, gamma, class_weights=None, unit_grad=False, beta_gamma_dev=None):
    """unit_grad=True: the caller promises to call `total.backward()` with the default gradient of 1 (a captured training step
    does): the stashed gradients are then used as they are, without the launch that multiplies them by the incoming gradient.
    beta_gamma_dev: device float32[2] overriding (beta, gamma): a captured step follows the beta warm-up without re-capture."""
    if not next(v[0] for v in terms.values() if v is not None).is_cuda:
        raise RuntimeError("the MI355X loss kernel needs CUDA/HIP tensors; there is no CPU fallback")
    with ops.pinned_stream():
        return _fused_loss(terms, beta, gamma, class_weights, unit_grad, beta_gamma_dev)


def _validate_loss_args(dev, B, ra, a, rb, b, lg, site, mu, lv, class_weights):
    """The kernel takes raw pointers: everything torch's own losses would reject (F.mse_loss / binary_cross_entropy /
    cross_entropy on mismatched devices or shapes, losses.py:31,34,39) is rejected here, before any launch."""
    for nm, pred, tgt in (("a", ra, a), ("b", rb, b), ("logvar", mu, lv)):
        if pred is None:
            continue
        if tgt.device != pred.device or pred.device != dev:
            raise RuntimeError(f"vae_loss: '{nm}' tensors are on different devices ({pred.device} vs {tgt.device}); "
                               "the MI355X loss kernel needs all of them on one CUDA/HIP device")
        if pred.dim() != 2 or tuple(tgt.shape) != tuple(pred.shape) or pred.shape[0] != B:
            raise RuntimeError(f"vae_loss: shape mismatch for '{nm}': prediction {tuple(pred.shape)} vs target {tuple(tgt.shape)} (batch {B})")
    if lg is not None:
        if site.device != dev or lg.device != dev:
            raise RuntimeError(f"vae_loss: site labels / logits are on another device ({site.device}, {lg.device}) than {dev}")
        if lg.dim() != 2 or lg.shape[0] != B or site.dim() != 1 or site.shape[0] != B:
            raise RuntimeError(f"vae_loss: site must have shape ({B},) for logits {tuple(lg.shape)}, got {tuple(site.shape)}")
        if site.dtype in (torch.float16, torch.float32, torch.float64, torch.bfloat16, torch.bool):
            raise RuntimeError(f"vae_loss: site labels must be integer class indices, got {site.dtype}")
        if class_weights is not None and class_weights.numel() != lg.shape[1]:
            raise RuntimeError(f"vae_loss: class_weights has {class_weights.numel()} entries for {lg.shape[1]} classes")


def _fused_loss(terms, beta, gamma, class_weights=None, unit_grad=False, beta_gamma_dev=None):
    """terms: dict with optional entries
         'a': (recon_a, a)  sum-MSE           'b': (recon_b, b)  sum-BCE (clamped logs)
         'c': (logits, site) weighted sum-CE  'kl': (mu, logvar)
    Returns (total (0-dim tensor, differentiable), out5 (device fp32 [total, recon, class, kld, labels out of range]))."""
    any_t = next(v[0] for v in terms.values() if v is not None)
    dev, B = any_t.device, any_t.shape[0]
    if not any_t.is_cuda:
        raise RuntimeError("the MI355X loss kernel needs CUDA/HIP tensors; there is no CPU fallback")
    ra, a = terms.get("a") or (None, None)
    rb, b = terms.get("b") or (None, None)
    lg, site = terms.get("c") or (None, None)
    mu, lv = terms.get("kl") or (None, None)
    _validate_loss_args(dev, B, ra, a, rb, b, lg, site, mu, lv, class_weights)
    if site is not None and site.dtype != torch.int64:
        site = site.long()
    if site is not None:
        site = site.contiguous()
    cw = None if class_weights is None else class_weights.to(device=dev, dtype=torch.float32).contiguous()
    need_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (ra, rb, lg, mu, lv))

    # reconstruction terms whose loss already ran inside the decoder's last GEMM (engine.VAEGraph.fused_recon): the "reconstruction"
    # is a placeholder, its loss sum sits in the forward's accumulators and its gradient in the forward's saved state
    fused, fused_idx = None, {}
    for key, (r, t) in (("a", (ra, a)), ("b", (rb, b))):
        tg = _tag_of(r) if r is not None else None
        fr = tg[0]["saved"].get("fused_recon") if tg is not None else None
        if fr is not None and tg[1] == "out" and tg[2] in fr["g"]:
            if fr["targets"][tg[2]].data_ptr() != t.data_ptr() or tuple(fr["targets"][tg[2]].shape) != tuple(t.shape):
                raise RuntimeError("fused reconstruction loss: the target passed to the loss is not the tensor the forward was given")
            fused, fused_idx[key] = fr, tg[2]
    if fused is not None and not (torch.is_grad_enabled()):
        raise RuntimeError("fused reconstruction loss needs the training step (gradients enabled)")
    if "a" in fused_idx:
        ra_keep, ra = ra, None
    if "b" in fused_idx:
        rb_keep, rb = rb, None

    def prep(x):
        return None if x is None else (x if (x.dtype == torch.float32 and x.stride(-1) == 1) else x.float().contiguous())
    ra_, a_, rb_, b_, lg_, mu_, lv_ = (prep(ra), prep(a), prep(rb), prep(b), prep(lg), prep(mu), prep(lv))
    if mu_ is not None:
        mu_, lv_ = mu_.contiguous(), lv_.contiguous()
    if fused is not None:
        sums, out4 = fused["sums"], fused["out5"]         # the decoders' GEMMs have already added their terms
    else:
        sums = out4 = None                                 # out4: [total, recon, class, kld, labels out of range]

    # ---- fused hand-off: all differentiable inputs are outputs of ONE forward of our model --------------------
    tags = [_tag_of(t) for t in (ra, rb, lg, mu, lv) if t is not None and t.requires_grad]
    tag = tags[0][0] if tags and all(t is not None and t[0] is tags[0][0] for t in tags) else None
    if need_grad and tag is not None and not tag["saved"].get("consumed") and "loss_grads" not in tag["saved"]:
        saved = tag["saved"]
        if sums is None:
            sums, out4 = saved.pop("loss_ws", None) or ops.loss_workspace(dev)     # zeroed by the forward's one memset
        adt = act_dtype(saved["prec"])
        n_dec = len(saved["dec"])
        g_outs = [None] * n_dec
        for key, i in fused_idx.items():
            g_outs[i] = fused["g"][i]
        ga = gb = gc = None
        if ra is not None and ra.requires_grad:
            ga = torch.empty(B, ceil_to(ra.shape[1], 8), dtype=adt, device=dev)
            g_outs[_tag_of(ra)[2]] = ga
        if rb is not None and rb.requires_grad:
            gb = torch.empty(B, ceil_to(rb.shape[1], 8), dtype=adt, device=dev)
            g_outs[_tag_of(rb)[2]] = gb
        if lg is not None and lg.requires_grad:
            gc = torch.empty(B, lg.shape[1], dtype=torch.float32, device=dev)
            g_outs[_tag_of(lg)[2]] = gc
        g_mu = torch.empty(B, mu.shape[1], dtype=torch.float32, device=dev) if mu is not None else None
        g_lv = torch.empty_like(g_mu) if mu is not None else None
        ops.vae_loss(B, recon_a=ra_, a=a_, recon_b=rb_, b=b_, logits=lg_, site=site, class_weights=cw, mu=mu_, logvar=lv_,
                     beta=beta, gamma=gamma, sums=sums, g_a=ga, g_b=gb, grad_b_wrt_logit=True, g_c=gc, g_mu=g_mu, g_lv=g_lv,
                     beta_gamma_dev=beta_gamma_dev)
        ops.loss_finalize(sums, beta, gamma, out4, beta_gamma_dev)
        stash = {"g_outs": g_outs, "g_mu": g_mu, "g_lv": g_lv, "scale": None, "unit_grad": bool(unit_grad)}
        if g_mu is None:                       # KL term absent: nothing flows into mu/logvar from this loss
            stash["g_mu"] = torch.zeros(B, saved["logvar"].shape[1], dtype=torch.float32, device=dev)
            stash["g_lv"] = torch.zeros_like(stash["g_mu"])
        saved["loss_grads"] = stash
        pads = [t for t in (ra, rb, lg, mu, lv) if t is not None]
        if "a" in fused_idx:
            pads.append(ra_keep)
        if "b" in fused_idx:
            pads.append(rb_keep)
        pads += [None] * (16 - len(pads))
        total = _LossHandleFn.apply(out4[0], stash, *pads)
        return total, out4

    if fused is not None:
        raise RuntimeError("fused reconstruction loss: the loss terms are not all outputs of the one forward that computed it")

    # ---- general path --------------------------------------------------------------------------------------------
    if sums is None:
        sums, out4 = ops.loss_workspace(dev)

    def spec(tensors):
        ga = torch.empty_like(ra_) if (need_grad and ra is not None and ra.requires_grad) else None
        gb = torch.empty_like(rb_) if (need_grad and rb is not None and rb.requires_grad) else None
        gc = torch.empty_like(lg_) if (need_grad and lg is not None and lg.requires_grad) else None
        gm = torch.empty_like(mu_) if (need_grad and mu is not None and mu.requires_grad) else None
        gl = torch.empty_like(lv_) if (need_grad and lv is not None and lv.requires_grad) else None
        ops.vae_loss(B, recon_a=ra_, a=a_, recon_b=rb_, b=b_, logits=lg_, site=site, class_weights=cw, mu=mu_, logvar=lv_,
                     beta=beta, gamma=gamma, sums=sums, g_a=ga, g_b=gb, grad_b_wrt_logit=False, g_c=gc, g_mu=gm, g_lv=gl,
                     beta_gamma_dev=beta_gamma_dev)
        ops.loss_finalize(sums, beta, gamma, out4, beta_gamma_dev)
        grads = []
        for t, g in ((ra, ga), (rb, gb), (lg, gc), (mu, gm), (lv, gl)):
            if t is not None:
                grads.append(g)
        return out4[0].clone(), grads

    inputs = [t for t in (ra, rb, lg, mu, lv) if t is not None]
    total = _LossGeneralFn.apply(spec, *inputs)
    return total, out4


def read_losses(out5):
    """The ONE host read of the loss (the reference's three `.item()` calls, losses.py:46): [total, recon, class, kld] as
    floats.  Raises if the kernel met a class index outside [0, n_sites) (torch's cross_entropy device-asserts there)."""
    vals = out5.tolist()
    if vals[4] != 0.0:
        raise RuntimeError(f"vae_loss: {int(vals[4])} class index(es) in `site` outside [0, n_classes)")
    return vals[:4]


class ReparamFn(torch.autograd.Function):
    """Stand-alone reparameterize(mu, logvar) (reference src/models/vae.py:11-15) on the HIP kernels: the fused
    mean-fusion + reparameterisation launch with ONE modality; backward d_mu = g, d_logvar = g * eps * exp(logvar / 2) / 2."""

    @staticmethod
    def forward(ctx, mu, logvar):
        if not (mu.is_cuda and logvar.is_cuda):
            raise RuntimeError(f"reparameterize: the MI355X path needs CUDA/HIP tensors (got {mu.device}); there is no CPU fallback")
        if mu.dim() != 2 or mu.shape != logvar.shape:
            raise RuntimeError(f"reparameterize: expected two (B, L) tensors, got {tuple(mu.shape)} and {tuple(logvar.shape)}")
        B, Ld = mu.shape
        heads = torch.cat([mu.detach().float(), logvar.detach().float()], dim=1).contiguous()
        eps = engine_noise().draw(B, [], Ld, mu.device)[1]
        mu_o, lv_o = torch.empty_like(heads[:, :Ld]).contiguous(), torch.empty_like(heads[:, :Ld]).contiguous()
        z = torch.empty(B, Ld, dtype=torch.float32, device=mu.device)
        with ops.pinned_stream():
            ops.fuse_reparam_fwd(B, Ld, heads, None, None, None, eps, mu_o, lv_o, z)
        ctx.save_for_backward(eps, lv_o)
        return z

    @staticmethod
    def backward(ctx, g):
        eps, lv = ctx.saved_tensors
        B, Ld = lv.shape
        d_heads = torch.empty(B, 2 * Ld, dtype=torch.float32, device=g.device)
        with ops.pinned_stream():
            ops.fuse_reparam_bwd(B, Ld, 1, None, None, [g.contiguous().float()], eps, lv, d_heads, None, None)
        return d_heads[:, :Ld], d_heads[:, Ld:]


# --------------------------------------------------------------------------------------------
# stand-alone blocks (EncoderA/B/C and DecoderA/B/C used on their own)
# --------------------------------------------------------------------------------------------
class BlockRuntime:
    """Weight preparation cache for one block used outside a VAEGraph."""

    def __init__(self, block):
        self.block, self._prep, self._key = block, None, None

    def ensure(self, prec, device):
        key = (prec, str(device)) + tuple(p.data_ptr() for p in self.block.params())
        if self._prep is None or key != self._key:
            pls = self.block.prepare(prec, device)
            self._prep = ops.WeightPrep(pls, device) if pls else None
            self._key = key
        if self._prep is not None:
            self._prep.run()


def _alloc_block_grads(block, device):
    params = block.params()
    flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=device)
    views, off = {}, 0
    for p in params:
        views[p] = flat[off:off + p.numel()].view(p.shape)
        off += p.numel()
    return views


class EncoderMLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rt, prec, train, noise, x, *params):
        from .engine import _check_input
        ctx.set_materialize_grads(False)
        x = _check_input(x.reshape(x.shape[0], -1), "x", rt.block.in_dim)
        rt.ensure(prec, x.device)
        masks = noise.draw(x.shape[0], rt.block.widths(), None, x.device)[0] if train else None
        heads, saved = rt.block.forward(prec, x, train, masks)
        ctx.rt, ctx.prec, ctx.saved, ctx.train = rt, prec, saved, train
        Ld = rt.block.latent
        return heads[:, :Ld], heads[:, Ld:]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_mu, g_lv):
        blk = ctx.rt.block
        Ld = blk.latent
        y = ctx.saved[-1][2]
        d_heads = torch.zeros(y.shape[0], 2 * Ld, dtype=torch.float32, device=y.device)
        if g_mu is not None:
            d_heads[:, :Ld].copy_(g_mu)
        if g_lv is not None:
            d_heads[:, Ld:].copy_(g_lv)
        grads = _alloc_block_grads(blk, y.device)
        blk.backward(ctx.prec, ctx.saved, d_heads, grads, train=ctx.train)
        out = [grads[p] for p in blk.params()]
        del grads
        return (None, None, None, None, None, *out)


class EmbedEncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, block, site, *params):
        ctx.set_materialize_grads(False)
        if not site.is_cuda:
            raise RuntimeError("the MI355X path needs CUDA/HIP tensors; there is no CPU fallback")
        site = site.long().contiguous()
        B, Ld, dev = site.shape[0], block.latent, site.device
        table = block.table()
        mu = torch.empty(B, Ld, dtype=torch.float32, device=dev)
        lv = torch.empty(B, Ld, dtype=torch.float32, device=dev)
        z = torch.empty(B, ceil_to(Ld, 8), dtype=torch.float32, device=dev)
        zeros = torch.zeros(B, Ld, dtype=torch.float32, device=dev)
        ops.fuse_reparam_fwd(B, Ld, None, None, table, site, zeros, mu, lv, z)     # gather only (eps = 0)
        ctx.block, ctx.site = block, site
        return mu, lv

    @staticmethod
    @once_differentiable
    def backward(ctx, g_mu, g_lv):
        blk, site = ctx.block, ctx.site
        B, Ld, dev = site.shape[0], blk.latent, site.device
        zeros = torch.zeros(B, Ld, dtype=torch.float32, device=dev)
        g_mu = zeros if g_mu is None else g_mu.contiguous().float()
        g_lv = zeros if g_lv is None else g_lv.contiguous().float()
        d_heads = torch.empty(B, 2 * Ld, dtype=torch.float32, device=dev)
        d_table = torch.zeros(blk.embedding.weight.shape[0], 2 * Ld, dtype=torch.float32, device=dev)
        # dz = 0 and eps = 0: the kernel reduces to the scatter-add of (g_mu | g_lv) by class
        ops.fuse_reparam_bwd(B, Ld, 1, g_mu, g_lv, [zeros], zeros, zeros, d_heads, d_table, site)
        grads = _alloc_block_grads(blk, dev)
        blk.backward(d_table, grads)
        out = [grads[p] for p in blk.params()]
        del grads
        return (None, None, *out)


class DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rt, prec, z, *params):
        ctx.set_materialize_grads(False)
        if not z.is_cuda:
            raise RuntimeError("the MI355X path needs CUDA/HIP tensors; there is no CPU fallback")
        blk = rt.block
        rt.ensure(prec, z.device)
        B, Ld = z.shape
        za = torch.zeros(B, ceil_to(Ld, 8), dtype=act_dtype(prec), device=z.device)
        za[:, :Ld].copy_(z)
        out, acts = blk.forward(prec, za)
        ctx.rt, ctx.prec, ctx.acts, ctx.out, ctx.Ld = rt, prec, acts, out.detach(), Ld      # detach: no ctx <-> output cycle
        ctx.z_needs = z.requires_grad
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        blk = ctx.rt.block
        if g is None:
            return (None,) * (3 + len(blk.params()))
        g = _as_rows(g)
        dz = torch.zeros(g.shape[0], ctx.Ld, dtype=torch.float32, device=g.device)
        grads = _alloc_block_grads(blk, g.device)
        blk.backward(ctx.prec, ctx.acts, ctx.out, g, False, dz, False, grads)
        out = [grads[p] for p in blk.params()]
        del grads
        return (None, None, dz if ctx.z_needs else None, *out)
