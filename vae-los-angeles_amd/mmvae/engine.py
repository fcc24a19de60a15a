"""Launch orchestration of the MultiModalVAE training path on one MI355X.

The compute blocks mirror the reference modules (file:line in each docstring) but each is a
short sequence of launches into libmmvae_hip.so; no torch arithmetic is used on the data path.
Activations between GEMMs are stored in the *activation type* of the precision mode (bf16 or
f32); BatchNorm statistics, loss sums, gradients of parameters and the optimiser are fp32.

What is saved for backward per BN layer is only the PRE-BatchNorm GEMM output, the per-column
(mean, rstd, scale, shift) and the dropout keep-mask: post-activation tensors are recomputed
inside the consumer GEMM's operand prologue.
"""
import os

import torch

from . import ops
from . import _lib as L_
from .ops import (PREC_BF16, PREC_F32, ACT_NONE, ACT_RELU, ACT_SIGMOID, EPI_RELU_MASK, EPI_BN_BWD, TILE, DROP_P,
                  ceil_to, act_dtype)

_PRECISIONS = {"bf16": PREC_BF16, "fp32": PREC_F32, "f32": PREC_F32}
_default_precision = _PRECISIONS[os.environ.get("MMVAE_PRECISION", "bf16").lower()]
_FUSE_BN_APPLY = os.environ.get("MMVAE_FUSE_BN_APPLY", "1") == "1"       # A/B switch: BN-backward correction of first layers inside the dW GEMM
_GROUP_TINY_DW = os.environ.get("MMVAE_GROUP_TINY_DW", "1") == "1"       # A/B switch: small-output dW GEMMs as grouped launches
_FOLD_BN_FINALIZE = os.environ.get("MMVAE_NO_FOLD_BN_FINALIZE") is None      # A/B switch: mmvae_bn_finalize / mmvae_bn_bwd_finalize inside their consumers
_TINY_DW_MAX = 16384                                                       # N*K at or below which a dW GEMM counts as small-output


def set_default_precision(name):
    """'bf16' (bf16 MFMA operands/activations, f32 accumulate) or 'fp32' (f32 MFMA)."""
    global _default_precision
    _default_precision = _PRECISIONS[name.lower()]


def default_precision():
    return _default_precision


# --------------------------------------------------------------------------------------------
# noise
# --------------------------------------------------------------------------------------------
class NoiseSource:
    """Dropout keep-masks and eps.  Default: ONE Philox launch per forward (all masks of the pass in one uint8
    buffer + eps), keyed by (torch.initial_seed() [+ rank], a DEVICE-resident running offset) -- the offset lives on
    the device so that a captured hipGraph draws fresh noise on every replay.  `inject` replays explicit arrays in the
    order the reference consumes its RNG (EncoderA mask, EncoderB masks, eps): parity tests."""

    def __init__(self):
        self._injected = None
        self._offsets = {}            # device -> int64[1] tensor (bit pattern of the uint64 Philox offset)
        self.stream_rank = None       # None: torch.distributed's rank; an int: draw THAT rank's stream (single-process restatement of N ranks)

    def inject(self, masks, eps):
        self._injected = (list(masks), eps)

    def clear(self):
        self._injected = None

    def _seed(self):
        seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        if self.stream_rank is not None:
            seed = (seed + 0x9E3779B97F4A7C15 * (int(self.stream_rank) + 1)) & 0xFFFFFFFFFFFFFFFF
        elif torch.distributed.is_available() and torch.distributed.is_initialized():
            seed = (seed + 0x9E3779B97F4A7C15 * (torch.distributed.get_rank() + 1)) & 0xFFFFFFFFFFFFFFFF
        return seed

    def offset_tensor(self, device):
        t = self._offsets.get(device)
        if t is None:
            t = self._offsets[device] = torch.zeros(L_.CTR_COPIES, dtype=torch.int64, device=device)    # identical copies, see mmvae_noise
        return t

    def state_dict(self, device):
        """Position of the Philox stream on `device` (one host read): part of a resumable checkpoint."""
        t = self._offsets.get(torch.device(device) if not isinstance(device, torch.device) else device)
        return {"offset": 0 if t is None else int(t[0].item())}

    def load_state_dict(self, state, device):
        device = torch.device(device) if not isinstance(device, torch.device) else device
        self.offset_tensor(device).fill_(int(state["offset"]))

    def draw(self, B, widths, Ld, device):
        """-> ([uint8 (B,w) keep-mask for w in widths], eps fp32 (B,Ld) or None if Ld is None)."""
        if self._injected is not None:
            masks = []
            for w in widths:
                m = self._injected[0].pop(0)
                if tuple(m.shape) != (B, w):
                    raise ValueError(f"injected mask shape {tuple(m.shape)} != {(B, w)}")
                masks.append(m.to(device=device, dtype=torch.uint8).contiguous())
            eps = None
            if Ld is not None:
                eps = self._injected[1]
                if tuple(eps.shape) != (B, Ld):
                    raise ValueError(f"injected eps shape {tuple(eps.shape)} != {(B, Ld)}")
                eps = eps.to(device=device, dtype=torch.float32).contiguous()
            return masks, eps
        segs, total = [], 0
        for w in widths:
            segs.append(total)
            total = ceil_to(total + B * w, 16)
        buf = torch.empty(total, dtype=torch.uint8, device=device) if total else None
        eps = torch.empty(B, Ld, dtype=torch.float32, device=device) if Ld is not None else None
        off = self.offset_tensor(device)
        ops.noise(buf, eps, 1.0 - DROP_P, self._seed(), 0, off, advance=True)        # the launch advances the device offset itself
        return [buf[o:o + B * w].view(B, w) for o, w in zip(segs, widths)], eps


GLOBAL_NOISE = NoiseSource()


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def _check_input(x, name, cols):
    if x.dim() != 2 or x.shape[1] != cols:
        raise RuntimeError(f"{name}: expected shape (B, {cols}), got {tuple(x.shape)}")
    if not x.is_cuda:
        raise RuntimeError(f"{name}: the MI355X path needs CUDA/HIP tensors (got {x.device}); there is no CPU fallback")
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    if x.stride(1) != 1:
        x = x.contiguous()
    return x


def zeros_pack(device, specs):
    """[(numel, dtype)] -> zero-filled tensors carved out of ONE allocation (one memset launch instead of one per tensor)."""
    offs, total = [], 0
    for n, dt in specs:
        offs.append(total)
        total += ceil_to(n * _ITEMSIZE[dt], 16)
    buf = torch.zeros(max(total, 16), dtype=torch.uint8, device=device)
    return [buf[o:o + n * _ITEMSIZE[dt]].view(dt) for o, (n, dt) in zip(offs, specs)]


_ITEMSIZE = {torch.float32: 4, torch.float64: 8, torch.uint8: 1, torch.int64: 8}


_MERGE_DECODER_STEM = os.environ.get("MMVAE_NO_DECODER_STEM") is None          # A/B switch


class BNState:
    """Per-layer BatchNorm vectors: rows of one [4][N] fp32 buffer (mean, rstd, scale, shift)."""

    def __init__(self, N, device):
        self.buf = torch.empty(4, N, dtype=torch.float32, device=device)
        self.mean, self.rstd, self.scale, self.shift = self.buf[0], self.buf[1], self.buf[2], self.buf[3]


# --------------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------------
class EncoderMLP:
    """EncoderA / EncoderB (reference src/models/encoders.py:8-23, 26-46):
    [Linear -> BatchNorm1d -> ReLU -> Dropout(0.1)] x n, then the fc_mu | fc_logvar heads
    computed as ONE GEMM with N = 2*latent."""

    def __init__(self, linears, bns, fc_mu, fc_logvar, name="enc"):
        self.linears, self.bns, self.fc_mu, self.fc_logvar = list(linears), list(bns), fc_mu, fc_logvar
        self.in_dim = self.linears[0].in_features
        self.latent = fc_mu.out_features
        self.name = name

    def prepare(self, prec, device):
        self.pl = [ops.PreparedLinear([l.weight], [l.bias], prec, device) for l in self.linears]
        self.pl_heads = ops.PreparedLinear([self.fc_mu.weight, self.fc_logvar.weight],
                                           [self.fc_mu.bias, self.fc_logvar.bias], prec, device)
        return self.pl + [self.pl_heads]

    def params(self):
        """Order of the gradient arena (heads adjacent so that one TN GEMM writes both)."""
        out = []
        for l, bn in zip(self.linears, self.bns):
            out += [l.weight, l.bias, bn.weight, bn.bias]
        out += [self.fc_mu.weight, self.fc_logvar.weight, self.fc_mu.bias, self.fc_logvar.bias]
        return out

    def widths(self):
        return [l.out_features for l in self.linears]

    def forward(self, prec, x, train, masks, stats_bufs=None, want_bwd=False):
        """masks: one uint8 (B, width) keep-mask per BN layer (training) or None (eval).
        stats_bufs: optional pre-zeroed float64 (2, N) accumulators, one per BN layer."""
        B, dev = x.shape[0], x.device
        adt = act_dtype(prec)
        saved = []
        h, pro = x, None
        fin = None            # the BatchNorm finalisation of the layer that produced h, still owed: it rides in the GEMM that consumes h

        def consume(N, K, out, bias, w, tag, stats=None, pro_out=None):
            """GEMM on (h, pro) that also finalises h's BatchNorm statistics when the library can fold that in; else the launch of its own."""
            nonlocal fin
            if fin is not None and _FOLD_BN_FINALIZE:
                try:
                    ops.gemm_nt(prec, h, w, N, K, out, bias=bias, prologue=pro, stats=stats, tag=tag, pro_out=pro_out, pro_finalize=fin)
                    fin = None
                    return pro_out
                except RuntimeError:
                    pass                                   # refused before anything was enqueued (argument check)
            if fin is not None:
                ops.bn_finalize(0, 0, None, None, None, None, None, None, None, None, None, None, args=fin)
                fin = None
            try:
                ops.gemm_nt(prec, h, w, N, K, out, bias=bias, prologue=pro, stats=stats, tag=tag, pro_out=pro_out)
                return pro_out
            except RuntimeError:
                if pro_out is None:
                    raise
                # the library did not take the problem on the kernel that writes pro_out: the ordinary call, and the backward redoes
                # the prologue on its operand load
                ops.gemm_nt(prec, h, w, N, K, out, bias=bias, prologue=pro, stats=stats, tag=tag)
                return None

        for lin, bn, pl in zip(self.linears, self.bns, self.pl):
            N, K = pl.N, pl.K
            y = torch.empty(B, ceil_to(N, 8), dtype=adt, device=dev)
            st = BNState(N, dev)
            # a hidden BN layer on the wave-specialised kernel: its producers also write the operand AFTER the prologue (the previous
            # layer's post-activation, 2 bytes per element), which lets this layer's dW GEMM run the plain LDS-DMA kernel instead of
            # redoing the prologue on its Q operand (EncoderB's second Linear: 52 -> 39 us for the dW GEMM)
            h_act = None
            if want_bwd and pro is not None and ops.can_keep_pro_out(prec, B, N, K, h, y) and pro[2] is not None \
                    and pro[2].stride(0) % 8 == 0 and pro[2].data_ptr() % 8 == 0:
                h_act = torch.empty(B, K, dtype=torch.bfloat16, device=dev)
            if train:
                stats = stats_bufs[len(saved)] if stats_bufs is not None else torch.zeros(2, N, dtype=torch.float64, device=dev)
                h_act = consume(N, K, y, pl.bias, pl.w, f"{self.name}.L{len(saved)}.fwd", stats=stats, pro_out=h_act)
                # the finalisation of THIS layer's statistics is owed to whoever reads y next (the next layer or the heads)
                fin = ops.bn_finalize_args(B, N, stats, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                           bn.num_batches_tracked, st.mean, st.rstd, st.scale, st.shift, bn.eps,
                                           bn.momentum if bn.momentum is not None else 0.1)
                mask = masks[len(saved)]
                new_pro = (st.scale, st.shift, mask, 1.0 / (1.0 - DROP_P))
            else:
                ops.gemm_nt(prec, h, pl.w, N, K, y, bias=pl.bias, prologue=pro, tag=f"{self.name}.L{len(saved)}.fwd")
                ops.bn_eval_coeffs(bn.weight, bn.bias, bn.running_mean, bn.running_var, st.scale, st.shift, bn.eps, st.mean, st.rstd)
                new_pro = (st.scale, st.shift, None, 1.0)
            saved.append((h, pro, y, st, new_pro) if h_act is None else (h_act, None, y, st, new_pro))      # backward's Q operand: plain when kept
            h, pro = y, new_pro
        heads = torch.empty(B, 2 * self.latent, dtype=torch.float32, device=dev)
        consume(2 * self.latent, self.pl_heads.K, heads, self.pl_heads.bias, self.pl_heads.w, f"{self.name}.heads.fwd")
        return heads, saved

    def backward(self, prec, saved, d_heads, grads, tn=ops.gemm_tn, stats_bufs=None, train=True, d_heads_lp=None):
        """d_heads: [B][2L] fp32.  grads: dict param -> fp32 view (pre-zeroed, accumulated).
        train=False: the forward ran in eval mode (running statistics, no dropout) -- torch's
        batch_norm(training=False) backward: dy = gamma * rstd * d, no batch-statistics correction."""
        B, dev = d_heads.shape[0], d_heads.device
        adt = act_dtype(prec)
        nt = (B + TILE - 1) // TILE
        L2 = 2 * self.latent
        h_in, pro_in, y, st, pro = saved[-1]
        Kl = self.pl_heads.K
        gw = grads[self.fc_mu.weight]            # fc_logvar.weight follows immediately in the arena
        gb = grads[self.fc_mu.bias]
        tn(prec, d_heads, y, _span(gw, L2 * Kl).view(L2, Kl), _span(gb, L2), L2, Kl, q_prologue=pro, tag=f"{self.name}.heads.dW")
        # gradient entering the last hidden layer: dX GEMM of the heads (A = d_heads, W = heads^T)
        src, src_wt, src_n, src_k = (d_heads_lp if d_heads_lp is not None else d_heads), self.pl_heads.wt, Kl, L2
        for i in reversed(range(len(self.linears))):
            lin, bn, pl = self.linears[i], self.bns[i], self.pl[i]
            h_in, pro_in, y, st, pro = saved[i]
            N, K = pl.N, pl.K
            bnargs = (st.scale, st.shift, st.mean, st.rstd, pro[2], pro[3])
            # BatchNorm/ReLU/Dropout backward of layer i: ONE contraction that stores d = dX * keep * relu' and accumulates
            # (sum d, sum d*xhat); then the BN correction in place (mmvae_bn_bwd_apply) or on the dW GEMM's operand load.
            stats = stats_bufs[i] if stats_bufs is not None else torch.zeros(2, N, dtype=torch.float64, device=dev)
            coef = torch.empty(3, N, dtype=torch.float32, device=dev)
            d = torch.empty(B, ceil_to(N, 8), dtype=adt, device=dev)                  # d := dL/dy_i
            ops.gemm_nt(prec, src, src_wt, src_n, src_k, d, epilogue=EPI_BN_BWD, h=y, bn=bnargs, bn_phase=2, stats=stats, tag=f"{self.name}.L{i}.dX")
            # The finalisation of the backward sums (dgamma, dbeta, the three constants per column) rides in the launch that consumes
            # them -- the first layer's dW GEMM or the in-place correction pass -- instead of a 5 us launch of its own (round 3); the
            # library answers ERR_ARG where its kernels cannot do that (small batches, unusual widths): then the separate launch.
            fin = (stats, bn.weight, grads[bn.weight], grads[bn.bias], not train)
            # (not for very wide inputs -- the scaled omics widths: the dW GEMM then has hundreds of K tiles and every one of them
            # would redo the correction of its P rows; one pass over d is cheaper)
            if i == 0 and _FUSE_BN_APPLY and K <= 4096:
                # first layer: only the dW GEMM consumes dL/dy -> the correction rides on its operand load, no pass over d
                if _FOLD_BN_FINALIZE and prec == PREC_BF16 and B >= 8192 and N >= 128 and K >= 256:
                    try:
                        tn(prec, d, h_in, grads[lin.weight], grads[lin.bias], N, K, q_prologue=pro_in,
                           p_prologue=(y, st.mean, st.rstd, None, fin), tag=f"{self.name}.L{i}.dW")
                        continue
                    except RuntimeError:
                        pass                                   # refused before anything was enqueued
                ops.bn_bwd_finalize(B, N, stats, bn.weight, st.rstd, grads[bn.weight], grads[bn.bias], coef, eval_mode=not train)
                tn(prec, d, h_in, grads[lin.weight], grads[lin.bias], N, K, q_prologue=pro_in,
                   p_prologue=(y, st.mean, st.rstd, coef), tag=f"{self.name}.L{i}.dW")
                continue
            done = False
            if _FOLD_BN_FINALIZE:
                try:
                    ops.bn_bwd_finalize_apply(d, y, B, N, st.mean, st.rstd, *fin)
                    done = True
                except RuntimeError:
                    pass
            if not done:
                ops.bn_bwd_finalize(B, N, stats, bn.weight, st.rstd, grads[bn.weight], grads[bn.bias], coef, eval_mode=not train)
                ops.bn_bwd_apply(d, y, N, st.mean, st.rstd, coef)
            tn(prec, d, h_in, grads[lin.weight], grads[lin.bias], N, K, q_prologue=pro_in, tag=f"{self.name}.L{i}.dW")
            src, src_wt, src_n, src_k = d, pl.wt, K, N


def _span(t, n):
    """View of n contiguous fp32 elements starting at tensor t (inside the gradient arena)."""
    return torch.as_strided(t, (n,), (1,))


class EmbedEncoder:
    """EncoderC (encoders.py:49-61): Embedding(S,E) + two heads == a [S][2L] table + gather."""

    def __init__(self, embedding, fc_mu, fc_logvar):
        self.embedding, self.fc_mu, self.fc_logvar = embedding, fc_mu, fc_logvar
        self.latent = fc_mu.out_features

    def prepare(self, prec, device):
        return []

    def params(self):
        return [self.embedding.weight, self.fc_mu.weight, self.fc_mu.bias, self.fc_logvar.weight, self.fc_logvar.bias]

    def table(self):
        emb = self.embedding.weight
        t = torch.empty(emb.shape[0], 2 * self.latent, dtype=torch.float32, device=emb.device)
        ops.embed_table_fwd(emb, self.fc_mu.weight, self.fc_mu.bias, self.fc_logvar.weight, self.fc_logvar.bias, t)
        return t

    def backward(self, d_table, grads):
        ops.embed_table_bwd(self.embedding.weight, self.fc_mu.weight, self.fc_logvar.weight, d_table,
                            grads[self.embedding.weight], grads[self.fc_mu.weight], grads[self.fc_mu.bias],
                            grads[self.fc_logvar.weight], grads[self.fc_logvar.bias])


class DecoderMLP:
    """DecoderA/B/C (src/models/decoders.py:8-50): Linear(+ReLU) chain, optional final Sigmoid.
    Hidden activations are stored in the activation type, the output in fp32 (it is returned
    to the caller)."""

    def __init__(self, linears, final_sigmoid, name="dec"):
        self.linears, self.final_sigmoid = list(linears), final_sigmoid
        self.out_dim = self.linears[-1].out_features
        self.name = name

    def prepare(self, prec, device):
        self.pl = [ops.PreparedLinear([l.weight], [l.bias], prec, device) for l in self.linears]
        return self.pl

    def params(self):
        out = []
        for l in self.linears:
            out += [l.weight, l.bias]
        return out

    def can_fuse_loss(self, prec):
        """The reconstruction loss can run inside the last layer's GEMM (EPI_LOSS_*): bf16 mode, a hidden layer in front of it
        (bf16 A operand) and more than one K step."""
        return prec == ops.PREC_BF16 and len(self.pl) >= 2 and self.pl[-1].K > 64

    def forward(self, prec, z, fused_loss=None, first=None):
        """fused_loss = (target fp32 [B][N], one-element float64 accumulator): the last layer does not store its output; its GEMM
        epilogue adds the reconstruction loss (sum-MSE, or sum-BCE behind the final Sigmoid: losses.py:31,34) to the accumulator and
        writes the bf16 gradient w.r.t. the pre-activation output.  Returns (gradient, acts) then instead of (output, acts).
        first: output of layer 0 computed elsewhere (VAEGraph's merged first layers of all decoders), a [B][N0] column slice."""
        B, dev = z.shape[0], z.device
        adt = act_dtype(prec)
        acts = [z]
        h = z
        if first is not None:
            acts.append(first)
            h = first
        for j, pl in enumerate(self.pl):
            if j == 0 and first is not None:
                continue
            last = j == len(self.pl) - 1
            if last and fused_loss is not None:
                target, acc = fused_loss
                if tuple(target.shape) != (B, pl.N) or target.dtype != torch.float32 or target.stride(1) != 1:
                    raise ValueError(f"fused reconstruction loss: target must be fp32 [{B}, {pl.N}], got {tuple(target.shape)} {target.dtype}")
                out = torch.empty(B, ceil_to(pl.N, 8), dtype=adt, device=dev)
                ops.gemm_nt(prec, h, pl.w, pl.N, pl.K, out, bias=pl.bias, epilogue=ops.EPI_LOSS_BCE_LOGIT if self.final_sigmoid else ops.EPI_LOSS_MSE,
                            h=target, loss_sum=acc, tag=f"{self.name}.L{j}.fwd")
            elif last:
                out = torch.empty(B, pl.N, dtype=torch.float32, device=dev)
                ops.gemm_nt(prec, h, pl.w, pl.N, pl.K, out, bias=pl.bias, act=ACT_SIGMOID if self.final_sigmoid else ACT_NONE, tag=f"{self.name}.L{j}.fwd")
            else:
                out = torch.empty(B, ceil_to(pl.N, 8), dtype=adt, device=dev)
                ops.gemm_nt(prec, h, pl.w, pl.N, pl.K, out, bias=pl.bias, act=ACT_RELU, tag=f"{self.name}.L{j}.fwd")
                acts.append(out)
            h = out
        return h, acts

    def backward(self, prec, acts, out, g_out, g_is_logit_grad, dz, accumulate_dz, grads, tn=ops.gemm_tn, d0_out=None):
        """g_out: gradient w.r.t. the decoder output ([B][>=N], fp32 or activation type).  For a
        sigmoid decoder it is w.r.t. the pre-sigmoid logits iff g_is_logit_grad.
        d0_out: where the gradient w.r.t. layer 0's output goes (a column slice of the buffer VAEGraph multiplies with the merged
        first-layer weights of all decoders); layer 0's own dX GEMM is then skipped and dz is not touched."""
        B, dev = g_out.shape[0], g_out.device
        adt = act_dtype(prec)
        d = g_out
        if self.final_sigmoid and not g_is_logit_grad:
            dl = torch.empty(B, ceil_to(self.out_dim, 8), dtype=adt, device=dev)
            ops.sigmoid_bwd(g_out, out, dl)
            d = dl
        for j in reversed(range(len(self.pl))):
            pl, lin = self.pl[j], self.linears[j]
            tn(prec, d, acts[j], grads[lin.weight], grads[lin.bias], pl.N, pl.K, tag=f"{self.name}.L{j}.dW")
            if j > 0:
                d_prev = d0_out if (j == 1 and d0_out is not None) else torch.empty(B, ceil_to(pl.K, 8), dtype=adt, device=dev)
                ops.gemm_nt(prec, d, pl.wt, pl.K, pl.N, d_prev, epilogue=EPI_RELU_MASK, h=acts[j], tag=f"{self.name}.L{j}.dX")
                d = d_prev
            elif d0_out is None:
                ops.gemm_nt(prec, d, pl.wt, pl.K, pl.N, dz, accumulate=accumulate_dz, tag=f"{self.name}.L{j}.dX")


# --------------------------------------------------------------------------------------------
# a VAE graph: encoders -> mean-fusion -> reparameterise -> decoders
# --------------------------------------------------------------------------------------------
class VAEGraph:
    """Compute graph shared by MultiModalVAE (vae.py:18-79), RNA2DNAVAE and DNA2RNAVAE
    (directional_vae.py:12-111): any subset of {MLP encoder a, MLP encoder b, embedding
    encoder}, mean fusion, reparameterisation, and a list of decoders."""

    def __init__(self, enc_a=None, enc_b=None, enc_c=None, decoders=()):
        self.enc_a, self.enc_b, self.enc_c = enc_a, enc_b, enc_c
        self.decoders = list(decoders)
        self.blocks = [b for b in (enc_a, enc_b, enc_c) if b is not None] + self.decoders
        self.latent = (enc_a or enc_b or enc_c).latent
        self._prep = None
        self._prep_key = None
        self.dec_stem = None
        self.noise = GLOBAL_NOISE
        self.grad_sync = None          # mmvae.parallel.GradAllReduce (early/final hooks) under data parallelism
        # (Independent chains on side HIP streams -- EncoderA beside EncoderB, the small decoders beside DecoderB, the dW GEMMs beside
        # the dX chain, the noise launch beside the first GEMMs -- were measured slower every time the kernels got faster: overlapped
        # bandwidth-bound kernels only take turns and the fork / join edges cost more than they buy.  One stream.)
        # Training-step fusion (mmvae.graphs): [target or None per decoder].  The next forward then computes those decoders'
        # reconstruction losses inside their last GEMM instead of returning the reconstruction (see DecoderMLP.forward).
        self.fused_recon = None

    def _late_decoder_params(self):
        """Decoder tensors whose dW GEMM is small-output (latent / class widths: the decoders' first layers, DecoderC): they are
        computed by the grouped launch at the END of backward, so under data parallelism they travel with the encoder half of the
        gradient arena instead of forcing an early flush of that launch."""
        out = []
        for d in self.decoders:
            for l in d.linears:
                if _GROUP_TINY_DW and l.weight.numel() <= _TINY_DW_MAX:
                    out += [l.weight, l.bias]
        return out

    def param_list(self):
        """Order of the flat gradient arena: encoders, the decoders' small-output tensors, then the large decoder tensors -- the tail
        (from early_cut() on) is final when the decoders' backward is done and is all-reduced under the encoder backward."""
        out = []
        for b in self.blocks:
            if b not in self.decoders:
                out += b.params()
        late = self._late_decoder_params()
        out += late
        ids = {id(p) for p in late}
        for d in self.decoders:
            out += [p for p in d.params() if id(p) not in ids]
        return out

    def early_cut(self):
        """Index into the arena where the early all-reduce bucket starts."""
        return sum(p.numel() for b in self.blocks if b not in self.decoders for p in b.params()) + sum(p.numel() for p in self._late_decoder_params())

    def _ensure_prepared(self, prec, device):
        key = (prec, str(device)) + tuple(p.data_ptr() for p in self.param_list())
        if self._prep is None or self._prep_key != key:
            pls = []
            for b in self.blocks:
                pls += b.prepare(prec, device)
            # The first layers of all decoders read the same z (K = latent: one K step): ONE GEMM with their weights concatenated
            # along N instead of one latency-bound launch per decoder, and ONE dX GEMM (K = sum of their widths) for dL/dz in
            # backward.  Each decoder continues from / writes into its column slice (widths must keep the slices 16-byte aligned).
            self.dec_stem = None
            decs = self.decoders
            if (_MERGE_DECODER_STEM and len(decs) >= 2 and all(isinstance(d, DecoderMLP) and len(d.linears) >= 2 for d in decs)
                    and all(d.linears[0].in_features == decs[0].linears[0].in_features and d.linears[0].out_features % 8 == 0 for d in decs)):
                self.dec_stem = ops.PreparedLinear([d.linears[0].weight for d in decs], [d.linears[0].bias for d in decs], prec, device)
                pls.append(self.dec_stem)
            self._prep = ops.WeightPrep(pls, device)
            self._prep_key = key
        self._prep.run()

    def forward(self, prec, xa, xb, site, train):
        """Returns (outs(list, fp32), mu, logvar, saved)."""
        ref = xa if xa is not None else (xb if xb is not None else site)
        if not ref.is_cuda:
            raise RuntimeError(f"the MI355X path needs inputs and parameters on one CUDA/HIP device (input on {ref.device}); "
                               "there is no CPU fallback")
        with ops.pinned_stream():
            return self._forward(prec, xa, xb, site, train)

    def _forward(self, prec, xa, xb, site, train):
        ref = xa if xa is not None else (xb if xb is not None else site)
        dev, B = ref.device, ref.shape[0]
        if not ref.is_cuda or any(p.device != dev for p in self.param_list()):
            raise RuntimeError(f"the MI355X path needs inputs and parameters on one CUDA/HIP device (input on {dev}); "
                               "there is no CPU fallback")
        self._ensure_prepared(prec, dev)
        saved = {"prec": prec, "B": B, "train": train}
        heads_a = heads_b = table = None
        Ld = self.latent
        widths_a = self.enc_a.widths() if (train and xa is not None) else []
        widths_b = self.enc_b.widths() if (train and xb is not None) else []
        masks, eps = self.noise.draw(B, widths_a + widths_b, Ld, dev)       # eps is sampled in eval mode too (vae.py:73)
        # ONE memset for everything this step needs zeroed: forward BatchNorm sums, the loss accumulators, and -- when a backward will
        # follow -- the flat gradient arena with the backward's BatchNorm sums and embedding-table gradient (three fills before)
        st_all = []
        # decided by the caller (functional.run_graph) BEFORE it enters the autograd.Function, inside which grad mode is always off
        want_bwd = train and (bool(getattr(self, "_want_bwd", False)) or (torch.is_grad_enabled() and any(p.requires_grad for p in self.param_list())))
        if train:
            specs = [(2 * w, torch.float64) for w in widths_a + widths_b]
            nst = len(specs)
            if want_bwd:
                specs += [(5, torch.float64), (5, torch.float32)] + self._grad_specs(xa is not None, xb is not None, site is not None)
            packed = zeros_pack(dev, specs)
            st_all = [t.view(2, -1) for t in packed[:nst]]
            if want_bwd:
                saved["loss_ws"] = (packed[nst], packed[nst + 1])
                saved["grad_pack"] = packed[nst + 2:]
        if xa is not None:
            xa = _check_input(xa, "a", self.enc_a.in_dim)
            heads_a, saved["enc_a"] = self.enc_a.forward(prec, xa, train, masks[:len(widths_a)] if train else None,
                                                         st_all[:len(widths_a)] if train else None, want_bwd=want_bwd)
        if xb is not None:
            xb = _check_input(xb.reshape(xb.shape[0], -1), "b", self.enc_b.in_dim)     # encoders.py:44 view
            heads_b, saved["enc_b"] = self.enc_b.forward(prec, xb, train, masks[len(widths_a):] if train else None,
                                                         st_all[len(widths_a):] if train else None, want_bwd=want_bwd)
        if site is not None:
            if site.dtype != torch.int64:
                site = site.long()
            site = site.contiguous()
            table = self.enc_c.table()
            saved["site"] = site
        mu = torch.empty(B, Ld, dtype=torch.float32, device=dev)
        logvar = torch.empty(B, Ld, dtype=torch.float32, device=dev)
        z = torch.empty(B, ceil_to(Ld, 8), dtype=act_dtype(prec), device=dev)
        ops.fuse_reparam_fwd(B, Ld, heads_a, heads_b, table, site, eps, mu, logvar, z)
        # .detach(): aliases of the RETURNED tensors, so the saved state holds no reference to objects that own the
        # autograd node (tensor -> grad_fn -> ctx -> saved -> tensor would be a cycle only the cyclic GC frees,
        # i.e. every step's activations would pile up in HBM until it runs)
        saved.update(eps=eps, logvar=logvar.detach(), n_mod=(heads_a is not None) + (heads_b is not None) + (table is not None))
        outs, saved["dec"] = [None] * len(self.decoders), [None] * len(self.decoders)
        order = sorted(range(len(self.decoders)), key=lambda i: -sum(l.weight.numel() for l in self.decoders[i].linears))
        stem = self.dec_stem
        firsts = [None] * len(self.decoders)
        if stem is not None:
            H0 = torch.empty(B, stem.N, dtype=act_dtype(prec), device=dev)
            ops.gemm_nt(prec, z, stem.w, stem.N, stem.K, H0, bias=stem.bias, act=ACT_RELU, tag="Decoders.L0.fwd")
            off = 0
            for i, d in enumerate(self.decoders):
                firsts[i] = H0[:, off:off + d.linears[0].out_features]
                off += d.linears[0].out_features
        fused = None
        want = self.fused_recon
        if want is not None and any(t is not None and self.decoders[i].can_fuse_loss(prec) for i, t in enumerate(want)):
            sums, out5 = saved["loss_ws"] if "loss_ws" in saved else ops.loss_workspace(dev)
            fused = saved["fused_recon"] = {"sums": sums, "out5": out5, "g": {}, "targets": {}}
        for i in order:                                 # largest decoder first
            dec = self.decoders[i]
            tgt = want[i] if (fused is not None and want[i] is not None and dec.can_fuse_loss(prec)) else None
            if tgt is not None:
                k = 1 if dec.final_sigmoid else 0                     # sums[0] = MSE, sums[1] = BCE (mmvae_vae_loss)
                g, acts = dec.forward(prec, z, fused_loss=(tgt, fused["sums"][k:k + 1]), first=firsts[i])
                fused["g"][i], fused["targets"][i] = g, tgt
                o = torch.empty(1, dtype=torch.float32, device=dev).expand(B, dec.out_dim)     # placeholder: no storage behind it
            else:
                o, acts = dec.forward(prec, z, first=firsts[i])
            outs[i] = o
            saved["dec"][i] = (acts, o.detach())
        return outs, mu, logvar, saved

    def _grad_specs(self, has_a, has_b, has_site):
        """zeros_pack specs of what a backward needs zeroed: [flat gradient arena, BatchNorm-backward sums per BN layer, table gradient]."""
        wa = self.enc_a.widths() if (has_a and self.enc_a is not None) else []
        wb = self.enc_b.widths() if (has_b and self.enc_b is not None) else []
        n_tab = self.enc_c.embedding.weight.shape[0] * 2 * self.latent * L_.TABLE_COPIES if has_site else 0
        total = sum(p.numel() for p in self.param_list())
        return [(total, torch.float32)] + [(2 * w, torch.float64) for w in wa + wb] + [(max(n_tab, 1), torch.float32)]

    def alloc_grads(self, device, extra=(), packed=None):
        """Flat zeroed gradient arena + views per parameter (+ extra zeroed tensors from the same memset; `packed`: tensors the
        forward already zeroed with _grad_specs)."""
        params = self.param_list()
        total = sum(p.numel() for p in params)
        if packed is None:
            packed = zeros_pack(device, [(total, torch.float32)] + list(extra))
        flat = packed[0]
        views, off = {}, 0
        for p in params:
            views[p] = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        return flat, views, packed[1:]

    def backward(self, saved, g_outs, g_logit_flags, g_mu, g_lv):
        with ops.pinned_stream():
            return self._backward(saved, g_outs, g_logit_flags, g_mu, g_lv)

    def _backward(self, saved, g_outs, g_logit_flags, g_mu, g_lv):
        """g_outs[i]: gradient w.r.t. decoder i's output or None; g_mu/g_lv fp32 [B][L] or None.
        Returns (flat_arena, {param: grad view})."""
        prec, B = saved["prec"], saved["B"]
        dev = saved["eps"].device
        Ld = self.latent
        wa = self.enc_a.widths() if "enc_a" in saved else []
        wb = self.enc_b.widths() if "enc_b" in saved else []
        site = saved.get("site")
        n_tab = self.enc_c.embedding.weight.shape[0] * 2 * Ld * L_.TABLE_COPIES if site is not None else 0
        flat, grads, extra = self.alloc_grads(dev, [(2 * w, torch.float64) for w in wa + wb] + [(max(n_tab, 1), torch.float32)],
                                              packed=saved.pop("grad_pack", None))
        st_bwd = [t.view(2, -1) for t in extra[:-1]]
        dzs = []                                       # one dL/dz per decoder; summed in mmvae_fuse_reparam_bwd
        # slab workspace for the split-batch dW GEMMs (<= 64 splits of the largest weight matrix); launches that use it
        # run one after another on one stream, so a single buffer serves them all
        big = max(p.numel() for p in self.param_list())
        slab = torch.empty(min(64 * big, 1 << 25), dtype=torch.float32, device=dev)     # <= 128 MiB; too small -> that GEMM uses atomics

        # Small-output dW GEMMs (latent / class widths: encoder heads, decoder first layers, DecoderC) are latency chains when
        # launched alone (~25 us each for a few MB + a reduce launch): they are DEFERRED and run as one grouped launch + one
        # reduce per flush -- after the decoders (their gradients are then final for the early all-reduce bucket) and at the end.
        tiny = []

        def tn(prec_, p, q, dw, db, N, K, q_prologue=None, p_prologue=None, tag=None):
            if _GROUP_TINY_DW and p_prologue is None and N * K <= _TINY_DW_MAX:
                tiny.append(dict(p=p, q=q, dw=dw, db=db, N=N, K=K, q_prologue=q_prologue))
                return
            ops.gemm_tn(prec_, p, q, dw, db, N, K, q_prologue=q_prologue, p_prologue=p_prologue, slab=slab, tag=tag)

        def flush_tiny(tag):
            if tiny:
                need = ops.TN_GROUP_SPLITS * sum(t_["N"] * t_["K"] for t_ in tiny)
                ops.gemm_tn_group(prec, tiny, torch.empty(need, dtype=torch.float32, device=dev), tag=tag)
                tiny.clear()
        stem = self.dec_stem if all(g is not None for g in g_outs) else None      # every decoder must fill its slice
        D0, off = None, 0
        if stem is not None:
            D0 = torch.empty(B, stem.N, dtype=act_dtype(prec), device=dev)
        for dec, (acts, out), g, is_logit in zip(self.decoders, saved["dec"], g_outs, g_logit_flags):
            if g is None:
                continue
            if stem is not None:
                n0 = dec.linears[0].out_features
                dec.backward(prec, acts, out, g, is_logit, None, False, grads, tn, d0_out=D0[:, off:off + n0])
                off += n0
                continue
            dz = torch.empty(B, Ld, dtype=torch.float32, device=dev)
            dec.backward(prec, acts, out, g, is_logit, dz, False, grads, tn)
            dzs.append(dz)
        if stem is not None:
            dz = torch.empty(B, Ld, dtype=torch.float32, device=dev)
            ops.gemm_nt(prec, D0, stem.wt, stem.K, stem.N, dz, tag="Decoders.L0.dX")
            dzs.append(dz)
        if not dzs:
            dzs.append(torch.zeros(B, Ld, dtype=torch.float32, device=dev))
        if self.grad_sync is not None:
            # the large decoder gradients (tail of the arena) are final: start reducing them under the encoder backward.  The decoders'
            # small-output tensors sit in front of the cut: their grouped dW launch stays ONE launch at the end of backward, as on one GPU
            # (round 3; before, data parallelism flushed that launch here: an extra grouped GEMM + reduce per step).
            self.grad_sync.early(flat, self.early_cut())
        n_mod = saved["n_mod"]
        d_heads = torch.empty(B, 2 * Ld, dtype=torch.float32, device=dev)
        d_table = extra[-1][:n_tab].view(L_.TABLE_COPIES, -1, 2 * Ld) if site is not None else None
        # bf16 mode: a bf16 copy of d_heads for the heads' dX GEMMs (they round it on load anyway; plain bf16 A -> LDS-DMA kernel)
        d_heads_lp = torch.empty(B, ceil_to(2 * Ld, 8), dtype=torch.bfloat16, device=dev) if prec == PREC_BF16 else None
        ops.fuse_reparam_bwd(B, Ld, n_mod, g_mu, g_lv, dzs, saved["eps"], saved["logvar"], d_heads, d_table, site, d_heads_lp=d_heads_lp)
        if "enc_a" in saved:
            self.enc_a.backward(prec, saved["enc_a"], d_heads, grads, tn, st_bwd[:len(wa)], train=saved["train"], d_heads_lp=d_heads_lp)
        if "enc_b" in saved:
            self.enc_b.backward(prec, saved["enc_b"], d_heads, grads, tn, st_bwd[len(wa):], train=saved["train"], d_heads_lp=d_heads_lp)
        if site is not None:
            self.enc_c.backward(d_table, grads)
        flush_tiny("tiny_dW.heads")
        if self.grad_sync is not None:
            self.grad_sync.final(flat)
        return flat, grads
