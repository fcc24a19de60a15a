"""Tensor-level wrappers over the C ABI (include/mmvae_hip.h).  PyTorch is used here only as
the owner of device memory and streams; all arithmetic happens in libmmvae_hip.so."""
import ctypes as C

import os

import torch

from . import _lib as L
from ._lib import (F32, BF16, PREC_F32, PREC_BF16, PRO_NONE, PRO_BN_RELU_DROP, PRO_BN_BWD_APPLY, EPI_STORE, EPI_RELU_MASK,
                   EPI_BN_BWD, EPI_LOSS_MSE, EPI_LOSS_BCE_LOGIT, ACT_NONE, ACT_RELU, ACT_SIGMOID, TILE)

DROP_P = 0.1                      # nn.Dropout(0.1), reference src/models/encoders.py:16,34,38
BN_EPS, BN_MOMENTUM = 1e-5, 0.1   # nn.BatchNorm1d defaults, encoders.py:14,32,36


class KernelProbe:
    """Optional per-launch timer: HIP events recorded on the launch stream around tagged GEMM
    launches (bench.py uses it for the live roofline numbers).  Off (None) by default."""

    def __init__(self, only=None):
        self.records = {}          # tag -> [(start_event, end_event, meta)]
        self.only = only           # optional set of tags to time (None = all)

    def wants(self, tag):
        return tag is not None and (self.only is None or tag in self.only)

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def end(self, tag, start, meta):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        self.records.setdefault(tag, []).append((start, ev, meta))

    def summary(self):
        """tag -> dict(calls, mean_ms, meta); call after torch.cuda.synchronize()."""
        out = {}
        for tag, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _ in recs]
            out[tag] = dict(calls=len(ms), mean_ms=sum(ms) / len(ms), meta=recs[0][2])
        return out


PROBE = None


class probe_span:
    """`with probe_span(tag, bytes)`: brackets a non-GEMM launch with events when a KernelProbe is installed (bench.py's
    survey of the step); `nbytes` = algorithmic HBM bytes of the launch (operands read once + results written once)."""

    def __init__(self, tag, nbytes):
        self.tag, self.nbytes, self.t0 = tag, nbytes, None

    def __enter__(self):
        if PROBE is not None and PROBE.wants(self.tag):
            self.t0 = PROBE.begin()
        return self

    def __exit__(self, *exc):
        if self.t0 is not None:
            PROBE.end(self.tag, self.t0, dict(kind="stream", bytes=int(self.nbytes() if callable(self.nbytes) else self.nbytes)))
        return False


def ceil_to(x, m):
    return (x + m - 1) // m * m


def act_dtype(prec):
    return torch.bfloat16 if prec == PREC_BF16 else torch.float32


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _p(t):
    return None if t is None else t.data_ptr()


_STREAM_OVERRIDE = None     # raw hipStream_t handle pinned by engine code for a span of launches (torch.cuda.current_stream()
                            # costs ~10 us per query, more than building the argument struct)


def _stream():
    return _STREAM_OVERRIDE if _STREAM_OVERRIDE is not None else torch.cuda.current_stream().cuda_stream


class pinned_stream:
    """Context manager: resolve the current stream ONCE and use its handle for every launch inside."""

    def __init__(self, stream=None):
        self.handle = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream

    def __enter__(self):
        global _STREAM_OVERRIDE
        self.prev, _STREAM_OVERRIDE = _STREAM_OVERRIDE, self.handle
        return self

    def __exit__(self, *exc):
        global _STREAM_OVERRIDE
        _STREAM_OVERRIDE = self.prev
        return False


def _mat(t, name):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a 2-D tensor with unit inner stride, got {tuple(t.shape)} / {t.stride()}")
    if not t.is_cuda:
        raise ValueError(f"{name}: must live on the GPU")
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# --------------------------------------------------------------------------------------------
# prepared weights
# --------------------------------------------------------------------------------------------
class PreparedLinear:
    """MFMA operand copies of one (possibly row-concatenated) Linear: W [ceil128(N)][ceil64(K)]
    and W^T [ceil128(K)][ceil64(N)] in the compute type, plus the fp32 bias (concatenated if
    the sources are)."""

    def __init__(self, weights, biases, prec, device):
        self.N = sum(w.shape[0] for w in weights)
        self.K = weights[0].shape[1]
        self.prec = prec
        dt = act_dtype(prec)
        self.w = torch.zeros(ceil_to(self.N, TILE), ceil_to(self.K, 64), dtype=dt, device=device)
        self.wt = torch.zeros(ceil_to(self.K, TILE), ceil_to(self.N, 64), dtype=dt, device=device)
        self.srcs = list(weights)
        self.bias_srcs = list(biases)
        if len(biases) == 1:
            self.bias = biases[0]
            self._bias_cat = None
        else:
            self._bias_cat = torch.zeros(self.N, dtype=torch.float32, device=device)
            self.bias = self._bias_cat

    def items(self):
        out = []
        dtc = _dt(self.w)
        row = 0
        esz = self.w.element_size()
        for i, w in enumerate(self.srcs):
            n = w.shape[0]
            last = i == len(self.srcs) - 1
            # plain copy: rows [row, row+n) (the last item also clears the padding rows)
            rows = (self.w.shape[0] - row) if last else n
            out.append(L.PrepItem(w.data_ptr(), self.w.data_ptr() + row * self.w.stride(0) * esz, n, self.K, w.stride(0),
                                  rows, self.w.shape[1], self.w.stride(0), 0, dtc))
            # transposed copy: columns [row, row+n)
            cols = (self.wt.shape[1] - row) if last else n
            out.append(L.PrepItem(w.data_ptr(), self.wt.data_ptr() + row * esz, n, self.K, w.stride(0),
                                  self.wt.shape[0], cols, self.wt.stride(0), 1, dtc))
            if self._bias_cat is not None:
                b = self.bias_srcs[i]
                out.append(L.PrepItem(b.data_ptr(), self._bias_cat.data_ptr() + row * 4, 1, n, n, 1, n, n, 0, F32))
            row += n
        return out


class WeightPrep:
    """All PreparedLinears of a module tree, refreshed from the fp32 masters in ONE launch."""

    def __init__(self, linears, device):
        self.linears = list(linears)
        items = [it for pl in self.linears for it in pl.items()]
        self.n = len(items)
        arr = (L.PrepItem * self.n)(*items)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.table = host.to(device)
        self._keys = self._current_keys()

    def _current_keys(self):
        return tuple(w.data_ptr() for pl in self.linears for w in pl.srcs + pl.bias_srcs)

    def stale(self):
        return self._keys != self._current_keys()

    def run(self):
        with probe_span("prep_weights", lambda: sum(6 * w.numel() for pl in self.linears for w in pl.srcs)):
            L.check(L.load().mmvae_prep_weights(self.table.data_ptr(), self.n, _stream()), "mmvae_prep_weights")


# --------------------------------------------------------------------------------------------
# GEMMs
# --------------------------------------------------------------------------------------------
def can_keep_pro_out(prec, M, N, K, a, out):
    """Will mmvae_gemm_nt write `pro_out` for this problem?  Mirrors the conditions of the wave-specialised kernel's prologue variant
    (include/mmvae_hip.h, gemm_ntp.hip): the library answers MMVAE_ERR_ARG when asked for pro_out outside them."""
    return (prec == PREC_BF16 and a.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and M >= 16384 and M % 128 == 0
            and N % 128 == 0 and N <= 256 and K % 64 == 0 and 64 < K <= 512 and _ld(a) % 8 == 0 and _ld(out) % 64 == 0
            and M * _ld(a) * 2 < 2 ** 31            # one launch: the row-block path of >= 4 GiB operands is never asked for pro_out
            and a.data_ptr() % 16 == 0 and out.data_ptr() % 128 == 0 and os.environ.get("MMVAE_NO_NTP") is None
            and os.environ.get("MMVAE_NO_NTP_PRO") is None and os.environ.get("MMVAE_NO_PRO_OUT") is None)


def gemm_nt(prec, a, w_lp, N, K, out, *, bias=None, act=ACT_NONE, accumulate=False, prologue=None,
            epilogue=EPI_STORE, h=None, bn=None, bn_coef=None, bn_phase=None, stats=None, loss_sum=None, tag=None, pro_out=None,
            pro_finalize=None):
    """out[M,N] = epi( pro(a)[M,K] @ W[N,K]^T ).  prologue = (scale, shift, mask|None, inv_keep);
    bn = (scale, shift, mean, rstd, mask|None, inv_keep) for EPI_BN_BWD (out=None, stats given:
    statistics phase; out and bn_coef given: apply phase).  stats: zeroed float64 [2][N] accumulator.
    EPI_LOSS_MSE / EPI_LOSS_BCE_LOGIT: h = fp32 target, out = bf16 gradient, loss_sum = one-element float64 view that is added to."""
    _mat(a, "a"); _mat(w_lp, "w")
    if out is not None:
        _mat(out, "out")
    M = a.shape[0]
    g = L.GemmNtArgs()
    g.prec, g.M, g.N, g.K = prec, M, N, K
    g.a, g.a_dtype, g.lda = a.data_ptr(), _dt(a), _ld(a)
    if prologue is not None:
        sc, sh, mask, inv_keep = prologue
        g.prologue = PRO_BN_RELU_DROP
        g.pro_scale, g.pro_shift, g.pro_mask = sc.data_ptr(), sh.data_ptr(), _p(mask)
        g.ld_pro_mask = _ld(mask) if mask is not None else 0
        g.pro_inv_keep = inv_keep
        if pro_out is not None:                 # the operand after the prologue (bf16 [M][>= K]); see can_keep_pro_out()
            g.pro_out, g.ld_pro_out = pro_out.data_ptr(), _ld(pro_out)
        if pro_finalize is not None:            # BnFinalizeArgs of the layer that produced `a` (bn_finalize_args()): finalised inside this launch
            g.pro_finalize = C.addressof(pro_finalize)
    g.w, g.ldw = w_lp.data_ptr(), w_lp.stride(0)
    g.epilogue = epilogue
    if out is not None:
        g.c, g.c_dtype, g.ldc = out.data_ptr(), _dt(out), _ld(out)
    g.bias, g.act, g.accumulate = _p(bias), act, int(accumulate)
    if h is not None:
        g.h, g.ldh = h.data_ptr(), _ld(h)
    if bn is not None:
        sc, sh, mean, rstd, mask, inv_keep = bn
        g.bn_scale, g.bn_shift, g.bn_mean, g.bn_rstd = sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        g.epi_mask, g.ld_epi_mask, g.epi_inv_keep = _p(mask), (_ld(mask) if mask is not None else 0), inv_keep
        g.bn_coef = _p(bn_coef)
        g.bn_phase = bn_phase if bn_phase is not None else int(bn_coef is not None)
    if stats is not None:
        assert stats.dtype == torch.float64 and stats.shape[0] == 2 and stats.shape[1] >= N and stats.stride(1) == 1
        g.stat1, g.stat2 = stats[0].data_ptr(), stats[1].data_ptr()
    if loss_sum is not None:
        assert loss_sum.dtype == torch.float64 and h is not None and h.dtype == torch.float32
        g.stat1 = loss_sum.data_ptr()
    t0 = PROBE.begin() if (PROBE is not None and PROBE.wants(tag)) else None
    L.check(L.load().mmvae_gemm_nt(C.byref(g), _stream()), "mmvae_gemm_nt")
    if t0 is not None:
        PROBE.end(tag, t0, dict(kind="nt", M=M, N=N, K=K, a_bytes=a.element_size(),
                                c_bytes=0 if out is None else out.element_size(),
                                pro=prologue is not None, pro_mask=prologue is not None and prologue[2] is not None,
                                epi=epilogue, epi_mask=bn is not None and bn[4] is not None, act_bytes=2 if prec == PREC_BF16 else 4))
    return out


def _tn_args(prec, p, q, dw, db, N, K, q_prologue, p_prologue, nsplit, slab):
    _mat(p, "p"); _mat(q, "q")
    g = L.GemmTnArgs()
    g.prec, g.M, g.N, g.K = prec, p.shape[0], N, K
    g.p, g.p_dtype, g.ldp = p.data_ptr(), _dt(p), _ld(p)
    g.q, g.q_dtype, g.ldq = q.data_ptr(), _dt(q), _ld(q)
    if q_prologue is not None:
        sc, sh, mask, inv_keep = q_prologue
        g.q_prologue = PRO_BN_RELU_DROP
        g.pro_scale, g.pro_shift, g.pro_mask = sc.data_ptr(), sh.data_ptr(), _p(mask)
        g.ld_pro_mask = _ld(mask) if mask is not None else 0
        g.pro_inv_keep = inv_keep
    if p_prologue is not None:
        py, mean, rstd, coef = p_prologue[:4]
        _mat(py, "p_y")
        g.p_prologue = PRO_BN_BWD_APPLY
        g.p_y, g.ld_py, g.p_mean, g.p_rstd = py.data_ptr(), _ld(py), mean.data_ptr(), rstd.data_ptr()
        if coef is not None:
            assert py.dtype == p.dtype and coef.shape[0] == 3 and coef.shape[1] == N and coef.is_contiguous()
            g.p_coef = coef.data_ptr()
        else:
            # mmvae_bn_bwd_finalize folded into the GEMM: p_prologue = (y, mean, rstd, None, (stats f64 [2][N], gamma, dgamma, dbeta, eval_mode))
            stats, gamma, dgamma, dbeta, eval_mode = p_prologue[4]
            assert py.dtype == p.dtype and stats.dtype == torch.float64 and stats.shape[0] == 2 and stats.stride(1) == 1
            g.p_sum_d, g.p_sum_dx, g.p_gamma = stats[0].data_ptr(), stats[1].data_ptr(), gamma.data_ptr()
            g.p_dgamma, g.p_dbeta, g.p_eval_mode = dgamma.data_ptr(), dbeta.data_ptr(), int(eval_mode)
    assert dw.dtype == torch.float32 and dw.is_contiguous()
    g.dw, g.lddw, g.db = dw.data_ptr(), K, _p(db)
    g.nsplit = nsplit
    if slab is not None:
        g.slab, g.slab_elems = slab.data_ptr(), slab.numel()
    return g


TN_GROUP_SPLITS = 256        # MMVAE_TN_GROUP_SPLITS: most batch splits a problem of a grouped launch can get (slab sizing)


def gemm_tn_group(prec, problems, slab, tag="tiny_dW.group"):
    """problems: list of dicts(p, q, dw, db, N, K, q_prologue=None): SMALL-output dW GEMMs (latent / class widths) launched as
    ONE grouped GEMM + ONE reduce.  `slab`: fp32 workspace carved into one region per problem."""
    lib = L.load()
    for i in range(0, len(problems), L.TN_GROUP_MAX):
        chunk = problems[i:i + L.TN_GROUP_MAX]
        arr = (L.GemmTnArgs * len(chunk))()
        off, nbytes = 0, 0
        for j, pr in enumerate(chunk):
            need = TN_GROUP_SPLITS * pr["N"] * pr["K"]
            if off + need > slab.numel():
                raise RuntimeError("gemm_tn_group: slab workspace too small")
            arr[j] = _tn_args(prec, pr["p"], pr["q"], pr["dw"], pr["db"], pr["N"], pr["K"], pr.get("q_prologue"), None, 0, slab[off:off + need])
            off += need
            nbytes += pr["p"].shape[0] * (pr["N"] * pr["p"].element_size() + pr["K"] * pr["q"].element_size()) + 4 * pr["N"] * pr["K"]
        with probe_span(tag if i == 0 else f"{tag}.{i}", nbytes):
            status = lib.mmvae_gemm_tn_group(C.cast(arr, C.c_void_p), len(chunk), _stream())
            if status == -1:
                # a problem outside the grouped kernel's operand combinations (odd widths / alignments): the entry point checks
                # every problem before it launches anything, so the same argument records go through the one-problem entry
                for j in range(len(chunk)):
                    L.check(lib.mmvae_gemm_tn(C.byref(arr[j]), _stream()), "mmvae_gemm_tn")
            else:
                L.check(status, "mmvae_gemm_tn_group")


def gemm_tn(prec, p, q, dw, db, N, K, *, q_prologue=None, p_prologue=None, nsplit=0, slab=None, tag=None):
    """dw[N,K] += pro_p(p)[M,N]^T @ pro(q)[M,K] ; db[N] += colsum(pro_p(p)).  dw/db fp32, pre-zeroed.
    p_prologue = (y, mean, rstd, coef): the BatchNorm-backward correction of mmvae_bn_bwd_apply applied on the load of p."""
    g = _tn_args(prec, p, q, dw, db, N, K, q_prologue, p_prologue, nsplit, slab)
    t0 = PROBE.begin() if (PROBE is not None and PROBE.wants(tag)) else None
    L.check(L.load().mmvae_gemm_tn(C.byref(g), _stream()), "mmvae_gemm_tn")
    if t0 is not None:
        PROBE.end(tag, t0, dict(kind="tn", M=p.shape[0], N=N, K=K, p_bytes=p.element_size() * (2 if p_prologue is not None else 1),
                                q_bytes=q.element_size(), pro_mask=q_prologue is not None and q_prologue[2] is not None))


# --------------------------------------------------------------------------------------------
# BatchNorm pieces
# --------------------------------------------------------------------------------------------
def bn_finalize_args(M, N, stats, gamma, beta, running_mean, running_var, nbt, mean, rstd, scale, shift,
                     eps=BN_EPS, momentum=BN_MOMENTUM):
    """The argument struct of mmvae_bn_finalize: launched on its own (bn_finalize) or handed to the consumer GEMM (gemm_nt(pro_finalize=))."""
    if M < 2:
        # same failure mode as torch.nn.BatchNorm1d in training mode
        raise ValueError(f"Expected more than 1 value per channel when training, got input size [{M}, {N}]")
    return L.BnFinalizeArgs(M, N, stats[0].data_ptr(), stats[1].data_ptr(),
                            gamma.data_ptr(), beta.data_ptr(), eps, momentum, _p(running_mean), _p(running_var), _p(nbt),
                            mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())


def bn_finalize(M, N, stats, gamma, beta, running_mean, running_var, nbt, mean, rstd, scale, shift,
                eps=BN_EPS, momentum=BN_MOMENTUM, args=None):
    a = args if args is not None else bn_finalize_args(M, N, stats, gamma, beta, running_mean, running_var, nbt, mean, rstd, scale, shift, eps, momentum)
    L.check(L.load().mmvae_bn_finalize(C.byref(a), _stream()), "mmvae_bn_finalize")


def bn_eval_coeffs(gamma, beta, running_mean, running_var, scale, shift, eps=BN_EPS, mean=None, rstd=None):
    L.check(L.load().mmvae_bn_eval_coeffs(gamma.numel(), gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(),
                                          running_var.data_ptr(), eps, scale.data_ptr(), shift.data_ptr(), _p(mean), _p(rstd), _stream()),
            "mmvae_bn_eval_coeffs")


def bn_bwd_finalize(M, N, stats, gamma, rstd, dgamma, dbeta, coef, eval_mode=False):
    a = L.BnBwdFinalizeArgs(M, N, stats[0].data_ptr(), stats[1].data_ptr(),
                            gamma.data_ptr(), rstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), coef.data_ptr(), int(eval_mode))
    L.check(L.load().mmvae_bn_bwd_finalize(C.byref(a), _stream()), "mmvae_bn_bwd_finalize")


def bn_bwd_finalize_apply(d, y, M, N, mean, rstd, stats, gamma, dgamma, dbeta, eval_mode=False):
    """mmvae_bn_bwd_finalize + mmvae_bn_bwd_apply in one launch (the hidden widths of the model; raises for others)."""
    with probe_span(f"bn_bwd_apply.N{N}", 3 * d.shape[0] * N * d.element_size()):
        L.check(L.load().mmvae_bn_bwd_finalize_apply(_dt(d), M, N, d.data_ptr(), _ld(d), y.data_ptr(), _ld(y), mean.data_ptr(), rstd.data_ptr(),
                                                     stats[0].data_ptr(), stats[1].data_ptr(), gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                     int(eval_mode), _stream()), "mmvae_bn_bwd_finalize_apply")


def bn_bwd_apply(d, y, N, mean, rstd, coef):
    with probe_span(f"bn_bwd_apply.N{N}", 3 * d.shape[0] * N * d.element_size()):
        L.check(L.load().mmvae_bn_bwd_apply(_dt(d), d.shape[0], N, d.data_ptr(), _ld(d), y.data_ptr(), _ld(y),
                                            mean.data_ptr(), rstd.data_ptr(), coef.data_ptr(), _stream()), "mmvae_bn_bwd_apply")


# --------------------------------------------------------------------------------------------
# EncoderC table, fusion, loss, noise, optimiser
# --------------------------------------------------------------------------------------------
def embed_table_fwd(emb, w_mu, b_mu, w_lv, b_lv, table):
    S, E = emb.shape
    L.check(L.load().mmvae_embed_table_fwd(S, E, w_mu.shape[0], emb.data_ptr(), w_mu.data_ptr(), b_mu.data_ptr(),
                                           w_lv.data_ptr(), b_lv.data_ptr(), table.data_ptr(), _stream()), "mmvae_embed_table_fwd")


def embed_table_bwd(emb, w_mu, w_lv, d_table, d_emb, d_w_mu, d_b_mu, d_w_lv, d_b_lv):
    """d_table: [S][2L] or [copies][S][2L] (the copies mmvae_fuse_reparam_bwd scattered into; summed here)."""
    S, E = emb.shape
    copies = d_table.shape[0] if d_table.dim() == 3 else 1
    L.check(L.load().mmvae_embed_table_bwd(S, E, w_mu.shape[0], emb.data_ptr(), w_mu.data_ptr(), w_lv.data_ptr(),
                                           d_table.data_ptr(), copies, d_emb.data_ptr(), d_w_mu.data_ptr(), d_b_mu.data_ptr(),
                                           d_w_lv.data_ptr(), d_b_lv.data_ptr(), _stream()), "mmvae_embed_table_bwd")


def fuse_reparam_fwd(B, Ld, heads_a, heads_b, table, site, eps, mu, logvar, z):
    n_mod = (heads_a is not None) + (heads_b is not None) + (table is not None)
    hd = heads_a if heads_a is not None else heads_b
    a = L.FuseFwdArgs(B, Ld, n_mod, _p(heads_a), _p(heads_b), _ld(hd) if hd is not None else 0,
                      _p(table), _p(site), table.shape[0] if table is not None else 0,
                      eps.data_ptr(), mu.data_ptr(), logvar.data_ptr(), z.data_ptr(), _dt(z), _ld(z))
    with probe_span("fuse_reparam_fwd", B * (8 * Ld * (n_mod - (table is not None)) + 8 * (table is not None) + 12 * Ld + z.element_size() * _ld(z))):
        L.check(L.load().mmvae_fuse_reparam_fwd(C.byref(a), _stream()), "mmvae_fuse_reparam_fwd")


def fuse_reparam_bwd(B, Ld, n_mod, g_mu, g_lv, dzs, eps, logvar, d_heads, d_table, site, d_heads_lp=None):
    """dzs: 1..3 fp32 (B, Ld) tensors with one leading dimension (dL/dz of each decoder); they are summed.
    d_table: zeroed [S][2L] or [copies][S][2L] (workgroups spread their scatter-adds over the copies)."""
    dzs = list(dzs) + [None] * (3 - len(dzs))
    copies = d_table.shape[0] if (d_table is not None and d_table.dim() == 3) else 1
    a = L.FuseBwdArgs(B, Ld, n_mod, _p(g_mu), _p(g_lv), dzs[0].data_ptr(), _p(dzs[1]), _p(dzs[2]), _ld(dzs[0]), eps.data_ptr(), logvar.data_ptr(),
                      d_heads.data_ptr(), _ld(d_heads), _p(d_table), _p(site), d_table.shape[-2] if d_table is not None else 0,
                      _p(d_heads_lp), _ld(d_heads_lp) if d_heads_lp is not None else 0, copies)
    n_dz = sum(1 for d in dzs if d is not None)
    with probe_span("fuse_reparam_bwd", B * Ld * 4 * (n_dz + (g_mu is not None) + (g_lv is not None) + 2 + 2)):
        L.check(L.load().mmvae_fuse_reparam_bwd(C.byref(a), _stream()), "mmvae_fuse_reparam_bwd")


def loss_workspace(device):
    """-> (sums float64[5] zeroed, out5 float32[5]): the accumulators of mmvae_vae_loss and the tuple mmvae_loss_finalize makes of them."""
    buf = torch.zeros(5 * 8 + 5 * 4 + 4, dtype=torch.uint8, device=device)
    return buf[:40].view(torch.float64), buf[40:60].view(torch.float32)


def vae_loss(B, *, recon_a=None, a=None, recon_b=None, b=None, logits=None, site=None, class_weights=None,
             mu=None, logvar=None, beta=1e-3, gamma=1.0, sums=None, g_a=None, g_b=None, grad_b_wrt_logit=False,
             g_c=None, g_mu=None, g_lv=None, beta_gamma_dev=None):
    """sums: zeroed float64[5] (four loss sums + the count of labels outside [0, S)).  beta_gamma_dev: optional device
    float32[2] = {beta, gamma} that overrides the by-value hyper-parameters (hipGraph replays follow the beta warm-up)."""
    x = L.LossArgs()
    x.B = B
    if recon_a is not None:
        x.A, x.recon_a, x.a, x.ld_ra, x.ld_a = recon_a.shape[1], recon_a.data_ptr(), a.data_ptr(), _ld(recon_a), _ld(a)
    if recon_b is not None:
        x.D, x.recon_b, x.b, x.ld_rb, x.ld_b = recon_b.shape[1], recon_b.data_ptr(), b.data_ptr(), _ld(recon_b), _ld(b)
    if logits is not None:
        x.S, x.logits, x.ld_logits, x.site, x.class_weights = logits.shape[1], logits.data_ptr(), _ld(logits), site.data_ptr(), _p(class_weights)
    if mu is not None:
        x.L, x.mu, x.logvar = mu.shape[1], mu.data_ptr(), logvar.data_ptr()
    x.beta, x.gamma, x.sums = beta, gamma, sums.data_ptr()
    if g_a is not None:
        x.g_a, x.g_a_dtype, x.ld_ga = g_a.data_ptr(), _dt(g_a), _ld(g_a)
    if g_b is not None:
        x.g_b, x.g_b_dtype, x.ld_gb, x.grad_b_wrt_logit = g_b.data_ptr(), _dt(g_b), _ld(g_b), int(grad_b_wrt_logit)
    if g_c is not None:
        x.g_c, x.ld_gc = g_c.data_ptr(), _ld(g_c)
    x.g_mu, x.g_lv = _p(g_mu), _p(g_lv)
    x.beta_gamma_dev = _p(beta_gamma_dev)
    assert sums.dtype == torch.float64 and sums.numel() >= 5

    def nbytes():
        n = 0
        for pred, g in ((recon_a, g_a), (recon_b, g_b), (logits, g_c)):
            if pred is not None:
                n += 2 * B * pred.shape[1] * 4 + (0 if g is None else B * pred.shape[1] * g.element_size())
        if mu is not None:
            n += B * mu.shape[1] * 4 * (2 + (g_mu is not None) + (g_lv is not None))
        return n
    with probe_span("vae_loss", nbytes):
        L.check(L.load().mmvae_vae_loss(C.byref(x), _stream()), "mmvae_vae_loss")


def loss_finalize(sums, beta, gamma, out5, beta_gamma_dev=None):
    assert out5.numel() >= 5
    L.check(L.load().mmvae_loss_finalize(sums.data_ptr(), beta, gamma, _p(beta_gamma_dev), out5.data_ptr(), _stream()), "mmvae_loss_finalize")


def sigmoid_bwd(g, p, out):
    L.check(L.load().mmvae_sigmoid_bwd(g.shape[0], g.shape[1], g.data_ptr(), _ld(g), p.data_ptr(), _ld(p), out.data_ptr(),
                                       _dt(out), _ld(out), _stream()), "mmvae_sigmoid_bwd")


def scale_many(tensors, scale):
    """x *= *scale (device scalar) unless it is 1, for every tensor of the list, in one launch per 8 tensors."""
    ts = [t for t in tensors if t is not None]
    for i in range(0, len(ts), 8):
        chunk = ts[i:i + 8]
        items = (L.ScaleItem * len(chunk))(*[L.ScaleItem(t.data_ptr(), t.numel(), _dt(t), 0) for t in chunk])
        L.check(L.load().mmvae_scale_many(items, len(chunk), scale.data_ptr(), _stream()), "mmvae_scale_many")


def scale_if_needed(x, scale):
    L.check(L.load().mmvae_scale_if_needed(x.data_ptr(), _dt(x), x.numel(), scale.data_ptr(), _stream()), "mmvae_scale_if_needed")


def noise(mask, eps, keep_prob, seed, offset, offset_dev=None, advance=False):
    """Fill `mask` (uint8, any shape, contiguous; may be None) and `eps` (fp32; may be None) from the Philox stream
    (seed, offset [+ *offset_dev]).  Returns the number of counter values consumed.  advance=True: offset_dev is an
    int64[CTR_COPIES] tensor of identical copies of the counter and the launch itself moves them past what it consumed."""
    n_mask = 0 if mask is None else mask.numel()
    n_eps = 0 if eps is None else eps.numel()
    if advance:
        assert offset_dev is not None and offset_dev.numel() == L.CTR_COPIES and offset_dev.dtype == torch.int64
    with probe_span("noise", n_mask + 4 * n_eps):
        L.check(L.load().mmvae_noise(_p(mask), n_mask, keep_prob, _p(eps), n_eps, seed, offset, _p(offset_dev), int(advance), _stream()), "mmvae_noise")
    return (n_mask + 15) // 16 * 4 + (n_eps + 3) // 4


def dropout_mask(mask, keep_prob, seed, offset):
    return noise(mask, None, keep_prob, seed, offset)


def randn(out, seed, offset):
    return noise(None, out, 1.0, seed, offset)


def counter_add(counter, inc):
    L.check(L.load().mmvae_counter_add(counter.data_ptr(), inc, _stream()), "mmvae_counter_add")


def gather_rows(pairs, idx, src_rows):
    """pairs: [(src (N, ...) row-major, dst (B, ...))]: dst[i] = src[idx[i]] for every pair in ONE launch (idx: int64 (B,) on the
    device).  The minibatch assembly of a device-resident dataset (reference: Dataset.__getitem__ + default collate per sample)."""
    items = (L.GatherItem * len(pairs))()
    B = idx.shape[0]
    nbytes = 0
    for j, (src, dst) in enumerate(pairs):
        if src.dtype != dst.dtype or src.shape[1:] != dst.shape[1:] or dst.shape[0] != B or src.shape[0] != src_rows:
            raise ValueError(f"gather_rows: pair {j}: {tuple(src.shape)} {src.dtype} -> {tuple(dst.shape)} {dst.dtype}")
        if not (src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous()):
            raise ValueError("gather_rows needs contiguous device tensors")
        row = src[0].numel() * src.element_size()
        items[j] = L.GatherItem(src.data_ptr(), dst.data_ptr(), row, row, row, 0)
        nbytes += 2 * B * row
    if idx.dtype != torch.int64 or not idx.is_cuda or not idx.is_contiguous():
        raise ValueError("gather_rows: idx must be a contiguous int64 device tensor")
    with probe_span("gather_rows", nbytes):
        L.check(L.load().mmvae_gather_rows(C.cast(items, C.c_void_p), len(pairs), idx.data_ptr(), B, src_rows, _stream()), "mmvae_gather_rows")


def adamw_step(items, lr, b1, b2, eps, wd, bc1, bc2, maximize=False, step_dev=None, lr_dev=None):
    """items: ctypes array of AdamWItem in host memory (device pointers inside).  step_dev: int64[CTR_COPIES] tensor of
    identical copies of the step count: bias corrections from the device counter, which the launch itself increments
    (<= 64 tensors; beyond that the copies are advanced by a fill after the launches)."""
    tick = step_dev is not None and len(items) <= 64
    if step_dev is not None:
        assert step_dev.numel() == L.CTR_COPIES and step_dev.dtype == torch.int64
    with probe_span("adamw", lambda: 28 * sum(it.n for it in items)):
        L.check(L.load().mmvae_adamw_step(C.cast(items, C.c_void_p), len(items), lr, b1, b2, eps, wd, bc1, bc2, int(maximize),
                                          _p(step_dev), int(tick), _p(lr_dev), _stream()), "mmvae_adamw_step")
    if step_dev is not None and not tick:
        step_dev.add_(1)
