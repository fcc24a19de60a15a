"""GPU parity of the two MFMA GEMM kernels (through the C ABI) against float64 matmuls.

bf16 mode: operands are rounded to bf16 on the host first, so both sides multiply the SAME
values and only the f32 accumulation order differs -> tolerance 2e-5 * sqrt(K) * scale.
f32 mode: v_mfma_f32_16x16x4_f32 is an exact fma chain -> same tolerance.
Outputs stored as bf16 add one bf16 rounding (2^-9 relative)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmvae import ops  # noqa: E402
from mmvae.ops import PREC_BF16, PREC_F32  # noqa: E402

DEV = "cuda"


def _round(x, prec):
    return x.to(torch.bfloat16).to(torch.float32) if prec == PREC_BF16 else x


def _prep(W, b, prec):
    pl = ops.PreparedLinear([W], [b], prec, DEV)
    ops.WeightPrep([pl], DEV).run()
    return pl


def _tol(K, scale, out_bf16=False):
    return 2e-5 * np.sqrt(K) * scale + (scale * 2.0 ** -8 if out_bf16 else 0.0)


@pytest.mark.parametrize("prec", [PREC_F32, PREC_BF16])
@pytest.mark.parametrize("M,N,K,a_bf16", [(300, 128, 782, False), (257, 512, 572, False), (128, 40, 77, False),
                                          (1000, 600, 256, True), (64, 24, 64, True), (31, 130, 20, True)])
def test_nt_store(prec, M, N, K, a_bf16):
    if prec == PREC_F32 and a_bf16:
        pytest.skip("bf16 activations only exist in bf16 mode")
    g = torch.Generator().manual_seed(M * 7 + N)
    A = _round(torch.randn(M, K, generator=g), prec)
    W = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    Wd, bd = W.to(DEV), b.to(DEV)
    pl = _prep(Wd, bd, prec)
    np.testing.assert_array_equal(pl.w[:N, :K].float().cpu().numpy(), _round(W, prec).numpy())
    np.testing.assert_array_equal(pl.wt[:K, :N].float().cpu().numpy(), _round(W, prec).t().numpy())
    assert float(pl.w.float().abs().sum()) == pytest.approx(float(_round(W, prec).abs().sum()), rel=1e-5)
    if a_bf16:
        Ad = torch.zeros(M, ops.ceil_to(K, 8), dtype=torch.bfloat16, device=DEV)
        Ad[:, :K] = A.to(DEV)
        Ad[:, K:] = float("nan")          # a caller's pad elements may hold anything: the kernel masks a row's last chunk when K % 8 != 0
    else:
        Ad = A.to(DEV)
    ref = A.double() @ _round(W, prec).double().t() + b.double()
    for out_dt in ([torch.float32] if prec == PREC_F32 else [torch.float32, torch.bfloat16]):
        for act in (ops.ACT_NONE, ops.ACT_RELU, ops.ACT_SIGMOID):
            out = torch.full((M, ops.ceil_to(N, 8)), 7.0, dtype=out_dt, device=DEV)
            st = torch.zeros(2, N, dtype=torch.float64, device=DEV)
            ops.gemm_nt(prec, Ad, pl.w, N, K, out, bias=pl.bias, act=act, stats=st)
            r = ref if act == 0 else (ref.clamp_min(0) if act == 1 else torch.sigmoid(ref))
            got = out[:, :N].float().cpu().double()
            tol = _tol(K, float(r.abs().max()), out_dt == torch.bfloat16)
            assert float((got - r).abs().max()) <= tol, (act, out_dt)
            assert torch.all((out[:, N:].float() == 7.0) | (out[:, N:].float() == 0.0))   # pad columns: untouched or zeroed
            np.testing.assert_allclose(st[0].cpu(), got.sum(0), rtol=1e-4, atol=1e-2)
            np.testing.assert_allclose(st[1].cpu(), (got ** 2).sum(0), rtol=1e-4, atol=1e-2)
    # accumulate
    out = torch.ones(M, N, device=DEV)
    ops.gemm_nt(prec, Ad, pl.w, N, K, out, bias=pl.bias, accumulate=True)
    assert float((out.cpu().double() - (ref + 1.0)).abs().max()) <= _tol(K, float(ref.abs().max()))


@pytest.mark.parametrize("prec", [PREC_F32, PREC_BF16])
@pytest.mark.parametrize("M,N,K", [(1000, 40, 128), (4096, 512, 572), (333, 128, 782), (2048, 782, 128), (130, 24, 64)])
def test_tn(prec, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    lp = prec == PREC_BF16
    P = _round(torch.randn(M, N, generator=g), prec)
    Q = _round(torch.randn(M, K, generator=g), prec)
    ref = P.double().t() @ Q.double()
    refb = P.double().sum(0)
    variants = [("f32", "f32")] + ([("bf16", "f32"), ("bf16", "bf16"), ("f32", "bf16")] if lp else [])
    for pk, qk in variants:
        def mk(X, kind):
            if kind == "f32":
                return X.to(DEV)
            t = torch.full((M, ops.ceil_to(X.shape[1], 8)), 7.0, dtype=torch.bfloat16, device=DEV)
            t[:, :X.shape[1]] = X.to(DEV)
            return t
        Pd, Qd = mk(P, pk), mk(Q, qk)
        for nsplit in (0, 1, 3):
            dw = torch.zeros(N, K, device=DEV)
            db = torch.zeros(N, device=DEV)
            ops.gemm_tn(prec, Pd, Qd, dw, db, N, K, nsplit=nsplit)
            tol = 2e-5 * np.sqrt(M) * float(ref.abs().max())
            assert float((dw.cpu().double() - ref).abs().max()) <= tol, (pk, qk, nsplit)
            assert float((db.cpu().double() - refb).abs().max()) <= 2e-5 * np.sqrt(M) * float(refb.abs().max()) + 1e-4
    # accumulation into non-zero dw
    dw = torch.ones(N, K, device=DEV); db = torch.ones(N, device=DEV)
    ops.gemm_tn(prec, P.to(DEV), Q.to(DEV), dw, db, N, K)
    assert float((dw.cpu().double() - ref - 1).abs().max()) <= 2e-5 * np.sqrt(M) * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K", [(8192, 512, 256), (16384, 500, 250), (8192, 782, 128), (12288, 256, 512)])
def test_tn_two_wave_groups(M, N, K):
    """Plain bf16 x bf16 dW problems whose automatic split fills the chip with ONE 8-wave workgroup per CU (8 or 7 output tiles,
    batch >= 8192) run gemm_tn_kernel<.., DMA, NG = 2>: two wave groups, 4-buffer DMA ring, partial tiles added through LDS.
    Against the fp64 product of the same bf16 operands, and against the 4-wave form (an explicit split count selects it)."""
    g = torch.Generator().manual_seed(M + N + K)
    P = _round(torch.randn(M, N, generator=g), PREC_BF16)
    Q = _round(torch.randn(M, K, generator=g), PREC_BF16)
    ref = P.double().t() @ Q.double()
    refb = P.double().sum(0)
    def mk(X):
        t = torch.full((M, ops.ceil_to(X.shape[1], 8)), 7.0, dtype=torch.bfloat16, device=DEV)       # pad columns must not leak in
        t[:, :X.shape[1]] = X.to(DEV)
        return t
    Pd, Qd = mk(P), mk(Q)
    slab = torch.empty(1 << 24, device=DEV)
    got = {}
    for nsplit in (0, 64):
        dw = torch.ones(N, K, device=DEV); db = torch.ones(N, device=DEV)                            # accumulates into what is there
        ops.gemm_tn(PREC_BF16, Pd, Qd, dw, db, N, K, nsplit=nsplit, slab=slab)
        tol = 2e-5 * np.sqrt(M) * float(ref.abs().max())
        assert float((dw.cpu().double() - 1 - ref).abs().max()) <= tol, nsplit
        assert float((db.cpu().double() - 1 - refb).abs().max()) <= 2e-5 * np.sqrt(M) * float(refb.abs().max()) + 1e-4, nsplit
        got[nsplit] = dw.clone()
        dw2 = torch.ones(N, K, device=DEV); db2 = torch.ones(N, device=DEV)
        ops.gemm_tn(PREC_BF16, Pd, Qd, dw2, db2, N, K, nsplit=nsplit, slab=slab)
        assert torch.equal(dw, dw2), nsplit                                                            # fixed-order sums: bitwise reproducible
    assert float((got[0] - got[64]).abs().max()) <= 1e-5 * np.sqrt(M) * float(ref.abs().max())


@pytest.mark.parametrize("prec", [PREC_F32, PREC_BF16])
@pytest.mark.parametrize("with_mask", [True, False])
def test_bn_relu_drop_prologue_and_bwd_epilogues(prec, with_mask):
    M, K, N = 700, 256, 136
    g = torch.Generator().manual_seed(5)
    adt = ops.act_dtype(prec)
    y = _round(torch.randn(M, K, generator=g), prec)
    scale = torch.rand(K, generator=g) + 0.5
    shift = torch.randn(K, generator=g) * 0.3
    mask = (torch.rand(M, K, generator=g) < 0.9).to(torch.uint8) if with_mask else None
    inv_keep = 1.0 / 0.9 if with_mask else 1.0
    W = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    pl = _prep(W.to(DEV), b.to(DEV), prec)
    # as the operand loader computes it: inv_keep folded into fp32 scale / shift, ONE fused multiply-add (exact product + one
    # rounding, emulated in float64), ReLU, times the keep byte -- so both sides round the same values to the compute type
    ik32 = torch.tensor(inv_keep, dtype=torch.float32)
    sc32, sh32 = scale * ik32, shift * ik32
    h = torch.relu((y.double() * sc32.double() + sh32.double()).float()) * (mask.float() if with_mask else 1.0)
    hq = _round(h, prec)
    ref = hq.double() @ _round(W, prec).double().t() + b.double()
    yd = y.to(DEV).to(adt)
    md = mask.to(DEV) if with_mask else None
    pro = (scale.to(DEV), shift.to(DEV), md, inv_keep)
    out = torch.zeros(M, N, device=DEV)
    ops.gemm_nt(prec, yd, pl.w, N, K, out, bias=pl.bias, prologue=pro)
    assert float((out.cpu().double() - ref).abs().max()) <= _tol(K, float(ref.abs().max()))
    # TN with the same prologue on Q
    P = _round(torch.randn(M, 40, generator=g), prec)
    dw = torch.zeros(40, K, device=DEV); db = torch.zeros(40, device=DEV)
    ops.gemm_tn(prec, P.to(DEV), yd, dw, db, 40, K, q_prologue=pro)
    refw = P.double().t() @ hq.double()
    assert float((dw.cpu().double() - refw).abs().max()) <= 2e-5 * np.sqrt(M) * float(refw.abs().max())

    # dX GEMM with EPI_RELU_MASK:  dH = (dY @ W) * (H > 0)
    dY = _round(torch.randn(M, N, generator=g), prec)
    H = _round(torch.relu(torch.randn(M, K, generator=g)), prec)
    refd = (dY.double() @ _round(W, prec).double()) * (H > 0)
    outd = torch.zeros(M, K, dtype=adt, device=DEV)
    ops.gemm_nt(prec, dY.to(DEV).to(adt), pl.wt, K, N, outd, epilogue=ops.EPI_RELU_MASK, h=H.to(DEV).to(adt))
    assert float((outd.float().cpu().double() - refd).abs().max()) <= _tol(N, float(refd.abs().max()), prec == PREC_BF16)

    # dX GEMM with EPI_BN_BWD: d = (dY @ W) * keep * (y*scale+shift > 0); partials (sum d, sum d*xhat)
    mean = torch.randn(K, generator=g) * 0.1
    rstd = torch.rand(K, generator=g) + 0.5
    st = torch.zeros(2, K, dtype=torch.float64, device=DEV)
    bnargs = (scale.to(DEV), shift.to(DEV), mean.to(DEV), rstd.to(DEV), md, inv_keep)
    ops.gemm_nt(prec, dY.to(DEV).to(adt), pl.wt, K, N, None, epilogue=ops.EPI_BN_BWD, h=yd, bn=bnargs, stats=st)
    keep = mask.double() * inv_keep if with_mask else 1.0
    refd = (dY.double() @ _round(W, prec).double()) * keep * ((y * scale + shift) > 0)
    xhat = (y.double() - mean.double()) * rstd.double()
    np.testing.assert_allclose(st[0].cpu(), refd.sum(0), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(st[1].cpu(), (refd * xhat).sum(0), rtol=1e-4, atol=2e-2)
    # phase 1: dy = c0 * (d - c1 - xhat * c2), subtraction done on the f32 accumulators
    coef = torch.rand(3, K, generator=g) + 0.25
    ops.gemm_nt(prec, dY.to(DEV).to(adt), pl.wt, K, N, outd, epilogue=ops.EPI_BN_BWD, h=yd, bn=bnargs, bn_coef=coef.to(DEV))
    refy = coef[0].double() * (refd - coef[1].double() - xhat * coef[2].double())
    got = outd.float().cpu().double()
    assert float((got - refy).abs().max()) <= _tol(N, float(refy.abs().max()), prec == PREC_BF16)


@pytest.mark.parametrize("N,K", [(256, 512), (512, 572), (512, 256)])
def test_nt_wide_tiles(N, K):
    """The 128x256-tile / 8-wave NT kernel (normally used from 256 row tiles up) at a small ragged M, all epilogues."""
    from mmvae import _lib
    lib = _lib.load()
    lib.mmvae_set_tuning(0, 1)
    try:
        prec, M = PREC_BF16, 389
        g = torch.Generator().manual_seed(N + K)
        A = _round(torch.randn(M, K, generator=g), prec)
        W = torch.randn(N, K, generator=g) / np.sqrt(K)
        b = torch.randn(N, generator=g)
        pl = _prep(W.to(DEV), b.to(DEV), prec)
        ref = A.double() @ _round(W, prec).double().t() + b.double()
        for Ad in (A.to(DEV), torch.nn.functional.pad(A, (0, ops.ceil_to(K, 8) - K)).to(DEV).bfloat16()):
            for out_dt in (torch.float32, torch.bfloat16):
                out = torch.full((M, N), 7.0, dtype=out_dt, device=DEV)
                st = torch.zeros(2, N, dtype=torch.float64, device=DEV)
                ops.gemm_nt(prec, Ad, pl.w, N, K, out, bias=pl.bias, act=ops.ACT_RELU, stats=st)
                r = ref.clamp_min(0)
                got = out.float().cpu().double()
                assert float((got - r).abs().max()) <= _tol(K, float(r.abs().max()), out_dt == torch.bfloat16)
                np.testing.assert_allclose(st[0].cpu(), got.sum(0), rtol=1e-4, atol=1e-2)
        # backward epilogues on the wide kernel: ReLU mask and both BatchNorm forms (output width N, reduction K)
        adt = torch.bfloat16
        dY = _round(torch.randn(M, K, generator=g), prec)
        Wt = torch.randn(N, K, generator=g) / np.sqrt(K)           # plays W^T: [N out][K red]
        plt = _prep(Wt.to(DEV), torch.zeros(N, device=DEV), prec)
        base = dY.double() @ _round(Wt, prec).double().t()
        H = _round(torch.relu(torch.randn(M, N, generator=g)), prec)
        outd = torch.zeros(M, N, dtype=adt, device=DEV)
        dYd = torch.nn.functional.pad(dY, (0, ops.ceil_to(K, 8) - K)).to(DEV).to(adt)      # activation buffers have 8-element rows
        ops.gemm_nt(prec, dYd, plt.w, N, K, outd, epilogue=ops.EPI_RELU_MASK, h=H.to(DEV).to(adt))
        refd = base * (H > 0)
        assert float((outd.float().cpu().double() - refd).abs().max()) <= _tol(K, float(refd.abs().max()), True)
        y = _round(torch.randn(M, N, generator=g), prec)
        scale, shift = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.3
        mean, rstd = torch.randn(N, generator=g) * 0.1, torch.rand(N, generator=g) + 0.5
        mask = (torch.rand(M, N, generator=g) < 0.9).to(torch.uint8)
        bnargs = (scale.to(DEV), shift.to(DEV), mean.to(DEV), rstd.to(DEV), mask.to(DEV), 1.0 / 0.9)
        st = torch.zeros(2, N, dtype=torch.float64, device=DEV)
        ops.gemm_nt(prec, dYd, plt.w, N, K, outd, epilogue=ops.EPI_BN_BWD, h=y.to(DEV).to(adt), bn=bnargs, bn_phase=2, stats=st)
        d = base * (mask.double() / 0.9) * ((y * scale + shift) > 0)
        xhat = (y.double() - mean.double()) * rstd.double()
        assert float((outd.float().cpu().double() - d).abs().max()) <= _tol(K, float(d.abs().max()), True)
        np.testing.assert_allclose(st[0].cpu(), d.sum(0), rtol=1e-4, atol=2e-2)
        np.testing.assert_allclose(st[1].cpu(), (d * xhat).sum(0), rtol=1e-4, atol=4e-2)
        # BN prologue on the A operand with the wide kernel
        Kp = 256
        yp = _round(torch.randn(M, Kp, generator=g), prec)
        sc, sh = torch.rand(Kp, generator=g) + 0.5, torch.randn(Kp, generator=g) * 0.3
        mk = (torch.rand(M, Kp, generator=g) < 0.9).to(torch.uint8)
        W2 = torch.randn(N, Kp, generator=g) / 16
        pl2 = _prep(W2.to(DEV), b.to(DEV), prec)
        hq = _round(torch.relu(yp * sc + sh) * mk.float() / 0.9, prec)
        ref2 = hq.double() @ _round(W2, prec).double().t() + b.double()
        out2 = torch.zeros(M, N, device=DEV)
        ops.gemm_nt(prec, yp.to(DEV).to(adt), pl2.w, N, Kp, out2, bias=pl2.bias, prologue=(sc.to(DEV), sh.to(DEV), mk.to(DEV), 1.0 / 0.9))
        assert float((out2.cpu().double() - ref2).abs().max()) <= _tol(Kp, float(ref2.abs().max()))
    finally:
        lib.mmvae_set_tuning(0, 256 * 128)


@pytest.mark.parametrize("prec", [PREC_F32, PREC_BF16])
@pytest.mark.parametrize("q_kind", ["f32", "act"])
def test_tn_bn_bwd_apply_prologue(prec, q_kind):
    """dW GEMM with the BatchNorm-backward correction on its P operand == mmvae_bn_bwd_apply followed by the plain dW GEMM
    (first layers: reference autograd native_batch_norm_backward + mm, optimize_hyperparameters.py:112)."""
    M, N, K = 1000, 256, 150
    g = torch.Generator().manual_seed(11)
    adt = ops.act_dtype(prec)
    d = _round(torch.randn(M, N, generator=g), prec)
    y = _round(torch.randn(M, N, generator=g) * 2 + 0.3, prec)
    mean, rstd = torch.randn(N, generator=g) * 0.2, torch.rand(N, generator=g) + 0.5
    coef = torch.stack([torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1, torch.randn(N, generator=g) * 0.1])
    Q = torch.randn(M, K, generator=g)
    Qd = Q.to(DEV) if q_kind == "f32" else _round(Q, prec).to(DEV).to(adt)
    if q_kind == "act" and prec == PREC_BF16:
        t = torch.zeros(M, ops.ceil_to(K, 8), dtype=adt, device=DEV); t[:, :K] = Qd; Qd = t
    Qref = Q if q_kind == "f32" and prec == PREC_F32 else _round(Q, prec)
    xh = (y.double() - mean.double()) * rstd.double()
    dy = coef[0].double() * (d.double() - coef[1].double() - xh * coef[2].double())
    dyq = _round(dy.float(), prec).double()                       # the operand is rounded to the compute type once
    ref, refb = dyq.t() @ Qref.double(), dyq.sum(0)
    dd, yd = d.to(DEV).to(adt), y.to(DEV).to(adt)
    dw = torch.zeros(N, K, device=DEV); db = torch.zeros(N, device=DEV)
    ops.gemm_tn(prec, dd, Qd, dw, db, N, K, p_prologue=(yd, mean.to(DEV), rstd.to(DEV), coef.to(DEV).contiguous()))
    tol = (2e-5 if prec == PREC_F32 else 3e-3) * np.sqrt(M) * float(ref.abs().max())     # bf16: an operand rounding may flip
    assert float((dw.cpu().double() - ref).abs().max()) <= tol
    assert float((db.cpu().double() - refb).abs().max()) <= (2e-5 if prec == PREC_F32 else 3e-3) * np.sqrt(M) * float(refb.abs().max()) + 1e-3
    # and against the two-launch form it replaces
    d2 = dd.clone()
    ops.bn_bwd_apply(d2, yd, N, mean.to(DEV), rstd.to(DEV), coef.to(DEV).contiguous())
    dw2 = torch.zeros(N, K, device=DEV); db2 = torch.zeros(N, device=DEV)
    ops.gemm_tn(prec, d2, Qd, dw2, db2, N, K)
    assert float((dw - dw2).abs().max()) <= 1e-3 * float(dw2.abs().max())


@pytest.mark.parametrize("prec", [PREC_F32, PREC_BF16])
@pytest.mark.parametrize("M,N", [(1000, 256), (4133, 512), (77, 24), (5, 64)])
def test_bn_bwd_apply(prec, M, N):
    """mmvae_bn_bwd_apply (in place dy = c0 (d - c1 - xhat c2)) against float64, for widths that take the column-resident kernel
    (256 % (N / V) == 0) and one that takes the generic kernel (N = 24); the model test covers it inside the backward."""
    g = torch.Generator().manual_seed(M + N)
    adt = ops.act_dtype(prec)
    d = _round(torch.randn(M, N, generator=g), prec)
    y = _round(torch.randn(M, N, generator=g) * 2 + 0.3, prec)
    mean, rstd = torch.randn(N, generator=g) * 0.2, torch.rand(N, generator=g) + 0.5
    coef = torch.stack([torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1, torch.randn(N, generator=g) * 0.1])
    xh = (y.double() - mean.double()) * rstd.double()
    ref = coef[0].double() * (d.double() - coef[1].double() - xh * coef[2].double())
    dd, yd = d.to(DEV).to(adt), y.to(DEV).to(adt)
    ops.bn_bwd_apply(dd, yd, N, mean.to(DEV), rstd.to(DEV), coef.to(DEV).contiguous())
    scale = float(ref.abs().max())
    tol = scale * (2.0 ** -8 if prec == PREC_BF16 else 1e-5)
    assert float((dd.float().cpu().double() - ref).abs().max()) <= tol


@pytest.mark.parametrize("case", ["store_f32_sigmoid", "store_bf16_stats", "relu_mask", "bn_bwd"])
def test_nt_kernel_generations_agree(case):
    """The two tile generations of the NT kernel (gemm_nt.hip register-staged; gemm_nt2.h LDS-DMA ring, persistent) on the SAME
    operands, switched inside one process with mmvae_set_tuning: identical products (the epilogue arithmetic is shared, the MFMA
    accumulation order over K is the same K-step order), at a size where both are eligible -- ragged N (572), K = 512, M not a
    multiple of the tile height."""
    from mmvae import _lib as L
    lib = L.load()
    dev = "cuda"
    M, N, K = 16384 + 300, (572 if case.startswith("store") else 512), (512 if case != "bn_bwd" else 256)
    g = torch.Generator().manual_seed(7)
    A = torch.randn(M, K, generator=g).to(dev).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    pl = ops.PreparedLinear([W], [bias], PREC_BF16, dev)
    ops.WeightPrep([pl], dev).run()
    H = torch.randn(M, ops.ceil_to(N, 8), generator=g).to(dev).bfloat16()
    mask = (torch.rand(M, N, generator=g) > 0.1).to(torch.uint8).to(dev)
    f = lambda: (torch.rand(N, generator=g) + 0.5).to(dev)
    bn = (f(), f() - 1.0, f() - 1.0, f(), mask, 1.0 / 0.9)

    def run():
        stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
        if case == "store_f32_sigmoid":
            out = torch.empty(M, N, device=dev)
            ops.gemm_nt(PREC_BF16, A, pl.w, N, K, out, bias=pl.bias, act=ops.ACT_SIGMOID)
        elif case == "store_bf16_stats":
            out = torch.zeros(M, ops.ceil_to(N, 8), dtype=torch.bfloat16, device=dev)
            ops.gemm_nt(PREC_BF16, A, pl.w, N, K, out, bias=pl.bias, stats=stats)
        elif case == "relu_mask":
            out = torch.zeros(M, ops.ceil_to(N, 8), dtype=torch.bfloat16, device=dev)
            ops.gemm_nt(PREC_BF16, A, pl.w, N, K, out, epilogue=ops.EPI_RELU_MASK, h=H)
        else:
            out = torch.zeros(M, ops.ceil_to(N, 8), dtype=torch.bfloat16, device=dev)
            ops.gemm_nt(PREC_BF16, A, pl.w, N, K, out, epilogue=ops.EPI_BN_BWD, h=H, bn=bn, bn_phase=2, stats=stats)
        torch.cuda.synchronize()
        return out.float(), stats

    res = {}
    try:
        for name, nt2 in (("gen1", 0), ("gen2", 1)):
            lib.mmvae_set_tuning(2, nt2)
            res[name] = run()
    finally:
        lib.mmvae_set_tuning(2, 1)
    assert lib.mmvae_set_tuning(1, 1) == -1 and lib.mmvae_set_tuning(5, 1) == -1      # retired kernel generations: keys rejected
    ref, ref_stats = res["gen1"]
    for name in ("gen2",):
        out, stats = res[name]
        assert torch.equal(out, ref), (name, float((out - ref).abs().max()))
        if case in ("store_bf16_stats", "bn_bwd"):
            assert torch.allclose(stats, ref_stats, rtol=1e-6, atol=1e-6 * float(ref_stats.abs().max())), name      # atomics order


@pytest.mark.parametrize("M,N,K", [(16384, 512, 572), (16384, 128, 782), (8192 + 96, 384, 300), (12000, 512, 572), (16384, 100 * 8, 256)])
def test_tn_wide_tiles(M, N, K):
    """The wide-tile dW kernels (gemm_tn_wide.hip: 256 x 288 LDS-DMA form, 128 x 448 register form; BatchNorm-corrected bf16 P x fp32
    Q, M >= 8192 with a slab) against float64 on the same rounded operands, against the 128 x 128 kernel (mmvae_set_tuning key 4),
    with N / K tails, a batch that is not a multiple of the 32-row step, and a row count whose last split is short."""
    prec = PREC_BF16
    g = torch.Generator().manual_seed(5)
    Np = ops.ceil_to(N, 8)
    d = torch.zeros(M, Np, dtype=torch.bfloat16); y = torch.zeros(M, Np, dtype=torch.bfloat16)
    d[:, :N] = torch.randn(M, N, generator=g).bfloat16(); y[:, :N] = (torch.randn(M, N, generator=g) * 2 + 0.3).bfloat16()
    mean, rstd = torch.randn(N, generator=g) * 0.2, torch.rand(N, generator=g) + 0.5
    coef = torch.stack([torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1, torch.randn(N, generator=g) * 0.1]).contiguous()
    Q = torch.randn(M, K, generator=g)
    xh = (y[:, :N].double() - mean.double()) * rstd.double()
    dy = coef[0].double() * (d[:, :N].double() - coef[1].double() - xh * coef[2].double())
    dyq = _round(dy.float(), prec).double()
    ref, refb = dyq.t() @ _round(Q, prec).double(), dyq.sum(0)
    slab = torch.empty(1 << 25, device=DEV)
    from mmvae import _lib
    lib = _lib.load()
    res = []
    try:
        for on in (1, 0):
            assert lib.mmvae_set_tuning(4, on) == 0
            dw = torch.zeros(N, K, device=DEV); db = torch.zeros(N, device=DEV)
            ops.gemm_tn(prec, d.to(DEV), Q.to(DEV), dw, db, N, K, p_prologue=(y.to(DEV), mean.to(DEV), rstd.to(DEV), coef.to(DEV)), slab=slab)
            res.append((dw.cpu().double(), db.cpu().double()))
    finally:
        lib.mmvae_set_tuning(4, 1)
    tol = 3e-3 * np.sqrt(M) * float(ref.abs().max())                 # a bf16 rounding of a corrected P element may flip (see above)
    for dw, db in res:
        assert float((dw - ref).abs().max()) <= tol
        assert float((db - refb).abs().max()) <= 3e-3 * np.sqrt(M) * float(refb.abs().max()) + 1e-3
    # wide vs 128 x 128: the same bf16 operands (same correction formula), only the fp32 summation order differs
    assert float((res[0][0] - res[1][0]).abs().max()) <= 2e-5 * np.sqrt(M) * float(ref.abs().max())
    assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-5 * np.sqrt(M) * float(refb.abs().max()) + 1e-4


@pytest.mark.parametrize("M,N,K,with_mask", [(1000, 512, 256, True), (777, 208, 128, True), (4096, 256, 192, False)])
def test_bn_bwd_epilogue_row_coalesced_form(M, N, K, with_mask):
    """BatchNorm+ReLU+Dropout backward epilogue of the dX GEMM (phase 2: store d, sums of d and d*xhat) in the row-coalesced LDS
    form of the second-generation kernel (mmvae_set_tuning key 6) against the accumulator-layout form and against float64:
    same d bits (same arithmetic on the same accumulators), statistics equal to summation order."""
    from mmvae import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    pl = _prep(W.to(DEV), torch.zeros(N, device=DEV), PREC_BF16)
    Np = ops.ceil_to(N, 8)
    Y = torch.zeros(M, Np, dtype=torch.bfloat16); Y[:, :N] = torch.randn(M, N, generator=g).bfloat16(); Y = Y.to(DEV)
    mask = (torch.rand(M, ops.ceil_to(N, 16), generator=g) > 0.1).to(torch.uint8).to(DEV) if with_mask else None
    f = lambda: (torch.rand(N, generator=g) + 0.5).to(DEV)
    sc, sh, mu, rs = f(), f() - 1.0, f() - 1.0, f()
    res = []
    try:
        for on in (0, 1):
            lib.mmvae_set_tuning(6, on)
            d = torch.full((M, Np), 5.0, dtype=torch.bfloat16, device=DEV)
            st = torch.zeros(2, N, dtype=torch.float64, device=DEV)
            ops.gemm_nt(PREC_BF16, A, pl.wt if False else pl.w, N, K, d, epilogue=ops.EPI_BN_BWD, h=Y, bn=(sc, sh, mu, rs, mask, 1.0 / 0.9), bn_phase=2, stats=st)
            res.append((d, st))
    finally:
        lib.mmvae_set_tuning(6, 1)
    assert torch.equal(res[0][0][:, :N].view(torch.int16), res[1][0][:, :N].view(torch.int16))
    assert float(res[1][0][:, N:].float().abs().max() if Np > N else 0.0) == 0.0                    # pad columns zeroed
    scale = res[0][1].abs().max(dim=1, keepdim=True).values
    assert float(((res[0][1] - res[1][1]).abs() / scale).max()) <= 1e-5
    acc = A.double().cpu() @ _round(W, PREC_BF16).double().t()
    y = Y[:, :N].double().cpu()
    keep = (mask[:, :N].double().cpu() / 0.9) if with_mask else 1.0
    dref = torch.where(y * sc.double().cpu() + sh.double().cpu() > 0, acc * keep, torch.zeros_like(acc))
    xh = (y - mu.double().cpu()) * rs.double().cpu()
    assert float((res[1][0][:, :N].double().cpu() - dref).abs().max()) <= _tol(K, float(dref.abs().max()), out_bf16=True)
    ref_st = torch.stack([dref.sum(0), (dref * xh).sum(0)])
    assert float((res[1][1].cpu() - ref_st).abs().max()) <= 3e-5 * np.sqrt(M) * float(ref_st.abs().max()) + 1e-3


@pytest.mark.parametrize("M,N,K", [(1000, 512, 572), (333, 200, 128), (4100, 128, 782)])
def test_relu_mask_epilogue_row_coalesced_form(M, N, K):
    """ReLU-backward dX epilogue in the row-coalesced LDS form (mmvae_set_tuning key 7) against the accumulator-layout form: same
    bits; output and saved activation as column slices of wider buffers (the merged decoder stem hands such slices over)."""
    from mmvae import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, ops.ceil_to(K, 8), generator=g).bfloat16().to(DEV)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    pl = _prep(W.to(DEV), torch.zeros(N, device=DEV), PREC_BF16)
    Np = ops.ceil_to(N, 8)
    Hbig = torch.randn(M, Np + 64, generator=g).bfloat16().to(DEV)
    H = Hbig[:, 32:32 + Np]
    res = []
    try:
        for on in (0, 1):
            lib.mmvae_set_tuning(7, on)
            Cbig = torch.full((M, Np + 64), 5.0, dtype=torch.bfloat16, device=DEV)
            C = Cbig[:, 16:16 + Np]
            ops.gemm_nt(PREC_BF16, A, pl.w, N, K, C, epilogue=ops.EPI_RELU_MASK, h=H)
            res.append(Cbig)
    finally:
        lib.mmvae_set_tuning(7, 1)
    assert torch.equal(res[0].view(torch.int16), res[1].view(torch.int16))          # incl. the untouched neighbours of the slice
    ref = torch.where(H[:, :N].double().cpu() > 0, A[:, :K].double().cpu() @ _round(W, PREC_BF16).double().t(), torch.zeros(M, N, dtype=torch.float64))
    assert float((res[1][:, 16:16 + N].double().cpu() - ref).abs().max()) <= _tol(K, float(ref.abs().max()), out_bf16=True)


@pytest.mark.parametrize("M,N,K", [(8192, 512, 2000), (8192, 256, 12000), (16384, 128, 6000)])
def test_tn_wide_tiles_plain_p_fp32_q(M, N, K):
    """Wide-tile dW kernel with a plain bf16 P and an fp32 Q (first encoder layers at very wide inputs, where the engine applies the
    BatchNorm correction in a pass of its own): LDS-DMA forms of both tile shapes, incl. the linear block map used below 8 batch
    splits (K = 12000 -> 42 tiles, 6 splits), against the 128 x 128 kernel (mmvae_set_tuning key 4) and float64."""
    from mmvae import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(N + K)
    P = torch.randn(M, N, generator=g).bfloat16()
    Q = torch.randn(M, K, generator=g)
    ref, refb = P.double().t() @ _round(Q, PREC_BF16).double(), P.double().sum(0)
    slab = torch.empty(1 << 25, device=DEV)
    Pd, Qd = P.to(DEV), Q.to(DEV)
    res = []
    try:
        for on in (1, 0):
            assert lib.mmvae_set_tuning(4, on) == 0
            dw = torch.zeros(N, K, device=DEV); db = torch.zeros(N, device=DEV)
            ops.gemm_tn(PREC_BF16, Pd, Qd, dw, db, N, K, slab=slab)
            res.append((dw.cpu().double(), db.cpu().double()))
    finally:
        lib.mmvae_set_tuning(4, 1)
    for dw, db in res:
        assert float((dw - ref).abs().max()) <= 2e-5 * np.sqrt(M) * float(ref.abs().max())
        assert float((db - refb).abs().max()) <= 2e-5 * np.sqrt(M) * float(refb.abs().max()) + 1e-4


def test_tn_wide_tiles_bf16_operands_large_output():
    """bf16 x bf16 dW with a very large output (the decoders' last layers at the scaled omics widths, N K >= 4 M elements) runs the
    wide-tile LDS-DMA kernel (3-stage ring, linear block map): against the 128 x 128 kernel (mmvae_set_tuning key 4) and float64;
    N and K tails."""
    from mmvae import _lib
    lib = _lib.load()
    M, N, K = 8192, 4100, 1100
    g = torch.Generator().manual_seed(77)
    P = torch.zeros(M, ops.ceil_to(N, 8), dtype=torch.bfloat16); P[:, :N] = torch.randn(M, N, generator=g).bfloat16()
    Q = torch.zeros(M, ops.ceil_to(K, 8), dtype=torch.bfloat16); Q[:, :K] = torch.randn(M, K, generator=g).bfloat16()
    ref, refb = P[:, :N].double().t() @ Q[:, :K].double(), P[:, :N].double().sum(0)
    slab = torch.empty(1 << 25, device=DEV)
    Pd, Qd = P.to(DEV), Q.to(DEV)
    res = []
    try:
        for on in (1, 0):
            assert lib.mmvae_set_tuning(4, on) == 0
            dw = torch.zeros(N, K, device=DEV); db = torch.zeros(N, device=DEV)
            ops.gemm_tn(PREC_BF16, Pd, Qd, dw, db, N, K, slab=slab)
            res.append((dw.cpu().double(), db.cpu().double()))
    finally:
        lib.mmvae_set_tuning(4, 1)
    for dw, db in res:
        assert float((dw - ref).abs().max()) <= 2e-5 * np.sqrt(M) * float(ref.abs().max())
        assert float((db - refb).abs().max()) <= 2e-5 * np.sqrt(M) * float(refb.abs().max()) + 1e-4
    assert float((res[0][0] - res[1][0]).abs().max()) <= 2e-5 * np.sqrt(M) * float(ref.abs().max())


def _set_tuning(key, value):
    from mmvae import _lib
    _lib.check(_lib.load().mmvae_set_tuning(key, value), "mmvae_set_tuning")


@pytest.mark.parametrize("M,N,K,lda,stats,act", [
    (4480, 512, 572, 572, True, ops.ACT_NONE),      # EncoderB.L0 shape: 16-byte rows, 128 x 256 tiles, row tiles padded to a multiple of 8
    (2048, 128, 782, 782, True, ops.ACT_NONE),      # EncoderA.L0 shape: 8-byte rows (16-byte loads at 8-byte alignment), K % 4 == 2: rotated tail piece
    (12288, 384, 200, 201, True, ops.ACT_RELU),     # odd leading dimension (4-byte aligned rows); 3 column tiles: a workgroup changes column tile
    (1024, 256, 130, 132, False, ops.ACT_NONE),     # K % 64 != 0 and K % 4 != 0 at 16-byte rows; no statistics
    (640, 128, 97, 97, True, ops.ACT_SIGMOID),      # K % 4 == 1
    (2056, 128, 782, 782, True, ops.ACT_NONE),      # a partial row tile: not taken by the wave-specialised kernel (both runs use the tile kernels)
])
def test_ntp_matches_tile_kernels(M, N, K, lda, stats, act):
    """The wave-specialised NT kernel (gemm_ntp.h: producer / consumer waves, persistent) against the tile kernels it replaces on the
    forward first layers: the same MFMA order over K -> bit-identical bf16 outputs; column statistics to fp32 summation order."""
    g = torch.Generator().manual_seed(M + N + K)
    Af = torch.randn(M, lda, generator=g).to(DEV)
    A = Af[:, :K]
    W = (torch.randn(N, K, generator=g) / np.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    pl = _prep(W, b, PREC_BF16)
    res = {}
    try:
        _set_tuning(9, 256)
        for on in (0, 1):
            _set_tuning(8, on)
            out = torch.full((M, ops.ceil_to(N, 8)), 7.0, dtype=torch.bfloat16, device=DEV)
            st = torch.zeros(2, N, dtype=torch.float64, device=DEV) if stats else None
            ops.gemm_nt(PREC_BF16, A, pl.w, N, K, out, bias=pl.bias, act=act, stats=st)
            torch.cuda.synchronize()
            res[on] = (out.clone(), None if st is None else st.clone())
    finally:
        _set_tuning(8, 1); _set_tuning(9, 16384)
    ref = A.to(torch.bfloat16).double() @ W.to(torch.bfloat16).double().t() + b.double()
    if act == ops.ACT_RELU:
        ref = ref.clamp_min(0)
    elif act == ops.ACT_SIGMOID:
        ref = torch.sigmoid(ref)
    got = res[1][0][:, :N].double()
    assert float((got - ref).abs().max()) <= _tol(K, float(ref.abs().max()), True)
    assert torch.equal(res[0][0][:, :N], res[1][0][:, :N])
    assert torch.all((res[1][0][:, N:].float() == 7.0) | (res[1][0][:, N:].float() == 0.0))
    if stats:
        np.testing.assert_allclose(res[1][1].cpu(), res[0][1].cpu(), rtol=2e-6, atol=1e-3)
        np.testing.assert_allclose(res[1][1][0].cpu(), got.sum(0).cpu(), rtol=1e-5, atol=1e-2)
        np.testing.assert_allclose(res[1][1][1].cpu(), (got ** 2).sum(0).cpu(), rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("M,N,K,masked,stats", [(1024, 256, 512, True, True),      # EncoderB's second Linear (encoders.py:35) behind BN + ReLU + Dropout
                                                (512, 128, 256, True, False),      # 128 x 128 tiles
                                                (768, 256, 128, False, True)])     # eval mode: no dropout mask
def test_ntp_prologue_matches_tile_kernels(M, N, K, masked, stats):
    """gemm_ntp.h with the producers' BatchNorm + ReLU + Dropout operand prologue (bf16 A) against the register-staged tile kernel with
    the same prologue (SrcBnReluDrop): same per-element arithmetic and MFMA order -> bit-identical bf16 outputs; and against fp64."""
    g = torch.Generator().manual_seed(M + N + K)
    Y = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    scale = (torch.rand(K, generator=g) + 0.5).to(DEV); shift = (torch.randn(K, generator=g) * 0.3).to(DEV)
    mask = (torch.rand(M, K, generator=g) > 0.1).to(torch.uint8).to(DEV) if masked else None
    inv_keep = 1.0 / 0.9 if masked else 1.0
    W = (torch.randn(N, K, generator=g) / np.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    pl = _prep(W, b, PREC_BF16)
    res = {}
    try:
        _set_tuning(9, 256)
        for on in (0, 1):
            _set_tuning(8, on)
            out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
            st = torch.zeros(2, N, dtype=torch.float64, device=DEV) if stats else None
            ops.gemm_nt(PREC_BF16, Y, pl.w, N, K, out, bias=pl.bias, prologue=(scale, shift, mask, inv_keep), stats=st)
            torch.cuda.synchronize()
            res[on] = (out.clone(), None if st is None else st.clone())
    finally:
        _set_tuning(8, 1); _set_tuning(9, 16384)
    h = torch.relu(Y.double() * (scale * inv_keep).double() + (shift * inv_keep).double())
    if masked:
        h = h * mask.double()
    ref = h.to(torch.float32).to(torch.bfloat16).double() @ W.to(torch.bfloat16).double().t() + b.double()
    got = res[1][0].double()
    assert float((got - ref).abs().max()) <= _tol(K, float(ref.abs().max()), True) + 2e-2 * float(ref.abs().max())      # + the prologue's bf16 rounding
    assert torch.equal(res[0][0], res[1][0])
    if stats:
        np.testing.assert_allclose(res[1][1].cpu(), res[0][1].cpu(), rtol=2e-6, atol=1e-3)


@pytest.mark.parametrize("M,N,K,lda,act", [(1024, 512, 256, 256, ops.ACT_RELU),      # DecoderB's hidden Linear + ReLU (decoders.py:29-30)
                                           (512, 128, 200, 208, ops.ACT_NONE)])      # K % 64 != 0, row pitch > K
def test_ntp_plain_bf16_matches_tile_kernels(M, N, K, lda, act):
    """gemm_ntp.h on a plain bf16 A operand (the producers copy 16-byte chunks into the ring) against the LDS-DMA tile kernel: bit-identical."""
    g = torch.Generator().manual_seed(M + N + K)
    Af = torch.zeros(M, lda, dtype=torch.bfloat16, device=DEV)
    Af[:, :K] = torch.randn(M, K, generator=g).to(DEV)
    A = Af[:, :K]
    W = (torch.randn(N, K, generator=g) / np.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    pl = _prep(W, b, PREC_BF16)
    res = {}
    try:
        _set_tuning(9, 256)
        for on in (0, 1):
            _set_tuning(8, on)
            out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
            ops.gemm_nt(PREC_BF16, A, pl.w, N, K, out, bias=pl.bias, act=act)
            torch.cuda.synchronize()
            res[on] = out.clone()
    finally:
        _set_tuning(8, 1); _set_tuning(9, 16384)
    ref = A.double() @ W.to(torch.bfloat16).double().t() + b.double()
    if act == ops.ACT_RELU:
        ref = ref.clamp_min(0)
    assert float((res[1].double() - ref).abs().max()) <= _tol(K, float(ref.abs().max()), True)
    assert torch.equal(res[0], res[1])


def test_ntp_prologue_output_is_the_post_activation():
    """pro_out of mmvae_gemm_nt (wave-specialised kernel only): the operand after BatchNorm-normalise + ReLU + Dropout, bit for bit what
    the dW GEMM's own operand prologue would form; the GEMM result is unchanged; problems the kernel does not take answer ERR_ARG."""
    M, N, K = 16384, 256, 512
    g = torch.Generator().manual_seed(3)
    Y = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    scale = (torch.rand(K, generator=g) + 0.5).to(DEV); shift = (torch.randn(K, generator=g) * 0.3).to(DEV)
    mask = (torch.rand(M, K, generator=g) > 0.1).to(torch.uint8).to(DEV)
    W = (torch.randn(N, K, generator=g) / np.sqrt(K)).to(DEV)
    pl = _prep(W, torch.zeros(N, device=DEV), PREC_BF16)
    pro = (scale, shift, mask, 1.0 / 0.9)
    out0 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); out1 = torch.empty_like(out0)
    H = torch.full((M, K), 7.0, dtype=torch.bfloat16, device=DEV)
    assert ops.can_keep_pro_out(PREC_BF16, M, N, K, Y, out1)
    ops.gemm_nt(PREC_BF16, Y, pl.w, N, K, out0, bias=pl.bias, prologue=pro)
    ops.gemm_nt(PREC_BF16, Y, pl.w, N, K, out1, bias=pl.bias, prologue=pro, pro_out=H)
    assert torch.equal(out0, out1)
    ik = torch.tensor(1.0 / 0.9, dtype=torch.float32, device=DEV)
    want = (torch.relu((Y.double() * (scale * ik).double() + (shift * ik).double()).float()) * mask.float()).bfloat16()     # one fused multiply-add, as the kernel
    assert torch.equal(H, want)
    # dW from the kept operand == dW through the operand prologue
    P = torch.randn(M, 40, generator=g).to(DEV)
    dw0 = torch.zeros(40, K, device=DEV); db0 = torch.zeros(40, device=DEV); dw1 = torch.zeros_like(dw0); db1 = torch.zeros_like(db0)
    ops.gemm_tn(PREC_BF16, P, Y, dw0, db0, 40, K, q_prologue=pro)
    ops.gemm_tn(PREC_BF16, P, H, dw1, db1, 40, K)
    assert float((dw0 - dw1).abs().max()) <= 1e-5 * float(dw0.abs().max())
    small = Y[:1024]
    with pytest.raises(RuntimeError):                         # below the kernel's minimum M: nobody would write pro_out
        ops.gemm_nt(PREC_BF16, small, pl.w, N, K, out0[:1024], bias=pl.bias, prologue=(scale, shift, mask[:1024], 1.0 / 0.9), pro_out=H[:1024])
