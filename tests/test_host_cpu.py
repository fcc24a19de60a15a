"""CPU-side checks of the host layer: the C-ABI library loads and exports every symbol declared in
include/mmvae_hip.h (no compute calls without a GPU), the drop-in surface has the reference's
names and state_dict ABI, and the product path refuses to run without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest
import torch

import np_oracle as O
from mmvae import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mmvae_hip.h")).read()
    declared = set(re.findall(r"^int (mmvae_\w+)\(", hdr, re.M))
    assert declared == set(_lib.EXPORTED), declared ^ set(_lib.EXPORTED)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mmvae_abi_version() == _lib.ABI_VERSION


def test_argument_checks_reject_before_launch():
    """Argument validation happens on the host before any launch, so it is testable without a GPU."""
    import ctypes as C
    lib = _lib.load()
    assert lib.mmvae_gemm_nt(None, None) == -1
    g = _lib.GemmNtArgs()
    assert lib.mmvae_gemm_nt(C.byref(g), None) == -1
    t = _lib.GemmTnArgs()
    assert lib.mmvae_gemm_tn(C.byref(t), None) == -1
    assert lib.mmvae_prep_weights(None, 0, None) == -1
    b = _lib.BnFinalizeArgs()
    b.M, b.N = 1, 8                        # BatchNorm1d training needs > 1 row
    assert lib.mmvae_bn_finalize(C.byref(b), None) == -1
    assert lib.mmvae_adamw_step(None, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, 1.0, 0, None, 0, None, None) == -1
    assert lib.mmvae_noise(None, 0, 0.9, None, 0, 1, 0, None, 0, None) == -1
    la = _lib.LossArgs()
    la.B = 4                               # no accumulator buffer
    assert lib.mmvae_vae_loss(C.byref(la), None) == -1


def test_ctypes_structs_match_c_layout(tmp_path):
    """sizeof/offsetof of every args struct as gcc lays the header out == the ctypes mirror."""
    import ctypes as C
    import subprocess
    pairs = {"mmvae_prep_item": _lib.PrepItem, "mmvae_gemm_nt_args": _lib.GemmNtArgs, "mmvae_gemm_tn_args": _lib.GemmTnArgs,
             "mmvae_bn_finalize_args": _lib.BnFinalizeArgs, "mmvae_bn_bwd_finalize_args": _lib.BnBwdFinalizeArgs,
             "mmvae_fuse_fwd_args": _lib.FuseFwdArgs, "mmvae_fuse_bwd_args": _lib.FuseBwdArgs, "mmvae_loss_args": _lib.LossArgs,
             "mmvae_adamw_item": _lib.AdamWItem, "mmvae_gather_item": _lib.GatherItem}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mmvae_hip.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0; }")
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


def test_binding_constants_match_the_header(tmp_path):
    """Enumerators and macros of include/mmvae_hip.h as gcc sees them == the constants of the ctypes binding."""
    import subprocess
    names = {"MMVAE_EPI_STORE": _lib.EPI_STORE, "MMVAE_EPI_RELU_MASK": _lib.EPI_RELU_MASK, "MMVAE_EPI_BN_BWD": _lib.EPI_BN_BWD,
             "MMVAE_EPI_LOSS_MSE": _lib.EPI_LOSS_MSE, "MMVAE_EPI_LOSS_BCE_LOGIT": _lib.EPI_LOSS_BCE_LOGIT,
             "MMVAE_CTR_COPIES": _lib.CTR_COPIES, "MMVAE_TN_GROUP_MAX": _lib.TN_GROUP_MAX, "MMVAE_TABLE_COPIES": _lib.TABLE_COPIES,
             "MMVAE_PRO_NONE": _lib.PRO_NONE, "MMVAE_PRO_BN_RELU_DROP": _lib.PRO_BN_RELU_DROP, "MMVAE_PRO_BN_BWD_APPLY": _lib.PRO_BN_BWD_APPLY,
             "MMVAE_F32": _lib.F32, "MMVAE_BF16": _lib.BF16, "MMVAE_PREC_F32": _lib.PREC_F32, "MMVAE_PREC_BF16": _lib.PREC_BF16}
    lines = ['#include <stdio.h>', '#include "mmvae_hip.h"', "int main(void) {"]
    lines += [f'printf("{n} %ld\\n", (long)({n}));' for n in names]
    lines.append("return 0; }")
    src = tmp_path / "consts.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "consts"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n, v in names.items():
        assert int(got[n]) == v, n


def test_drop_in_surface_and_state_dict_abi():
    import src.models as M
    import src.utils as U
    from src.utils.directional_losses import rna2dna_loss, dna2rna_loss  # noqa: F401
    for name in ("MultiModalVAE", "reparameterize", "EncoderA", "EncoderB", "EncoderC", "DecoderA", "DecoderB", "DecoderC",
                 "RNA2DNAVAE", "DNA2RNAVAE"):
        assert hasattr(M, name), name
    assert U.__all__ == ["vae_loss"]
    A, D, S, L, E = 782, 572, 24, 20, 32
    m = M.MultiModalVAE(A, D, S, L)
    P, Bf = O.make_params(0, A, D, S, L, E)
    assert set(m.state_dict()) == set(P) | set(Bf)
    assert sum(p.numel() for p in m.parameters()) == 1_081_114
    assert [k for k, _ in m.named_parameters()] == [k for k, _ in O.param_shapes(A, D, S, L, E)]
    d = M.RNA2DNAVAE(A, D, S, L)
    assert {k.split(".")[0] for k in d.state_dict()} == {"encoder_rna", "encoder_site", "decoder_dna"}
    d = M.DNA2RNAVAE(A, D, S, L)
    assert {k.split(".")[0] for k in d.state_dict()} == {"encoder_dna", "encoder_site", "decoder_rna"}


def test_same_seed_initialisation_as_stock_modules():
    """Parameter containers are stock nn.Linear/BatchNorm1d/Embedding created in the reference's
    order, so torch.manual_seed(s) gives the same initial weights as the reference would."""
    import src.models as M
    torch.manual_seed(3)
    m = M.EncoderA(30, 4)
    torch.manual_seed(3)
    ref = torch.nn.Linear(30, 128)
    assert torch.equal(m.fc[0].weight, ref.weight) and torch.equal(m.fc[0].bias, ref.bias)


def test_no_cpu_fallback():
    import src.models as M
    from src.utils import vae_loss
    m = M.MultiModalVAE(16, 12, 3, 2)
    assert m() == (None, None, None, None, None)
    x = torch.rand(4, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(a=x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vae_loss(torch.rand(4, 16), x, torch.rand(4, 12), torch.rand(4, 12), torch.rand(4, 3), torch.zeros(4, dtype=torch.long),
                 torch.zeros(4, 2), torch.zeros(4, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M.DecoderA(2, 16)(torch.rand(4, 2))


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MMVAELibraryError, match="no CPU / eager fallback"):
        _lib.load()


def test_fused_adamw_is_gpu_only():
    from mmvae.optim import FusedAdamW
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FusedAdamW([p]).step()
    with pytest.raises(ValueError):
        FusedAdamW([p], lr=-1.0)


def test_balanced_class_weights_match_the_reference_closed_form():
    """optimize_hyperparameters.py:33-44: sklearn compute_class_weight('balanced') over the classes PRESENT in the training
    labels (n / (k_present * count)), weight 1 for classes that do not occur."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "vae-los-angeles_amd"))
    from trainer import balanced_class_weights
    from sklearn.utils.class_weight import compute_class_weight
    rng = np.random.default_rng(0)
    for n_classes, present in ((24, list(range(24))), (24, [0, 3, 4, 9, 23]), (5, [2])):
        y = rng.choice(present, size=997, p=np.random.default_rng(1).dirichlet(np.ones(len(present))))
        uniq = np.unique(y)
        want = np.ones(n_classes, dtype=np.float32)
        want[uniq] = compute_class_weight(class_weight="balanced", classes=uniq, y=y)        # the reference's own lines 37-43
        got = balanced_class_weights(torch.from_numpy(y), n_classes).numpy()
        np.testing.assert_allclose(got, want, rtol=1e-6)
