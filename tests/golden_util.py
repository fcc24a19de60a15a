"""Helpers to compare arrays with the (possibly sampled) entries of tests/golden/*.npz."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def has(fx, key):
    return key in fx.files or (key + "@idx") in fx.files


def expect(fx, key, arr, rtol, atol, what="", scale_atol=0.0, outlier_frac=0.0, outlier_atol=0.0):
    """Assert `arr` matches fixture entry `key` (full or sampled).

    The absolute tolerance is atol + scale_atol * max|expected| (a tensor-scale term for
    sum-reduced quantities such as weight gradients).  Up to `outlier_frac` of the
    elements may miss that tolerance as long as they are within `outlier_atol` (Adam turns
    a gradient that is pure rounding noise into a full +-lr step)."""
    def _cmp(got, ref, atol, msg):
        bad = np.abs(got - ref) > atol + rtol * np.abs(ref)
        if outlier_frac and bad.mean() <= outlier_frac:
            assert np.all(np.abs(got - ref)[bad] <= outlier_atol), msg
            return
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=msg)
    arr = np.asarray(arr, dtype=np.float64)
    if key in fx.files:
        ref = fx[key].astype(np.float64)
        atol = atol + scale_atol * (float(np.max(np.abs(ref))) if ref.size else 0.0)
        assert ref.shape == arr.shape, (key, ref.shape, arr.shape)
        _cmp(arr, ref, atol, f"{what}{key}")
        return float(np.max(np.abs(arr - ref))) if ref.size else 0.0
    idx = fx[key + "@idx"]
    shape = tuple(fx[key + "@shape"])
    assert shape == arr.shape, (key, shape, arr.shape)
    val = fx[key + "@val"].astype(np.float64)
    atol = atol + scale_atol * float(np.max(np.abs(val)))
    got = arr.reshape(-1)[idx]
    _cmp(got, val, atol, f"{what}{key} (sampled)")
    ssq = float(fx[key + "@sumsq"])
    np.testing.assert_allclose((arr ** 2).sum(), ssq, rtol=max(rtol * 10, 1e-6), atol=atol,
                               err_msg=f"{what}{key} (sumsq)")
    return float(np.max(np.abs(got - val)))
