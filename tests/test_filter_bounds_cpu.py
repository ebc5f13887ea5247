"""
The error bound of the one-product fp16 pre-filter (segk_kmeans.hip, k_kmeans_score_h1), checked on a numpy
model of its operands: rows and means are scaled by a power of two so that the largest element lies in
[2^12, 2^13) and rounded once to fp16 (np.float16: round-to-nearest-even with gradual underflow, as
v_cvt_f16_f32); the products are exact in fp32 and their sum is taken here in float64, so what is measured
is the operand rounding alone -- the term 1.01 * 2^-10 |x| M of tau_A.  (The fp32 accumulation error is
covered by the split-precision margin E1', tested on the device in tests/test_gpu_kmeans.py.)
"""
import numpy as np
import pytest


def _scaled_f16(A):
    vmax = np.abs(A).max()
    e = 13 - np.frexp(np.float32(vmax))[1] if vmax > 0 else 0
    return np.ldexp(A.astype(np.float32), e).astype(np.float16).astype(np.float64), e


def _worst_ratio(X, Mn):
    x1, ea = _scaled_f16(X)
    m1, eb = _scaled_f16(Mn)
    approx = np.ldexp(x1 @ m1.T, -(ea + eb))
    true = X.astype(np.float64) @ Mn.astype(np.float64).T
    xn = np.linalg.norm(X.astype(np.float64), axis=1)
    Mmax = np.linalg.norm(Mn.astype(np.float64), axis=1).max()
    bound = 1.01 * 2.0 ** -10 * xn * Mmax
    return (np.abs(approx - true) / bound[:, None]).max()


@pytest.mark.parametrize("D", [8, 40, 100, 128])
def test_operand_rounding_bound_random(D):
    rs = np.random.RandomState(D)
    for scale in (1.0, 1e-3, 3e4):
        X = (rs.randn(600, D) * scale).astype(np.float32)
        Mn = (rs.randn(300, D) * scale).astype(np.float32)
        assert _worst_ratio(X, Mn) < 0.5


def test_operand_rounding_bound_adversarial():
    """Same-sign operands sitting just above fp16 rounding midpoints (every product errs the same way), a
    wide dynamic range that flushes most elements, and rows equal to a mean."""
    rs = np.random.RandomState(1)
    D = 128
    base = 1.0 + (2.0 ** -11) * (1 + 2.0 ** -9)               # just past the midpoint between two fp16 values
    X = (base * 2.0 ** rs.randint(0, 3, size=(200, D))).astype(np.float32)
    Mn = (base * 2.0 ** rs.randint(0, 3, size=(100, D))).astype(np.float32)
    r = _worst_ratio(X, Mn)
    assert 0.2 < r < 1.0, r                                   # the bound is nearly attained, never exceeded
    X = (rs.randn(400, D) * 10.0 ** rs.uniform(-9, 0, size=(400, D))).astype(np.float32)
    Mn = (rs.randn(150, D) * 10.0 ** rs.uniform(-9, 0, size=(150, D))).astype(np.float32)
    Mn[3] = X[5]
    assert _worst_ratio(X, Mn) < 1.0
