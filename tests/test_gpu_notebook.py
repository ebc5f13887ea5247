"""
BASELINE.json configs[0]: the example notebook's toy data (examples/clustering_examples.ipynb: N = 100, D = 2
float64, K = 4) through the PRODUCT's `KMeans.fit` and `FBGMM.gibbs_sample` on the device, against the
trajectories captured from the reference (tests/golden/notebook.npz) and the log lines the notebook publishes
(ipynb:172-175, 272-280).
"""
import random

import numpy as np
import numpy.testing as npt
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    torch.cuda.set_device(0)
    from segmentalist_amd import _abi
    _abi.ctx()
    return torch


def test_notebook_kmeans_fit_through_the_product(gpu, golden):
    """ipynb:272-280: KMeans(X, 4, "spread").fit(20) converges in nine iterations; the objective and the number of
    mean updates per iteration are published in the notebook (Python-2 shuffle: segmentalist_amd.rng)."""
    from segmentalist_amd import rng
    from segmentalist_amd.kmeans import KMeans
    g = golden("notebook")
    X = g["X"]
    np.random.set_state(("MT19937", g["np_state_keys"], int(g["np_state_pos"][0]), int(g["np_state_pos"][1]),
                         float(g["np_state_gauss"])))
    random.setstate((3, tuple(int(v) for v in g["py_state"]), None))
    rng.set_shuffle("py2")
    try:
        km = KMeans(X, 4, "spread")
    finally:
        rng.set_shuffle("py3")
    c = km.components
    assert np.array_equal(c.assignments, g["kmeans_init_assign"])
    assert np.array_equal(c.random_means, g["kmeans_random_means"])
    rec = km.fit(20)
    published = [-618.585465615, -223.041596617, -220.219963349, -219.615938349, -207.450606173,
                 -126.321787187, -109.921387903, -108.302238117, -108.302238117]
    assert rec["n_mean_updates"] == [69, 18, 1, 1, 4, 11, 4, 1, 0]
    npt.assert_allclose(rec["sum_neg_sqrd_norm"], published, rtol=0, atol=5e-10)
    # the record metric sums per-token terms on the device in a different order than the reference's per-component
    # numpy sums: equal to the captured values to rounding, the state itself bit for bit
    npt.assert_allclose(rec["sum_neg_sqrd_norm"], g["kmeans_sum_neg_sqrd_norm"], rtol=1e-12)
    assert rec["components"] == [4] * 9
    assert np.array_equal(c.assignments, g["kmeans_final_assign"])
    assert np.array_equal(c.means, g["kmeans_final_means"])
    assert c.means.dtype == np.float64


def test_notebook_fbgmm_gibbs_through_the_product(gpu, golden, monkeypatch):
    """ipynb:150-191: FBGMM(X, prior, alpha=1, K=4, "rand", fixed covariance).gibbs_sample(20); the first log_marg
    values are published (-692.428422807, -637.265652593, -507.64571257, -439.574417911).  The uniforms the reference
    consumed are replayed."""
    from segmentalist_amd import fbgmm
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    g = golden("notebook")
    X = g["X"]
    D = 2
    mu_scale, covar_scale, var_scale = 4.0, 0.7, 0.5
    k_0 = covar_scale ** 2 / mu_scale ** 2
    var = covar_scale ** 2 * np.ones(D) * var_scale
    prior = FixedVarPrior(var, np.zeros(D), var / k_0)
    fm = fbgmm.FBGMM(X, prior, 1., 4, g["fbgmm_init_assign"].copy(), covariance_type="fixed")
    assert np.array_equal(fm.components.assignments, g["fbgmm_init_assign"])
    stream = iter(g["fbgmm_uniforms"])
    monkeypatch.setattr(random, "random", lambda: float(next(stream)))
    rec = fm.gibbs_sample(20)
    npt.assert_allclose(rec["log_marg"], g["fbgmm_log_marg"], rtol=1e-10)
    npt.assert_allclose(rec["log_marg"][:4], [-692.428422807, -637.265652593, -507.64571257, -439.574417911], rtol=0,
                        atol=5e-9)
    assert np.array_equal(fm.components.assignments, g["fbgmm_final_assign"])
