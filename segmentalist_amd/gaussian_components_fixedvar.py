"""
Drop-in for segmentalist/gaussian_components_fixedvar.py: fixed-variance diagonal Gaussian
components of a Bayesian GMM, statistics resident in HBM (include/segk.h `segk_fbgmm`,
cov_type 0).  Hot-path methods (log_post_pred, log_prior, add/del_item, del_component) run in
libsegk.so; `log_marg_k` / `log_marg` are per-sweep record metrics evaluated on the host from
snapshots of the device statistics.
"""
import math

import numpy as np

from ._gauss_base import _DeviceGaussianComponents


class FixedVarPrior(object):
    """gaussian_components_fixedvar.py:349-357."""

    def __init__(self, var, mu_0, var_0):
        self.var = var
        self.mu_0 = mu_0
        self.var_0 = var_0


class GaussianComponentsFixedVar(_DeviceGaussianComponents):
    _cov_type = 0

    def __init__(self, X, prior, assignments=None, K_max=None, lm=None, _corpus=None, _alpha=1.0, _lms=1.0):
        assert K_max is not None, "always require `K_max`"          # as the reference (:89-91)
        self.precision = 1. / np.asarray(prior.var, dtype=np.float64)
        self.mu_0 = np.asarray(prior.mu_0, dtype=np.float64)
        self.precision_0 = 1. / np.asarray(prior.var_0, dtype=np.float64)
        self.lm = lm          # bigram_lms.BigramSmoothLM whose counts follow del_component (:204-221)
        self._setup(X, assignments, K_max, self.precision, self.mu_0, self.precision_0, 0.0, 0.0, _alpha, _lms,
                    _corpus, lm=lm)

    # statistics (host snapshots, reference names)
    @property
    def mu_N_numerators(self):
        return self.dev.stat_a.cpu().numpy()

    @property
    def precision_Ns(self):
        return self.dev.stat_b.cpu().numpy()

    @property
    def log_prod_precision_preds(self):
        return self.dev.log_prod.cpu().numpy()

    @property
    def precision_preds(self):
        return self.dev.pred.cpu().numpy()

    # A3 ----------------------------------------------------------------------------------
    def log_post_pred(self, i):
        """gaussian_components_fixedvar.py:242-253 (vector over the K active components)."""
        return self._logits_parts(i)[0]

    def log_prior(self, i):
        """gaussian_components_fixedvar.py:224-231."""
        return self._logits_parts(i)[1]

    def log_post_pred_k(self, i, k):
        return self.log_post_pred(i)[k]

    def _logits_parts(self, i):
        return self.dev.pred_vector(i)

    def _snapshot(self):
        return dict(assignments=self.assignments, K=self.K, counts=self.counts)

    def _log_marg_k(self, k, snap, rows):
        """gaussian_components_fixedvar.py:261-283 (record metric, host)."""
        X = self.X[rows]
        N = snap["counts"][k]
        p, p0, m0 = self.precision, self.precision_0, self.mu_0
        return np.sum(
            (N - 1) / 2. * np.log(p) - 0.5 * N * math.log(2 * np.pi) - 0.5 * np.log(N / p0 + 1. / p)
            - 0.5 * p * np.square(X).sum(axis=0) - 0.5 * p0 * np.square(m0)
            + 0.5 * (np.square(X.sum(axis=0)) * p / p0 + np.square(m0) * p0 / p + 2 * X.sum(axis=0) * m0)
            / (N / p0 + 1. / p))
