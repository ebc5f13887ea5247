"""
Drop-in for segmentalist/gaussian_components_diag.py: diagonal-covariance (normal-inverse-
chi-squared) components, statistics resident in HBM (`segk_fbgmm`, cov_type 1).
"""
import math

import numpy as np
from scipy.special import gammaln

from ._gauss_base import _DeviceGaussianComponents


class GaussianComponentsDiag(_DeviceGaussianComponents):
    _cov_type = 1

    def __init__(self, X, prior, assignments=None, K_max=None, _corpus=None, _alpha=1.0, _lms=1.0):
        self.prior = prior
        if K_max is None:
            K_max = X.shape[0]
        assert len(np.asarray(prior.S_0).shape) == 1, "For diagonal covariance, S_0 needs to be vector."
        self._setup(X, assignments, K_max, np.asarray(prior.S_0, np.float64), np.asarray(prior.m_0, np.float64),
                    None, prior.k_0, prior.v_0, _alpha, _lms, _corpus)

    @property
    def m_N_numerators(self):
        return self.dev.stat_a.cpu().numpy()

    @property
    def S_N_partials(self):
        return self.dev.stat_b.cpu().numpy()

    @property
    def log_prod_vars(self):
        return self.dev.log_prod.cpu().numpy()

    @property
    def inv_vars(self):
        return self.dev.pred.cpu().numpy()

    # A2 (vector API), on the device ---------------------------------------------------------------
    def log_prior(self, i):
        """gaussian_components_diag.py:215-222."""
        return self.dev.pred_vector(i)[1]

    def log_post_pred(self, i):
        """gaussian_components_diag.py:237-259."""
        return self.dev.pred_vector(i)[0]

    def log_post_pred_k(self, i, k):
        return self.log_post_pred(i)[k]

    def _snapshot(self):
        return dict(assignments=self.assignments, K=self.K, counts=self.counts, a=self.m_N_numerators,
                    b=self.S_N_partials)

    def _log_marg_k(self, k, snap, rows):
        """gaussian_components_diag.py:271-290 (record metric, host)."""
        p = self.prior
        cnt = snap["counts"][k]
        k_N = p.k_0 + cnt
        v_N = p.v_0 + cnt
        m_N = snap["a"][k] / k_N
        S_N = snap["b"][k] - k_N * np.square(m_N)
        return (-cnt * self.D / 2. * math.log(np.pi) + self.D / 2. * math.log(p.k_0) - self.D / 2. * math.log(k_N)
                + p.v_0 / 2. * np.log(p.S_0).sum() - v_N / 2. * np.log(S_N).sum()
                + self.D * (gammaln(v_N / 2.) - gammaln(p.v_0 / 2.)))
