"""
Drop-in for segmentalist/fbgmm.py: finite Bayesian Gaussian mixture model, device backed.
"""
import logging
import random
import time

import numpy as np
from scipy.special import gammaln

from .gaussian_components_diag import GaussianComponentsDiag
from .gaussian_components_fixedvar import GaussianComponentsFixedVar
from .kmeans import _consecutive

logger = logging.getLogger(__name__)


class FBGMM(object):
    def __init__(self, X, prior, alpha, K, assignments="rand", covariance_type="full", lms=1.0, _corpus=None):
        self.alpha = alpha
        self.prior = prior
        self.covariance_type = covariance_type
        self.lms = lms
        self._corpus = _corpus
        self.setup_components(K, assignments, X)

    def setup_components(self, K, assignments="rand", X=None):
        """fbgmm.py:93-137."""
        if X is None:
            assert hasattr(self, "components")
            X = self.components.X
        N, D = X.shape
        if isinstance(assignments, str) and assignments == "rand":
            assignments = np.random.randint(0, K, N)
        elif isinstance(assignments, str) and assignments == "each-in-own":
            assignments = np.arange(N)
        assignments = _consecutive(np.asarray(assignments))
        kw = dict(_corpus=self._corpus, _alpha=self.alpha, _lms=self.lms)
        if self.covariance_type == "diag":
            self.components = GaussianComponentsDiag(X, self.prior, assignments, K_max=K, **kw)
        elif self.covariance_type == "fixed":
            self.components = GaussianComponentsFixedVar(X, self.prior, assignments, K_max=K, **kw)
        elif self.covariance_type == "full":
            raise NotImplementedError(
                "full-covariance components are outside the accelerated hot path (SURVEY.md section 2, #8)")
        else:
            assert False, "Invalid covariance type."

    # record metrics (host, from device snapshots) ------------------------------------------------
    def log_prob_z(self):
        """fbgmm.py:208-225."""
        counts = self.components.counts
        K_max = self.components.K_max
        return (gammaln(self.alpha) - gammaln(self.alpha + np.sum(counts))
                + np.sum(gammaln(counts + float(self.alpha) / K_max) - gammaln(self.alpha / K_max)))

    def log_prob_X_given_z(self):
        return self.components.log_marg()

    def log_marg(self):
        return self.log_prob_z() + self.log_prob_X_given_z()

    # hot path --------------------------------------------------------------------------------------
    def log_marg_i(self, i):
        """fbgmm.py:256-285 (A4), on the device."""
        assert i != -1
        return float(self.components.dev.log_marg_rows([i])[0])

    def gibbs_sample_inside_loop_i(self, i, anneal_temp=1):
        """fbgmm.py:422-463 (A10); consumes one random.random() like the reference."""
        self.components.dev.assign_item(i, random.random(), anneal_temp, map_assign=False)

    def map_assign_i(self, i):
        """fbgmm.py:465-494."""
        self.components.dev.assign_item(i, 0.0, 1.0, map_assign=True)

    def gibbs_sample(self, n_iter, consider_unassigned=True, anneal_schedule=None, anneal_start_temp_inv=0.1,
                     anneal_end_temp_inv=1, n_anneal_steps=-1):
        """fbgmm.py:288-420 on the device: per iteration one kernel walks the items in index order
        (cache statistics, del_item, logits, draw, restore or add_item); the uniforms are the
        `random.random()` values the reference would consume -- one per considered item."""
        record_dict = {k: [] for k in ["sample_time", "log_marg", "log_prob_z", "log_prob_X_given_z",
                                       "anneal_temp", "components"]}
        start_time = time.time()
        if anneal_schedule is None:
            get_anneal_temp = iter([])
        elif anneal_schedule == "linear":
            if n_anneal_steps == -1:
                n_anneal_steps = n_iter
            get_anneal_temp = iter(1. / np.linspace(anneal_start_temp_inv, anneal_end_temp_inv, n_anneal_steps))
        elif anneal_schedule == "step":
            assert not n_anneal_steps == -1, "`n_anneal_steps` of -1 not allowed for step annealing schedule"
            n_iter_per_step = int(round(float(n_iter) / n_anneal_steps))
            anneal_list = 1. / np.linspace(anneal_start_temp_inv, anneal_end_temp_inv, n_anneal_steps)
            get_anneal_temp = iter(np.repeat(anneal_list, n_iter_per_step))
        else:
            assert False, "invalid anneal_schedule"
        c = self.components
        dev = c.dev
        for i_iter in range(n_iter):
            anneal_temp = next(get_anneal_temp, anneal_end_temp_inv)
            n_draws = c.N if consider_unassigned else int(np.count_nonzero(c.assignments != -1))
            used = dev.gibbs_items([random.random() for _ in range(n_draws)], consider_unassigned, anneal_temp)
            dev.check_status()
            assert used == n_draws
            record_dict["sample_time"].append(time.time() - start_time)
            start_time = time.time()
            record_dict["log_marg"].append(self.log_marg())
            record_dict["log_prob_z"].append(self.log_prob_z())
            record_dict["log_prob_X_given_z"].append(self.log_prob_X_given_z())
            record_dict["anneal_temp"].append(anneal_temp)
            record_dict["components"].append(c.K)
            info = "iteration: " + str(i_iter)
            for key in sorted(record_dict):
                info += ", " + key + ": " + str(record_dict[key][-1])
            logger.info(info)
        return record_dict

    def get_n_assigned(self):
        """Number of assigned items (the reference counts `assignments != -1`): the sum of the component counts, read
        from the device without copying the assignment vector to the host."""
        return int(self.components.dev.counts.sum().item())
