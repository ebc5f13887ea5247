"""Shared host-side shell of the device-backed Gaussian component classes."""
import numpy as np

from .device import DeviceCorpus, DeviceFbgmm


class _DeviceGaussianComponents(object):
    """Attributes and methods common to GaussianComponentsFixedVar / GaussianComponentsDiag.
    The statistics live in HBM (`self.dev`); array attributes are host snapshots."""

    _cov_type = None

    def _setup(self, X, assignments, K_max, prior_a, prior_b, prior_c, k_0, v_0, alpha=1.0, lms=1.0,
               _corpus=None, lm=None):
        self.X = X
        self.N, self.D = X.shape
        self.K_max = K_max
        if assignments is None:
            assignments = -1 * np.ones(self.N, np.int64)
        else:
            assignments = np.asarray(assignments, np.int64)
            assert (self.N,) == assignments.shape
            # apart from unassigned (-1), components are labelled from 0
            assert set(assignments).difference([-1]) == set(range(assignments.max() + 1))
        corpus = _corpus if _corpus is not None else DeviceCorpus(X)
        self.dev = DeviceFbgmm(corpus, self._cov_type, K_max, alpha, lms, prior_a, prior_b, prior_c, k_0, v_0,
                               assignments, lm=lm)

    # state snapshots -------------------------------------------------------------------
    @property
    def K(self):
        return int(self.dev.K.item())

    @property
    def counts(self):
        return self.dev.counts.cpu().numpy()

    @property
    def assignments(self):
        return self.dev.assignments.cpu().numpy().astype(np.int64)

    # mutators (A11) ------------------------------------------------------------------------
    def add_item(self, i, k):
        assert not i == -1
        self.dev.update(1, item=i, k=k)

    def del_item(self, i):
        assert not i == -1
        self.dev.update(2, item=i)

    def del_component(self, k):
        self.dev.update(4, k=k)

    def get_assignments(self, list_of_i):
        return self.assignments[np.asarray(list_of_i)]

    def log_marg(self):
        """Sum of log_marg_k over the active components (record metric, host).  One snapshot of
        the device state is shared by all components and the rows of a component are found by one
        stable sort of the assignments instead of a scan per component (O(N log N + N D) for the
        whole call; the values are those of the reference's expression, row order included)."""
        snap = self._snapshot()
        a = snap["assignments"]
        order = np.argsort(a, kind="stable")
        bounds = np.searchsorted(a[order], np.arange(snap["K"] + 1))
        total = 0.
        for k in range(snap["K"]):
            total += self._log_marg_k(k, snap, order[bounds[k]:bounds[k + 1]])
        return total

    def log_marg_k(self, k):
        snap = self._snapshot()
        return self._log_marg_k(k, snap, np.where(snap["assignments"] == k)[0])
