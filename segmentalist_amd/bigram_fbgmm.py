"""
Drop-in for segmentalist/bigram_fbgmm.py: the acoustic model of the bigram segmenter -- a bank of
Gaussian components whose assignment prior is a language model instead of a Dirichlet-multinomial.
"""
import numpy as np

from .gaussian_components_diag import GaussianComponentsDiag
from .gaussian_components_fixedvar import GaussianComponentsFixedVar
from .kmeans import _consecutive


class BigramFBGMM(object):
    """bigram_fbgmm.py:19-100."""

    def __init__(self, X, prior, K, assignments="rand", covariance_type="full", lms=1.0, lm=None, _corpus=None):
        self.prior = prior
        self.covariance_type = covariance_type
        self.lms = lms
        self._corpus = _corpus
        self.setup_components(K, assignments, X, lm)

    def setup_components(self, K, assignments="rand", X=None, lm=None):
        if X is None:
            assert hasattr(self, "components")
            X = self.components.X
        N, D = X.shape
        if isinstance(assignments, str) and assignments == "rand":
            assignments = np.random.randint(0, K, N)
        elif isinstance(assignments, str) and assignments == "each-in-own":
            assignments = np.arange(N)
        assignments = _consecutive(np.asarray(assignments))
        kw = dict(_corpus=self._corpus, _lms=self.lms)
        if self.covariance_type == "diag":
            # as in the reference the LM is not tied to diagonal components (bigram_fbgmm.py:88-89),
            # which makes its sampler assert on the first deleted component
            self.components = GaussianComponentsDiag(X, self.prior, assignments, K_max=K, **kw)
        elif self.covariance_type == "fixed":
            self.components = GaussianComponentsFixedVar(X, self.prior, assignments, K_max=K, lm=lm, **kw)
        elif self.covariance_type == "full":
            raise NotImplementedError(
                "full-covariance components are outside the accelerated hot path (SURVEY.md section 2, #8)")
        else:
            assert False, "Invalid covariance type."

    def log_prob_X_given_z(self):
        return self.components.log_marg()

    def get_n_assigned(self):
        """Number of assigned items (the reference counts `assignments != -1`): the sum of the component counts, read
        from the device without copying the assignment vector to the host."""
        return int(self.components.dev.counts.sum().item())
