"""
Drop-in for segmentalist/kmeans_components.py with the state in HBM.

`KMeansComponents` keeps the reference's attributes (`X`, `N`, `D`, `K_max`, `K`, `means`,
`mean_numerators`, `counts`, `random_means`, `assignments`) and methods; every method is a
call into libsegk.so (include/segk.h).  Array attributes are read-only host snapshots of the
device buffers, fetched on access.

dtype contract (measured on the reference): `means` has the dtype of X
(kmeans_components.py:75-76), so with float32 embeddings `neg_sqrd_norm` is float32
arithmetic end to end; the device kernels reproduce it bit for bit.
"""
import logging

import numpy as np

from .device import DeviceCorpus, DeviceKMeans

logger = logging.getLogger(__name__)


class KMeansComponents(object):
    def __init__(self, X, assignments, K_max, _corpus=None, _shard=None):
        self.X = X
        self.N, self.D = X.shape
        self.K_max = K_max

        assignments = np.asarray(assignments, np.int64)
        assert (self.N,) == assignments.shape
        # apart from unassigned (-1), components are labelled from 0 (kmeans_components.py:68)
        assert set(assignments).difference([-1]) == set(range(assignments.max() + 1))

        self.setup_random_means()
        corpus = _corpus if _corpus is not None else DeviceCorpus(X)
        # device: counts / mean_numerators / means / K from the assignments, summed in the order
        # of the reference's add_item loop (:79-81)
        # _shard = (row_lo, row_hi): `corpus` holds only those rows of X (numbered from 0); the initial statistics then come
        # from the host, which has all of X
        self.dev = DeviceKMeans(corpus, K_max, assignments, self.random_means,
                                shard=None if _shard is None else (int(_shard[0]), int(_shard[1]), X))

    def setup_random_means(self):
        # kmeans_components.py:90-91 (consumes np.random exactly like the reference)
        self.random_means = self.X[np.random.choice(range(self.N), self.K_max, replace=True), :]

    # ---------------------------------------------------------------- state snapshots
    @property
    def K(self):
        return int(self.dev.K.item())

    @property
    def means(self):
        return self.dev.means.cpu().numpy()

    @property
    def mean_numerators(self):
        return self.dev.mean_numerators.cpu().numpy()

    @property
    def counts(self):
        return self.dev.counts.cpu().numpy()

    @property
    def assignments(self):
        return self.dev.global_assignments().astype(np.int64)

    # ---------------------------------------------------------------- mutators (A11)
    def add_item(self, i, k):
        """kmeans_components.py:93-111."""
        assert not i == -1
        self.dev.add_item(i, k)
        self.dev.check_status()

    def del_item(self, i):
        """kmeans_components.py:113-132."""
        assert not i == -1
        self.dev.del_item(i)

    def del_component(self, k):
        """kmeans_components.py:149-166."""
        assert k < self.K
        self.dev.del_component(k)

    def clean_components(self):
        """kmeans_components.py:263-266."""
        self.dev.clean_components()

    # ---------------------------------------------------------------- scores (A1)
    def neg_sqrd_norm(self, i):
        """kmeans_components.py:169-226: vector over all K_max rows, dtype of X."""
        return self.dev.neg_sqrd_norm(i)

    def max_neg_sqrd_norm_i(self, i):
        mx, _, _ = self.dev.exact_max([i])
        return self.X.dtype.type(mx[0])

    def argmax_neg_sqrd_norm_i(self, i):
        _, am, _ = self.dev.exact_max([i])
        return int(am[0])

    def sum_neg_sqrd_norm(self):
        """kmeans_components.py:234-247 (record metric)."""
        return self.dev.sum_neg_sqrd_norm()

    def get_assignments(self, list_of_i):
        return self.assignments[np.asarray(list_of_i)]

    def get_max_assignments(self, list_of_i):
        """kmeans_components.py:256-261, one fused device call for the whole list."""
        list_of_i = list(list_of_i)
        if not list_of_i:
            return []
        # python's X[-1] semantics for a -1 id (kmeans_components.py:225)
        ids = [i if i >= 0 else self.N + i for i in list_of_i]
        _, am, _ = self.dev.exact_max(ids)
        return [int(k) for k in am]
