"""
Drop-in for segmentalist/kmeans.py: the k-means acoustic model (`KMeans`).
"""
import logging
import time

import numpy as np

from . import rng
from .kmeans_components import KMeansComponents

logger = logging.getLogger(__name__)


def _consecutive(assignments):
    """Relabel so that the used labels are 0..max without gaps (kmeans.py:88-92)."""
    for k in range(assignments.max()):
        while len(np.nonzero(assignments == k)[0]) == 0:
            assignments[np.where(assignments > k)] -= 1
        if assignments.max() == k:
            break
    return assignments


class KMeans(object):
    def __init__(self, X, K, assignments="rand", _corpus=None, _shard=None):
        self._corpus = _corpus
        self._shard = _shard         # (row_lo, row_hi): the device holds a shard of X (multi-rank batch mode)
        self.setup_components(K, assignments, X)

    def setup_components(self, K, assignments="rand", X=None):
        """kmeans.py:52-94."""
        if X is None:
            assert hasattr(self, "components")
            X = self.components.X
        N, D = X.shape
        if isinstance(assignments, str) and assignments == "rand":
            assignments = np.random.randint(0, K, N)
        elif isinstance(assignments, str) and assignments == "each-in-own":
            assignments = np.arange(N)
        elif isinstance(assignments, str) and assignments == "spread":
            spread = (list(range(K)) * int(np.ceil(float(N) / K)))[:N]
            rng.shuffle(spread)
            assignments = np.array(spread)
        assignments = _consecutive(np.asarray(assignments))
        self.components = KMeansComponents(X, assignments, K, _corpus=self._corpus, _shard=getattr(self, "_shard", None))

    def fit(self, n_iter, consider_unassigned=True, no_empty=True):
        """
        kmeans.py:97-173: batch (Lloyd) iterations -- every considered item gets the argmax of
        `neg_sqrd_norm` against the means frozen at the start of the iteration, then the changed
        items are moved (del_item/add_item in item order) and empty components removed.
        """
        c = self.components
        dk = c.dev
        dk.require_whole_corpus("KMeans.fit")
        record_dict = {"sum_neg_sqrd_norm": [], "components": [], "n_mean_updates": [], "sample_time": []}
        start_time = time.time()
        for i_iter in range(n_iter):
            old = c.assignments
            items = np.arange(c.N) if consider_unassigned else np.where(old != -1)[0]
            _, new_k, _ = dk.exact_max(items.astype(np.int32)) if len(items) else (None, np.zeros(0, int), 0)
            moved = np.where(new_k != old[items])[0]
            for j in moved:
                i = int(items[j])
                dk.del_item(i)
                dk.add_item(i, int(new_k[j]))
            dk.check_status()
            dk.clean_components()

            record_dict["sum_neg_sqrd_norm"].append(c.sum_neg_sqrd_norm())
            record_dict["components"].append(c.K)
            record_dict["n_mean_updates"].append(len(moved))
            record_dict["sample_time"].append(time.time() - start_time)
            start_time = time.time()
            info = "iteration: " + str(i_iter)
            for key in sorted(record_dict):
                info += ", " + key + ": " + str(record_dict[key][-1])
            logger.info(info)
            if len(moved) == 0:
                break
        return record_dict

    def get_n_assigned(self):
        """Number of assigned items (the reference counts `assignments != -1`): the sum of the component counts, read
        from the device without copying the assignment vector to the host."""
        return int(self.components.dev.counts.sum().item())
