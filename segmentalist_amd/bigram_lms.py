"""
Drop-in for segmentalist/bigram_lms.py: interpolated, additively smoothed bigram language model
over component labels.  The unigram / bigram count tables live in HBM because the sampler reads
them inside its kernels (assignment prior of every new segment, `segk_fbgmm_assign`) and
`del_component` rewires them (gaussian_components_fixedvar.py:204-221); the attributes
`unigram_counts` / `bigram_counts` are host snapshots of those tables.  The probability helpers
below are O(K) views for inspection and tests -- the sampler does not call them.
"""
import numpy as np

from .device import _dev, _torch


class BigramSmoothLM(object):
    """bigram_lms.py:17-114."""

    def __init__(self, intrp_lambda, a, b, K):
        torch = _torch()
        self.intrp_lambda = intrp_lambda
        self.a = a
        self.b = b
        self.K = K
        self._unigram = torch.zeros(K, dtype=torch.int64, device=_dev())
        self._bigram = torch.zeros((K, K), dtype=torch.int64, device=_dev())

    # snapshots --------------------------------------------------------------------------------
    @property
    def unigram_counts(self):
        return self._unigram.cpu().numpy()

    @unigram_counts.setter
    def unigram_counts(self, v):
        self._unigram.copy_(_torch().as_tensor(np.asarray(v, dtype=np.int64)))

    @property
    def bigram_counts(self):
        return self._bigram.cpu().numpy()

    @bigram_counts.setter
    def bigram_counts(self, v):
        self._bigram.copy_(_torch().as_tensor(np.asarray(v, dtype=np.int64)))

    # probabilities (bigram_lms.py:49-91) -----------------------------------------------------------
    def prob_i(self, i):
        u = self.unigram_counts
        return (u[i] + float(self.a) / self.K) / (int(np.sum(u)) + self.a)

    def prob_i_given_j(self, i, j):
        u, bg = self.unigram_counts, self.bigram_counts
        return (self.intrp_lambda * ((u[i] + float(self.a) / self.K) / (int(np.sum(u)) + self.a))
                + (1 - self.intrp_lambda) * ((bg[j, i] + float(self.b) / self.K) / (u[j] + float(self.b))))

    def log_prob_vec_i(self):
        u = self.unigram_counts
        return np.log(u + float(self.a) / self.K) - np.log(int(np.sum(u)) + self.a)

    def prob_vec_i(self):
        u = self.unigram_counts
        return (u + float(self.a) / self.K) / (int(np.sum(u)) + self.a)

    def log_prob_vec_given_j(self, j):
        return np.log(self.prob_vec_given_j(j))

    def prob_vec_given_j(self, j):
        u, bg = self.unigram_counts, self.bigram_counts
        return (self.intrp_lambda * ((u + float(self.a) / self.K) / (int(np.sum(u)) + self.a))
                + (1 - self.intrp_lambda) * (bg[j, :] + float(self.b) / self.K) / (u[j] + float(self.b)))

    # counts (bigram_lms.py:93-114) -----------------------------------------------------------------
    def _count(self, utterance, sign):
        u, bg = self.unigram_counts, self.bigram_counts
        j_prev = None
        for i_cur in utterance:
            u[i_cur] += sign
            if j_prev is not None:
                bg[j_prev, i_cur] += sign
            j_prev = i_cur
        self.unigram_counts, self.bigram_counts = u, bg

    def counts_from_data(self, data):
        for utterance in data:
            self.counts_from_utterance(utterance)

    def counts_from_utterance(self, utterance):
        self._count(utterance, 1)

    def remove_counts_from_utterance(self, utterance):
        self._count(utterance, -1)
