"""
np_oracle.py -- numpy/py3 CPU restatement of the segmentalist hot path (class level).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package `segmentalist_amd` never
imports anything from `oracle/`.

Parity status: PINNED against (a) golden vectors captured from the reference
itself (tests/golden/make_golden.py) and (b) the constants asserted by the
reference's own tests (tests/test_oracle_golden.py).

The restatement keeps the reference's *algorithmic structure* (one numpy K x D
evaluation per candidate embedding, one DP per utterance, sequential updates of
the component statistics) so that timing it is a fair single-core stand-in for
the reference ("faithful" CPU baseline), and keeps the reference's
*floating-point evaluation order* wherever results are compared bit-for-bit
(k-means scores, max-plus DP, mean updates).

File:line citations are relative to /root/reference/segmentalist/.
"""
import math
import random as _pyrandom

import numpy as np
from scipy.special import gammaln, logsumexp as _sp_logsumexp

NEG_INF = -np.inf


# --------------------------------------------------------------------------- #
# RNG indirection: python-3 `random.shuffle` differs from python 2.
# --------------------------------------------------------------------------- #
def shuffle_py2(x):
    """Python-2.7 `random.shuffle` (one random.random() per element, reversed)."""
    for i in reversed(range(1, len(x))):
        j = int(_pyrandom.random() * (i + 1))
        x[i], x[j] = x[j], x[i]


_shuffle = _pyrandom.shuffle


def set_shuffle(kind):
    """kind: "py3" (stdlib) or "py2" (reference-era algorithm)."""
    global _shuffle
    _shuffle = shuffle_py2 if kind == "py2" else _pyrandom.shuffle


# --------------------------------------------------------------------------- #
# A9  scalar kernels (_cython_utils.pyx:13-25, 75-89; utils.py:10-21)
# --------------------------------------------------------------------------- #
def logsumexp(a):
    a = np.asarray(a, dtype=np.float64)
    mx = a[0]
    for v in a[1:]:
        if v > mx:
            mx = v
    s = np.float64(0.0)
    with np.errstate(all="ignore"):
        for v in a:
            s += np.exp(v - mx)
        return np.log(s) + mx


def draw(p_k, u=None):
    if u is None:
        u = _pyrandom.random()
    for i in range(len(p_k)):
        u = u - p_k[i]
        if u < 0:
            return i
    return len(p_k) - 1


def consecutive_labels(assignments):
    """The "make labels consecutive" fix-up (fbgmm.py:124-128, kmeans.py:88-92)."""
    for k in range(assignments.max()):
        while len(np.nonzero(assignments == k)[0]) == 0:
            assignments[np.where(assignments > k)] -= 1
        if assignments.max() == k:
            break
    return assignments


# --------------------------------------------------------------------------- #
# A13  Utterances (utterances.py:14-174) and process_embeddings
#      (unigram_acoustic_wordseg.py:571-646)
# --------------------------------------------------------------------------- #
def process_embeddings(embedding_mats, vec_ids_dict):
    rows, vec_ids, labels = [], [], []
    base = 0
    for utt in sorted(embedding_mats):
        labels.append(utt)
        src = vec_ids_dict[utt]
        cur = src.copy()
        mat = embedding_mats[utt]
        for i_row in range(len(mat)):
            rows.append(mat[i_row])
            cur[np.where(src == i_row)[0]] = base + i_row
        base += len(mat)
        vec_ids.append(cur)
    return np.asarray(rows), vec_ids, labels


class Utterances(object):
    def __init__(self, lengths, vec_ids, durations, landmarks, seed_boundaries=None,
                 p_boundary_init=0.5, n_slices_min=0, n_slices_max=6, min_duration=0):
        assert lengths == [len(i) for i in landmarks]
        self.lengths = lengths
        self.D = len(lengths)
        self.N_max = max(lengths)
        self.landmarks = landmarks
        tri = self.N_max * (self.N_max + 1) // 2
        self.vec_ids = -1 * np.ones((self.D, tri), dtype=np.int64)
        for i, v in enumerate(vec_ids):
            self.vec_ids[i, :len(v)] = v
        self.durations = np.full((self.D, tri), np.nan)
        for i, dv in enumerate(durations):
            if not (min_duration == 0 or len(dv) == 1):          # :96-101
                cur = np.array(dv, dtype=np.float64)
                cur[cur < min_duration] = np.nan
                if np.all(np.isnan(cur)):
                    cur[np.argmax(dv)] = np.max(dv)
                dv = cur
            self.durations[i, :len(dv)] = dv
        self.boundaries = np.zeros((self.D, self.N_max), dtype=bool)
        if seed_boundaries is not None:                          # :106-115
            for i_utt, bounds in enumerate(seed_boundaries):
                lm = landmarks[i_utt]
                closest = [int(np.argmin([abs(b - l) for l in lm])) for b in bounds]
                self.boundaries[i_utt, closest] = True
        elif p_boundary_init == 0:                               # :128-135
            for i in range(self.D):
                self.boundaries[i, self.lengths[i] - 1] = True
        else:                                                    # :141-157
            for i in range(self.D):
                N = self.lengths[i]
                while True:
                    self.boundaries[i, 0:N] = (np.random.rand(N) < p_boundary_init)
                    self.boundaries[i, N - 1] = True
                    if np.all(np.asarray(self.get_segmented_embeds_i(i)) == -1):
                        continue
                    spans = [b - a for a, b in self.get_segmented_landmark_indices(i)]
                    if (max(spans) <= n_slices_max and min(spans) >= n_slices_min) or N <= n_slices_min:
                        break

    def _segments(self, i):
        j_prev = 0
        for j in range(self.lengths[i]):
            if self.boundaries[i, j]:
                yield j_prev, j
                j_prev = j + 1

    def get_segmented_embeds_i(self, i):                          # :159-174
        return [self.vec_ids[i, (j + 1) * j // 2 + s] for s, j in self._segments(i)]

    def get_segmented_durations_i(self, i):
        return [self.durations[i, (j + 1) * j // 2 + s] for s, j in self._segments(i)]

    def get_segmented_landmark_indices(self, i):
        return [(s, j + 1) for s, j in self._segments(i)]


# --------------------------------------------------------------------------- #
# A1/A11  KMeansComponents (kmeans_components.py:18-266)
# --------------------------------------------------------------------------- #
class KMeansComponents(object):
    def __init__(self, X, assignments, K_max):
        self.X = X
        self.N, self.D = X.shape
        self.K_max = K_max
        self.mean_numerators = np.zeros((K_max, self.D), np.float64)
        self.counts = np.zeros(K_max, np.int64)
        self.K = 0
        assignments = np.asarray(assignments, np.int64)
        assert (self.N,) == assignments.shape
        assert set(assignments).difference([-1]) == set(range(assignments.max() + 1))
        self.assignments = -1 * np.ones(self.N, dtype=np.int64)
        self.random_means = self.X[np.random.choice(range(self.N), K_max, replace=True), :]   # :91
        self.means = self.random_means.copy()
        for k in range(assignments.max() + 1):
            for i in np.where(assignments == k)[0]:
                self.add_item(i, k)

    def add_item(self, i, k):                                     # :93-111
        assert not i == -1
        assert self.assignments[i] == -1
        if k > self.K:
            k = self.K
        if k == self.K:
            self.K += 1
        self.mean_numerators[k, :] += self.X[i]
        self.counts[k] += 1
        self.means[k, :] = self.mean_numerators[k, :] / self.counts[k]
        self.assignments[i] = k

    def del_item(self, i):                                        # :113-132
        assert not i == -1
        k = self.assignments[i]
        if k != -1:
            self.counts[k] -= 1
            self.assignments[i] = -1
            self.mean_numerators[k, :] -= self.X[i]
            if self.counts[k] != 0:
                self.means[k, :] = self.mean_numerators[k, :] / self.counts[k]

    def del_component(self, k):                                   # :149-166
        assert k < self.K
        self.K -= 1
        if k != self.K:
            self.mean_numerators[k] = self.mean_numerators[self.K]
            self.counts[k] = self.counts[self.K]
            self.means[k, :] = self.mean_numerators[self.K, :] / self.counts[self.K]
            self.assignments[np.where(self.assignments == self.K)] = k
        self.mean_numerators[self.K].fill(0.)
        self.counts[self.K] = 0
        self.means[self.K] = self.random_means[self.K]

    def neg_sqrd_norm(self, i):                                   # :225-226
        deltas = self.means - self.X[i]
        return -(deltas * deltas).sum(axis=1)

    def max_neg_sqrd_norm_i(self, i):
        return np.max(self.neg_sqrd_norm(i))

    def argmax_neg_sqrd_norm_i(self, i):
        return np.argmax(self.neg_sqrd_norm(i))

    def sum_neg_sqrd_norm(self):                                  # :234-247
        objective = 0
        for k in range(self.K):
            Xk = self.X[np.where(self.assignments == k)]
            mean = self.mean_numerators[k, :] / self.counts[k]
            deltas = mean - Xk
            objective += -np.sum(deltas * deltas)
        return objective

    def get_assignments(self, list_of_i):
        return self.assignments[np.asarray(list_of_i)]

    def get_max_assignments(self, list_of_i):
        return [self.argmax_neg_sqrd_norm_i(i) for i in list_of_i]

    def clean_components(self):                                   # :263-266
        for k in np.where(self.counts[:self.K] == 0)[0][::-1]:
            self.del_component(k)


class KMeans(object):                                             # kmeans.py:26-177
    def __init__(self, X, K, assignments="rand"):
        N = X.shape[0]
        if isinstance(assignments, str) and assignments == "rand":
            assignments = np.random.randint(0, K, N)
        elif isinstance(assignments, str) and assignments == "each-in-own":
            assignments = np.arange(N)
        elif isinstance(assignments, str) and assignments == "spread":
            lst = (list(range(K)) * int(np.ceil(float(N) / K)))[:N]
            _shuffle(lst)
            assignments = np.array(lst)
        assignments = consecutive_labels(assignments)
        self.components = KMeansComponents(X, assignments, K)

    def fit(self, n_iter, consider_unassigned=True):              # :97-173
        c = self.components
        rec = {"sum_neg_sqrd_norm": [], "components": [], "n_mean_updates": []}
        for _ in range(n_iter):
            updates = []
            for i in range(c.N):
                k_old = c.assignments[i]
                if not consider_unassigned and k_old == -1:
                    continue
                k = np.argmax(c.neg_sqrd_norm(i))
                if k != k_old:
                    updates.append((i, k))
            for i, k in updates:
                c.del_item(i)
                c.add_item(i, k)
            c.clean_components()
            rec["sum_neg_sqrd_norm"].append(c.sum_neg_sqrd_norm())
            rec["components"].append(c.K)
            rec["n_mean_updates"].append(len(updates))
            if len(updates) == 0:
                break
        return rec

    def get_n_assigned(self):
        return len(np.where(self.components.assignments != -1)[0])


# --------------------------------------------------------------------------- #
# DP functions A6/A7/A8 -- python restatements (the C twins live in
# segk_oracle.c; both are pinned to the same golden vectors).
# --------------------------------------------------------------------------- #
def _win(a, t, i, n_max):
    return a[i:i + t][-n_max:] if n_max else a[i:i + t]


def forward_backward_kmeans_viterbi(vec, N, n_slices_min=0, n_slices_max=0, i_utt=None):
    """kmeans_acoustic_wordseg.py:449-555 (n_slices_min in {0,1})."""
    boundaries = np.zeros(N, dtype=bool)
    boundaries[-1] = True
    g = np.ones(N)
    g[0] = 0.0
    i = 0
    for t in range(1, N):
        q = _win(vec, t, i, n_slices_max) + (g[:t][-n_slices_max:] if n_slices_max else g[:t])
        g[t] = -np.inf if np.all(q == -np.inf) else np.max(q)
        i += t
    t = N
    total = 0.
    while True:
        i = (t - 1) * t // 2
        q = _win(vec, t, i, n_slices_max) + (g[:t][-n_slices_max:] if n_slices_max else g[:t])
        if np.all(q == -np.inf):
            while np.all(q == -np.inf):
                t = t - 1
                if t == 0:
                    break
                i = (t - 1) * t // 2
                q = _win(vec, t, i, n_slices_max) + (g[:t][-n_slices_max:] if n_slices_max else g[:t])
            boundaries[t - 1] = True
        k = int(np.argmax(q[::-1])) + 1
        total += vec[i + t - k]
        if t - k - 1 < 0:
            break
        boundaries[t - k - 1] = True
        t = t - k
    return total, boundaries


def forward_alphas(vec, log_p_continue, N, n_slices_max=0):
    """The forward filter of forward_backward alone (unigram_acoustic_wordseg.py:684-703): alphas[t], t < N."""
    a = np.ones(N)
    a[0] = 0.0
    i = 0
    for t in range(1, N):
        q = _win(vec, t, i, n_slices_max) + (a[:t][-n_slices_max:] if n_slices_max else a[:t])
        a[t] = -np.inf if np.all(q == -np.inf) else logsumexp(q) + log_p_continue
        i += t
    return a


def forward_backward(vec, log_p_continue, N, n_slices_min=0, n_slices_max=0, i_utt=None,
                     anneal_temp=1, uniforms=None):
    """unigram_acoustic_wordseg.py:653-756.  `uniforms` (iterator) replaces random.random()."""
    boundaries = np.zeros(N, dtype=bool)
    boundaries[-1] = True
    a = np.ones(N)
    a[0] = 0.0
    i = 0
    for t in range(1, N):
        q = _win(vec, t, i, n_slices_max) + (a[:t][-n_slices_max:] if n_slices_max else a[:t])
        a[t] = -np.inf if np.all(q == -np.inf) else logsumexp(q) + log_p_continue
        i += t
    t = N
    log_prob = np.float64(0.)
    while True:
        i = (t - 1) * t // 2
        q = _win(vec, t, i, n_slices_max) + (a[:t][-n_slices_max:] if n_slices_max else a[:t])
        if np.all(q == -np.inf):
            while np.all(q == -np.inf):
                t = t - 1
                if t == 0:
                    break
                i = (t - 1) * t // 2
                q = _win(vec, t, i, n_slices_max) + (a[:t][-n_slices_max:] if n_slices_max else a[:t])
            boundaries[t - 1] = True
        with np.errstate(invalid="ignore"):
            if anneal_temp != 1:
                lq = q[::-1] - logsumexp(q)
                lqa = 1. / anneal_temp * lq - logsumexp(1. / anneal_temp * lq)
                p = np.exp(lqa)
            else:
                p = np.exp(q[::-1] - logsumexp(q))
        k = draw(p, None if uniforms is None else next(uniforms)) + 1
        log_prob += vec[i + t - k]
        if t - k - 1 < 0:
            break
        boundaries[t - k - 1] = True
        t = t - k
    assert log_prob != -np.inf
    return log_prob, boundaries


def forward_backward_viterbi(vec, log_p_continue, N, n_slices_min=0, n_slices_max=0, i_utt=None,
                             anneal_temp=None):
    """unigram_acoustic_wordseg.py:759-864."""
    boundaries = np.zeros(N, dtype=bool)
    boundaries[-1] = True
    a = np.ones(N)
    a[0] = 0.0
    i = 0
    for t in range(1, N):
        q = _win(vec, t, i, n_slices_max) + (a[:t][-n_slices_max:] if n_slices_max else a[:t])
        a[t] = -np.inf if np.all(q == -np.inf) else np.max(q)
        i += t
    t = N
    log_prob = 0.
    while True:
        i = (t - 1) * t // 2
        q = _win(vec, t, i, n_slices_max) + (a[:t][-n_slices_max:] if n_slices_max else a[:t])
        if np.all(q == -np.inf):
            while np.all(q == -np.inf):
                t = t - 1
                if t == 0:
                    break
                i = (t - 1) * t // 2
                q = _win(vec, t, i, n_slices_max) + (a[:t][-n_slices_max:] if n_slices_max else a[:t])
            boundaries[t - 1] = True
        with np.errstate(invalid="ignore"):
            p = np.exp(q[::-1] - logsumexp(q))
        k = int(np.argmax(p)) + 1
        log_prob += vec[i + t - k]
        if t - k - 1 < 0:
            break
        boundaries[t - k - 1] = True
        t = t - k
    return log_prob, boundaries


# --------------------------------------------------------------------------- #
# A12  SegmentalKMeansWordseg (kmeans_acoustic_wordseg.py:27-447)
# --------------------------------------------------------------------------- #
class SegmentalKMeansWordseg(object):
    def __init__(self, am_K, embedding_mats, vec_ids_dict, durations_dict, landmarks_dict,
                 seed_boundaries_dict=None, seed_assignments_dict=None, n_slices_min=0,
                 n_slices_max=20, min_duration=0, p_boundary_init=0.5,
                 init_am_assignments="rand", wip=0):
        assert seed_assignments_dict is None
        self.n_slices_min, self.n_slices_max, self.wip = n_slices_min, n_slices_max, wip
        embeddings, vec_ids, labels = process_embeddings(embedding_mats, vec_ids_dict)
        self.ids_to_utterance_labels = labels
        N = embeddings.shape[0]
        seeds = [seed_boundaries_dict[i] for i in labels] if seed_boundaries_dict is not None else None
        self.utterances = Utterances(
            [len(landmarks_dict[i]) for i in labels], vec_ids,
            [durations_dict[i] for i in labels], [landmarks_dict[i] for i in labels],
            seed_boundaries=seeds, p_boundary_init=p_boundary_init, n_slices_min=n_slices_min,
            n_slices_max=n_slices_max, min_duration=min_duration)
        init = []
        for i in range(self.utterances.D):
            init.extend(self.utterances.get_segmented_embeds_i(i))
        init = np.array(init, dtype=int)
        init = init[np.where(init != -1)]
        assignments = -1 * np.ones(N, dtype=int)
        if init_am_assignments == "rand":                          # :183-194
            a = consecutive_labels(np.random.randint(0, am_K, len(init)))
            assignments[init] = a
        elif init_am_assignments == "spread":                      # :196-205
            n = len(init)
            lst = (list(range(am_K)) * int(np.ceil(float(n) / am_K)))[:n]
            _shuffle(lst)
            assignments[init] = np.array(lst)
        else:
            assert False
        self.acoustic_model = KMeans(embeddings, am_K, assignments)

    def get_vec_embed_neg_len_sqrd_norms(self, vec_ids, durations):   # :334-351
        out = -np.inf * np.ones(len(vec_ids))
        c = self.acoustic_model.components
        for j, e in enumerate(vec_ids):
            if e == -1:
                continue
            out[j] = c.max_neg_sqrd_norm_i(e)
            if np.isnan(durations[j]):
                out[j] = -np.inf
            else:
                out[j] *= durations[j]
        return out + self.wip

    def segment_i(self, i):                                            # :225-332
        u, c = self.utterances, self.acoustic_model.components
        old = u.get_segmented_embeds_i(i)
        N = u.lengths[i]
        tri = (N * N + N) // 2
        vec = self.get_vec_embed_neg_len_sqrd_norms(u.vec_ids[i, :tri], u.durations[i, :tri])
        total, u.boundaries[i, :N] = forward_backward_kmeans_viterbi(
            vec, N, self.n_slices_min, self.n_slices_max, i)
        new = u.get_segmented_embeds_i(i)
        new_k = c.get_max_assignments(new)
        for e in old:
            if e == -1:
                continue
            c.del_item(e)
        for e, k in zip(new, new_k):
            c.add_item(e, k)
        c.clean_components()
        return total

    def segment(self, n_iter, n_iter_inbetween_kmeans=0):              # :353-426
        rec = {"sum_neg_sqrd_norm": [], "sum_neg_len_sqrd_norm": [], "components": [], "n_tokens": []}
        for _ in range(n_iter):
            order = list(range(self.utterances.D))
            _shuffle(order)
            tot = 0
            for i_utt in order:
                tot += self.segment_i(i_utt)
            rec["sum_neg_sqrd_norm"].append(self.acoustic_model.components.sum_neg_sqrd_norm())
            rec["sum_neg_len_sqrd_norm"].append(tot)
            rec["components"].append(self.acoustic_model.components.K)
            rec["n_tokens"].append(self.acoustic_model.get_n_assigned())
            if n_iter_inbetween_kmeans > 0:
                self.acoustic_model.fit(n_iter_inbetween_kmeans, consider_unassigned=False)
        return rec

    def get_unsup_transcript_i(self, i):
        return list(self.acoustic_model.components.get_assignments(
            self.utterances.get_segmented_embeds_i(i)))


# --------------------------------------------------------------------------- #
# Batch-synchronous k-means sweep -- the SPEC of the multi-GPU throughput mode
# (new design, DESIGN.md "batch mode"; not a reference code path).  Built from
# the reference primitives above with component statistics frozen for the whole
# sweep:
#   1. every utterance is scored + Viterbi-segmented against the frozen means
#      (segment_i steps :253-290, :313) -- independent per utterance;
#   2. all old items are deleted, all new items added in utterance-index order
#      with the reference's `k > K -> K` clamp (kmeans_components.py:103-106);
#   3. counts / mean_numerators are rebuilt FROM SCRATCH in a fixed summation
#      order: utterances are cut into `n_blocks` contiguous blocks; inside a block
#      items are summed sequentially in token order (utterance, then segment); the block partials are
#      combined by a fixed balanced binary tree.  (Bit-identical for 1/2/4/8
#      GPUs.)  means = mean_numerators / counts;
#   4. clean_components (:263-266).
# --------------------------------------------------------------------------- #
def block_bounds(n_utt, n_blocks):
    return [(b * n_utt) // n_blocks for b in range(n_blocks + 1)]


def tree_sum(parts):
    parts = list(parts)
    while len(parts) > 1:
        nxt = [parts[j] + parts[j + 1] for j in range(0, len(parts) - 1, 2)]
        if len(parts) % 2:
            nxt.append(parts[-1])
        parts = nxt
    return parts[0]


def kmeans_batch_sweep(seg, n_blocks=8):
    """One batch-synchronous sweep of a SegmentalKMeansWordseg oracle object (in place)."""
    u, c = seg.utterances, seg.acoustic_model.components
    D = u.D
    new_tokens = []          # per utterance: list of (embed_id, k_raw)
    totals = np.zeros(D)
    for i in range(D):
        N = u.lengths[i]
        tri = (N * N + N) // 2
        vec = seg.get_vec_embed_neg_len_sqrd_norms(u.vec_ids[i, :tri], u.durations[i, :tri])
        totals[i], bnd = forward_backward_kmeans_viterbi(vec, N, seg.n_slices_min, seg.n_slices_max, i)
        old = u.get_segmented_embeds_i(i)
        u.boundaries[i, :N] = bnd
        new = u.get_segmented_embeds_i(i)
        assert -1 not in new
        new_tokens.append((old, new, c.get_max_assignments(new)))
    # delete all old, add all new (assignment / K bookkeeping only)
    for old, _, _ in new_tokens:
        for e in old:
            if e != -1:
                c.assignments[e] = -1
    K = c.K
    for _, new, ks in new_tokens:
        for e, k in zip(new, ks):
            if k > K:
                k = K
            if k == K:
                K += 1
            c.assignments[e] = k
    c.K = K
    # rebuild statistics in the fixed order
    bb = block_bounds(D, n_blocks)
    part_sum, part_cnt = [], []
    for b in range(n_blocks):
        s = np.zeros((c.K_max, c.D), np.float64)
        n = np.zeros(c.K_max, np.int64)
        for i in range(bb[b], bb[b + 1]):
            for e in new_tokens[i][1]:           # token order: utterance, then segment
                k = c.assignments[e]
                s[k] += c.X[e]
                n[k] += 1
        part_sum.append(s)
        part_cnt.append(n)
    c.mean_numerators = tree_sum(part_sum)
    c.counts = tree_sum(part_cnt)
    for k in range(c.K):
        if c.counts[k] != 0:
            c.means[k] = c.mean_numerators[k] / c.counts[k]
    c.clean_components()
    blk = [np.float64(0.)] * n_blocks
    for b in range(n_blocks):
        acc = np.float64(0.)
        for i in range(bb[b], bb[b + 1]):
            acc += totals[i]
        blk[b] = acc
    return tree_sum(blk)


def minibatch_ranges(n_utt, n_blocks, n_batches):
    """Utterance ranges [lo, hi) of (statistics block b, mini-batch j): every block is cut into n_batches contiguous runs."""
    bb = block_bounds(n_utt, n_blocks)
    out = []
    for b in range(n_blocks):
        sub = block_bounds(bb[b + 1] - bb[b], n_batches)
        out.append([(bb[b] + sub[j], bb[b] + sub[j + 1]) for j in range(n_batches)])
    return out


def kmeans_minibatch_sweep(seg, n_blocks=8, n_batches=1, totals=None):
    """SPEC of the mini-batch form of the batch-synchronous sweep ("or per mini-batch of B utterances for
    fresher stats"; the reference itself refreshes the means after EVERY utterance, kmeans_acoustic_wordseg.py:314-320,
    393-399).  The sweep is n_batches steps; step j resegments, against the means as they stand at the start of the step,
    the utterances of run j of EVERY statistics block (so that every GPU has an equal share of every step), then rebuilds
    counts / mean_numerators from scratch over the CURRENT tokens of ALL utterances -- resegmented ones and not yet
    resegmented ones alike -- in the fixed order of kmeans_batch_sweep, and cleans the components.  n_batches = 1 is
    kmeans_batch_sweep.  `totals` (float64 [n_utt], kept by the caller across sweeps; default: a fresh array) holds every
    utterance's objective from the step that resegmented it; the return value is their block-sequential, tree-combined sum.
    Results do not depend on the number of GPUs."""
    u, c = seg.utterances, seg.acoustic_model.components
    D = u.D
    if totals is None:
        totals = np.zeros(D)
    bb = block_bounds(D, n_blocks)
    ranges = minibatch_ranges(D, n_blocks, n_batches)
    for j in range(n_batches):
        step = [i for b in range(n_blocks) for i in range(*ranges[b][j])]          # utterance-index order
        new_tokens = {}
        for i in step:
            N = u.lengths[i]
            tri = (N * N + N) // 2
            vec = seg.get_vec_embed_neg_len_sqrd_norms(u.vec_ids[i, :tri], u.durations[i, :tri])
            totals[i], bnd = forward_backward_kmeans_viterbi(vec, N, seg.n_slices_min, seg.n_slices_max, i)
            old = u.get_segmented_embeds_i(i)
            u.boundaries[i, :N] = bnd
            new = u.get_segmented_embeds_i(i)
            assert -1 not in new
            new_tokens[i] = (old, new, c.get_max_assignments(new))
        for i in step:                                   # delete the old items of the step's utterances ...
            for e in new_tokens[i][0]:
                if e != -1:
                    c.assignments[e] = -1
        K = c.K
        for i in step:                                   # ... add the new ones in utterance order, add_item's clamp replayed
            for e, k in zip(new_tokens[i][1], new_tokens[i][2]):
                if k > K:
                    k = K
                if k == K:
                    K += 1
                c.assignments[e] = k
        c.K = K
        # statistics from scratch over ALL utterances' current tokens, in the fixed order
        part_sum, part_cnt = [], []
        for b in range(n_blocks):
            s = np.zeros((c.K_max, c.D), np.float64)
            n = np.zeros(c.K_max, np.int64)
            for i in range(bb[b], bb[b + 1]):
                for e in u.get_segmented_embeds_i(i):    # token order: utterance, then segment
                    if e == -1:                          # a segment of the initial segmentation without an embedding
                        continue
                    k = c.assignments[e]
                    s[k] += c.X[e]
                    n[k] += 1
            part_sum.append(s)
            part_cnt.append(n)
        c.mean_numerators = tree_sum(part_sum)
        c.counts = tree_sum(part_cnt)
        for k in range(c.K):
            if c.counts[k] != 0:
                c.means[k] = c.mean_numerators[k] / c.counts[k]
        c.clean_components()
    blk = []
    for b in range(n_blocks):
        acc = np.float64(0.)
        for i in range(bb[b], bb[b + 1]):
            acc += totals[i]
        blk.append(acc)
    return tree_sum(blk)


# --------------------------------------------------------------------------- #
# A3/A11  GaussianComponentsFixedVar (gaussian_components_fixedvar.py:20-338)
# --------------------------------------------------------------------------- #
class FixedVarPrior(object):
    def __init__(self, var, mu_0, var_0):
        self.var, self.mu_0, self.var_0 = var, mu_0, var_0


class NIW(object):
    def __init__(self, m_0, k_0, v_0, S_0):
        self.m_0, self.k_0, self.v_0, self.S_0 = m_0, k_0, v_0, S_0


class _GaussBase(object):
    def _init_items(self, assignments):
        self.K = 0
        if assignments is None:
            self.assignments = -1 * np.ones(self.N, np.int64)
            return
        assignments = np.asarray(assignments, np.int64)
        assert (self.N,) == assignments.shape
        assert set(assignments).difference([-1]) == set(range(assignments.max() + 1))
        self.assignments = assignments
        for k in range(self.assignments.max() + 1):
            for i in np.where(self.assignments == k)[0]:
                self.add_item(i, k)

    def get_assignments(self, list_of_i):
        return self.assignments[np.asarray(list_of_i)]

    def log_marg(self):
        return sum((self.log_marg_k(k) for k in range(self.K)), 0.)


class GaussianComponentsFixedVar(_GaussBase):
    def __init__(self, X, prior, assignments=None, K_max=None, lm=None):
        self.X = X
        self.precision = 1. / prior.var
        self.mu_0 = prior.mu_0
        self.precision_0 = 1. / prior.var_0
        self.N, self.D = X.shape
        self.K_max = K_max
        self.lm = lm
        self.mu_N_numerators = np.zeros((K_max, self.D))
        self.precision_Ns = np.zeros((K_max, self.D))
        self.log_prod_precision_preds = np.zeros(K_max)
        self.precision_preds = np.zeros((K_max, self.D))
        self.counts = np.zeros(K_max, np.int64)
        self._c = -0.5 * self.D * math.log(2. * np.pi)
        self._init_items(assignments)

    def _update(self, k):                                           # :317-325
        pp = self.precision_Ns[k] * self.precision / (self.precision_Ns[k] + self.precision)
        self.log_prod_precision_preds[k] = np.log(pp).sum()
        self.precision_preds[k, :] = pp

    def cache_component_stats(self, k):                              # :128-141
        return (self.mu_N_numerators[k].copy(), self.precision_Ns[k].copy(), self.log_prod_precision_preds[k],
                self.precision_preds[k].copy(), self.counts[k])

    def restore_component_from_stats(self, k, a, b, lp, pp, count):   # :143-151
        self.mu_N_numerators[k, :] = a
        self.precision_Ns[k, :] = b
        self.log_prod_precision_preds[k] = lp
        self.precision_preds[k, :] = pp
        self.counts[k] = count

    def add_item(self, i, k):                                       # :153-170
        assert not i == -1
        if k == self.K:
            self.K += 1
            self.mu_N_numerators[k, :] = self.precision_0 * self.mu_0
            self.precision_Ns[k, :] = self.precision_0
        self.mu_N_numerators[k, :] += self.precision * self.X[i]
        self.precision_Ns[k, :] += self.precision
        self.counts[k] += 1
        self._update(k)
        self.assignments[i] = k

    def del_item(self, i):                                          # :172-188
        assert not i == -1
        k = self.assignments[i]
        if k != -1:
            self.counts[k] -= 1
            self.assignments[i] = -1
            if self.counts[k] == 0:
                self.del_component(k)
            else:
                self.mu_N_numerators[k, :] -= self.precision * self.X[i]
                self.precision_Ns[k, :] -= self.precision
                self._update(k)

    def del_component(self, k):                                     # :190-221
        self.K -= 1
        K = self.K
        if k != K:
            self.mu_N_numerators[k] = self.mu_N_numerators[K]
            self.precision_Ns[k, :] = self.precision_Ns[K, :]
            self.log_prod_precision_preds[k] = self.log_prod_precision_preds[K]
            self.precision_preds[k, :] = self.precision_preds[K, :]
            self.counts[k] = self.counts[K]
            self.assignments[np.where(self.assignments == K)] = k
            if self.lm is not None:
                self.lm.unigram_counts[k] = self.lm.unigram_counts[K]
                self.lm.bigram_counts[k, :] = self.lm.bigram_counts[K, :]
                self.lm.bigram_counts[:, k] = self.lm.bigram_counts[:, K]
        self.mu_N_numerators[K].fill(0.)
        self.precision_Ns[K, :].fill(0.)
        self.log_prod_precision_preds[K] = 0.
        self.precision_preds[K, :].fill(0.)
        self.counts[K] = 0
        if self.lm is not None:
            self.lm.unigram_counts[K] = 0
            self.lm.bigram_counts[K, :].fill(0)
            self.lm.bigram_counts[:, K].fill(0)

    def log_prior(self, i):                                         # :224-231, :328-338
        delta = self.X[i, :] - self.mu_0
        slog = math.log(self.precision_0[0])
        for v in self.precision_0[1:]:
            slog += math.log(v)
        ss = 0.0
        for a, b in zip(delta, self.precision_0):
            ss += a * a * b
        return self._c + 0.5 * slog - 0.5 * ss

    def log_post_pred(self, i):                                     # :242-253
        K = self.K
        deltas = self.mu_N_numerators[:K] / self.precision_Ns[:K] - self.X[i]
        return (self._c + 0.5 * self.log_prod_precision_preds[:K]
                - 0.5 * ((deltas * deltas) * self.precision_preds[:K]).sum(axis=1))

    def log_marg_k(self, k):                                        # :261-283
        X = self.X[np.where(self.assignments == k)]
        N = self.counts[k]
        return np.sum(
            (N - 1) / 2. * np.log(self.precision) - 0.5 * N * math.log(2 * np.pi)
            - 0.5 * np.log(N / self.precision_0 + 1. / self.precision)
            - 0.5 * self.precision * np.square(X).sum(axis=0)
            - 0.5 * self.precision_0 * np.square(self.mu_0)
            + 0.5 * (np.square(X.sum(axis=0)) * self.precision / self.precision_0
                     + np.square(self.mu_0) * self.precision_0 / self.precision
                     + 2 * X.sum(axis=0) * self.mu_0) / (N / self.precision_0 + 1. / self.precision))


# --------------------------------------------------------------------------- #
# A2/A11  GaussianComponentsDiag (gaussian_components_diag.py:19-360)
# --------------------------------------------------------------------------- #
class GaussianComponentsDiag(_GaussBase):
    def __init__(self, X, prior, assignments=None, K_max=None):
        self.X = X
        self.prior = prior
        self.N, self.D = X.shape
        self.K_max = self.N if K_max is None else K_max
        K_max = self.K_max
        assert len(prior.S_0.shape) == 1
        self.m_N_numerators = np.zeros((K_max, self.D))
        self.S_N_partials = np.zeros((K_max, self.D))
        self.log_prod_vars = np.zeros(K_max)
        self.inv_vars = np.zeros((K_max, self.D))
        self.counts = np.zeros(K_max, np.int64)
        self._sq_m0 = np.square(prior.m_0)
        self._sq = np.square(X)
        n = np.concatenate([[1], np.arange(1, prior.v_0 + self.N + 2)])
        self._log_v = np.log(n)
        self._gl2 = gammaln(n / 2.)
        self._log_pi = math.log(np.pi)
        self._init_items(assignments)

    def _update(self, k):                                           # :332-345
        k_N = self.prior.k_0 + self.counts[k]
        v_N = self.prior.v_0 + self.counts[k]
        m_N = self.m_N_numerators[k] / k_N
        var = (k_N + 1.) / (k_N * v_N) * (self.S_N_partials[k] - k_N * np.square(m_N))
        self.log_prod_vars[k] = np.log(var).sum()
        self.inv_vars[k, :] = 1. / var

    def cache_component_stats(self, k):                              # :137-150
        return (self.m_N_numerators[k].copy(), self.S_N_partials[k].copy(), self.log_prod_vars[k],
                self.inv_vars[k].copy(), self.counts[k])

    def restore_component_from_stats(self, k, a, b, lp, iv, count):   # :152-161
        self.m_N_numerators[k, :] = a
        self.S_N_partials[k, :] = b
        self.log_prod_vars[k] = lp
        self.inv_vars[k, :] = iv
        self.counts[k] = count

    def add_item(self, i, k):                                       # :162-177
        if k == self.K:
            self.K += 1
            self.m_N_numerators[k, :] = self.prior.k_0 * self.prior.m_0
            self.S_N_partials[k, :] = self.prior.S_0 + self.prior.k_0 * self._sq_m0
        self.m_N_numerators[k, :] += self.X[i]
        self.S_N_partials[k, :] += self._sq[i]
        self.counts[k] += 1
        self._update(k)
        self.assignments[i] = k

    def del_item(self, i):                                          # :179-194
        k = self.assignments[i]
        if k != -1:
            self.counts[k] -= 1
            self.assignments[i] = -1
            if self.counts[k] == 0:
                self.del_component(k)
            else:
                self.m_N_numerators[k, :] -= self.X[i]
                self.S_N_partials[k, :] -= self._sq[i]
                self._update(k)

    def del_component(self, k):                                     # :196-213
        self.K -= 1
        K = self.K
        if k != K:
            self.m_N_numerators[k] = self.m_N_numerators[K]
            self.S_N_partials[k, :] = self.S_N_partials[K, :]
            self.log_prod_vars[k] = self.log_prod_vars[K]
            self.inv_vars[k, :] = self.inv_vars[K, :]
            self.counts[k] = self.counts[K]
            self.assignments[np.where(self.assignments == K)] = k
        self.m_N_numerators[K].fill(0.)
        self.S_N_partials[K, :].fill(0.)
        self.log_prod_vars[K] = 0.
        self.inv_vars[K, :].fill(0.)
        self.counts[K] = 0

    def _students_t(self, i, mu, log_prod_var, inv_var, v):         # :347-360
        delta = self.X[i, :] - mu
        return (self.D * (self._gl2[v + 1] - self._gl2[v] - 0.5 * self._log_v[v] - 0.5 * self._log_pi)
                - 0.5 * log_prod_var
                - (v + 1.) / 2. * (np.log(1. + 1. / v * np.square(delta) * inv_var)).sum())

    def log_prior(self, i):                                         # :215-222
        p = self.prior
        var = (p.k_0 + 1.) / (p.k_0 * p.v_0) * p.S_0
        return self._students_t(i, p.m_0, np.log(var).sum(), 1. / var, p.v_0)

    def log_post_pred(self, i):                                     # :237-259
        K = self.K
        k_Ns = self.prior.k_0 + self.counts[:K]
        v_Ns = self.prior.v_0 + self.counts[:K]
        deltas = self.m_N_numerators[:K] / k_Ns[:, np.newaxis] - self.X[i]
        g = self._gl2[v_Ns + 1] - self._gl2[v_Ns]
        return (self.D * (g - 0.5 * self._log_v[v_Ns] - 0.5 * self._log_pi)
                - 0.5 * self.log_prod_vars[:K]
                - (v_Ns + 1) / 2. * np.einsum("ij->i", np.log(
                    1 + np.square(deltas) * self.inv_vars[:K] * (1. / v_Ns[:, np.newaxis]))))

    def log_marg_k(self, k):                                        # :271-290
        p = self.prior
        k_N = p.k_0 + self.counts[k]
        v_N = p.v_0 + self.counts[k]
        m_N = self.m_N_numerators[k] / k_N
        S_N = self.S_N_partials[k] - k_N * np.square(m_N)
        return (-self.counts[k] * self.D / 2. * self._log_pi
                + self.D / 2. * math.log(p.k_0) - self.D / 2. * math.log(k_N)
                + p.v_0 / 2. * np.log(p.S_0).sum() - v_N / 2. * np.log(S_N).sum()
                + self.D * (self._gl2[v_N] - self._gl2[p.v_0]))


# --------------------------------------------------------------------------- #
# A4/A10  FBGMM (fbgmm.py:27-494)
# --------------------------------------------------------------------------- #
def anneal_temps(n_iter, schedule, start_inv, end_inv, n_steps):
    """The three annealing schedules shared by the samplers (fbgmm.py:330-347)."""
    if schedule is None:
        return []
    if schedule == "linear":
        if n_steps == -1:
            n_steps = n_iter
        return list(1. / np.linspace(start_inv, end_inv, n_steps))
    assert schedule == "step" and n_steps != -1
    per = int(round(float(n_iter) / n_steps))
    return list(np.repeat(1. / np.linspace(start_inv, end_inv, n_steps), per))


class FBGMM(object):
    def __init__(self, X, prior, alpha, K, assignments="rand", covariance_type="full", lms=1.0):
        self.alpha, self.prior, self.covariance_type, self.lms = alpha, prior, covariance_type, lms
        N = X.shape[0]
        if isinstance(assignments, str) and assignments == "rand":
            assignments = np.random.randint(0, K, N)
        elif isinstance(assignments, str) and assignments == "each-in-own":
            assignments = np.arange(N)
        assignments = consecutive_labels(np.asarray(assignments))
        if covariance_type == "diag":
            self.components = GaussianComponentsDiag(X, prior, assignments, K_max=K)
        elif covariance_type == "fixed":
            self.components = GaussianComponentsFixedVar(X, prior, assignments, K_max=K)
        else:
            raise ValueError("full covariance is outside the hot-path scope (SURVEY section 2 #8)")

    def _logits(self, i, with_lms=True):
        c = self.components
        z = np.ones(c.K_max) * np.log(float(self.alpha) / c.K_max + c.counts)
        if with_lms:
            z = self.lms * z
        z[:c.K] += c.log_post_pred(i)
        z[c.K:] += c.log_prior(i)
        return z

    def log_prob_z(self):                                           # :208-225
        c = self.components
        return (gammaln(self.alpha) - gammaln(self.alpha + np.sum(c.counts))
                + np.sum(gammaln(c.counts + float(self.alpha) / c.K_max) - gammaln(self.alpha / c.K_max)))

    def log_prob_X_given_z(self):
        return self.components.log_marg()

    def log_marg(self):
        return self.log_prob_z() + self.log_prob_X_given_z()

    def log_marg_i(self, i):                                        # :256-285
        c = self.components
        z = self.lms * (np.log(float(self.alpha) / c.K_max + c.counts)
                        - np.log(int(np.sum(c.counts)) + self.alpha))
        z[:c.K] += c.log_post_pred(i)
        z[c.K:] += c.log_prior(i)
        return logsumexp(z)

    def gibbs_sample_inside_loop_i(self, i, anneal_temp=1, u=None):  # :422-463
        c = self.components
        z = self._logits(i)
        if anneal_temp != 1:
            z = z - _sp_logsumexp(z)
            za = 1. / anneal_temp * z - _sp_logsumexp(1. / anneal_temp * z)
            p = np.exp(za)
        else:
            p = np.exp(z - _sp_logsumexp(z))
        assert not np.isnan(np.sum(p))
        k = draw(p, u)
        if k > c.K:
            k = c.K
        c.add_item(i, k)
        return k

    def gibbs_sample(self, n_iter, consider_unassigned=True, anneal_schedule=None, anneal_start_temp_inv=0.1,
                     anneal_end_temp_inv=1, n_anneal_steps=-1):        # :288-420
        rec = {"log_marg": [], "log_prob_z": [], "log_prob_X_given_z": [], "anneal_temp": [], "components": []}
        temps = iter(anneal_temps(n_iter, anneal_schedule, anneal_start_temp_inv, anneal_end_temp_inv,
                                  n_anneal_steps))
        c = self.components
        for _ in range(n_iter):
            anneal_temp = next(temps, anneal_end_temp_inv)
            for i in range(c.N):
                k_old = c.assignments[i]
                if not consider_unassigned and k_old == -1:
                    continue
                K_old = c.K
                stats_old = c.cache_component_stats(k_old)
                c.del_item(i)
                z = self._logits(i)
                if anneal_temp != 1:
                    z = z - _sp_logsumexp(z)
                    p = np.exp(1. / anneal_temp * z - _sp_logsumexp(1. / anneal_temp * z))
                else:
                    p = np.exp(z - _sp_logsumexp(z))
                k = draw(p)
                if k > c.K:
                    k = c.K
                if k == k_old and c.K == K_old:
                    c.restore_component_from_stats(k_old, *stats_old)
                    c.assignments[i] = k_old
                else:
                    c.add_item(i, k)
            rec["log_marg"].append(self.log_marg())
            rec["log_prob_z"].append(self.log_prob_z())
            rec["log_prob_X_given_z"].append(self.log_prob_X_given_z())
            rec["anneal_temp"].append(anneal_temp)
            rec["components"].append(c.K)
        return rec

    def map_assign_i(self, i):                                      # :465-494
        c = self.components
        z = self._logits(i, with_lms=False)
        p = np.exp(z - _sp_logsumexp(z))
        k = int(np.argmax(p))
        if k > c.K:
            k = c.K
        c.add_item(i, k)
        return k

    def get_n_assigned(self):
        return len(np.where(self.components.assignments != -1)[0])


# --------------------------------------------------------------------------- #
# A12  UnigramAcousticWordseg (unigram_acoustic_wordseg.py:27-564)
# --------------------------------------------------------------------------- #
class UnigramAcousticWordseg(object):
    def __init__(self, am_class, am_alpha, am_K, am_param_prior, embedding_mats, vec_ids_dict,
                 durations_dict, landmarks_dict, seed_boundaries_dict=None, seed_assignments_dict=None,
                 covariance_type="fixed", n_slices_min=0, n_slices_max=20, min_duration=0,
                 p_boundary_init=0.5, beta_sent_boundary=2.0, lms=1., wip=0., fb_type="standard",
                 init_am_assignments="rand", time_power_term=1.):
        assert seed_assignments_dict is None
        self.n_slices_min, self.n_slices_max = n_slices_min, n_slices_max
        self.beta_sent_boundary, self.wip, self.time_power_term = beta_sent_boundary, wip, time_power_term
        self.fb_type = fb_type
        self.fb_func = {"standard": forward_backward, "viterbi": forward_backward_viterbi}[fb_type]
        embeddings, vec_ids, labels = process_embeddings(embedding_mats, vec_ids_dict)
        self.ids_to_utterance_labels = labels
        N = embeddings.shape[0]
        seeds = [seed_boundaries_dict[i] for i in labels] if seed_boundaries_dict is not None else None
        self.utterances = Utterances(
            [len(landmarks_dict[i]) for i in labels], vec_ids,
            [durations_dict[i] for i in labels], [landmarks_dict[i] for i in labels],
            seed_boundaries=seeds, p_boundary_init=p_boundary_init, n_slices_min=n_slices_min,
            n_slices_max=n_slices_max, min_duration=min_duration)
        init = []
        for i in range(self.utterances.D):
            init.extend(self.utterances.get_segmented_embeds_i(i))
        init = np.array(init, dtype=int)
        init = init[np.where(init != -1)]
        assignments = -1 * np.ones(N, dtype=int)
        assert init_am_assignments == "rand"
        assignments[init] = consecutive_labels(np.random.randint(0, am_K, len(init)))   # :210-217
        self.acoustic_model = am_class(embeddings, am_param_prior, am_alpha, am_K, assignments,
                                       covariance_type=covariance_type, lms=lms)

    def get_vec_embed_log_probs(self, vec_ids, durations):           # :474-511
        out = -np.inf * np.ones(len(vec_ids))
        for j, e in enumerate(vec_ids):
            if e == -1:
                continue
            out[j] = self.acoustic_model.log_marg_i(e)
            if np.isnan(durations[j]):
                out[j] = -np.inf
            else:
                out[j] *= durations[j] ** self.time_power_term
        return out + self.wip

    def gibbs_sample_i(self, i, anneal_temp=1, anneal_gibbs_am=False, uniforms=None):   # :252-360
        u, am = self.utterances, self.acoustic_model
        for e in u.get_segmented_embeds_i(i):
            if e == -1:
                continue
            am.components.del_item(e)
        N = u.lengths[i]
        tri = (N * N + N) // 2
        vec = self.get_vec_embed_log_probs(u.vec_ids[i, :tri], u.durations[i, :tri])
        assert self.beta_sent_boundary == -1                          # :520-521
        if self.fb_type == "standard":
            log_prob, u.boundaries[i, :N] = forward_backward(
                vec, 0.0, N, self.n_slices_min, self.n_slices_max, i, anneal_temp, uniforms=uniforms)
        else:
            log_prob, u.boundaries[i, :N] = forward_backward_viterbi(
                vec, 0.0, N, self.n_slices_min, self.n_slices_max, i, anneal_temp)
        for e in u.get_segmented_embeds_i(i):
            if e == -1:
                continue
            if self.fb_type == "standard":
                am.gibbs_sample_inside_loop_i(
                    e, anneal_temp if anneal_gibbs_am else 1,
                    None if uniforms is None else next(uniforms))
            else:
                am.map_assign_i(e)
        return log_prob

    def gibbs_sample(self, n_iter, anneal_temp=1, am_n_iter=0):       # :362-472 (no annealing schedule)
        rec = {"log_marg": [], "log_marg*length": [], "log_prob_z": [], "log_prob_X_given_z": [],
               "components": [], "n_tokens": []}
        for _ in range(n_iter):
            if am_n_iter > 0:                                            # :440-443
                self.acoustic_model.gibbs_sample(am_n_iter, consider_unassigned=False)
            order = list(range(self.utterances.D))
            _shuffle(order)
            lp = 0
            for i_utt in order:
                lp += self.gibbs_sample_i(i_utt, anneal_temp)
            am = self.acoustic_model
            rec["log_marg"].append(am.log_marg())
            rec["log_marg*length"].append(lp)
            rec["log_prob_z"].append(am.log_prob_z())
            rec["log_prob_X_given_z"].append(am.log_prob_X_given_z())
            rec["components"].append(am.components.K)
            rec["n_tokens"].append(am.get_n_assigned())
        return rec

    def get_unsup_transcript_i(self, i):
        return list(self.acoustic_model.components.get_assignments(
            self.utterances.get_segmented_embeds_i(i)))


# --------------------------------------------------------------------------- #
# Bigram variant (config 5): BigramSmoothLM (bigram_lms.py:17-114), BigramFBGMM
# (bigram_fbgmm.py:19-100), BigramAcousticWordseg (bigram_acoustic_wordseg.py:32-725).
# Only fb_type="unigram" works in the reference (the bigram DP is a stub, :694-695).
# --------------------------------------------------------------------------- #
class BigramSmoothLM(object):
    def __init__(self, intrp_lambda, a, b, K):
        self.intrp_lambda, self.a, self.b, self.K = intrp_lambda, a, b, K
        self.unigram_counts = np.zeros(K, np.int64)
        self.bigram_counts = np.zeros((K, K), np.int64)

    def prob_i(self, i):
        return (self.unigram_counts[i] + float(self.a) / self.K) / (int(np.sum(self.unigram_counts)) + self.a)

    def prob_i_given_j(self, i, j):
        p = (self.bigram_counts[j, i] + float(self.b) / self.K) / (self.unigram_counts[j] + float(self.b))
        return self.intrp_lambda * self.prob_i(i) + (1 - self.intrp_lambda) * p

    def log_prob_vec_i(self):                                         # :64-69
        return (np.log(self.unigram_counts + float(self.a) / self.K)
                - np.log(int(np.sum(self.unigram_counts)) + self.a))

    def prob_vec_i(self):
        return (self.unigram_counts + float(self.a) / self.K) / (int(np.sum(self.unigram_counts)) + self.a)

    def prob_vec_given_j(self, j):                                    # :84-91
        return (self.intrp_lambda * self.prob_vec_i() + (1 - self.intrp_lambda)
                * (self.bigram_counts[j, :] + float(self.b) / self.K) / (self.unigram_counts[j] + float(self.b)))

    def counts_from_utterance(self, utterance):                       # :98-105
        j_prev = None
        for i_cur in utterance:
            self.unigram_counts[i_cur] += 1
            if j_prev is not None:
                self.bigram_counts[j_prev, i_cur] += 1
            j_prev = i_cur

    def remove_counts_from_utterance(self, utterance):                # :107-114
        j_prev = None
        for i_cur in utterance:
            self.unigram_counts[i_cur] -= 1
            if j_prev is not None:
                self.bigram_counts[j_prev, i_cur] -= 1
            j_prev = i_cur


class BigramFBGMM(object):
    def __init__(self, X, prior, K, assignments="rand", covariance_type="fixed", lms=1.0, lm=None):
        self.prior, self.covariance_type, self.lms = prior, covariance_type, lms
        N = X.shape[0]
        if isinstance(assignments, str) and assignments == "rand":
            assignments = np.random.randint(0, K, N)
        elif isinstance(assignments, str) and assignments == "each-in-own":
            assignments = np.arange(N)
        assignments = consecutive_labels(np.asarray(assignments))
        if covariance_type == "diag":
            self.components = GaussianComponentsDiag(X, prior, assignments, K_max=K)
        elif covariance_type == "fixed":
            self.components = GaussianComponentsFixedVar(X, prior, assignments, K_max=K, lm=lm)
        else:
            raise ValueError("full covariance is outside the hot-path scope")

    def log_prob_X_given_z(self):
        return self.components.log_marg()

    def get_n_assigned(self):
        return len(np.where(self.components.assignments != -1)[0])


class BigramAcousticWordseg(object):
    def __init__(self, am_K, am_param_prior, lm_params, embedding_mats, vec_ids_dict, durations_dict,
                 landmarks_dict, seed_boundaries_dict=None, seed_assignments_dict=None, covariance_type="fixed",
                 n_slices_min=0, n_slices_max=20, min_duration=0, p_boundary_init=0.5, beta_sent_boundary=2.0,
                 lms=1., wip=0., fb_type="bigram", init_am_assignments="rand", time_power_term=1.):
        assert seed_assignments_dict is None
        assert fb_type == "unigram", "the bigram DP is a stub in the reference"
        self.n_slices_min, self.n_slices_max = n_slices_min, n_slices_max
        self.beta_sent_boundary, self.wip, self.lms = beta_sent_boundary, wip, lms
        self.time_power_term, self.fb_type = time_power_term, fb_type
        embeddings, vec_ids, labels = process_embeddings(embedding_mats, vec_ids_dict)
        self.ids_to_utterance_labels = labels
        N = embeddings.shape[0]
        seeds = [seed_boundaries_dict[i] for i in labels] if seed_boundaries_dict is not None else None
        self.utterances = Utterances(
            [len(landmarks_dict[i]) for i in labels], vec_ids, [durations_dict[i] for i in labels],
            [landmarks_dict[i] for i in labels], seed_boundaries=seeds, p_boundary_init=p_boundary_init,
            n_slices_min=n_slices_min, n_slices_max=n_slices_max, min_duration=min_duration)
        init = []
        for i in range(self.utterances.D):
            init.extend(self.utterances.get_segmented_embeds_i(i))
        init = np.array(init, dtype=int)
        init = init[np.where(init != -1)]
        assert lm_params["type"] == "smooth"
        self.lm = BigramSmoothLM(lm_params["intrp_lambda"], lm_params["a"], lm_params["b"], am_K)
        assignments = -1 * np.ones(N, dtype=int)
        assert init_am_assignments == "rand"
        assignments[init] = consecutive_labels(np.random.randint(0, am_K, len(init)))
        self.acoustic_model = BigramFBGMM(embeddings, am_param_prior, am_K, assignments,
                                          covariance_type=covariance_type, lms=lms, lm=self.lm)
        for i_utt in range(self.utterances.D):                         # set_lm_counts :271-276
            self.lm.counts_from_utterance(self.get_unsup_transcript_i(i_utt))

    def get_unsup_transcript_i(self, i):
        return list(self.acoustic_model.components.get_assignments(self.utterances.get_segmented_embeds_i(i)))

    def log_prob_z(self):                                               # :287-305
        tmp = BigramSmoothLM(self.lm.intrp_lambda, self.lm.a, self.lm.b, self.lm.K)
        lp = 0.
        for i_utt in range(self.utterances.D):
            j_prev = None
            for i_cur in self.get_unsup_transcript_i(i_utt):
                if j_prev is not None:
                    lp += np.log(tmp.prob_i_given_j(i_cur, j_prev))
                    tmp.bigram_counts[j_prev, i_cur] += 1
                else:
                    lp += np.log(tmp.prob_i(i_cur))
                tmp.unigram_counts[i_cur] += 1
                # NB: the reference never updates j_prev here (:298-304), so the bigram branch is
                # dead code and this is a unigram predictive likelihood -- kept as is.
        return lp

    def log_marg(self):
        return self.log_prob_z() + self.acoustic_model.log_prob_X_given_z()

    def log_marg_i_embed_unigram(self, e):                              # :314-329
        c = self.acoustic_model.components
        z = self.lms * self.lm.log_prob_vec_i()
        z[:c.K] += c.log_post_pred(e)
        z[c.K:] += c.log_prior(e)
        return logsumexp(z)

    def gibbs_sample_inside_loop_i_embed(self, e, j_prev=None, anneal_temp=1, u=None):   # :332-384
        c = self.acoustic_model.components
        if j_prev is not None:
            z = np.log(self.lm.prob_vec_given_j(j_prev))
        else:
            z = self.lm.log_prob_vec_i()
        z = z * self.lms
        z[:c.K] += c.log_post_pred(e)
        z[c.K:] += c.log_prior(e)
        if anneal_temp != 1:
            z = z - logsumexp(z)
            za = 1. / anneal_temp * z - logsumexp(1. / anneal_temp * z)
            p = np.exp(za)
        else:
            p = np.exp(z - logsumexp(z))
        k = draw(p, u)
        if k > c.K:
            k = c.K
        c.add_item(e, k)
        return k

    def get_vec_embed_log_probs(self, vec_ids, durations):             # :673-692
        out = -np.inf * np.ones(len(vec_ids))
        for j, e in enumerate(vec_ids):
            if e == -1:
                continue
            out[j] = self.log_marg_i_embed_unigram(e)
            if np.isnan(durations[j]):
                out[j] = -np.inf
            else:
                out[j] *= durations[j] ** self.time_power_term
        return out + self.wip

    def gibbs_sample_i(self, i, anneal_temp=1, anneal_gibbs_am=False, uniforms=None):   # :386-551
        u, am = self.utterances, self.acoustic_model
        self.lm.remove_counts_from_utterance(self.get_unsup_transcript_i(i))
        for e in u.get_segmented_embeds_i(i):
            if e == -1:
                continue
            am.components.del_item(e)
        N = u.lengths[i]
        tri = (N * N + N) // 2
        vec = self.get_vec_embed_log_probs(u.vec_ids[i, :tri], u.durations[i, :tri])
        assert self.beta_sent_boundary == -1
        log_prob, u.boundaries[i, :N] = forward_backward(
            vec, 0.0, N, self.n_slices_min, self.n_slices_max, i, anneal_temp, uniforms=uniforms)
        j_prev = None
        for e in u.get_segmented_embeds_i(i):
            if e == -1:
                continue
            j_prev = self.gibbs_sample_inside_loop_i_embed(
                e, j_prev, anneal_temp if anneal_gibbs_am else 1, None if uniforms is None else next(uniforms))
        self.lm.counts_from_utterance(self.get_unsup_transcript_i(i))
        return log_prob

    def gibbs_sample(self, n_iter, anneal_temp=1):                      # :553-671
        rec = {"log_marg": [], "log_marg*length": [], "log_prob_z": [], "log_prob_X_given_z": [],
               "components": [], "n_tokens": []}
        for _ in range(n_iter):
            order = list(range(self.utterances.D))
            _shuffle(order)
            lp = 0
            for i_utt in order:
                lp += self.gibbs_sample_i(i_utt, anneal_temp)
            rec["log_marg"].append(self.log_marg())
            rec["log_marg*length"].append(lp)
            rec["log_prob_z"].append(self.log_prob_z())
            rec["log_prob_X_given_z"].append(self.acoustic_model.log_prob_X_given_z())
            rec["components"].append(self.acoustic_model.components.K)
            rec["n_tokens"].append(self.acoustic_model.get_n_assigned())
        return rec
