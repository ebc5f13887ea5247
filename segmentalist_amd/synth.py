"""
Synthetic corpora in the reference's input format.

The reference ships no data; its own tests build inputs by hand in exactly this
shape (tests/test_unigram_acoustic_wordseg.py:34-57): per utterance a matrix of
acoustic word embeddings, a triangular `vec_ids` vector (entry t(t-1)/2 + s is
the row of the embedding of span [s, t), -1 = no embedding), a `durations`
vector of the same shape (frames) and the list of landmark frames.
"""
import numpy as np


def span_table(N, n_slices_max):
    """vec_ids (local row numbering) for an utterance of N landmarks; spans longer than
    n_slices_max landmarks have no embedding (-1).  Row numbering = start-major, as the
    reference's test builder."""
    tri = N * (N + 1) // 2
    vec_ids = -1 * np.ones(tri, dtype=np.int64)
    starts = np.empty(tri, dtype=np.int64)
    ends = np.empty(tri, dtype=np.int64)
    for t in range(1, N + 1):
        i = t * (t - 1) // 2
        starts[i:i + t] = np.arange(t)
        ends[i:i + t] = t
    n = 0
    for s in range(N):
        for e in range(s, min(N, s + n_slices_max)):
            t = e + 1
            vec_ids[t * (t - 1) // 2 + s] = n
            n += 1
    return vec_ids, starts, ends, n


def make_corpus(n_utt, D, K, seed=0, N=20, ragged=False, n_slices_max=6, dtype=np.float32,
                noise=0.3, normalise=True, N_range=(8, 32)):
    """
    Returns (embedding_mats, vec_ids_dict, durations_dict, landmarks_dict).

    N fixed (throughput runs: N=20, n_slices_max=6 -> 105 spans per utterance) or, with
    ragged=True, N ~ U{N_range[0]..N_range[1]} per utterance.  Embeddings x = mu_c + noise*eps with
    K_true = max(1, K//2) cluster centres, L2-normalised, cast to `dtype`.
    """
    rs = np.random.RandomState(seed)
    K_true = max(1, K // 2)
    mu = rs.randn(K_true, D)
    embedding_mats, vec_ids_dict, durations_dict, landmarks_dict = {}, {}, {}, {}
    cache = {}
    for u in range(n_utt):
        Nu = int(rs.randint(N_range[0], N_range[1] + 1)) if ragged else N
        if Nu not in cache:
            cache[Nu] = span_table(Nu, n_slices_max)
        vec_ids, starts, ends, n_emb = cache[Nu]
        gaps = rs.randint(3, 12, size=Nu)
        landmarks = np.cumsum(gaps)
        lm0 = np.concatenate([[0], landmarks])
        durations = (lm0[ends] - lm0[starts]).astype(np.int64)
        c = rs.randint(0, K_true, size=n_emb)
        x = mu[c] + noise * rs.randn(n_emb, D)
        if normalise:
            x /= np.linalg.norm(x, axis=1, keepdims=True)
        key = "utt%06d" % u
        embedding_mats[key] = x.astype(dtype)
        vec_ids_dict[key] = vec_ids.copy()
        durations_dict[key] = durations
        landmarks_dict[key] = [int(v) for v in landmarks]
    return embedding_mats, vec_ids_dict, durations_dict, landmarks_dict
