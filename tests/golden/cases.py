"""
Seeded input builders shared by tests/golden/make_golden.py (which feeds them to the
reference) and by the tests (which feed the same inputs to the oracle and to the HIP
path).  Only *inputs* are generated here; expected outputs live in the .npz fixtures.
np.random.RandomState legacy streams are stable across numpy versions.
"""
import numpy as np

# (name, D, K, n_rows, dtype)
A1_CASES = [
    ("f32_d3_k4", 3, 4, 8, "float32"),
    ("f32_d7_k3", 7, 3, 8, "float32"),
    ("f32_d8_k5", 8, 5, 8, "float32"),
    ("f32_d39_k100", 39, 100, 8, "float32"),
    ("f32_d100_k1000", 100, 1000, 8, "float32"),
    ("f32_d129_k7", 129, 7, 6, "float32"),
    ("f32_d300_k5", 300, 5, 6, "float32"),
    ("f64_d2_k4", 2, 4, 8, "float64"),
    ("f64_d39_k100", 39, 100, 6, "float64"),
    ("f64_d100_k64", 100, 64, 6, "float64"),
    ("f64_d260_k5", 260, 5, 4, "float64"),
]


def a1_inputs(name, D, K, n, dtype):
    rs = np.random.RandomState(abs(hash_name(name)) % (2 ** 31))
    X = rs.randn(n, D)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    means = rs.randn(K, D) * 0.3
    # engineered exact ties / near ties: duplicate a row, and put one mean on a data point
    if K >= 4:
        means[K - 1] = means[1]
        means[2] = X[0]
    return X.astype(dtype), means.astype(dtype)


def hash_name(name):
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def dp_cases():
    """List of dicts(vec, N, n_min, n_max) for the three DP functions."""
    rs = np.random.RandomState(1234)
    out = []
    for N in [1, 2, 3, 4, 5, 8, 12, 20]:
        for n_max in [0, 2, 6]:
            for n_min in [0, 1]:
                for kind in ["dense", "banded", "holes", "ints", "deadend"]:
                    tri = N * (N + 1) // 2
                    vec = -rs.rand(tri) * 30.0
                    if kind == "ints":
                        vec = -rs.randint(0, 4, tri).astype(np.float64)   # many exact ties
                    for t in range(1, N + 1):
                        i = t * (t - 1) // 2
                        for s in range(t):
                            span = t - s
                            if kind in ("banded", "holes", "deadend") and span > 3:
                                vec[i + s] = -np.inf
                            if kind == "holes" and rs.rand() < 0.25:
                                vec[i + s] = -np.inf
                    if kind == "deadend" and N >= 3:
                        # nothing may end at the last landmark(s): forces back-tracking
                        t = N
                        vec[t * (t - 1) // 2: t * (t - 1) // 2 + t] = -np.inf
                    out.append(dict(vec=vec, N=N, n_min=n_min, n_max=n_max, kind=kind))
    return out


def gauss_state(D, K_max, n_items, seed, dtype=np.float32):
    """Data + a random partial assignment for component-statistics tests (A2/A3/A4)."""
    rs = np.random.RandomState(seed)
    K_true = max(2, K_max // 2)
    mu = rs.randn(K_true, D)
    z = rs.randint(0, K_true, n_items)
    X = (mu[z] + 0.3 * rs.randn(n_items, D))
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X = X.astype(dtype)
    K_used = max(1, K_max - 2)
    assign = rs.randint(0, K_used, n_items)
    assign[rs.rand(n_items) < 0.3] = -1
    # labels must be consecutive from 0
    present = sorted(set(assign.tolist()) - {-1})
    remap = {k: j for j, k in enumerate(present)}
    assign = np.array([remap.get(int(a), -1) for a in assign], dtype=np.int64)
    return X, assign


def fixed_prior_params(D):
    var = 0.002 * np.ones(D)
    mu_0 = np.zeros(D)
    var_0 = var / 0.05
    return var, mu_0, var_0


def diag_prior_params(D):
    m_0 = np.zeros(D)
    k_0 = 0.05
    v_0 = D + 3
    S_0 = 0.002 * v_0 * np.ones(D)
    return m_0, k_0, v_0, S_0


# chain-level corpora: (name, n_utt, D, K, seed, ragged, N, n_slices_max, dtype)
KMEANS_CHAINS = [
    ("km_tiny", 4, 3, 3, 11, True, 0, 4, "float32"),
    ("km_small", 12, 8, 6, 12, True, 0, 6, "float32"),
    ("km_mid", 40, 16, 12, 13, True, 0, 6, "float32"),
    ("km_f64", 8, 5, 4, 14, True, 0, 5, "float64"),
]
UNIGRAM_CHAINS = [
    ("ug_fixed", 5, 6, 4, 21, True, 0, 4, "float32", "fixed"),
    ("ug_diag", 5, 6, 4, 22, True, 0, 4, "float32", "diag"),
    ("ug_fixed_mid", 16, 10, 8, 23, True, 0, 6, "float32", "fixed"),
    ("ug_diag_mid", 16, 10, 8, 24, True, 0, 6, "float32", "diag"),
]


# FBGMM.gibbs_sample (fbgmm.py:288-420) standalone: (name, D, K_max, n_items, seed, cov, consider_unassigned,
# anneal_schedule); data and the partial initial assignment come from gauss_state(D, K_max, n_items, seed)
AM_GIBBS = [
    ("amg_fixed", 5, 6, 40, 51, "fixed", True, None),
    ("amg_diag", 5, 6, 40, 52, "diag", True, None),
    ("amg_fixed_assigned", 8, 10, 90, 53, "fixed", False, None),
    ("amg_diag_assigned", 8, 10, 90, 54, "diag", False, "linear"),
    ("amg_fixed_anneal", 6, 8, 60, 55, "fixed", True, "step"),
]
# unigram chains with intermediate acoustic-model sweeps (unigram_acoustic_wordseg.py:440-443)
AM_ITER_CHAINS = [c for c in UNIGRAM_CHAINS if c[0] in ("ug_fixed", "ug_diag_mid")]

# (name, n_utt, D, K, seed, ragged, N, n_slices_max, dtype, cov).  Only "fixed": the reference's
# del_component rewires the LM counts for fixed-variance components only
# (gaussian_components_fixedvar.py:204-221); with "diag" its LM counts go negative and it asserts.
BIGRAM_CHAINS = [
    ("bg_fixed", 6, 6, 5, 41, True, 0, 4, "float32", "fixed"),
    ("bg_fixed_mid", 16, 10, 8, 42, True, 0, 6, "float32", "fixed"),
    ("bg_fixed_k", 10, 8, 12, 43, True, 0, 5, "float32", "fixed"),
]
BIGRAM_LM = {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5}


def chain_corpus(n_utt, D, K, seed, ragged, N, n_slices_max, dtype):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from segmentalist_amd.synth import make_corpus
    # short utterances (3..9 landmarks) keep the fixtures small
    return make_corpus(n_utt, D, K, seed=seed, N=N, ragged=ragged, n_slices_max=n_slices_max,
                       dtype=np.dtype(dtype).type, N_range=(3, 9))


# ------------------------------------------------------------------ edge cases of the drivers
# (name, driver, n_utt, D, K, seed, N_range, n_slices_max, driver kwargs): very short utterances (1 and 2
# landmarks), spans shorter than min_duration (NaN durations, utterances.py:96-101), n_slices_min = 1,
# a single initial span (p_boundary_init = 0), seed boundaries snapped to landmarks (utterances.py:106-115).
EDGE_CHAINS = [
    ("e_km_short", "kmeans", 14, 6, 7, 61, (1, 5), 4, dict(p_boundary_init=0.5)),
    ("e_km_mindur", "kmeans", 12, 6, 6, 62, (2, 8), 4, dict(p_boundary_init=0.5, min_duration=9)),
    ("e_km_onespan", "kmeans", 10, 5, 5, 63, (3, 6), 6, dict(p_boundary_init=0.0)),
    ("e_ug_short", "unigram_fixed", 12, 6, 6, 64, (1, 5), 4, dict(p_boundary_init=0.5)),
    ("e_ug_mindur", "unigram_diag", 12, 6, 6, 65, (2, 8), 4, dict(p_boundary_init=0.5, min_duration=9)),
    ("e_ug_nmin1", "unigram_fixed", 12, 6, 6, 66, (3, 8), 4, dict(p_boundary_init=0.5, n_slices_min=1)),
    ("e_ug_seeded", "unigram_diag", 10, 6, 6, 67, (4, 9), 5, dict(seed_bounds=True)),
    ("e_bg_short", "bigram", 12, 6, 6, 68, (1, 6), 4, dict(p_boundary_init=0.5)),
]


def edge_corpus(n_utt, D, K, seed, N_range, n_slices_max):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from segmentalist_amd.synth import make_corpus
    return make_corpus(n_utt, D, K, seed=seed, ragged=True, n_slices_max=n_slices_max, dtype=np.float32, N_range=N_range)


def edge_seed_bounds(corpus, seed):
    """Seed boundaries in frames, deliberately off the landmarks (they get snapped)."""
    rs = np.random.RandomState(seed)
    out = {}
    for key, lm in corpus[3].items():
        n = max(1, len(lm) // 2)
        picks = sorted(set(rs.choice(len(lm), size=n, replace=False).tolist()) | {len(lm) - 1})
        out[key] = [int(lm[i]) + int(rs.randint(-1, 2)) for i in picks]
        out[key][-1] = int(lm[-1])
    return out


def edge_build(mods, case):
    """Construct the driver of an EDGE_CHAINS entry from a module namespace `mods` (the reference,
    the oracle or the product: same constructor signatures)."""
    name, driver, n_utt, D, K, seed, N_range, nmax, kw = case
    corpus = edge_corpus(n_utt, D, K, seed, N_range, nmax)
    kw = dict(kw)
    seeds = edge_seed_bounds(corpus, seed) if kw.pop("seed_bounds", False) else None
    common = dict(n_slices_min=kw.pop("n_slices_min", 0), n_slices_max=nmax, min_duration=kw.pop("min_duration", 0),
                  p_boundary_init=kw.pop("p_boundary_init", 0.5), seed_boundaries_dict=seeds)
    if driver == "kmeans":
        return mods["SegmentalKMeansWordseg"](K, *corpus, init_am_assignments="rand", wip=0, **common)
    fixed = mods["FixedVarPrior"](*fixed_prior_params(D))
    if driver == "bigram":
        return mods["BigramAcousticWordseg"](K, fixed, dict(BIGRAM_LM), *corpus, covariance_type="fixed",
                                             beta_sent_boundary=-1, lms=1.0, wip=0.0, fb_type="unigram",
                                             init_am_assignments="rand", time_power_term=1.0, **common)
    cov = driver.split("_")[1]
    prior = fixed if cov == "fixed" else mods["NIW"](*diag_prior_params(D))
    return mods["UnigramAcousticWordseg"](mods["FBGMM"], 1.0, K, prior, *corpus, covariance_type=cov,
                                          beta_sent_boundary=-1, lms=1.0, wip=0.0, fb_type="standard",
                                          init_am_assignments="rand", time_power_term=1.0, **common)
