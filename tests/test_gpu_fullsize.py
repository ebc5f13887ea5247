"""FBGMM / bigram drivers at BASELINE sizes (configs[1]: 1 000 utterances, D = 39, K = 100; configs[4]:
10 000 utterances, D = 100, K = 1 000 -- the whole corpus on one GPU), where the oracle is too slow to
replay a sweep: sampled span scores against the specification's log_marg_i on the same state, and size-
independent properties -- conservation of tokens, statistics equal to a from-scratch recount,
language-model tables equal to a recount of the transcripts, valid segmentations, determinism."""
import random

import numpy as np
import numpy.testing as npt
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    torch.cuda.set_device(0)
    from segmentalist_amd import _abi
    _abi.ctx()
    return torch


def _build(kind, n_utt, D, K, sync, seed=0, **kw):
    from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(n_utt, D, K, seed=0, N=20, n_slices_max=6)
    random.seed(seed)
    np.random.seed(seed)
    args = dict(n_slices_min=0, n_slices_max=6, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
                init_am_assignments="rand", time_power_term=1.0, sync=sync)
    args.update(kw)
    fixed = FixedVarPrior(0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D))
    if kind == "bigram":
        return baw.BigramAcousticWordseg(K, fixed, {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5}, *corpus,
                                         covariance_type="fixed", fb_type="unigram", **args)
    prior = fixed if kind == "fixed" else NIW(np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D))
    return uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, prior, *corpus, covariance_type=kind, fb_type="standard", **args)


def _check_state(seg, kind):
    """Invariants of the reference's view of the state."""
    u, c = seg.utterances, seg.acoustic_model.components
    a = c.assignments
    K = c.K
    # segmentation: last landmark closed, every segment is a scored span of at most n_slices_max slices
    tokens = []
    for i in range(u.D):
        N = u.lengths[i]
        assert u.boundaries[i, N - 1]
        embeds = u.get_segmented_embeds_i(i)
        assert all(e >= 0 for e in embeds)
        tokens.append(list(embeds))
    flat = np.concatenate(tokens)
    # exactly the segmented embeddings are assigned, labels are contiguous 0..K-1
    assert np.array_equal(np.sort(np.where(a >= 0)[0]), np.sort(flat))
    assert set(a[flat]) == set(range(K))
    counts = c.counts
    assert np.array_equal(counts[:K], np.bincount(a[flat], minlength=K)[:K]) and not counts[K:].any()
    # statistics equal to a recount from scratch
    X = c.X.astype(np.float64)
    sx = np.zeros((K, c.D))
    np.add.at(sx, a[flat], X[flat])
    if kind == "diag":
        npt.assert_allclose(c.m_N_numerators[:K], c.prior.k_0 * c.prior.m_0 + sx, rtol=1e-9, atol=1e-9)
    else:
        npt.assert_allclose(c.mu_N_numerators[:K], c.precision_0 * c.mu_0 + c.precision * sx, rtol=1e-9, atol=1e-9)
        npt.assert_allclose(c.precision_Ns[:K], c.precision_0 + counts[:K, None] * c.precision, rtol=1e-12)
    if kind == "bigram":
        uni = np.zeros(seg.lm.K, np.int64)
        big = np.zeros((seg.lm.K, seg.lm.K), np.int64)
        for t in tokens:
            ks = a[np.asarray(t, dtype=int)]
            np.add.at(uni, ks, 1)
            np.add.at(big, (ks[:-1], ks[1:]), 1)
        assert np.array_equal(seg.lm.unigram_counts, uni)
        assert np.array_equal(seg.lm.bigram_counts, big)
    return len(flat)


@pytest.mark.parametrize("kind", ["diag", "fixed", "bigram"])
def test_config2_serial_chain_properties(gpu, kind):
    """BASELINE config 2 (1 000 utterances, D = 39, K = 100), the reference's serial chain."""
    seg = _build(kind, 1000, 39, 100, "sequential")
    n0 = _check_state(seg, kind)
    rec = seg.gibbs_sample(2)
    n2 = _check_state(seg, kind)
    assert rec["n_tokens"][-1] == n2 and rec["components"][-1] == seg.acoustic_model.components.K
    assert np.all(np.isfinite(rec["log_marg"])) and rec["log_marg"][1] > rec["log_marg"][0] - abs(rec["log_marg"][0])
    assert n0 > 0


@pytest.mark.parametrize("kind,n_utt,D,K", [("diag", 1000, 39, 100), ("bigram", 10000, 100, 1000)])
def test_batch_sampler_properties_and_determinism(gpu, kind, n_utt, D, K):
    """Batch sampler at config-2 size and at config-5 shape: invariants after every sweep, the
    log-probability of the segmentation improves from the random start, two runs coincide."""
    finals = []
    for run in range(2):
        seg = _build(kind, n_utt, D, K, "batch", n_gibbs_blocks=8, n_stat_blocks=8, batch_seed=1)
        lps = []
        for sw in range(3):
            seg.batch_sweep_async()
            gpu.cuda.synchronize()
            seg._df.check_status()
            lps.append(float(seg._df.out_logprob.sum().item()))
        seg.materialise()
        _check_state(seg, kind)
        sweeper = seg._get_sweeper()
        cnt, tot, occ = sweeper.totals()
        assert occ == seg.acoustic_model.components.K and tot == seg.acoustic_model.get_n_assigned()
        assert lps[-1] > lps[0]
        finals.append((seg.utterances.boundaries.copy(), seg.acoustic_model.components.assignments.copy(), lps))
    assert np.array_equal(finals[0][0], finals[1][0]) and np.array_equal(finals[0][1], finals[1][1])
    assert finals[0][2] == finals[1][2]


@pytest.mark.parametrize("prec,tol", [("f16", 1e-4), ("f64", 1e-9)])
def test_config5_full_size_span_scores_against_the_specification(gpu, prec, tol):
    """BASELINE configs[4] at its real size (BigramAcousticWordseg, 10 000 utterances, D = 100, K = 1 000; score
    precision f16 is what `bench.py --workload bigram_c5` runs, f64 the default of the API).  After two sweeps the
    specification object (oracle/np_fbgmm_batch.py) is built from the materialised state; the span scores the
    next sweep's first Gibbs step computes for block 0 must equal the specification's log_marg_i
    (bigram_acoustic_wordseg.py:314-329) under "everything but block 0" within `tol` RELATIVE to
    max(|log_marg_i|, 1) -- the 1e-4 contract of the path for the matrix-core score, 1e-9 for the fp64 kernel."""
    from oracle import np_fbgmm_batch as nb
    seg = _build("bigram", 10000, 100, 1000, "batch", n_gibbs_blocks=8, n_stat_blocks=8, batch_seed=1, score_precision=prec)
    for _ in range(2):
        seg.batch_sweep_async()
    gpu.cuda.synchronize()
    seg._df.check_status()
    seg.materialise()
    spec = nb.FbgmmBatch(seg, n_gibbs_blocks=8, n_stat_blocks=8, seed=1)
    d = spec.derive(*spec.stats_excluding(0))
    uni, big = spec.uni.copy(), spec.big.copy()
    for s_ in range(spec.S):
        for i in range(*spec.ranges[s_][0]):
            spec._lm_count(uni, big, spec.tr[i], -1)
    seg.batch_sweep_async()
    gpu.cuda.synchronize()
    seg._df.check_status()
    score = seg._df.score.cpu().numpy()
    sw = seg._get_sweeper()
    rs = np.random.RandomState(0)
    rows = np.concatenate([rs.randint(sw.row_range_np[s_, 0, 0], sw.row_range_np[s_, 0, 1], size=150)
                           for s_ in range(sw.S)])
    worst = 0.0
    for row in rows:
        want = spec.log_marg(d, spec.X[row], uni, big)
        worst = max(worst, abs(score[row] - want) / max(abs(want), 1.0))
    print("configs[4] full size, %s span scores: worst error relative to max(|log_marg_i|, 1) = %.3g" % (prec, worst))
    assert worst < tol, worst


def test_config5_split_remainder_rows_give_the_bits_of_whole_row_blocks(gpu, monkeypatch):
    """The log-sum-exp span score (k_kmeans_score_sp, MODE 1) sends the row blocks behind the last whole round of workgroups
    -- two of 1 026 per Gibbs step at configs[4] -- through workgroups that take one chunk of tiles each; the association of
    a row's sum is the same either way, so a sweep must leave the same scores, boundaries and slots bit for bit as with
    SEGK_LSE_SPLIT=0 (every row block walked by one workgroup)."""
    out = []
    for mode in ("1", "0"):
        monkeypatch.setenv("SEGK_LSE_SPLIT", mode)
        seg = _build("bigram", 10000, 100, 1000, "batch", n_gibbs_blocks=8, n_stat_blocks=8, batch_seed=1, score_precision="f16")
        for _ in range(2):
            seg.batch_sweep_async()
        gpu.cuda.synchronize()
        seg._df.check_status()
        sw = seg._get_sweeper()
        out.append((seg._df.score.cpu().numpy().copy(), seg._dev_bounds.cpu().numpy().copy(), sw.slot.cpu().numpy().copy()))
    assert np.all(np.isfinite(out[0][0]))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
