"""GPU parity of the batch-synchronous (blocked parallel Gibbs) FBGMM / bigram sampler against its
executable specification oracle/np_fbgmm_batch.py (the reference has no parallel mode).  Sampled
boundaries and slots must coincide with the specification's (same counter-based uniforms);
log-probabilities to 1e-9 relative (contract 1e-4)."""
import random

import numpy as np
import numpy.testing as npt
import pytest

from oracle import np_fbgmm_batch as nb
from oracle import np_oracle as no
from tests.golden import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    torch.cuda.set_device(0)
    from segmentalist_amd import _abi
    _abi.ctx()
    return torch


def _pair(kind, n_utt, D, K, cseed, nmax, B, S, seed=5, dtype="float32", score_precision="f64", n_landmarks=0, **kw):
    """(oracle segmenter + batch state, product segmenter) from identical initial states.  n_landmarks > 0: every utterance
    that long (default: ragged, 3 to 9 landmarks)."""
    from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    corpus = cases.chain_corpus(n_utt, D, K, cseed, n_landmarks == 0, n_landmarks, nmax, dtype)
    args = dict(n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
                init_am_assignments="rand", time_power_term=1.0)
    args.update(kw)
    bargs = dict(sync="batch", n_gibbs_blocks=B, n_stat_blocks=S, batch_seed=11, score_precision=score_precision)
    out = []
    for side in ("oracle", "product"):
        random.seed(seed)
        np.random.seed(seed)
        if kind == "bigram":
            if side == "oracle":
                seg = no.BigramAcousticWordseg(K, no.FixedVarPrior(*cases.fixed_prior_params(D)), dict(cases.BIGRAM_LM),
                                               *corpus, covariance_type="fixed", fb_type="unigram", **args)
            else:
                seg = baw.BigramAcousticWordseg(K, FixedVarPrior(*cases.fixed_prior_params(D)), dict(cases.BIGRAM_LM),
                                                *corpus, covariance_type="fixed", fb_type="unigram", **args, **bargs)
        else:
            if side == "oracle":
                prior = (no.FixedVarPrior(*cases.fixed_prior_params(D)) if kind == "fixed"
                         else no.NIW(*cases.diag_prior_params(D)))
                seg = no.UnigramAcousticWordseg(no.FBGMM, 1.0, K, prior, *corpus, covariance_type=kind,
                                                fb_type="standard", **args)
            else:
                prior = FixedVarPrior(*cases.fixed_prior_params(D)) if kind == "fixed" else NIW(*cases.diag_prior_params(D))
                seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, prior, *corpus, covariance_type=kind,
                                                 fb_type="standard", **args, **bargs)
        out.append(seg)
    ref, seg = out
    return ref, nb.FbgmmBatch(ref, n_gibbs_blocks=B, n_stat_blocks=S, seed=11), seg


CASES = [
    ("fixed", 24, 8, 10, 77, 5, 3, 4, {}),
    ("diag", 24, 8, 10, 78, 5, 3, 4, {}),
    ("fixed", 40, 12, 30, 79, 6, 4, 8, dict(lms=0.7, wip=-0.2, time_power_term=1.2)),
    ("diag", 33, 70, 12, 80, 4, 2, 2, {}),          # D > 64: two dimension chunks
    ("fixed", 19, 6, 300, 81, 5, 5, 1, {}),         # K_max > workgroup width
    ("diag", 130, 8, 10, 82, 5, 9, 12, {}),         # more than eight blocks and more than eight slices: k_fbb_prepare's second load rounds
    ("diag", 20, 256, 12, 83, 4, 2, 2, {}),         # the widest rows the batch sampler takes (D <= 256)
    ("fixed", 20, 200, 12, 84, 4, 2, 2, {}),
    ("bigram", 30, 100, 40, 85, 5, 3, 4, {}),
]


@pytest.mark.parametrize("kind,n_utt,D,K,cseed,nmax,B,S,kw", CASES,
                         ids=["%s_u%d_D%d_K%d_B%d_S%d" % (c[0], c[1], c[2], c[3], c[6], c[7]) for c in CASES])
def test_batch_sweeps_match_specification(gpu, kind, n_utt, D, K, cseed, nmax, B, S, kw):
    ref, spec, seg = _pair(kind, n_utt, D, K, cseed, nmax, B, S, **kw)
    for sw in range(3):
        lp = spec.sweep(sw)
        seg.batch_sweep_async()
        gpu.cuda.synchronize()
        seg._df.check_status()
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), sw
        slots = seg._get_sweeper().slot.cpu().numpy()
        assert np.array_equal(slots, spec.slot), sw
        npt.assert_allclose(seg._df.out_logprob.cpu().numpy(), lp, rtol=1e-9)
        # the reference's view
        seg.materialise()
        a, Kc = spec.canonical()
        c = seg.acoustic_model.components
        assert c.K == Kc
        assert np.array_equal(c.assignments, a)
        cnt = spec.stats_excluding(-1)[0]
        assert np.array_equal(c.counts[:Kc], cnt[cnt > 0])


@pytest.mark.parametrize("kind,nmax,n_landmarks", [("fixed", 20, 24), ("diag", 20, 24), ("fixed", 12, 24), ("diag", 6, 64), ("fixed", 30, 64)])
def test_batch_sweeps_with_a_wide_window(gpu, kind, nmax, n_landmarks):
    """Utterances of 24 landmarks and windows of 20 and 12 slices: the boundary sampler's register path beyond one row of
    sixteen lanes (fb_dp_sample: delay line by wave_shr, maxima across rows) and inside it with more than eight candidates."""
    ref, spec, seg = _pair(kind, 8, 8, 10, 93, nmax, 2, 2, n_landmarks=n_landmarks)
    assert int(np.max(seg.utterances.lengths)) == n_landmarks          # (64: the most the batch sampler takes)
    for sw in range(2):
        lp = spec.sweep(sw)
        seg.batch_sweep_async()
        gpu.cuda.synchronize()
        seg._df.check_status()
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), sw
        assert np.array_equal(seg._get_sweeper().slot.cpu().numpy(), spec.slot), sw
        npt.assert_allclose(seg._df.out_logprob.cpu().numpy(), lp, rtol=1e-9)


@pytest.mark.parametrize("kind", ["fixed", "diag"])
def test_batch_sweeps_without_the_prior_row_table(gpu, kind):
    """segk_fbatch.prior_rows = NULL: the score and assignment kernels evaluate the rows' prior predictive themselves (the
    table of segk_fbb_prior_rows is the default and is what the tests above run) -- the same chain, and the table holds the
    values the kernels would compute."""
    ref, spec, seg = _pair(kind, 24, 8, 30, 91, 5, 3, 4)          # K_max > the number of tokens' components: empty slots
    sweeper = seg._get_sweeper()
    assert sweeper.bt.prior_rows, "the sweeper keeps the table by default"
    table = sweeper.prior_rows.cpu().numpy()
    want = np.array([ref.acoustic_model.components.log_prior(i) for i in range(len(table))])
    npt.assert_allclose(table, want, rtol=1e-12)
    sweeper.bt.prior_rows = None
    for sw in range(2):
        spec.sweep(sw)
        seg.batch_sweep_async()
        gpu.cuda.synchronize()
        seg._df.check_status()
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), sw
        assert np.array_equal(sweeper.slot.cpu().numpy(), spec.slot), sw


def test_batch_statistics_and_scores_match_specification(gpu):
    """The prepared statistics and the span scores of one step, value by value."""
    ref, spec, seg = _pair("diag", 24, 8, 10, 78, 5, 3, 4)
    sw = seg._get_sweeper()
    sw.enter(seg._dev_bounds)
    from segmentalist_amd import _abi
    from segmentalist_amd._abi import check, ptr
    L, ctx, cp, fp, bp, st = sw._args()
    for b in (0, 2, -1):
        check(L.segk_fbb_prepare(ctx, cp, fp, bp, b, st))
        cnt, sx, sxx = spec.stats_excluding(b)
        d = spec.derive(cnt, sx, sxx)
        assert np.array_equal(sw.cnt.cpu().numpy(), cnt.astype(np.float64))
        occ = cnt > 0
        npt.assert_allclose(sw.mean_t.cpu().numpy().T[occ], d["mean"][occ], rtol=1e-13)
        npt.assert_allclose(sw.q_t.cpu().numpy().T[occ], d["q"][occ], rtol=1e-12)
        npt.assert_allclose(sw.lconst.cpu().numpy()[occ], d["const"][occ], rtol=1e-12)
    b = 1
    check(L.segk_fbb_prepare(ctx, cp, fp, bp, b, st))
    check(L.segk_fbb_score(ctx, cp, fp, bp, sw.s_lo, sw.s_n, b, sw._n_rows[b], ptr(seg._df.score), st))
    d = spec.derive(*spec.stats_excluding(b))
    score = seg._df.score.cpu().numpy()
    for s in range(sw.S):
        lo, hi = sw.row_range_np[s, b]
        for row in range(lo, hi):
            npt.assert_allclose(score[row], spec.log_marg(d, spec.X[row]), rtol=1e-11)


def test_bigram_batch_sweeps_match_specification(gpu):
    ref, spec, seg = _pair("bigram", 30, 8, 12, 91, 5, 3, 4)
    for sw in range(3):
        lp = spec.sweep(sw)
        seg.batch_sweep_async()
        gpu.cuda.synchronize()
        seg._df.check_status()
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), sw
        assert np.array_equal(seg._get_sweeper().slot.cpu().numpy(), spec.slot), sw
        assert np.array_equal(seg._get_sweeper().lm_big.cpu().numpy(), spec.big), sw
        npt.assert_allclose(seg._df.out_logprob.cpu().numpy(), lp, rtol=1e-9)
    seg.materialise()
    a, Kc = spec.canonical()
    assert np.array_equal(seg.acoustic_model.components.assignments, a)
    # LM tables of the LM object in the reference's labelling
    cnt = spec.stats_excluding(-1)[0]
    occ = np.where(cnt > 0)[0]
    assert np.array_equal(seg.lm.unigram_counts[:Kc], cnt[occ])
    assert np.array_equal(seg.lm.bigram_counts[:Kc, :Kc], spec.big[np.ix_(occ, occ)])


def test_sequential_after_batch_continues_from_the_materialised_state(gpu):
    """Mode switch: batch sweeps, then the reference's serial chain on the same object, compared
    with the oracle continuing from the specification's canonical state."""
    ref, spec, seg = _pair("fixed", 16, 6, 8, 93, 4, 2, 2)
    spec.sweep(0)
    seg.batch_sweep_async()
    seg.materialise()
    a, Kc = spec.canonical()
    # oracle: rebuild its components from the canonical assignment
    prior = no.FixedVarPrior(*cases.fixed_prior_params(6))
    ref.acoustic_model.components = no.GaussianComponentsFixedVar(ref.acoustic_model.components.X, prior, a.copy(), K_max=8)
    random.seed(3)
    st = random.getstate()
    ref.gibbs_sample(1)
    random.setstate(st)
    seg.sync = "sequential"
    rec = seg.gibbs_sample(1)
    assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries)
    assert np.array_equal(seg.acoustic_model.components.assignments, ref.acoustic_model.components.assignments)
    assert rec["components"][0] == ref.acoustic_model.components.K


@pytest.mark.parametrize("prec", ["f32", "f16"])
@pytest.mark.parametrize("kind,n_utt,D,K,nmax", [("fixed", 40, 12, 30, 6), ("bigram", 30, 100, 64, 5), ("fixed", 25, 39, 100, 6)])
def test_matrix_core_span_scores_within_tolerance(gpu, kind, n_utt, D, K, nmax, prec):
    """score_precision="f32": the MFMA log-sum-exp kernel against the specification's log_marg_i on the
    same state -- 1e-4 relative is the contract of the path (BASELINE north_star); measured ~1e-6."""
    from segmentalist_amd._abi import check, ptr
    ref, spec, seg = _pair(kind, n_utt, D, K, 123, nmax, 3, 2, score_precision=prec)
    sw = seg._get_sweeper()
    assert sw.score_f32 and sw.score_f16 == (prec == "f16")
    sw.enter(seg._dev_bounds)
    L, ctx, cp, fp, bp, st = sw._args()
    worst = worst_abs = 0.0
    mags = []
    for b in range(sw.B):
        check(L.segk_fbb_prepare(ctx, cp, fp, bp, b, st))
        check(L.segk_fbb_score_f32(ctx, cp, fp, bp, ptr(sw._block_rows[b]), sw._block_rows[b].numel(), ptr(seg._df.score), st))
        d = spec.derive(*spec.stats_excluding(b))
        # with a language model the unigram counts of "all other blocks" are the slot counts
        uni, big = (d["cnt"], spec.big) if kind == "bigram" else (None, None)
        score = seg._df.score.cpu().numpy()
        for s in range(sw.S):
            lo, hi = sw.row_range_np[s, b]
            for row in range(lo, hi):
                want = spec.log_marg(d, spec.X[row], uni, big)
                mags.append(abs(want))
                worst_abs = max(worst_abs, abs(score[row] - want))
                worst = max(worst, abs(score[row] - want) / max(abs(want), 1.0))
    print("%s span score: worst abs err %.3g, worst err relative to max(|log_marg_i|, 1) %.3g, median |log_marg| %.3g"
          % (prec, worst_abs, worst, float(np.median(mags))))
    # 1e-4 relative is the contract of the path; a span whose log-marginal is within 1 of zero is held to the same
    # absolute error as a span of magnitude 1 (the same floor as the diagonal and the token-likelihood tests)
    assert worst < 1e-4, (worst, worst_abs)


@pytest.mark.parametrize("kind,prec", [("fixed", "f32"), ("fixed", "f16"), ("bigram", "f16"), ("diag", "f32")])
def test_matrix_core_mode_samples_a_valid_chain(gpu, kind, prec):
    """Full sweeps with matrix-core scores (f16: also the token likelihoods of the assignment step; diagonal components
    in f32: span scores AND token likelihoods from float32 Student-t terms, segk_fbb_assign_diag32): the
    chain stays close to the f64 chain (equal boundaries and slots for the vast majority of utterances
    after one sweep -- a draw only flips when a uniform falls within ~1e-5 of a cumulative boundary) and
    the state invariants hold."""
    ref, spec, seg = _pair(kind, 60, 16, 24, 321, 6, 3, 4, score_precision=prec)
    lp = spec.sweep(0)
    seg.batch_sweep_async()
    gpu.cuda.synchronize()
    seg._df.check_status()
    same = np.mean(np.all(seg.utterances.boundaries == ref.utterances.boundaries, axis=1))
    assert same > 0.9, same
    slots = seg._get_sweeper().slot.cpu().numpy()
    both = (slots >= 0) & (spec.slot >= 0)
    assert np.mean(slots[both] == spec.slot[both]) > 0.9
    npt.assert_allclose(float(seg._df.out_logprob.sum().item()), lp.sum(), rtol=5e-2)
    for _ in range(2):
        seg.batch_sweep_async()
    seg.materialise()
    c = seg.acoustic_model.components
    assert c.counts[:c.K].sum() == seg.acoustic_model.get_n_assigned()
    if kind == "bigram":
        assert seg.lm.unigram_counts.sum() == seg.acoustic_model.get_n_assigned()


@pytest.mark.parametrize("n_utt,D,K,nmax,scale", [(40, 12, 30, 6, 1.0), (25, 39, 100, 6, 1.0), (30, 8, 12, 5, 1.0)],
                         ids=["D12_K30_with_near_zero_values", "c2_shape_D39_K100", "D8_K12"])
def test_diag_float32_span_scores_within_the_contract(gpu, n_utt, D, K, nmax, scale):
    """score_precision="f32" with diagonal components (k_fbb_score_diag32: Student-t terms in float32 with v_log_f32)
    against the specification's log_marg_i (fbgmm.py:256-285 over gaussian_components_diag.py:237-259) on the same
    state.  The contract of the path is 1e-4 RELATIVE to |log_marg_i|; a span whose log-marginal is within 1 of zero
    is held to the same ABSOLUTE error as a span of magnitude 1.  On the D = 12 corpus one span in seven has
    |log_marg_i| < 1 (asserted below), so the near-zero regime is part of the measurement."""
    from segmentalist_amd import fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd._abi import check
    from segmentalist_amd.niw import NIW
    corpus = cases.chain_corpus(n_utt, D, K, 321, True, 0, nmax, "float32")
    if scale != 1.0:
        corpus = ({k: (v * scale).astype(np.float32) for k, v in corpus[0].items()},) + tuple(corpus[1:])
    args = dict(n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
                init_am_assignments="rand", time_power_term=1.0)
    random.seed(5); np.random.seed(5)
    ref = no.UnigramAcousticWordseg(no.FBGMM, 1.0, K, no.NIW(*cases.diag_prior_params(D)), *corpus, covariance_type="diag",
                                    fb_type="standard", **args)
    spec = nb.FbgmmBatch(ref, n_gibbs_blocks=3, n_stat_blocks=2, seed=11)
    random.seed(5); np.random.seed(5)
    seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, NIW(*cases.diag_prior_params(D)), *corpus, covariance_type="diag",
                                     fb_type="standard", sync="batch", n_gibbs_blocks=3, n_stat_blocks=2, batch_seed=11,
                                     score_precision="f32", **args)
    sw = seg._get_sweeper()
    assert sw.score_diag32 and not sw.score_f32
    sw.enter(seg._dev_bounds)
    L, ctx, cp, fp, bp, st = sw._args()
    worst, n_small, mags = 0.0, 0, []
    for b in range(sw.B):
        check(L.segk_fbb_prepare(ctx, cp, fp, bp, b, st))
        check(L.segk_fbb_score_diag32(ctx, cp, fp, bp, sw.s_lo, sw.s_n, b, sw._n_rows[b], _abi_ptr(seg._df.score), st))
        d = spec.derive(*spec.stats_excluding(b))
        score = seg._df.score.cpu().numpy()
        for s_ in range(sw.S):
            lo, hi = sw.row_range_np[s_, b]
            for row in range(lo, hi):
                want = spec.log_marg(d, spec.X[row])
                mags.append(abs(want))
                n_small += abs(want) < 1.0
                worst = max(worst, abs(score[row] - want) / max(abs(want), 1.0))
    print("diag f32 span score: worst error relative to max(|log_marg_i|, 1) = %.3g; %d of %d spans with |log_marg_i| < 1, "
          "median |log_marg_i| %.3g" % (worst, n_small, len(mags), float(np.median(mags))))
    assert worst < 1e-4, worst
    if D == 12:
        assert n_small >= len(mags) // 20, (n_small, len(mags))
    # and a whole sweep in this mode samples a valid chain
    seg.batch_sweep_async()
    gpu.cuda.synchronize()
    seg._df.check_status()
    seg.materialise()
    c = seg.acoustic_model.components
    assert c.counts[:c.K].sum() == seg.acoustic_model.get_n_assigned()


def _abi_ptr(t):
    from segmentalist_amd._abi import ptr
    return ptr(t)


@pytest.mark.parametrize("kind", ["fixed", "bigram"])
def test_sorted_partial_sums_are_the_same_bits(gpu, monkeypatch, kind):
    """Banks of 256 slots and more bucket a block's tokens by slot (k_fbb_sort: stable counting sort) before the partial
    sums; SEGK_FBB_SORT=0 keeps the one-step kernel that walks every utterance per slot.  Same additions in the same order:
    the sampler's whole state after three sweeps is identical."""
    states = []
    for mode in ("0", "1"):
        monkeypatch.setenv("SEGK_FBB_SORT", mode)
        _, _, seg = _pair(kind, 50, 12, 300, 99, 6, 3, 4)
        for _ in range(3):
            seg.batch_sweep_async()
        gpu.cuda.synchronize()
        seg._df.check_status()
        sw = seg._get_sweeper()
        states.append((sw.partials.cpu().numpy().copy(), sw.slot.cpu().numpy().copy(), seg.utterances.boundaries.copy()))
    for a, b in zip(*states):
        assert np.array_equal(a, b)
