"""
Direct tolerance tests of the modes `bench.py --workload fbgmm_diag_c2 / bigram_c5` time (score_precision "f32" / "f16").
north_star: "log-likelihoods within 1e-4 rel for FBGMM".  The span scores of these modes are held to that bound in
tests/test_gpu_fbgmm_batch.py; here the OTHER places where reduced-precision arithmetic enters the sampler are measured
value by value against the fp64 specification (oracle/np_fbgmm_batch.py, built from the reference's densities:
gaussian_components_diag.py:237-259, gaussian_components_fixedvar.py:242-253, unigram_acoustic_wordseg.py:684-703), at
configs[1] (D = 39, K = 100) and configs[4] (D = 100, K = 1000) shapes, through the library's probes
(segk_fbb_set_probe):

  * the token log-likelihoods the assignment draws are made from -- float32 Student-t terms with v_log_f32
    (segk_fbb_assign_diag32), the fp16x2 matrix-core contraction (segk_fbb_token_scores -> segk_fbb_assign, with and
    without a language model: the block-wide form and the one-wave-per-utterance form);
  * the forward filter's alphas of the boundary sampler with hardware exp / log (segk_fbatch.fast_dp = 1).

Bound: |got - want| <= 1e-4 * max(|want|, 1).
"""
import ctypes as C
import random

import numpy as np
import pytest

from oracle import np_fbgmm_batch as nb
from oracle import np_oracle as no

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    torch.cuda.set_device(0)
    from segmentalist_amd import _abi
    _abi.ctx()
    return torch


def _pair(kind, n_utt, D, K, prec, B=3, S=2, seed=5):
    from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(n_utt, D, K, seed=17, N=20, n_slices_max=6)          # the bench generator: 105 spans per utterance
    args = dict(n_slices_min=0, n_slices_max=6, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
                init_am_assignments="rand", time_power_term=1.0)
    bargs = dict(sync="batch", n_gibbs_blocks=B, n_stat_blocks=S, batch_seed=11, score_precision=prec)
    fixed = (0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D))
    niw = (np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D))
    lm = {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5}
    out = []
    for side in ("oracle", "product"):
        random.seed(seed)
        np.random.seed(seed)
        if kind == "bigram":
            seg = (no.BigramAcousticWordseg(K, no.FixedVarPrior(*fixed), dict(lm), *corpus, covariance_type="fixed",
                                            fb_type="unigram", **args) if side == "oracle" else
                   baw.BigramAcousticWordseg(K, FixedVarPrior(*fixed), dict(lm), *corpus, covariance_type="fixed",
                                             fb_type="unigram", **args, **bargs))
        elif side == "oracle":
            prior = no.FixedVarPrior(*fixed) if kind == "fixed" else no.NIW(*niw)
            seg = no.UnigramAcousticWordseg(no.FBGMM, 1.0, K, prior, *corpus, covariance_type=kind, fb_type="standard", **args)
        else:
            prior = FixedVarPrior(*fixed) if kind == "fixed" else NIW(*niw)
            seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, prior, *corpus, covariance_type=kind, fb_type="standard",
                                             **args, **bargs)
        out.append(seg)
    ref, seg = out
    return ref, nb.FbgmmBatch(ref, n_gibbs_blocks=B, n_stat_blocks=S, seed=11), seg


def _one_step(sw, seg, b, sweep=0, fused=False):
    """Gibbs step b of the sampler as FbgmmBatchSweeper.sweep enqueues it, without the partial-sum refresh (so that every
    step of this test conditions on the INITIAL state of the other blocks, which is what the specification object holds)."""
    import torch
    from segmentalist_amd._abi import check, ptr
    df = seg._df
    L, ctx, cp, fp, bp, st = sw._args()
    if sw.lm_tok is not None:
        check(L.segk_fbb_lm_apply(ctx, cp, fp, bp, b, -1, st))
    check(L.segk_fbb_prepare(ctx, cp, fp, bp, b, st))
    if fused:                  # the three calls below as one launch (diagonal components, float32 terms)
        check(L.segk_fbb_step_diag32(ctx, cp, fp, bp, sw.s_lo, sw.s_n, b, sw._n_utts[b], sweep, 0, 6, 0.0, 1.0, 1.0, 1.0,
                                     ptr(df.score), ptr(seg._dev_bounds), ptr(df.new_tok), ptr(df.n_new), ptr(df.out_logprob),
                                     ptr(df.status), st))
        torch.cuda.synchronize()
        df.check_status()
        return
    if sw.score_f32:
        check(L.segk_fbb_score_f32(ctx, cp, fp, bp, ptr(sw._block_rows[b]), sw._block_rows[b].numel(), ptr(df.score), st))
    elif sw.score_diag32:
        check(L.segk_fbb_score_diag32(ctx, cp, fp, bp, sw.s_lo, sw.s_n, b, sw._n_rows[b], ptr(df.score), st))
    else:
        check(L.segk_fbb_score(ctx, cp, fp, bp, sw.s_lo, sw.s_n, b, sw._n_rows[b], ptr(df.score), st))
    check(L.segk_fbb_segment(ctx, cp, fp, bp, sw.s_lo, sw.s_n, b, sw._n_utts[b], sweep, 0, 6, 0.0, 1.0, 1.0, ptr(df.score),
                             ptr(seg._dev_bounds), ptr(df.new_tok), ptr(df.n_new), ptr(df.out_logprob), ptr(df.status), st))
    if sw.ll_mat is not None:
        tm = sw._tok_map[b]
        rows = sw._tok_rows[:tm.numel()]
        torch.index_select(df.new_tok.view(-1), 0, tm, out=rows)
        check(L.segk_fbb_token_scores(ctx, cp, fp, bp, ptr(rows), tm.numel(), ptr(sw.ll_mat), sw.ll_ld, st))
        check(L.segk_fbb_assign(ctx, cp, fp, bp, sw.s_lo, sw.s_n, b, sw._n_utts[b], sweep, 1.0, ptr(df.new_tok), ptr(df.n_new),
                                ptr(sw.ll_mat), sw.ll_ld, st))
    elif sw.score_diag32:
        check(L.segk_fbb_assign_diag32(ctx, cp, fp, bp, sw.s_lo, sw.s_n, b, sw._n_utts[b], sweep, 1.0, ptr(df.new_tok),
                                       ptr(df.n_new), st))
    else:
        check(L.segk_fbb_assign(ctx, cp, fp, bp, sw.s_lo, sw.s_n, b, sw._n_utts[b], sweep, 1.0, ptr(df.new_tok), ptr(df.n_new),
                                None, 0, st))
    if sw.lm_tok is not None:          # put the block's old transcripts back: the next step starts from the initial tables
        check(L.segk_fbb_lm_apply(ctx, cp, fp, bp, b, 1, st))
    torch.cuda.synchronize()
    df.check_status()


CASES = [
    ("diag", "f32", 48, 39, 100, None),            # configs[1] shape: float32 Student-t token likelihoods, fast_dp
    ("diag", "f32", 48, 39, 100, "fused"),         # ... the Gibbs step as one launch (segk_fbb_step_diag32)
    ("diag", "f32", 40, 20, 160, "fused"),         # ... with three chunks of 64 slots
    ("fixed", "f16", 48, 39, 100, None),           # fp16x2 token likelihoods, block-wide draw kernel, fast_dp
    ("fixed", "f32", 48, 39, 100, None),           # fp32 MFMA span scores, fp64 token likelihoods, fast_dp
    ("bigram", "f16", 48, 100, 1000, None),        # configs[4] shape: one wave per utterance, hardware log / exp
    ("bigram", "f16", 48, 100, 1000, "0"),         # ... and the block-wide form it replaced (SEGK_FBB_ASSIGN_WAVE=0)
    ("fixed", "f16", 36, 100, 1000, None),
    ("bigram", "f16", 40, 60, 1100, None),         # more than 1 024 slots: the wave kernel's per-slot loops behind the column table
]


@pytest.mark.parametrize("kind,prec,n_utt,D,K,wave", CASES,
                         ids=["%s_%s_D%d_K%d%s" % (c[0], c[1], c[3], c[4], "" if c[5] is None else "_fused" if c[5] == "fused" else "_blockwide")
                              for c in CASES])
def test_token_likelihoods_and_forward_filter_within_the_contract(gpu, monkeypatch, kind, prec, n_utt, D, K, wave):
    torch = gpu
    from segmentalist_amd import _abi
    from segmentalist_amd._abi import check, ptr
    fused = wave == "fused"
    if wave is not None and not fused:
        monkeypatch.setenv("SEGK_FBB_ASSIGN_WAVE", wave)
    ref, spec, seg = _pair(kind, n_utt, D, K, prec)
    sw = seg._get_sweeper()
    assert sw.bt.fast_dp == 1
    sw.enter(seg._dev_bounds)
    u = ref.utterances
    N_max = seg._corpus.N_max
    alpha = torch.full((n_utt, N_max), float("nan"), dtype=torch.float64, device="cuda")
    ll = torch.full((n_utt * N_max, K), float("nan"), dtype=torch.float64, device="cuda")
    L, ctx = _abi.lib(), _abi.ctx()
    check(L.segk_fbb_set_probe(ctx, ptr(alpha), ptr(ll), K))
    try:
        for b in range(sw.B):
            _one_step(sw, seg, b, fused=fused)
    finally:
        check(L.segk_fbb_set_probe(ctx, None, None, 0))
    alpha, ll = alpha.cpu().numpy(), ll.cpu().numpy().reshape(n_utt, N_max, K)
    score = seg._df.score.cpu().numpy()
    new_tok, n_new = seg._df.new_tok.cpu().numpy(), seg._df.n_new.cpu().numpy()
    worst_ll = worst_a_own = worst_a_spec = worst_draw = 0.0
    n_tok = n_occ = n_drawn_empty = 0
    slots = sw.slot.cpu().numpy()
    for b in range(sw.B):
        d = spec.derive(*spec.stats_excluding(b))
        uni = big = None
        if kind == "bigram":           # LM counts of all other blocks
            uni, big = spec.uni.copy(), spec.big.copy()
            for s in range(spec.S):
                for i in range(*spec.ranges[s][b]):
                    spec._lm_count(uni, big, spec.tr[i], -1)
        for s in range(spec.S):
            for i in range(*spec.ranges[s][b]):
                N = u.lengths[i]
                tri = N * (N + 1) // 2
                # --- forward filter: (i) against the fp64 recurrence on the device's own span scores, (ii) against the
                # specification end to end (fp64 scores, library exp / log)
                vec_own = -np.inf * np.ones(tri)
                vec_spec = -np.inf * np.ones(tri)
                for j in range(tri):
                    e = u.vec_ids[i, j]
                    if e == -1 or np.isnan(u.durations[i, j]):
                        continue
                    vec_own[j] = score[e] * u.durations[i, j]
                    vec_spec[j] = spec.log_marg(d, spec.X[e], uni, big) * u.durations[i, j]
                a_own = no.forward_alphas(vec_own, 0.0, N, 6)
                a_spec = no.forward_alphas(vec_spec, 0.0, N, 6)
                got = alpha[i, :N]
                assert np.all(np.isfinite(got)), (i, got)
                worst_a_own = max(worst_a_own, float(np.max(np.abs(got - a_own) / np.maximum(np.abs(a_own), 1.0))))
                worst_a_spec = max(worst_a_spec, float(np.max(np.abs(got - a_spec) / np.maximum(np.abs(a_spec), 1.0))))
                # --- token log-likelihoods, every slot
                assert n_new[i] > 0
                for t in range(n_new[i]):
                    want = spec.loglik(d, spec.X[new_tok[i, t]])
                    g = ll[i, t]
                    assert np.all(np.isfinite(g)), (i, t)
                    worst_ll = max(worst_ll, float(np.max(np.abs(g - want) / np.maximum(np.abs(want), 1.0))))
                    n_tok += 1
                    n_occ += int(d["active"].sum())
                # --- the draws: slot k was drawn with the token's uniform u iff cum(k - 1) <= u < cum(k), cum the running sum
                # of the probabilities in slot order (utils.draw).  Probabilities out of the device's own fp64 token
                # log-likelihoods and the specification's prior (the device: v_log_f32 / v_exp_f32, ~3e-6 relative; the
                # one-wave kernel sums the occupied slots and the block of equal empty ones separately)
                if kind == "bigram" or fused:
                    j_prev = None
                    for t in range(n_new[i]):
                        z = spec.prior_z(d, j_prev, uni, big) + ll[i, t]
                        pr = np.exp(z - z.max())
                        cum = np.cumsum(pr / pr.sum())
                        k = int(slots[new_tok[i, t]])
                        assert 0 <= k < K
                        uu = nb.u01(spec.seed, 0, i, N_max + t)
                        below = cum[k - 1] if k > 0 else 0.0
                        worst_draw = max(worst_draw, below - uu, uu - cum[k])
                        n_drawn_empty += int(not d["active"][k])
                        j_prev = k if kind == "bigram" else None
    print("%s %s D=%d K=%d: token log-likelihoods worst %.3g (over %d tokens x %d slots, %.0f occupied per token); "
          "alphas worst %.3g against fp64 on the device's scores, %.3g against the specification"
          % (kind, prec, D, K, worst_ll, n_tok, K, n_occ / max(n_tok, 1), worst_a_own, worst_a_spec))
    if kind == "bigram" or fused:
        print("draws: worst distance of a token's uniform from its slot's interval %.3g; %d of %d tokens drew an empty slot"
              % (worst_draw, n_drawn_empty, n_tok))
        assert worst_draw < 1e-4, worst_draw
    assert n_tok >= 4 * n_utt
    assert worst_ll < TOL, worst_ll
    assert worst_a_own < TOL, worst_a_own
    assert worst_a_spec < TOL, worst_a_spec


def test_fused_gibbs_step_gives_the_span_scores_and_boundaries_of_the_three_launches(gpu):
    """segk_fbb_step_diag32 against segk_fbb_score_diag32 + segk_fbb_segment + segk_fbb_assign_diag32 from identical states,
    every block of a sweep: the span scores and the boundaries bit for bit (the same arithmetic and the same uniforms); the
    slots agree wherever the token's uniform does not fall within the last bits of a cumulative boundary (the token
    likelihoods of the two forms differ in their last float32 bits)."""
    out = []
    for fused in (False, True):
        ref, spec, seg = _pair("diag", 64, 39, 100, "f32")
        sw = seg._get_sweeper()
        sw.enter(seg._dev_bounds)
        for b in range(sw.B):
            _one_step(sw, seg, b, fused=fused)
        out.append((seg._df.score.cpu().numpy().copy(), seg._dev_bounds.cpu().numpy().copy(), sw.slot.cpu().numpy().copy(),
                    seg._df.out_logprob.cpu().numpy().copy(), seg._df.n_new.cpu().numpy().copy()))
    a, b_ = out
    assert np.array_equal(a[0], b_[0])
    assert np.array_equal(a[1], b_[1])
    assert np.array_equal(a[3], b_[3]) and np.array_equal(a[4], b_[4])
    both = (a[2] >= 0) | (b_[2] >= 0)
    assert np.mean(a[2][both] == b_[2][both]) > 0.98, np.mean(a[2][both] == b_[2][both])
