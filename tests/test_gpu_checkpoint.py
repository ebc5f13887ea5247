"""state_dict() / load_state_dict(): a chain resumed from a checkpoint in a freshly constructed
segmenter continues bit for bit like the uninterrupted one -- serial chains and batch samplers."""
import pickle
import random

import numpy as np
import pytest

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


def _make(kind, sync):
    from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, kmeans_acoustic_wordseg as kaw
    from segmentalist_amd import unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    D, K = 8, 10
    corpus = cases.chain_corpus(20, D, K, 4711, True, 0, 5, "float32")
    random.seed(7)
    np.random.seed(7)
    if kind == "kmeans":
        return kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_max=5, init_am_assignments="rand", sync=sync, n_stat_blocks=4)
    kw = dict(n_slices_min=0, n_slices_max=5, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
              init_am_assignments="rand", time_power_term=1.0, sync=sync, n_gibbs_blocks=2, n_stat_blocks=4, batch_seed=3)
    fixed = FixedVarPrior(*cases.fixed_prior_params(D))
    if kind == "bigram":
        return baw.BigramAcousticWordseg(K, fixed, dict(cases.BIGRAM_LM), *corpus, covariance_type="fixed", fb_type="unigram", **kw)
    return uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, NIW(*cases.diag_prior_params(D)), *corpus, covariance_type="diag",
                                      fb_type="standard", **kw)


def _run(seg, kind, n):
    return seg.segment(n) if kind == "kmeans" else seg.gibbs_sample(n)


def _snapshot(seg, kind):
    c = seg.acoustic_model.components
    out = [seg.utterances.boundaries.copy(), c.assignments.copy(), c.counts.copy(), np.array(c.K)]
    if kind == "kmeans":
        out += [c.means.copy(), c.mean_numerators.copy()]
    else:
        out += [c.dev.stat_a.cpu().numpy(), c.dev.stat_b.cpu().numpy()]
    if kind == "bigram":
        out += [seg.lm.unigram_counts, seg.lm.bigram_counts]
    return out


@pytest.mark.parametrize("kind", ["kmeans", "diag", "bigram"])
@pytest.mark.parametrize("sync", ["sequential", "batch"])
def test_resume_is_bit_identical(gpu, kind, sync):
    a = _make(kind, sync)
    _run(a, kind, 2)
    sd = pickle.loads(pickle.dumps(a.state_dict()))     # plain data: survives a round trip through pickle
    rec_a = _run(a, kind, 2)
    want = _snapshot(a, kind)
    b = _make(kind, sync)                                 # same arguments, fresh random initial state
    random.seed(999)
    np.random.seed(999)
    b.load_state_dict(sd)
    rec_b = _run(b, kind, 2)
    got = _snapshot(b, kind)
    for x, y in zip(want, got):
        assert np.array_equal(x, y)
    key = "sum_neg_len_sqrd_norm" if kind == "kmeans" else "log_marg"
    assert list(rec_a[key]) == list(rec_b[key])
