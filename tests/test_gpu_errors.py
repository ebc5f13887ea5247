"""Error behaviour of the drop-in surface: the reference's assertions and the library's own refusals
must surface as Python exceptions, never as silent wrong results."""
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


def _kmeans(**kw):
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    corpus = cases.chain_corpus(8, 6, 5, 11, True, 0, 4, "float32")
    random.seed(1)
    np.random.seed(1)
    return kaw.SegmentalKMeansWordseg(5, *corpus, n_slices_max=4, init_am_assignments="rand", **kw)


def test_kmeans_components_assertions(gpu):
    """kmeans_components.py:100-101: add_item(-1, .) and adding an assigned item assert."""
    seg = _kmeans()
    c = seg.acoustic_model.components
    with pytest.raises(AssertionError):
        c.add_item(-1, 0)
    i = int(np.where(c.assignments >= 0)[0][0])
    with pytest.raises(AssertionError):
        c.add_item(i, 0)
        c.dev.check_status()


def test_unsupported_reference_paths_raise(gpu):
    """Settings the reference itself rejects (unigram...:520-521 `assert False, "to check"`; the
    k-means driver without seed assignments / one-by-one init, kmeans_...:148-149,207-208)."""
    from segmentalist_amd import fbgmm, kmeans_acoustic_wordseg as kaw, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    corpus = cases.chain_corpus(6, 6, 5, 12, True, 0, 4, "float32")
    prior = FixedVarPrior(*cases.fixed_prior_params(6))
    seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, 5, prior, *corpus, covariance_type="fixed", n_slices_max=4,
                                     beta_sent_boundary=2.0)
    with pytest.raises(AssertionError):
        seg.gibbs_sample_i(0)
    with pytest.raises(AssertionError):
        kaw.SegmentalKMeansWordseg(5, *corpus, n_slices_max=4, init_am_assignments="one-by-one")
    with pytest.raises(NotImplementedError):
        fbgmm.FBGMM(np.zeros((4, 3), np.float32), prior, 1.0, 2, covariance_type="full")
    with pytest.raises(AssertionError):
        uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, 5, prior, *corpus, covariance_type="fixed", fb_type="nope")


def test_library_refusals(gpu):
    """Arguments outside the supported domain come back as SegkError with a message."""
    import ctypes as C
    from segmentalist_amd import _abi
    from segmentalist_amd._abi import SegkError
    seg = _kmeans()
    dk = seg._dk
    L, ctx = _abi.lib(), _abi.ctx()
    # n_slices_min >= 2 is undefined behaviour in the reference: refused
    with pytest.raises(SegkError, match="n_slices_min"):
        dk.segment(seg._dev_bounds, 2, 4, 0.0)
    # a row range outside the corpus
    with pytest.raises(SegkError, match="row range"):
        _abi.check(L.segk_kmeans_filter(ctx, dk._cp(), C.byref(dk.m), None, 0, seg._corpus.n_emb + 1,
                                        C.byref(dk.cand), _abi.stream()))
    # batch mode needs the number of statistics blocks to divide over the ranks, and at least two Gibbs blocks
    from segmentalist_amd import fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    corpus = cases.chain_corpus(6, 6, 5, 12, True, 0, 4, "float32")
    prior = FixedVarPrior(*cases.fixed_prior_params(6))
    useg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, 5, prior, *corpus, covariance_type="fixed", n_slices_max=4,
                                      beta_sent_boundary=-1, sync="batch", n_gibbs_blocks=1, n_stat_blocks=2)
    with pytest.raises(SegkError, match="n_blocks"):
        useg.batch_sweep_async()
    from segmentalist_amd.niw import NIW
    dseg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, 5, NIW(*cases.diag_prior_params(6)), *corpus, covariance_type="diag",
                                      n_slices_max=4, beta_sent_boundary=-1, sync="batch", n_gibbs_blocks=2, n_stat_blocks=2,
                                      score_precision="f16")
    with pytest.raises(SegkError, match="fixed-variance"):
        dseg.batch_sweep_async()


def test_uniform_stream_exhaustion_is_reported(gpu):
    """The serial chain consumes a pre-drawn block of uniforms; running past it is an error, not a
    silent 0.5."""
    from segmentalist_amd import fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd._abi import SegkError
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    corpus = cases.chain_corpus(6, 6, 5, 12, True, 0, 4, "float32")
    seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, 5, FixedVarPrior(*cases.fixed_prior_params(6)), *corpus,
                                     covariance_type="fixed", n_slices_max=4, beta_sent_boundary=-1)
    seg._df.set_uniform_stream(np.array([0.3]))        # far too short for an utterance
    seg._gibbs_i_async(0, 1, False)
    with pytest.raises(SegkError, match="uniform stream"):
        seg._df.check_status()
