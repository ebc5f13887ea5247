"""Models built one after another in ONE process share the library context (workspaces, queues, cached tables) and, through
the caching allocator, usually the device ADDRESSES of the model before them.  Model B must come out the same whether it is
the first model of the process or follows a model A on another corpus of the same shape (round 4: a table cached under the
addresses of its inputs served the wrong corpus in exactly this sequence)."""
import gc
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


def _scaled(corpus, scale):
    corpus = list(corpus)
    corpus[0] = {k: (v * scale).astype(np.float32) for k, v in corpus[0].items()}
    return corpus


def _kmeans(sync):
    def run(torch, cseed, scale):
        from segmentalist_amd import kmeans_acoustic_wordseg as kaw
        from segmentalist_amd.synth import make_corpus
        n_utt = 3000 if sync == "batch" else 300
        corpus = _scaled(make_corpus(n_utt, 100, 1000, seed=cseed, N=20, n_slices_max=6), scale)
        random.seed(1); np.random.seed(1)
        seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", **({"sync": "batch"} if sync == "batch" else {}))
        if sync == "batch":
            for _ in range(5):                           # (from the third sweep on through the hinted score path)
                seg.batch_sweep_async()
            torch.cuda.synchronize()
            seg._dk.check_status()
        else:
            seg.segment(3)
        c = seg.acoustic_model.components
        return seg.utterances.boundaries.copy(), c.assignments.copy(), c.means.copy()
    return run


def _fb(kind, sync, prec="f64"):
    def run(torch, cseed, scale):
        from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, unigram_acoustic_wordseg as uaw
        from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
        from segmentalist_amd.niw import NIW
        from segmentalist_amd.synth import make_corpus
        D, K = 24, 40
        corpus = _scaled(make_corpus(80, D, K, seed=cseed, N=10, n_slices_max=5), scale)
        random.seed(3); np.random.seed(3)
        args = dict(n_slices_min=0, n_slices_max=5, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
                    init_am_assignments="rand", time_power_term=1.0)
        if sync == "batch":
            args.update(sync="batch", n_gibbs_blocks=4, n_stat_blocks=4, batch_seed=11, score_precision=prec)
        if kind == "bigram":
            seg = baw.BigramAcousticWordseg(K, FixedVarPrior(*cases.fixed_prior_params(D)), dict(cases.BIGRAM_LM), *corpus,
                                            covariance_type="fixed", fb_type="unigram", **args)
        else:
            prior = FixedVarPrior(*cases.fixed_prior_params(D)) if kind == "fixed" else NIW(*cases.diag_prior_params(D))
            seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, prior, *corpus, covariance_type=kind, fb_type="standard", **args)
        if sync == "batch":
            for _ in range(3):
                seg.batch_sweep_async()
            torch.cuda.synchronize()
            seg._df.check_status()
            seg.materialise()
            extra = seg._df.out_logprob.cpu().numpy().copy()
        else:
            extra = np.array(seg.gibbs_sample(3)["log_marg"])
        return seg.utterances.boundaries.copy(), seg.acoustic_model.components.assignments.copy(), extra
    return run


DRIVERS = [("kmeans-batch", _kmeans("batch")), ("kmeans-sequential", _kmeans("sequential")),
           ("fbgmm-fixed-chain", _fb("fixed", "sequential")), ("fbgmm-diag-chain", _fb("diag", "sequential")),
           ("bigram-chain", _fb("bigram", "sequential")), ("fbgmm-diag-batch-f64", _fb("diag", "batch", "f64")),
           ("fbgmm-diag-batch-f32", _fb("diag", "batch", "f32")), ("fbgmm-fixed-batch-f16", _fb("fixed", "batch", "f16")),
           ("bigram-batch-f16", _fb("bigram", "batch", "f16"))]


@pytest.mark.parametrize("name,fn", DRIVERS, ids=[d[0] for d in DRIVERS])
def test_a_model_is_the_same_after_another_model_of_the_same_shape(gpu, name, fn):
    def once(cseed, scale):
        out = fn(gpu, cseed, scale)
        gc.collect()
        gpu.cuda.synchronize()
        return out
    first = once(5, 1.7)
    once(4, 1.0)
    again = once(5, 1.7)
    for x, y in zip(first, again):
        assert np.array_equal(x, y), name
