"""Edge cases of the drivers on the GPU against trajectories captured from the reference
(tests/golden/edge.npz): 1- and 2-landmark utterances, spans below min_duration (NaN durations),
n_slices_min = 1, a single initial span (p_boundary_init = 0), seed boundaries snapped to landmarks.
Also the same corpora through the batch modes against their specifications."""
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


def _product_mods(**extra):
    from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, kmeans_acoustic_wordseg as kaw
    from segmentalist_amd import unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW

    def wrap(cls):
        return (lambda *a, **k: cls(*a, **dict(k, **extra))) if extra else cls
    return dict(SegmentalKMeansWordseg=wrap(kaw.SegmentalKMeansWordseg), UnigramAcousticWordseg=wrap(uaw.UnigramAcousticWordseg),
                BigramAcousticWordseg=wrap(baw.BigramAcousticWordseg), FBGMM=fbgmm.FBGMM, FixedVarPrior=FixedVarPrior, NIW=NIW)


def _oracle_mods():
    return dict(SegmentalKMeansWordseg=no.SegmentalKMeansWordseg, UnigramAcousticWordseg=no.UnigramAcousticWordseg,
                BigramAcousticWordseg=no.BigramAcousticWordseg, FBGMM=no.FBGMM, FixedVarPrior=no.FixedVarPrior, NIW=no.NIW)


@pytest.mark.parametrize("case", cases.EDGE_CHAINS, ids=[c[0] for c in cases.EDGE_CHAINS])
def test_edge_case_chains_vs_reference(gpu, golden, case):
    g = golden("edge")
    name, driver = case[0], case[1]
    random.seed(1)
    np.random.seed(1)
    seg = cases.edge_build(_product_mods(), case)
    c = seg.acoustic_model.components
    assert np.array_equal(seg.utterances.boundaries, g[name + "_init_bounds"])
    assert np.array_equal(c.assignments, g[name + "_init_assign"])
    for it in range(3):
        rec = seg.segment(1) if driver == "kmeans" else seg.gibbs_sample(1)
        assert np.array_equal(seg.utterances.boundaries, g[name + "_bounds"][it]), it
        assert np.array_equal(c.assignments, g[name + "_assign"][it]), it
        key = "sum_neg_len_sqrd_norm" if driver == "kmeans" else "log_marg"
        npt.assert_allclose(rec[key][0], g[name + "_rec_" + key][it], rtol=1e-8)
        assert rec["components"][0] == g[name + "_rec_components"][it]
        assert rec["n_tokens"][0] == g[name + "_rec_n_tokens"][it]


@pytest.mark.parametrize("case", [c for c in cases.EDGE_CHAINS if c[1] != "kmeans"],
                         ids=[c[0] for c in cases.EDGE_CHAINS if c[1] != "kmeans"])
def test_edge_cases_batch_sampler_vs_specification(gpu, case):
    random.seed(2)
    np.random.seed(2)
    ref = cases.edge_build(_oracle_mods(), case)
    spec = nb.FbgmmBatch(ref, n_gibbs_blocks=2, n_stat_blocks=2, seed=3)
    random.seed(2)
    np.random.seed(2)
    seg = cases.edge_build(_product_mods(sync="batch", n_gibbs_blocks=2, n_stat_blocks=2, batch_seed=3), case)
    for sw in range(3):
        lp = spec.sweep(sw)
        seg.batch_sweep_async()
        gpu.cuda.synchronize()
        seg._df.check_status()
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), sw
        assert np.array_equal(seg._get_sweeper().slot.cpu().numpy(), spec.slot), sw
        npt.assert_allclose(seg._df.out_logprob.cpu().numpy(), lp, rtol=1e-9)


@pytest.mark.parametrize("case", [c for c in cases.EDGE_CHAINS if c[1] == "kmeans"],
                         ids=[c[0] for c in cases.EDGE_CHAINS if c[1] == "kmeans"])
def test_edge_cases_kmeans_batch_vs_specification(gpu, case):
    random.seed(2)
    np.random.seed(2)
    ref = cases.edge_build(_oracle_mods(), case)
    random.seed(2)
    np.random.seed(2)
    seg = cases.edge_build(_product_mods(sync="batch", n_stat_blocks=2), case)
    for sw in range(3):
        tot = no.kmeans_batch_sweep(ref, n_blocks=2)
        rec = seg.segment(1)
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), sw
        c, rc = seg.acoustic_model.components, ref.acoustic_model.components
        assert np.array_equal(c.assignments, rc.assignments), sw
        assert np.array_equal(c.means, rc.means), sw
        assert rec["sum_neg_len_sqrd_norm"][0] == tot
