"""
state_dict() / load_state_dict() for the three segmenters (absent in the reference).

A checkpoint holds everything the chains depend on -- boundaries, component statistics exactly as
they stand on the device (the serial chain updates them incrementally, so they are saved, not
recomputed), language-model tables, the batch samplers' slot labels and sweep counter, and the
states of both host RNG streams -- as numpy arrays / plain Python objects.  Loading it into a
segmenter constructed with the same arguments and data resumes the chain bit for bit.
"""
import random

import numpy as np


def _np(t):
    return t.detach().cpu().numpy().copy()


def _put(t, a):
    import torch
    t.copy_(torch.from_numpy(np.ascontiguousarray(a)).to(t.dtype))


_KMEANS_FIELDS = ["means", "mean_numerators", "counts", "random_means", "assignments", "K"]
_FBGMM_FIELDS = ["stat_a", "stat_b", "log_prod", "pred", "kconst", "counts", "assignments", "K"]


def state_dict(seg):
    sd = {"class": type(seg).__name__, "boundaries": seg.utterances.boundaries.copy(),
          "py_random": random.getstate(), "np_random": np.random.get_state()}
    if hasattr(seg, "_dk"):                              # SegmentalKMeansWordseg
        dk = seg._dk
        # (multi-rank batch mode: a collective, like the boundaries above; a sharded corpus: every rank's rows assembled)
        assign = dk.global_assignments()
        for name in _KMEANS_FIELDS:
            sd["km_" + name] = assign.copy() if name == "assignments" else _np(getattr(dk, name))
    else:                                                # Unigram / Bigram drivers
        sw = seg._sweeper
        if sw is not None and sw.in_batch_state:
            seg.materialise()
            sd["batch_slot"] = _np(sw.slot)
            sd["batch_sweep_index"] = sw.sweep_index
            if sw.lm_tok is not None:
                sd["batch_lm_big"] = _np(sw.lm_big)
        df = seg._df
        for name in _FBGMM_FIELDS:
            sd["fb_" + name] = _np(getattr(df, name))
        if df.lm is not None:
            sd["lm_unigram"], sd["lm_bigram"] = _np(df.lm._unigram), _np(df.lm._bigram)
    return sd


def load_state_dict(seg, sd):
    assert sd["class"] == type(seg).__name__, "checkpoint of a %s" % sd["class"]
    seg.utterances.boundaries = sd["boundaries"]
    if hasattr(seg, "_dk"):
        dk = seg._dk
        for name in _KMEANS_FIELDS:
            if name == "assignments":
                dk.set_global_assignments(sd["km_" + name])
            else:
                _put(getattr(dk, name), sd["km_" + name])
        dk.assign_stale = None
        dk.bounds_stale = None
        dk.prepare()                                     # the MFMA operand image of the means
    else:
        df = seg._df
        for name in _FBGMM_FIELDS:
            _put(getattr(df, name), sd["fb_" + name])
        if df.lm is not None:
            _put(df.lm._unigram, sd["lm_unigram"])
            _put(df.lm._bigram, sd["lm_bigram"])
        if seg._sweeper is not None:
            seg._sweeper.invalidate()
        if "batch_slot" in sd:
            sw = seg._get_sweeper()
            sw.enter(seg._dev_bounds)                    # token lists, then the saved slot labels
            _put(sw.slot, sd["batch_slot"])
            if sw.lm_tok is not None:
                _put(sw.lm_big, sd["batch_lm_big"])
            sw.rebuild_from_slots(seg._dev_bounds)
            sw.sweep_index = int(sd["batch_sweep_index"])
    random.setstate(sd["py_random"])
    np.random.set_state(sd["np_random"])
