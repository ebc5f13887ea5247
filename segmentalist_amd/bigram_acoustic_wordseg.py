"""
Drop-in for segmentalist/bigram_acoustic_wordseg.py: word segmentation with a bigram language
model over component labels tied to fixed-variance Gaussian components -- on the device.

What runs where.  `gibbs_sample_i` is one asynchronous chain of kernels per utterance
(DeviceFbgmm.gibbs_utt): remove the utterance's LM counts, delete its segments, score every
candidate span with `lms*lm.log_prob_vec_i()` as assignment prior (log_marg_i_embed_unigram),
forward filtering / backward sampling, then assign the new segments one after the other with the
bigram prior of the previous segment's component, and add the new LM counts.  The RNG protocol
is UnigramAcousticWordseg's (a pre-drawn block of `random.random()` values consumed on the
device, host stream rewound and advanced by the count consumed).

As in the reference only fb_type="unigram" is functional (the bigram DP and its score vector are
stubs there, bigram_acoustic_wordseg.py:694-695,728-759) and only covariance_type="fixed" ties
the LM to the components (bigram_fbgmm.py:86-91).
"""
import logging
import os
import math
import random
import time

import numpy as np

from . import unigram_acoustic_wordseg
from .bigram_fbgmm import BigramFBGMM
from .bigram_lms import BigramSmoothLM
from .device import DeviceCorpus, to_dev
from .kmeans import _consecutive
from .unigram_acoustic_wordseg import UnigramAcousticWordseg
from .utterances import Utterances, process_embeddings  # noqa: F401

logger = logging.getLogger(__name__)
i_debug_monitor = 0
debug_gibbs_only = False


def forward_backward(vec_embed_log_probs, log_p_continue, N, n_slices_min=0, n_slices_max=0, i_utt=None,
                     anneal_temp=1):
    """bigram_acoustic_wordseg.py:728-759: a stub in the reference (returns None)."""
    return None


class BigramAcousticWordseg(object):
    def __init__(self, am_K, am_param_prior, lm_params, embedding_mats, vec_ids_dict, durations_dict,
                 landmarks_dict, seed_boundaries_dict=None, seed_assignments_dict=None, covariance_type="fixed",
                 n_slices_min=0, n_slices_max=20, min_duration=0, p_boundary_init=0.5, beta_sent_boundary=2.0,
                 lms=1., wip=0., fb_type="bigram", init_am_assignments="rand", time_power_term=1.,
                 sync="sequential", n_gibbs_blocks=8, n_stat_blocks=8, batch_seed=0, process_group=None,
                 score_precision="f64"):
        """Same arguments as the reference (bigram_acoustic_wordseg.py:129-136) plus the execution
        mode of UnigramAcousticWordseg (sync="batch": oracle/np_fbgmm_batch.py)."""
        logger.info("Initializing")
        assert sync in ("sequential", "batch")
        self.sync = sync
        self._batch_args = (n_gibbs_blocks, n_stat_blocks, batch_seed, process_group, score_precision)
        self._sweeper = None
        assert seed_assignments_dict is None or seed_boundaries_dict is not None
        self.n_slices_min = n_slices_min
        self.n_slices_max = n_slices_max
        self.beta_sent_boundary = beta_sent_boundary
        self.wip = wip
        self.lms = lms
        self.time_power_term = time_power_term
        self.set_fb_type(fb_type)

        embeddings, vec_ids, ids_to_utterance_labels = process_embeddings(embedding_mats, vec_ids_dict)
        self.ids_to_utterance_labels = ids_to_utterance_labels
        self._row_start = vec_ids.row_start
        N = embeddings.shape[0]
        seed_boundaries = None
        if seed_boundaries_dict is not None:
            seed_boundaries = [seed_boundaries_dict[i] for i in ids_to_utterance_labels]
        lengths = [len(landmarks_dict[i]) for i in ids_to_utterance_labels]
        landmarks = [landmarks_dict[i] for i in ids_to_utterance_labels]
        durations = [durations_dict[i] for i in ids_to_utterance_labels]
        self.utterances = Utterances(
            lengths, vec_ids, durations, landmarks, seed_boundaries=seed_boundaries,
            p_boundary_init=p_boundary_init, n_slices_min=n_slices_min, n_slices_max=n_slices_max,
            min_duration=min_duration)
        u = self.utterances
        # (banded span tables: what the segmentation kernels read when no embedding lies outside the window)
        self._corpus = DeviceCorpus(embeddings, u.vec_ids, u.durations, u.lengths, band=u.complete_band_tables(n_slices_max))

        init_embeds = []
        for i in range(u.D):
            init_embeds.extend(u.get_segmented_embeds_i(i))
        init_embeds = np.array(init_embeds, dtype=int)
        init_embeds = init_embeds[np.where(init_embeds != -1)]

        if lm_params["type"] == "smooth":                      # bigram_acoustic_wordseg.py:184-190
            intrp_lambda, a, b = lm_params["intrp_lambda"], lm_params["a"], lm_params["b"]
            self.lm = BigramSmoothLM(intrp_lambda, a, b, am_K)
        else:
            assert False, "invalid language model type: " + str(lm_params["type"])

        assignments = -1 * np.ones(N, dtype=int)
        if seed_assignments_dict is not None:
            assert False, "to-do"                                # :193 in the reference
        elif init_am_assignments == "rand":
            assignments[init_embeds] = _consecutive(np.random.randint(0, am_K, len(init_embeds)))
            self.acoustic_model = BigramFBGMM(embeddings, am_param_prior, am_K, assignments,
                                              covariance_type=covariance_type, lms=lms, lm=self.lm,
                                              _corpus=self._corpus)
        elif init_am_assignments == "one-by-one":
            assert False                                          # :235 in the reference
        else:
            assert False, "invalid value for `init_am_assignments`: " + init_am_assignments

        self._df = self.acoustic_model.components.dev
        self._dev_bounds = to_dev(u.boundaries.astype(np.uint8))
        u.bind_device(self._dev_bounds)
        self.set_lm_counts()

    # the RNG plumbing and the span-score helper are the unigram segmenter's
    _open_stream = UnigramAcousticWordseg._open_stream
    _close_stream = UnigramAcousticWordseg._close_stream
    calc_p_continue = UnigramAcousticWordseg.calc_p_continue
    get_unsup_transcript_i = UnigramAcousticWordseg.get_unsup_transcript_i
    _get_sweeper = UnigramAcousticWordseg._get_sweeper
    batch_sweep_async = UnigramAcousticWordseg.batch_sweep_async
    materialise = UnigramAcousticWordseg.materialise
    _leave_batch = UnigramAcousticWordseg._leave_batch

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        from . import checkpoint
        return checkpoint.state_dict(self)

    def load_state_dict(self, sd):
        from . import checkpoint
        checkpoint.load_state_dict(self, sd)

    def set_fb_type(self, fb_type):
        self.fb_type = fb_type
        if fb_type == "bigram":
            self.fb_func = forward_backward
            self.get_vec_embed_log_probs = self.get_vec_embed_log_probs_bigram
        elif fb_type == "unigram":
            self.fb_func = unigram_acoustic_wordseg.forward_backward
            self.get_vec_embed_log_probs = self.get_vec_embed_log_probs_unigram
        else:
            assert False, "invalid `fb_type`: " + fb_type

    def set_lm_counts(self):
        """bigram_acoustic_wordseg.py:271-276, for the whole corpus in one launch."""
        if self.acoustic_model.components.lm is None:
            for i_utt in range(self.utterances.D):
                self.lm.counts_from_utterance(self.get_unsup_transcript_i(i_utt))
        else:
            self._df.update(6, utt=-1, boundaries=self._dev_bounds)

    # ------------------------------------------------------------------ record metrics (host)
    def log_prob_z(self):
        """bigram_acoustic_wordseg.py:287-305.  The reference never advances `j_prev` inside its
        loop, so what it accumulates is the sequential unigram predictive probability of all
        tokens in utterance order; evaluated here in that form (same floating-point terms)."""
        comps = self.acoustic_model.components
        assignments = comps.assignments
        K = self.lm.K
        a = self.lm.a
        counts = np.zeros(K, np.int64)
        total = 0
        log_prob_z = 0.
        for i_utt in range(self.utterances.D):
            for i_cur in assignments[np.asarray(self.utterances.get_segmented_embeds_i(i_utt), dtype=int)]:
                log_prob_z += np.log((counts[i_cur] + float(a) / K) / (total + a))
                counts[i_cur] += 1
                total += 1
        return log_prob_z

    def log_marg(self):
        return self.log_prob_z() + self.acoustic_model.log_prob_X_given_z()

    # ------------------------------------------------------------------ per-embedding API
    def log_marg_i_embed_unigram(self, i_embed):
        """bigram_acoustic_wordseg.py:314-329, on the device."""
        assert i_embed != -1
        return float(self._df.log_marg_rows([i_embed])[0])

    def gibbs_sample_inside_loop_i_embed(self, i_embed, j_prev_assignment=None, anneal_temp=1, i_utt=None):
        """bigram_acoustic_wordseg.py:332-384; consumes one random.random(); returns the component."""
        k = self._df.assign_item(i_embed, random.random(), anneal_temp, map_assign=False,
                                 j_prev=j_prev_assignment)
        return k

    # ------------------------------------------------------------------ one utterance
    def _gibbs_i_async(self, i, anneal_temp, anneal_gibbs_am, assignments_only=False):
        assert assignments_only or self.fb_type == "unigram", "the bigram forward-backward is a stub in the reference"
        log_p_continue = 0.0 if assignments_only else math.log(self.calc_p_continue())
        self._df.gibbs_utt(self._dev_bounds, i, False, self.n_slices_min, self.n_slices_max, self.wip,
                           self.time_power_term, log_p_continue, anneal_temp,
                           anneal_temp if anneal_gibbs_am else 1.0, assignments_only=assignments_only)
        self.utterances.mark_device_dirty()

    def gibbs_sample_i(self, i, anneal_temp=1, anneal_gibbs_am=False, assignments_only=False):
        """bigram_acoustic_wordseg.py:386-551."""
        self._leave_batch()
        self._open_stream([i])
        self._gibbs_i_async(i, anneal_temp, anneal_gibbs_am, assignments_only)
        self._close_stream()
        self._df.check_status()
        return 0. if assignments_only else float(self._df.out_logprob[i].item())

    # ------------------------------------------------------------------ sweeps
    def gibbs_sample(self, n_iter, am_n_iter=0, anneal_schedule=None, anneal_start_temp_inv=0.1,
                     anneal_end_temp_inv=1, n_anneal_steps=-1, anneal_gibbs_am=False, assignments_only=False):
        """bigram_acoustic_wordseg.py:553-671; same record keys."""
        import torch
        if anneal_schedule is None:
            get_anneal_temp = iter([])
        elif anneal_schedule == "linear":
            if n_anneal_steps == -1:
                n_anneal_steps = n_iter
            get_anneal_temp = iter(1. / np.linspace(anneal_start_temp_inv, anneal_end_temp_inv, n_anneal_steps))
        elif anneal_schedule == "step":
            assert not n_anneal_steps == -1, "`n_anneal_steps` of -1 not allowed for step annealing schedule"
            n_iter_per_step = int(round(float(n_iter) / n_anneal_steps))
            anneal_list = 1. / np.linspace(anneal_start_temp_inv, anneal_end_temp_inv, n_anneal_steps)
            get_anneal_temp = iter(np.repeat(anneal_list, n_iter_per_step))
        else:
            assert False, "invalid anneal_schedule"

        record_dict = {k: [] for k in ["sample_time", "log_marg", "log_marg*length", "log_prob_z",
                                       "log_prob_X_given_z", "anneal_temp", "components", "n_tokens"]}
        am = self.acoustic_model
        for i_iter in range(n_iter):
            start_time = time.time()
            if am_n_iter > 0:
                assert False, "to-do"                             # :632 in the reference
            anneal_temp = next(get_anneal_temp, anneal_end_temp_inv)
            utt_order = list(range(self.utterances.D))
            from . import rng
            rng.shuffle(utt_order)
            if debug_gibbs_only:
                utt_order = [i_debug_monitor]
            if self.sync == "batch":
                assert self.fb_type == "unigram" and not assignments_only
                self.batch_sweep_async(anneal_temp, anneal_gibbs_am)
                torch.cuda.synchronize()
                self._df.check_status()
                log_prob = float(np.sum(self._get_sweeper().utt_values(self._df.out_logprob)))
                self.materialise()
            else:
                self._leave_batch()
                self._open_stream(utt_order)
                # the whole chain of the sweep by one library call where it applies (unigram_acoustic_wordseg.py does the same)
                whole = (os.environ.get("SEGK_SEQ_PER_UTT", "0") != "1" and not debug_gibbs_only and not assignments_only
                         and self._df.sequential_sweep(self._dev_bounds, utt_order, self._row_start, False, self.n_slices_min,
                                                       self.n_slices_max, self.wip, self.time_power_term,
                                                       math.log(self.calc_p_continue()), anneal_temp,
                                                       anneal_temp if anneal_gibbs_am else 1.0))
                if whole:
                    self.utterances.mark_device_dirty()
                else:
                    for i_utt in utt_order:
                        self._gibbs_i_async(i_utt, anneal_temp, anneal_gibbs_am, assignments_only)
                torch.cuda.synchronize()
                self._close_stream()
                self._df.check_status()
                lps = self._df.out_logprob.cpu().numpy()
                log_prob = 0
                for i_utt in utt_order:
                    log_prob += 0. if assignments_only else lps[i_utt]

            record_dict["sample_time"].append(time.time() - start_time)
            # the record metrics (bigram_acoustic_wordseg.py:287-305, components.log_marg) in one device call instead of a
            # Python loop over every token and numpy on host snapshots; SEGK_HOST_METRICS=1 keeps the host expressions
            if os.environ.get("SEGK_HOST_METRICS", "0") == "1":
                lpz, lpx, n_comp, n_tok = self.log_prob_z(), am.log_prob_X_given_z(), am.components.K, am.get_n_assigned()
            else:
                lpz, lpx, n_comp, n_tok = self._df.record_metrics(urn=True, urn_a=self.lm.a)
            record_dict["log_marg"].append(lpz + lpx)
            record_dict["log_marg*length"].append(log_prob)
            record_dict["log_prob_z"].append(lpz)
            record_dict["log_prob_X_given_z"].append(lpx)
            record_dict["anneal_temp"].append(anneal_temp)
            record_dict["components"].append(n_comp)
            record_dict["n_tokens"].append(n_tok)
            info = "iteration: " + str(i_iter)
            for key in sorted(record_dict):
                info += ", " + key + ": " + str(record_dict[key][-1])
            logger.info(info)
        return record_dict

    # ------------------------------------------------------------------ helpers of the reference API
    def get_vec_embed_log_probs_unigram(self, vec_ids, durations):
        """bigram_acoustic_wordseg.py:673-692."""
        vec_ids = np.asarray(vec_ids)
        out = -np.inf * np.ones(len(vec_ids))
        valid = np.where(vec_ids != -1)[0]
        if len(valid):
            lm = self._df.log_marg_rows(vec_ids[valid])
            d = np.asarray(durations, dtype=np.float64)[valid]
            with np.errstate(invalid="ignore"):
                out[valid] = np.where(np.isnan(d), -np.inf, lm * d ** self.time_power_term)
        return out + self.wip

    def get_vec_embed_log_probs_bigram(self, vec_ids, durations):
        """A stub in the reference (bigram_acoustic_wordseg.py:694-695)."""
        return None
