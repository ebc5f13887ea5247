"""
Drop-in for segmentalist/unigram_acoustic_wordseg.py: unigram word segmentation of speech with a
Bayesian GMM over acoustic word embeddings, blocked Gibbs sampling -- on the device.

RNG protocol.  The reference consumes the process-global Python `random`
stream: `random.shuffle(utt_order)` once per sweep, then per utterance one `random.random()` per
backward-sampling step followed by one per new segment assignment.  Here a block of uniforms is
pre-drawn from the same stream, handed to the device, consumed there through a cursor in exactly
that order, and afterwards the host stream is rewound and advanced by the number actually
consumed -- so seeded runs stay aligned with the reference draw for draw.
"""
import logging
import os
import math
import random
import time

import numpy as np

from . import rng
from .device import DeviceCorpus, to_dev
from .kmeans import _consecutive
from .kmeans_acoustic_wordseg import _dp_tri
from .utterances import Utterances, process_embeddings  # noqa: F401  (re-exported as in the reference)

logger = logging.getLogger(__name__)
i_debug_monitor = 0
debug_gibbs_only = False


class UnigramAcousticWordseg(object):
    def __init__(self, am_class, am_alpha, am_K, am_param_prior, embedding_mats, vec_ids_dict, durations_dict,
                 landmarks_dict, seed_boundaries_dict=None, seed_assignments_dict=None, covariance_type="fixed",
                 n_slices_min=0, n_slices_max=20, min_duration=0, p_boundary_init=0.5, beta_sent_boundary=2.0,
                 lms=1., wip=0., fb_type="standard", init_am_assignments="rand", time_power_term=1.,
                 sync="sequential", n_gibbs_blocks=8, n_stat_blocks=8, batch_seed=0, process_group=None,
                 score_precision="f64"):
        """Same arguments as the reference (unigram_acoustic_wordseg.py:107-123), plus the execution
        mode: sync="sequential" is the reference's serial chain (draw for draw); sync="batch" the
        batch-synchronous blocked Gibbs sampler specified in oracle/np_fbgmm_batch.py
        (`n_gibbs_blocks` steps per sweep, statistics summed over `n_stat_blocks` slices, sharded
        over the ranks of `process_group` when torch.distributed is initialised).
        score_precision="f32" / "f16" (batch mode, fixed-variance components) evaluates the span
        scores on the matrix cores (fp32 MFMA, or two-way fp16 splits of the fp32 operands on the
        16-bit pipe) -- within the 1e-4 tolerance of the path, 5x / 10x faster; with diagonal
        components "f32" evaluates the Student-t terms in float32 with the hardware logarithm (same
        tolerance); "f64" reproduces the specification to the last draw."""
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

        assignments = -1 * np.ones(N, dtype=int)
        am_kw = dict(covariance_type=covariance_type, lms=lms, _corpus=self._corpus)
        if seed_assignments_dict is not None:                     # unigram_acoustic_wordseg.py:176-204
            self.seed_to_cluster = {}
            i_cluster = 0
            for i_utt, utt in enumerate(ids_to_utterance_labels):
                utt_init_embeds = np.array(u.get_segmented_embeds_i(i_utt), dtype=int)
                utt_init_assignments = np.array(seed_assignments_dict[utt][:])
                utt_init_assignments = utt_init_assignments[np.where(utt_init_embeds != -1)]
                utt_init_embeds = utt_init_embeds[np.where(utt_init_embeds != -1)]
                for seed in utt_init_assignments:
                    if seed not in self.seed_to_cluster:
                        if isinstance(seed, (int, np.integer)):
                            self.seed_to_cluster[seed] = seed
                        else:
                            self.seed_to_cluster[seed] = i_cluster
                            i_cluster += 1
                assignments[utt_init_embeds] = [self.seed_to_cluster[i] for i in utt_init_assignments]
            if am_K is None:
                am_K = max(self.seed_to_cluster.values()) + 1
            else:
                assert am_K >= max(self.seed_to_cluster.values()) + 1
            self.acoustic_model = am_class(embeddings, am_param_prior, am_alpha, am_K, assignments, **am_kw)
        elif init_am_assignments == "rand":                       # :206-223
            assignments[init_embeds] = _consecutive(np.random.randint(0, am_K, len(init_embeds)))
            self.acoustic_model = am_class(embeddings, am_param_prior, am_alpha, am_K, assignments, **am_kw)
        elif init_am_assignments == "one-by-one":                 # :225-236
            self.acoustic_model = am_class(embeddings, am_param_prior, am_alpha, am_K, assignments, **am_kw)
            for i_embed in init_embeds:
                self.acoustic_model.gibbs_sample_inside_loop_i(i_embed)
        else:
            assert False, "invalid value for `init_am_assignments`: " + init_am_assignments

        self._df = self.acoustic_model.components.dev
        self._dev_bounds = to_dev(u.boundaries.astype(np.uint8))
        u.bind_device(self._dev_bounds)

    def set_fb_type(self, fb_type):
        self.fb_type = fb_type
        if fb_type == "standard":
            self.fb_func = forward_backward
        elif fb_type == "viterbi":
            self.fb_func = forward_backward_viterbi
        else:
            assert False, "invalid `fb_type`: " + fb_type

    # ------------------------------------------------------------------ RNG stream plumbing
    def _open_stream(self, utts):
        """Pre-draw the uniforms a visit of `utts` can consume at most (two per landmark)."""
        n = int(sum(2 * self.utterances.lengths[i] for i in utts)) + 2
        self._rng_state = random.getstate()
        self._df.set_uniform_stream(np.array([random.random() for _ in range(n)]))

    def _close_stream(self):
        used = int(self._df.ucursor.item())
        random.setstate(self._rng_state)
        for _ in range(used):
            random.random()
        return used

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        from . import checkpoint
        return checkpoint.state_dict(self)

    def load_state_dict(self, sd):
        from . import checkpoint
        checkpoint.load_state_dict(self, sd)

    # ------------------------------------------------------------------ batch mode
    def _get_sweeper(self):
        if self._sweeper is None:
            from .device import FbgmmBatchSweeper
            B, S, seed, group, prec = self._batch_args
            self._sweeper = FbgmmBatchSweeper(self._df, self._row_start, B, S, seed, group, prec)
        return self._sweeper

    def batch_sweep_async(self, anneal_temp=1, anneal_gibbs_am=False):
        """Enqueue one batch-synchronous sweep (fb_type "standard" sampling); results stay on the device."""
        self._get_sweeper().sweep(self._dev_bounds, self.n_slices_min, self.n_slices_max, self.wip,
                                  self.time_power_term, anneal_temp, anneal_temp if anneal_gibbs_am else 1.0)
        self.utterances.mark_device_dirty()

    def materialise(self):
        """Bring the reference's view (components.K / assignments / statistics) up to date with the
        batch state."""
        if self._sweeper is not None:
            self._sweeper.materialise(self._dev_bounds)
            self.utterances.mark_device_dirty()

    def _leave_batch(self):
        if self._sweeper is not None and self._sweeper.in_batch_state:
            self.materialise()
            self._sweeper.invalidate()

    # ------------------------------------------------------------------ one utterance
    def _gibbs_i_async(self, i, anneal_temp, anneal_gibbs_am):
        viterbi = self.fb_type == "viterbi"
        log_p_continue = math.log(self.calc_p_continue())
        self._df.gibbs_utt(self._dev_bounds, i, viterbi, self.n_slices_min, self.n_slices_max, self.wip,
                           self.time_power_term, log_p_continue, anneal_temp,
                           anneal_temp if anneal_gibbs_am else 1.0)
        self.utterances.mark_device_dirty()

    def gibbs_sample_i(self, i, anneal_temp=1, anneal_gibbs_am=False):
        """unigram_acoustic_wordseg.py:252-360."""
        self._leave_batch()
        self._open_stream([i])
        self._gibbs_i_async(i, anneal_temp, anneal_gibbs_am)
        self._close_stream()
        self._df.check_status()
        return float(self._df.out_logprob[i].item())

    # ------------------------------------------------------------------ sweeps
    def gibbs_sample(self, n_iter, am_n_iter=0, anneal_schedule=None, anneal_start_temp_inv=0.1,
                     anneal_end_temp_inv=1, n_anneal_steps=-1, anneal_gibbs_am=False):
        """unigram_acoustic_wordseg.py:362-472; same record keys."""
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
            if am_n_iter > 0:                                     # unigram_acoustic_wordseg.py:440-443
                self._leave_batch()
                self.acoustic_model.gibbs_sample(am_n_iter, consider_unassigned=False)
            anneal_temp = next(get_anneal_temp, anneal_end_temp_inv)
            if self.sync == "batch":
                assert self.fb_type == "standard", "batch mode samples boundaries (fb_type=\"standard\")"
                self.batch_sweep_async(anneal_temp, anneal_gibbs_am)
                torch.cuda.synchronize()
                self._df.check_status()
                log_prob = float(np.sum(self._get_sweeper().utt_values(self._df.out_logprob)))
                self.materialise()                                # the record metrics read the reference's view
            else:
                self._leave_batch()
                utt_order = list(range(self.utterances.D))
                rng.shuffle(utt_order)
                if debug_gibbs_only:
                    utt_order = [i_debug_monitor]
                self._open_stream(utt_order)
                # the whole chain of the sweep by one library call (a persistent kernel per stretch of utterances between two
                # emptied components) where it applies; SEGK_SEQ_PER_UTT=1 / SEGK_FB_CHAIN=0 keep the four launches per utterance
                whole = os.environ.get("SEGK_SEQ_PER_UTT", "0") != "1" and not debug_gibbs_only and self._df.sequential_sweep(
                    self._dev_bounds, utt_order, self._row_start, self.fb_type == "viterbi", self.n_slices_min, self.n_slices_max,
                    self.wip, self.time_power_term, math.log(self.calc_p_continue()), anneal_temp,
                    anneal_temp if anneal_gibbs_am else 1.0)
                if whole:
                    self.utterances.mark_device_dirty()
                else:
                    for i_utt in utt_order:
                        self._gibbs_i_async(i_utt, anneal_temp, anneal_gibbs_am)
                torch.cuda.synchronize()
                self._close_stream()
                self._df.check_status()
                lps = self._df.out_logprob.cpu().numpy()
                log_prob = 0
                for i_utt in utt_order:
                    log_prob += lps[i_utt]

            record_dict["sample_time"].append(time.time() - start_time)
            # the record metrics (fbgmm.py:208-225, components.log_marg) in one device call instead of numpy on host
            # snapshots of the assignments; SEGK_HOST_METRICS=1 keeps the host expressions (tests compare the two)
            if os.environ.get("SEGK_HOST_METRICS", "0") == "1":
                lpz, lpx, n_comp, n_tok = am.log_prob_z(), am.log_prob_X_given_z(), am.components.K, am.get_n_assigned()
            else:
                lpz, lpx, n_comp, n_tok = self._df.record_metrics()
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
    def get_vec_embed_log_probs(self, vec_ids, durations):
        """unigram_acoustic_wordseg.py:474-511."""
        vec_ids = np.asarray(vec_ids)
        out = -np.inf * np.ones(len(vec_ids))
        valid = np.where(vec_ids != -1)[0]
        if len(valid):
            lm = self._df.log_marg_rows(vec_ids[valid])
            d = np.asarray(durations, dtype=np.float64)[valid]
            with np.errstate(invalid="ignore"):
                out[valid] = np.where(np.isnan(d), -np.inf, lm * d ** self.time_power_term)
        return out + self.wip

    def calc_p_continue(self):
        """unigram_acoustic_wordseg.py:513-531 (only beta_sent_boundary == -1 works in the reference)."""
        if self.beta_sent_boundary != -1:
            assert False, "to check"
        return 1.0

    def get_unsup_transcript_i(self, i):
        return list(self.acoustic_model.components.get_assignments(self.utterances.get_segmented_embeds_i(i)))

    def get_log_margs_i(self, i):
        """unigram_acoustic_wordseg.py:539-564."""
        comps = self.acoustic_model.components
        segmented_embeds = self.utterances.get_segmented_embeds_i(i)
        assignments = comps.get_assignments(segmented_embeds)
        for e in segmented_embeds:
            if e == -1:
                continue
            comps.del_item(e)
        log_margs = [self.acoustic_model.log_marg_i(j) for j in segmented_embeds if j != -1]
        for e, k in zip(segmented_embeds, assignments):
            comps.add_item(e, k)
        return log_margs


def _with_rng_stream(N, run):
    """Pre-draw N+1 uniforms, run, rewind and advance by the number consumed."""
    state = random.getstate()
    u = [random.random() for _ in range(N + 1)]
    result, used = run(u)
    random.setstate(state)
    for _ in range(used):
        random.random()
    return result


def forward_backward(vec_embed_log_probs, log_p_continue, N, n_slices_min=0, n_slices_max=0, i_utt=None,
                     anneal_temp=1):
    """unigram_acoustic_wordseg.py:653-756 on the device (consumes `random` like the reference)."""
    def run(u):
        tot, bounds, nd, st = _dp_tri(2, vec_embed_log_probs, N, n_slices_min, n_slices_max, log_p_continue,
                                      anneal_temp, u)
        if st:
            raise AssertionError("log_prob == -inf")
        return (np.float64(tot), bounds), nd
    return _with_rng_stream(N, run)


def forward_backward_viterbi(vec_embed_log_probs, log_p_continue, N, n_slices_min=0, n_slices_max=0, i_utt=None,
                             anneal_temp=None):
    """unigram_acoustic_wordseg.py:759-864 on the device."""
    tot, bounds, _, _ = _dp_tri(1, vec_embed_log_probs, N, n_slices_min, n_slices_max)
    return tot, bounds
