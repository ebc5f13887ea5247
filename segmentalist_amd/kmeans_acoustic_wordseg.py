"""
Drop-in for segmentalist/kmeans_acoustic_wordseg.py: segmental k-means word segmentation.

`SegmentalKMeansWordseg` keeps the reference's constructor, attributes, methods and record
keys.  Two execution modes:

  sync="sequential" (default)  the reference's chain: utterances are visited one by one in
        `random.shuffle` order and the component statistics are updated between
        utterances, all on the device (score -> DP -> del/add/clean per utterance, enqueued
        asynchronously).  Bit-identical boundaries / assignments / means to the reference.
  sync="batch"                 every utterance of a sweep is scored against the statistics
        frozen at the start of the sweep (one MFMA GEMM over all embeddings), one workgroup
        per utterance runs the DP, and the statistics are rebuilt once per sweep in a fixed
        summation order; utterances shard over the ranks of a torch.distributed job.  A
        different (AD-LDA style) chain; bit-identical for 1/2/4/8 GPUs and to the CPU
        restatement of the same specification (oracle/np_oracle.py kmeans_batch_sweep).
        n_batches > 1: the sweep runs as that many steps, each resegmenting 1/n_batches of every
        statistics block against the means the previous step left (kmeans_minibatch_sweep):
        between the whole-sweep batch chain and the reference's per-utterance refresh.
"""
import ctypes as C
import logging
import os
import random
import time

import numpy as np

from . import _abi, rng
from .comm import get_comm
from .device import DeviceCorpus, KMeansBatchSweeper, Partition, to_dev
from .kmeans import KMeans, _consecutive
from .utterances import Utterances, process_embeddings

logger = logging.getLogger(__name__)
i_debug_monitor = 0          # kept for API compatibility (debug traces are not reproduced)
segment_debug_only = False


class SegmentalKMeansWordseg(object):
    def __init__(self, am_K, embedding_mats, vec_ids_dict, durations_dict, landmarks_dict,
                 seed_boundaries_dict=None, seed_assignments_dict=None, n_slices_min=0,
                 n_slices_max=20, min_duration=0, p_boundary_init=0.5, init_am_assignments="rand",
                 wip=0, sync="sequential", n_stat_blocks=8, flag_cap=None, process_group=None, n_batches=1,
                 shard_corpus=None):
        logger.info("Initializing")
        assert seed_assignments_dict is None or seed_boundaries_dict is not None
        assert sync in ("sequential", "batch")
        assert n_batches >= 1
        self.n_batches = int(n_batches)      # batch mode: statistics refreshed n_batches times per sweep (mini-batches)
        self.n_slices_min = n_slices_min
        self.n_slices_max = n_slices_max
        self.wip = wip
        self.sync = sync

        embeddings, vec_ids, ids_to_utterance_labels = process_embeddings(embedding_mats, vec_ids_dict)
        self.ids_to_utterance_labels = ids_to_utterance_labels
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

        # embeddings in the initial segmentation (kmeans_acoustic_wordseg.py:136-141)
        init_embeds = []
        for i in range(self.utterances.D):
            init_embeds.extend(self.utterances.get_segmented_embeds_i(i))
        init_embeds = np.array(init_embeds, dtype=int)
        init_embeds = init_embeds[np.where(init_embeds != -1)]
        logger.info("No. initial embeddings: " + str(init_embeds.shape[0]))

        assignments = -1 * np.ones(N, dtype=int)
        if seed_assignments_dict is not None:
            assert False, "to-do"                       # as the reference (:148-149)
        elif init_am_assignments == "rand":             # :183-194
            a = _consecutive(np.random.randint(0, am_K, len(init_embeds)))
            assignments[init_embeds] = a
        elif init_am_assignments == "spread":           # :196-205
            n = len(init_embeds)
            spread = (list(range(am_K)) * int(np.ceil(float(n) / am_K)))[:n]
            rng.shuffle(spread)
            assignments[init_embeds] = np.array(spread)
        elif init_am_assignments == "one-by-one":
            assert False, "to-do"                       # as the reference (:207-208)
        else:
            assert False, "invalid value for `init_am_assignments`: " + init_am_assignments

        # device images
        u = self.utterances
        # banded span tables for the DP kernels' fast path (windows of at most 8 slices, at most 64 landmarks)
        band = u.band_tables(n_slices_max) if (1 <= n_slices_max <= 8 and u.N_max <= 64) else None
        # Multi-rank batch mode ("each GPU keeps its shard's X rows"): this rank's device holds only the rows of
        # its own utterances -- the float32 matrix, both fp16 planes, the per-row work arrays -- numbered from 0; the span
        # tables name them by those local numbers.  shard_corpus=False keeps every row on every rank (needed to mix in
        # sequential-mode calls or KMeans.fit, which walk the whole matrix).
        comm = get_comm(process_group)
        if shard_corpus is None:
            shard_corpus = sync == "batch" and comm.world > 1
        shard = None
        if shard_corpus and comm.world > 1:
            assert sync == "batch", "a sharded corpus exists in batch mode only (the sequential chain does not shard)"
            pt = Partition(u.D, vec_ids.row_start, n_stat_blocks, comm.rank, comm.world)
            shard = (pt.row_lo, pt.row_hi)
            vi = np.full(np.asarray(u.vec_ids).shape, -1, dtype=np.int32)
            own = np.asarray(u.vec_ids)[pt.utt_lo:pt.utt_hi]
            vi[pt.utt_lo:pt.utt_hi] = np.where(own >= 0, own - pt.row_lo, -1)
            if band is not None:
                bi = np.full(band[0].shape, -1, dtype=np.int32)
                bown = band[0][pt.utt_lo:pt.utt_hi]
                bi[pt.utt_lo:pt.utt_hi] = np.where(bown >= 0, bown - pt.row_lo, -1)
                band = (bi, band[1])
            self._corpus = DeviceCorpus(embeddings[pt.row_lo:pt.row_hi], vi, u.durations, u.lengths, band=band)
        else:
            self._corpus = DeviceCorpus(embeddings, u.vec_ids, u.durations, u.lengths, band=band)
        self.acoustic_model = KMeans(embeddings, am_K, assignments, _corpus=self._corpus, _shard=shard)
        self._dk = self.acoustic_model.components.dev
        self._dev_bounds = to_dev(u.boundaries.astype(np.uint8))
        u.bind_device(self._dev_bounds, refresh=self._dk.ensure_boundaries)
        self._row_start = vec_ids.row_start

        # batch mode plumbing (single process unless torch.distributed is initialised)
        self._sweeper = None
        # flag_cap: tokens per statistics block and sweep that may found new components (more: an error, never a wrong result).
        # Default 4 096; 1 024 with a sharded corpus, whose records carry flag_cap embedding rows per block over the wire.
        if flag_cap is None:
            flag_cap = 1024 if shard is not None else 4096
        self._batch_args = (n_stat_blocks, flag_cap, comm)

    # ------------------------------------------------------------------ sequential mode
    def segment_i(self, i):
        """kmeans_acoustic_wordseg.py:225-332.  Returns the length-weighted objective of the
        utterance (computed with the means before the update, as in the reference)."""
        self._segment_i_async(i)
        self._dk.check_status()
        return float(self._dk.out_total[i].item())

    def _segment_i_async(self, i):
        self._dk.segment_utt_sequential(self._dev_bounds, i, self.n_slices_min, self.n_slices_max, self.wip)
        self.utterances.mark_device_dirty()

    def get_vec_embed_neg_len_sqrd_norms(self, vec_ids, durations):
        """kmeans_acoustic_wordseg.py:334-351 for arbitrary (vec_ids, durations) vectors."""
        vec_ids = np.asarray(vec_ids)
        out = -np.inf * np.ones(len(vec_ids))
        valid = np.where(vec_ids != -1)[0]
        if len(valid):
            mx, _, _ = self._dk.exact_max(vec_ids[valid].astype(np.int32))
            d = np.asarray(durations, dtype=np.float64)[valid]
            with np.errstate(invalid="ignore"):
                out[valid] = np.where(np.isnan(d), -np.inf, mx * d)
        return out + self.wip

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
            n_blocks, cap, group = self._batch_args
            comm = get_comm(group)
            part = Partition(self.utterances.D, self._row_start, n_blocks, comm.rank, comm.world)
            self._sweeper = KMeansBatchSweeper(self._dk, part, cap, comm)
        return self._sweeper

    def batch_sweep_async(self):
        """Enqueue one batch-synchronous sweep; results stay on the device."""
        if self.n_batches > 1:
            self._get_sweeper().sweep_minibatch(self._dev_bounds, self.n_slices_min, self.n_slices_max, self.wip, self.n_batches)
        else:
            self._get_sweeper().sweep(self._dev_bounds, self.n_slices_min, self.n_slices_max, self.wip)
        self.utterances.mark_device_dirty()

    # ------------------------------------------------------------------ driver
    def segment(self, n_iter, n_iter_inbetween_kmeans=0):
        """kmeans_acoustic_wordseg.py:353-426; same record keys."""
        import torch
        logger.info("Segmenting for " + str(n_iter) + " iterations")
        record_dict = {"sum_neg_sqrd_norm": [], "sum_neg_len_sqrd_norm": [], "components": [],
                       "sample_time": [], "n_tokens": []}
        c = self.acoustic_model.components
        for i_iter in range(n_iter):
            start_time = time.time()
            batch_rec = None
            if self.sync == "sequential":
                utt_order = list(range(self.utterances.D))
                rng.shuffle(utt_order)
                if segment_debug_only:
                    utt_order = [i_debug_monitor]
                # the whole chain of the sweep enqueued by one library call (float32 data); SEGK_SEQ_PER_UTT=1 keeps the
                # per-utterance calls (score through the filter machinery, DP, update + image refresh)
                if os.environ.get("SEGK_SEQ_PER_UTT", "0") == "1" or not self._dk.sequential_sweep(
                        self._dev_bounds, utt_order, self.n_slices_min, self.n_slices_max, self.wip):
                    for i_utt in utt_order:
                        self._segment_i_async(i_utt)
                self.utterances.mark_device_dirty()
                torch.cuda.synchronize()
                self._dk.check_status()
                totals = self._dk.out_total.cpu().numpy()
                # the reference's `sum += ...` over the utterances in visiting order: a running sum is that sequence of
                # additions (np.cumsum accumulates left to right; the Python loop over 10 000 numpy scalars took 2 ms)
                sum_neg_len_sqrd_norm = np.cumsum(totals[np.asarray(utt_order, dtype=np.int64)])[-1] if len(utt_order) else 0
            else:
                self.batch_sweep_async()
                pt = self._get_sweeper().part
                if pt.world == 1:
                    # every record value of the sweep from one device-to-host copy (the metric from the token lists)
                    sum_neg_len_sqrd_norm, n_comp, n_tok, sum_neg = self._dk.batch_record(pt.utt_lo, pt.utt_hi)
                    batch_rec = (sum_neg, n_comp, n_tok)
                else:
                    torch.cuda.synchronize()
                    self._dk.check_status()
                    sum_neg_len_sqrd_norm = float(self._dk.out_scalars[0].item())
            record_dict["sample_time"].append(time.time() - start_time)
            if batch_rec is not None:
                record_dict["sum_neg_sqrd_norm"].append(batch_rec[0])
                record_dict["sum_neg_len_sqrd_norm"].append(sum_neg_len_sqrd_norm)
                record_dict["components"].append(batch_rec[1])
                record_dict["n_tokens"].append(batch_rec[2])
            else:
                record_dict["sum_neg_sqrd_norm"].append(c.sum_neg_sqrd_norm())
                record_dict["sum_neg_len_sqrd_norm"].append(sum_neg_len_sqrd_norm)
                record_dict["components"].append(c.K)
                record_dict["n_tokens"].append(self.acoustic_model.get_n_assigned())
            info = "iteration: " + str(i_iter)
            for key in sorted(record_dict):
                info += ", " + key + ": " + str(record_dict[key][-1])
            logger.info(info)
            if n_iter_inbetween_kmeans > 0:
                self.acoustic_model.fit(n_iter_inbetween_kmeans, consider_unassigned=False)
        return record_dict

    def get_unsup_transcript_i(self, i):
        return list(self.acoustic_model.components.get_assignments(
            self.utterances.get_segmented_embeds_i(i)))

    def get_max_unsup_transcript_i(self, i):
        return self.acoustic_model.components.get_max_assignments(
            self.utterances.get_segmented_embeds_i(i))


def _dp_tri(kind, vec, N, n_slices_min, n_slices_max, log_p_continue=0.0, anneal_temp=1.0, uniforms=None):
    """Run one triangular-layout DP problem on the device (segk_dp_tri)."""
    import torch
    vec_t = to_dev(np.asarray(vec, dtype=np.float64))
    dev = vec_t.device
    Ns = torch.tensor([N], dtype=torch.int32, device=dev)
    offs = torch.zeros(1, dtype=torch.int64, device=dev)
    bounds = torch.zeros(N, dtype=torch.uint8, device=dev)
    totals = torch.zeros(1, dtype=torch.float64, device=dev)
    nd = torch.zeros(1, dtype=torch.int32, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    work = torch.zeros(3 * N + 2, dtype=torch.float64, device=dev)
    u_t = to_dev(np.asarray(uniforms, dtype=np.float64)) if uniforms is not None else None
    _abi.check(_abi.lib().segk_dp_tri(
        _abi.ctx(), kind, _abi.ptr(vec_t), _abi.ptr(Ns), _abi.ptr(offs), 1, int(n_slices_min),
        int(n_slices_max), float(log_p_continue), float(anneal_temp), _abi.ptr(u_t),
        0 if u_t is None else u_t.numel(), _abi.ptr(bounds), N, _abi.ptr(totals), _abi.ptr(nd), _abi.ptr(st),
        _abi.ptr(work), 3 * N + 2, _abi.stream()))
    return float(totals.item()), bounds.cpu().numpy().astype(bool), int(nd.item()), int(st.item())


def forward_backward_kmeans_viterbi(vec_embed_neg_len_sqrd_norms, N, n_slices_min=0, n_slices_max=0,
                                    i_utt=None):
    """kmeans_acoustic_wordseg.py:449-555 on the device."""
    tot, bounds, _, _ = _dp_tri(0, vec_embed_neg_len_sqrd_norms, N, n_slices_min, n_slices_max)
    return tot, bounds
