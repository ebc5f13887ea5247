"""
Device-resident state of the hot path and thin wrappers over the C ABI (include/segk.h).

PyTorch is used for exactly three things here: allocating device buffers, host<->device
copies, and the stream handle.  All arithmetic of the hot path runs in libsegk.so.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import SEGK_F32, SEGK_F64, SegkError, check, ptr
from .comm import SingleComm, get_comm


def _torch():
    import torch
    return torch


def _dev():
    torch = _torch()
    if not torch.cuda.is_available():
        raise SegkError("segmentalist_amd needs an MI355X: no ROCm device is visible and there is no "
                        "CPU fallback for the hot path")
    return torch.device("cuda", torch.cuda.current_device())


def to_dev(a, dtype=None):
    torch = _torch()
    a = np.ascontiguousarray(a if dtype is None else np.asarray(a, dtype=dtype))
    return torch.from_numpy(a).to(_dev())


def _host_init_stats(X, assignments, K_max, random_means):
    """KMeansComponents.__init__'s statistics (kmeans_components.py:79-81: add_item for k ascending, i ascending within k) on
    the host, exactly as k_kmeans_init_stats computes them on the device -- float64 sums of a component's rows in ascending row
    order, means = the float64 quotient rounded to the dtype of X, empty slots = the random means --, for the ranks of a
    sharded corpus, none of which holds every row on its device."""
    X = np.asarray(X)
    a = np.asarray(assignments, dtype=np.int64)
    D = X.shape[1]
    numer = np.zeros((K_max, D), dtype=np.float64)
    counts = np.zeros(K_max, dtype=np.int64)
    means = np.array(random_means, dtype=X.dtype, copy=True)
    rows = np.flatnonzero((a >= 0) & (a < K_max))
    order = rows[np.argsort(a[rows], kind="stable")]          # by component, ascending rows inside a component
    ks, starts = np.unique(a[order], return_index=True)
    ends = np.append(starts[1:], len(order))
    for k, s, e in zip(ks, starts, ends):
        numer[k] = np.cumsum(X[order[s:e]].astype(np.float64), axis=0)[-1]      # cumsum: strictly sequential additions
        counts[k] = e - s
        means[k] = (numer[k] / float(e - s)).astype(X.dtype)
    K = int(ks.max()) + 1 if len(ks) else 0
    return means, numer, counts, K


class DeviceCorpus(object):
    """Device image of the embedding matrix and (optionally) of `Utterances`
    (utterances.py:74-105): vec_ids int32 [n_utt, tri], durations f64, lengths int32."""

    def __init__(self, X, vec_ids=None, durations=None, lengths=None, band=None):
        torch = _torch()
        X = np.asarray(X)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        assert X.ndim == 2
        self.x_np_dtype = X.dtype
        self.x_dtype = SEGK_F32 if X.dtype == np.float32 else SEGK_F64
        self.n_emb, self.D = X.shape
        if self.n_emb >= 2 ** 31:
            raise SegkError("more than 2^31 embeddings are not supported (int32 row ids)")
        self.ld32 = (self.D + 3) // 4 * 4
        dev = _dev()
        if self.x_dtype == SEGK_F32 and self.ld32 == self.D:
            self.X = to_dev(X)
            self.X32 = self.X
            self.ldx = self.D
            x32_out = None
        else:
            self.X = to_dev(X)
            self.ldx = self.D
            self.X32 = torch.empty((self.n_emb, self.ld32), dtype=torch.float32, device=dev)
            x32_out = self.X32
        self.xnorm = torch.empty(self.n_emb, dtype=torch.float32, device=dev)
        if vec_ids is not None:
            vec_ids = np.asarray(vec_ids)
            self.n_utt = vec_ids.shape[0]
            self.lengths_np = np.asarray(lengths, dtype=np.int32)
            self.N_max = int(self.lengths_np.max())
            tri = self.N_max * (self.N_max + 1) // 2
            assert vec_ids.shape[1] == tri
            self.vec_ids = to_dev(vec_ids, np.int32)
            self.durations = to_dev(durations, np.float64)
            self.lengths = to_dev(self.lengths_np)
        else:
            self.n_utt, self.N_max = 0, 0
            self.vec_ids = self.durations = self.lengths = None
            self.lengths_np = None
        self.tri = self.N_max * (self.N_max + 1) // 2
        # banded span tables (ids int32 [n_utt, N_max, W], durations f64): what the per-utterance kernels read
        self.band_ids = self.band_dur = None
        self.band_W = 0
        if band is not None and vec_ids is not None:
            bi, bd = band
            assert bi.shape == bd.shape == (self.n_utt, self.N_max, bi.shape[2])
            self.band_ids, self.band_dur, self.band_W = to_dev(bi, np.int32), to_dev(bd, np.float64), int(bi.shape[2])
        # bf16x3 image of the rows for the k-means filter (float32 data, 8 <= D <= 128); built on demand
        self.Xb3 = None
        self.c = _abi.Corpus(
            X=self.X.data_ptr(), X32=self.X32.data_ptr(), x_dtype=self.x_dtype, D=self.D,
            n_emb=self.n_emb, ldx=self.ldx, ld32=self.ld32, xnorm=self.xnorm.data_ptr(),
            vec_ids=self.vec_ids.data_ptr() if self.vec_ids is not None else None,
            durations=self.durations.data_ptr() if self.durations is not None else None,
            lengths=self.lengths.data_ptr() if self.lengths is not None else None,
            n_utt=self.n_utt, N_max=self.N_max, band_W=self.band_W,
            band_ids=self.band_ids.data_ptr() if self.band_ids is not None else None,
            band_dur=self.band_dur.data_ptr() if self.band_dur is not None else None)
        check(_abi.lib().segk_corpus_prepare(_abi.ctx(), C.byref(self.c), ptr(x32_out), ptr(self.xnorm),
                                             _abi.stream()))

    def ensure_b3(self):
        """The rows split into 16-bit pieces (segk_corpus_prepare_b3) for the split-precision k-means
        filter: fp16x2 by default, bf16x3 with SEGK_SCORE_B3=3, none (fp32 MFMA filter) with 0."""
        import os
        pieces = int(os.environ.get("SEGK_SCORE_B3", "2") or 2)
        if pieces not in (2, 3):
            return False
        if self.Xb3 is None and self.x_dtype == SEGK_F32 and 8 <= self.D <= 128:
            torch = _torch()
            nbytes = int(_abi.lib().segk_corpus_b3_bytes(self.n_emb, self.D))
            self.Xb3 = torch.empty(nbytes, dtype=torch.uint8, device=self.X.device)
            check(_abi.lib().segk_corpus_prepare_b3(_abi.ctx(), C.byref(self.c), ptr(self.Xb3), pieces, _abi.stream()))
            self.c.Xb3 = self.Xb3.data_ptr()
            self.c.sp_pieces = pieces
        return self.Xb3 is not None

    @property
    def torch_xdtype(self):
        torch = _torch()
        return torch.float32 if self.x_dtype == SEGK_F32 else torch.float64


class DeviceKMeans(object):
    """Device image of `KMeansComponents` (kmeans_components.py:18-91) plus the derived MFMA
    operands and the per-sweep work buffers."""

    def __init__(self, corpus, K_max, assignments, random_means, shard=None):
        torch = _torch()
        dev = _dev()
        self.corpus = corpus
        c = corpus
        self.K_max = int(K_max)
        # shard = (row_lo, row_hi, X_host): `corpus` holds the rows [row_lo, row_hi) of the embedding matrix, numbered from 0
        # (multi-rank batch mode, SURVEY 8(e): "each GPU keeps its shard's X rows").  Everything row-indexed on the device --
        # the span tables' ids, token lists, candidates, `assignments` -- is then in that LOCAL numbering; the host-facing
        # views (components.assignments, checkpoints) are assembled from all ranks.  Component statistics are replicated.
        self.row_base, self.n_rows_global, self.shard = 0, c.n_emb, None
        assignments = np.asarray(assignments)
        init = None
        if shard is not None:
            lo, hi, X_host = shard
            assert hi - lo == c.n_emb and len(assignments) == X_host.shape[0]
            self.row_base, self.n_rows_global, self.shard = lo, len(assignments), (lo, hi)
            init = _host_init_stats(X_host, assignments, self.K_max, np.asarray(random_means, dtype=c.x_np_dtype))
            assignments = assignments[lo:hi]
        xd = c.torch_xdtype
        self.means = torch.empty((self.K_max, c.D), dtype=xd, device=dev)
        self.mean_numerators = torch.zeros((self.K_max, c.D), dtype=torch.float64, device=dev)
        self.counts = torch.zeros(self.K_max, dtype=torch.int64, device=dev)
        self.random_means = to_dev(np.asarray(random_means, dtype=c.x_np_dtype))
        self.assignments = to_dev(assignments, np.int32)
        self.K = torch.zeros(1, dtype=torch.int32, device=dev)
        n_tiles_f = _abi.lib().segk_kmeans_tiles_floats(self.K_max, c.D)
        self.tiles = torch.zeros(int(n_tiles_f), dtype=torch.float32, device=dev)
        self.mnorm_max = torch.zeros(1, dtype=torch.float64, device=dev)
        self.tiles_b3 = None
        if c.ensure_b3():
            self.tiles_b3 = torch.zeros(int(_abi.lib().segk_kmeans_tiles_b3_floats(self.K_max, c.D)), dtype=torch.float32,
                                        device=dev)
        self.m = _abi.KMeansDev(
            tiles_b3=self.tiles_b3.data_ptr() if self.tiles_b3 is not None else None,
            means=self.means.data_ptr(), mean_numerators=self.mean_numerators.data_ptr(),
            counts=self.counts.data_ptr(), random_means=self.random_means.data_ptr(),
            assignments=self.assignments.data_ptr(), K=self.K.data_ptr(), K_max=self.K_max,
            tiles=self.tiles.data_ptr(), mnorm_max=self.mnorm_max.data_ptr())
        # candidates of the filter stage, indexed by embedding row
        self.cand_k = torch.zeros(c.n_emb, dtype=torch.int32, device=dev)
        self.cand_f = torch.zeros((c.n_emb, 2), dtype=torch.float32, device=dev)
        self.cand_s = torch.zeros(c.n_emb, dtype=torch.float64, device=dev)
        self.cand_queue = torch.zeros(c.n_emb, dtype=torch.int32, device=dev)
        self.cand_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.cand = _abi.CandDev(k=self.cand_k.data_ptr(), f=self.cand_f.data_ptr(), s=self.cand_s.data_ptr(),
                                 queue=self.cand_queue.data_ptr(), count=self.cand_count.data_ptr())
        self.status = torch.zeros(8, dtype=torch.int32, device=dev)
        self.assign_stale = None
        # multi-rank batch sweeps: the communicator of the sweeper (ensure_assignments / ensure_boundaries are
        # collectives on it) and the boundary buffer whose rows of other ranks' utterances are out of date
        self.batch_comm = SingleComm()
        self.bounds_stale = None
        self._L = _abi.lib()
        self._ctx = _abi.ctx()
        if init is None:
            check(self._L.segk_kmeans_init_stats(self._ctx, C.byref(c.c), C.byref(self.m), _abi.stream()))
        else:
            means, numer, counts, K = init
            self.means.copy_(torch.from_numpy(means))
            self.mean_numerators.copy_(torch.from_numpy(numer))
            self.counts.copy_(torch.from_numpy(counts))
            self.K.fill_(int(K))
            self.prepare()
        if c.n_utt:
            self._alloc_utt_buffers()

    # ------------------------------------------------------------------ buffers for segmentation
    def _alloc_utt_buffers(self):
        torch = _torch()
        dev = _dev()
        c = self.corpus
        nu, nm = c.n_utt, c.N_max
        self.old_tok = torch.zeros((nu, nm), dtype=torch.int32, device=dev)
        self.new_tok = torch.zeros((nu, nm), dtype=torch.int32, device=dev)
        self.new_k = torch.zeros((nu, nm), dtype=torch.int32, device=dev)
        self.n_old = torch.zeros(nu, dtype=torch.int32, device=dev)
        self.n_new = torch.zeros(nu, dtype=torch.int32, device=dev)
        self.n_flag = torch.zeros(nu, dtype=torch.int32, device=dev)
        self.out_total = torch.zeros(nu, dtype=torch.float64, device=dev)
        self.utt_arange = torch.arange(nu, dtype=torch.int32, device=dev)
        self.remap = torch.zeros(self.K_max, dtype=torch.int32, device=dev)
        self.out_scalars = torch.zeros(8, dtype=torch.float64, device=dev)
        # batch sweeps leave `assignments` untouched: `assign_stale` = (lo, hi, world) of the token
        # lists it must be rebuilt from, or None when it is current

    # ------------------------------------------------------------------ single calls
    def _cp(self):
        return C.byref(self.corpus.c)

    def require_whole_corpus(self, what):
        if self.shard is not None:
            raise SegkError("%s needs every row of the embedding matrix on this device, but the segmenter holds a shard (rows "
                            "%d..%d of %d): construct it with shard_corpus=False" % (what, self.shard[0], self.shard[1],
                                                                                     self.n_rows_global))

    def global_assignments(self):
        """`assignments` of ALL rows as a numpy array (a collective on the sweeper's communicator when anything is stale or
        the corpus is sharded: every rank contributes the labels of its own rows)."""
        self.ensure_assignments()
        a = self.assignments.cpu().numpy()
        if self.shard is None:
            return a
        out = np.full(self.n_rows_global, -1, dtype=a.dtype)
        for lo, hi, part in self.batch_comm.all_gather_object((self.shard[0], self.shard[1], a)):
            out[lo:hi] = part
        return out

    def set_global_assignments(self, a):
        """Upload `assignments` given for all rows (checkpoint resume)."""
        torch = _torch()
        a = np.asarray(a)
        if self.shard is not None:
            a = a[self.shard[0]:self.shard[1]]
        self.assignments.copy_(torch.from_numpy(np.ascontiguousarray(a)).to(self.assignments.dtype))

    def ensure_assignments(self, group=None):
        """Materialise `assignments` after batch sweeps (they only maintain the token lists).  With more than
        one rank this is a COLLECTIVE on the sweeper's communicator (every rank holds the tokens of its own
        utterances only): all ranks must call it -- directly or through `components.assignments`,
        `sum_neg_sqrd_norm()`, `state_dict()` ... -- together."""
        if self.assign_stale is None:
            return
        lo, hi, world = self.assign_stale
        check(self._L.segk_kmeans_assignments_from_tokens(self._ctx, self._cp(), C.byref(self.m), lo, hi,
                                                          ptr(self.new_tok), ptr(self.new_k), ptr(self.n_new),
                                                          _abi.stream()))
        if world > 1 and self.shard is None:
            (self.batch_comm if group is None else get_comm(group)).all_reduce_max(self.assignments)
        self.assign_stale = None

    def ensure_boundaries(self):
        """After multi-rank batch sweeps every rank has resegmented its own utterances only: fetch the rows of
        the others (COLLECTIVE on the sweeper's communicator, like ensure_assignments), so that
        `utterances.boundaries`, transcripts and checkpoints are the same, complete state on every rank."""
        if self.bounds_stale is None:
            return
        bounds, pt = self.bounds_stale
        torch = _torch()
        mine = torch.zeros_like(bounds)
        mine[pt.utt_lo:pt.utt_hi] = bounds[pt.utt_lo:pt.utt_hi]
        self.batch_comm.all_reduce_max(mine)
        bounds.copy_(mine)
        self.bounds_stale = None

    def ensure_state(self):
        """Both of the above: what the sequential-mode entry points need after multi-rank batch sweeps (they read the
        old boundaries and labels of utterances other ranks own).  A collective when anything is stale."""
        self.ensure_assignments()
        self.ensure_boundaries()

    def prepare(self):
        check(self._L.segk_kmeans_prepare(self._ctx, self._cp(), C.byref(self.m), _abi.stream()))

    def score_rows(self, ids=None, row0=0, n=None, hint_remap=None):
        """A1 over rows (device int32 tensor `ids`, or the range row0..row0+n): afterwards
        cand_k / cand_s hold np.argmax / np.max of neg_sqrd_norm for those rows.
        hint_remap (device int32 [K_max]): cand_k of these rows holds what the previous call left there, in the labelling
        that table translates into the current one -- the library then verifies those hints against the dense filter values
        instead of tracking the winner's index (segk_kmeans_score_hinted); the results are the same bits either way."""
        if ids is not None:
            n = ids.numel()
            p = ptr(ids)
        else:
            p = None
            n = self.corpus.n_emb - row0 if n is None else n
        if hint_remap is not None:
            check(self._L.segk_kmeans_score_hinted(self._ctx, self._cp(), C.byref(self.m), p, int(row0), int(n),
                                                   C.byref(self.cand), ptr(hint_remap), ptr(self.status), _abi.stream()))
            return
        check(self._L.segk_kmeans_score(self._ctx, self._cp(), C.byref(self.m), p, int(row0), int(n),
                                        C.byref(self.cand), ptr(self.status), _abi.stream()))

    def score_ptr(self, ids_ptr, n):
        check(self._L.segk_kmeans_score(self._ctx, self._cp(), C.byref(self.m), C.c_void_p(ids_ptr), 0, int(n),
                                        C.byref(self.cand), ptr(self.status), _abi.stream()))

    def exact_max(self, ids):
        """np.max / np.argmax of neg_sqrd_norm for rows `ids` (host ints) ->
        (float64[n], int32[n], number of rows that needed the full scan)."""
        torch = _torch()
        ids_t = to_dev(self._local_rows(ids), np.int32)
        n = ids_t.numel()
        out_max = torch.empty(n, dtype=torch.float64, device=ids_t.device)
        out_arg = torch.empty(n, dtype=torch.int32, device=ids_t.device)
        self.score_rows(ids_t)
        check(self._L.segk_kmeans_exact_max(self._ctx, self._cp(), C.byref(self.m), ptr(ids_t), n,
                                            C.byref(self.cand), ptr(out_max), ptr(out_arg), _abi.stream()))
        return out_max.cpu().numpy(), out_arg.cpu().numpy(), int(self.cand_count.item())

    def _local_rows(self, ids):
        """Global row ids -> the device's numbering (identity unless the corpus is sharded: then they must lie in the shard)."""
        if self.shard is None:
            return ids
        a = np.asarray(ids, dtype=np.int64)
        if a.size and (a.min() < self.shard[0] or a.max() >= self.shard[1]):
            self.require_whole_corpus("scoring rows outside this rank's shard")
        return a - self.shard[0]

    def neg_sqrd_norm(self, row):
        torch = _torch()
        row = int(self._local_rows([row])[0])
        out = torch.empty(self.K_max, dtype=self.corpus.torch_xdtype, device=self.means.device)
        check(self._L.segk_kmeans_neg_sqrd_norm(self._ctx, self._cp(), C.byref(self.m), int(row), ptr(out),
                                                _abi.stream()))
        return out.cpu().numpy()

    def add_item(self, i, k):
        self.require_whole_corpus("add_item")
        self.ensure_assignments()
        check(self._L.segk_kmeans_add_item(self._ctx, self._cp(), C.byref(self.m), int(i), int(k),
                                           ptr(self.status), _abi.stream()))

    def del_item(self, i):
        self.require_whole_corpus("del_item")
        self.ensure_assignments()
        check(self._L.segk_kmeans_del_item(self._ctx, self._cp(), C.byref(self.m), int(i), ptr(self.status),
                                           _abi.stream()))

    def del_component(self, k):
        self.require_whole_corpus("del_component")
        self.ensure_assignments()
        check(self._L.segk_kmeans_del_component(self._ctx, self._cp(), C.byref(self.m), int(k),
                                                ptr(self.status), _abi.stream()))

    def clean_components(self):
        self.require_whole_corpus("clean_components")
        self.ensure_assignments()
        check(self._L.segk_kmeans_clean_components(self._ctx, self._cp(), C.byref(self.m), ptr(self.status),
                                                   _abi.stream()))

    def sum_neg_sqrd_norm(self):
        self.ensure_assignments()
        torch = _torch()
        out = torch.zeros(1, dtype=torch.float64, device=self.means.device)
        check(self._L.segk_kmeans_sum_neg_sqrd_norm(self._ctx, self._cp(), C.byref(self.m), ptr(out),
                                                    _abi.stream()))
        if self.shard is not None:      # every rank's rows, added in rank order (a record metric: tolerance-level parity)
            return float(sum(self.batch_comm.all_gather_object(float(out.item()))))
        return float(out.item())

    # ------------------------------------------------------------------ segmentation
    def segment(self, boundaries, n_slices_min, n_slices_max, wip, utt=None, utt0=0, n_utts=None):
        """A5+A8+argmax for one utterance (`utt`) or the range utt0..utt0+n_utts."""
        c = self.corpus
        if utt is not None:
            up = C.c_void_p(self.utt_arange.data_ptr() + 4 * int(utt))
            utt0, n = 0, 1
        else:
            up = None
            n = c.n_utt - utt0 if n_utts is None else n_utts
        check(self._L.segk_kmeans_segment(
            self._ctx, self._cp(), C.byref(self.m), up, int(utt0), int(n), int(n_slices_min), int(n_slices_max),
            float(wip), C.byref(self.cand), ptr(boundaries), ptr(self.old_tok),
            ptr(self.new_tok), ptr(self.new_k), ptr(self.n_old), ptr(self.n_new), ptr(self.n_flag),
            ptr(self.out_total), ptr(self.status), _abi.stream()))

    def segment_utt_sequential(self, boundaries, i, n_slices_min, n_slices_max, wip):
        """The whole of segment_i (kmeans_acoustic_wordseg.py:225-332) for utterance i, enqueued
        asynchronously: score its spans, DP, del/add/clean in the reference's order."""
        c = self.corpus
        self.require_whole_corpus("the sequential chain (segment_i)")
        self.ensure_state()
        N = int(c.lengths_np[i])
        tri_i = N * (N + 1) // 2
        self.score_ptr(c.vec_ids.data_ptr() + 4 * i * c.tri, tri_i)
        self.segment(boundaries, n_slices_min, n_slices_max, wip, utt=i)
        check(self._L.segk_kmeans_update_utt(self._ctx, self._cp(), C.byref(self.m), int(i), ptr(self.old_tok),
                                             ptr(self.new_tok), ptr(self.new_k), ptr(self.n_old),
                                             ptr(self.n_new), ptr(self.status), _abi.stream()))

    def sequential_sweep(self, boundaries, order, n_slices_min, n_slices_max, wip):
        """segment_i for every utterance of `order` in turn by ONE library call (segk_kmeans_sequential_sweep): where the
        configuration allows, one persistent kernel per stretch of utterances between two emptied components -- the call
        then SYNCHRONISES the stream after every such launch --, otherwise three launches per utterance (direct exact score,
        DP, update), all enqueued.  Returns False when the library does not support the data (float64): the caller then
        walks the utterances itself."""
        if self.corpus.x_dtype != SEGK_F32 or self.corpus.N_max > 63:
            return False
        torch = _torch()
        self.require_whole_corpus("the sequential chain (segment)")
        self.ensure_state()
        if getattr(self, "_seq_keys", None) is None:
            self._seq_keys = torch.zeros(self.corpus.tri + 2, dtype=torch.int64, device=self.means.device)
        arr = np.ascontiguousarray(order, dtype=np.int32)       # (a ctypes array built element by element: 2 ms of a 135 ms sweep)
        check(self._L.segk_kmeans_sequential_sweep(
            self._ctx, self._cp(), C.byref(self.m), arr.ctypes.data, len(order), int(n_slices_min), int(n_slices_max), float(wip),
            C.byref(self.cand), ptr(self._seq_keys), ptr(boundaries), ptr(self.old_tok), ptr(self.new_tok), ptr(self.new_k),
            ptr(self.n_old), ptr(self.n_new), ptr(self.n_flag), ptr(self.out_total), ptr(self.status), _abi.stream()))
        return True

    def batch_record(self, lo, hi):
        """The record values of the batch sweep just enqueued, in ONE device-to-host copy (segk_kmeans_batch_record):
        (sum_neg_len_sqrd_norm, K, n_tokens, sum_neg_sqrd_norm); raises like check_status."""
        check(self._L.segk_kmeans_batch_record(self._ctx, self._cp(), C.byref(self.m), int(lo), int(hi), ptr(self.new_tok),
                                               ptr(self.new_k), ptr(self.status), ptr(self.out_scalars), _abi.stream()))
        o = self.out_scalars.cpu().numpy()
        self._raise_status(int(o[5]))
        return float(o[0]), int(o[1]), int(o[2]), float(o[4])

    def _raise_status(self, bits):
        if bits & 1:
            raise AssertionError("a new segment has no embedding (vec_id == -1): the reference asserts in "
                                 "KMeansComponents.add_item (kmeans_components.py:100)")
        if bits & 2:
            raise AssertionError("add_item on an item that is already assigned (kmeans_components.py:101)")
        if bits & 4:
            raise SegkError("batch sweep: in one statistics block more new tokens chose an inactive component than "
                            "`flag_cap`; the tokens beyond were dropped, so the statistics of this sweep are not usable: "
                            "rebuild the segmenter (or load a checkpoint) with a larger flag_cap")

    def check_status(self):
        st = self.status.cpu().numpy()
        self._raise_status(int(st[0]))
        return int(st[1])


class Partition(object):
    """Static split of the utterances into `n_blocks` statistics blocks (the fixed summation
    tree of the batch mode) and of the blocks over `world` ranks (rank r owns a contiguous
    run of blocks, hence of utterances and of embedding rows)."""

    def __init__(self, n_utt, utt_row_start, n_blocks=8, rank=0, world=1):
        if n_blocks % world != 0:
            raise SegkError("the number of statistics blocks (%d) must be a multiple of the number of "
                            "ranks (%d)" % (n_blocks, world))
        self.n_blocks, self.rank, self.world = n_blocks, rank, world
        self.nbl = n_blocks // world
        self.bounds = np.array([(b * n_utt) // n_blocks for b in range(n_blocks + 1)], dtype=np.int32)
        self.utt_lo = int(self.bounds[rank * self.nbl])
        self.utt_hi = int(self.bounds[(rank + 1) * self.nbl])
        self.row_lo = int(utt_row_start[self.utt_lo])
        self.row_hi = int(utt_row_start[self.utt_hi])
        self.local_bounds = self.bounds[rank * self.nbl:(rank + 1) * self.nbl + 1].copy()
        self._row_start = utt_row_start

    def minibatch(self, n_batches):
        """Step j of a mini-batch sweep resegments run j of EVERY statistics block (oracle/np_oracle.py minibatch_ranges):
        for this rank's blocks -> per step (utterance ids int32, embedding rows int32, contiguous?) as numpy arrays."""
        steps = []
        for j in range(n_batches):
            utts, rows = [], []
            for b in range(self.rank * self.nbl, (self.rank + 1) * self.nbl):
                lo_b, hi_b = int(self.bounds[b]), int(self.bounds[b + 1])
                lo = lo_b + (j * (hi_b - lo_b)) // n_batches
                hi = lo_b + ((j + 1) * (hi_b - lo_b)) // n_batches
                utts.append(np.arange(lo, hi, dtype=np.int32))
                rows.append(np.arange(int(self._row_start[lo]), int(self._row_start[hi]), dtype=np.int32))
            u, r = np.concatenate(utts), np.concatenate(rows)
            contiguous = self.nbl == 1 or (len(u) and u[-1] - u[0] + 1 == len(u))
            steps.append((u, r, bool(contiguous)))
        return steps


class KMeansBatchSweeper(object):
    """One batch-synchronous sweep = score -> segment -> partials -> [ONE all-gather] -> finalize, all enqueued
    on the current stream.  With world == 1 there is no collective; with world > 1 every rank contributes one
    packed record -- the partial sums, totals and counts of its statistics blocks and, per block, the (normally
    empty) list of tokens that found new components -- of segk_kmeans_batch_record_words() 8-byte words
    (nbl * (K_max * (D + 1) + 1 + flag words): 0.8 MB per rank at 8 GPUs on the headline corpus) over RCCL."""

    def __init__(self, dk, part, flag_cap=4096, group=None):
        torch = _torch()
        dev = _dev()
        self.dk, self.part, self.cap, self.comm = dk, part, int(flag_cap), get_comm(group)
        assert self.comm.world == part.world and self.comm.rank == part.rank
        c = dk.corpus
        W = part.world
        # a sharded corpus: the embedding rows of the tokens that found new components travel in the record (the other ranks
        # cannot read them from their own X)
        self.flag_rows = 1 if dk.shard is not None else 0
        self.rank_stride = int(dk._L.segk_kmeans_batch_record_words(
            dk.K_max, c.D, part.nbl, self.cap, c.D * (4 if c.x_dtype == SEGK_F32 else 8) if self.flag_rows else 0))
        self.pack_all = torch.zeros((W, self.rank_stride), dtype=torch.float64, device=dev)
        self.pack = self.pack_all[part.rank]
        self.blk_lo = to_dev(part.local_bounds, np.int32)
        # sort scratch: one region per (local block, component range), addressed by the kernels as sorted + p0 * NR with p0 the
        # block's first GLOBAL token slot (utterance * N_max): only this rank's slots are backed by memory -- the pointer
        # handed to the library is the address slot 0 would have (the kernels touch the regions of the local blocks only)
        ws, wk = C.c_int64(), C.c_int64()
        n_loc = (part.utt_hi - part.utt_lo) * c.N_max
        check(dk._L.segk_kmeans_batch_scratch_words(dk.K_max, n_loc, part.nbl, C.byref(ws), C.byref(wk)))
        self.sorted = torch.zeros(ws.value, dtype=torch.int32, device=dev)
        nr = ws.value // max(n_loc, 1)
        self._sorted_ptr = C.c_void_p(self.sorted.data_ptr() - 4 * part.utt_lo * c.N_max * nr)
        self.koff = torch.zeros(wk.value, dtype=torch.int32, device=dev)
        # SEGK_SWEEP_GRAPH=1: the sweep replayed as a hipGraph from its second run on.  Default 0: measured on MI355X
        # the replay is SLOWER than the plain launches it replaces (full corpus 0.661 vs 0.634 ms per sweep, a
        # 1 250-utterance shard 0.258 vs 0.216 ms: the launches of sweep i + 1 are enqueued while sweep i runs, so
        # their host cost is already hidden, and a replay adds ~10 us of its own -- profiles/README.md r02_b)
        import os
        self.use_graph = os.environ.get("SEGK_SWEEP_GRAPH", "0") == "1"
        self._graph, self._graph_args, self._side, self._warm = None, None, None, 0
        # Hints for the score stage (segk_kmeans_score_hinted): every row's argmax of the previous sweep + the relabelling of that
        # sweep's finalize.  They are passed from the THIRD sweep of a chain on (the first sweep has none; the second would hint
        # with labels found against the initial means -- on a fresh chain more than half of them are wrong, and verifying a wrong
        # hint costs more than not hinting), and left out for one sweep whenever the library reports that the certificate of a
        # recent hinted sweep failed on more than HINT_MISS_PERMILLE of the rows (segk_kmeans_hint_feedback: no synchronisation).
        # Results are identical either way.
        self._sweeps_done = 0
        _l, _s, _p = C.c_uint32(0), C.c_uint32(0), C.c_int32(0)
        check(dk._L.segk_kmeans_hint_feedback(dk._ctx, C.byref(_l), C.byref(_s), C.byref(_p)))
        self._fb_seen = _l.value     # figures of hinted calls enqueued before this sweeper existed are not its business
        self._mb = None              # (n_batches, per-step launch tables) of sweep_minibatch
        dk.batch_comm = self.comm

    HINT_MISS_PERMILLE = 450

    def _use_hints(self):
        if self._sweeps_done < 2:
            return False
        launched, seen, permille = C.c_uint32(0), C.c_uint32(0), C.c_int32(0)
        check(self.dk._L.segk_kmeans_hint_feedback(self.dk._ctx, C.byref(launched), C.byref(seen), C.byref(permille)))
        if seen.value > self._fb_seen:
            self._fb_seen = seen.value
            if permille.value > self.HINT_MISS_PERMILLE:
                return False
        return True

    # ------------------------------------------------------------------ mini-batch sweeps
    def _minibatch_tables(self, n_batches):
        if self._mb is None or self._mb[0] != n_batches:
            steps = []
            for u, r, contiguous in self.part.minibatch(n_batches):
                r = r - self.dk.row_base                 # rows in the device's numbering (a shard starts at 0)
                if contiguous:
                    steps.append((int(u[0]) if len(u) else 0, len(u), int(r[0]) if len(r) else 0, len(r), None, None))
                else:
                    steps.append((0, len(u), 0, len(r), to_dev(u, np.int32), to_dev(r, np.int32)))
            self._mb = (n_batches, steps)
        return self._mb[1]

    def _tokens_from_state(self, boundaries):
        """Token lists of ALL utterances from the boundaries and `assignments` (the state a fresh segmenter or the
        sequential mode leaves): what a mini-batch step needs for the utterances it does not resegment."""
        torch = _torch()
        dk, c = self.dk, self.dk.corpus
        check(dk._L.segk_fbb_collect(dk._ctx, dk._cp(), ptr(boundaries), ptr(dk.new_tok), ptr(dk.n_new), _abi.stream()))
        live = torch.arange(c.N_max, device=dk.n_new.device, dtype=torch.int32)[None, :] < dk.n_new[:, None]
        lab = dk.assignments[dk.new_tok.clamp(min=0).long()]
        dk.new_k.copy_(torch.where(live, lab, torch.full_like(lab, -1)))
        dk.n_flag.zero_()

    def sweep_minibatch(self, boundaries, n_slices_min, n_slices_max, wip, n_batches):
        """One sweep as n_batches steps (specification: oracle/np_oracle.py kmeans_minibatch_sweep): step j scores and
        resegments run j of every local statistics block against the current means, then the statistics are rebuilt from the
        current tokens of ALL utterances exactly as in a whole-sweep batch step (sort, partial sums, ONE all-gather,
        finalize) -- n_batches all-gathers per sweep, n_batches times fresher statistics."""
        dk, pt = self.dk, self.part
        L, ctx, cp, mp, st = dk._L, dk._ctx, dk._cp(), C.byref(dk.m), _abi.stream()
        if dk.assign_stale is None:
            dk.ensure_boundaries()
            self._tokens_from_state(boundaries)
        # (the rows of step j carry the labels of step j of the PREVIOUS sweep, n_batches finalizes old, of which only the last
        # one's relabelling is in dk.remap: while components are still being removed some hints are stale -- they are verified
        # like any other and cost one pass of the band stage, never a wrong result)
        hints = self._use_hints()
        for (utt0, n_utts, row0, n_rows, utts, rows) in self._minibatch_tables(n_batches):
            remap = dk.remap if hints else None
            if rows is None:
                dk.score_rows(row0=row0, n=n_rows, hint_remap=remap)
                dk.segment(boundaries, n_slices_min, n_slices_max, wip, utt0=utt0, n_utts=n_utts)
            else:
                dk.score_rows(ids=rows, hint_remap=remap)
                check(L.segk_kmeans_segment(ctx, cp, mp, ptr(utts), 0, n_utts, int(n_slices_min), int(n_slices_max), float(wip),
                                            C.byref(dk.cand), ptr(boundaries), ptr(dk.old_tok), ptr(dk.new_tok), ptr(dk.new_k),
                                            ptr(dk.n_old), ptr(dk.n_new), ptr(dk.n_flag), ptr(dk.out_total), ptr(dk.status), st))
            check(L.segk_kmeans_batch_partials(ctx, cp, mp, ptr(self.blk_lo), pt.nbl, ptr(dk.new_tok), ptr(dk.new_k),
                                               ptr(dk.n_flag), ptr(dk.out_total), self._sorted_ptr, ptr(self.koff),
                                               ptr(self.pack), self.cap, self.flag_rows, ptr(dk.out_scalars), st))
            if pt.world > 1:
                self.comm.all_gather_rows(self.pack_all, self.pack)
            self._enqueue_back()
        self._sweeps_done += 1
        dk.assign_stale = (pt.utt_lo, pt.utt_hi, pt.world)
        dk.bounds_stale = (boundaries, pt) if pt.world > 1 else None

    def _enqueue_front(self, boundaries, n_slices_min, n_slices_max, wip):
        dk, pt = self.dk, self.part
        L, ctx, cp, mp, st = dk._L, dk._ctx, dk._cp(), C.byref(dk.m), _abi.stream()
        # from the second sweep on cand_k holds every row's argmax of the previous sweep and dk.remap the relabelling of
        # that sweep's finalize: hints for the score stage (same results)
        dk.score_rows(row0=pt.row_lo - dk.row_base, n=pt.row_hi - pt.row_lo, hint_remap=dk.remap if self._use_hints() else None)
        dk.segment(boundaries, n_slices_min, n_slices_max, wip, utt0=pt.utt_lo, n_utts=pt.utt_hi - pt.utt_lo)
        check(L.segk_kmeans_batch_partials(ctx, cp, mp, ptr(self.blk_lo), pt.nbl, ptr(dk.new_tok), ptr(dk.new_k),
                                           ptr(dk.n_flag), ptr(dk.out_total), self._sorted_ptr, ptr(self.koff),
                                           ptr(self.pack), self.cap, self.flag_rows, ptr(dk.out_scalars), st))

    def _enqueue_back(self):
        dk, pt = self.dk, self.part
        L, ctx, cp, mp, st = dk._L, dk._ctx, dk._cp(), C.byref(dk.m), _abi.stream()
        check(L.segk_kmeans_batch_finalize(ctx, cp, mp, pt.utt_lo, pt.utt_hi, ptr(self.pack_all), pt.n_blocks, pt.nbl,
                                           self.rank_stride, self.cap, self.flag_rows, pt.rank, ptr(dk.new_k), ptr(dk.remap),
                                           ptr(dk.out_scalars), ptr(dk.status), st))

    def sweep(self, boundaries, n_slices_min, n_slices_max, wip):
        dk, pt = self.dk, self.part
        if self.use_graph:
            self._sweep_graph(boundaries, n_slices_min, n_slices_max, wip)
        else:
            self._enqueue_front(boundaries, n_slices_min, n_slices_max, wip)
            if pt.world > 1:
                self.comm.all_gather_rows(self.pack_all, self.pack)
            self._enqueue_back()
        dk.assign_stale = (pt.utt_lo, pt.utt_hi, pt.world)
        dk.bounds_stale = (boundaries, pt) if pt.world > 1 else None
        self._sweeps_done += 1

    # ------------------------------------------------------------------ hipGraph replay
    def _capture(self, fn):
        """Run `fn` (library launches only) under stream capture on the sweeper's side stream -> executable graph."""
        dk = self.dk
        st = _abi.stream()
        check(dk._L.segk_graph_begin(dk._ctx, st))
        try:
            fn()
        finally:
            ex = C.c_void_p()
            rc = dk._L.segk_graph_end(dk._ctx, st, C.byref(ex))
        check(rc)
        return ex

    def _sweep_graph(self, boundaries, n_slices_min, n_slices_max, wip):
        """The sweep as one hipGraph (two around the all-gather with more than one rank), replayed on a stream of
        the sweeper's own (a capture cannot run on the legacy default stream); the caller's current stream is
        ordered before and after it.  The first sweep runs eagerly: it creates what a capture cannot contain
        (workspaces, kernel attributes)."""
        torch = _torch()
        dk, pt = self.dk, self.part
        args = (boundaries.data_ptr(), int(n_slices_min), int(n_slices_max), float(wip))
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            if self._warm < 1 or (self._graph is not None and self._graph_args != args):
                self._drop_graph()
            if self._warm < 1:
                self._enqueue_front(boundaries, n_slices_min, n_slices_max, wip)
                if pt.world > 1:
                    self.comm.all_gather_rows(self.pack_all, self.pack)
                self._enqueue_back()
                self._warm += 1
            else:
                if self._graph is None:
                    if pt.world > 1:
                        front = self._capture(lambda: self._enqueue_front(boundaries, n_slices_min, n_slices_max, wip))
                        back = self._capture(self._enqueue_back)
                        self._graph = (front, back)
                    else:
                        def whole():
                            self._enqueue_front(boundaries, n_slices_min, n_slices_max, wip)
                            self._enqueue_back()
                        self._graph = (self._capture(whole),)
                    self._graph_args = args
                st = _abi.stream()
                check(dk._L.segk_graph_launch(dk._ctx, self._graph[0], st))
                if pt.world > 1:
                    self.comm.all_gather_rows(self.pack_all, self.pack)
                    check(dk._L.segk_graph_launch(dk._ctx, self._graph[1], st))
        cur.wait_stream(self._side)

    def _drop_graph(self):
        if self._graph is not None:
            for ex in self._graph:
                self.dk._L.segk_graph_destroy(self.dk._ctx, ex)
        self._graph = None

    def __del__(self):
        try:
            self._drop_graph()
        except Exception:
            pass


class DeviceFbgmm(object):
    """Device image of FBGMM + its Gaussian components (fixed-variance or diagonal), see
    include/segk.h `segk_fbgmm`."""

    def __init__(self, corpus, cov_type, K_max, alpha, lms, prior_a, prior_b, prior_c, k_0, v_0, assignments,
                 lm=None):
        torch = _torch()
        dev = _dev()
        self.corpus = corpus
        c = corpus
        self.K_max = int(K_max)
        self.cov_type = int(cov_type)
        f64 = torch.float64
        self.prior_a = to_dev(prior_a, np.float64)
        self.prior_b = to_dev(prior_b, np.float64)
        self.prior_c = to_dev(prior_c if prior_c is not None else np.zeros(c.D), np.float64)
        self.stat_a = torch.zeros((self.K_max, c.D), dtype=f64, device=dev)
        self.stat_b = torch.zeros((self.K_max, c.D), dtype=f64, device=dev)
        self.log_prod = torch.zeros(self.K_max, dtype=f64, device=dev)
        self.pred = torch.zeros((self.K_max, c.D), dtype=f64, device=dev)
        self.kconst = torch.zeros(self.K_max + 1, dtype=f64, device=dev)
        self.counts = torch.zeros(self.K_max, dtype=torch.int64, device=dev)
        self.assignments = to_dev(assignments, np.int32)
        self.K = torch.zeros(1, dtype=torch.int32, device=dev)
        self.f = _abi.FbgmmDev(
            cov_type=self.cov_type, K_max=self.K_max, alpha=float(alpha), lms=float(lms), k_0=float(k_0),
            v_0=float(v_0), prior_a=self.prior_a.data_ptr(), prior_b=self.prior_b.data_ptr(),
            prior_c=self.prior_c.data_ptr(), stat_a=self.stat_a.data_ptr(), stat_b=self.stat_b.data_ptr(),
            log_prod=self.log_prod.data_ptr(), pred=self.pred.data_ptr(), counts=self.counts.data_ptr(),
            assignments=self.assignments.data_ptr(), K=self.K.data_ptr(), kconst=self.kconst.data_ptr())
        self.lm = lm
        if lm is not None:        # bigram_lms.BigramSmoothLM: its count tensors are tied to the components
            assert lm.K == self.K_max and self.cov_type == 0
            self.f.lm_unigram = lm._unigram.data_ptr()
            self.f.lm_bigram = lm._bigram.data_ptr()
            self.f.lm_lambda, self.f.lm_a, self.f.lm_b = float(lm.intrp_lambda), float(lm.a), float(lm.b)
        self.score = torch.zeros(c.n_emb, dtype=f64, device=dev)
        self.status = torch.zeros(8, dtype=torch.int32, device=dev)
        self._L = _abi.lib()
        self._ctx = _abi.ctx()
        # scratch for single-item calls
        self._tok1 = torch.zeros(max(c.N_max, 1), dtype=torch.int32, device=dev)
        self._n1 = torch.ones(1, dtype=torch.int32, device=dev)
        self._u1 = torch.zeros(1, dtype=f64, device=dev)
        self._cur1 = torch.zeros(1, dtype=torch.int64, device=dev)
        check(self._L.segk_fbgmm_init_stats(self._ctx, self._cp(), C.byref(self.f), _abi.stream()))
        if c.n_utt:
            nu, nm = c.n_utt, c.N_max
            self.new_tok = torch.zeros((nu, nm), dtype=torch.int32, device=dev)
            self.n_new = torch.zeros(nu, dtype=torch.int32, device=dev)
            self.out_logprob = torch.zeros(nu, dtype=f64, device=dev)
            self.ucursor = torch.zeros(1, dtype=torch.int64, device=dev)
            self.ustream = torch.zeros(1, dtype=f64, device=dev)

    def _cp(self):
        return C.byref(self.corpus.c)

    def set_lms_alpha(self, alpha, lms):
        self.f.alpha, self.f.lms = float(alpha), float(lms)

    def update(self, op, utt=0, item=0, k=0, boundaries=None):
        check(self._L.segk_fbgmm_update(self._ctx, self._cp(), C.byref(self.f), int(op), int(utt), int(item),
                                        int(k), ptr(boundaries), _abi.stream()))

    def score_rows(self, ids_ptr=None, row0=0, n=None):
        n = self.corpus.n_emb - row0 if n is None else n
        p = C.c_void_p(ids_ptr) if ids_ptr is not None else None
        check(self._L.segk_fbgmm_score(self._ctx, self._cp(), C.byref(self.f), p, int(row0), int(n),
                                       ptr(self.score), _abi.stream()))

    def log_marg_rows(self, ids):
        """log_marg_i for a list of rows -> numpy float64."""
        ids_t = to_dev(ids, np.int32)
        check(self._L.segk_fbgmm_score(self._ctx, self._cp(), C.byref(self.f), ptr(ids_t), 0, ids_t.numel(),
                                       ptr(self.score), _abi.stream()))
        return self.score[ids_t.long()].cpu().numpy()

    def pred_vector(self, row):
        """(log_post_pred(row)[:K], log_prior(row)) evaluated on the device."""
        torch = _torch()
        out = torch.zeros(self.K_max + 1, dtype=torch.float64, device=self.stat_a.device)
        check(self._L.segk_fbgmm_pred_vector(self._ctx, self._cp(), C.byref(self.f), int(row), ptr(out),
                                             _abi.stream()))
        o = out.cpu().numpy()
        return o[:int(self.K.item())].copy(), float(o[self.K_max])

    def assign_item(self, i, u, anneal_temp=1.0, map_assign=False, j_prev=None):
        """gibbs_sample_inside_loop_i / map_assign_i for one row with the uniform `u`; with an LM
        attached gibbs_sample_inside_loop_i_embed given the previous component `j_prev`.
        Returns the component the row went to."""
        self._tok1[0] = int(i)
        self._u1[0] = float(u)
        self._cur1.zero_()
        check(self._L.segk_fbgmm_assign(self._ctx, self._cp(), C.byref(self.f), 0, 1 if map_assign else 0,
                                        -1 if j_prev is None else int(j_prev), float(anneal_temp),
                                        ptr(self._tok1), ptr(self._n1), ptr(self._u1), ptr(self._cur1), 1,
                                        ptr(self.status), _abi.stream()))
        return int(self.assignments[int(i)].item())

    def gibbs_items(self, uniforms, consider_unassigned=True, anneal_temp=1.0, ids=None):
        """One pass of FBGMM.gibbs_sample's inner loop (fbgmm.py:352-405) over `ids` (default: all
        rows, in index order) consuming `uniforms` in order."""
        torch = _torch()
        u = to_dev(np.asarray(uniforms, dtype=np.float64)) if len(uniforms) else torch.zeros(1, dtype=torch.float64,
                                                                                             device=self.K.device)
        cur = torch.zeros(1, dtype=torch.int64, device=self.K.device)
        ids_t = None if ids is None else to_dev(ids, np.int32)
        n = self.corpus.n_emb if ids is None else ids_t.numel()
        check(self._L.segk_fbgmm_gibbs_items(self._ctx, self._cp(), C.byref(self.f), ptr(ids_t), int(n),
                                             1 if consider_unassigned else 0, float(anneal_temp), ptr(u), ptr(cur),
                                             len(uniforms), ptr(self.status), _abi.stream()))
        return int(cur.item())

    def set_uniform_stream(self, u):
        self.ustream = to_dev(u, np.float64)
        self.ucursor.zero_()

    def gibbs_utt(self, boundaries, i, viterbi, n_slices_min, n_slices_max, wip, time_power_term,
                  log_p_continue, anneal_temp_fb, anneal_temp_am, assignments_only=False, map_assign=None):
        """gibbs_sample_i (unigram_acoustic_wordseg.py:252-360; with an LM attached
        bigram_acoustic_wordseg.py:386-551) for utterance i, enqueued asynchronously: [remove its
        LM counts,] remove its segments, score its spans, sample boundaries, assign[, add its LM
        counts]."""
        c = self.corpus
        L, ctx, cp, fp, st = self._L, self._ctx, self._cp(), C.byref(self.f), _abi.stream()
        N = int(c.lengths_np[i])
        tri_i = N * (N + 1) // 2
        if map_assign is None:
            map_assign = viterbi
        if self.lm is not None:
            check(L.segk_fbgmm_update(ctx, cp, fp, 5, int(i), 0, 0, ptr(boundaries), st))
        check(L.segk_fbgmm_update(ctx, cp, fp, 0, int(i), 0, 0, ptr(boundaries), st))
        if not assignments_only:
            check(L.segk_fbgmm_score(ctx, cp, fp, C.c_void_p(c.vec_ids.data_ptr() + 4 * i * c.tri), 0, tri_i,
                                     ptr(self.score), st))
        mode = 2 if assignments_only else (1 if viterbi else 0)
        check(L.segk_unigram_segment(ctx, cp, int(i), mode, int(n_slices_min), int(n_slices_max),
                                     float(wip), float(time_power_term), float(log_p_continue),
                                     float(anneal_temp_fb), ptr(self.score), ptr(self.ustream), ptr(self.ucursor),
                                     self.ustream.numel(), ptr(boundaries), ptr(self.new_tok), ptr(self.n_new),
                                     ptr(self.out_logprob), ptr(self.status), st))
        check(L.segk_fbgmm_assign(ctx, cp, fp, int(i), 1 if map_assign else 0, -1, float(anneal_temp_am),
                                  ptr(self.new_tok), ptr(self.n_new), ptr(self.ustream), ptr(self.ucursor),
                                  self.ustream.numel(), ptr(self.status), st))
        if self.lm is not None:
            check(L.segk_fbgmm_update(ctx, cp, fp, 6, int(i), 0, 0, ptr(boundaries), st))

    def sequential_sweep(self, boundaries, order, row_start, viterbi, n_slices_min, n_slices_max, wip, time_power_term,
                         log_p_continue, anneal_temp_fb, anneal_temp_am, map_assign=None):
        """gibbs_utt for every utterance of `order` in turn by ONE library call (segk_fbgmm_sequential_sweep: a persistent
        kernel per stretch of utterances between two emptied components; the call synchronises the stream; with a language
        model attached the bigram sampler's gibbs_sample_i, bigram_acoustic_wordseg.py:386-551).  Returns False, with nothing
        enqueued, where the kernel does not apply (a model too large for a workgroup's LDS, ...): the caller then walks the
        utterances itself."""
        if self.corpus.N_max > 64:
            return False
        if getattr(self, "_row_start_dev", None) is None:
            self._row_start_dev = to_dev(np.asarray(row_start, dtype=np.int32))
        arr = np.ascontiguousarray(order, dtype=np.int32)
        if map_assign is None:
            map_assign = viterbi
        rc = self._L.segk_fbgmm_sequential_sweep(
            self._ctx, self._cp(), C.byref(self.f), arr.ctypes.data, len(arr), ptr(self._row_start_dev), 1 if viterbi else 0,
            1 if map_assign else 0, int(n_slices_min), int(n_slices_max), float(wip), float(time_power_term),
            float(log_p_continue), float(anneal_temp_fb), float(anneal_temp_am), ptr(self.score), ptr(self.ustream),
            ptr(self.ucursor), self.ustream.numel(), ptr(boundaries), ptr(self.new_tok), ptr(self.n_new),
            ptr(self.out_logprob), ptr(self.status), _abi.stream())
        if rc == _abi.SEGK_ERR_UNSUPPORTED:
            return False
        check(rc)
        return True

    def record_metrics(self, urn=False, urn_a=0.0):
        """(log_prob_z, log_prob_X_given_z, K, n_assigned) of the sequential-mode state, computed on the device (segk_fbgmm_record_metrics): fbgmm.py:208-225 or, with urn=True,
        bigram_acoustic_wordseg.py:287-305; gaussian_components_{fixedvar,diag}.py log_marg."""
        torch = _torch()
        if getattr(self, "_metric_out", None) is None:
            self._metric_out = torch.zeros(4, dtype=torch.float64, device=self.K.device)
        check(self._L.segk_fbgmm_record_metrics(self._ctx, self._cp(), C.byref(self.f), 1 if urn else 0, float(urn_a),
                                                ptr(self._metric_out), _abi.stream()))
        o = self._metric_out.cpu().numpy()
        return float(o[0]), float(o[1]), int(o[2]), int(o[3])

    def check_status(self):
        st = int(self.status[0].item())
        if st & 8:
            raise SegkError("uniform stream exhausted")
        if st & 16:
            raise AssertionError("forward_backward: log_prob == -inf (unigram_acoustic_wordseg.py:753)")


class FbgmmBatchSweeper(object):
    """Batch-synchronous ("blocked parallel Gibbs") sweeps of the FBGMM / bigram samplers --
    specification oracle/np_fbgmm_batch.py, C ABI `segk_fbb_*` (include/segk.h).

    The utterances are cut into `n_stat_blocks` slices (rank r owns a contiguous run of them) x
    `n_gibbs_blocks` blocks.  Per Gibbs step b every rank resamples block b of its slices against
    the statistics of all other blocks, recomputes the block's partial sums and all-gathers them
    (one RCCL all-gather of S/P * K_max*(2D+1) doubles per rank and step; with a language model a
    second one of the block's transcripts).  All partial sums are replicated, so every rank derives
    the same statistics in the same fixed order: results do not depend on the number of ranks."""

    def __init__(self, df, row_start, n_gibbs_blocks=8, n_stat_blocks=8, seed=0, group=None, score_precision="f64"):
        torch = _torch()
        dev = _dev()
        self.df, self.comm = df, get_comm(group)
        assert score_precision in ("f64", "f32", "f16")
        if score_precision == "f16" and df.cov_type != 0:
            raise SegkError("score_precision='f16' (matrix-core span score) exists for fixed-variance components only")
        if score_precision == "f16" and 2 * df.corpus.D > 208:
            raise SegkError("score_precision='f16' supports D <= 104")
        # fixed variance: the span score as a matrix-core contraction (f32 / f16); diagonal: the Student-t terms in
        # float32 with the hardware logarithm (f32)
        self.score_f32 = score_precision in ("f32", "f16") and df.cov_type == 0
        self.score_f16 = score_precision == "f16"
        self.score_diag32 = score_precision == "f32" and df.cov_type == 1
        self._fused = None             # segk_fbb_step_diag32: None = not tried yet, False = the library refused (unsupported shape)
        c = df.corpus
        self.S, self.B = int(n_stat_blocks), int(n_gibbs_blocks)
        rank, world = self.comm.rank, self.comm.world
        if self.S % world != 0:
            raise SegkError("the number of statistics slices (%d) must be a multiple of the number of ranks (%d)"
                            % (self.S, world))
        self.rank, self.world = rank, world
        self.s_n = self.S // world
        self.s_lo = rank * self.s_n
        n_utt = c.n_utt
        sb = [(s * n_utt) // self.S for s in range(self.S + 1)]
        ur = np.zeros((self.S, self.B, 2), np.int32)
        rr = np.zeros((self.S, self.B, 2), np.int32)
        row_start = np.asarray(row_start, dtype=np.int64)
        for s in range(self.S):
            n_s = sb[s + 1] - sb[s]
            for b in range(self.B):
                lo, hi = sb[s] + (b * n_s) // self.B, sb[s] + ((b + 1) * n_s) // self.B
                ur[s, b] = (lo, hi)
                rr[s, b] = (row_start[lo], row_start[hi])
        self.utt_range_np, self.row_range_np = ur, rr
        self.utt_range, self.row_range = to_dev(ur), to_dev(rr)
        K, D = df.K_max, c.D
        self.rec = K * (2 * D + 1)
        f64 = torch.float64
        self.partials = torch.zeros((self.B, self.S, self.rec), dtype=f64, device=dev)
        self.cnt = torch.zeros(K, dtype=f64, device=dev)
        self.mean_t = torch.zeros((D, K), dtype=f64, device=dev)
        self.q_t = torch.zeros((D, K), dtype=f64, device=dev)
        self.lconst = torch.zeros(K, dtype=f64, device=dev)
        self.zconst = torch.zeros(K, dtype=f64, device=dev)
        self.half = torch.zeros(K, dtype=f64, device=dev)
        self.scal = torch.zeros(2, dtype=f64, device=dev)
        self.slot = torch.zeros(c.n_emb, dtype=torch.int32, device=dev)
        self.remap = torch.zeros(K, dtype=torch.int32, device=dev)
        self.u_max = int((ur[:, :, 1] - ur[:, :, 0]).max())
        self.lm_tok = None
        # the batch state keeps its own bigram table (indexed by slots); the sequential-mode tables of
        # the LM object are only written by materialise()
        self.f = _abi.FbgmmDev.from_buffer_copy(df.f)
        if df.lm is not None:
            self.lm_tok = torch.full((self.B, self.S, self.u_max, c.N_max), -1, dtype=torch.int32, device=dev)
            self.lm_big = torch.zeros((K, K), dtype=torch.int64, device=dev)
            self.f.lm_bigram = self.lm_big.data_ptr()
        self.y = self.tiles32 = None
        ldy = (2 * D + 3) // 4 * 4
        if self.score_f32:
            self.y = torch.zeros((c.n_emb, ldy), dtype=torch.float32, device=dev)
            self.tiles32 = torch.zeros(int(_abi.lib().segk_kmeans_tiles_floats(K + 1, 2 * D)), dtype=torch.float32,
                                       device=dev)
        self.y16 = self.tiles16 = self.rows32 = self.consts16 = None
        if self.score_f16:
            L = _abi.lib()
            self.y16 = torch.zeros(int(L.segk_corpus_b3_bytes(c.n_emb, 2 * D)), dtype=torch.uint8, device=dev)
            self.tiles16 = torch.zeros(int(L.segk_kmeans_tiles_b3_floats(K + 1, 2 * D)), dtype=torch.float32, device=dev)
            self.rows32 = torch.zeros((K + 1, 2 * D), dtype=torch.float32, device=dev)
            self.consts16 = torch.zeros(2 * (K + 2) + 32, dtype=f64, device=dev)
        # the prior predictive of every row (an empty slot's likelihood): a constant of corpus and prior, evaluated once
        # instead of in the score and assignment kernels of every Gibbs step (same values)
        self.prior_rows = torch.zeros(c.n_emb, dtype=f64, device=dev)
        check(df._L.segk_fbb_prior_rows(df._ctx, df._cp(), C.byref(self.f), self.prior_rows.data_ptr(), _abi.stream()))
        self.bt = _abi.FbatchDev(
            prior_rows=self.prior_rows.data_ptr(),
            y16=self.y16.data_ptr() if self.y16 is not None else None,
            tiles16=self.tiles16.data_ptr() if self.tiles16 is not None else None,
            rows32=self.rows32.data_ptr() if self.rows32 is not None else None,
            consts16=self.consts16.data_ptr() if self.consts16 is not None else None,
            y=self.y.data_ptr() if self.y is not None else None, ldy=ldy,
            tiles32=self.tiles32.data_ptr() if self.tiles32 is not None else None,
            n_slices=self.S, n_blocks=self.B, u_max=self.u_max if df.lm is not None else 0, fast_dp=1 if score_precision != "f64" else 0,
            utt_range=self.utt_range.data_ptr(), row_range=self.row_range.data_ptr(),
            partials=self.partials.data_ptr(), cnt=self.cnt.data_ptr(), mean_t=self.mean_t.data_ptr(),
            q_t=self.q_t.data_ptr(), lconst=self.lconst.data_ptr(), zconst=self.zconst.data_ptr(),
            half=self.half.data_ptr(), scal=self.scal.data_ptr(), slot=self.slot.data_ptr(),
            lm_tok=self.lm_tok.data_ptr() if self.lm_tok is not None else None, seed=int(seed) & (2 ** 64 - 1))
        # host-side per-step launch tables
        I32 = C.c_int32 * self.s_n
        self._n_rows = [I32(*[int(rr[self.s_lo + i, b, 1] - rr[self.s_lo + i, b, 0]) for i in range(self.s_n)])
                        for b in range(self.B)]
        self._n_utts = [I32(*[int(ur[self.s_lo + i, b, 1] - ur[self.s_lo + i, b, 0]) for i in range(self.s_n)])
                        for b in range(self.B)]
        self.ll_mat = None
        if self.score_f16:
            # token likelihoods of the assignment step from the matrix cores: per block, the positions of its
            # utterances' token slots in the flat new_tok array (slice order) and the matrix they fill
            nm = c.N_max
            self._tok_map = [to_dev(np.concatenate([np.arange(ur[self.s_lo + i, b, 0] * nm, ur[self.s_lo + i, b, 1] * nm)
                                                    for i in range(self.s_n)]).astype(np.int64))
                             for b in range(self.B)]
            self._tok_rows = torch.zeros(max(t.numel() for t in self._tok_map), dtype=torch.int32, device=dev)
            self.ll_ld = (K + 1 + 31) // 32 * 32
            self.ll_mat = torch.zeros((self._tok_rows.numel(), self.ll_ld), dtype=torch.float32, device=dev)
        if self.score_f32:
            # the rows of block b over the local slices, one launch per step
            self._block_rows = [to_dev(np.concatenate([np.arange(rr[self.s_lo + i, b, 0], rr[self.s_lo + i, b, 1])
                                                       for i in range(self.s_n)]).astype(np.int32))
                                for b in range(self.B)]
            check(df._L.segk_fbb_make_y(df._ctx, df._cp(), C.byref(self.bt), _abi.stream()))
        self.in_batch_state = False
        self.sweep_index = 0

    # ------------------------------------------------------------------ plumbing
    def _args(self):
        df = self.df
        self.f.alpha, self.f.lms = df.f.alpha, df.f.lms
        return df._L, df._ctx, df._cp(), C.byref(self.f), C.byref(self.bt), _abi.stream()

    def _gather(self, full, b):
        """in-place all-gather of full[b] ([S, ...]): every rank contributes its slices."""
        if self.world == 1:
            return
        out = full[b].view(self.world, -1)
        self.comm.all_gather_rows(out, out[self.rank])

    def enter(self, boundaries):
        """Build the batch state (slots, token lists, all partial sums, transcripts) from the
        sequential-mode state of the components."""
        df = self.df
        L, ctx, cp, fp, bp, st = self._args()
        self.slot.copy_(df.assignments)
        if df.lm is not None:
            self.lm_big.copy_(df.lm._bigram)
        check(L.segk_fbb_collect(ctx, cp, ptr(boundaries), ptr(df.new_tok), ptr(df.n_new), st))
        for b in range(self.B):
            check(L.segk_fbb_partials(ctx, cp, fp, bp, self.s_lo, self.s_n, b, ptr(df.new_tok), ptr(df.n_new), st))
            self._gather(self.partials, b)
            if self.lm_tok is not None:
                check(L.segk_fbb_lm_fill(ctx, cp, fp, bp, self.s_lo, self.s_n, b, self._n_utts[b], ptr(df.new_tok),
                                         ptr(df.n_new), st))
                self._gather(self.lm_tok, b)
        self.in_batch_state = True

    def rebuild_from_slots(self, boundaries):
        """Partial sums and transcripts from the current `slot` labels (checkpoint resume)."""
        df = self.df
        L, ctx, cp, fp, bp, st = self._args()
        check(L.segk_fbb_collect(ctx, cp, ptr(boundaries), ptr(df.new_tok), ptr(df.n_new), st))
        for b in range(self.B):
            check(L.segk_fbb_partials(ctx, cp, fp, bp, self.s_lo, self.s_n, b, ptr(df.new_tok), ptr(df.n_new), st))
            self._gather(self.partials, b)
            if self.lm_tok is not None:
                check(L.segk_fbb_lm_fill(ctx, cp, fp, bp, self.s_lo, self.s_n, b, self._n_utts[b], ptr(df.new_tok),
                                         ptr(df.n_new), st))
                self._gather(self.lm_tok, b)
        self.in_batch_state = True

    def sweep(self, boundaries, n_slices_min, n_slices_max, wip, time_power_term, anneal_temp_fb=1.0,
              anneal_temp_am=1.0):
        """One sweep = n_gibbs_blocks steps, all enqueued on the current stream."""
        torch = _torch()
        df = self.df
        if not self.in_batch_state:
            self.enter(boundaries)
        L, ctx, cp, fp, bp, st = self._args()
        sw = self.sweep_index
        for b in range(self.B):
            if self.lm_tok is not None:
                check(L.segk_fbb_lm_apply(ctx, cp, fp, bp, b, -1, st))
            check(L.segk_fbb_prepare(ctx, cp, fp, bp, b, st))
            if self.score_diag32 and self.lm_tok is None and self._fused is not False:
                # span scores, boundaries and slots of the block by ONE launch where the library can (segk_fbb_step_diag32)
                rc = L.segk_fbb_step_diag32(ctx, cp, fp, bp, self.s_lo, self.s_n, b, self._n_utts[b], sw, int(n_slices_min),
                                            int(n_slices_max), float(wip), float(time_power_term), float(anneal_temp_fb),
                                            float(anneal_temp_am), ptr(df.score), ptr(boundaries), ptr(df.new_tok),
                                            ptr(df.n_new), ptr(df.out_logprob), ptr(df.status), st)
                if rc == _abi.SEGK_ERR_UNSUPPORTED:
                    self._fused = False
                else:
                    check(rc)
                    self._fused = True
                    check(L.segk_fbb_partials(ctx, cp, fp, bp, self.s_lo, self.s_n, b, ptr(df.new_tok), ptr(df.n_new), st))
                    self._gather(self.partials, b)
                    continue
            if self.score_f32:
                check(L.segk_fbb_score_f32(ctx, cp, fp, bp, ptr(self._block_rows[b]), self._block_rows[b].numel(),
                                           ptr(df.score), st))
            elif self.score_diag32:
                check(L.segk_fbb_score_diag32(ctx, cp, fp, bp, self.s_lo, self.s_n, b, self._n_rows[b], ptr(df.score), st))
            else:
                check(L.segk_fbb_score(ctx, cp, fp, bp, self.s_lo, self.s_n, b, self._n_rows[b], ptr(df.score), st))
            check(L.segk_fbb_segment(ctx, cp, fp, bp, self.s_lo, self.s_n, b, self._n_utts[b], sw, int(n_slices_min),
                                     int(n_slices_max), float(wip), float(time_power_term), float(anneal_temp_fb),
                                     ptr(df.score), ptr(boundaries), ptr(df.new_tok), ptr(df.n_new),
                                     ptr(df.out_logprob), ptr(df.status), st))
            if self.ll_mat is not None:
                tm = self._tok_map[b]
                rows = self._tok_rows[:tm.numel()]
                torch.index_select(df.new_tok.view(-1), 0, tm, out=rows)
                check(L.segk_fbb_token_scores(ctx, cp, fp, bp, ptr(rows), tm.numel(), ptr(self.ll_mat), self.ll_ld, st))
                check(L.segk_fbb_assign(ctx, cp, fp, bp, self.s_lo, self.s_n, b, self._n_utts[b], sw, float(anneal_temp_am),
                                        ptr(df.new_tok), ptr(df.n_new), ptr(self.ll_mat), self.ll_ld, st))
            elif self.score_diag32:
                check(L.segk_fbb_assign_diag32(ctx, cp, fp, bp, self.s_lo, self.s_n, b, self._n_utts[b], sw,
                                               float(anneal_temp_am), ptr(df.new_tok), ptr(df.n_new), st))
            else:
                check(L.segk_fbb_assign(ctx, cp, fp, bp, self.s_lo, self.s_n, b, self._n_utts[b], sw, float(anneal_temp_am),
                                        ptr(df.new_tok), ptr(df.n_new), None, 0, st))
            if self.lm_tok is not None:
                check(L.segk_fbb_lm_fill(ctx, cp, fp, bp, self.s_lo, self.s_n, b, self._n_utts[b], ptr(df.new_tok),
                                         ptr(df.n_new), st))
                self._gather(self.lm_tok, b)
                check(L.segk_fbb_lm_apply(ctx, cp, fp, bp, b, 1, st))
            check(L.segk_fbb_partials(ctx, cp, fp, bp, self.s_lo, self.s_n, b, ptr(df.new_tok), ptr(df.n_new), st))
            self._gather(self.partials, b)
        self.sweep_index += 1

    def utt_values(self, t):
        """numpy copy of a per-utterance device vector with every rank's own utterances filled in."""
        v = t.cpu().numpy().copy()
        if self.world > 1:
            lo = int(self.utt_range_np[self.s_lo, 0, 0])
            hi = int(self.utt_range_np[self.s_lo + self.s_n - 1, -1, 1])
            for plo, phi, x in self.comm.all_gather_object((lo, hi, v[lo:hi])):
                v[plo:phi] = x
        return v

    def totals(self):
        """(counts per slot as float64 numpy, total, occupied) of the current state."""
        L, ctx, cp, fp, bp, st = self._args()
        check(L.segk_fbb_prepare(ctx, cp, fp, bp, -1, st))
        sc = self.scal.cpu().numpy()
        return self.cnt.cpu().numpy(), float(sc[0]), int(sc[1])

    def invalidate(self):
        """The sequential-mode state was mutated: rebuild the batch state before the next sweep."""
        self.in_batch_state = False

    def materialise(self, boundaries=None):
        """The reference's view of the current batch state, without touching it: contiguous
        component labels in `assignments` / K, the LM tables of the LM object, and the
        sequential-mode statistics rebuilt from the assignments (Components.__init__ order)."""
        if not self.in_batch_state:
            return
        torch = _torch()
        df = self.df
        L, ctx, cp, fp, bp, st = self._args()
        check(L.segk_fbb_prepare(ctx, cp, fp, bp, -1, st))
        check(L.segk_fbb_canonical(ctx, cp, C.byref(df.f), bp, ptr(self.remap), st))
        if self.world > 1:
            # every rank holds the slots of its own rows only
            lo, hi = int(self.row_range_np[self.s_lo, 0, 0]), int(self.row_range_np[self.s_lo + self.s_n - 1, -1, 1])
            for plo, phi, t in self.comm.all_gather_object((lo, hi, df.assignments[lo:hi].cpu())):
                df.assignments[plo:phi] = t.to(df.assignments.device)
            if boundaries is not None:          # the boundaries of the other ranks' utterances
                ulo = int(self.utt_range_np[self.s_lo, 0, 0])
                uhi = int(self.utt_range_np[self.s_lo + self.s_n - 1, -1, 1])
                for plo, phi, t in self.comm.all_gather_object((ulo, uhi, boundaries[ulo:uhi].cpu())):
                    boundaries[plo:phi] = t.to(boundaries.device)
        if df.lm is not None:
            occ = torch.nonzero(self.remap >= 0).flatten()
            K = occ.numel()
            big = df.lm._bigram
            big.zero_()
            big[:K, :K] = self.lm_big[occ][:, occ]
            df.lm._unigram.zero_()
            df.lm._unigram[:K] = self.cnt[occ].round().long()
        check(L.segk_fbgmm_init_stats(ctx, cp, C.byref(df.f), st))
