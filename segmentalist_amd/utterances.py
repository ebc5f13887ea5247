"""
Drop-in for segmentalist/utterances.py: the segmentation state of a corpus.

Host-side container (A13): construction and the RNG it consumes follow the reference
(utterances.py:74-157) so that seeded runs start from the same boundaries; the arrays it
holds (`vec_ids`, `durations`, `boundaries`, `lengths`) are what the HIP kernels read and
write.  When a device-backed segmenter owns this object, `boundaries` is mirrored lazily
from HBM (see `bind_device`).
"""
import logging

import numpy as np

logger = logging.getLogger(__name__)


class Utterances(object):
    def __init__(self, lengths, vec_ids, durations, landmarks, seed_boundaries=None,
                 p_boundary_init=0.5, n_slices_min=0, n_slices_max=6, min_duration=0):
        assert lengths == [len(i) for i in landmarks]
        self.lengths = lengths
        self.D = len(lengths)
        assert self.D == len(vec_ids)
        self.N_max = max(lengths)
        self.landmarks = landmarks
        tri = self.N_max * (self.N_max + 1) // 2

        # vec_ids / durations, padded to the longest utterance (utterances.py:91-102).  Uniform corpora (every
        # utterance the same number of landmarks, no duration floor) are stacked in one numpy call instead of the
        # reference's loop over utterances
        uniform = (min_duration == 0 and self.D > 0 and all(len(v) == tri for v in vec_ids)
                   and all(len(d) == tri for d in durations))
        if uniform:
            self.vec_ids = np.stack([np.asarray(v) for v in vec_ids]).astype(np.int64)
            self.durations = np.stack([np.asarray(d, dtype=np.float64) for d in durations])
        else:
            self.vec_ids = np.full((self.D, tri), -1, dtype=np.int64)
            self.durations = np.full((self.D, tri), np.nan, dtype=np.float64)
        for i in range(0 if uniform else self.D):
            v = np.asarray(vec_ids[i])
            self.vec_ids[i, :len(v)] = v
            dv = durations[i]
            if min_duration != 0 and len(dv) != 1:
                masked = np.array(dv, dtype=np.float64)
                masked[masked < min_duration] = np.nan
                if np.all(np.isnan(masked)):
                    masked[np.argmax(dv)] = np.max(dv)
                dv = masked
            self.durations[i, :len(dv)] = dv

        self._boundaries = np.zeros((self.D, self.N_max), dtype=bool)
        self._dev_boundaries = None
        self._host_stale = False
        b = self._boundaries
        if seed_boundaries is not None:
            # nearest-landmark snapping (utterances.py:106-115); no RNG
            for i_utt, bounds in enumerate(seed_boundaries):
                lm = np.asarray(landmarks[i_utt])
                for bound in bounds:
                    b[i_utt, int(np.argmin(np.abs(bound - lm)))] = True
        elif p_boundary_init == 0:
            for i in range(self.D):
                b[i, self.lengths[i] - 1] = True
        else:
            # rejection sampling of a valid random segmentation (utterances.py:141-157):
            # one np.random.rand(N) per attempt
            for i in range(self.D):
                N = self.lengths[i]
                while True:
                    b[i, :N] = np.random.rand(N) < p_boundary_init
                    b[i, N - 1] = True
                    if all(e == -1 for e in self.get_segmented_embeds_i(i)):
                        continue
                    spans = [e - s for s, e in self.get_segmented_landmark_indices(i)]
                    if (max(spans) <= n_slices_max and min(spans) >= n_slices_min) or N <= n_slices_min:
                        break

    def band_tables(self, W):
        """Banded image of `vec_ids` / `durations` for a window of W slices: entry [i, t - 1, w] is the
        span [t - 1 - w, t) of utterance i, i.e. the triangular entry t(t-1)/2 + (t-1-w); -1 / NaN where the span does not
        exist.  The per-utterance kernels only ever read this band; built here in one vectorised gather."""
        N = self.N_max
        t = np.arange(1, N + 1)[:, None]
        w = np.arange(W)[None, :]
        s = t - 1 - w
        ok = s >= 0
        j = np.where(ok, t * (t - 1) // 2 + s, 0)
        ids = np.where(ok[None], self.vec_ids[:, j], -1).astype(np.int32)
        dur = np.where(ok[None], self.durations[:, j], np.nan)
        return np.ascontiguousarray(ids), np.ascontiguousarray(dur)

    def complete_band_tables(self, W):
        """`band_tables(W)` when the band holds every embedding of the corpus -- no triangular entry outside it names one --
        else None.  The FBGMM / bigram kernels read nothing but the band then (segk.h: segk_corpus.band_ids); with
        embeddings outside the window they keep to the triangle."""
        if not (1 <= W < self.N_max):
            return None
        ids, dur = self.band_tables(W)
        if int(np.count_nonzero(ids >= 0)) != int(np.count_nonzero(np.asarray(self.vec_ids) >= 0)):
            return None
        return ids, dur

    # ---------------------------------------------------------------- device mirroring
    def bind_device(self, dev_boundaries, refresh=None):
        """`dev_boundaries`: torch uint8 [D, N_max] owned by the segmenter.  `refresh`: called before the device
        buffer is read back (multi-rank batch mode: fetches the rows other ranks own -- a collective)."""
        self._dev_boundaries = dev_boundaries
        self._dev_refresh = refresh

    def mark_device_dirty(self):
        self._host_stale = True

    @property
    def boundaries(self):
        if self._host_stale and self._dev_boundaries is not None:
            if getattr(self, "_dev_refresh", None) is not None:
                self._dev_refresh()
            self._boundaries = self._dev_boundaries.cpu().numpy().astype(bool)
            self._host_stale = False
        return self._boundaries

    @boundaries.setter
    def boundaries(self, value):
        self._boundaries = np.asarray(value, dtype=bool)
        self._host_stale = False
        if self._dev_boundaries is not None:
            import torch
            self._dev_boundaries.copy_(torch.from_numpy(self._boundaries.astype(np.uint8)))

    def push_boundaries(self):
        """Host -> device after in-place edits of `boundaries`."""
        self.boundaries = self._boundaries

    # ---------------------------------------------------------------- queries (utterances.py:159-232)
    def _segments(self, i):
        b = self.boundaries
        start = 0
        for j in range(self.lengths[i]):
            if b[i, j]:
                yield start, j + 1
                start = j + 1

    def get_segmented_embeds_i(self, i):
        return [self.vec_ids[i, e * (e - 1) // 2 + s] for s, e in self._segments(i)]

    def get_segmented_durations_i(self, i):
        return [self.durations[i, e * (e - 1) // 2 + s] for s, e in self._segments(i)]

    def get_original_segmented_embeds_i(self, i):
        v = self.vec_ids[i]
        return list(np.asarray(self.get_segmented_embeds_i(i)) - np.min(v[v != -1]))

    def get_segmented_landmark_indices(self, i):
        return list(self._segments(i))

    def get_segmented_landmarks(self, i):
        assert self.landmarks is not None
        out, prev = [], 0
        for _, e in self._segments(i):
            out.append((prev, self.landmarks[i][e - 1]))
            prev = self.landmarks[i][e - 1]
        return out


def process_embeddings(embedding_mats, vec_ids_dict):
    """
    Stack the per-utterance embedding matrices (utterances in sorted-key order) and remap the
    utterance-local vec_ids to global rows (unigram_acoustic_wordseg.py:571-646).  Returns
    (embeddings, list of vec_ids, utterance labels); additionally exposes the first global row
    of every utterance via the `row_start` attribute of the returned list (length D+1).
    """
    labels = sorted(embedding_mats)
    mats = [np.asarray(embedding_mats[u]) for u in labels]
    starts = np.concatenate([[0], np.cumsum([len(m) for m in mats])]).astype(np.int64)
    vec_ids = _VecIdList()
    srcs = [np.asarray(vec_ids_dict[u]) for u in labels]
    if srcs and all(v.shape == srcs[0].shape for v in srcs):
        # every utterance has the same span table shape: one stacked remap instead of the loop
        V = np.stack(srcs)
        lens = np.asarray([len(m) for m in mats], dtype=V.dtype)[:, None]
        ok = (V >= 0) & (V < lens)
        V = np.where(ok, V + starts[:-1, None].astype(V.dtype), V)
        vec_ids.extend(list(V))
    else:
        for n, src in enumerate(srcs):
            cur = src.copy()
            ok = (src >= 0) & (src < len(mats[n]))
            cur[ok] = src[ok] + starts[n]
            vec_ids.append(cur)
    vec_ids.row_start = starts
    if len(mats) and all(len(m) for m in mats):
        embeddings = np.concatenate(mats, axis=0)
    else:
        embeddings = np.asarray([row for m in mats for row in m])
    return embeddings, vec_ids, labels


class _VecIdList(list):
    row_start = None
