"""
np_fbgmm_batch.py -- executable SPECIFICATION of the batch-synchronous ("blocked parallel Gibbs")
sweep of the FBGMM / bigram word-segmentation samplers.

TEST INFRASTRUCTURE ONLY (same rules as np_oracle.py): imported by tests/ and tools/, never by
the product package.

The reference (unigram_acoustic_wordseg.py:252-472, bigram_acoustic_wordseg.py:386-671) is a
strictly serial Markov chain: utterance by utterance, and inside an utterance segment by
segment, every draw conditions on the statistics left by the previous one.  It has no parallel
mode, so there is no reference behaviour to match here; this module DEFINES the parallel sampler
the device implements, in terms of the reference's own building blocks (`logsumexp`, `draw`,
`forward_backward`, the predictive densities of gaussian_components_{fixedvar,diag}.py), and the
device path is tested for parity against it.

Definition.  The utterances are cut into `n_stat_blocks` contiguous *slices* (the unit of GPU
ownership and of the fixed summation order) and every slice into `n_gibbs_blocks` contiguous
*blocks*.  A sweep is `n_gibbs_blocks` steps; step b resamples, in parallel and independently of
each other, all utterances of block b of every slice, conditioned on the segments currently held
by all OTHER blocks:

  1. statistics without block b: per component slot counts, sum x, sum x^2 of the tokens of all
     other blocks -- per (slice, block) partial sums accumulated sequentially in token order,
     blocks b' != b added in increasing b', slices combined by a fixed balanced tree;
  2. every candidate span of the block's utterances is scored: log_marg_i (fbgmm.py:256-285) under
     those statistics; a slot with count 0 is an empty component (prior predictive);
  3. boundaries: forward filtering / backward sampling (unigram...:653-756) with the uniforms
     u01(seed, sweep, utterance, step);
  4. every new segment draws its slot from softmax(logits) (fbgmm.py:436-457) by inverse CDF
     (`draw_chunked`, a two-level walk of utils.draw's cumulative sum) with
     u01(seed, sweep, utterance, N_max + position) -- all segments of the block against the same
     statistics of step 1 (the serial chain would see the utterance's earlier segments);
     with a language model the prior of segment t is lm.prob_vec_given_j(slot of segment t-1),
     the LM counts being those of all other blocks;
  5. the block's partial sums are replaced by those of its new tokens.

Slots are not renumbered during a sweep: the finite mixture is symmetric in its K_max slots, an
empty slot is just a slot whose count is 0 (the reference keeps the occupied ones contiguous as a
bookkeeping device, `del_component`).  `canonical()` gives the reference's view (occupied slots
relabelled 0..K-1 in increasing slot order).

With one utterance per block the statistics seen by an utterance are exactly those of the serial
chain (everything but itself); the remaining difference to the reference is step 4.
Results do not depend on how slices are distributed over GPUs.
"""
import math

import numpy as np
from scipy.special import gammaln, logsumexp as _sp_logsumexp

from . import np_oracle as no

_M = (1 << 64) - 1


def u01(seed, sweep, utt, j):
    """Counter-based uniform in [0, 1): two rounds of the splitmix64 finaliser over a linear
    combination of the counters.  Same integer arithmetic on the device (segk_u01)."""
    z = (seed * 0x9E3779B97F4A7C15 + sweep * 0xBF58476D1CE4E5B9 + utt * 0x94D049BB133111EB
         + j * 0xD6E8FEB86659FD93 + 0x2545F4914F6CDD1D) & _M
    for _ in range(2):
        z ^= z >> 30
        z = (z * 0xBF58476D1CE4E5B9) & _M
        z ^= z >> 27
        z = (z * 0x94D049BB133111EB) & _M
        z ^= z >> 31
    return (z >> 11) * (1.0 / 9007199254740992.0)


def draw_chunked(p, u):
    """Inverse-CDF draw of the batch sampler: same distribution as utils.draw (utils.py:10-21) with
    a two-level walk -- the n probabilities are cut into 64 runs of ceil(n/64) consecutive entries,
    run sums are accumulated left to right, the runs are walked in order subtracting whole run sums
    from u, and the run in which the remainder would turn negative is walked entry by entry."""
    n = len(p)
    per = (n + 63) // 64
    r = u
    for l in range(64):
        lo, hi = min(l * per, n), min(l * per + per, n)
        s = np.float64(0.0)
        for q in range(lo, hi):
            s = s + p[q]
        if r - s < 0:
            for q in range(lo, hi):
                r = r - p[q]
                if r < 0:
                    return q
            return max(hi - 1, 0)
        r = r - s
    return n - 1


class FbgmmBatch(object):
    """Batch sampler state built from an oracle UnigramAcousticWordseg / BigramAcousticWordseg."""

    def __init__(self, seg, n_gibbs_blocks=8, n_stat_blocks=8, seed=0, rank=0, world=1, all_gather_object=None):
        """rank / world / all_gather_object: the rank-split protocol of the multi-GPU path (rank r
        resamples only its slices and exchanges the block's partial sums -- and transcripts, with a
        language model -- after every step); world = 1 is the plain specification."""
        self.seg = seg
        assert n_stat_blocks % world == 0
        self.rank, self.world, self.ago = rank, world, all_gather_object
        self.s_lo, self.s_hi = rank * (n_stat_blocks // world), (rank + 1) * (n_stat_blocks // world)
        am = seg.acoustic_model
        c = am.components
        self.X = c.X
        self.K_max = c.K_max
        self.D = c.D
        self.cov = am.covariance_type
        self.lms = am.lms
        self.alpha = getattr(am, "alpha", None)
        self.lm = getattr(seg, "lm", None)
        self.seed = seed
        self.B = n_gibbs_blocks
        self.S = n_stat_blocks
        u = seg.utterances
        self.slot = c.assignments.copy()               # slot of every embedding row, -1 = unassigned
        sb = no.block_bounds(u.D, self.S)
        # utterance ranges [lo, hi) of (slice s, block b)
        self.ranges = [[None] * self.B for _ in range(self.S)]
        for s in range(self.S):
            bb = no.block_bounds(sb[s + 1] - sb[s], self.B)
            for b in range(self.B):
                self.ranges[s][b] = (sb[s] + bb[b], sb[s] + bb[b + 1])
        if self.cov == "fixed":
            self.prec = np.asarray(c.precision, np.float64)
            self.mu_0 = np.asarray(c.mu_0, np.float64)
            self.prec_0 = np.asarray(c.precision_0, np.float64)
        else:
            p = c.prior
            self.m_0, self.k_0, self.v_0, self.S_0 = (np.asarray(p.m_0, np.float64), float(p.k_0), float(p.v_0),
                                                      np.asarray(p.S_0, np.float64))
        # np.square(X) in the dtype of X, as gaussian_components_diag.py:125 caches it
        self.XX = np.square(self.X)
        self.P = [[self._partial(s, b) for b in range(self.B)] for s in range(self.S)]
        if self.lm is not None:
            self.uni = self.lm.unigram_counts.copy()
            self.big = self.lm.bigram_counts.copy()
            # replicated transcripts (slots of every utterance's segments)
            self.tr = [[self.slot[e] for e in self._tokens(i)] for i in range(u.D)]

    # ------------------------------------------------------------------ statistics
    def _tokens(self, i):
        return [e for e in self.seg.utterances.get_segmented_embeds_i(i) if e != -1]

    def _partial(self, s, b):
        cnt = np.zeros(self.K_max, np.int64)
        sx = np.zeros((self.K_max, self.D), np.float64)
        sxx = np.zeros((self.K_max, self.D), np.float64)
        lo, hi = self.ranges[s][b]
        for i in range(lo, hi):
            for e in self._tokens(i):                   # token order: utterance, then segment
                k = self.slot[e]
                cnt[k] += 1
                sx[k] += self.X[e]
                sxx[k] += self.XX[e]
        return cnt, sx, sxx

    def stats_excluding(self, b):
        """(counts, sum x, sum x^2) of all tokens outside block b; b = -1: of all tokens."""
        per_slice = []
        for s in range(self.S):
            cnt = np.zeros(self.K_max, np.int64)
            sx = np.zeros((self.K_max, self.D), np.float64)
            sxx = np.zeros((self.K_max, self.D), np.float64)
            for bp in range(self.B):
                if bp == b:
                    continue
                cnt = cnt + self.P[s][bp][0]
                sx = sx + self.P[s][bp][1]
                sxx = sxx + self.P[s][bp][2]
            per_slice.append((cnt, sx, sxx))
        return (no.tree_sum([p[0] for p in per_slice]), no.tree_sum([p[1] for p in per_slice]),
                no.tree_sum([p[2] for p in per_slice]))

    # ------------------------------------------------------------------ densities
    def derive(self, cnt, sx, sxx):
        """Per-slot parameters of the predictive densities from the sums."""
        n = cnt.astype(np.float64)[:, None]
        d = {}
        with np.errstate(all="ignore"):
            if self.cov == "fixed":       # gaussian_components_fixedvar.py:153-170, 317-325
                pN = self.prec_0 + n * self.prec
                d["mean"] = (self.prec_0 * self.mu_0 + self.prec * sx) / pN
                d["pp"] = pN * self.prec / (pN + self.prec)
                d["const"] = -0.5 * self.D * math.log(2. * np.pi) + 0.5 * np.log(d["pp"]).sum(axis=1)
            else:                         # gaussian_components_diag.py:162-177, 332-345, 237-259
                k_N, v_N = self.k_0 + n, self.v_0 + n
                m_N = (self.k_0 * self.m_0 + sx) / k_N
                var = (k_N + 1.) / (k_N * v_N) * (self.S_0 + self.k_0 * np.square(self.m_0) + sxx - k_N * np.square(m_N))
                d["mean"] = m_N
                d["q"] = 1. / var * (1. / v_N)
                vn = v_N[:, 0]
                d["half"] = (vn + 1.) / 2.
                d["const"] = (self.D * (gammaln((vn + 1.) / 2.) - gammaln(vn / 2.) - 0.5 * np.log(vn) - 0.5 * math.log(np.pi))
                              - 0.5 * np.log(var).sum(axis=1))
        d["active"] = cnt > 0
        d["cnt"] = cnt
        return d

    def log_prior_pred(self, x):
        x = x.astype(np.float64)
        if self.cov == "fixed":           # gaussian_components_fixedvar.py:224-231
            return (-0.5 * self.D * math.log(2. * np.pi) + 0.5 * np.log(self.prec_0).sum()
                    - 0.5 * (np.square(x - self.mu_0) * self.prec_0).sum())
        var = (self.k_0 + 1.) / (self.k_0 * self.v_0) * self.S_0       # gaussian_components_diag.py:215-222
        v = self.v_0
        return (self.D * (gammaln((v + 1.) / 2.) - gammaln(v / 2.) - 0.5 * math.log(v) - 0.5 * math.log(np.pi))
                - 0.5 * np.log(var).sum() - (v + 1.) / 2. * np.log(1. + 1. / v * np.square(x - self.m_0) / var).sum())

    def loglik(self, d, x):
        """log predictive of x under every slot: occupied slots their posterior predictive, empty
        slots the prior predictive."""
        x = x.astype(np.float64)
        with np.errstate(all="ignore"):
            if self.cov == "fixed":
                ll = d["const"] - 0.5 * (np.square(d["mean"] - x) * d["pp"]).sum(axis=1)
            else:
                ll = d["const"] - d["half"] * np.log(1. + np.square(d["mean"] - x) * d["q"]).sum(axis=1)
        return np.where(d["active"], ll, self.log_prior_pred(x))

    def prior_z(self, d, j_prev, uni, big):
        """Unnormalised assignment prior over the slots (times lms)."""
        K = self.K_max
        if self.lm is None:               # fbgmm.py:436-440
            return self.lms * np.log(float(self.alpha) / K + d["cnt"])
        lm = self.lm
        tot = int(np.sum(uni))
        pi = (uni + float(lm.a) / K) / (tot + lm.a)
        if j_prev is None:                # bigram_lms.py:64-69
            return (np.log(uni + float(lm.a) / K) - np.log(tot + lm.a)) * self.lms
        pij = (1 - lm.intrp_lambda) * (big[j_prev, :] + float(lm.b) / K) / (uni[j_prev] + float(lm.b))
        return np.log(lm.intrp_lambda * pi + pij) * self.lms

    def log_marg(self, d, x, uni=None, big=None):
        """Score of one span: fbgmm.py:256-285 / bigram_acoustic_wordseg.py:314-329."""
        if self.lm is None:
            z = self.lms * (np.log(float(self.alpha) / self.K_max + d["cnt"]) - math.log(int(np.sum(d["cnt"])) + self.alpha))
        else:
            z = self.prior_z(d, None, uni, big)
        return _sp_logsumexp(z + self.loglik(d, x))

    # ------------------------------------------------------------------ one sweep
    def sweep(self, sweep_index, anneal_temp=1.0, anneal_gibbs_am=False):
        seg, u = self.seg, self.seg.utterances
        log_probs = np.zeros(u.D)
        for b in range(self.B):
            cnt, sx, sxx = self.stats_excluding(b)
            d = self.derive(cnt, sx, sxx)
            uni = big = None
            if self.lm is not None:       # LM counts of all other blocks (integers: exact)
                uni, big = self.uni.copy(), self.big.copy()
                for s in range(self.S):
                    for i in range(*self.ranges[s][b]):
                        self._lm_count(uni, big, self.tr[i], -1)
            new_state = {}
            for s in range(self.s_lo, self.s_hi):
                for i in range(*self.ranges[s][b]):
                    N = u.lengths[i]
                    tri = (N * N + N) // 2
                    vec = -np.inf * np.ones(tri)
                    for j in range(tri):
                        e = u.vec_ids[i, j]
                        if e == -1:
                            continue
                        dur = u.durations[i, j]
                        vec[j] = -np.inf if np.isnan(dur) else self.log_marg(d, self.X[e], uni, big) * dur ** seg.time_power_term
                    vec = vec + seg.wip
                    uniforms = (u01(self.seed, sweep_index, i, j) for j in range(10 ** 9))
                    lp, bnd = no.forward_backward(vec, 0.0, N, seg.n_slices_min, seg.n_slices_max, i, anneal_temp,
                                                  uniforms=uniforms)
                    old = self._tokens(i)
                    bounds_new = np.asarray(bnd, dtype=bool)
                    new_state[i] = (lp, bounds_new, old)
            # all utterances of the block were sampled from the same statistics: apply
            for i, (lp, bounds_new, old) in new_state.items():
                for e in old:
                    self.slot[e] = -1
            for i, (lp, bounds_new, old) in new_state.items():
                N = u.lengths[i]
                u.boundaries[i, :N] = bounds_new
                log_probs[i] = lp
                j_prev = None
                temp = anneal_temp if anneal_gibbs_am else 1.0
                for t, e in enumerate(self._tokens(i)):
                    z = self.prior_z(d, j_prev, uni, big) + self.loglik(d, self.X[e])
                    if temp != 1:
                        z = z - _sp_logsumexp(z)
                        p = np.exp(1. / temp * z - _sp_logsumexp(1. / temp * z))
                    else:
                        p = np.exp(z - _sp_logsumexp(z))
                    k = draw_chunked(p, u01(self.seed, sweep_index, i, u.N_max + t))
                    self.slot[e] = k
                    j_prev = k if self.lm is not None else None
            mine = {s: self._partial(s, b) for s in range(self.s_lo, self.s_hi)}
            for part in ([mine] if self.world == 1 else self.ago(mine)):
                for s, p in part.items():
                    self.P[s][b] = p
            if self.lm is not None:
                mine = {i: [self.slot[e] for e in self._tokens(i)] for i in new_state}
                for part in ([mine] if self.world == 1 else self.ago(mine)):
                    for i, t in part.items():
                        self.tr[i] = t
                for s in range(self.S):
                    for i in range(*self.ranges[s][b]):
                        self._lm_count(uni, big, self.tr[i], +1)
                self.uni, self.big = uni, big
        return log_probs

    @staticmethod
    def _lm_count(uni, big, transcript, sign):
        j_prev = None
        for k in transcript:
            uni[k] += sign
            if j_prev is not None:
                big[j_prev, k] += sign
            j_prev = k

    # ------------------------------------------------------------------ views
    def canonical(self):
        """(assignments relabelled like the reference keeps them, K)."""
        cnt = self.stats_excluding(-1)[0]
        order = np.where(cnt > 0)[0]
        remap = -np.ones(self.K_max, np.int64)
        remap[order] = np.arange(len(order))
        out = np.where(self.slot >= 0, remap[np.maximum(self.slot, 0)], -1)
        return out, len(order)
