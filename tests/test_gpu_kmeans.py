"""
GPU parity tests of the k-means path: the HIP kernels (through the C ABI) against
  - the golden vectors captured from the reference (tests/golden/*.npz),
  - the oracle on the same seeded inputs,
  - size-independent properties at the headline size.
Bit-exact everywhere (integer / index results and the float32 or float64 scores, whose
arithmetic is specified by the reference: DESIGN.md "bit-exact contract").
"""
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
    _abi.ctx()          # raises if libsegk.so is missing or the device is not gfx950
    return torch


def _components(X, means, K_max=None):
    """A device KMeansComponents whose `means` are overwritten with the given matrix."""
    import torch
    from segmentalist_amd.kmeans_components import KMeansComponents
    n = X.shape[0]
    K = means.shape[0] if K_max is None else K_max
    np.random.seed(0)
    c = KMeansComponents(X, np.zeros(n, dtype=int), K)
    c.dev.means.copy_(torch.from_numpy(np.ascontiguousarray(means)).to(c.dev.means.device))
    c.dev.prepare()
    return c


# ------------------------------------------------------------------ A1
@pytest.mark.parametrize("case", cases.A1_CASES, ids=[c[0] for c in cases.A1_CASES])
def test_a1_scores_bit_exact_vs_reference(gpu, golden, case):
    g = golden("kernels")
    name, D, K, n, dtype = case
    X, means = cases.a1_inputs(*case)
    c = _components(X, means)
    want = g["a1_%s_scores" % name]
    for i in range(n):
        got = c.neg_sqrd_norm(i)
        assert got.dtype == want.dtype
        assert np.array_equal(got, want[i]), (name, i)
    mx, am, nbrute = c.dev.exact_max(np.arange(n))
    assert np.array_equal(mx, g["a1_%s_max" % name].astype(np.float64))
    assert np.array_equal(am, g["a1_%s_argmax" % name])


@pytest.mark.parametrize("D,K,n,dtype", [(100, 1000, 4096, "float32"), (39, 100, 3000, "float32"),
                                          (130, 257, 1500, "float32"), (17, 33, 700, "float64"),
                                          (200, 64, 600, "float32"), (300, 40, 300, "float32")])
@pytest.mark.parametrize("b3", ["2", "3", "0"], ids=["fp16x2", "bf16x3", "fp32mfma"])
def test_a1_max_argmax_vs_oracle_random(gpu, monkeypatch, D, K, n, dtype, b3):
    """Both filters (SEGK_SCORE_B3=0 forces the fp32-MFMA one where the bf16x3 one would be chosen)."""
    from oracle import c_oracle as co
    monkeypatch.setenv("SEGK_SCORE_B3", b3)
    rs = np.random.RandomState(D * 1000 + K)
    K_true = max(2, K // 2)
    mu = rs.randn(K_true, D)
    X = mu[rs.randint(0, K_true, n)] + 0.3 * rs.randn(n, D)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X = X.astype(dtype)
    means = (mu[rs.randint(0, K_true, K)] + 0.05 * rs.randn(K, D))
    means /= np.linalg.norm(means, axis=1, keepdims=True)
    means = means.astype(dtype)
    means[K // 2] = means[1]          # exact duplicate of row 1 ...
    X[5] = means[1]                   # ... and a data point sitting on it: an exact tie, lowest index wins
    c = _components(X, means)
    mx, am, nbrute = c.dev.exact_max(np.arange(n))
    want_mx, want_am = co.kmeans_max_argmax(means, X)
    assert np.array_equal(am, want_am)
    assert np.array_equal(mx, want_mx)
    assert am[5] == 1
    assert 1 <= nbrute < n             # the tie takes the full scan; the filter decides most rows



@pytest.mark.parametrize("D,K,n", [(100, 1000, 6000), (8, 33, 3000), (40, 257, 5000), (128, 2300, 4000), (12, 64, 2000),
                                    (64, 700, 9000)])
def test_a1_hinted_path_vs_oracle_random(gpu, monkeypatch, D, K, n):
    """segk_kmeans_score_hinted forced at small sizes (SEGK_SCORE_HINT=1) on random data with engineered exact ties: hints
    that are right, wrong, garbage and duplicates of the winner; row ranges and id lists (with -1 entries and an offset);
    one, two and four LDS ranges of tile images (K = 2300: 72 tiles).  cand_k / cand_s == the C oracle, bit for bit."""
    import torch
    from oracle import c_oracle as co
    monkeypatch.setenv("SEGK_SCORE_HINT", "1")
    rs = np.random.RandomState(D * 7 + K)
    K_true = max(2, K // 2)
    mu = rs.randn(K_true, D)
    X = mu[rs.randint(0, K_true, n)] + 0.3 * rs.randn(n, D)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X = X.astype(np.float32)
    means = mu[rs.randint(0, K_true, K)] + 0.05 * rs.randn(K, D)
    means /= np.linalg.norm(means, axis=1, keepdims=True)
    means = means.astype(np.float32)
    means[K // 2] = means[1]          # exact duplicate of row 1 ...
    X[5] = means[1]                   # ... and a data point sitting on it
    c = _components(X, means)
    from segmentalist_amd import _abi
    from segmentalist_amd._abi import ptr
    import ctypes as C
    _abi.check(_abi.lib().segk_kmeans_mark_duplicates(c.dev._ctx, c.dev._cp(), C.byref(c.dev.m), None, _abi.stream()))
    want_s, want_k = co.kmeans_max_argmax(means, X)
    ident = torch.arange(K, dtype=torch.int32, device="cuda")

    def run(hints, ids=None, row0=0, nrows=None, remap=ident):
        c.dev.cand_k.copy_(torch.from_numpy(hints.astype(np.int32)).cuda())
        c.dev.cand_s.fill_(float("nan"))
        _abi.check(_abi.lib().segk_profile_enable(c.dev._ctx, 1))
        if ids is None:
            c.dev.score_rows(row0=row0, n=nrows, hint_remap=remap)
        else:
            c.dev.score_rows(ids=torch.from_numpy(ids.astype(np.int32)).cuda(), hint_remap=remap)
        torch.cuda.synchronize()
        kind = int(_abi.lib().segk_profile_last_kind(c.dev._ctx))
        _abi.check(_abi.lib().segk_profile_enable(c.dev._ctx, 0))
        assert kind == 5, kind
        return c.dev.cand_k.cpu().numpy(), c.dev.cand_s.cpu().numpy()

    # right hints, whole range
    k, s = run(want_k)
    assert np.array_equal(k, want_k) and np.array_equal(s, want_s.astype(np.float64))
    assert k[5] == 1
    # hint = the duplicate of the winner
    adv = want_k.copy()
    adv[want_k == 1] = K // 2
    k, s = run(adv)
    assert np.array_equal(k, want_k) and np.array_equal(s, want_s.astype(np.float64))
    # wrong / garbage hints on a sub-range with an offset
    junk = rs.choice(np.array([-1, -5, K, 2 ** 29 | 1, 2 ** 30 | 2, 2 ** 31 - 1], dtype=np.int64), n)
    hints = np.where(rs.rand(n) < 0.3, want_k, np.where(rs.rand(n) < 0.5, rs.randint(0, K, n), junk))
    r0, nr = 37, n - 100
    k, s = run(hints, row0=r0, nrows=nr)
    assert np.array_equal(k[r0:r0 + nr], want_k[r0:r0 + nr]) and np.array_equal(s[r0:r0 + nr], want_s[r0:r0 + nr].astype(np.float64))
    assert np.array_equal(k[:r0], hints[:r0].astype(np.int32)) and np.isnan(s[:r0]).all()       # rows outside the call untouched
    # an id list with skipped entries, through a relabel table
    perm = rs.permutation(K).astype(np.int32)
    inv = np.argsort(perm)
    ids = rs.permutation(n)[: n // 2].astype(np.int64)
    ids[::17] = -1
    k, s = run(inv[want_k], ids=ids, remap=torch.from_numpy(perm).cuda())
    live = ids[ids >= 0]
    assert np.array_equal(k[live], want_k[live]) and np.array_equal(s[live], want_s[live].astype(np.float64))


@pytest.mark.parametrize("K,spread", [(40, 3e-4), (200, 1e-7), (1000, 2e-4)], ids=["tens_of_candidates", "band_overflow", "four_ranges"])
def test_band_stage_with_crowded_bands(gpu, monkeypatch, K, spread):
    """The band stage of the hinted path (segk_score_band.hip) where MANY components lie inside the band of a row's largest
    filter value: the means are tiny perturbations of three base vectors, so every row has its base's whole family as
    candidates -- a dozen per row (several passes of the candidate list), hundreds (more than a wave's list holds: those rows
    take the full scan), and the same on four LDS ranges.  Hints right, wrong and absent: cand_k / cand_s == the C oracle."""
    import torch
    from oracle import c_oracle as co
    from segmentalist_amd import _abi
    monkeypatch.setenv("SEGK_SCORE_HINT", "1")
    rs = np.random.RandomState(K)
    D, n = 16, 3000
    base = rs.randn(3, D)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    means = (base[rs.randint(0, 3, K)] * (1.0 + spread * rs.randn(K, 1)) + spread * rs.randn(K, D)).astype(np.float32)
    X = base[rs.randint(0, 3, n)] + 0.2 * rs.randn(n, D)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X = X.astype(np.float32)
    c = _components(X, means)
    want_s, want_k = co.kmeans_max_argmax(means, X)
    ident = torch.arange(K, dtype=torch.int32, device="cuda")
    for hints in (want_k, rs.randint(0, K, n), np.full(n, -1)):
        c.dev.cand_k.copy_(torch.from_numpy(np.asarray(hints, dtype=np.int32)).cuda())
        c.dev.cand_s.fill_(float("nan"))
        _abi.check(_abi.lib().segk_profile_enable(c.dev._ctx, 1))
        c.dev.score_rows(hint_remap=ident)
        torch.cuda.synchronize()
        assert int(_abi.lib().segk_profile_last_kind(c.dev._ctx)) == 5
        _abi.check(_abi.lib().segk_profile_enable(c.dev._ctx, 0))
        assert np.array_equal(c.dev.cand_k.cpu().numpy(), want_k)
        assert np.array_equal(c.dev.cand_s.cpu().numpy(), want_s.astype(np.float64))
    c.dev.check_status()


def test_a1_filter_candidate_is_within_margin(gpu):
    """The fp32 MFMA filter's winner must be the true argmax or inside the proven margin."""
    import torch
    rs = np.random.RandomState(7)
    n, D, K = 2048, 100, 1000
    X = rs.randn(n, D).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    means = (rs.randn(K, D) * 0.2).astype(np.float32)
    c = _components(X, means)
    c.dev.score_rows()
    torch.cuda.synchronize()
    f = X.astype(np.float64) @ means.astype(np.float64).T - 0.5 * (means.astype(np.float64) ** 2).sum(1)
    ck = c.dev.cand_k.cpu().numpy()
    cf = c.dev.cand_f.cpu().numpy()
    top = np.sort(f, axis=1)[:, ::-1]
    assert np.max(np.abs(cf[:, 0] - top[:, 0])) < 2e-5
    assert np.max(np.abs(cf[:, 1] - top[:, 1])) < 2e-5
    assert np.mean(ck == np.argmax(f, axis=1)) > 0.999


# ------------------------------------------------------------------ A6 / A7 / A8 on the device
def test_dp_functions_vs_reference(gpu, golden):
    import torch
    from segmentalist_amd import _abi
    from segmentalist_amd.device import to_dev
    g = golden("kernels")
    dpc = cases.dp_cases()
    for n_min in (0, 1):
        for n_max in (0, 2, 6):
            sel = [i for i, c in enumerate(dpc) if c["n_min"] == n_min and c["n_max"] == n_max]
            Ns = np.array([dpc[i]["N"] for i in sel], dtype=np.int32)
            offs = np.concatenate([[0], np.cumsum([len(dpc[i]["vec"]) for i in sel])]).astype(np.int64)
            vecs = np.concatenate([dpc[i]["vec"] for i in sel])
            Nmax = int(Ns.max())
            vecs_t, Ns_t, offs_t = to_dev(vecs), to_dev(Ns), to_dev(offs)   # keep the buffers alive
            ob = np.concatenate([[0], np.cumsum([c["N"] for c in dpc])])
            ou = np.concatenate([[0], np.cumsum([c["N"] + 1 for c in dpc])])
            for kind, key, temp, lpc in [(0, "km", 1.0, 0.0), (1, "vt", 1.0, 0.0), (2, "fb", 1.0, -0.25),
                                         (2, "fa", 1.7, -0.25)]:
                P = len(sel)
                bounds = torch.zeros((P, Nmax), dtype=torch.uint8, device="cuda")
                totals = torch.zeros(P, dtype=torch.float64, device="cuda")
                nd = torch.zeros(P, dtype=torch.int32, device="cuda")
                st = torch.zeros(P, dtype=torch.int32, device="cuda")
                work = torch.zeros((P, 3 * Nmax + 2), dtype=torch.float64, device="cuda")
                u = None
                if kind == 2:
                    un = np.full((P, Nmax + 1), 0.5)
                    for r, i in enumerate(sel):
                        src = g["dp_%s_uniforms" % key][ou[i]:ou[i + 1]]
                        un[r, :len(src)] = np.nan_to_num(src, nan=0.5)
                    u = to_dev(un)
                _abi.check(_abi.lib().segk_dp_tri(
                    _abi.ctx(), kind, _abi.ptr(vecs_t), _abi.ptr(Ns_t), _abi.ptr(offs_t), P,
                    n_min, n_max, lpc, temp, _abi.ptr(u), Nmax + 1, _abi.ptr(bounds), Nmax, _abi.ptr(totals),
                    _abi.ptr(nd), _abi.ptr(st), _abi.ptr(work), 3 * Nmax + 2, _abi.stream()))
                B = bounds.cpu().numpy().astype(bool)
                T = totals.cpu().numpy()
                for r, i in enumerate(sel):
                    N = dpc[i]["N"]
                    want_t = g["dp_%s_total" % key][i]
                    if kind == 2 and np.isnan(want_t):
                        assert st[r].item() == 1
                        continue
                    assert np.array_equal(B[r, :N], g["dp_%s_bounds" % key][ob[i]:ob[i + 1]]), (key, i)
                    if kind == 0:
                        assert T[r] == want_t or (np.isnan(T[r]) and np.isnan(want_t)), (key, i)
                    else:
                        assert np.isclose(T[r], want_t, rtol=1e-13, atol=0) or T[r] == want_t, (key, i)
                    if kind == 2:
                        assert nd[r].item() == g["dp_%s_ndraws" % key][i]


def test_module_level_viterbi_function(gpu, golden):
    from segmentalist_amd.kmeans_acoustic_wordseg import forward_backward_kmeans_viterbi
    g = golden("kernels")
    dpc = cases.dp_cases()
    ob = np.concatenate([[0], np.cumsum([c["N"] for c in dpc])])
    for i in range(0, len(dpc), 17):
        c = dpc[i]
        tot, b = forward_backward_kmeans_viterbi(c["vec"], c["N"], c["n_min"], c["n_max"])
        assert np.array_equal(b, g["dp_km_bounds"][ob[i]:ob[i + 1]])
        assert tot == g["dp_km_total"][i] or np.isnan(tot)


# ------------------------------------------------------------------ A11 mutators
def test_component_mutators_vs_oracle(gpu):
    from oracle import np_oracle as no
    from segmentalist_amd.kmeans_components import KMeansComponents
    for dtype in (np.float32, np.float64):
        rs = np.random.RandomState(11)
        X = rs.randn(60, 7).astype(dtype)
        assign = rs.randint(0, 5, 60)
        assign[rs.rand(60) < 0.4] = -1
        assign = no.consecutive_labels(assign)
        np.random.seed(4)
        ref = no.KMeansComponents(X, assign.copy(), 8)
        np.random.seed(4)
        dev = KMeansComponents(X, assign.copy(), 8)

        def same():
            assert dev.K == ref.K
            assert np.array_equal(dev.counts, ref.counts)
            assert np.array_equal(dev.assignments, ref.assignments)
            assert np.array_equal(dev.mean_numerators, ref.mean_numerators)
            assert np.array_equal(dev.means, ref.means)
            assert dev.means.dtype == ref.means.dtype
        same()
        free = list(np.where(ref.assignments == -1)[0])
        used = list(np.where(ref.assignments != -1)[0])
        for step in range(40):
            op = rs.randint(0, 3)
            if op == 0 and free:
                i = free.pop(rs.randint(len(free)))
                k = int(rs.randint(0, 8))
                ref.add_item(i, k)
                dev.add_item(i, k)
                used.append(i)
            elif op == 1 and used:
                i = used.pop(rs.randint(len(used)))
                ref.del_item(i)
                dev.del_item(i)
                free.append(i)
            else:
                ref.clean_components()
                dev.clean_components()
            same()
        assert abs(dev.sum_neg_sqrd_norm() - ref.sum_neg_sqrd_norm()) <= 1e-9 * abs(ref.sum_neg_sqrd_norm())
        assert dev.get_max_assignments(list(range(10))) == [int(k) for k in ref.get_max_assignments(range(10))]


# ------------------------------------------------------------------ chains, sequential mode (reference chain)
@pytest.mark.parametrize("chain", cases.KMEANS_CHAINS, ids=[c[0] for c in cases.KMEANS_CHAINS])
@pytest.mark.parametrize("init", ["spread", "rand"])
def test_sequential_chain_bit_exact_vs_reference(gpu, golden, chain, init):
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    g = golden("chains")
    name, n_utt, D, K, seed, ragged, N, nmax, dtype = chain
    corpus = cases.chain_corpus(n_utt, D, K, seed, ragged, N, nmax, dtype)
    random.seed(1)
    np.random.seed(1)
    seg = kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5,
                                     init_am_assignments=init, wip=0)
    c = seg.acoustic_model.components
    tag = "%s_%s" % (name, init)
    assert np.array_equal(seg.utterances.boundaries, g[tag + "_init_bounds"])
    assert np.array_equal(c.assignments, g[tag + "_init_assign"])
    assert np.array_equal(c.random_means, g[tag + "_random_means"])
    for it in range(3):
        rec = seg.segment(1)
        assert np.array_equal(seg.utterances.boundaries, g[tag + "_bounds"][it]), it
        assert np.array_equal(c.assignments, g[tag + "_assign"][it]), it
        assert np.array_equal(c.means, g[tag + "_means"][it]), it
        assert c.K == g[tag + "_K"][it]
        assert rec["sum_neg_len_sqrd_norm"][0] == g[tag + "_rec_sum_neg_len_sqrd_norm"][it]
        assert np.isclose(rec["sum_neg_sqrd_norm"][0], g[tag + "_rec_sum_neg_sqrd_norm"][it], rtol=1e-10)
        assert rec["n_tokens"][0] == g[tag + "_rec_n_tokens"][it]
        assert rec["components"][0] == g[tag + "_rec_components"][it]
    assert np.array_equal(c.mean_numerators, g[tag + "_mean_numerators"])
    assert np.array_equal(c.counts, g[tag + "_counts"])
    recf = seg.acoustic_model.fit(3, consider_unassigned=False)
    assert np.array_equal(c.assignments, g[tag + "_fit_assign"])
    assert np.array_equal(recf["n_mean_updates"], g[tag + "_fit_n_mean_updates"])


def test_get_vec_embed_matches_oracle(gpu):
    from oracle import np_oracle as no
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    corpus = cases.chain_corpus(6, 8, 5, 99, True, 0, 5, "float32")
    random.seed(2); np.random.seed(2)
    ref = no.SegmentalKMeansWordseg(5, *corpus, n_slices_max=5, wip=-0.3)
    random.seed(2); np.random.seed(2)
    seg = kaw.SegmentalKMeansWordseg(5, *corpus, n_slices_max=5, wip=-0.3)
    for i in range(6):
        a = ref.get_vec_embed_neg_len_sqrd_norms(ref.utterances.vec_ids[i], ref.utterances.durations[i])
        b = seg.get_vec_embed_neg_len_sqrd_norms(seg.utterances.vec_ids[i], seg.utterances.durations[i])
        assert np.array_equal(a, b)
        assert seg.segment_i(i) == ref.segment_i(i)
        assert seg.get_unsup_transcript_i(i) == ref.get_unsup_transcript_i(i)


# ------------------------------------------------------------------ batch mode vs its CPU specification
@pytest.mark.parametrize("n_utt,D,K,nmax,dtype,n_blocks", [
    (12, 8, 6, 6, "float32", 8), (40, 16, 12, 6, "float32", 8), (40, 16, 12, 6, "float32", 1),
    (25, 5, 30, 4, "float64", 4), (64, 39, 40, 6, "float32", 8)])
def test_batch_sweep_bit_exact_vs_spec(gpu, n_utt, D, K, nmax, dtype, n_blocks):
    from oracle import np_oracle as no
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    corpus = cases.chain_corpus(n_utt, D, K, 1000 + n_utt, True, 0, nmax, dtype)
    for init in ("spread", "rand"):
        random.seed(5); np.random.seed(5)
        ref = no.SegmentalKMeansWordseg(K, *corpus, n_slices_max=nmax, init_am_assignments=init)
        random.seed(5); np.random.seed(5)
        seg = kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_max=nmax, init_am_assignments=init,
                                         sync="batch", n_stat_blocks=n_blocks)
        cr, cd = ref.acoustic_model.components, seg.acoustic_model.components
        for it in range(4):
            want = no.kmeans_batch_sweep(ref, n_blocks=n_blocks)
            rec = seg.segment(1)
            assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), it
            assert np.array_equal(cd.assignments, cr.assignments), it
            assert cd.K == cr.K
            assert np.array_equal(cd.counts, cr.counts)
            assert np.array_equal(cd.mean_numerators, cr.mean_numerators), it
            assert np.array_equal(cd.means, cr.means), it
            assert rec["sum_neg_len_sqrd_norm"][0] == want
            assert rec["n_tokens"][0] == ref.acoustic_model.get_n_assigned()


@pytest.mark.parametrize("n_utt,D,K,scale", [(600, 100, 130, 1.0), (300, 16, 40, 1.0), (300, 16, 40, 37.0)])
def test_operand_images_after_a_batch_sweep_equal_a_fresh_prepare(gpu, n_utt, D, K, scale):
    """The finalize kernel writes its rows' part of the fp16x2 image with the exponent the image had (the post kernel rebuilds
    it only when the exponent of the new means differs): after every sweep both operand images are, bit for bit, what
    segk_kmeans_prepare + segk_kmeans_mark_duplicates build from the same means -- also when the scale of the data moves the
    exponent between the initial means and the first sweep's."""
    import ctypes as C
    from segmentalist_amd import _abi, kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    corpus = list(make_corpus(n_utt, D, K, seed=77, N=12, n_slices_max=5))
    corpus[0] = {k: (v * scale).astype(np.float32) for k, v in corpus[0].items()}
    random.seed(3); np.random.seed(3)
    seg = kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_max=5, init_am_assignments="spread", sync="batch")
    dk = seg._dk
    L, ctx = _abi.lib(), _abi.ctx()
    for it in range(4):
        seg.batch_sweep_async()
        gpu.cuda.synchronize()
        dk.check_status()
        got32, got16 = dk.tiles.clone(), dk.tiles_b3.clone()
        dk.prepare()
        _abi.check(L.segk_kmeans_mark_duplicates(ctx, dk._cp(), C.byref(dk.m), None, _abi.stream()))
        gpu.cuda.synchronize()
        assert gpu.equal(got32.view(gpu.int32), dk.tiles.view(gpu.int32)), it
        a, b = got16.view(gpu.int32), dk.tiles_b3.view(gpu.int32)
        assert gpu.equal(a[:2], b[:2]), (it, a[:4].tolist(), b[:4].tolist())          # exponent, E_m
        assert gpu.equal(a[1024:], b[1024:]), it


@pytest.mark.parametrize("n_utt,D,K,N,nmax,n_blocks,sweeps,p_b", [(150, 16, 2500, 0, 6, 8, 3, 0.5), (1700, 8, 12, 20, 4, 1, 2, 0.7),
                                                                (1000, 8, 2, 20, 1, 1, 2, 1.0), (900, 12, 70, 20, 2, 2, 2, 0.8),
                                                                (400, 16, 600, 0, 6, 16, 3, 0.5), (260, 12, 900, 0, 5, 12, 3, 0.5),
                                                                (2000, 8, 300, 12, 4, 40, 3, 0.5), (200, 8, 8192, 0, 6, 8, 2, 0.5),
                                                                (300, 16, 2049, 0, 6, 4, 2, 0.5)],
                         ids=["ranges_of_128_components", "block_beyond_the_preloaded_keys", "compaction_overflow",
                              "two_large_blocks", "sixteen_blocks", "twelve_blocks", "forty_blocks", "largest_bank_8192",
                              "first_bank_beyond_the_band_stage_2049"])
def test_batch_statistics_kernel_fallbacks_vs_spec(gpu, n_utt, D, K, N, nmax, n_blocks, sweeps, p_b):
    """k_batch_sort_sum (csrc/segk_stats.hip) beyond the headline shape, against oracle/np_oracle.py kmeans_batch_sweep bit for bit:
    K_max > 2048 (ranges of 128 components instead of 32); a statistics block of more than 32 768 slots (its keys are not
    preloaded, the placement pass walks all slots again); more than 512 in-range tokens per wave (a window of one slice makes
    every landmark a token, two components share one range: the compacted list overflows and the workgroup falls back to the
    walk); two blocks of 9 000 slots with long per-component lists (several 32-row batches per list, component boundaries
    inside a batch); more than eight statistics blocks, with components founded in the first sweeps (the finalize kernel stages
    the flagged tokens of blocks 8.. without the speculative fetch of the first eight, and sums by the general tree)."""
    from oracle import np_oracle as no
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(n_utt, D, K, seed=4000 + n_utt, N=N, ragged=(N == 0), n_slices_max=nmax, N_range=(3, 9))
    random.seed(5); np.random.seed(5)
    ref = no.SegmentalKMeansWordseg(K, *corpus, n_slices_max=nmax, init_am_assignments="spread", p_boundary_init=p_b)
    random.seed(5); np.random.seed(5)
    seg = kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_max=nmax, init_am_assignments="spread", p_boundary_init=p_b, sync="batch",
                                     n_stat_blocks=n_blocks, flag_cap=40000)       # (one block: every token near an inactive row is on its list)
    cr, cd = ref.acoustic_model.components, seg.acoustic_model.components
    for it in range(sweeps):
        want = no.kmeans_batch_sweep(ref, n_blocks=n_blocks)
        rec = seg.segment(1)
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), it
        assert np.array_equal(cd.assignments, cr.assignments), it
        assert cd.K == cr.K
        assert np.array_equal(cd.counts, cr.counts)
        assert np.array_equal(cd.mean_numerators, cr.mean_numerators), it
        assert np.array_equal(cd.means, cr.means), it
        assert rec["sum_neg_len_sqrd_norm"][0] == want
        assert rec["n_tokens"][0] == ref.acoustic_model.get_n_assigned()


@pytest.mark.parametrize("n_utt,D,K,nmax,n_blocks,n_batches", [(40, 16, 12, 6, 4, 2), (64, 8, 9, 5, 8, 4), (33, 12, 30, 4, 2, 8),
                                                             (24, 5, 4, 5, 1, 3), (300, 100, 130, 6, 8, 2)])
def test_minibatch_sweep_bit_exact_vs_spec(gpu, n_utt, D, K, nmax, n_blocks, n_batches):
    """sync="batch", n_batches > 1 ("or per mini-batch of B utterances for fresher stats") against the executable
    specification oracle/np_oracle.py kmeans_minibatch_sweep: after every sweep boundaries, assignments, K, counts, numerators,
    means and the record total, bit for bit -- from a fresh segmenter (whose token lists are first derived from the initial
    segmentation), with more mini-batches than some blocks have utterances, with one block, and with an interleaved step of
    the sequential chain (the token lists are then rebuilt from the state it leaves)."""
    from oracle import np_oracle as no
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    corpus = cases.chain_corpus(n_utt, D, K, 2000 + n_utt, True, 0, nmax, "float32")
    for init in ("spread", "rand"):
        random.seed(5); np.random.seed(5)
        ref = no.SegmentalKMeansWordseg(K, *corpus, n_slices_max=nmax, init_am_assignments=init)
        random.seed(5); np.random.seed(5)
        seg = kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_max=nmax, init_am_assignments=init, sync="batch",
                                         n_stat_blocks=n_blocks, n_batches=n_batches)
        cr, cd = ref.acoustic_model.components, seg.acoustic_model.components
        totals = np.zeros(ref.utterances.D)
        for it in range(4):
            if it == 2:                                  # one utterance through the reference's sequential step on both sides
                want_i = ref.segment_i(3)
                assert seg.segment_i(3) == want_i
                totals[3] = want_i
            want = no.kmeans_minibatch_sweep(ref, n_blocks, n_batches, totals)
            rec = seg.segment(1)
            assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), it
            assert np.array_equal(cd.assignments, cr.assignments), it
            assert cd.K == cr.K
            assert np.array_equal(cd.counts, cr.counts)
            assert np.array_equal(cd.mean_numerators, cr.mean_numerators), it
            assert np.array_equal(cd.means, cr.means), it
            assert rec["sum_neg_len_sqrd_norm"][0] == want, it
            assert rec["n_tokens"][0] == ref.acoustic_model.get_n_assigned()


def test_hint_policy_of_the_batch_sweeper(gpu, monkeypatch):
    """The batch sweeper's use of hints (device.KMeansBatchSweeper._use_hints) at a size where the hinted path applies
    (3 000 utterances of the headline shape, whole sweeps and two mini-batches per sweep): no hints in the first two sweeps,
    hints afterwards -- the hinted kernels run (profile kind 5) and the rows the certificate leaves undecided stay a minority
    (a hint-miss regression -- stale or unmapped hints -- would show here) --, and the state after six sweeps is bit-identical
    to the same chain with SEGK_SCORE_HINT=0."""
    import ctypes as C
    import torch
    from segmentalist_amd import _abi, kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(3000, 100, 1000, seed=0, N=20, n_slices_max=6)
    L, ctx = _abi.lib(), _abi.ctx()
    for n_batches in (1, 2):
        states = {}
        for mode in ("default", "0"):
            if mode == "0":
                monkeypatch.setenv("SEGK_SCORE_HINT", "0")
            else:
                monkeypatch.delenv("SEGK_SCORE_HINT", raising=False)
            random.seed(0); np.random.seed(0)
            seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch",
                                             n_batches=n_batches)
            n_rows = seg._corpus.n_emb // n_batches
            for it in range(6):
                _abi.check(L.segk_profile_enable(ctx, 1))
                seg.batch_sweep_async()
                torch.cuda.synchronize()
                kind = int(L.segk_profile_last_kind(ctx))
                _abi.check(L.segk_profile_enable(ctx, 0))
                sc = (C.c_int32 * 2)()
                _abi.check(L.segk_kmeans_stage_counts(ctx, C.byref(seg._dk.cand), sc, _abi.stream()))
                if mode == "default":
                    assert (kind == 5) == (it >= 2), (n_batches, it, kind)
                    if it >= 3:
                        assert 0 < sc[0] < 0.35 * n_rows, (n_batches, it, sc[0], n_rows)
                else:
                    assert kind != 5
            seg._dk.check_status()
            c = seg.acoustic_model.components
            states[mode] = (c.assignments.copy(), c.means.copy(), c.counts.copy(), seg.utterances.boundaries.copy())
        for a, b in zip(states["default"], states["0"]):
            assert np.array_equal(a, b), n_batches


def test_batch_sweep_headline_shape_properties(gpu):
    """BASELINE config 3 shape (D=100, K=1000, 20 landmarks, n_slices_max=6) at 1500 utterances:
    size-independent properties + spot parity against the C oracle."""
    import torch
    from oracle import c_oracle as co
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(1500, 100, 1000, seed=0, N=20, n_slices_max=6)
    random.seed(0); np.random.seed(0)
    seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
    c = seg.acoustic_model.components
    X = c.X
    for it in range(2):
        means_before, K_before = c.means, c.K
        seg.segment(1)
        b = seg.utterances.boundaries
        assert b[:, -1].all()                                     # last landmark always a boundary
        assign = c.assignments
        toks = np.where(assign != -1)[0]
        assert len(toks) == b.sum()                               # one token per segment
        counts = c.counts
        assert counts.sum() == len(toks) and (counts[:c.K] > 0).all() and (counts[c.K:] == 0).all()
        # every token sits on the argmax of the means it was scored against (C oracle, ALL tokens): the raw argmax,
        # add_item's clamp replayed in token order (kmeans_components.py:102-106), then the row moves of
        # clean_components (the device's relabelling table)
        order = np.concatenate([np.asarray(seg.utterances.get_segmented_embeds_i(i), dtype=np.int64)
                                for i in range(seg.utterances.D)])
        assert np.array_equal(np.sort(order), toks)
        _, am = co.kmeans_max_argmax(means_before, X, order)
        remap = seg._dk.remap.cpu().numpy()
        K_cur = K_before
        for e, k in zip(order, am):
            k = int(k)
            if k > K_cur:
                k = K_cur
            if k == K_cur:
                K_cur += 1
            assert assign[e] == remap[k], (it, e, k)
        # statistics are exactly the sums of the assigned rows
        k0 = int(assign[toks[0]])
        rows = toks[assign[toks] == k0]
        assert np.allclose(c.mean_numerators[k0], X[rows].astype(np.float64).sum(0), rtol=1e-12, atol=1e-12)
    # idempotence of the kernel: re-running the score+segment stage on unchanged statistics
    # reproduces the same boundaries
    seg._dk.score_rows()
    before = seg._dev_bounds.clone()
    seg._dk.segment(seg._dev_bounds, 0, 6, 0.0)
    torch.cuda.synchronize()
    again = seg._dev_bounds.clone()
    seg._dk.segment(seg._dev_bounds, 0, 6, 0.0)
    torch.cuda.synchronize()
    assert torch.equal(again, seg._dev_bounds)
    assert before.shape == again.shape


@pytest.mark.parametrize("D,K,scale", [(100, 1000, 1.0), (128, 513, 1.0), (40, 257, 30.0), (16, 64, 1e-3)])
@pytest.mark.parametrize("pieces", ["2", "3"], ids=["fp16x2", "bf16x3"])
def test_split_precision_filter_error_is_far_inside_the_proven_bound(gpu, monkeypatch, D, K, scale, pieces):
    """The split-precision filter's values against float64: the observed error must sit well inside
    E1' (the bound assumes one rounding of size u per accumulated product and the stated split
    residuals; a factor-4 head-room shows that neither assumption is violated by the matrix pipe)."""
    import torch
    monkeypatch.setenv("SEGK_SCORE_B3", pieces)
    rs = np.random.RandomState(D + K)
    n = 4096
    X = (rs.randn(n, D) * scale).astype(np.float32)            # full 24-bit significands, mixed signs
    means = (rs.randn(K, D) * scale).astype(np.float32)
    means[: K // 4] *= 3.0                                      # unequal norms: the constants -|m|^2/2 matter
    c = _components(X, means)
    assert c.dev.corpus.Xb3 is not None and c.dev.corpus.c.sp_pieces == int(pieces), "split images missing"
    c.dev.score_rows()
    torch.cuda.synchronize()
    ck = c.dev.cand_k.cpu().numpy().astype(np.int64)
    cf = c.dev.cand_f.cpu().numpy().astype(np.float64)
    X64, M64 = X.astype(np.float64), means.astype(np.float64)
    f1 = np.einsum("nd,nd->n", X64, M64[ck]) - 0.5 * (M64[ck] ** 2).sum(1)
    u = 2.0 ** -24
    KP = (D + 15) // 16 * 16
    xn = np.linalg.norm(X64, axis=1)
    Mmax = np.sqrt((M64 ** 2).sum(1).max())
    e1 = (1.02 * (KP + 16) + (16 if pieces == "2" else 0)) * u * (xn * Mmax + 0.5 * Mmax ** 2)
    ratio = np.abs(cf[:, 0] - f1) / e1
    assert ratio.max() < 0.25, ratio.max()
    # and the decisions are the reference's: exact argmax / max after the exact stage
    from oracle import c_oracle as co
    mx, am, _ = c.dev.exact_max(np.arange(n))
    want_mx, want_am = co.kmeans_max_argmax(means, X)
    assert np.array_equal(am, want_am) and np.array_equal(mx, want_mx)


@pytest.mark.parametrize("pieces", ["2", "3"], ids=["fp16x2", "bf16x3"])
def test_split_precision_filter_wide_dynamic_range(gpu, monkeypatch, pieces):
    """Rows and means whose elements span nine decades (most of them subnormal or flushed in fp16 after
    the power-of-two scaling), a corpus-wide scale far from 1, duplicated means: the decisions after the
    exact stage are still the reference's, bit for bit, and the filter error stays inside the bound."""
    import torch
    from oracle import c_oracle as co
    monkeypatch.setenv("SEGK_SCORE_B3", pieces)
    rs = np.random.RandomState(99)
    n, D, K = 3000, 64, 200
    for scale in (1.0, 3e-4, 7e3):
        X = (rs.randn(n, D) * 10.0 ** rs.uniform(-9, 0, size=(n, D)) * scale).astype(np.float32)
        means = (rs.randn(K, D) * 10.0 ** rs.uniform(-9, 0, size=(K, D)) * scale).astype(np.float32)
        means[7] = means[3]                                   # exact duplicate: ties go to the lowest index
        X[11] = means[3]
        c = _components(X, means)
        assert c.dev.corpus.c.sp_pieces == int(pieces)
        mx, am, nbrute = c.dev.exact_max(np.arange(n))
        want_mx, want_am = co.kmeans_max_argmax(means, X)
        assert np.array_equal(am, want_am) and np.array_equal(mx, want_mx)
        assert am[11] == 3
        ck = c.dev.cand_k.cpu().numpy().astype(np.int64)
        cf = c.dev.cand_f.cpu().numpy().astype(np.float64)
        X64, M64 = X.astype(np.float64), means.astype(np.float64)
        f1 = np.einsum("nd,nd->n", X64, M64[ck]) - 0.5 * (M64[ck] ** 2).sum(1)
        u = 2.0 ** -24
        xn = np.linalg.norm(X64, axis=1)
        Mmax = np.sqrt((M64 ** 2).sum(1).max())
        e1 = (1.02 * (64 + 16) + (16 if pieces == "2" else 0)) * u * (xn * Mmax + 0.5 * Mmax ** 2)
        sel = np.isfinite(cf[:, 0])
        assert (np.abs(cf[sel, 0] - f1[sel]) / e1[sel]).max() < 0.5


# ------------------------------------------------------------------ one-product pre-filter
@pytest.mark.parametrize("D,K,n,scale", [(100, 1000, 4096, 1.0), (128, 513, 3000, 1.0), (40, 257, 2500, 30.0),
                                         (16, 64, 1500, 1e-3), (8, 31, 700, 1.0), (64, 33, 5000, 7e3),
                                         (108, 130, 3000, 1.0), (12, 40, 2000, 1.0), (28, 70, 2500, 0.1), (100, 3100, 3000, 1.0)])
def test_prefilter_decisions_are_the_references(gpu, monkeypatch, D, K, n, scale):
    """SEGK_SCORE_PRE=1 forces the one-product fp16 pre-filter (normally used above one round of the
    chip) in front of the split-precision kernel: max / argmax after the exact stage stay the
    reference's bit for bit -- clustered rows, an exact tie, a duplicated mean inside one PAIR of
    components (the pre-filter tracks pairs and lets the exact stage pick the member).  The shapes cover the four
    compile-time layouts of the exact stage (D = 16 KS - 4 V, V = 0..3), tables of one and several LDS ranges, and one that
    needs more than eight (K = 3100: the exact stage that transposes whole rows through LDS, k_kmeans_exact_pair3)."""
    from oracle import c_oracle as co
    monkeypatch.setenv("SEGK_SCORE_PRE", "1")
    rs = np.random.RandomState(D * 1000 + K + 1)
    K_true = max(2, K // 2)
    mu = rs.randn(K_true, D)
    X = mu[rs.randint(0, K_true, n)] + 0.3 * rs.randn(n, D)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X = (X * scale).astype(np.float32)
    means = (mu[rs.randint(0, K_true, K)] + 0.05 * rs.randn(K, D))
    means /= np.linalg.norm(means, axis=1, keepdims=True)
    means = (means * scale).astype(np.float32)
    means[K // 2] = means[1]          # exact duplicate in another pair
    means[7] = means[6]               # exact duplicate inside the pair (6, 7)
    X[5] = means[1]
    X[9] = means[6]
    c = _components(X, means)
    mx, am, nbrute = c.dev.exact_max(np.arange(n))
    want_mx, want_am = co.kmeans_max_argmax(means, X)
    assert np.array_equal(am, want_am)
    assert np.array_equal(mx, want_mx)
    assert am[5] == 1 and am[9] == 6
    assert 1 <= nbrute < n // 4


def test_prefilter_wide_dynamic_range(gpu, monkeypatch):
    """Elements spanning nine decades (mostly flushed or subnormal in fp16), NaN-free: decisions stay exact."""
    from oracle import c_oracle as co
    monkeypatch.setenv("SEGK_SCORE_PRE", "1")
    rs = np.random.RandomState(98)
    n, D, K = 3000, 64, 200
    for scale in (1.0, 3e-4, 7e3):
        X = (rs.randn(n, D) * 10.0 ** rs.uniform(-9, 0, size=(n, D)) * scale).astype(np.float32)
        means = (rs.randn(K, D) * 10.0 ** rs.uniform(-9, 0, size=(K, D)) * scale).astype(np.float32)
        means[7] = means[3]
        X[11] = means[3]
        c = _components(X, means)
        mx, am, nbrute = c.dev.exact_max(np.arange(n))
        want_mx, want_am = co.kmeans_max_argmax(means, X)
        assert np.array_equal(am, want_am) and np.array_equal(mx, want_mx)
        assert am[11] == 3


@pytest.mark.parametrize("n,force", [(70000, True), (262144 + 1000, False), (262144 + 5000, False)])
def test_prefilter_large_launches(gpu, monkeypatch, n, force):
    """Above four rounds of the chip (262 144 rows on 256 CUs) the pre-filter is the default.  The three
    sizes take the 256-row workgroups alone (forced), whole rounds of 512-row workgroups + a queued
    remainder, and whole rounds + a second launch for the remainder; the contiguous-row (ids = NULL) form
    is what the sweeps use."""
    if force:
        monkeypatch.setenv("SEGK_SCORE_PRE", "1")
    import torch
    from oracle import c_oracle as co
    rs = np.random.RandomState(n % 1000)
    D, K = 100, 500
    mu = rs.randn(K // 2, D)
    X = mu[rs.randint(0, K // 2, n)] + 0.3 * rs.randn(n, D)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X = X.astype(np.float32)
    means = (mu[rs.randint(0, K // 2, K)] + 0.1 * rs.randn(K, D)).astype(np.float32)
    means /= np.linalg.norm(means, axis=1, keepdims=True)
    c = _components(X, means)
    c.dev.score_rows()
    torch.cuda.synchronize()
    nbrute = int(c.dev.cand_count.item())
    am = c.dev.cand_k.cpu().numpy()
    mx = c.dev.cand_s.cpu().numpy()
    want_mx, want_am = co.kmeans_max_argmax(means, X)
    assert np.array_equal(am, want_am)
    assert np.array_equal(mx, want_mx.astype(np.float64))
    assert nbrute < n // 50          # near-duplicate means: a third of the rows pass to the second stage, few beyond


def test_prefilter_row_lists_with_skipped_entries(gpu, monkeypatch):
    """ids with -1 entries (skipped) and an arbitrary order through the forced pre-filter path: the listed
    rows get the reference's max / argmax, rows that are not listed are not touched."""
    import torch
    from oracle import c_oracle as co
    from segmentalist_amd.device import to_dev
    monkeypatch.setenv("SEGK_SCORE_PRE", "1")
    rs = np.random.RandomState(21)
    n, D, K = 3000, 40, 129
    mu = rs.randn(K // 2, D)
    X = (mu[rs.randint(0, K // 2, n)] + 0.3 * rs.randn(n, D)).astype(np.float32)
    means = (mu[rs.randint(0, K // 2, K)] + 0.05 * rs.randn(K, D)).astype(np.float32)
    c = _components(X, means)
    ids = rs.permutation(n)[:1777].astype(np.int32)
    ids[::13] = -1
    c.dev.cand_k.fill_(-7)
    c.dev.cand_s.fill_(123.0)
    ids_t = to_dev(ids, np.int32)
    c.dev.score_rows(ids_t)
    torch.cuda.synchronize()
    am = c.dev.cand_k.cpu().numpy()
    mx = c.dev.cand_s.cpu().numpy()
    want_mx, want_am = co.kmeans_max_argmax(means, X)
    listed = np.zeros(n, dtype=bool)
    listed[ids[ids >= 0]] = True
    assert np.array_equal(am[listed], want_am[listed])
    assert np.array_equal(mx[listed], want_mx[listed].astype(np.float64))
    assert (am[~listed] == -7).all() and (mx[~listed] == 123.0).all()


def test_prefilter_randomized_sweep(gpu):
    """tools/stress_prefilter.py: 48 random (D, K, n, scale, tie structure) cases through the forced pre-filter
    path -- unit rows, nine-decade dynamic range, exact duplicate means, n = 1 ... 20 000 -- against the C oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_prefilter.py"), "7", "48"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout


@pytest.mark.parametrize("pre", ["0", "1"], ids=["fp16x2", "prefilter"])
def test_duplicate_means_are_taken_out_of_the_filters(gpu, monkeypatch, pre):
    """segk_kmeans_mark_duplicates: rows of `means` that repeat an earlier row exactly (-0.0 == +0.0 counts) stop
    being candidates of the filters -- the decisions stay the reference's (first maximum), but rows near a
    duplicated component no longer need the full scan."""
    import ctypes as C
    import torch
    from oracle import c_oracle as co
    from segmentalist_amd import _abi
    from segmentalist_amd.device import ptr
    monkeypatch.setenv("SEGK_SCORE_PRE", pre)
    rs = np.random.RandomState(77)
    n, D, K = 6000, 40, 300
    mu = rs.randn(K // 2, D)
    X = (mu[rs.randint(0, K // 2, n)] + 0.2 * rs.randn(n, D)).astype(np.float32)
    means = (mu[rs.randint(0, K // 2, K)] + 0.3 * rs.randn(K, D)).astype(np.float32)
    dup_of = {250: 3, 251: 3, 299: 40, 41: 40, 120: 7}
    for j, i in dup_of.items():
        means[j] = means[i]
    means[7, 5] = 0.0
    means[120, 5] = -0.0                                  # equal by value, different bits
    means[200] = means[9]
    means[200, 0] = np.nextafter(means[200, 0], np.float32(10))   # one ulp off: NOT a duplicate
    want_mx, want_am = co.kmeans_max_argmax(means, X)
    c = _components(X, means)
    _, _, nbrute_before = c.dev.exact_max(np.arange(n))
    n_marked = torch.zeros(1, dtype=torch.int32, device="cuda")
    _abi.check(_abi.lib().segk_kmeans_mark_duplicates(c.dev._ctx, c.dev._cp(), C.byref(c.dev.m), ptr(n_marked), _abi.stream()))
    torch.cuda.synchronize()
    assert int(n_marked.item()) == len(dup_of)
    mx, am, nbrute_after = c.dev.exact_max(np.arange(n))
    assert np.array_equal(am, want_am) and np.array_equal(mx, want_mx)
    assert not np.isin(am, list(dup_of)).any()            # a duplicate never wins
    assert nbrute_after <= nbrute_before // 2


@pytest.mark.parametrize("n_utt,D,K,N,nmax,ragged,mindur", [(300, 100, 1000, 20, 6, False, 0), (200, 40, 130, 0, 5, True, 0),
                                                             (150, 16, 9, 12, 8, False, 0), (120, 8, 300, 0, 3, True, 0),
                                                             (150, 8, 20, 0, 4, True, 9), (100, 12, 40, 0, 6, True, 14),
                                                             (120, 8, 1500, 10, 5, False, 0), (100, 8, 5000, 0, 4, True, 0),
                                                             (60, 128, 50, 32, 6, False, 0)])
def test_persistent_sequential_chain_equals_the_three_launch_form(gpu, monkeypatch, n_utt, D, K, N, nmax, ragged, mindur):
    """segk_seq_chain.hip (one persistent kernel per sweep: owner-computes components, one grid barrier per utterance, the DP
    replicated in every workgroup) against the three launches per utterance it replaces (SEGK_SEQ_CHAIN=0), which
    test_sequential_chain_bit_exact_vs_reference pins to the reference's captured chains: boundaries, labels, means,
    numerators, counts and the record values after every sweep, bit for bit -- headline shape, ragged utterances shorter than
    the window (no banded table), a window of eight, many components emptying (the stop / clean / relaunch path), spans
    shorter than min_duration (NaN durations, utterances.py:96-101: span ends whose candidates are all -inf, the backward
    pass's step-back branch, kmeans_acoustic_wordseg.py:516-530), K_max = 1 500 and 5 000 (sixteen and sixty-four components per
    workgroup: the score phase's lane groups of eight and thirty-two), and the largest shape the kernel takes (32 landmarks,
    128 dimensions)."""
    from segmentalist_amd import device as dev_mod, kmeans_acoustic_wordseg as kaw
    corpus = cases.chain_corpus(n_utt, D, K, 7 * n_utt + D, ragged, N, nmax, "float32")
    ran = []
    real = dev_mod.DeviceKMeans.sequential_sweep
    monkeypatch.setattr(dev_mod.DeviceKMeans, "sequential_sweep", lambda self, *a, **k: ran.append(real(self, *a, **k)) or ran[-1])
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SEGK_SEQ_CHAIN", mode)
        random.seed(5)
        np.random.seed(5)
        seg = kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5,
                                         init_am_assignments="rand", wip=-0.1, min_duration=mindur)
        if mindur:
            # the case is here for spans that HAVE an embedding and a NaN duration
            ut = seg.utterances
            assert (np.isnan(ut.durations) & (ut.vec_ids >= 0)).any()
        c = seg.acoustic_model.components
        states = []
        for it in range(3):
            rec = seg.segment(1)
            states.append((seg.utterances.boundaries.copy(), c.assignments.copy(), c.means.copy(), c.mean_numerators.copy(),
                           c.counts.copy(), c.K, rec["sum_neg_len_sqrd_norm"][0], rec["n_tokens"][0]))
        out[mode] = states
        if mode == "1":
            assert ran and all(ran), "the persistent kernel did not take the sweeps"
        ran.clear()
    for it in range(3):
        a, b = out["0"][it], out["1"][it]
        for x, y in zip(a[:5], b[:5]):
            assert np.array_equal(x, y), it
        assert a[5:] == b[5:], it


def test_sequential_sweep_with_a_repeated_utterance(gpu, monkeypatch):
    """segk_kmeans_sequential_sweep with an utterance listed twice in `order` (adjacent and apart): the persistent kernel
    prefetches utterance order[q + 1] while order[q] is being updated, so such an order is routed to the launches per
    utterance -- same state as SEGK_SEQ_CHAIN=0, and as visiting the utterances one call at a time."""
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    corpus = cases.chain_corpus(60, 16, 12, 977, False, 12, 6, "float32")
    order = [5, 5, 9, 3, 17, 9, 40, 41, 5, 2]
    out = []
    for mode in ("default", "0", "one_by_one"):
        monkeypatch.delenv("SEGK_SEQ_CHAIN", raising=False)
        if mode == "0":
            monkeypatch.setenv("SEGK_SEQ_CHAIN", "0")
        random.seed(5)
        np.random.seed(5)
        seg = kaw.SegmentalKMeansWordseg(12, *corpus, n_slices_min=0, n_slices_max=6, p_boundary_init=0.5,
                                         init_am_assignments="rand", wip=-0.1)
        c = seg.acoustic_model.components
        if mode == "one_by_one":
            for i in order:
                seg.segment_i(i)
        else:
            assert seg._dk.sequential_sweep(seg._dev_bounds, order, 0, 6, -0.1)
            seg.utterances.mark_device_dirty()
            gpu.cuda.synchronize()
            seg._dk.check_status()
        out.append((seg.utterances.boundaries.copy(), c.assignments.copy(), c.means.copy(), c.mean_numerators.copy(),
                    c.counts.copy(), c.K))
    for other in out[1:]:
        for x, y in zip(out[0][:5], other[:5]):
            assert np.array_equal(x, y)
        assert out[0][5] == other[5]


@pytest.mark.parametrize("n_utt,D,K,N,nmax,ragged", [(400, 16, 40, 20, 6, False), (300, 8, 25, 0, 8, True), (200, 12, 30, 0, 3, True),
                                                     (150, 8, 20, 44, 8, False), (120, 8, 20, 64, 5, False)])
def test_segment_kernels_agree(gpu, monkeypatch, n_utt, D, K, N, nmax, ragged):
    """The per-utterance DP by the whole wave (seg_w8_wave: eight lanes per step, DPP maxima, token lists from ballots; the
    default below 4 096 utterances), with eight utterances per wave (k_kmeans_segment_oct, the default above; forced here with
    SEGK_SEGMENT_OCT=1) and the generic kernel (SEGK_SEGMENT_GENERIC=1): identical boundaries, labels and statistics after
    three batch sweeps -- uniform utterances, ragged ones shorter than the window, a window of eight, 44 and 64 landmarks
    (boundary masks beyond 32 bits, several steps of eight bits in the token lists)."""
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    corpus = cases.chain_corpus(n_utt, D, K, 31 * n_utt + D, ragged, N, nmax, "float32")
    out = []
    for env in ({"SEGK_SEGMENT_OCT": "0"}, {"SEGK_SEGMENT_OCT": "1"}, {"SEGK_SEGMENT_GENERIC": "1"}):
        for k in ("SEGK_SEGMENT_OCT", "SEGK_SEGMENT_GENERIC"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        random.seed(9)
        np.random.seed(9)
        seg = kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5,
                                         init_am_assignments="rand", wip=-0.2, sync="batch")
        recs = []
        for _ in range(3):
            recs.append(seg.segment(1)["sum_neg_len_sqrd_norm"][0])
        c = seg.acoustic_model.components
        out.append((seg.utterances.boundaries.copy(), c.assignments.copy(), c.means.copy(), c.counts.copy(), recs))
    for other in out[1:]:
        for a, b in zip(out[0][:4], other[:4]):
            assert np.array_equal(a, b)
        assert out[0][4] == other[4]


def test_two_models_on_one_context_with_interleaved_sweeps(gpu):
    """Two segmenters alive on the process's one library context (its workspaces, queues, hint buffers and feedback words are
    shared), whole-sweep batches on one and mini-batches on the other, both at sizes where the hinted score path runs, sweeps
    interleaved: each ends in the state it reaches alone."""
    import torch
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus

    def build(seed, n_utt, K, n_batches):
        corpus = make_corpus(n_utt, 100, K, seed=seed, N=20, n_slices_max=6)
        random.seed(seed); np.random.seed(seed)
        return kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch", n_batches=n_batches)

    def state(seg):
        torch.cuda.synchronize()
        seg._dk.check_status()
        c = seg.acoustic_model.components
        return seg.utterances.boundaries.copy(), c.assignments.copy(), c.means.copy(), c.K

    specs = ((1, 3000, 1000, 1), (2, 2500, 700, 2))
    alone = []
    for sp in specs:
        s = build(*sp)
        for _ in range(6):
            s.batch_sweep_async()
        alone.append(state(s))
        del s
    segs = [build(*sp) for sp in specs]
    for _ in range(6):
        for s in segs:
            s.batch_sweep_async()
    for i, s in enumerate(segs):
        st = state(s)
        for a, b in zip(st[:3], alone[i][:3]):
            assert np.array_equal(a, b), i
        assert st[3] == alone[i][3]
