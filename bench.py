#!/usr/bin/env python3
"""
bench.py -- Gibbs/segmental-k-means sweeps per second on BASELINE.json config 3:
SegmentalKMeansWordseg, 10 000 utterances x 20 landmarks (n_slices_max = 6 -> 105 candidate
spans each, 1.05 M embeddings), D = 100, K = 1000, synthetic unit-norm float32 embeddings
(SURVEY.md 8(d)), batch-synchronous sweeps, sharded over N GPUs (strong scaling:
the corpus is fixed, each rank owns 1/N of the utterances; one small RCCL all-gather per
sweep).  `--windows` (default 9) back-to-back windows of exactly `--steps` sweeps are timed, each
bracketed by barrier + synchronize; the line reports the median window and the spread.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fp32 MFMA score
kernel), its duration measured live with HIP events on the launch stream inside the timed
region; `cpu_baseline` times the oracle's faithful single-core restatement of the reference
on a bounded sample of the same corpus (rank 0, N = 1 only).
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

METRIC = "Gibbs sweeps/sec (10k utts, D=100, K=1000) at 1/2/4/8 MI355X"
PEAK_FP32_MATRIX_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MATRIX_TFLOPS = 2516.6    # v_mfma_f32_32x32x16_bf16: 256 CUs x 4 SIMDs x 512 MAC/clk x 2 x 2.4 GHz ("~2.5 PF dense")


def cpu_blas_courtesy(corpus, n_utts_total, K, n_slices_max, budget_utts):
    """NOT the reference: a BLAS-vectorised CPU version of the same batch sweep (one sgemm scores every
    embedding of the sample against every mean via |x|^2 - 2 x.m + |m|^2, the C oracle's Viterbi per utterance,
    vectorised statistics), all host cores through the BLAS threads -- the courtesy upper bound BASELINE.md
    section 4 asks for, so that the GPU number is not only read against a single-threaded Python loop."""
    from oracle import c_oracle as co
    emb, vid, dur, lm = corpus
    keys = sorted(emb)[:budget_utts]
    X = np.concatenate([emb[k] for k in keys]).astype(np.float32)
    rs = np.random.RandomState(0)
    means = X[rs.choice(X.shape[0], K, replace=False)].copy()
    offs = np.cumsum([0] + [emb[k].shape[0] for k in keys])
    # per-utterance tables prepared once, as the reference's constructor does (global row of every span, durations)
    vids = [np.asarray(vid[k]) for k in keys]
    gids = [np.where(v >= 0, offs[u] + np.maximum(v, 0), 0) for u, v in enumerate(vids)]
    durs = [np.where(v >= 0, np.asarray(dur[k], dtype=np.float64), 1.0) for v, k in zip(vids, keys)]
    masks = [v < 0 for v in vids]
    Ns = [len(lm[k]) for k in keys]
    from oracle.c_oracle import lib as _olib, _d, _bp
    L = _olib()
    t0 = time.perf_counter()
    for _ in range(2):                                   # two sweeps: scoring, DP, means update
        xx = np.einsum("ij,ij->i", X, X)
        mm = np.einsum("ij,ij->i", means, means)
        sc = X @ means.T
        sc *= 2.0
        sc -= mm[None, :]
        arg = sc.argmax(axis=1)
        best = (sc[np.arange(sc.shape[0]), arg] - xx).astype(np.float64)
        tok_rows = []
        b = np.zeros(max(Ns), np.uint8)
        g = np.empty(max(Ns), np.float64)
        for u in range(len(keys)):
            vec = best[gids[u]] * durs[u]
            vec[masks[u]] = -np.inf
            N = Ns[u]
            L.orc_fb_kmeans_viterbi(_d(vec), N, 0, n_slices_max, b.ctypes.data_as(_bp), _d(g))
            ends = np.flatnonzero(b[:N]) + 1
            starts = np.concatenate([[0], ends[:-1]])
            tok_rows.append(gids[u][ends * (ends - 1) // 2 + starts])
        rows = np.concatenate(tok_rows)
        ks = arg[rows]
        order = np.argsort(ks, kind="stable")
        cnts = np.bincount(ks, minlength=K)
        sums = np.add.reduceat(X[rows[order]].astype(np.float64), np.concatenate([[0], np.cumsum(cnts)[:-1]])[cnts > 0], axis=0)
        nz = cnts > 0
        means[nz] = (sums / cnts[nz, None]).astype(np.float32)
    dt = (time.perf_counter() - t0) / 2
    return {"value": 1.0 / (dt / len(keys) * n_utts_total), "unit": "sweeps/s", "cores": os.cpu_count(),
            "note": "NOT the reference: BLAS-vectorised CPU batch sweep (sgemm scores + C Viterbi + vectorised statistics), "
                    "%d utterances, %.3f s per sweep = %.3f ms/utterance, extrapolated linearly; a courtesy upper bound for "
                    "a CPU, different arithmetic than the reference's" % (len(keys), dt, 1e3 * dt / len(keys))}


def cpu_baseline(corpus, n_utts_total, K, n_slices_max, budget_utts):
    """Oracle (port of the reference's per-embedding numpy path) on the first `budget_utts`
    utterances of the same corpus; extrapolated linearly to the whole corpus (the cost per
    utterance is size independent: BASELINE.md section 2)."""
    from oracle import np_oracle as no
    emb, vid, dur, lm = corpus
    keys = sorted(emb)[:budget_utts]
    sub = tuple({k: d[k] for k in keys} for d in (emb, vid, dur, lm))
    random.seed(0)
    np.random.seed(0)
    seg = no.SegmentalKMeansWordseg(K, *sub, n_slices_max=n_slices_max, init_am_assignments="spread")
    t0 = time.perf_counter()
    order = list(range(seg.utterances.D))
    random.shuffle(order)
    for i in order:
        seg.segment_i(i)
    dt = time.perf_counter() - t0
    per_utt = dt / len(keys)
    out = {
        "value": 1.0 / (per_utt * n_utts_total),
        "unit": "sweeps/s",
        "cores": 1,
        "kind": "port",
        "sample": "oracle/np_oracle.py SegmentalKMeansWordseg.segment_i (per-embedding numpy K x D scoring + "
                  "Viterbi DP + sequential mean updates, the reference's structure) on the first %d of %d "
                  "utterances, %.2f s wall = %.2f ms/utterance, extrapolated linearly to the full sweep; "
                  "host has %d cores, the reference path is single-threaded"
                  % (len(keys), n_utts_total, dt, 1e3 * per_utt, os.cpu_count()),
    }
    cal = os.path.join(ROOT, "profiles", "cpu_calibration.json")
    if os.path.exists(cal):
        cj = json.load(open(cal))
        out["calibration"] = {
            "port_over_reference": cj["ratio_port_over_reference"],
            "note": "ms/utterance of this port / ms/utterance of the py3 translation of the reference itself, same inputs "
                    "and seeds, identical resulting state, measured in the build container (tools/cpu_calibration.py: "
                    "%.2f vs %.2f ms/utterance on %d utterances); the reference cannot travel to the GPU box"
                    % (cj["oracle_port_ms_per_utt"], cj["reference_py3_translation_ms_per_utt"], cj["n_utts"])}
    try:
        out["blas_courtesy"] = cpu_blas_courtesy(corpus, n_utts_total, K, n_slices_max, min(4 * budget_utts, n_utts_total))
    except Exception as e:                               # the courtesy number must never cost the line
        out["blas_courtesy"] = {"error": repr(e)}
    return out


def self_launch(gpus):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks as a CHILD
    `python -m torch.distributed.run` job BEFORE this process touches a GPU (never an exec after GPU initialisation), relay
    the ranks' output -- rank 0 prints the one JSON line -- and leave with the child's exit code.  The torchrun form the
    driver uses for N > 1 sets WORLD_SIZE and never comes here.  Backend: RCCL ("nccl") when the box has a GPU per rank;
    with fewer GPUs than ranks (a rehearsal on a one-GPU box) gloo with the ranks sharing the cards, unless
    SEGK_BENCH_BACKEND says otherwise."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "SEGK_BENCH_BACKEND" not in env:
        # GPUs of the box counted by a short-lived child that is gone before the ranks start: this process never opens the GPU
        try:
            n_dev = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], env=env,
                                       stdout=subprocess.PIPE, text=True, timeout=300).stdout.strip().splitlines()[-1])
        except Exception:
            n_dev = gpus
        if n_dev < gpus:
            env["SEGK_BENCH_BACKEND"] = "gloo"
    with socket.socket() as so:                        # a free rendezvous port
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:                           # rank 0's JSON line to stdout; anything else the ranks print (gloo's
        dst = sys.stdout if line.startswith("{") else sys.stderr      # connection chatter) to stderr
        dst.write(line)
        dst.flush()
    sys.exit(proc.wait())


def early_sweeps(kaw, corpus, args, n_sweeps=10):
    """Sweeps 1..n_sweeps of a FRESH chain (the sweeps in which a k-means run does its work: 57 % of the rows change their
    component in sweep 1, 16 % in sweep 2, 6-10 % in sweeps 3-10), each timed with HIP events on the launch stream, once on
    the default (hinted) score path and once with SEGK_SCORE_HINT=0 (the library reads the switch at every call); outside the
    main timed region.  second_stage_rows / full_scan_rows: what the score stages of that sweep passed on."""
    import ctypes as C
    import torch
    from segmentalist_amd import _abi
    out = {}
    saved = os.environ.get("SEGK_SCORE_HINT")
    try:
        for label, env in (("hinted", None), ("unhinted", "0")):
            if env is None:
                os.environ.pop("SEGK_SCORE_HINT", None)
                if saved is not None:
                    os.environ["SEGK_SCORE_HINT"] = saved
            else:
                os.environ["SEGK_SCORE_HINT"] = env
            random.seed(0)
            np.random.seed(0)
            seg = kaw.SegmentalKMeansWordseg(args.K, *corpus, n_slices_max=args.n_slices_max,
                                             init_am_assignments="spread", sync="batch")
            seg._get_sweeper()          # (as the main run does before its warm-up: the sweeper's tables are construction, not sweep 1)
            ms, second, full, comps = [], [], [], []
            sc = (C.c_int32 * 2)()
            for _ in range(n_sweeps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                seg.batch_sweep_async()
                e1.record()
                torch.cuda.synchronize()
                ms.append(float(e0.elapsed_time(e1)))
                _abi.check(_abi.lib().segk_kmeans_stage_counts(_abi.ctx(), C.byref(seg._dk.cand), sc, _abi.stream()))
                second.append(int(sc[0]))
                full.append(int(sc[1]))
                comps.append(int(seg._dk.K.item()))
            seg._dk.check_status()
            out[label] = {"ms": ms, "second_stage_rows": second, "full_scan_rows": full, "components": comps}
            del seg
    finally:
        if saved is None:
            os.environ.pop("SEGK_SCORE_HINT", None)
        else:
            os.environ["SEGK_SCORE_HINT"] = saved
    out["note"] = ("sweeps 1..%d of a fresh chain (spread initialisation), one HIP-event pair around each whole sweep with the "
                   "stream drained in between (so ~10 us of launch latency per sweep that back-to-back sweeps hide); sweep 1 has "
                   "no hints on either path" % n_sweeps)
    return out


def minibatch_rate(kaw, corpus, args, n_batches, n_sweeps):
    """sweeps/s of the mini-batch form (n_batches statistics refreshes per sweep) on the same corpus:
    5 warm sweeps, then n_sweeps timed between two synchronisations; outside the main timed region."""
    import torch
    random.seed(0)
    np.random.seed(0)
    seg = kaw.SegmentalKMeansWordseg(args.K, *corpus, n_slices_max=args.n_slices_max, init_am_assignments="spread",
                                     sync="batch", n_batches=n_batches)
    for _ in range(5):
        seg.batch_sweep_async()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_sweeps):
        seg.batch_sweep_async()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    seg._dk.check_status()
    return {"n_batches": n_batches, "sweeps_per_s": n_sweeps / dt, "ms_per_sweep": 1e3 * dt / n_sweeps,
            "components_after": int(seg._dk.K.item()), "sweeps": 5 + n_sweeps}


def main_fbgmm(args):
    """Secondary workloads (BASELINE.json configs[1] and configs[4]): sweeps/s of the batch-synchronous
    blocked Gibbs sampler of the FBGMM / bigram drivers, same JSON contract."""
    import ctypes as C
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("SEGK_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    from segmentalist_amd import _abi, bigram_acoustic_wordseg as baw, fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    bigram = args.workload == "bigram_c5"
    n_utt, D, K = (10000, 100, 1000) if bigram else (1000, 39, 100)
    corpus = make_corpus(n_utt, D, K, seed=0, N=args.landmarks, n_slices_max=args.n_slices_max)
    random.seed(0)
    np.random.seed(0)
    kw = dict(n_slices_min=0, n_slices_max=args.n_slices_max, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
              init_am_assignments="rand", time_power_term=1.0, sync="batch", n_gibbs_blocks=8, n_stat_blocks=8)
    if bigram:
        prior = FixedVarPrior(0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D))
        seg = baw.BigramAcousticWordseg(K, prior, {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5}, *corpus,
                                        covariance_type="fixed", fb_type="unigram", score_precision=os.environ.get("SEGK_FBB_PRECISION", "f16"), **kw)
    else:
        prior = NIW(np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D))
        # score precision of the diagonal Student-t span score: f32 (float32 terms with v_log_f32; within the 1e-4
        # contract of the path, tests/test_gpu_fbgmm_batch.py) unless SEGK_FBB_PRECISION=f64 asks for the fp64 kernel
        dprec = os.environ.get("SEGK_FBB_PRECISION", "f32")
        seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, prior, *corpus, covariance_type="diag", fb_type="standard",
                                         score_precision="f32" if dprec in ("f32", "f16") else "f64", **kw)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        seg.batch_sweep_async()
    barrier()
    seg._df.check_status()
    diag32 = (not bigram) and getattr(seg._get_sweeper(), "score_diag32", False)
    if bigram or diag32:
        _abi.check(_abi.lib().segk_profile_enable(_abi.ctx(), 1))
    win = []
    for _ in range(max(1, args.windows)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            seg.batch_sweep_async()
        barrier()
        win.append(time.perf_counter() - t0)
    seg._df.check_status()
    if world > 1:
        t = torch.tensor(win, dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        win = [float(v) for v in t.cpu()]
    elapsed = float(np.median(win))
    if rank == 0:
        sw = seg._get_sweeper()
        cnt, tot, occ = sw.totals()
        out = {
            "metric": "Gibbs sweeps/sec (%s)" % ("BigramAcousticWordseg, 10k utts, D=100, K=1000" if bigram
                                                  else "UnigramAcousticWordseg + FBGMM diag, 1k utts, D=39, K=100"),
            "value": args.steps / elapsed, "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "windows": len(win),
            "window_ms_per_step": {"min": 1e3 * min(win) / args.steps, "median": 1e3 * elapsed / args.steps,
                                   "max": 1e3 * max(win) / args.steps},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("f32 span scores and token likelihoods, f64 sampling" if diag32 else
                      "f32 span scores, f64 sampling" if bigram else "f64"), "data": "synthetic",
            "config": {"workload": "%s batch-synchronous blocked Gibbs sweep, 8 blocks (BASELINE.json configs[%d])"
                                   % ("BigramAcousticWordseg" if bigram else "UnigramAcousticWordseg + FBGMM (diag)",
                                      4 if bigram else 1),
                       "utterances": n_utt, "landmarks_per_utt": args.landmarks, "n_slices_max": args.n_slices_max,
                       "embeddings": int(seg._corpus.n_emb), "D": D, "K": K,
                       "parallelism": "utterance slices x%d, one all-gather of partial sums per Gibbs step" % world,
                       "components_after": occ, "tokens": int(tot)},
        }
        if bigram:
            nmax = 256
            ms = (C.c_float * nmax)()
            rows = (C.c_int64 * nmax)()
            got = _abi.lib().segk_profile_read(_abi.ctx(), ms, rows, nmax)
            if got > 0:
                score_ms = float(np.mean(ms[:got]))
                n_rows = float(np.mean(rows[:got]))
                # [x^2, x] . [-pp/2, pp*mu] over the occupied slots + the empty-slot row (the reference scores the components
                # that exist, fbgmm.py:256-285); the fp16x2 image holds just those, packed into the leading tiles
                # (segk.h: segk_fbatch.consts16), as of the last Gibbs step of the timed region
                cols, tiles_used = occ + 1, (K + 1 + 31) // 32
                if getattr(sw, "score_f16", False) and sw.consts16 is not None:
                    import torch as _t
                    cm = sw.consts16[K + 2:].view(_t.int32)[:K + 2].cpu()
                    cols, tiles_used = int(cm[K]) + 1, int(cm[K + 1])
                flops = 2.0 * n_rows * cols * (2 * D)
                achieved = flops / (score_ms * 1e-3) / 1e12
                if getattr(sw, "score_f16", False):
                    kp, kpad = (2 * D + 15) // 16 * 16, 32 * tiles_used
                    executed = 3 * 2.0 * n_rows * kpad * kp
                    ex_tf = executed / (score_ms * 1e-3) / 1e12
                    out["dtype"] = "fp16x2 span scores, f64 sampling"
                    out["roofline"] = {"bound": "mfma", "kernel": "k_kmeans_score_sp<%d, 4, 2, 1> (log-sum-exp mode on two-way fp16 "
                                       "splits, one launch per Gibbs step)" % (kp // 16), "achieved": achieved,
                                       "peak": PEAK_BF16_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_MATRIX_TFLOPS,
                                       "traffic": None, "ms_per_launch": score_ms, "flops_per_launch": flops,
                                       "columns": cols, "tiles_of_32": tiles_used,
                                       "executed_flops_per_launch": executed, "executed_achieved": ex_tf,
                                       "executed_frac": ex_tf / PEAK_BF16_MATRIX_TFLOPS,
                                       "achieved_over_fp32_matrix_peak": achieved / PEAK_FP32_MATRIX_TFLOPS}
                else:
                    out["roofline"] = {"bound": "mfma", "kernel": "k_kmeans_score<50, 1, 4, 0, 1> (log-sum-exp mode, one launch per "
                                       "Gibbs step)", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                                       "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": None, "ms_per_launch": score_ms,
                                       "flops_per_launch": flops}
            _abi.check(_abi.lib().segk_profile_enable(_abi.ctx(), 0))
        if diag32:
            # dominant kernel: k_fbb_score_diag32, one launch per Gibbs step.  Algorithmic work = one Student-t term
            # log(1 + delta^2 q) per (row, slot, dimension): rows x K_max x D terms per launch; the roofline is what the
            # vector ALUs sustain on exactly that term with operands in registers (segk_calibrate_vlog, measured here)
            nmax = 256
            ms = (C.c_float * nmax)()
            rows = (C.c_int64 * nmax)()
            got = _abi.lib().segk_profile_read(_abi.ctx(), ms, rows, nmax)
            _abi.check(_abi.lib().segk_profile_enable(_abi.ctx(), 0))
            peak = C.c_double(0.0)
            _abi.check(_abi.lib().segk_calibrate_vlog(_abi.ctx(), C.byref(peak), _abi.stream()))
            if got > 0:
                score_ms = float(np.mean(ms[:got]))
                fused = getattr(sw, "_fused", None) is True
                # (the fused step's events count utterances; its terms: every row of the block against the OCCUPIED slots --
                # an empty slot's logit needs no term -- as of the final state)
                n_rows = float(seg._corpus.n_emb) / sw.B if fused else float(np.mean(rows[:got]))
                terms = n_rows * (occ if fused else K) * D
                achieved = terms / (score_ms * 1e-3) / 1e9
                out["roofline"] = {"bound": "valu", "kernel": ("k_fbb_step_diag32 (one launch per Gibbs step: float32 Student-t terms with "
                                   "v_log_f32 of %d rows x %d occupied slots x %d dimensions, then -- the same workgroups -- span "
                                   "scores, the sampling DP and the draws, which are most of its time)" % (int(n_rows), occ, D)) if fused else
                                   ("k_fbb_score_diag32 (float32 Student-t terms with v_log_f32, one launch per "
                                    "Gibbs step: %d rows x %d slots x %d dimensions)" % (int(n_rows), K, D)),
                                   "achieved": achieved, "peak": peak.value / 1e9, "unit": "Gterm/s", "frac": achieved / (peak.value / 1e9),
                                   "traffic": None, "ms_per_launch": score_ms, "terms_per_launch": terms,
                                   "peak_source": "k_vlog_calibrate measured in this run: the kernel's inner term (v_sub, v_mul, v_fma, "
                                                  "v_log_f32, v_add) from registers on all %d CUs" % torch.cuda.get_device_properties(0).multi_processor_count}
        if world == 1 and args.cpu_utts > 0:
            from oracle import np_oracle as no
            n_cpu = min(args.cpu_utts, 300 if bigram else 1000) // (10 if bigram else 1) or 1
            keys = sorted(corpus[0])[:n_cpu]
            sub = tuple({k: d[k] for k in keys} for d in corpus)
            random.seed(0)
            np.random.seed(0)
            okw = dict(kw)
            for k in ("sync", "n_gibbs_blocks", "n_stat_blocks"):
                okw.pop(k)
            if bigram:
                ref = no.BigramAcousticWordseg(K, no.FixedVarPrior(0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D)),
                                               {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5}, *sub,
                                               covariance_type="fixed", fb_type="unigram", **okw)
            else:
                ref = no.UnigramAcousticWordseg(no.FBGMM, 1.0, K, no.NIW(np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D)),
                                                *sub, covariance_type="diag", fb_type="standard", **okw)
            t0 = time.perf_counter()
            for i in range(len(keys)):
                ref.gibbs_sample_i(i)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": len(keys) / dt / n_utt, "unit": "sweeps/s", "cores": 1, "kind": "port",
                                   "sample": "oracle/np_oracle.py gibbs_sample_i (the reference's serial chain) on the first %d "
                                             "of %d utterances, %.2f s wall = %.2f ms/utterance, extrapolated linearly"
                                             % (len(keys), n_utt, dt, 1e3 * dt / len(keys))}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main_sequential(args):
    """The reference's sequential chain (every utterance sees the means the previous one left; bit-identical to the
    reference, tests/test_gpu_kmeans.py) on the headline corpus: sweeps per second on ONE GPU -- the chain does not shard."""
    import torch
    assert args.gpus == 1, "the sequential chain is one Markov chain: one GPU"
    torch.cuda.set_device(0)
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    steps, warmup = args.steps, args.warmup
    corpus = make_corpus(args.utts, args.dim, args.K, seed=0, N=args.landmarks, n_slices_max=args.n_slices_max)
    random.seed(0)
    np.random.seed(0)
    seg = kaw.SegmentalKMeansWordseg(args.K, *corpus, n_slices_max=args.n_slices_max, init_am_assignments="spread")
    for _ in range(warmup):
        seg.segment(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = seg.segment(steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    print(json.dumps({
        "metric": "Gibbs sweeps/sec (sequential reference chain, 10k utts, D=100, K=1000)",
        "value": steps / elapsed, "unit": "sweeps/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps, "us_per_utterance": 1e6 * elapsed / steps / args.utts,
        "higher_is_better": True, "scaling": "none (one chain)", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "SegmentalKMeansWordseg sync='sequential' (the reference's chain, bit-identical state), "
                               "one persistent kernel per stretch of utterances between two emptied components",
                   "utterances": args.utts, "landmarks_per_utt": args.landmarks, "n_slices_max": args.n_slices_max,
                   "D": args.dim, "K": args.K, "components_after": int(seg.acoustic_model.components.K),
                   "sweep_seconds_including_record_keeping": [float(t) for t in rec["sample_time"]]},
        "note": "`value` is wall time over the whole segment() call, record keeping of every sweep included; "
                "record['sample_time'] (the reference's own bracket: the per-utterance loop only) is listed in config",
    }))


def main_fbgmm_sequential(args, bigram=False):
    """configs[1] through the reference's own serial Gibbs chain (UnigramAcousticWordseg + FBGMM with diagonal components, the API
    default sync='sequential'; draws identical to the reference, tests/test_gpu_chain_parity_fullshape.py): sweeps per second
    on ONE GPU -- the chain does not shard.  One persistent kernel per stretch of utterances between two emptied components
    (segk_fbgmm_sequential_sweep).  bigram: the same corpus through BigramAcousticWordseg (fixed-variance components, bigram
    language model over the component labels), the kernel's language-model variant."""
    import torch
    assert args.gpus == 1, "the sequential chain is one Markov chain: one GPU"
    torch.cuda.set_device(0)
    from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    n_utt, D, K = 1000, 39, 100
    corpus = make_corpus(n_utt, D, K, seed=0, N=args.landmarks, n_slices_max=args.n_slices_max)
    random.seed(0)
    np.random.seed(0)
    common = dict(n_slices_min=0, n_slices_max=args.n_slices_max, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
                  init_am_assignments="rand", time_power_term=1.0)
    if bigram:
        seg = baw.BigramAcousticWordseg(K, FixedVarPrior(0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D)),
                                        {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5}, *corpus,
                                        covariance_type="fixed", fb_type="unigram", **common)
    else:
        seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, NIW(np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D)), *corpus,
                                         covariance_type="diag", fb_type="standard", **common)
    for _ in range(args.warmup):
        seg.gibbs_sample(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = seg.gibbs_sample(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t_loop = float(np.mean(rec["sample_time"]))
    print(json.dumps({
        "metric": "Gibbs sweeps/sec (sequential reference chain, %s, 1k utts, D=39, K=100)"
                  % ("BigramAcousticWordseg fixed-variance" if bigram else "UnigramAcousticWordseg + FBGMM diag"),
        "value": args.steps / elapsed, "unit": "sweeps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "us_per_utterance": 1e6 * t_loop / n_utt,
        "higher_is_better": True, "scaling": "none (one chain)", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("BigramAcousticWordseg sync='sequential' (fixed-variance components, bigram language model; the "
                                "configs[1] corpus through the reference's serial chain), the persistent kernel's language-model "
                                "variant" if bigram else
                                "UnigramAcousticWordseg sync='sequential' + FBGMM diag (BASELINE.json configs[1] through the "
                                "reference's serial chain), one persistent kernel per stretch of utterances between two emptied "
                                "components"),
                   "utterances": n_utt, "landmarks_per_utt": args.landmarks, "n_slices_max": args.n_slices_max, "D": D, "K": K,
                   "components_after": int(rec["components"][-1]),
                   "sample_time_seconds": [float(t) for t in rec["sample_time"]]},
        "note": "`value` is wall time over the whole gibbs_sample() call, record metrics of every sweep included; "
                "us_per_utterance from record['sample_time'] (the reference's own bracket: the per-utterance loop)",
    }))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed sweeps per window (default 50; kmeans_c3_sequential: 3)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed sweeps first (default 5; kmeans_c3_sequential: 1)")
    ap.add_argument("--utts", type=int, default=10000)
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--K", type=int, default=1000)
    ap.add_argument("--landmarks", type=int, default=20)
    ap.add_argument("--n-slices-max", type=int, default=6)
    ap.add_argument("--windows", type=int, default=9, help="timed windows of --steps sweeps each (at least; more are added "
                    "until --min-seconds of timed sweeps have run); the median window is reported")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="lower bound of the total timed region")
    ap.add_argument("--no-seq-chain", action="store_true", help="skip the sequential_chain extra key (N = 1)")
    ap.add_argument("--no-early", action="store_true", help="skip the early_sweeps and minibatch extra keys (N = 1)")
    ap.add_argument("--cpu-utts", type=int, default=2000, help="utterances timed for cpu_baseline (0 = skip)")
    ap.add_argument("--no-events", action="store_true", help="do not bracket the score kernel with events")
    ap.add_argument("--event-period", type=int, default=8,
                    help="bracket the score kernel of every Nth sweep of the timed region with HIP events (roofline.achieved)")
    ap.add_argument("--workload", default="kmeans_c3", choices=["kmeans_c3", "fbgmm_diag_c2", "bigram_c5", "kmeans_c3_sequential", "fbgmm_c2_sequential",
                             "bigram_c2_sequential"],
                    help="kmeans_c3 (default) is the headline of BASELINE.json (configs[2]); fbgmm_diag_c2 = configs[1] "
                         "(UnigramAcousticWordseg + FBGMM diag, 1k utterances, D=39, K=100), bigram_c5 = configs[4] "
                         "(BigramAcousticWordseg, 10k utterances, D=100, K=1000): the batch (blocked Gibbs) sampler; "
                         "kmeans_c3_sequential = the headline corpus through the reference's own sequential chain "
                         "(sync='sequential', the API default; one GPU; --steps sweeps after --warmup, default 3 after 1); "
                         "fbgmm_c2_sequential / bigram_c2_sequential = the configs[1] corpus through the serial Gibbs chain of "
                         "UnigramAcousticWordseg + FBGMM / of BigramAcousticWordseg (one persistent kernel per stretch)")
    args = ap.parse_args()
    seq = args.workload in ("kmeans_c3_sequential", "fbgmm_c2_sequential", "bigram_c2_sequential")
    if args.steps is None:
        args.steps = 3 if seq else 50
    if args.warmup is None:
        args.warmup = 1 if seq else 5
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not seq:
        self_launch(args.gpus)                           # does not return
    if args.workload in ("fbgmm_c2_sequential", "bigram_c2_sequential"):
        return main_fbgmm_sequential(args, bigram=args.workload == "bigram_c2_sequential")
    if seq:
        return main_sequential(args)
    if args.workload != "kmeans_c3":
        return main_fbgmm(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if world > 1:
        import torch.distributed as dist
        # RCCL over xGMI; SEGK_BENCH_BACKEND=gloo exists only to rehearse the N > 1 code path with
        # several ranks sharing one GPU (RCCL refuses two ranks on one device)
        backend = os.environ.get("SEGK_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus

    corpus = make_corpus(args.utts, args.dim, args.K, seed=0, N=args.landmarks, n_slices_max=args.n_slices_max)
    random.seed(0)
    np.random.seed(0)
    seg = kaw.SegmentalKMeansWordseg(args.K, *corpus, n_slices_max=args.n_slices_max,
                                     init_am_assignments="spread", sync="batch")
    sweeper = seg._get_sweeper()
    n_emb = seg.acoustic_model.components.N          # (seg._corpus holds this rank's shard of the rows when world > 1)
    rows_local = sweeper.part.row_hi - sweeper.part.row_lo

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        seg.batch_sweep_async()
    barrier()
    seg._dk.check_status()

    # HIP events around the main launch of the score kernel, recorded by the library on the launch
    # stream (segk_profile_enable / segk_profile_read, include/segk.h)
    import ctypes as C
    from segmentalist_amd import _abi
    use_ev = not args.no_events
    # a hipGraph capture cannot contain the event records: with SEGK_SWEEP_GRAPH=1 the timed windows replay the graph
    # and the score kernel is timed afterwards over `steps` further sweeps launched plainly
    ev_after = use_ev and sweeper.use_graph
    if use_ev and not ev_after:
        # every `--event-period`-th sweep of the timed region has its score kernel bracketed by events (an event record between
        # two kernels costs the stream ~3 us of bubble: around every launch that was 6.5 us, 1.2 %, of the sweep it measures)
        _abi.check(_abi.lib().segk_profile_enable(_abi.ctx(), max(1, args.event_period)))
    # `windows` back-to-back timed windows of EXACTLY `steps` sweeps each, every one bracketed by barrier +
    # synchronize on both sides and reduced with MAX over the ranks; the line reports the MEDIAN window (a 13 ms
    # window moves by percent with one clock ramp; the spread is reported beside it)
    win = []
    n_windows = max(1, args.windows)
    w = 0
    while w < n_windows:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            seg.batch_sweep_async()
        barrier()
        win.append(time.perf_counter() - t0)
        w += 1
        if w == 1 and args.min_seconds > 0:
            # as many windows as it takes for the timed region to last --min-seconds (the driver's independent busy
            # sampling must be able to see the run); every rank derives the same count from the slowest rank's first window
            t1 = torch.tensor([win[0]], dtype=torch.float64, device="cuda" if (world > 1 and dist.get_backend() == "nccl") else "cpu")
            if world > 1:
                dist.all_reduce(t1, op=dist.ReduceOp.MAX)
            n_windows = int(min(4000, max(n_windows, np.ceil(args.min_seconds / max(float(t1.item()), 1e-6)))))
    seg._dk.check_status()
    if world > 1:
        t = torch.tensor(win, dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        win = [float(v) for v in t.cpu()]
    elapsed = float(np.median(win))
    if ev_after:
        sweeper.use_graph = False
        _abi.check(_abi.lib().segk_profile_enable(_abi.ctx(), 1))
        for _ in range(args.steps):
            seg.batch_sweep_async()
        barrier()
        sweeper.use_graph = True

    score_ms, score_rows = None, rows_local
    if use_ev:
        nmax = min(args.steps * len(win), 256)
        ms = (C.c_float * nmax)()
        rows = (C.c_int64 * nmax)()
        got = _abi.lib().segk_profile_read(_abi.ctx(), ms, rows, nmax)
        if got > 0:
            score_ms = float(np.mean(ms[:got]))
            score_rows = int(rows[got - 1])
        _abi.check(_abi.lib().segk_profile_enable(_abi.ctx(), 0))

    # rows the hinted exact stage passed on to the second stage / rows of the full scan, last sweep (diagnostic)
    sc = (C.c_int32 * 2)()
    _abi.check(_abi.lib().segk_kmeans_stage_counts(_abi.ctx(), C.byref(seg._dk.cand), sc, _abi.stream()))
    stage_counts = {"second_stage_rows": int(sc[0]), "full_scan_rows": int(sc[1]), "rows": int(rows_local)}

    # N > 1: the cost of the per-sweep collective, measured outside the timed windows -- `steps` further sweeps with the
    # all-gather bracketed by stream synchronisation on every rank (comm.TorchComm.timed)
    gather_info = None
    if world > 1:
        comm = sweeper.comm
        comm.timed, comm.gather_us, comm.gather_calls = True, 0.0, 0
        for _ in range(args.steps):
            seg.batch_sweep_async()
        barrier()
        comm.timed = False
        g = torch.tensor([comm.gather_us / max(comm.gather_calls, 1)], dtype=torch.float64,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(g, op=dist.ReduceOp.MAX)
        gather_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                       "all_gather_us_per_sweep_max_over_ranks": float(g.item()),
                       "record_bytes_per_rank": int(sweeper.rank_stride * 8),
                       "note": "host wall time of the one all-gather of a sweep between two stream synchronisations "
                               "(includes the launch of the collective), measured on sweeps outside the timed windows"}

    # N = 1: the reference's own chain (sync='sequential', the API default; bit-identical to the reference) on the same
    # corpus, outside the timed region: one warm sweep, two timed by the chain's own record['sample_time'] bracket
    seq_chain = None
    if world == 1 and rank == 0 and not args.no_seq_chain:
        try:
            random.seed(0)
            np.random.seed(0)
            seg_s = kaw.SegmentalKMeansWordseg(args.K, *corpus, n_slices_max=args.n_slices_max, init_am_assignments="spread")
            seg_s.segment(1)
            rec_s = seg_s.segment(2)
            t_s = float(np.mean(rec_s["sample_time"]))
            seq_chain = {"sweeps_per_s": 1.0 / t_s, "us_per_utterance": 1e6 * t_s / args.utts,
                         "note": "SegmentalKMeansWordseg(sync='sequential').segment: the reference's chain, every utterance sees "
                                 "the means the previous one left (kmeans_acoustic_wordseg.py:393-399), bit-identical state "
                                 "(tests/test_gpu_chain_parity_fullshape.py); one GPU, mean of record['sample_time'] over two "
                                 "sweeps after one warm sweep; does not shard"}
            del seg_s
        except Exception as e:                           # the extra key must never cost the line
            seq_chain = {"error": repr(e)}

    # N = 1: sweeps 1-10 of a fresh chain on both score paths, and the mini-batch form beside `value` (outside the timed region)
    early, mb = None, None
    if world == 1 and rank == 0 and not args.no_early:
        try:
            early = early_sweeps(kaw, corpus, args)
        except Exception as e:                           # the extra keys must never cost the line
            early = {"error": repr(e)}
        try:
            mb = minibatch_rate(kaw, corpus, args, 8, 20)
        except Exception as e:
            mb = {"error": repr(e)}

    if rank == 0:
        # algorithmic: 2 rows K D of the timed launch.  K = all K_max slots: those beyond the active components hold
        # random means and are candidates like any other, as in the reference (kmeans_components.py:149-166, 225-226)
        flops_per_launch = 2.0 * score_rows * args.K * args.dim
        out = {
            "metric": METRIC,
            "value": args.steps / elapsed,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "windows": len(win),
            "window_ms_per_step": {"min": 1e3 * min(win) / args.steps, "median": 1e3 * elapsed / args.steps,
                                   "max": 1e3 * max(win) / args.steps},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "SegmentalKMeansWordseg batch-synchronous sweep (BASELINE.json configs[2])",
                "utterances": args.utts, "landmarks_per_utt": args.landmarks,
                "n_slices_max": args.n_slices_max, "embeddings": int(n_emb), "D": args.dim, "K": args.K,
                "parallelism": "utterance shards x%d, ONE all-gather of the packed block statistics per sweep" % world,
                "collective": ("none (single process)" if world == 1 else
                               "%s, world size %d as reported by torch.distributed" % (dist.get_backend(), dist.get_world_size())),
                "sweep_launch": "hipGraph replay" if sweeper.use_graph else "plain launches",
                "corpus": ("rows sharded over the ranks: %d of %d on rank 0 (float32 matrix, fp16 planes, per-row work arrays); "
                           "component statistics replicated" % (seg._corpus.n_emb, n_emb)) if world > 1 else "one device holds all rows",
                "device_memory_allocated_bytes_rank0": int(torch.cuda.max_memory_allocated()),
                "components_after": int(seg.acoustic_model.components.K),
            },
        }
        if score_ms is not None:
            # HBM bytes per launch of the same kernel from the PMC passes committed under profiles/
            # (collected in separate rocprofv3 --pmc runs, gfx950-corrected); only quoted for the
            # workload they were measured on
            traffic = None
            b3_on = getattr(seg._corpus, "Xb3", None) is not None and os.environ.get("SEGK_SCORE_B3", "2") != "0"
            tpath = os.path.join(ROOT, "profiles", "score_kernel_traffic_b3.json" if b3_on else "score_kernel_traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                wl = tj["workload"]
                if (wl["utterances"], wl["landmarks_per_utt"], wl["n_slices_max"], wl["D"], wl["K"], wl["n_gpus"]) == \
                        (args.utts, args.landmarks, args.n_slices_max, args.dim, args.K, world):
                    traffic = tj["traffic_bytes_per_launch"]
            achieved = flops_per_launch / (score_ms * 1e-3) / 1e12
            b3 = getattr(seg._corpus, "Xb3", None) is not None and os.environ.get("SEGK_SCORE_B3", "2") != "0"
            kind = int(_abi.lib().segk_profile_last_kind(_abi.ctx())) if use_ev else -1
            launches = int(_abi.lib().segk_profile_last_launches(_abi.ctx())) if use_ev else 1
            if b3 and kind == 5:
                # segk_kmeans_score_hinted (every sweep but the first): k_kmeans_top2_rs scores every row against every
                # component with ONE v_mfma_f32_32x32x16_f16 product per 16 dimensions and keeps the two largest values per
                # row; the exact stage verifies the previous sweep's winner against them (segk_score_hint.hip).  `achieved`
                # / `frac`: the contract's ALGORITHMIC flops 2*rows*K*D over this launch; executed_*: the padded product.
                kp, kpad = (args.dim + 15) // 16 * 16, (args.K + 31) // 32 * 32
                executed = 2.0 * score_rows * kpad * kp
                ex_tf = executed / (score_ms * 1e-3) / 1e12
                tp = os.path.join(ROOT, "profiles", "score_kernel_traffic_hint.json")
                traffic = None
                if os.path.exists(tp):
                    tj = json.load(open(tp))
                    wl = tj["workload"]
                    if (wl["utterances"], wl["landmarks_per_utt"], wl["n_slices_max"], wl["D"], wl["K"], wl["n_gpus"]) == \
                            (args.utts, args.landmarks, args.n_slices_max, args.dim, args.K, world):
                        traffic = tj["traffic_bytes_per_launch"]
                out["dtype"] = "fp16 one-product filter + fp16x2 second stage, float32 reference arithmetic for the results"
                out["roofline"] = {
                    "bound": "mfma",
                    "kernel": "k_kmeans_top2_rs<%d> (all %d rows of the rank x all %d component slots; one fp16 product on the 16-bit "
                              "matrix pipe, value-only top-2, tile images resident in LDS; the winners are verified in reference "
                              "arithmetic by k_kmeans_hint_exact, results bit-identical to the float32 reference)"
                              % (kp // 16, score_rows, args.K),
                    "achieved": achieved, "peak": PEAK_BF16_MATRIX_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_BF16_MATRIX_TFLOPS, "traffic": traffic,
                    "traffic_source": ("profiles/score_kernel_traffic_hint.json (static: separate rocprofv3 --pmc passes of this "
                                       "command on this workload, FETCH_SIZE x2 + WRITE_SIZE, committed; not measured in this run)"
                                       if traffic is not None else None),
                    "ms_per_launch": score_ms,
                    "launches_in_interval": 1,
                    "timed_launches": "HIP events around the kernel in every %d-th sweep of the timed region" % max(1, args.event_period),
                    "flops_per_launch": flops_per_launch,
                    "executed_flops_per_launch": executed, "executed_achieved": ex_tf,
                    "executed_frac": ex_tf / PEAK_BF16_MATRIX_TFLOPS,
                    "achieved_over_fp32_matrix_peak": achieved / PEAK_FP32_MATRIX_TFLOPS,
                }
            elif b3 and kind == 1:
                # One-product fp16 pre-filter (k_kmeans_score_h1) in front of the split-precision kernel: the timed
                # launch scores every row against every component with ONE v_mfma_f32_32x32x16_f16 product per
                # 16 dimensions; rows it cannot decide (6 % here) pass to the three-product kernel.  `achieved`
                # / `frac` are the contract's ALGORITHMIC flops 2*rows*K*D over this launch; executed_* count the
                # padded product the matrix pipe really does.
                kp, kpad = (args.dim + 15) // 16 * 16, (args.K + 31) // 32 * 32
                executed = 2.0 * score_rows * kpad * kp
                ex_tf = executed / (score_ms * 1e-3) / 1e12
                tp = os.path.join(ROOT, "profiles", "score_kernel_traffic_pre.json")
                traffic = None
                if os.path.exists(tp):
                    tj = json.load(open(tp))
                    wl = tj["workload"]
                    if (wl["utterances"], wl["landmarks_per_utt"], wl["n_slices_max"], wl["D"], wl["K"], wl["n_gpus"]) == \
                            (args.utts, args.landmarks, args.n_slices_max, args.dim, args.K, world):
                        traffic = tj["traffic_bytes_per_launch"]
                out["dtype"] = "fp16 pre-filter + fp16x2"
                out["roofline"] = {
                    "bound": "mfma",
                    "kernel": "k_kmeans_score_h1<%d, 4> (main launch: %d of %d rows; one-product fp16 pre-filter on the 16-bit "
                              "matrix pipe, undecided rows re-scored by k_kmeans_score_sp, results bit-identical to the float32 "
                              "reference)" % (kp // 16, score_rows, rows_local),
                    "achieved": achieved, "peak": PEAK_BF16_MATRIX_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_BF16_MATRIX_TFLOPS, "traffic": traffic, "ms_per_launch": score_ms / launches,
                    "launches_in_interval": launches,
                    "flops_per_launch": flops_per_launch / launches,
                    "executed_flops_per_launch": executed / launches, "executed_achieved": ex_tf,
                    "executed_frac": ex_tf / PEAK_BF16_MATRIX_TFLOPS,
                    "achieved_over_fp32_matrix_peak": achieved / PEAK_FP32_MATRIX_TFLOPS,
                }
            elif b3:
                # The filter runs as three fp16 (or six bf16) products per float32 multiply-add on
                # v_mfma_f32_32x32x16_{f16,bf16} (splits of both operands).  `achieved` / `frac` follow
                # the contract (ALGORITHMIC flops 2*rows*K*D over the kernel's duration, against the dense peak
                # of the dtype the matrix pipe computes in); the executed_* keys count what the pipe really
                # does: 3 (6) products x (K padded to 32) x (D padded to 16).
                kp, kpad = (args.dim + 15) // 16 * 16, (args.K + 31) // 32 * 32
                pieces = int(seg._corpus.c.sp_pieces)
                n_prod = 3 if pieces == 2 else 6
                executed = n_prod * 2.0 * score_rows * kpad * kp
                ex_tf = executed / (score_ms * 1e-3) / 1e12
                out["dtype"] = "fp16x2" if pieces == 2 else "bf16x3"
                out["roofline"] = {
                    "bound": "mfma",
                    "kernel": "k_kmeans_score_sp<%d, 4, %d> (main launch: %d of %d rows; float32 contraction as %s "
                              "splits on the 16-bit matrix pipe, results bit-identical to the float32 reference)"
                              % (kp // 16, pieces, score_rows, rows_local, "two-way fp16" if pieces == 2 else "three-way bf16"),
                    "achieved": achieved, "peak": PEAK_BF16_MATRIX_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_BF16_MATRIX_TFLOPS, "traffic": traffic, "ms_per_launch": score_ms,
                    "flops_per_launch": flops_per_launch,
                    "executed_flops_per_launch": executed, "executed_achieved": ex_tf,
                    "executed_frac": ex_tf / PEAK_BF16_MATRIX_TFLOPS,
                    "achieved_over_fp32_matrix_peak": achieved / PEAK_FP32_MATRIX_TFLOPS,
                }
            else:
                out["roofline"] = {
                    "bound": "mfma", "kernel": "k_kmeans_score<25, 1, 4, 0> (main launch: %d of %d rows; the last partial "
                                               "round runs split-K in k_kmeans_score<..., 1>)" % (score_rows, rows_local),
                    "achieved": achieved,
                    "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MATRIX_TFLOPS,
                    "traffic": traffic, "ms_per_launch": score_ms,
                    "flops_per_launch": flops_per_launch,
                }
        out["stage_counts"] = stage_counts
        if gather_info is not None:
            out["config"]["collective_measured"] = gather_info
        if seq_chain is not None:
            out["sequential_chain"] = seq_chain
        if early is not None:
            out["early_sweeps"] = early
        if mb is not None:
            out["batch_forms"] = {"n_batches_1": {"n_batches": 1, "sweeps_per_s": args.steps / elapsed,
                                                  "components_after": int(seg.acoustic_model.components.K)},
                                  "n_batches_8": mb,
                                  "note": "`value` is n_batches = 1 (statistics frozen for a whole sweep); n_batches = 8 refreshes them "
                                          "eight times per sweep (closer to the reference's per-utterance refresh: more components "
                                          "survive) at the cost of eight statistics passes per sweep"}
        if world == 1 and args.cpu_utts > 0:
            out["cpu_baseline"] = cpu_baseline(corpus, args.utts, args.K, args.n_slices_max, args.cpu_utts)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
