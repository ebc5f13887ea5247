"""Worker of tests/test_gpu_dist.py: batch sweeps of the FBGMM / bigram PRODUCT samplers under
torch.distributed (any world size, gloo or nccl); rank 0 writes the merged final state."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, backend, n_sweeps, kind = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    prec = sys.argv[5] if len(sys.argv) > 5 else "f64"
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    ngpu = torch.cuda.device_count()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(ngpu, 1))
    if world > 1:
        dist.init_process_group(backend)
    from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    D, K = 20, 30
    corpus = make_corpus(90, D, K, seed=3, ragged=True, n_slices_max=5, N_range=(4, 14))
    random.seed(11)
    np.random.seed(11)
    kw = dict(n_slices_min=0, n_slices_max=5, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
              init_am_assignments="rand", time_power_term=1.0, sync="batch", n_gibbs_blocks=3, n_stat_blocks=8,
              batch_seed=5)
    fixed = (0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D))
    if kind == "bigram":
        seg = baw.BigramAcousticWordseg(K, FixedVarPrior(*fixed), {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5},
                                        *corpus, covariance_type="fixed", fb_type="unigram", score_precision=prec, **kw)
    else:
        seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, NIW(np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D)),
                                         *corpus, covariance_type="diag", fb_type="standard", **kw)
    rec = seg.gibbs_sample(n_sweeps)
    c = seg.acoustic_model.components
    sw = seg._get_sweeper()
    state = dict(assignments=c.assignments, counts=c.counts, K=np.array(c.K), stat_a=c.dev.stat_a.cpu().numpy(),
                 stat_b=c.dev.stat_b.cpu().numpy(), log_marg=np.array(rec["log_marg"]),
                 lml=np.array(rec["log_marg*length"]), partials=sw.partials.cpu().numpy())
    if kind == "bigram":
        state["unigram"], state["bigram"] = seg.lm.unigram_counts, seg.lm.bigram_counts
    b = seg._dev_bounds.cpu()
    if world > 1:
        gathered = [torch.empty_like(b) for _ in range(world)]
        dist.all_gather(gathered, b)
        full = b.clone()
        for r in range(world):
            lo = int(sw.utt_range_np[r * sw.s_n, 0, 0])
            hi = int(sw.utt_range_np[(r + 1) * sw.s_n - 1, -1, 1])
            full[lo:hi] = gathered[r][lo:hi]
        b = full
    state["boundaries"] = b.numpy()
    if rank == 0:
        np.savez(out_path, **state)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
