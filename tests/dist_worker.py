"""Worker of tests/test_gpu_dist.py: runs batch sweeps of the PRODUCT under torch.distributed
(any world size, gloo or nccl) and writes the final state of rank 0 to an .npz file."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, backend, n_sweeps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    ngpu = torch.cuda.device_count()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(ngpu, 1))
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend)
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(96, 24, 40, seed=3, ragged=True, n_slices_max=5, N_range=(4, 14))
    random.seed(11)
    np.random.seed(11)
    # (world > 1: every rank's device holds the rows of its own utterances only, unless --no-shard)
    seg = kaw.SegmentalKMeansWordseg(40, *corpus, n_slices_max=5, init_am_assignments="rand", sync="batch",
                                     n_stat_blocks=8, shard_corpus=False if "--no-shard" in sys.argv else None)
    if world > 1:
        assert (seg._dk.shard is None) == ("--no-shard" in sys.argv)
        if seg._dk.shard is not None:
            assert seg._corpus.n_emb == seg._dk.shard[1] - seg._dk.shard[0] < seg.acoustic_model.components.N
    # optional: --load CKPT resumes from a checkpoint (possibly written under another world size) before sweeping,
    # --save CKPT writes one after the sweeps (every rank builds it: state_dict() is a collective; rank 0 stores it)
    extra = sys.argv[4:]
    load = extra[extra.index("--load") + 1] if "--load" in extra else None
    save = extra[extra.index("--save") + 1] if "--save" in extra else None
    if load:
        import pickle
        seg.load_state_dict(pickle.load(open(load, "rb")))
    rec = seg.segment(n_sweeps)
    c = seg.acoustic_model.components
    state = dict(assignments=c.assignments, means=c.means, mean_numerators=c.mean_numerators, counts=c.counts,
                 K=np.array(c.K), totals=np.array(rec["sum_neg_len_sqrd_norm"]),
                 n_tokens=np.array(rec["n_tokens"]))
    # the boundaries of the utterances owned by other ranks are fetched by the product (a collective: every rank
    # reads the attribute)
    state["boundaries"] = seg.utterances.boundaries.astype(np.uint8)
    sd = seg.state_dict()
    assert np.array_equal(sd["boundaries"], seg.utterances.boundaries)
    assert np.array_equal(sd["km_assignments"], c.assignments)
    if save and rank == 0:
        import pickle
        pickle.dump(sd, open(save, "wb"))
    if rank == 0:
        np.savez(out_path, **state)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
