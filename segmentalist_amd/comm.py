"""
The collectives of the multi-GPU batch modes behind one small interface.

The reference has no parallelism of any kind; the batch modes shard the utterances over the ranks and exchange
packed records of component statistics (k-means: ONE all-gather per sweep or mini-batch; FBGMM / bigram: one per
Gibbs step).  Everything the drivers need from a communicator is:

    rank, world
    all_gather_rows(out, inp)     out[r] <- inp of rank r; `inp` is out[rank] (in place)
    all_reduce_max(t)             element-wise maximum over the ranks, in place
    all_gather_object(obj)        -> list of every rank's picklable `obj`

`TorchComm` maps them onto torch.distributed: backend "nccl" IS RCCL on ROCm (device buffers directly, over xGMI);
under "gloo" (CPU tests, several ranks sharing one GPU) the buffers are staged through host memory.  Any object with
the same attributes can be passed to the drivers as `process_group` instead of a torch.distributed group -- the tests
use that to run EIGHT ranks inside one process on a one-GPU box (tests/virtual_ranks.py), where eight processes on one
card are not allowed.

`TorchComm.gather_us` accumulates the host-side wall time of the row all-gathers bracketed by stream synchronisation
when `timed` is set (bench.py: the measured cost of the collective per sweep); untimed, nothing synchronises.
"""
import time


class SingleComm(object):
    """One rank: nothing to exchange."""
    rank, world, backend = 0, 1, "none"

    def all_gather_rows(self, out, inp):
        pass

    def all_reduce_max(self, t):
        pass

    def all_gather_object(self, obj):
        return [obj]


class TorchComm(object):
    def __init__(self, group=None):
        import torch.distributed as dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.timed = False
        self.gather_us = 0.0
        self.gather_calls = 0

    def all_gather_rows(self, out, inp):
        import torch
        import torch.distributed as dist
        if self.timed:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if self.backend == "nccl":
            dist.all_gather_into_tensor(out.view(-1), inp.reshape(-1), group=self.group)
        else:
            host = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(self.world)]
            dist.all_gather(host, inp.cpu(), group=self.group)
            for r, h in enumerate(host):
                out[r].copy_(h)
        if self.timed:
            torch.cuda.synchronize()
            self.gather_us += (time.perf_counter() - t0) * 1e6
            self.gather_calls += 1

    def all_reduce_max(self, t):
        import torch.distributed as dist
        if self.backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
            t.copy_(h)

    def all_gather_object(self, obj):
        import torch.distributed as dist
        parts = [None] * self.world
        dist.all_gather_object(parts, obj, group=self.group)
        return parts


def get_comm(group=None):
    """`group`: None (the default torch.distributed group when one is initialised, else a single rank), a
    torch.distributed process group, or any object with the communicator interface above."""
    if group is not None and hasattr(group, "all_gather_rows"):
        return group
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return TorchComm(group)
    except ImportError:
        pass
    return SingleComm()
