"""
N ranks of the batch modes inside ONE process (test infrastructure).

A GPU test box has one card and admits at most six processes on it, so configs[3]'s eight ranks cannot be eight
torch.distributed processes there.  The drivers accept any object with the communicator interface of
segmentalist_amd/comm.py as `process_group`; `VirtualWorld` provides one per rank, each rank being a Python thread.
Exactly ONE thread runs at any time (a baton lock, handed over only inside collectives), so the process-global RNG
streams and the library context see the same strictly sequential call pattern as separate processes would produce per
rank; a collective completes when every rank has entered it, exactly like the real thing, and a rank that skips one
deadlocks the test (with a timeout) instead of passing.  All ranks share the device's default stream, which orders the
exchange copies after the kernels that produced the rows.
"""
import threading


class _Abort(Exception):
    pass


class VirtualComm(object):
    backend = "virtual"

    def __init__(self, world_obj, rank):
        self.w, self.rank, self.world = world_obj, rank, world_obj.world
        self.calls = {"all_gather_rows": 0, "all_reduce_max": 0, "all_gather_object": 0}

    def _collective(self, payload, do):
        w = self.w
        w.slots[self.rank] = payload
        w.baton.release()
        try:
            idx = w.barrier.wait(timeout=w.timeout)
            if idx == 0:
                with w.baton:
                    w.result = do(list(w.slots))
            w.barrier.wait(timeout=w.timeout)
        except threading.BrokenBarrierError:
            w.baton.acquire()
            raise _Abort("another rank failed or skipped a collective")
        w.baton.acquire()
        return w.result

    def all_gather_rows(self, out, inp):
        self.calls["all_gather_rows"] += 1

        def do(slots):
            for r, (o, _) in enumerate(slots):
                for q, (_, i) in enumerate(slots):
                    if q != r:
                        o[q].copy_(i.view(o[q].shape))
        self._collective((out, inp), do)

    def all_reduce_max(self, t):
        self.calls["all_reduce_max"] += 1

        def do(slots):
            import torch
            m = slots[0].clone()
            for s in slots[1:]:
                torch.maximum(m, s, out=m)
            for s in slots:
                s.copy_(m)
        self._collective(t, do)

    def all_gather_object(self, obj):
        self.calls["all_gather_object"] += 1
        return self._collective(obj, lambda slots: slots)


class VirtualWorld(object):
    def __init__(self, world, timeout=600):
        self.world, self.timeout = world, timeout
        self.baton = threading.Lock()
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.result = None

    def run(self, fn):
        """fn(comm) on every rank; returns the list of results (re-raises the first failure)."""
        results, errors, aborted = [None] * self.world, [], []

        def body(rank):
            import torch
            if torch.cuda.is_available():
                torch.cuda.set_device(0)
            self.baton.acquire()
            try:
                results[rank] = fn(VirtualComm(self, rank))
            except _Abort as e:
                aborted.append((rank, e))
            except BaseException as e:          # noqa: B902 -- report, then unblock the other ranks
                errors.append((rank, e))
                self.barrier.abort()
            finally:
                self.baton.release()

        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0][1]
        if aborted:
            raise RuntimeError("ranks %s waited in a collective that the others never entered" % [r for r, _ in aborted])
        return results
