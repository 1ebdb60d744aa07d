"""What a multi-rank driver exchanges, behind one small interface with two carriers.

The reference's only inter-process traffic is the synthesizer's MPI gathers (synthesizer/src/lib.rs:466-583).  Here a
rank exchanges (i) a few flags per step and (ii) the synthesizer's sums per dump index.

  LibraryComm  the product path: RCCL inside libmsm_hip.so (msm_comm_*), one rank per GPU.  Only the 128-byte
               communicator id travels outside it, through `exchange_id` (a launcher's key-value store, a file, ...).
  GlooComm     torch.distributed on the CPU for rehearsals of the driver logic without GPUs (tests); also what carries
               the sums when several ranks have to share one GPU, which RCCL refuses.

Both order their collectives by the host's call order: every rank must make the same calls in the same order.
"""
import numpy as np

SUM, MAX, MIN = 0, 2, 3          # RCCL's ncclRedOp_t codes, as msm_comm_allreduce_f64 takes them


class NoComm:
    """One rank: every reduction is the identity."""
    rank, world = 0, 1

    def allreduce(self, values, op=SUM):
        return [float(v) for v in values]

    def synth_allreduce(self, ev, slot, n_cells=None):
        """Sums slot `slot` of ev's synthesizer over the ranks in place; returns the global stream count.
        n_cells: cells per grid, for a rank that holds no device context at all (ev is None)."""
        return ev.synth_device(slot)[1]

    def barrier(self):
        pass


class LibraryComm(NoComm):
    def __init__(self, ev, rank: int, world: int, exchange_id):
        """ev: the rank's Evolver.  exchange_id(id_or_None) -> id: rank 0 passes the id it created and gets it back,
        the other ranks pass None and receive rank 0's."""
        self.ev, self.rank, self.world = ev, rank, world
        cid = exchange_id(type(ev).comm_unique_id() if rank == 0 else None)
        ev.comm_init(bytes(cid), rank, world)

    def allreduce(self, values, op=SUM):
        return self.ev.comm_allreduce(values, op)

    def synth_allreduce(self, ev, slot, n_cells=None):
        ev.synth_allreduce(slot)             # sums and stream count over the ranks; zeros from a rank without the slot
        return ev.synth_device(slot)[1]

    def barrier(self):
        self.ev.comm_allreduce([0.0], SUM)


class GlooComm(NoComm):
    def __init__(self, dist, to_tensor=None):
        """dist: an initialised torch.distributed module.  to_tensor(handle, n_doubles) wraps what ev.synth_device
        returns as a tensor the backend can reduce (default: a device pointer, staged through the host)."""
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()
        self._to_tensor = to_tensor

    def allreduce(self, values, op=SUM):
        import torch
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        self.dist.all_reduce(t, op={SUM: self.dist.ReduceOp.SUM, MAX: self.dist.ReduceOp.MAX, MIN: self.dist.ReduceOp.MIN}[op])
        return t.tolist()

    def synth_allreduce(self, ev, slot, n_cells=None):
        import torch
        n_doubles = 2 * 4 * int(n_cells if ev is None else np.prod(ev.shape))
        try:
            handle, count = ev.synth_device(slot)
        except Exception:                    # no context, or nothing accumulated here: contribute zeros
            handle, count = None, 0
        if handle is None:
            t = torch.zeros(n_doubles, dtype=torch.float64)
            self.dist.all_reduce(t)
        elif self._to_tensor is not None:
            self.dist.all_reduce(self._to_tensor(handle, n_doubles))
        else:
            # a raw device pointer and a CPU backend: through torch's view of the device buffer and back
            dev = torch.as_tensor(_DeviceView(handle, n_doubles), device="cuda")
            host = dev.cpu()
            self.dist.all_reduce(host)
            dev.copy_(host)
        total = int(round(self.allreduce([count], SUM)[0]))
        if handle is not None:
            ev.synth_set_count(slot, max(total, 1))
        return total

    def barrier(self):
        self.dist.barrier()


class _DeviceView:
    """Minimal __cuda_array_interface__ holder so that torch can wrap a raw device pointer."""

    def __init__(self, ptr: int, n_doubles: int):
        self.__cuda_array_interface__ = {"shape": (n_doubles,), "typestr": "<f8", "data": (ptr, False), "version": 3}
