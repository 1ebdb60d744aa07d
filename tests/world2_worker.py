"""One rank of the world-size-2 library test (started by tests/test_gpu_world2.py, one process per rank, both on GPU 0
with the test double of librccl).  Every rank holds a block of the streams, joins the communicator through an id file,
switches the per-step density all-reduce on, steps, and writes what it saw."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import sp_oracle as o  # noqa: E402  (initial conditions only)
from msm_amd import Evolver  # noqa: E402
from msm_amd.sharding import partition_streams  # noqa: E402

TOPHAT = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.05, num_data_dumps=200, total_mass=1e11,
              hbar_=0.05, k2_cutoff=0.95, alias_threshold=0.02)


def stream_psi(N, s):
    rng = np.random.default_rng(100 + s)
    return o.normalize(1.0 + 0.05 * (rng.standard_normal((N, N, N)) + 1j * rng.standard_normal((N, N, N))), 30.0 / N, 3)


def main():
    rank, world, N, total, steps, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    first, count = partition_streams(total, world, rank)
    idfile = os.path.join(out, "comm.id")
    with Evolver(size=N, n_streams=count, batch_streams=2, **TOPHAT) as ev:
        for i in range(count):
            ev.set_psi(i, stream_psi(N, first + i))
        if rank == 0:
            with open(idfile + ".tmp", "wb") as f:
                f.write(Evolver.comm_unique_id())
            os.rename(idfile + ".tmp", idfile)
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120:
                raise SystemExit("no communicator id")
            time.sleep(0.02)
        ev.comm_init(open(idfile, "rb").read(), rank, world)
        assert "fake_rccl" in Evolver.comm_origin(), Evolver.comm_origin()
        got = ev.comm_allreduce([float(rank + 1), -float(rank)], Evolver.COMM_SUM)
        assert got == [sum(range(1, world + 1)), -sum(range(world))], got
        assert ev.comm_allreduce([float(rank)], Evolver.COMM_MAX) == [float(world - 1)]
        assert ev.comm_allreduce([float(rank)], Evolver.COMM_MIN) == [0.0]
        ev.mean_density_begin(total)
        means = []
        for i in range(steps):
            ev.update()
            if i % 2 == 0:                       # a flag exchange between the steps, as the drivers do
                ev.comm_allreduce([1.0], Evolver.COMM_MAX)
            means.append(ev.mean_density())
        # the synthesizer's sums of one slot over the ranks; rank 1 adds nothing to slot 7
        for i in range(count):
            ev.synth_accumulate(i, 3)
            if rank == 0:
                ev.synth_accumulate(i, 7)
        ev.synth_allreduce(3)
        ev.synth_allreduce(7)
        np.savez(os.path.join(out, f"rank{rank}.npz"), means=np.array(means),
                 psi=np.array([ev.get_psi(i) for i in range(count)]),
                 synth_psi=ev.synth_fetch(3, "psi"), synth_psi2=ev.synth_fetch(3, "psi2"), synth_psik2=ev.synth_fetch(3, "psik2"),
                 count3=ev.synth_device(3)[1], count7=ev.synth_device(7)[1], qx=ev.synth_qx(3),
                 dt=np.array([r.dt for r in ev.status()]))
        ev.comm_allreduce([0.0])                 # leave together


if __name__ == "__main__":
    main()
