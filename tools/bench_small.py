"""Step rate of small and medium grids (launch-bound regime): ms per step and stream-cell updates/s.

    python tools/bench_small.py [--size 128] [--streams 1] [--steps 200] [--precision f64]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msm_amd import Evolver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--precision", default="f64")
ap.add_argument("--batch", type=int, default=0, help="streams per kernel launch (0: the library's choice)")
args = ap.parse_args()
N, ns = args.size, args.streams
base = dict(axis_length=30.0, final_sim_time=1e9, cfl=0.05, num_data_dumps=2, total_mass=1e11, hbar_=0.05,
            k2_cutoff=0.95, alias_threshold=0.02)
rng = np.random.default_rng(1)
psi = 1.0 + 0.05 * (rng.standard_normal((N, N, N)) + 1j * rng.standard_normal((N, N, N)))
psi /= np.sqrt(np.sum(np.abs(psi) ** 2) * (30.0 / N) ** 3)
with Evolver(size=N, n_streams=ns, precision=args.precision, batch_streams=args.batch, **base) as ev:
    ev.set_psi(0, psi.astype(ev.cplx))
    for s in range(1, ns):
        ev.copy_psi(s, 0)
    for _ in range(5):
        ev.update()
    ev.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev.update()
    ev.sync()
    dt = (time.perf_counter() - t0) / args.steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev.update()
        ev.status()
    dts = (time.perf_counter() - t0) / args.steps
print(f"{N}^3 {args.precision} x {ns} (batch {args.batch or 'auto'}): {dt * 1e6:.1f} us/step queued, {dts * 1e6:.1f} us/step with a status read per step, "
      f"{N ** 3 * ns / dt:.3e} updates/s")
