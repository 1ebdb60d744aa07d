"""Dump-path timing at 256^3 fp64: synchronous msm_get_psi into pageable numpy planes vs msm_get_psi_async into
pinned planes overlapped with the following steps.  Prints GB/s of host-visible dump data and the step slow-down."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msm_amd import Evolver
from msm_amd.ics import plane_wave_psi

N, ns = 256, 32
base = dict(axis_length=60.0, final_sim_time=2000.0, cfl=0.1, num_data_dumps=64, total_mass=3e16, hbar_=0.01,
            ntot=1e10, k2_cutoff=0.95, alias_threshold=0.001)
with Evolver(size=N, n_streams=ns, **base) as ev:
    ev.set_psi(0, plane_wave_psi(N))
    for s in range(1, ns):
        ev.copy_psi(s, 0)
    for s in range(ns):
        ev.sample(s, "Wigner", s + 1)
    for _ in range(2):
        ev.update()
    ev.sync()
    t0 = time.perf_counter()
    for _ in range(4):
        ev.update()
    ev.sync()
    step = (time.perf_counter() - t0) / 4
    gb = N ** 3 * 16 / 1e9
    t0 = time.perf_counter()
    for s in range(8):
        ev.get_psi(s)
    sync_t = (time.perf_counter() - t0) / 8
    planes = [ev.pinned_planes() for _ in range(8)]
    t0 = time.perf_counter()
    tickets = [ev.get_psi_async(s, planes[s]) for s in range(8)]
    issue_t = time.perf_counter() - t0
    for _ in range(4):
        ev.update()
    ev.sync()
    for t in tickets:
        ev.download_wait(t)
    total = time.perf_counter() - t0
    for p in planes:
        ev.free_pinned(p)
    print(f"step {step*1e3:.1f} ms for {ns} streams; sync dump {sync_t*1e3:.1f} ms/stream = {gb/sync_t:.1f} GB/s; "
          f"async: issue {issue_t*1e3/8:.1f} ms/stream, 8 dumps + 4 steps in {total*1e3:.1f} ms "
          f"(4 steps alone {4*step*1e3:.1f} ms) => {8*gb/max(total-4*step,1e-9):.1f} GB/s of exposed dump time")
