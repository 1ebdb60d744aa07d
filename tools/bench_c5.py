"""Per-GPU share of the largest benchmark configuration (512^3, single precision, expanding box): ms per step,
stream-cell updates/s and the fraction of the single-precision roofline (112 B per update), with per-pass timings.

    python tools/bench_c5.py [--streams 64] [--batch 32] [--steps 5] [--two-pass 0|2] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=64)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--two-pass", type=int, default=None, help="0: generic engine, 2: two passes per transform (default: the library's choice)")
ap.add_argument("--json", default=None)
args = ap.parse_args()
if args.two_pass is not None:
    os.environ["MSM_TWO_PASS"] = str(args.two_pass)
from msm_amd import Evolver  # noqa: E402

N, ns = args.size, args.streams
cosmo = dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01)
base = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.05, num_data_dumps=200, total_mass=1e11, hbar_=0.05,
            k2_cutoff=0.95, alias_threshold=0.02)
rng = np.random.default_rng(5)
psi = (1.0 + 0.05 * rng.standard_normal((N, N, N), dtype=np.float32)
       + 0.05j * rng.standard_normal((N, N, N), dtype=np.float32)).astype(np.complex64)
with Evolver(size=N, n_streams=ns, precision="f32", cosmology=cosmo, batch_streams=args.batch, **base) as ev:
    mode = ev.derived()["transform_passes_mode"]
    dv = ev.derived()["dx"] ** 3
    psi *= np.float32(1.0 / np.sqrt(np.sum(np.abs(psi).astype(np.float64) ** 2) * dv))
    ev.set_psi(0, psi)
    for s in range(1, ns):
        ev.copy_psi(s, 0)
    for s in range(ns):
        ev.sample(s, "Husimi", s + 1)

    def step():
        ev.step_begin([0.5] * ns, [1e9] * ns)
        st = ev.status()
        ev.step_end([1.0] * ns, [s.time + s.dt for s in st], [s.tau + s.dt for s in st], [0.0] * ns)

    for _ in range(2):
        step()
    ev.sync()
    ev.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ev.sync()
    dt = (time.perf_counter() - t0) / args.steps
    prof = {k: (ms / cnt, cnt / args.steps) for k, (ms, cnt) in ev.profile_report().items() if cnt}
    st = ev.status()
    assert all(r.n_steps == 2 + args.steps and r.aliased == 0 for r in st)
    norm = float(np.sum(np.abs(ev.get_psi(ns - 1)).astype(np.float64) ** 2) * dv)
out = dict(size=N, streams=ns, batch=args.batch, passes_per_transform=2 if mode >= 2 else 3, ms_per_step=dt * 1e3,
           updates_per_s=N ** 3 * ns / dt, frac_of_112B_roofline=N ** 3 * ns / dt * 112 / 8e12,
           per_pass_ms={k: round(v[0], 3) for k, v in prof.items()}, launches_per_step={k: v[1] for k, v in prof.items()},
           norm_last_stream=norm)
for k, (ms, cnt) in prof.items():
    print(f"{k:16s} {ms:8.3f} ms x {cnt:.0f}/step")
print(f"{N}^3 fp32 expanding x {ns} streams ({out['passes_per_transform']} passes per transform): {dt * 1e3:.1f} ms/step, "
      f"{out['updates_per_s']:.3e} updates/s, {out['frac_of_112B_roofline'] * 100:.1f}% of the fp32 roofline (112 B/update); "
      f"norm of the last stream {norm:.6f}")
if args.json:
    with open(args.json, "w") as f:
        json.dump(out, f)
