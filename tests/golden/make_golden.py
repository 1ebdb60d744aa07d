"""Generates the committed golden vectors from the numpy oracle (oracle/sp_oracle.py).

Run from the repo root:  python tests/golden/make_golden.py
Each file holds psi0 and, after steps 1, 2 and 10 of the split-step update,
psi, calculate_potential(psi), and the per-step scalars (dt, dump flag,
alias mass, max|phi|, time).  Inputs are the reference's own example
parameters (examples/spherical-tophat.toml, tomls/planeWave3d_e10_sym.toml)
at reduced grid sizes.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import sp_oracle as o  # noqa: E402

TOPHAT = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11,
              hbar_=0.05, sim_name="spherical-tophat", k2_cutoff=0.95, alias_threshold=0.02, dims=3,
              ics=dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0))
PLANE = dict(axis_length=60.0, final_sim_time=2000.0, cfl=0.1, num_data_dumps=64, total_mass=3e16,
             hbar_=0.01, ntot=1e10, sim_name="planeWave3d_e10_sym", k2_cutoff=0.95, alias_threshold=0.001,
             dims=3, ics=dict(type="UserSpecified", path=os.path.join(HERE, "planeWave3d_e10_sym.npz")),
             sampling=dict(seeds="1 to 2", scheme="Wigner"))


def run(toml, size, which, dtype, nsteps=10, keep=(1, 2, 10)):
    toml = dict(toml, size=size)
    params = list(o.simulation_iter(toml, dtype=dtype))[which]
    sim = o.SimulationObject(params)
    out = {"psi0": sim.psi.copy()}
    scal = []
    for i in range(1, nsteps + 1):
        info = sim.update()
        scal.append([float(info["dt"]), float(info["dump"]), float(info["alias_mass"]),
                     float(info["max_phi"]), float(info["time"])])
        if i in keep:
            out[f"psi_{i}"] = sim.psi.copy()
            # potential of the new psi, as the reference's dump recomputes it (simulation_object.rs:1167-1168)
            out[f"phi_{i}"] = sim.calculate_potential().copy()
    out["scalars"] = np.array(scal)
    out["seed"] = np.array(-1 if params.sampling is None else params.sampling[0])
    return out


if __name__ == "__main__":
    jobs = {
        "tophat_n8_f64": (TOPHAT, 8, -1, np.float64),
        "tophat_n16_f64": (TOPHAT, 16, -1, np.float64),
        "tophat_n16_f32": (TOPHAT, 16, -1, np.float32),
        "planewave_n16_mft_f64": (PLANE, 16, -1, np.float64),
        "planewave_n16_wigner1_f64": (PLANE, 16, 0, np.float64),
    }
    for name, (toml, size, which, dtype) in jobs.items():
        res = run(toml, size, which, dtype)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **res)
        print(name, os.path.getsize(path) // 1024, "KiB", res["scalars"][:2].tolist())
