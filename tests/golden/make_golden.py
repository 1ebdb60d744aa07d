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


COSMO = dict(TOPHAT, sim_name="spherical-tophat-cosmo", final_sim_time=0.6, num_data_dumps=3,
             cosmology=dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01))


def run_dumps(toml, size, expanding, keep):
    """Whole run of the mean-field stream to final_sim_time (main.rs:61-69); keeps the dumps listed in
    `keep` (examples/spherical-tophat.toml: 200 dumps; examples/spherical-tophat-cosmo.toml under the
    feature "expanding", shortened to 3 dumps)."""
    params = list(o.simulation_iter(dict(toml, size=size), expanding=expanding))[-1]
    sim = o.SimulationObject(params)
    sim.dump()
    while sim.not_finished():
        sim.update()
    out = {f"dump_{k}": psi for k, psi in sim.dumps if k in keep}
    out["n_dumps"] = np.array(len(sim.dumps))
    out["n_steps"] = np.array(sim.parameters.n_steps)
    out["final_time"] = np.array(float(sim.parameters.time))
    return out


def run_expanding(toml, size, dtype, nsteps=6, keep=(1, 2, 6)):
    """Mean-field stream of an expanding box, step by step: beside the fields, the host-side scalars a caller of
    msm_step_begin / msm_step_end supplies (scale factor, tau to the next dump, a1 + a2 of the half-kicks, new
    time / tau, tau of the next dump time), so that the vectors drive the C ABI without the oracle."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    import expanding_host as xh
    params = list(o.simulation_iter(dict(toml, size=size), expanding=True, dtype=dtype))[-1]
    sim = o.SimulationObject(params)
    out = {"psi0": sim.psi.copy()}
    scal, host = [], []
    for i in range(1, nsteps + 1):
        a_now, tau_next, snapped = xh.step_inputs(sim)
        shadow = __import__("copy").deepcopy(sim)
        dtau = float(shadow.update()["dt"])
        a_sum, t_new, tau_new = xh.half_kicks(sim, dtau)
        info = sim.update()
        assert float(info["dt"]) == dtau
        scal.append([float(info["dt"]), float(info["dump"]), float(info["alias_mass"]), float(info["max_phi"]),
                     float(info["time"]), float(sim.parameters.tau)])
        host.append([a_now, tau_next, snapped, a_sum, t_new, tau_new])
        if i in keep:
            out[f"psi_{i}"] = sim.psi.copy()
    out["scalars"] = np.array(scal)
    out["host"] = np.array(host)
    return out


if __name__ == "__main__":
    jobs = {
        "tophat_n8_f64": (TOPHAT, 8, -1, np.float64),
        "tophat_n16_f64": (TOPHAT, 16, -1, np.float64),
        "tophat_n16_f32": (TOPHAT, 16, -1, np.float32),
        "tophat_n12_f64": (TOPHAT, 12, -1, np.float64),       # an edge that is not a power of two
        "planewave_n16_mft_f64": (PLANE, 16, -1, np.float64),
        "planewave_n16_wigner1_f64": (PLANE, 16, 0, np.float64),
    }
    extra = {
        "tophat_n32_f64": lambda: run(TOPHAT, 32, -1, np.float64, keep=(10,)),
        "tophat_n16_run200_f64": lambda: run_dumps(TOPHAT, 16, False, (0, 1, 200)),
        "tophat_cosmo_n16_run3_f64": lambda: run_dumps(COSMO, 16, True, (0, 1, 2, 3)),
        # examples/spherical-tophat-cosmo.toml in single precision under the feature "expanding": the mode of the
        # largest benchmark configuration (512^3 fp32), at a size the vectors stay small
        "tophat_cosmo_n32_steps_f32": lambda: run_expanding(dict(COSMO, cfl=0.05), 32, np.float32),
        "tophat_cosmo_n16_steps_f64": lambda: run_expanding(dict(COSMO, cfl=0.05), 16, np.float64),
    }
    for name, fn in extra.items():
        res = fn()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **res)
        print(name, os.path.getsize(path) // 1024, "KiB")
    for name, (toml, size, which, dtype) in jobs.items():
        res = run(toml, size, which, dtype)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **res)
        print(name, os.path.getsize(path) // 1024, "KiB", res["scalars"][:2].tolist())
