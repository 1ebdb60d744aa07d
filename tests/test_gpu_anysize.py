"""GPU: grids whose edge is not a power of two (any even N; the reference asserts only evenness,
utils/fft.rs:105, and ArrayFire transforms any length) and N = 1024, against the numpy oracle.
These sizes run mixed-radix or chirp-z line transforms and the unfused step.  Tolerances as everywhere: 1e-11 * max|psi| after
10 steps (fp64), dt relative 1e-12; fp32 2e-4."""
import numpy as np
import pytest

import sp_oracle as o
from msm_amd import Evolver

pytestmark = pytest.mark.gpu

TOPHAT = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11,
              hbar_=0.05, k2_cutoff=0.95, alias_threshold=0.02)
ICS = dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0)


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def oracle_sim(dims, N, dtype=np.float64, psi=None, **over):
    toml = dict(TOPHAT, sim_name="t", dims=dims, size=N, ics=ICS, **over)
    return o.SimulationObject(list(o.simulation_iter(toml, dtype=dtype))[-1], psi=psi)


# 2-3-5-smooth edges take the mixed-radix line transform, every other even edge (a factor 7, 11, 13 ... or a large prime)
# the chirp-z one
@pytest.mark.parametrize("dims,N", [(1, 6), (1, 10), (1, 100), (1, 750), (1, 1000), (1, 1024), (2, 6), (2, 12), (2, 96), (2, 600),
                                    (3, 6), (3, 12), (3, 20), (3, 24), (3, 48), (3, 100),
                                    (1, 14), (1, 154), (1, 998), (2, 28), (2, 202), (3, 14), (3, 22), (3, 98)])
def test_fft_roundtrip_and_parseval(dims, N):
    rng = np.random.default_rng(dims * 10000 + N)
    psi = rng.standard_normal((N,) * dims) + 1j * rng.standard_normal((N,) * dims)
    with Evolver(size=N, n_streams=2, dims=dims, **TOPHAT) as ev:
        ev.set_psi(1, psi)
        ev.forward(1)
        k = ev.get_psi(1)
        assert rel_err(k, o.forward(psi)) < 1e-12
        assert abs(np.sum(np.abs(k) ** 2) / np.sum(np.abs(psi) ** 2) - 1) < 1e-12
        ev.inverse(1)
        assert rel_err(ev.get_psi(1), psi) < 1e-12


@pytest.mark.parametrize("dims,N,precision,tol,dtol", [(3, 12, "f64", 1e-11, 1e-12), (3, 24, "f64", 1e-11, 1e-12),
                                                       (3, 40, "f64", 1e-11, 1e-12), (2, 48, "f64", 1e-11, 1e-12),
                                                       (1, 100, "f64", 1e-11, 1e-12), (1, 768, "f64", 1e-11, 1e-12),
                                                       (3, 24, "f32", 2e-4, 2e-4), (2, 96, "f32", 2e-4, 2e-4),
                                                       (3, 14, "f64", 1e-11, 1e-12), (3, 26, "f64", 1e-11, 1e-12),
                                                       (2, 44, "f64", 1e-11, 1e-12), (1, 182, "f64", 1e-11, 1e-12),
                                                       (3, 28, "f32", 2e-4, 2e-4)])
def test_steps_potential_and_alias_vs_oracle(dims, N, precision, tol, dtol):
    dtype = np.float64 if precision == "f64" else np.float32
    sim = oracle_sim(dims, N, dtype)
    with Evolver(size=N, n_streams=3, dims=dims, precision=precision, batch_streams=2, **TOPHAT) as ev:
        d = ev.derived()
        assert d["transform_passes_mode"] == 0
        assert abs(d["k2_max"] / float(sim.parameters.k2_max) - 1) < 1e-6
        for s in range(3):
            ev.set_psi(s, sim.psi)
        ev.calculate_potential(2, 1)
        assert rel_err(ev.get_potential(2), sim.calculate_potential()) < (1e-12 if precision == "f64" else 2e-4)
        for i in range(10):
            ev.update()
            info = sim.update()
            for st in ev.status():
                assert abs(st.dt / float(info["dt"]) - 1) <= dtol, (i, st.dt, info["dt"])
                assert bool(st.dump) == info["dump"] and st.current_dumps == sim.parameters.current_dumps
                assert abs(st.time / float(info["time"]) - 1) <= dtol
                assert abs(st.max_phi / float(info["max_phi"]) - 1) <= max(dtol, 1e-11)
                assert abs(st.alias_mass - float(info["alias_mass"])) <= max(1e-3 * float(info["alias_mass"]), 1e-12 if precision == "f32" else 1e-20)
        for s in range(3):
            assert rel_err(ev.get_psi(s), sim.psi) < tol
            assert rel_err(ev.get_psik(s), sim.psik) < tol
        ev.calculate_potential(0, 2)
        assert rel_err(ev.get_potential(1), sim.calculate_potential()) < (1e-11 if precision == "f64" else 2e-4)


def test_wigner_streams_and_synth_3d_n20():
    N, ns = 20, 4
    sim0 = oracle_sim(3, N, total_mass=3e16, ntot=1e10)
    dx = 30.0 / N
    sims = []
    base = dict(TOPHAT, total_mass=3e16, ntot=1e10)
    with Evolver(size=N, n_streams=ns, **base) as ev:
        ev.set_psi(0, sim0.psi)
        for s in range(1, ns):
            ev.copy_psi(s, 0)
        for s in range(ns):
            ev.sample(s, "Wigner", s + 1)
            p = o.sample_quantum_perturbation(sim0.psi, dx, 3, 3e16, 3e16 / 1e10, "Wigner", s + 1)
            assert rel_err(ev.get_psi(s), p) < 1e-13
            sims.append(oracle_sim(3, N, psi=p, total_mass=3e16, ntot=1e10))
        for _ in range(3):
            ev.update()
            for sim in sims:
                sim.update()
        for s in range(ns):
            assert rel_err(ev.get_psi(s), sims[s].psi) < 1e-11
            ev.synth_accumulate(s, 0)
        want = o.reference_synthesis([sim.psi for sim in sims], dx ** 3)
        for f in ("psi", "psi2", "psik", "psik2"):
            assert rel_err(ev.synth_fetch(0, f), want[f]) < 1e-11, f
        assert abs(ev.synth_qx(0) - want["Qx"]) < 1e-11


def test_chirpz_and_mixed_radix_agree(monkeypatch):
    """MSM_CHIRPZ=1 sends a 2-3-5-smooth edge through the chirp-z transform as well: both line transforms give the
    same steps to rounding."""
    N = 48
    sim = oracle_sim(3, N)
    out = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("MSM_CHIRPZ", "1")
        with Evolver(size=N, n_streams=1, **TOPHAT) as ev:
            ev.set_psi(0, sim.psi)
            for _ in range(5):
                ev.update()
            out.append((ev.get_psi(0), ev.get_psik(0), ev.status()[0].dt))
    assert rel_err(out[0][0], out[1][0]) < 1e-12 and rel_err(out[0][1], out[1][1]) < 1e-12
    assert abs(out[0][2] / out[1][2] - 1) < 1e-13
    for _ in range(5):
        sim.update()
    assert rel_err(out[1][0], sim.psi) < 1e-11
