"""GPU: parity of the modes and sizes the benchmark configurations run in, against the numpy oracle and the committed
vectors -- single precision in an expanding box (the mode of the 512^3 configuration), the 256^3 engine (two passes
per transform) in static and expanding boxes and both precisions, and one 512^3 single-precision expanding step.

Tolerances (floating point):  fp64 psi <= 1e-11 max|psi|, dt <= 1e-12 relative;
                              fp32 psi <= 2e-4 max|psi|,  dt <= 2e-6 relative (also against the fp64 oracle).
"""
import os

import numpy as np
import pytest

import expanding_host as xh
import sp_oracle as o
from msm_amd import Evolver

pytestmark = pytest.mark.gpu

TOPHAT = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11,
              hbar_=0.05, k2_cutoff=0.95, alias_threshold=0.02)
TOPHAT_ICS = dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0)
COSMO = dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01)
COSMO_BASE = dict(TOPHAT, final_sim_time=0.6, num_data_dumps=3, cfl=0.05)     # tests/golden/make_golden.py: COSMO


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def oracle_sims(base, N, n_streams, dtype=np.float64, expanding=False):
    """The mean-field stream and Husimi streams of seeds 1.. of one configuration, as oracle objects."""
    toml = dict(base, sim_name="t", dims=3, size=N, ics=TOPHAT_ICS, ntot=1e6)
    if expanding:
        toml["cosmology"] = COSMO
    if n_streams > 1:
        toml["sampling"] = dict(seeds=f"1 to {n_streams - 1}", scheme="Husimi")
    params = list(o.simulation_iter(toml, expanding=expanding, dtype=dtype))
    return [o.SimulationObject(p) for p in [params[-1]] + params[:-1]]


def evolver_kwargs(base, expanding):
    kw = dict(base, ntot=1e6)
    if expanding:
        kw["cosmology"] = COSMO
    return kw


@pytest.mark.parametrize("name,precision,tol,dtol", [("tophat_cosmo_n32_steps_f32", "f32", 2e-4, 2e-6),
                                                     ("tophat_cosmo_n16_steps_f64", "f64", 1e-11, 1e-12)])
def test_expanding_golden_steps(name, precision, tol, dtol, golden_dir):
    """examples/spherical-tophat-cosmo.toml under the feature "expanding" (update :669-873), six steps of the
    mean-field stream driven through msm_step_begin / msm_step_end from the vectors' host scalars alone."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    N = g["psi0"].shape[0]
    with Evolver(size=N, n_streams=1, precision=precision, cosmology=COSMO, **COSMO_BASE) as ev:
        ev.set_psi(0, g["psi0"])
        for i in range(1, 7):
            dt, dump, alias, maxphi, time, tau = g["scalars"][i - 1]
            a_now, tau_next, snapped, a_sum, t_new, tau_new = g["host"][i - 1]
            ev.step_begin([a_now], [tau_next])
            st = ev.status()[0]
            assert abs(st.dt / dt - 1) <= dtol and bool(st.dump) == bool(dump), (i, st.dt, dt)
            ev.step_end([a_sum], [t_new], [tau_new], [snapped])
            st = ev.status()[0]
            assert abs(st.time / time - 1) <= dtol and abs(st.tau / tau - 1) <= dtol, (i, st.time, time, st.tau, tau)
            assert abs(st.max_phi / maxphi - 1) <= max(dtol, 1e-12) and st.n_steps == i and st.aliased == 0
            assert abs(st.alias_mass - alias) <= max(1e-3 * alias, 1e-12 if precision == "f32" else 1e-30)
            if f"psi_{i}" in g:
                assert rel_err(ev.get_psi(0), g[f"psi_{i}"]) < tol, i
        assert ev.status()[0].current_dumps == 3 and ev.status()[0].active == 0


@pytest.mark.parametrize("N,precision,tol,dtol", [(32, "f32", 2e-4, 2e-6), (32, "f64", 1e-11, 1e-12), (64, "f32", 2e-4, 2e-6),
                                                  (128, "f64", 1e-11, 1e-12), (128, "f32", 2e-4, 2e-6)])
def test_expanding_streams_vs_oracle(N, precision, tol, dtol):
    """Three streams (mean field + two Husimi seeds, one Poisson pair and one unpaired) in an expanding box, every step
    against the oracle in the same precision: the generic engine at N = 32, 64, the 128^3 two-pass tiling at N = 128."""
    dtype = np.float32 if precision == "f32" else np.float64
    sims = oracle_sims(COSMO_BASE, N, 3, dtype, expanding=True)
    with Evolver(size=N, n_streams=3, precision=precision, **evolver_kwargs(COSMO_BASE, True)) as ev:
        for s, sim in enumerate(sims):
            ev.set_psi(s, sim.psi)
        for i in range(4):
            infos = xh.update(ev, sims)
            st = ev.status()
            for s, info in enumerate(infos):
                assert abs(st[s].dt / float(info["dt"]) - 1) <= dtol, (i, s)
                assert bool(st[s].dump) == info["dump"] and st[s].current_dumps == sims[s].parameters.current_dumps
                assert abs(st[s].time / float(info["time"]) - 1) <= dtol
                assert abs(st[s].tau / float(sims[s].parameters.tau) - 1) <= dtol
        for s, sim in enumerate(sims):
            assert rel_err(ev.get_psi(s), sim.psi) < tol, s
            assert rel_err(ev.get_psik(s), sim.psik) < tol, s


# ---------------------------------------------------------------------------
# 256^3: the two-pass engine against the numpy oracle directly
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def oracle_256_static():
    o.FFT_WORKERS = 8
    try:
        base = dict(TOPHAT, cfl=0.05)
        sims = oracle_sims(base, 256, 2)
        psi0 = [sim.psi.copy() for sim in sims]
        rows = []
        for _ in range(3):
            rows.append([sim.update() for sim in sims])
        return base, psi0, rows, [sim.psi for sim in sims], [sim.psik for sim in sims], [sim.calculate_potential() for sim in sims]
    finally:
        o.FFT_WORKERS = 0


@pytest.mark.parametrize("precision,tol,dtol", [("f64", 1e-11, 1e-12), ("f32", 2e-4, 2e-6)])
def test_two_pass_256_static_vs_numpy_oracle(oracle_256_static, precision, tol, dtol):
    """Three static steps of a Poisson pair (mean field + one Husimi stream) at 256^3: dt, max|phi|, time per step,
    psi, psi_k and the potential afterwards; single precision against the same double-precision oracle run."""
    base, psi0, rows, psi, psik, phi = oracle_256_static
    with Evolver(size=256, n_streams=2, precision=precision, **evolver_kwargs(base, False)) as ev:
        assert ev.derived()["transform_passes_mode"] == 2.0
        for s in range(2):
            ev.set_psi(s, psi0[s])
        for i in range(3):
            ev.update()
            st = ev.status()
            for s in range(2):
                info = rows[i][s]
                assert abs(st[s].dt / float(info["dt"]) - 1) <= dtol, (i, s, st[s].dt, info["dt"])
                assert bool(st[s].dump) == info["dump"] and abs(st[s].time / float(info["time"]) - 1) <= dtol
                assert abs(st[s].max_phi / float(info["max_phi"]) - 1) <= max(dtol, 1e-11)
                assert st[s].aliased == 0
        for s in range(2):
            assert rel_err(ev.get_psi(s), psi[s]) < tol, s
            assert rel_err(ev.get_psik(s), psik[s]) < tol, s
        ev.calculate_potential(0, 2)
        for s in range(2):
            assert rel_err(ev.get_potential(s), phi[s]) < (1e-11 if precision == "f64" else 2e-3), s


def test_benchmark_workload_256_vs_numpy_oracle():
    """The headline benchmark's own inputs at its own size (bench.py: plane-wave ICs of sim.py regenerated at N = 256,
    tomls/planeWave3d_e10_sym.toml parameters, Wigner noise per stream seed): the sampled psi_0 of two seeds against the
    oracle's sampler (ics.rs:560-602), then two steps of both streams against the numpy oracle."""
    from msm_amd.ics import plane_wave_psi
    PLANE = dict(axis_length=60.0, final_sim_time=2000.0, cfl=0.1, num_data_dumps=64, total_mass=3e16,
                 hbar_=0.01, ntot=1e10, k2_cutoff=0.95, alias_threshold=0.001)      # = bench.py PLANE
    N, seeds = 256, (1, 2)
    psi0 = plane_wave_psi(N)
    assert rel_err(psi0, o.plane_wave_ics(N)) < 1e-13     # two restatements of sim.py:124-183, equal to rounding
    o.FFT_WORKERS = 8
    try:
        toml = dict(PLANE, sim_name="t", dims=3, size=N, ics=TOPHAT_ICS)       # ics unused: psi is handed over below
        params = [list(o.simulation_iter(toml))[-1] for _ in seeds]
        want0 = [o.sample_quantum_perturbation(psi0, 60.0 / N, 3, PLANE["total_mass"], PLANE["total_mass"] / PLANE["ntot"],
                                               "Wigner", seed) for seed in seeds]
        sims = [o.SimulationObject(p, psi=w) for p, w in zip(params, want0)]
        with Evolver(size=N, n_streams=2, **PLANE) as ev:
            assert ev.derived()["transform_passes_mode"] == 2.0
            ev.set_psi(0, psi0)
            ev.copy_psi(1, 0)
            for s, seed in enumerate(seeds):
                ev.sample(s, "Wigner", seed)
            for s in range(2):
                noise = np.max(np.abs(want0[s] - psi0))
                assert noise > 0 and np.max(np.abs(ev.get_psi(s) - want0[s])) < 1e-12 * noise + 1e-15 * np.max(np.abs(psi0)), s
            for i in range(2):
                ev.update()
                st = ev.status()
                for s, sim in enumerate(sims):
                    info = sim.update()
                    assert abs(st[s].dt / float(info["dt"]) - 1) <= 1e-12, (i, s, st[s].dt, info["dt"])
                    assert bool(st[s].dump) == info["dump"] and abs(st[s].time / float(info["time"]) - 1) <= 1e-12
                    assert abs(st[s].max_phi / float(info["max_phi"]) - 1) <= 1e-11 and st[s].aliased == 0
                    assert abs(st[s].alias_mass - float(info["alias_mass"])) <= max(1e-4 * float(info["alias_mass"]), 1e-20)
            assert st[0].dt != st[1].dt                   # two seeds, two trajectories
            for s, sim in enumerate(sims):
                assert rel_err(ev.get_psi(s), sim.psi) < 1e-11, s
                assert rel_err(ev.get_psik(s), sim.psik) < 1e-11, s
    finally:
        o.FFT_WORKERS = 0


@pytest.fixture(scope="module")
def oracle_256_expanding():
    o.FFT_WORKERS = 8
    try:
        sims = oracle_sims(COSMO_BASE, 256, 2, expanding=True)
        psi0 = [sim.psi.copy() for sim in sims]
        rows = []
        for _ in range(2):
            row = []
            for sim in sims:
                pre = xh.step_inputs(sim)
                import copy
                shadow = copy.deepcopy(sim)
                dtau = float(shadow.update()["dt"])
                del shadow
                post = xh.half_kicks(sim, dtau)
                info = sim.update()
                row.append((pre, post, info, float(sim.parameters.tau)))
            rows.append(row)
        return psi0, rows, [sim.psi for sim in sims], [sim.psik for sim in sims]
    finally:
        o.FFT_WORKERS = 0


@pytest.mark.parametrize("precision,tol,dtol", [("f64", 1e-11, 1e-12), ("f32", 2e-4, 2e-6)])
def test_two_pass_256_expanding_vs_numpy_oracle(oracle_256_expanding, precision, tol, dtol):
    """Two expanding steps (update :669-873) of a pair at 256^3 on the two-pass engine, host scalars from the oracle."""
    psi0, rows, psi, psik = oracle_256_expanding
    with Evolver(size=256, n_streams=2, precision=precision, **evolver_kwargs(COSMO_BASE, True)) as ev:
        assert ev.derived()["transform_passes_mode"] == 2.0
        for s in range(2):
            ev.set_psi(s, psi0[s])
        for row in rows:
            ev.step_begin([r[0][0] for r in row], [r[0][1] for r in row])
            st = ev.status()
            for s, r in enumerate(row):
                assert abs(st[s].dt / float(r[2]["dt"]) - 1) <= dtol and bool(st[s].dump) == r[2]["dump"]
            ev.step_end([r[1][0] for r in row], [r[1][1] for r in row], [r[1][2] for r in row], [r[0][2] for r in row])
            st = ev.status()
            for s, r in enumerate(row):
                assert abs(st[s].time / float(r[2]["time"]) - 1) <= dtol and abs(st[s].tau / r[3] - 1) <= dtol
                assert abs(st[s].max_phi / float(r[2]["max_phi"]) - 1) <= max(dtol, 1e-11)
        for s in range(2):
            assert rel_err(ev.get_psi(s), psi[s]) < tol, s
            assert rel_err(ev.get_psik(s), psik[s]) < tol, s


@pytest.mark.parametrize("two_pass", [0, 2], ids=["three-pass", "two-pass"])
def test_512_f32_expanding_step_vs_oracle(two_pass):
    """One expanding step of one stream at 512^3 in single precision (the per-GPU grid of the largest benchmark
    configuration) against the oracle run in the same precision, on the generic three-pass engine (MSM_TWO_PASS=0) and
    on the two-pass tiling (the default at this size in single precision)."""
    N = 512
    o.FFT_WORKERS = 8
    os.environ["MSM_TWO_PASS"] = str(two_pass)
    try:
        sims = oracle_sims(COSMO_BASE, N, 1, np.float32, expanding=True)
        with Evolver(size=N, n_streams=1, precision="f32", **evolver_kwargs(COSMO_BASE, True)) as ev:
            ev.set_psi(0, sims[0].psi)
            infos = xh.update(ev, sims)
            st = ev.status()[0]
            assert abs(st.dt / float(infos[0]["dt"]) - 1) <= 2e-6 and bool(st.dump) == infos[0]["dump"]
            assert abs(st.max_phi / float(infos[0]["max_phi"]) - 1) <= 1e-4
            assert abs(st.time / float(infos[0]["time"]) - 1) <= 2e-6 and st.aliased == 0
            assert ev.derived()["transform_passes_mode"] == float(two_pass)
            assert rel_err(ev.get_psi(0), sims[0].psi) < 2e-4
            assert rel_err(ev.get_psik(0), sims[0].psik) < 2e-4
    finally:
        o.FFT_WORKERS = 0
        os.environ.pop("MSM_TWO_PASS", None)


def test_512_f64_mean_field_example_step_vs_oracle():
    """examples/gaussian-overdensity-mft.toml as shipped -- size 512, double precision (the reference binary's
    `type T = f64`, simulator/src/main.rs:19), ColdGauss ICs, the mean-field stream alone -- two steps on the generic
    three-pass engine against the numpy oracle: dt, max|phi|, time, alias mass, psi and psi_k."""
    N = 512
    o.FFT_WORKERS = 16
    try:
        base = dict(axis_length=30.0, final_sim_time=400.0, cfl=0.02, num_data_dumps=200, total_mass=1e10, hbar_=0.02,
                    k2_cutoff=0.95, alias_threshold=0.02)
        toml = dict(base, sim_name="gaussian-overdensity-512-mft", dims=3, size=N,
                    ics=dict(type="ColdGauss", mean=[15.0, 15.0, 15.0], std=[10.0, 10.0, 10.0]))
        sim = o.SimulationObject(list(o.simulation_iter(toml))[-1])
        with Evolver(size=N, n_streams=1, **base) as ev:
            assert ev.derived()["transform_passes_mode"] == 0.0          # the generic engine
            ev.set_psi(0, sim.psi)
            for i in range(2):
                ev.update()
                info = sim.update()
                st = ev.status()[0]
                assert abs(st.dt / float(info["dt"]) - 1) <= 1e-12 and bool(st.dump) == info["dump"], (i, st.dt, info["dt"])
                assert abs(st.time / float(info["time"]) - 1) <= 1e-12 and st.aliased == 0
                assert abs(st.max_phi / float(info["max_phi"]) - 1) <= 1e-11
                assert abs(st.alias_mass - float(info["alias_mass"])) <= 1e-9 * max(float(info["alias_mass"]), 1e-30) + 1e-25
            assert rel_err(ev.get_psi(0), sim.psi) < 1e-11
            assert rel_err(ev.get_psik(0), sim.psik) < 1e-11
    finally:
        o.FFT_WORKERS = 0


def _run_512(mode, cosmology, batch=2):
    """Three single-precision streams at 512^3 in sub-batches of two (a pair and an unpaired stream) or in one sub-batch
    (a pair and an unpaired stream side by side), two steps."""
    N = 512
    os.environ["MSM_TWO_PASS"] = str(mode)
    try:
        base = dict(TOPHAT, cfl=0.05)
        with Evolver(size=N, n_streams=3, batch_streams=batch, cosmology=cosmology, precision="f32", **base) as ev:
            assert ev.derived()["transform_passes_mode"] == float(mode)
            rng = np.random.default_rng(7)
            for s in range(3):
                psi = (1.0 + 0.02 * rng.standard_normal((N, N, N), dtype=np.float32)
                       + 0.02j * rng.standard_normal((N, N, N), dtype=np.float32)).astype(np.complex64)
                psi *= np.float32(1.0 / np.sqrt(np.sum(np.abs(psi).astype(np.float64) ** 2) * ev.derived()["dx"] ** 3))
                ev.set_psi(s, psi)
            rows = []
            for i in range(2):
                if cosmology:
                    ev.step_begin([0.5] * 3, [1e9] * 3)
                    st = ev.status()
                    ev.step_end([1.0] * 3, [st[s].time + st[s].dt for s in range(3)],
                                [st[s].tau + st[s].dt for s in range(3)], [0.0] * 3)
                else:
                    ev.update()
                rows.append([(r.dt, r.time, r.max_phi, r.alias_mass, r.active, r.aliased, r.n_steps) for r in ev.status()])
            psi = [ev.get_psi(s) for s in range(3)]
            psik = ev.get_psik(2)
            ev.calculate_potential(0, 2)
            phi = [ev.get_potential(s) for s in range(2)]
        return rows, psi, psik, phi
    finally:
        os.environ.pop("MSM_TWO_PASS", None)


@pytest.mark.parametrize("cosmology,batch", [(None, 2), (COSMO, 2), (None, 4)], ids=["static", "expanding", "static-one-batch"])
def test_two_pass_engine_matches_three_pass_512_f32(cosmology, batch):
    """The single-precision 512^3 tiling (two passes per transform) against the generic engine (three passes), which the
    oracle tests pin at every size: same steps and flags, fields equal to single-precision rounding.  The sub-batch
    shapes cover every way the tiling forms its densities: a whole pair (packed in registers by the second stream's
    pass), an unpaired stream alone, and an unpaired stream behind a pair in one launch."""
    r3, psi3, psik3, phi3 = _run_512(0, cosmology, batch)
    r2, psi2, psik2, phi2 = _run_512(2, cosmology, batch)
    for a, b in zip(r3, r2):
        for x, y in zip(a, b):
            assert x[4:] == y[4:]
            assert abs(x[0] / y[0] - 1) < 2e-5 and abs(x[1] / y[1] - 1) < 2e-5
            assert abs(x[2] / y[2] - 1) < 5e-3          # max|phi| of a nearly uniform density is ill-conditioned
    errs = [rel_err(psi2[s], psi3[s]) for s in range(3)]
    if max(errs) >= 2e-5:
        # Both engines are bitwise reproducible (no atomics, fixed reduction orders), so a second run of each says what
        # this is: engines that reproduce themselves and still differ fail here; a run that its own engine does not
        # reproduce was a one-off on this machine -- reported as a warning, and the comparison is made on the second pair.
        import warnings
        r3b, psi3b, psik3b, phi3b = _run_512(0, cosmology, batch)
        r2b, psi2b, psik2b, phi2b = _run_512(2, cosmology, batch)
        again3 = [float(np.max(np.abs(psi3b[s] - psi3[s]))) for s in range(3)]
        again2 = [float(np.max(np.abs(psi2b[s] - psi2[s]))) for s in range(3)]
        what = (f"engines differ by {errs}; three-pass engine against its own second run {again3}, "
                f"tiling against its own second run {again2}")
        assert max(again3) > 0 or max(again2) > 0, what
        warnings.warn("non-reproducible first run: " + what)
        psi3, psi2, psik3, psik2, phi3, phi2 = psi3b, psi2b, psik3b, psik2b, phi3b, phi2b
        for s in range(3):
            assert rel_err(psi2[s], psi3[s]) < 2e-5, (s, what)
    assert rel_err(psik2, psik3) < 2e-5
    for s in range(2):
        assert rel_err(phi2[s], phi3[s]) < 1e-2, s
