"""GPU: parity of the HIP path (through the C ABI) against the numpy oracle and
the committed golden vectors.

Stated tolerances (floating point, so not bit-exact):
  fp64  FFT / potential      <= 2e-13 * max|value|
  fp64  psi after k steps    <= 1e-11 * max|psi|      (k <= 10)
  fp64  dt                   <= 1e-12 relative, identical dump flags
  fp32  psi after k steps    <= 2e-4  * max|psi|
"""
import os

import numpy as np
import pytest

import sp_oracle as o
from msm_amd import Evolver, MsmError

pytestmark = pytest.mark.gpu

TOPHAT = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11,
              hbar_=0.05, k2_cutoff=0.95, alias_threshold=0.02)
TOPHAT_ICS = dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0)
PLANE = dict(axis_length=60.0, final_sim_time=2000.0, cfl=0.1, num_data_dumps=64, total_mass=3e16,
             hbar_=0.01, ntot=1e10, k2_cutoff=0.95, alias_threshold=0.001)


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def rand_psi(N, seed, cdt=np.complex128):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((N, N, N)) + 1j * rng.standard_normal((N, N, N))).astype(cdt)


def oracle_sim(base, N, ics=None, psi=None, dtype=np.float64, sampling=None):
    toml = dict(base, sim_name="t", dims=3, size=N, ics=ics or TOPHAT_ICS)
    params = list(o.simulation_iter(toml, dtype=dtype))[-1]
    params.sampling = sampling
    return o.SimulationObject(params, psi=psi)


# ---------------------------------------------------------------------------
# FFT wrappers (reference: simulator/tests/fft.rs, simulator/src/utils/fft.rs)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("N", [2, 8])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_fft_roundtrip_constant(N, precision):
    # tests/fft.rs: constant 0+2i arrays, forward then inverse, sum |x - y| <= 1e-6, unitarity
    with Evolver(size=N, n_streams=1, precision=precision, **TOPHAT) as ev:
        values = np.full((N, N, N), 2j, dtype=ev.cplx)
        ev.set_psi(0, values)
        ev.forward(0)
        k = ev.get_psi(0)
        assert abs(np.sum(np.abs(k) ** 2) - np.sum(np.abs(values) ** 2)) <= 1e-5 * np.sum(np.abs(values) ** 2)
        assert abs(k[0, 0, 0] - 2j * N ** 1.5) <= 1e-5 * N ** 1.5   # all mass in the k = 0 mode
        ev.inverse(0)
        assert np.sum(np.abs(ev.get_psi(0) - values)) <= 1e-6 * N ** 3


@pytest.mark.parametrize("N", [2, 4, 8, 16, 32, 64, 128, 256])
def test_fft_vs_oracle_f64(N):
    psi = rand_psi(N, 100 + N)
    with Evolver(size=N, n_streams=2, **TOPHAT) as ev:
        ev.set_psi(1, psi)
        ev.forward(1)
        assert rel_err(ev.get_psi(1), o.forward(psi)) < 2e-13
        ev.inverse(1)
        assert rel_err(ev.get_psi(1), psi) < 2e-13
        ev.inverse(1)
        assert rel_err(ev.get_psi(1), o.inverse(psi)) < 2e-13


@pytest.mark.parametrize("N", [16, 64, 128, 512])
def test_fft_vs_oracle_f32(N):
    psi = rand_psi(N, 7 + N, np.complex64)
    with Evolver(size=N, n_streams=1, precision="f32", **TOPHAT) as ev:
        ev.set_psi(0, psi)
        ev.forward(0)
        want = o.forward(psi.astype(np.complex128))
        assert rel_err(ev.get_psi(0), want) < 5e-6
        ev.inverse(0)
        assert rel_err(ev.get_psi(0), psi) < 5e-6


def test_fft_512_f64():
    N = 512
    psi = rand_psi(N, 3)
    with Evolver(size=N, n_streams=1, **TOPHAT) as ev:
        ev.set_psi(0, psi)
        ev.forward(0)
        got = ev.get_psi(0)
        assert rel_err(got, o.forward(psi)) < 3e-13
        # linearity + shift property need no oracle: F[psi rolled by 1 along x] = F[psi] e^{-2 pi i kx / N}
        ev.set_psi(0, np.roll(psi, 1, axis=2))
        ev.forward(0)
        k = np.exp(-2j * np.pi * np.fft.fftfreq(N) * 1.0)[None, None, :]
        assert rel_err(ev.get_psi(0), got * k) < 3e-13


# ---------------------------------------------------------------------------
# calculate_potential (simulation_object.rs:1066-1110)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("N,nstreams", [(8, 1), (16, 3), (32, 2), (64, 5), (128, 2)])
def test_potential_vs_oracle(N, nstreams):
    with Evolver(size=N, n_streams=nstreams, **TOPHAT) as ev:
        sims = []
        for s in range(nstreams):
            psi = o.normalize(rand_psi(N, 10 * N + s) + 3.0, 30.0 / N, 3)
            ev.set_psi(s, psi)
            sims.append(oracle_sim(TOPHAT, N, psi=psi))
        ev.calculate_potential(0, nstreams)
        st = ev.status()
        for s in range(nstreams):
            want = sims[s].calculate_potential()
            assert rel_err(ev.get_potential(s), want) < 2e-13, s
            assert abs(st[s].max_phi / np.max(np.abs(want)) - 1) < 1e-13
        # psi itself is untouched
        assert np.array_equal(ev.get_psi(0), sims[0].psi)


def test_potential_k0_mode_is_zeroed():
    # uniform density has only the k = 0 mode: phi must be exactly 0 (simulation_object.rs:1097-1102)
    N = 16
    with Evolver(size=N, n_streams=1, **TOPHAT) as ev:
        ev.set_psi(0, np.full((N, N, N), 0.25 + 0.5j))
        ev.calculate_potential(0, 1)
        phi = ev.get_potential(0)
        assert np.all(np.isfinite(phi)) and np.max(np.abs(phi)) < 1e-20


# ---------------------------------------------------------------------------
# update (simulation_object.rs:475-661) against the oracle and the goldens
# ---------------------------------------------------------------------------
def check_steps(ev, stream, sim, nsteps, tol, dt_tol=1e-12):
    for i in range(nsteps):
        ev.update()
        info = sim.update()
        st = ev.status()[stream]
        assert abs(st.dt / float(info["dt"]) - 1) <= dt_tol, (i, st.dt, info["dt"])
        assert bool(st.dump) == info["dump"], i
        assert st.current_dumps == sim.parameters.current_dumps
        assert abs(st.time / float(info["time"]) - 1) <= dt_tol
        assert abs(st.max_phi / float(info["max_phi"]) - 1) <= max(dt_tol, 1e-12)
        assert st.n_steps == i + 1 and st.aliased == 0
    got = ev.get_psi(stream)
    assert rel_err(got, sim.psi) < tol
    assert rel_err(ev.get_psik(stream), sim.psik) < tol
    return got


@pytest.mark.parametrize("N", [8, 16, 32, 64])
def test_static_steps_tophat_f64(N):
    sim = oracle_sim(TOPHAT, N)
    with Evolver(size=N, n_streams=1, **TOPHAT) as ev:
        ev.set_psi(0, sim.psi)
        check_steps(ev, 0, sim, 10, 1e-11)
        # alias mass: tiny numbers, compare at the scale of the total mass (1.0)
        assert abs(ev.status()[0].alias_mass - float(sim.check_alias()[0])) < 1e-20


def test_static_steps_plain_launches(monkeypatch):
    """Small grids replay their step from a captured HIP graph; MSM_GRAPH=0 takes the plain launch path with the same
    kernels.  Both must follow the oracle, so that a change to the capture cannot hide a launch-path bug."""
    monkeypatch.setenv("MSM_GRAPH", "0")
    N = 16
    sim = oracle_sim(TOPHAT, N)
    with Evolver(size=N, n_streams=1, **TOPHAT) as ev:
        ev.set_psi(0, sim.psi)
        check_steps(ev, 0, sim, 10, 1e-11)
        assert abs(ev.status()[0].alias_mass - float(sim.check_alias()[0])) < 1e-20


def test_static_steps_128_f64():
    N = 128
    sim = oracle_sim(dict(TOPHAT, cfl=0.05), N)   # small cfl: dt limited by the kinetic bound, not the dump
    with Evolver(size=N, n_streams=1, **dict(TOPHAT, cfl=0.05)) as ev:
        ev.set_psi(0, sim.psi)
        check_steps(ev, 0, sim, 3, 1e-11)


@pytest.mark.parametrize("name", ["tophat_n8_f64", "tophat_n16_f64", "planewave_n16_mft_f64",
                                  "planewave_n16_wigner1_f64", "tophat_n16_f32", "tophat_n32_f64", "tophat_n12_f64"])
def test_golden_vectors(name, golden_dir):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    base = TOPHAT if name.startswith("tophat") else PLANE
    f32 = name.endswith("f32")
    N = g["psi0"].shape[0]
    tol, dtol = (2e-4, 2e-6) if f32 else (1e-11, 1e-12)
    with Evolver(size=N, n_streams=1, precision="f32" if f32 else "f64", **base) as ev:
        ev.set_psi(0, g["psi0"])
        for i in range(1, 11):
            ev.update()
            st = ev.status()[0]
            dt, dump, alias, maxphi, time = g["scalars"][i - 1]
            assert abs(st.dt / dt - 1) <= dtol and bool(st.dump) == bool(dump)
            assert abs(st.time / time - 1) <= dtol and abs(st.max_phi / maxphi - 1) <= dtol
            assert abs(st.alias_mass - alias) <= max(1e-4 * alias, 1e-12 if f32 else 1e-20)
            if f"psi_{i}" in g:
                assert rel_err(ev.get_psi(0), g[f"psi_{i}"]) < tol, i
                ev.calculate_potential(0, 1)    # as dump() does with output_potential (simulation_object.rs:1167-1168)
                assert rel_err(ev.get_potential(0), g[f"phi_{i}"]) < (2e-4 if f32 else 1e-11), i


def test_streams_are_independent_of_batching():
    # 5 streams, batch of 2 (pairs + a ragged last batch): every stream matches its own
    # single-stream oracle run, and its own run alone on the device, bit for bit
    N = 16
    sims, psis = [], []
    base = oracle_sim(TOPHAT, N).psi
    for s in range(5):
        psi = o.sample_quantum_perturbation(base, 30.0 / N, 3, 1e11, 1e3, "Husimi", seed=s + 1)
        psis.append(psi)
        sims.append(oracle_sim(TOPHAT, N, psi=psi))
    with Evolver(size=N, n_streams=5, batch_streams=2, **TOPHAT) as ev:
        for s in range(5):
            ev.set_psi(s, psis[s])
        for _ in range(4):
            ev.update()
        outs = [ev.get_psi(s) for s in range(5)]
    for s in range(5):
        for _ in range(4):
            sims[s].update()
        assert rel_err(outs[s], sims[s].psi) < 1e-11
    with Evolver(size=N, n_streams=1, **TOPHAT) as ev:
        ev.set_psi(0, psis[3])
        for _ in range(4):
            ev.update()
        alone = ev.get_psi(0)
    # stream 3 shared a Poisson pair with stream 2 above and ran unpaired here: the packed transform
    # differs only at rounding level
    assert rel_err(alone, outs[3]) < 1e-13


def test_dump_schedule_and_finish():
    # examples/spherical-tophat.toml at N = 8 with T_f shortened: every step is a dump step,
    # time snaps to k T_f / n (simulation_object.rs:620-631) and the stream goes inactive at T_f
    N = 8
    cfgd = dict(TOPHAT, final_sim_time=2.0, num_data_dumps=10)
    sim = oracle_sim(cfgd, N)
    with Evolver(size=N, n_streams=1, **cfgd) as ev:
        ev.set_psi(0, sim.psi)
        steps = 0
        while sim.not_finished():
            sim.update()
            ev.update()
            steps += 1
            st = ev.status()[0]
            assert st.current_dumps == sim.parameters.current_dumps
            assert st.time == float(sim.parameters.time)
            assert bool(st.active) == sim.not_finished()
        assert steps == 10 and st.current_dumps == 10 and st.time == 2.0
        before = ev.get_psi(0)
        ev.update()                      # finished streams are left alone
        assert ev.status()[0].n_steps == 10
        assert np.array_equal(ev.get_psi(0), before)


def test_alias_detection():
    # a wavefunction with all its mass at the Nyquist corner trips check_alias
    # (simulation_object.rs:1249-1293); the library reports it instead of panicking
    N = 8
    idx = np.indices((N, N, N)).sum(axis=0)
    psi = o.normalize(((-1.0) ** idx).astype(np.complex128), 30.0 / N, 3)
    sim = oracle_sim(TOPHAT, N, psi=psi)
    with pytest.raises(o.AliasError):
        sim.update()
    with Evolver(size=N, n_streams=1, **TOPHAT) as ev:
        ev.set_psi(0, psi)
        ev.update()
        st = ev.status()[0]
        assert st.aliased == 1 and st.active == 0 and abs(st.alias_mass - 1.0) < 1e-12


# ---------------------------------------------------------------------------
# sampling (ics.rs:434-648) -- same Philox/Box-Muller spec as the oracle
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("scheme", ["Wigner", "Husimi"])
def test_sampling_matches_oracle_spec(scheme, golden_dir):
    N = 16
    psi0 = o.user_specified_ics(os.path.join(golden_dir, "planeWave3d_e10_sym.npz"), 3, N)
    with Evolver(size=N, n_streams=3, **PLANE) as ev:
        ev.set_psi(0, psi0)
        ev.copy_psi(2, 0)
        ev.sample(2, scheme, 12345678901)
        got = ev.get_psi(2)
        want = o.sample_quantum_perturbation(psi0, 60.0 / N, 3, 3e16, 3e16 / 1e10, scheme, 12345678901)
        noise = np.max(np.abs(want - psi0))
        assert np.max(np.abs(got - want)) < 1e-12 * noise + 1e-15 * np.max(np.abs(psi0))
        assert np.array_equal(ev.get_psi(0), psi0)
        with pytest.raises(MsmError):
            ev._check(ev._lib.msm_sample(ev._ctx, 1, 7, 1), "msm_sample")    # unknown scheme code


def test_error_conventions():
    with pytest.raises(MsmError):
        Evolver(size=25, n_streams=1, **TOPHAT)          # odd edge: get_kgrid asserts size % 2 == 0 (utils/fft.rs:105)
    with pytest.raises(MsmError):
        Evolver(size=2048, n_streams=1, dims=1, **TOPHAT)  # beyond the largest line transform
    with pytest.raises(MsmError):
        Evolver(size=16, n_streams=1, dims=4, **TOPHAT)  # dims is 1, 2 or 3
    with Evolver(size=8, n_streams=2, **TOPHAT) as ev:
        with pytest.raises(MsmError):
            ev.set_psi(5, np.zeros((8, 8, 8), complex))
        with pytest.raises(MsmError):
            ev.get_psik(0)                                # psi_k undefined before the first step
        with pytest.raises(MsmError):
            ev.get_potential(1)                           # no potential resident yet
        with pytest.raises(ValueError):
            ev.set_psi(0, np.zeros((4, 4, 4), complex))


# ---------------------------------------------------------------------------
# full-size, oracle-free properties (256^3, the grid of the headline benchmark)
# ---------------------------------------------------------------------------
def test_full_size_properties_256():
    N = 256
    psi = o.normalize(rand_psi(N, 1) * 0.05 + 1.0, 30.0 / N, 3)
    with Evolver(size=N, n_streams=2, **dict(TOPHAT, cfl=0.05)) as ev:
        d = ev.derived()
        ev.set_psi(0, psi)
        ev.copy_psi(1, 0)
        # Parseval + round trip
        ev.forward(0)
        k = ev.get_psi(0)
        assert abs(np.sum(np.abs(k) ** 2) / np.sum(np.abs(psi) ** 2) - 1) < 1e-13
        ev.inverse(0)
        assert rel_err(ev.get_psi(0), psi) < 2e-13
        # Poisson solve is consistent with its definition: -k^2 phi_k = C rho_k, zero mean
        ev.calculate_potential(0, 2)
        phi = ev.get_potential(1)
        assert abs(phi.mean()) < 1e-13 * np.abs(phi).max()
        rho = 1e11 * np.abs(psi) ** 2
        k1 = 2 * np.pi * np.fft.fftfreq(N, d=30.0 / N)
        k2 = k1[:, None, None] ** 2 + k1[None, :, None] ** 2 + k1[None, None, :] ** 2
        lap = np.fft.ifftn(-k2 * np.fft.fftn(phi)).real
        want = o.POIS_CONST * (rho - rho.mean())
        assert np.max(np.abs(lap - want)) < 1e-10 * np.max(np.abs(want))
        # two identical streams stay identical and the step is unitary
        for _ in range(2):
            ev.update()
        a, b = ev.get_psi(0), ev.get_psi(1)
        assert rel_err(a, b) < 1e-13
        assert abs(np.sum(np.abs(a) ** 2) * d["dx"] ** 3 - 1) < 1e-12
        st = ev.status()
        assert st[0].dt == st[1].dt and st[0].dt <= d["kinetic_dt"]


def test_steps_and_potential_256_vs_cpu_port():
    """256^3 (the size the two-pass Poisson path serves): three streams -- one pair and one
    unpaired -- against the C restatement of the reference step (oracle/sp_cpu.c)."""
    import sp_cpu
    N, base = 256, dict(TOPHAT, cfl=0.05)
    psis = [o.normalize(rand_psi(N, 10 + s) * 0.05 + 1.0, 30.0 / N, 3) for s in range(3)]
    sims = []
    for s in range(3):
        sim = sp_cpu.CpuSim(N, **base)
        sim.set_psi(psis[s])
        sims.append(sim)
    with Evolver(size=N, n_streams=3, **base) as ev:
        for s in range(3):
            ev.set_psi(s, psis[s])
        ev.calculate_potential(0, 3)
        for s in range(3):
            pm = sims[s].potential()
            assert rel_err(ev.get_potential(s), sims[s].get_phi()) < 1e-12
            assert abs(ev.status()[s].max_phi / pm - 1) < 1e-12
        for i in range(2):
            ev.update()
            infos = [sim.update() for sim in sims]
            st = ev.status()
            for s in range(3):
                assert abs(st[s].dt / infos[s]["dt"] - 1) < 1e-12
                assert abs(st[s].max_phi / infos[s]["max_phi"] - 1) < 1e-11
                assert abs(st[s].time / infos[s]["time"] - 1) < 1e-12
        for s in range(3):
            assert rel_err(ev.get_psi(s), sims[s].get_psi()) < 1e-11
            assert rel_err(ev.get_psik(s), sims[s].get_psik()) < 1e-11
        ev.calculate_potential(0, 3)
        for s in range(3):
            sims[s].potential()
            assert rel_err(ev.get_potential(s), sims[s].get_phi()) < 1e-11
    for sim in sims:
        sim.close()


def _run_mode_256(mode, cosmology=None, precision="f64", N=256):
    """Five streams in sub-batches of four (one unpaired), stream 1 carrying grid-scale noise so that
    it aliases on the first step and sits out the rest; returns status rows and fields."""
    os.environ["MSM_TWO_PASS"] = str(mode)
    try:
        base = dict(TOPHAT, cfl=0.05, alias_threshold=1e-4, k2_cutoff=0.5)
        with Evolver(size=N, n_streams=5, batch_streams=4, cosmology=cosmology, precision=precision, **base) as ev:
            assert ev.derived()["transform_passes_mode"] == float(mode)
            for s in range(5):
                psi = rand_psi(N, 20 + s) * (0.5 if s == 1 else 1e-3) + 1.0
                ev.set_psi(s, o.normalize(psi, 30.0 / N, 3).astype(ev.cplx))
            rows = []
            for i in range(3):
                if cosmology:
                    ev.step_begin([0.5] * 5, [1e9] * 5)
                    st = ev.status()
                    ev.step_end([0.5] * 5, [st[s].time + st[s].dt for s in range(5)],
                                [st[s].tau + st[s].dt for s in range(5)], [0.0] * 5)
                else:
                    ev.update()
                rows.append([(r.dt, r.time, r.max_phi, r.alias_mass, r.active, r.aliased, r.n_steps) for r in ev.status()])
            psi = [ev.get_psi(s) for s in range(5)]
            psik = [ev.get_psik(s) for s in range(5)]
            ev.calculate_potential(0, 4)
            phi = [ev.get_potential(s) for s in range(4)]
        return rows, psi, psik, phi
    finally:
        os.environ.pop("MSM_TWO_PASS", None)


@pytest.mark.parametrize("N", [256, 128])
@pytest.mark.parametrize("cosmology", [None, dict(omega_matter_now=1.0, omega_radiation_now=0.0, h=0.7, z0=9.0, max_dloga=0.01)],
                         ids=["static", "expanding"])
@pytest.mark.parametrize("precision,tol,stol", [("f64", 1e-11, 1e-11), ("f32", 2e-4, 2e-5)])
def test_two_pass_engine_matches_three_pass_256(cosmology, precision, tol, stol, N):
    """The two-pass engines (256^3 and 128^3: blocked work-array layouts, state kept as the pass-A image of psi) against
    the plain three-passes-per-transform engine that the small-N oracle tests pin: same steps, same
    activity flags, fields equal to rounding -- with an unpaired stream and a stream that drops out."""
    r3, psi3, psik3, phi3 = _run_mode_256(0, cosmology, precision, N)
    r2, psi2, psik2, phi2 = _run_mode_256(2, cosmology, precision, N)
    assert [row[1][4] for row in r3] == [0, 0, 0] and r3[0][1][5] == 1       # stream 1 aliased at step 1
    for a, b in zip(r3, r2):
        for x, y in zip(a, b):
            assert x[4:] == y[4:]
            assert abs(x[0] / y[0] - 1) < stol and abs(x[1] / y[1] - 1) < stol
            # phi of a nearly uniform density (1 + 1e-3 noise) is a difference of nearly equal numbers: the relative
            # rounding error of max|phi| is ~1e3 times that of psi
            assert abs(x[2] / y[2] - 1) < (5 * stol if precision == "f64" else 5e-3)
            assert abs(x[3] - y[3]) <= (1e-9 if precision == "f64" else 1e-3) * max(abs(x[3]), 1e-30) + (1e-22 if precision == "f64" else 1e-12)
    for s in range(5):
        assert rel_err(psi2[s], psi3[s]) < tol
        assert rel_err(psik2[s], psik3[s]) < tol
    for s in range(4):
        assert rel_err(phi2[s], phi3[s]) < (5 * tol if precision == "f64" else 1e-2)     # ill-conditioned, as max|phi| above


def test_config5_shape_512_f32_expanding():
    """512^3 fp32, expanding box (the shape of the largest benchmark configuration, one pair of streams): runs on the
    generic three-pass engine.  Oracle-free properties: Parseval and round trip at fp32 accuracy,
    the two streams of a pair (real / imaginary lane of one Poisson solve) stay equal to fp32 rounding,
    the step conserves the norm, dt obeys the kinetic bound."""
    N = 512
    cosmo = dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01)
    rng = np.random.default_rng(5)
    psi = (1.0 + 0.05 * rng.standard_normal((N, N, N), dtype=np.float32)
           + 0.05j * rng.standard_normal((N, N, N), dtype=np.float32)).astype(np.complex64)
    with Evolver(size=N, n_streams=2, precision="f32", cosmology=cosmo, **dict(TOPHAT, cfl=0.05)) as ev:
        d = ev.derived()
        dv = d["dx"] ** 3
        psi *= np.float32(1.0 / np.sqrt(np.sum(np.abs(psi).astype(np.float64) ** 2) * dv))
        ev.set_psi(0, psi)
        ev.copy_psi(1, 0)
        ev.forward(0)
        k = ev.get_psi(0)
        assert abs(np.sum(np.abs(k).astype(np.float64) ** 2) / np.sum(np.abs(psi).astype(np.float64) ** 2) - 1) < 1e-5
        ev.inverse(0)
        assert rel_err(ev.get_psi(0), psi) < 2e-5
        ev.copy_psi(0, 1)
        for _ in range(2):
            ev.step_begin([0.5, 0.5], [1e9, 1e9])
            st = ev.status()
            assert st[0].dt == st[1].dt and 0 < st[0].dt <= d["kinetic_dt"] * (1 + 1e-6)
            ev.step_end([0.5, 0.5], [s.time + s.dt for s in st], [s.tau + s.dt for s in st], [0.0, 0.0])
        a, b = ev.get_psi(0), ev.get_psi(1)
        assert rel_err(a, b) < 2e-5
        assert abs(np.sum(np.abs(a).astype(np.float64) ** 2) * dv - 1) < 1e-4
        st = ev.status()
        assert st[0].n_steps == 2 and st[0].aliased == 0 and st[0].active == 1


def test_config3_shape_256_64_streams():
    """256^3 fp64, 64 Wigner streams in two sub-batches of 32 (the single-GPU benchmark configuration with stochastic
    resampling on).  Size-independent properties: the step is unitary (norm of every checked stream stays 1), a
    stream's trajectory does not depend on where it sits -- stream 63 repeats the seed of stream 0 from the other
    sub-batch and the other lane of its Poisson pair -- and Parseval holds on the stored state."""
    from msm_amd.ics import plane_wave_psi
    N, ns = 256, 64
    with Evolver(size=N, n_streams=ns, batch_streams=32, **PLANE) as ev:
        ev.set_psi(0, plane_wave_psi(N))
        for s in range(1, ns):
            ev.copy_psi(s, 0)
        for s in range(ns):
            ev.sample(s, "Wigner", 1 if s == 63 else s + 1)
        norm0 = {s: np.sum(np.abs(ev.get_psi(s)) ** 2) for s in (0, 5, 31, 32)}   # sim.py's ICs are not unit-normalised
        for _ in range(3):
            ev.update()
        st = ev.status()
        assert all(r.n_steps == 3 and r.active == 1 and r.aliased == 0 for r in st)
        assert abs(st[63].dt / st[0].dt - 1) < 1e-12 and abs(st[63].time / st[0].time - 1) < 1e-12
        assert abs(st[1].dt / st[0].dt - 1) > 1e-9                     # other seeds really are other trajectories
        a, b = ev.get_psi(0), ev.get_psi(63)
        assert rel_err(b, a) < 1e-12
        for s, psi in ((0, a), (31, ev.get_psi(31)), (32, ev.get_psi(32))):
            assert abs(np.sum(np.abs(psi) ** 2) / norm0[s] - 1) < 1e-12, s
        assert rel_err(ev.get_psik(63), ev.get_psik(0)) < 1e-12
        assert abs(np.sum(np.abs(ev.get_psik(5)) ** 2) / norm0[5] - 1) < 1e-12   # Parseval on the stored state


def test_poisson_resampling(golden_dir):
    """Poisson scheme (ics.rs:495-558): counts from the stream's seeded Philox sequence.  Against the
    oracle's restatement of the same sampler (a count may differ where a rejection test lands within
    rounding of its threshold: at most a handful of cells), phases kept exactly, statistics as Poisson."""
    N = 16
    psi0 = o.user_specified_ics(os.path.join(golden_dir, "planeWave3d_e10_sym.npz"), 3, N)
    for ntot in (1e10, 2e4):                        # lambda ~ 2.4e6 (rejection branch) and ~ 5 (product branch)
        base = dict(PLANE, ntot=ntot)
        with Evolver(size=N, n_streams=2, **base) as ev:
            ev.set_psi(0, psi0)
            ev.copy_psi(1, 0)
            ev.sample(0, "Poisson", 3)
            ev.sample(1, "Poisson", 4)
            a, b = ev.get_psi(0), ev.get_psi(1)
        want = o.sample_quantum_perturbation(psi0, 60.0 / N, 3, PLANE["total_mass"], PLANE["total_mass"] / ntot, "Poisson", 3)
        same = np.isclose(a, want, rtol=1e-12, atol=1e-14 * np.abs(want).max())
        assert same.mean() > 0.999
        nz = np.abs(a) > 0
        assert np.allclose(a[nz] / np.abs(a[nz]), psi0[nz] / np.abs(psi0[nz]), atol=1e-12)
        dv = (60.0 / N) ** 3
        counts, lam = np.abs(a) ** 2 * dv * ntot, np.abs(psi0) ** 2 * dv * ntot
        assert np.allclose(counts, np.round(counts), atol=1e-6 * max(1.0, lam.max() * 1e-6))
        z = (counts - lam) / np.sqrt(lam)
        assert abs(z.mean()) < 0.1 and abs(z.var() - 1) < 0.1
        assert not np.array_equal(a, b)


def test_async_download_and_restart_clock():
    """msm_get_psi_async into pinned planes overlaps the following steps and returns the state of the
    moment it was issued; msm_set_clock places a fresh context at a dump boundary (restart)."""
    N = 32
    sim = oracle_sim(TOPHAT, N)
    with Evolver(size=N, n_streams=2, **TOPHAT) as ev:
        ev.set_psi(0, sim.psi)
        ev.copy_psi(1, 0)
        planes = [ev.pinned_planes() for _ in range(6)]
        tickets, wants = [], []
        for i in range(6):                     # more downloads in flight than staging grids
            ev.update()
            sim.update()
            tickets.append(ev.get_psi_async(0, planes[i]))
            wants.append(sim.psi.copy())
        ev.update()
        for i in range(6):
            ev.download_wait(tickets[i])
            assert rel_err(planes[i][0][0] + 1j * planes[i][1][0], wants[i]) < 1e-11, i
        for p in planes:
            ev.free_pinned(p)
        k = ev.status()[0].current_dumps
        psi_k, t_k = ev.get_psi(0), ev.status()[0].time
        ev.update()
        after = ev.get_psi(0)
    with Evolver(size=N, n_streams=1, **TOPHAT) as ev2:
        ev2.set_psi(0, psi_k)
        ev2.set_clock(0, t_k, 0.0, k)
        ev2.update()
        st = ev2.status()[0]
        assert st.current_dumps == k + 1 and abs(st.time / (t_k + st.dt) - 1) < 1e-12
        assert rel_err(ev2.get_psi(0), after) < 1e-12
        with pytest.raises(MsmError):
            ev2.set_clock(0, 0.0, 0.0, 0)      # only before the first step


@pytest.mark.parametrize("N,batch", [(16, 2), (256, 2)])
def test_streams_finished_before_the_first_step_keep_their_field(N, batch):
    """A restart can hand over streams that are already at the final time (msm_set_clock): they never step, alone in
    a batch or next to a live stream, and their psi and psi_k still download correctly after the others have moved on."""
    rng = np.random.default_rng(N)
    psis = [o.normalize(1.0 + 1e-3 * (rng.standard_normal((N,) * 3) + 1j * rng.standard_normal((N,) * 3)), 30.0 / N, 3)
            for _ in range(4)]
    with Evolver(size=N, n_streams=4, batch_streams=batch, **TOPHAT) as ev:
        for s in range(4):
            ev.set_psi(s, psis[s])
        ev.set_clock(1, TOPHAT["final_sim_time"], 0.0, TOPHAT["num_data_dumps"])      # next to live stream 0
        ev.set_clock(2, TOPHAT["final_sim_time"], 0.0, TOPHAT["num_data_dumps"])      # batch {2, 3}: both finished
        ev.set_clock(3, TOPHAT["final_sim_time"], 0.0, TOPHAT["num_data_dumps"])
        ev.update()
        ev.update()
        st = ev.status()
        assert [r.active for r in st] == [1, 0, 0, 0] and [r.n_steps for r in st] == [2, 0, 0, 0]
        for s in (1, 2, 3):
            assert rel_err(ev.get_psi(s), psis[s]) < 1e-13
            assert rel_err(ev.get_psik(s), o.forward(psis[s])) < 1e-13
        assert rel_err(ev.get_psi(0), psis[0]) > 1e-9


_DENSITY_SCRIPT = r"""
import sys
import numpy as np
import torch                      # first: libmsm_hip.so must bind to the HIP runtime torch already loaded
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import sp_oracle as o
from msm_amd import Evolver
TOPHAT = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11,
              hbar_=0.05, k2_cutoff=0.95, alias_threshold=0.02)
N, ns = 16, 5
toml = dict(TOPHAT, sim_name="t", dims=3, size=N, ics=dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0))
base = o.SimulationObject(list(o.simulation_iter(toml))[-1]).psi
dens = torch.zeros(N ** 3, dtype=torch.float64, device="cuda")
dens2 = torch.zeros(N ** 3, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
with Evolver(size=N, n_streams=ns, batch_streams=2, hip_stream=torch.cuda.current_stream().cuda_stream or None, **TOPHAT) as ev:
    for s in range(ns):
        ev.set_psi(s, o.sample_quantum_perturbation(base, 30.0 / N, 3, 1e11, 1e3, "Husimi", seed=s + 1))
    ev.density_sum(dens2.data_ptr()); ev.sync(); torch.cuda.synchronize()
    want0 = sum(np.abs(ev.get_psi(s)) ** 2 for s in range(ns)).reshape(-1)
    assert np.allclose(dens2.cpu().numpy(), want0, rtol=1e-13, atol=0)
    ev.step_density(dens.data_ptr())
    for _ in range(2):
        ev.update()
    ev.sync(); torch.cuda.synchronize()
    want = sum(np.abs(ev.get_psi(s)) ** 2 for s in range(ns)).reshape(-1)
    assert np.allclose(dens.cpu().numpy(), want, rtol=1e-12, atol=0), np.abs(dens.cpu().numpy() - want).max()
    ev.step_density(None)
    ev.density_sum(dens2.data_ptr()); ev.sync(); torch.cuda.synchronize()
    assert np.allclose(dens2.cpu().numpy(), want, rtol=1e-12, atol=0)
print("density ok")
"""


def test_step_density_product():
    # msm_step_density: sum_s |psi_s(t + dt)|^2 left by the step in a caller-owned device buffer (a torch
    # tensor, as bench.py uses it before the RCCL all-reduce), and msm_density_sum computed on demand, both
    # against the downloaded wavefunctions.  Own process: torch has to be imported before the library is loaded.
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _DENSITY_SCRIPT, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "density ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
