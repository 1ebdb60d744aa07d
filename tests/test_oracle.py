"""CPU: pins the numpy oracle against the reference's own known answers.

Each test names the reference test or fixture it restates (paths relative to
the reference checkout).
"""
import math
import os

import numpy as np
import pytest

import sp_oracle as o


def test_k_grid():
    # simulator/src/utils/fft.rs:163-171 (f32) and :173-182 (f64)
    assert o.get_kgrid(0.25, 4, np.float32).tolist() == [0.0, 1.0, -2.0, -1.0]
    assert o.get_kgrid(0.25, 4, np.float64).tolist() == [0.0, 1.0, -2.0, -1.0]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spec_grid(dtype):
    # simulator/src/utils/fft.rs:184-247: element i + 4 j + 16 k, exact equality
    size = 4
    T = np.dtype(dtype).type
    kg = o.get_kgrid(0.25, size, dtype)
    sg = o.spec_grid(0.25, 3, size, dtype)
    host = sg.reshape(-1)  # C order [k][j][i] -> flat i + 4 j + 16 k
    two_pi_sq = T(T(2.0 * math.pi) ** T(2.0))
    for i in range(size):
        for j in range(size):
            for k in range(size):
                want = T(T(kg[i] * kg[i] + kg[j] * kg[j]) + kg[k] * kg[k]) * two_pi_sq
                assert host[i + j * size + k * size * size] == want


@pytest.mark.parametrize("dims,size", [(1, 8), (1, 2), (2, 2), (3, 2)])
@pytest.mark.parametrize("cdt", [np.complex64, np.complex128])
def test_fft_roundtrip_constant(dims, size, cdt):
    # simulator/tests/fft.rs: constant arrays, forward then inverse, sum |x - y| <= 1e-6,
    # and the unitary transform keeps sum |psi|^2 dx with dk = dx (:28-45)
    length = 128.0
    values = np.full((size,) * dims, 1j * length ** (-0.5) if dims == 1 else 2j, dtype=cdt)
    k = o.forward(values)
    assert abs(np.sum(np.abs(k) ** 2) - np.sum(np.abs(values) ** 2)) < 1e-4 * np.sum(np.abs(values) ** 2)
    back = o.inverse(k)
    assert np.sum(np.abs(back - values)) <= 1e-6 * max(1.0, size ** dims)


@pytest.mark.parametrize("dims", [1, 2, 3])
@pytest.mark.parametrize("cdt", [np.complex64, np.complex128])
def test_normalize(dims, cdt):
    # simulator/src/utils/grid.rs:107-270: constant 1+1i on 8^dims, dx = 1/8
    S = 8
    dx = 1.0 / S
    arr = o.normalize(np.full((S,) * dims, 1 + 1j, dtype=cdt), dx, dims)
    assert abs(np.sum(np.abs(arr) ** 2) * dx ** dims - 1.0) < 1e-6
    assert o.check_norm(arr, dx, dims)


def test_seeds_grammar():
    # common/src/parameters.rs:121-144
    assert o.parse_seeds("0..=55") == list(range(56))
    assert o.parse_seeds("0 to 55") == list(range(56))
    assert o.parse_seeds("[1, 3]") == [1, 3]
    assert o.parse_seeds("1, 3") == [1, 3]
    # simulator/src/utils/io.rs:268,292: `{seeds:?}` of 1..=64
    assert o.parse_seeds("1..=64") == list(range(1, 65))
    with pytest.raises(ValueError):
        o.parse_seeds("none")


def test_planewave_fixture(golden_dir):
    # initial_conditions/planeWave3d_e10_sym.npz is what sim.py:getPsi() writes at N = 16;
    # the oracle's restatement of sim.py:124-183 reproduces it bit for bit
    d = np.load(os.path.join(golden_dir, "planeWave3d_e10_sym.npz"))
    psi = o.plane_wave_ics(16)
    assert np.array_equal(psi.real, d["real"]) and np.array_equal(psi.imag, d["imag"])
    assert abs(np.sum(np.abs(psi) ** 2) * (60.0 / 16) - 1.0) < 1e-12   # sim.py:176 normalises with dx
    ics = o.user_specified_ics(os.path.join(golden_dir, "planeWave3d_e10_sym.npz"), 3, 16)
    assert np.array_equal(ics, psi)
    with pytest.raises(ValueError):
        o.user_specified_ics(os.path.join(golden_dir, "planeWave3d_e10_sym.npz"), 3, 32)
    with pytest.raises(ValueError):
        o.user_specified_ics(os.path.join(golden_dir, "planeWave3d_e10_sym.npz"), 2, 16)


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    kat = [
        ([0, 0, 0, 0], (0, 0), [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, (0xFFFFFFFF, 0xFFFFFFFF), [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], (0xA4093822, 0x299F31D0),
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, want in kat:
        got = o.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0].tolist()
        assert got == want


def test_sampling_statistics():
    # ics.rs:560-646: per-component sigma 1/(2 sqrt n) (Wigner), 1/sqrt(2 n) (Husimi)
    N, dx = 16, 0.5
    n = 1.0e4
    psi = np.full((N, N, N), 0.3 + 0.1j, dtype=np.complex128)
    for scheme, sigma in (("Wigner", 1 / (2 * math.sqrt(n))), ("Husimi", 1 / math.sqrt(2 * n))):
        out = o.sample_quantum_perturbation(psi, dx, 3, n, 1.0, scheme, seed=5)
        d = (out - psi) * math.sqrt(dx ** 3)
        assert abs(d.real.std() / sigma - 1) < 0.05 and abs(d.imag.std() / sigma - 1) < 0.05
        assert abs(d.real.mean()) < 4 * sigma / math.sqrt(N ** 3)
    a = o.sample_quantum_perturbation(psi, dx, 3, n, 1.0, "Wigner", seed=5)
    b = o.sample_quantum_perturbation(psi, dx, 3, n, 1.0, "Wigner", seed=5)
    c = o.sample_quantum_perturbation(psi, dx, 3, n, 1.0, "Wigner", seed=6)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_simulation_iter_names():
    # simulator/src/utils/io.rs:183-241: N seeds -> N streams named {sim}-stream{seed:05}, then the MFT run
    toml = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11,
                hbar_=0.05, sim_name="spherical-tophat", k2_cutoff=0.95, alias_threshold=0.02, dims=3, size=8,
                ics=dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0),
                sampling=dict(seeds="1 to 10", scheme="Husimi"))
    ps = list(o.simulation_iter(toml))
    assert len(ps) == 11
    assert ps[0].sim_name == "spherical-tophat-stream00001" and ps[0].sampling == (1, "Husimi")
    assert ps[-1].sim_name == "spherical-tophat" and ps[-1].sampling is None
    assert ps[0].particle_mass == o.HBAR / 0.05


def test_static_step_invariants():
    # the update is unitary: sum |psi|^2 dx^3 is conserved; psi_k = F psi; dt obeys the three bounds
    toml = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11,
                hbar_=0.05, sim_name="t", k2_cutoff=0.95, alias_threshold=0.02, dims=3, size=16,
                ics=dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0))
    sim = o.SimulationObject(list(o.simulation_iter(toml))[-1])
    p = sim.parameters
    assert o.check_norm(sim.psi, p.dx, 3)
    for i in range(3):
        info = sim.update()
        assert abs(np.sum(np.abs(sim.psi) ** 2) * p.dx ** 3 - 1.0) < 1e-12
        assert np.allclose(o.forward(sim.psi), sim.psik, atol=1e-13)
        assert info["dt"] <= sim.last["kinetic_dt"] and info["dt"] <= sim.last["potential_dt"]
        # every step of this example is dump-limited: dt = T_f / 200 and time snaps to k T_f / 200
        assert info["dump"] and info["time"] == (i + 1) * 40.0 / 200
    assert p.current_dumps == 3 and len(sim.dumps) == 3


def test_scale_factor_einstein_de_sitter():
    # Omega_m = 1 (tomls/planeWave3d_e10_sym.toml:22): a(t) = (a0^{3/2} + 3/2 H0 t)^{2/3}
    cosmo = o.CosmologyParameters(1.0, 0.0, 0.7, 9.0, 0.01)
    s = o.ScaleFactorSolver(cosmo)
    h0 = 0.7 * o.LITTLE_H_TO_BIG_H
    t = 0.0
    for dt in (50.0, 200.0, 1000.0, -300.0):
        s.step(dt)
        t += dt
        want = (0.1 ** 1.5 + 1.5 * h0 * t) ** (2.0 / 3.0)
        assert abs(s.get_a() / want - 1) < 1e-9
    # tau(t) = int sqrt(1.5 Om H0^2) / a^2 dt, closed form for EdS
    tgt = 500.0
    tau = o.get_tau(tgt, cosmo)
    pref = math.sqrt(1.5) * h0
    f = lambda tt: -3.0 / (1.5 * h0) * (0.1 ** 1.5 + 1.5 * h0 * tt) ** (-1.0 / 3.0)
    assert abs(tau / (pref * (f(tgt) - f(0.0))) - 1) < 1e-6


def test_planewave_1d_fixture(golden_dir):
    # the reference's own 1-D initial conditions (planeWave1d.npz beside examples/repro.toml): 256 points, unit norm
    path = os.path.join(golden_dir, "planeWave1d.npz")
    psi = o.user_specified_ics(path, 1, 256)
    assert psi.shape == (256,) and psi.dtype == np.complex128
    assert abs(np.sum(np.abs(psi) ** 2) * 60.0 / 256 - 1.0) < 1e-12
    with pytest.raises(ValueError):
        o.user_specified_ics(path, 1, 128)
    with pytest.raises(ValueError):
        o.user_specified_ics(path, 2, 256)


def test_goldens_regenerate(golden_dir):
    # the committed vectors are what the oracle produces today (guards oracle drift)
    g = np.load(os.path.join(golden_dir, "tophat_n8_f64.npz"))
    toml = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11,
                hbar_=0.05, sim_name="spherical-tophat", k2_cutoff=0.95, alias_threshold=0.02, dims=3, size=8,
                ics=dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0))
    sim = o.SimulationObject(list(o.simulation_iter(toml))[-1])
    assert np.allclose(sim.psi, g["psi0"], rtol=0, atol=1e-15)
    for i in range(2):
        sim.update()
    assert np.allclose(sim.psi, g["psi_2"], rtol=0, atol=1e-13)


def test_expanding_golden_regenerates(golden_dir):
    # the committed expanding-box run (examples/spherical-tophat-cosmo.toml, shortened) is what the oracle produces today
    g = np.load(os.path.join(golden_dir, "tophat_cosmo_n16_run3_f64.npz"))
    toml = dict(axis_length=30.0, final_sim_time=0.6, cfl=0.5, num_data_dumps=3, total_mass=1e11, hbar_=0.05,
                sim_name="spherical-tophat-cosmo", k2_cutoff=0.95, alias_threshold=0.02, dims=3, size=16,
                ics=dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0),
                cosmology=dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01))
    sim = o.SimulationObject(list(o.simulation_iter(toml, expanding=True))[-1])
    sim.dump()
    while sim.not_finished():
        sim.update()
    assert len(sim.dumps) == int(g["n_dumps"]) == 4 and sim.parameters.n_steps == int(g["n_steps"])
    for k, psi in sim.dumps:
        assert np.allclose(psi, g[f"dump_{k}"], rtol=0, atol=1e-13)


def test_poisson_counts_distribution():
    # the stand-in for rand_distr::Poisson (ics.rs:519-520): mean and variance = lambda on both branches
    for lam, n in ((0.3, 200000), (6.5, 200000), (40.0, 200000), (2.4e6, 4000)):
        k = o.poisson_counts(np.full(n, lam), 9)
        assert abs(k.mean() / lam - 1) < 5 / np.sqrt(n * lam) + 1e-9
        assert abs(k.var() / lam - 1) < 0.05 + (0.1 if n < 10000 else 0)
        assert np.all(k >= 0) and np.all(k == np.floor(k))
    assert np.array_equal(o.poisson_counts(np.full(100, 3.0), 1), o.poisson_counts(np.full(100, 3.0), 1))
    assert not np.array_equal(o.poisson_counts(np.full(100, 3.0), 1), o.poisson_counts(np.full(100, 3.0), 2))
    assert np.all(o.poisson_counts(np.zeros(10), 3) == 0)


def test_cold_gauss_kspace_norm():
    psi = o.cold_gauss_kspace([0.1, 0.0, -0.1], [0.2, 0.3, 0.25], 30 / 16, 3, 16, 7)
    assert abs(np.sum(np.abs(psi) ** 2) * (30 / 16) ** 3 - 1) < 1e-12


def test_product_plane_wave_generator_equals_the_shipped_npz(golden_dir):
    # msm_amd.ics.plane_wave_psi is what bench.py and the 256^3 tests start from: at the fixture's size it must be the
    # reference's own initial_conditions/planeWave3d_e10_sym.npz (written by sim.py:185-186), bit for bit
    from msm_amd.ics import plane_wave_psi
    g = np.load(os.path.join(golden_dir, "planeWave3d_e10_sym.npz"))
    psi = plane_wave_psi(16)
    assert psi.shape == (16, 16, 16)
    assert np.array_equal(psi.real, g["real"]) and np.array_equal(psi.imag, g["imag"])
    assert np.array_equal(plane_wave_psi(8), o.plane_wave_ics(8))
