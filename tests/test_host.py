"""CPU: the host-side C++ pieces behind the CLI (libmsm_host.so: TOML schema, seeds grammar,
stream iterator, IC builders, NPY/NPZ codec, scale-factor solver) against the reference's own
tests and the numpy oracle."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import sp_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "msm_amd", "libmsm_host.so")


@pytest.fixture(scope="module")
def host():
    lib = C.CDLL(LIB)
    lib.msmh_last_error.restype = C.c_char_p
    lib.msmh_get_tau.restype = C.c_double
    lib.msmh_get_tau.argtypes = [C.c_double, C.POINTER(C.c_double), C.c_int]
    lib.msmh_supercomoving_boxsize.restype = C.c_double
    lib.msmh_supercomoving_boxsize.argtypes = [C.c_double, C.POINTER(C.c_double), C.c_double]
    lib.msmh_dt_from_dtau.restype = C.c_double
    lib.msmh_dt_from_dtau.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_double, C.c_double, C.c_double]
    lib.msmh_build_ics.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    return lib


def seeds(lib, s):
    out = (C.c_ulonglong * 4096)()
    n = lib.msmh_parse_seeds(s.encode(), out, 4096)
    if n < 0:
        raise ValueError(lib.msmh_last_error().decode())
    return list(out[:n])


def describe(lib, path, expanding=False):
    buf = C.create_string_buffer(1 << 16)
    if lib.msmh_describe_toml(str(path).encode(), int(expanding), buf, len(buf)) != 0:
        raise ValueError(lib.msmh_last_error().decode())
    return json.loads(buf.value.decode())


TOPHAT_TOML = """axis_length                 = 30
final_sim_time              = 40
cfl                         = 0.5
num_data_dumps              = 200
total_mass                  = 1e11
#particle_mass              = None
hbar_                       = 0.05
sim_name                    = "spherical-tophat"
k2_cutoff                   = 0.95
alias_threshold             = 0.02
dims                        = 3
size                        = 16

[ics]
type = "SphericalTophat"
radius = 5.0
slope = 50
delta = 100

[sampling]
seeds = "1 to 10"
scheme = "Husimi"
"""


def test_seeds_grammar(host):
    # common/src/parameters.rs:121-144
    assert seeds(host, "0..=55") == list(range(56))
    assert seeds(host, "0 to 55") == list(range(56))
    assert seeds(host, "[1, 3]") == [1, 3]
    assert seeds(host, "1, 3") == [1, 3]
    with pytest.raises(ValueError):
        seeds(host, "none")


def test_deserialize_toml(host, tmp_path):
    # simulator/src/utils/io.rs:248-326
    text = """\
axis_length                 = 30
final_sim_time              = 400
cfl                         = 0.5
num_data_dumps              = 100
total_mass                  = 10000000000
hbar_                       = 0.02
sim_name                    = "gaussian-overdensity-512-mft"
k2_cutoff                   = 0.95
alias_threshold             = 0.02
dims                        = 3
size                        = 512

[ics]
type = "ColdGaussKSpace"
mean = [15.0, 15.0, 15.0]
std = [10.0, 10.0, 10.0]

[sampling]
seeds = "1..=64"
scheme = "Husimi"
"""
    p = tmp_path / "t.toml"
    p.write_text(text)
    d = describe(host, p)
    assert d["axis_length"] == 30.0 and d["final_sim_time"] == 400.0 and d["cfl"] == 0.5
    assert d["num_data_dumps"] == 100 and d["total_mass"] == 1e10 and d["hbar_"] == 0.02
    assert d["sim_name"] == "gaussian-overdensity-512-mft" and d["k2_cutoff"] == 0.95
    assert d["alias_threshold"] == 0.02 and d["dims"] == 3 and d["size"] == 512
    assert d["ics"]["type"] == "ColdGaussKSpace" and d["ics"]["mean"] == [15.0] * 3 and d["ics"]["std"] == [10.0] * 3
    assert d["sampling"] == {"scheme": "Husimi", "seeds": list(range(1, 65))}
    assert d["output_potential"] is False and d["time"] is None


def test_stream_iterator_and_unknown_keys(host, tmp_path):
    # io.rs:183-241; unknown keys (num_streams, [remote_storage_parameters]) are ignored like serde does
    p = tmp_path / "t.toml"
    p.write_text(TOPHAT_TOML + "num_streams = 16\n\n[remote_storage_parameters]\nkeypair = \"dev.json\"\n")
    d = describe(host, p)
    names = [s["sim_name"] for s in d["streams"]]
    assert names == [f"spherical-tophat-stream{k:05d}" for k in range(1, 11)] + ["spherical-tophat"]
    assert [s["sampled"] for s in d["streams"]] == [True] * 10 + [False]
    assert d["resolved_hbar_"] == 0.05 and d["resolved_particle_mass"] == o.HBAR / 0.05
    assert d["cosmology"] is None
    with pytest.raises(ValueError, match="cosmology"):
        describe(host, p, expanding=True)          # feature "expanding" makes [cosmology] mandatory
    q = tmp_path / "bad.toml"
    q.write_text(TOPHAT_TOML.replace("cfl                         = 0.5\n", ""))
    with pytest.raises(ValueError, match="cfl"):
        describe(host, q)
    with pytest.raises(ValueError, match="failed to read"):
        describe(host, tmp_path / "missing.toml")


def build(host, toml_path, N, dims, dx, dtype=np.float64, expanding=False):
    re = np.zeros((N,) * dims, dtype=dtype)
    im = np.zeros((N,) * dims, dtype=dtype)
    rc = host.msmh_build_ics(str(toml_path).encode(), int(expanding), 0 if dtype == np.float64 else 1, dx,
                             re.ctypes.data, im.ctypes.data)
    if rc != 0:
        raise ValueError(host.msmh_last_error().decode())
    return re + 1j * im


@pytest.mark.parametrize("dtype,tol", [(np.float64, 4e-15), (np.float32, 3e-6)])
def test_spherical_tophat(host, tmp_path, dtype, tol):
    p = tmp_path / "t.toml"
    p.write_text(TOPHAT_TOML)
    got = build(host, p, 16, 3, 30.0 / 16, dtype)
    want = o.spherical_tophat(30.0, 30.0 / 16, 3, 16, 5.0, 100.0, 50.0, dtype)
    assert np.max(np.abs(got - want)) <= tol * np.max(np.abs(want))
    assert o.check_norm(got, 30.0 / 16, 3)


def test_cold_gauss(host, tmp_path):
    # examples/gaussian-overdensity-mft.toml at a small size; mean[0] belongs to the fastest axis
    text = TOPHAT_TOML.split("[ics]")[0] + "[ics]\ntype = \"ColdGauss\"\nmean = [10, 15, 20]\nstd = [10, 5, 8]\n"
    p = tmp_path / "t.toml"
    p.write_text(text)
    got = build(host, p, 16, 3, 30.0 / 16)
    want = o.cold_gauss([10, 15, 20], [10, 5, 8], 30.0 / 16, 3, 16)
    assert np.max(np.abs(got - want)) <= 1e-14 * np.max(np.abs(want))
    assert abs(np.sum(np.abs(got) ** 2) * (30.0 / 16) ** 3 - 1) < 1e-12
    # widest along the axis whose std is 10: the LAST numpy axis
    prof = [np.abs(got).sum(axis=tuple(a for a in range(3) if a != ax)) for ax in range(3)]
    assert np.std(prof[2] / prof[2].sum()) < np.std(prof[1] / prof[1].sum())


def test_cold_gauss_kspace_field(host, tmp_path):
    # ics.rs:282-424: the k-space product with seeded uniform phases (before the transform to real space)
    text = TOPHAT_TOML.split("[ics]")[0] + "[ics]\ntype = \"ColdGaussKSpace\"\nmean = [0.1, 0.0, -0.1]\nstd = [0.2, 0.3, 0.25]\nphase_seed = 7\n"
    p = tmp_path / "t.toml"
    p.write_text(text)
    got = build(host, p, 16, 3, 30.0 / 16)
    want = o.cold_gauss_kspace_k([0.1, 0.0, -0.1], [0.2, 0.3, 0.25], 30.0 / 16, 3, 16, 7)
    assert np.max(np.abs(got - want)) <= 1e-14 * np.max(np.abs(want))
    assert abs(np.sum(np.abs(got) ** 2) * (30.0 / 16) ** 3 - 1) < 1e-12
    # the phases are a function of (element, seed) only
    u = np.angle(got.reshape(-1)[:4]) / (2 * np.pi) % 1.0
    assert np.allclose(u, o.uniforms(4, 7), atol=1e-12)
    # default seed is 0
    p.write_text(text.replace("phase_seed = 7\n", ""))
    assert np.allclose(build(host, p, 16, 3, 30.0 / 16), o.cold_gauss_kspace_k([0.1, 0.0, -0.1], [0.2, 0.3, 0.25], 30.0 / 16, 3, 16, None),
                       rtol=0, atol=1e-14)


@pytest.mark.parametrize("compressed", [False, True])
def test_user_specified_ics(host, tmp_path, golden_dir, compressed):
    # ics.rs:650-730; both plain np.savez (what sim.py writes) and deflated members
    d = np.load(os.path.join(golden_dir, "planeWave3d_e10_sym.npz"))
    npz = tmp_path / "ics.npz"
    (np.savez_compressed if compressed else np.savez)(npz, real=d["real"], imag=d["imag"])
    text = TOPHAT_TOML.split("[ics]")[0] + f"[ics]\ntype = \"UserSpecified\"\npath = \"{npz}\"\n"
    p = tmp_path / "t.toml"
    p.write_text(text)
    got = build(host, p, 16, 3, 60.0 / 16)
    assert np.array_equal(got.real, d["real"]) and np.array_equal(got.imag, d["imag"])
    for bad, msg in ((text.replace("size                        = 16", "size                        = 32"), "Grid size"),
                     (text.replace("dims                        = 3", "dims                        = 2"), "Dimensions"),
                     (text.replace(str(npz), str(tmp_path / "nope.npz")), "does not exist")):
        p.write_text(bad)
        with pytest.raises(ValueError, match=msg):
            build(host, p, 32, 3, 1.0)
    # the reference reads the members as T: an f32 file is rejected for an f64 run
    np.savez(npz, real=d["real"].astype(np.float32), imag=d["imag"].astype(np.float32))
    p.write_text(text)
    with pytest.raises(ValueError, match="real part"):
        build(host, p, 16, 3, 60.0 / 16)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_npy_writer(host, tmp_path, dtype):
    # simulator/tests/io.rs:1-47 and the dump shape (N,N,N,1) of utils/io.rs:63-68
    a = np.arange(4 * 4 * 4, dtype=dtype).reshape(4, 4, 4) * 0.5
    path = tmp_path / "psi_00001_real"          # dumps have no .npy extension
    shape = (C.c_ulonglong * 4)(4, 4, 4, 1)
    assert host.msmh_write_npy(str(path).encode(), a.ctypes.data_as(C.c_void_p), 0 if dtype == np.float64 else 1, shape, 4) == 0
    with open(path, "rb") as f:
        back = np.load(f)
    assert back.shape == (4, 4, 4, 1) and back.dtype == dtype and np.array_equal(back[..., 0], a)
    assert os.path.getsize(path) % 64 == a.nbytes % 64     # header padded to 64 bytes


def test_scale_factor_and_tau(host):
    cosmo = (C.c_double * 5)(0.7, 0.0, 0.7, 1.0, 0.01)
    oc = o.CosmologyParameters(0.7, 0.0, 0.7, 1.0, 0.01)
    dts = np.array([10.0, 250.0, 0.0, -40.0, 1000.0])
    a = np.zeros(5)
    assert host.msmh_scale_factor(cosmo, 1, dts.ctypes.data_as(C.POINTER(C.c_double)), 5, a.ctypes.data_as(C.POINTER(C.c_double))) == 0
    s = o.ScaleFactorSolver(oc)
    want = [s.step(float(dt)) for dt in dts]
    assert np.allclose(a, want, rtol=1e-14, atol=0)
    for t in (0.2, 40.0, 2000.0):
        assert abs(host.msmh_get_tau(t, cosmo, 1) / o.get_tau(t, oc) - 1) < 1e-13
    assert host.msmh_supercomoving_boxsize(0.05, cosmo, 30.0) == pytest.approx(o.get_supercomoving_boxsize(0.05, oc, 30.0), rel=1e-15)
    # Einstein-de Sitter closed form (Omega_m = 1, tomls/planeWave3d_e10_sym.toml:22)
    eds = (C.c_double * 5)(1.0, 0.0, 0.7, 9.0, 0.01)
    out = np.zeros(1)
    one = np.array([500.0])
    host.msmh_scale_factor(eds, 1, one.ctypes.data_as(C.POINTER(C.c_double)), 1, out.ctypes.data_as(C.POINTER(C.c_double)))
    h0 = 0.7 * o.LITTLE_H_TO_BIG_H
    assert abs(out[0] / (0.1 ** 1.5 + 1.5 * h0 * 500.0) ** (2 / 3) - 1) < 1e-9


def test_dt_from_dtau(host):
    # simulation_object.rs:1344-1388 against the oracle's restatement
    cosmo = (C.c_double * 5)(0.7, 0.0, 0.7, 1.0, 0.01)
    oc = o.CosmologyParameters(0.7, 0.0, 0.7, 1.0, 0.01)
    toml = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11, hbar_=0.05,
                sim_name="t", k2_cutoff=0.95, alias_threshold=0.02, dims=3, size=8,
                ics=dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0),
                cosmology=dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01))
    sim = o.SimulationObject(list(o.simulation_iter(toml, expanding=True))[-1])
    sim.scale_factor_solver.step(12.5)
    sim.parameters.time = np.float64(12.5)
    sim.parameters.tau = np.float64(o.get_tau(12.5, oc))
    want = sim._calculate_dt_from_dtau(3e-4)
    got = host.msmh_dt_from_dtau(cosmo, 1, 12.5, float(sim.parameters.tau), 3e-4)
    assert abs(got / want - 1) < 1e-12


@pytest.mark.parametrize("precision,dtype,rtol", [("f64", np.float64, 1e-12), ("f32", np.float32, 1e-6)])
def test_expanding_host_loop_matches_oracle_bookkeeping(precision, dtype, rtol):
    """host/expanding_loop.hpp (what msm-simulator and the benchmark put between msm_step_begin and msm_step_end) against
    the same scalars taken from an oracle object (simulation_object.rs:726-760, :939-990) over six steps, two streams;
    d tau comes from the oracle's own step, the streams cross dump times on the way."""
    import copy

    import expanding_host as xh
    from msm_amd.expanding import ExpandingHost

    cosmo = dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01)
    toml = dict(axis_length=30.0, final_sim_time=0.6, cfl=0.05, num_data_dumps=3, total_mass=1e11, hbar_=0.05, ntot=1e6,
                sim_name="t", k2_cutoff=0.95, alias_threshold=0.02, dims=3, size=8,
                ics=dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0), cosmology=cosmo,
                sampling=dict(seeds="1", scheme="Husimi"))
    sims = [o.SimulationObject(p) for p in o.simulation_iter(toml, expanding=True, dtype=dtype)]
    host = ExpandingHost(cosmo, 2, precision, 0.6, 3)
    dumps_seen = 0
    for step in range(6):
        live = [1 if sim.not_finished() else 0 for sim in sims]
        if not any(live):
            break
        pre = [xh.step_inputs(sim) for sim in sims]
        a_now, tau_next, snapped = host.begin([sim.parameters.current_dumps for sim in sims])
        for s in range(2):
            if live[s]:
                assert np.allclose([a_now[s], tau_next[s], snapped[s]], pre[s], rtol=rtol, atol=0), (step, s)
        dtau = []
        for s, sim in enumerate(sims):
            shadow = copy.deepcopy(sim)
            dtau.append(float(shadow.update()["dt"]) if live[s] else 0.0)
        post = [xh.half_kicks(sim, dtau[s]) for s, sim in enumerate(sims)]
        kick_a, new_time, new_tau = host.advance(dtau, live)
        for s in range(2):
            if live[s]:
                assert np.allclose([kick_a[s], new_time[s], new_tau[s]], post[s], rtol=rtol, atol=0), (step, s)
        for s, sim in enumerate(sims):
            if live[s]:
                dumps_seen += int(sim.update()["dump"])
        host.commit([float(sim.parameters.time) for sim in sims], [float(sim.parameters.tau) for sim in sims])
    assert dumps_seen >= 2
    host.close()
