"""GPU: the msm-simulator process surface end to end (TOML in, npz in, NPY dumps out) against
the numpy oracle run on the same configuration.  Tolerance: 1e-11 * max|psi| per dump (fp64)."""
import os
import subprocess

import numpy as np
import pytest

import sp_oracle as o

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "msm_amd", "bin", "msm-simulator")


def toml_text(d):
    out = []
    for k, v in d.items():
        if isinstance(v, dict):
            continue
        out.append(f"{k} = " + (f'"{v}"' if isinstance(v, str) else str(v).lower() if isinstance(v, bool) else repr(v)))
    for k, v in d.items():
        if isinstance(v, dict):
            out.append(f"\n[{k}]")
            for kk, vv in v.items():
                out.append(f"{kk} = " + (f'"{vv}"' if isinstance(vv, str) else repr(vv)))
    return "\n".join(out) + "\n"


def run_cli(tmp_path, cfg, *flags, expect=0):
    path = tmp_path / "run.toml"
    path.write_text(toml_text(cfg))
    r = subprocess.run([BIN, "--toml", str(path), *flags], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == expect, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return r


def load_dump(tmp_path, name, k, what="psi"):
    base = tmp_path / "sim-data" / name / f"{what}_{k:05d}"
    with open(str(base) + "_real", "rb") as f:
        re = np.load(f)
    with open(str(base) + "_imag", "rb") as f:
        im = np.load(f)
    assert re.ndim == 4 and re.shape[3] == 1 and re.dtype == np.float64
    return re[..., 0] + 1j * im[..., 0]


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def oracle_runs(cfg, expanding=False):
    sims = []
    for params in o.simulation_iter(cfg, expanding=expanding):
        sim = o.SimulationObject(params)
        sim.dump()                       # main.rs:61, dump 0
        while sim.not_finished():
            sim.update()
        sims.append(sim)
    return sims


TOPHAT = dict(axis_length=30.0, final_sim_time=1.0, cfl=0.5, num_data_dumps=5, total_mass=1e11, ntot=1e6,
              hbar_=0.05, sim_name="spherical-tophat", k2_cutoff=0.95, alias_threshold=0.02, dims=3, size=16,
              ics=dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0),
              sampling=dict(seeds="1 to 2", scheme="Husimi"))


def test_static_tophat_streams_and_mft(tmp_path):
    # examples/spherical-tophat.toml with a short run: 2 Husimi streams + the mean-field run
    r = run_cli(tmp_path, TOPHAT, "--verbose")
    assert "Working on simulation spherical-tophat-stream00001" in r.stdout and "Finished all streams" in r.stdout
    sims = oracle_runs(TOPHAT)
    assert [s.parameters.sim_name for s in sims] == ["spherical-tophat-stream00001", "spherical-tophat-stream00002",
                                                      "spherical-tophat"]
    for sim in sims:
        assert len(sim.dumps) == 6
        for k, psi in sim.dumps:
            assert rel(load_dump(tmp_path, sim.parameters.sim_name, k), psi) < 1e-11, (sim.parameters.sim_name, k)
    assert not (tmp_path / "sim-data" / "spherical-tophat" / "psi_00006_real").exists()


def test_user_specified_wigner_and_potential(tmp_path, golden_dir):
    # tomls/planeWave3d_e10_sym.toml: npz initial conditions, Wigner streams, unknown keys, output_potential
    cfg = dict(axis_length=60.0, final_sim_time=5e-6, cfl=0.1, num_data_dumps=2, total_mass=3e16, hbar_=0.01,
               sim_name="planeWave3d_e10_sym", ntot=1e10, k2_cutoff=0.95, alias_threshold=0.001, dims=3, size=16,
               output_potential=True,
               ics=dict(type="UserSpecified", path=os.path.join(golden_dir, "planeWave3d_e10_sym.npz")),
               cosmology=dict(omega_matter_now=1.0, omega_radiation_now=0.0, h=1e-07, z0=99.0, max_dloga=0.01),
               sampling=dict(num_streams=16, seeds="3, 7", scheme="Wigner"),
               remote_storage_parameters=dict(keypair="dev.json", storage_account="streams"))
    run_cli(tmp_path, cfg)
    sims = oracle_runs(cfg)
    assert [s.parameters.sim_name for s in sims][:2] == ["planeWave3d_e10_sym-stream00003", "planeWave3d_e10_sym-stream00007"]
    for sim in sims:
        assert len(sim.dumps) == 3 and sim.parameters.n_steps >= 5
        for k, psi in sim.dumps:
            assert rel(load_dump(tmp_path, sim.parameters.sim_name, k), psi) < 1e-11
        # potential_{k} = calculate_potential of the dumped psi, imaginary plane zero (simulation_object.rs:1167-1179)
        k, psi = sim.dumps[-1]
        phi = load_dump(tmp_path, sim.parameters.sim_name, k, "potential")
        sim.psi = psi
        assert rel(phi.real, sim.calculate_potential()) < 1e-11 and not phi.imag.any()


def test_repro_example_1d_aliases_as_the_oracle_does(tmp_path, golden_dir):
    # examples/repro.toml as a static box: the reference's 1-D fixture planeWave1d.npz (256 points), Wigner streams.
    # With these constants the first stream collapses onto the grid scale and trips check_alias after a few hundred
    # steps (simulation_object.rs:607-617): the run must die at the same step with the same aliased mass.
    cfg = dict(axis_length=60.0, final_sim_time=1000.0, cfl=0.1, num_data_dumps=128, total_mass=3e16, hbar_=0.01,
               sim_name="planeWave1d", ntot=1e12, k2_cutoff=0.95, alias_threshold=0.001, dims=1, size=256,
               ics=dict(type="UserSpecified", path=os.path.join(golden_dir, "planeWave1d.npz")),
               cosmology=dict(omega_matter_now=1.0, omega_radiation_now=0.0, h=1e-07, z0=99.0, max_dloga=0.01),
               sampling=dict(num_streams=128, seeds="1 to 1", scheme="Wigner"))
    r = run_cli(tmp_path, cfg, expect=101)
    sim = o.SimulationObject(next(iter(o.simulation_iter(cfg))))
    psi0 = sim.psi.copy()
    with pytest.raises(RuntimeError) as exc:
        for _ in range(5000):
            sim.update()
    want = float(str(exc.value).split("p_mass is ")[1].split()[0])
    got = float(r.stderr.split("p_mass is ")[1].split()[0])
    assert 200 < sim.parameters.n_steps < 1000 and abs(got / want - 1) < 1e-5, (got, want)
    assert "(planeWave1d-stream00001)" in r.stderr
    base = tmp_path / "sim-data" / "planeWave1d-stream00001" / "psi_00000"
    re, im = np.load(open(str(base) + "_real", "rb")), np.load(open(str(base) + "_imag", "rb"))
    assert re.shape == (256, 1, 1, 1) and rel((re + 1j * im).reshape(256), psi0) < 1e-13


def test_expanding_box(tmp_path):
    # examples/spherical-tophat-cosmo.toml under the reference's feature "expanding" (test.sh:6)
    cfg = dict(TOPHAT, sim_name="spherical-tophat-cosmo", final_sim_time=0.6, num_data_dumps=3,
               sampling=dict(seeds="1 to 1", scheme="Husimi"),
               cosmology=dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01))
    r = run_cli(tmp_path, cfg, "--expanding")
    assert "z = " in r.stdout
    sims = oracle_runs(cfg, expanding=True)
    for sim in sims:
        assert len(sim.dumps) == 4
        for k, psi in sim.dumps:
            assert rel(load_dump(tmp_path, sim.parameters.sim_name, k), psi) < 1e-11, (sim.parameters.sim_name, k)


def test_test_flag_and_errors(tmp_path, golden_dir):
    run_cli(tmp_path, TOPHAT, "--test")
    assert not (tmp_path / "sim-data").exists()          # main.rs:59: ICs only, no dump, no evolution
    bad = dict(TOPHAT, ics=dict(type="UserSpecified", path=str(tmp_path / "missing.npz")))
    r = run_cli(tmp_path, bad, expect=1)
    assert "ics file does not exist" in r.stderr
    r = run_cli(tmp_path, dict(TOPHAT, size=32, ics=dict(type="UserSpecified",
                path=os.path.join(golden_dir, "planeWave3d_e10_sym.npz"))), expect=1)
    assert "Grid size of user-provided data (16) does not match the size specified in the toml (32)" in r.stderr
    r = run_cli(tmp_path, TOPHAT, "--expanding", expect=1)
    assert "cosmology" in r.stderr
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 2 and "--toml" in r.stderr


def test_aliasing_aborts_like_the_reference(tmp_path):
    # a Nyquist-corner wavefunction trips check_alias on the first step; the reference panics
    # (simulation_object.rs:607-617), the CLI exits with Rust's panic status and the same message
    N = 8
    idx = np.indices((N, N, N)).sum(axis=0)
    psi = o.normalize(((-1.0) ** idx).astype(np.complex128), 30.0 / N, 3)
    np.savez(tmp_path / "nyq.npz", real=psi.real, imag=psi.imag)
    cfg = dict(TOPHAT, size=N, ics=dict(type="UserSpecified", path=str(tmp_path / "nyq.npz")))
    del cfg["sampling"]
    r = run_cli(tmp_path, cfg, expect=101)
    assert "simulation aliased" in r.stderr and "k2_cutoff = 0.95" in r.stderr
    assert (tmp_path / "sim-data" / "spherical-tophat" / "psi_00000_real").exists()
    assert not (tmp_path / "sim-data" / "spherical-tophat" / "psi_00001_real").exists()


def test_full_example_run_against_golden(tmp_path, golden_dir):
    # examples/spherical-tophat.toml as shipped (size 16, 200 dumps) minus [sampling]: the mean-field stream,
    # start to finish, against the committed dumps 0, 1 and 200
    g = np.load(os.path.join(golden_dir, "tophat_n16_run200_f64.npz"))
    cfg = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11, hbar_=0.05,
               sim_name="spherical-tophat", k2_cutoff=0.95, alias_threshold=0.02, dims=3, size=16,
               ics=dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0))
    run_cli(tmp_path, cfg)
    for k, tol in ((0, 1e-14), (1, 1e-11), (200, 1e-10)):
        assert rel(load_dump(tmp_path, "spherical-tophat", k), g[f"dump_{k}"]) < tol, k
    assert not (tmp_path / "sim-data" / "spherical-tophat" / "psi_00201_real").exists()


def test_expanding_run_against_golden(tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, "tophat_cosmo_n16_run3_f64.npz"))
    cfg = dict(axis_length=30.0, final_sim_time=0.6, cfl=0.5, num_data_dumps=3, total_mass=1e11, hbar_=0.05,
               sim_name="spherical-tophat-cosmo", k2_cutoff=0.95, alias_threshold=0.02, dims=3, size=16,
               ics=dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0),
               cosmology=dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01))
    run_cli(tmp_path, cfg, "--expanding")
    for k in range(4):
        assert rel(load_dump(tmp_path, "spherical-tophat-cosmo", k), g[f"dump_{k}"]) < 1e-11, k


def test_synthesize_flag_matches_post_processing(tmp_path):
    # --synthesize: the -combined/ products equal the reference synthesizer's arithmetic applied to the
    # stream dumps this very run wrote (synthesizer/src/main.rs:63-187); the mean-field run is excluded
    cfg = dict(TOPHAT, sampling=dict(seeds="1 to 3", scheme="Husimi"))
    run_cli(tmp_path, cfg, "--synthesize")
    names = [f"spherical-tophat-stream{seed:05d}" for seed in (1, 2, 3)]
    dv = (cfg["axis_length"] / cfg["size"]) ** 3
    qx = load_dump(tmp_path, "spherical-tophat-combined", 0, "psi")  # placeholder to assert layout below
    assert qx.shape == (16, 16, 16)
    with open(tmp_path / "sim-data" / "spherical-tophat-combined" / "Qx_real", "rb") as f:
        qre = np.load(f)
    assert qre.shape == (cfg["num_data_dumps"] + 1, 1, 1, 1)
    for k in range(cfg["num_data_dumps"] + 1):
        want = o.reference_synthesis([load_dump(tmp_path, n, k) for n in names], dv)
        for field in ("psi", "psi2", "psik", "psik2"):
            got = load_dump(tmp_path, "spherical-tophat-combined", k, field)
            assert rel(got, want[field]) < 1e-12, (k, field)
        assert abs(qre[k, 0, 0, 0] - want["Qx"].real) < 1e-12


def test_cold_gauss_kspace_and_poisson_streams(tmp_path):
    # [ics] ColdGaussKSpace with phase_seed + [sampling] Poisson: dump 0 of the mean-field run is the oracle's field
    cfg = dict(TOPHAT, sim_name="cgk", final_sim_time=0.4, num_data_dumps=2,
               ics=dict(type="ColdGaussKSpace", mean=[0.05, 0.0, -0.05], std=[0.1, 0.12, 0.08], phase_seed=11),
               sampling=dict(seeds="5", scheme="Poisson"))
    run_cli(tmp_path, cfg)
    want0 = o.cold_gauss_kspace([0.05, 0.0, -0.05], [0.1, 0.12, 0.08], 30.0 / 16, 3, 16, 11)
    assert rel(load_dump(tmp_path, "cgk", 0), want0) < 1e-12
    sims = oracle_runs(dict(cfg, sampling=None) if False else {k: v for k, v in cfg.items() if k != "sampling"})
    for k, psi in sims[-1].dumps:
        assert rel(load_dump(tmp_path, "cgk", k), psi) < 1e-10, k
    stream0 = load_dump(tmp_path, "cgk-stream00005", 0)
    counts = np.abs(stream0) ** 2 * (30.0 / 16) ** 3 * 1e6        # ntot = 1e6
    assert np.allclose(counts, np.round(counts), atol=1e-6) and abs(counts.sum() / 1e6 - 1) < 0.01


@pytest.mark.parametrize("dims,size,flags", [(1, 64, ()), (2, 32, ()), (2, 32, ("--expanding",))])
def test_one_and_two_dimensional_runs(tmp_path, dims, size, flags):
    # dims = 1, 2 (examples/repro.toml is a 1-D run): dump shapes (N,1,1,1) / (N,N,1,1) (simulation_object.rs:1149-1160)
    cfg = dict(TOPHAT, sim_name=f"low{dims}", dims=dims, size=size, final_sim_time=0.3, num_data_dumps=3,
               sampling=dict(seeds="4", scheme="Wigner"))
    if flags:
        cfg["cosmology"] = dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01)
    run_cli(tmp_path, cfg, *flags)
    sims = oracle_runs(cfg, expanding=bool(flags))
    assert [s.parameters.sim_name for s in sims] == [f"low{dims}-stream00004", f"low{dims}"]
    for sim in sims:
        assert len(sim.dumps) == 4
        for k, psi in sim.dumps:
            base = tmp_path / "sim-data" / sim.parameters.sim_name / f"psi_{k:05d}"
            re, im = np.load(open(str(base) + "_real", "rb")), np.load(open(str(base) + "_imag", "rb"))
            assert re.shape == (size,) * dims + (1,) * (4 - dims)
            got = (re + 1j * im).reshape((size,) * dims)
            assert rel(got, psi) < 1e-11, (sim.parameters.sim_name, k)


@pytest.mark.parametrize("size,flags", [(12, ()), (20, ("--expanding",))])
def test_non_power_of_two_edge(tmp_path, size, flags):
    # any even edge (utils/fft.rs:105): chirp-z line transforms and the unfused step behind the same process surface
    cfg = dict(TOPHAT, sim_name="npo2", size=size, final_sim_time=0.3, num_data_dumps=3, sampling=dict(seeds="7", scheme="Husimi"))
    if flags:
        cfg["cosmology"] = dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01)
    run_cli(tmp_path, cfg, *flags)
    sims = oracle_runs(cfg, expanding=bool(flags))
    assert [s.parameters.sim_name for s in sims] == ["npo2-stream00007", "npo2"]
    for sim in sims:
        assert len(sim.dumps) == 4
        for k, psi in sim.dumps:
            assert rel(load_dump(tmp_path, sim.parameters.sim_name, k), psi) < 1e-11, (sim.parameters.sim_name, k)


def test_resume_from_last_dump(tmp_path):
    # --resume: a run cut after dump 2 continues from that dump and reproduces the uninterrupted run's later dumps
    # (the reference cannot restart: simulation_object.rs:627,915)
    cfg = dict(TOPHAT, sampling=dict(seeds="1 to 2", scheme="Husimi"))
    full = tmp_path / "full"
    part = tmp_path / "part"
    full.mkdir(); part.mkdir()
    run_cli(full, cfg)
    run_cli(part, dict(cfg, final_sim_time=cfg["final_sim_time"] * 2 / 5, num_data_dumps=2))     # same dump times 1, 2
    names = ["spherical-tophat-stream00001", "spherical-tophat-stream00002", "spherical-tophat"]
    for n in names:
        for k in range(3):
            assert rel(load_dump(part, n, k), load_dump(full, n, k)) < 1e-12
    # drop stream 2's dump 2 imaginary plane: that stream must fall back to dump 1
    os.remove(part / "sim-data" / names[1] / "psi_00002_imag")
    r = run_cli(part, cfg, "--resume")
    assert f"({names[0]}) resuming from dump 2" in r.stdout and f"({names[1]}) resuming from dump 1" in r.stdout
    for n in names:
        for k in range(6):
            assert rel(load_dump(part, n, k), load_dump(full, n, k)) < 1e-11, (n, k)


def test_rank_world_partition(tmp_path):
    # --rank/--world: block partition of the seed streams over processes, the mean-field run on rank 0
    cfg = dict(TOPHAT, sampling=dict(seeds="1 to 3", scheme="Husimi"))
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(); two.mkdir()
    run_cli(one, cfg)
    r0 = run_cli(two, cfg, "--rank", "0", "--world", "2")
    r1 = run_cli(two, cfg, "--rank", "1", "--world", "2")
    assert "stream00003" not in r0.stdout and "stream00001" not in r1.stdout
    for n in ["spherical-tophat-stream00001", "spherical-tophat-stream00002", "spherical-tophat-stream00003", "spherical-tophat"]:
        for k in range(6):
            assert rel(load_dump(two, n, k), load_dump(one, n, k)) < 1e-12, (n, k)   # pair partners differ: rounding only
    run_cli(two, cfg, "--rank", "2", "--world", "2", expect=2)


def test_cold_gauss_ics_run(tmp_path):
    # [ics] ColdGauss (ics.rs:24-162) with different mean/std per axis: mean[0]/std[0] belong to the fastest axis,
    # i.e. the LAST axis of the dumped NPY (ics.rs:88-89,140-141); Husimi stream + mean-field run against the oracle
    cfg = dict(TOPHAT, sim_name="cg", final_sim_time=0.4, num_data_dumps=2,
               ics=dict(type="ColdGauss", mean=[10.0, 15.0, 21.0], std=[3.0, 5.0, 8.0]),
               sampling=dict(seeds="2", scheme="Husimi"))
    run_cli(tmp_path, cfg)
    want0 = o.cold_gauss([10.0, 15.0, 21.0], [3.0, 5.0, 8.0], 30.0 / 16, 3, 16)
    got0 = load_dump(tmp_path, "cg", 0)
    assert rel(got0, want0) < 1e-13
    # the narrowest Gaussian (std 3, mean 10) lies along the last axis of the dump
    prof = [np.abs(got0).sum(axis=tuple(a for a in range(3) if a != ax)) for ax in range(3)]
    widths = [float(np.sqrt(np.sum(p * (np.arange(16) - np.sum(p * np.arange(16)) / p.sum()) ** 2) / p.sum())) for p in prof]
    assert widths[2] < widths[1] < widths[0]
    sims = oracle_runs(cfg)
    assert [s.parameters.sim_name for s in sims] == ["cg-stream00002", "cg"]
    for sim in sims:
        assert len(sim.dumps) == 3
        for k, psi in sim.dumps:
            assert rel(load_dump(tmp_path, sim.parameters.sim_name, k), psi) < 1e-11, (sim.parameters.sim_name, k)


GAUSSIAN_MFT = dict(axis_length=30, final_sim_time=400, cfl=0.02, num_data_dumps=200, total_mass=1e10, hbar_=0.02,
                    sim_name="gaussian-overdensity-512-mft", k2_cutoff=0.95, alias_threshold=0.02, dims=3, size=512,
                    ics=dict(type="ColdGauss", mean=[15, 15, 15], std=[10, 10, 10]))


def test_gaussian_overdensity_mft_example(tmp_path):
    # examples/gaussian-overdensity-mft.toml (keys as shipped: integer-valued numbers, no [sampling] => the mean-field
    # run alone) with `size` overridden to 64: --test builds the ICs and stops (main.rs:59); a shortened run then
    # follows the oracle through its first dumps
    cfg = dict(GAUSSIAN_MFT, size=64)
    r = run_cli(tmp_path, cfg, "--test", "--verbose")
    assert "Working on simulation gaussian-overdensity-512-mft" in r.stdout and not (tmp_path / "sim-data").exists()
    short = dict(cfg, final_sim_time=4, num_data_dumps=2)       # same dump spacing as the shipped file (2 time units)
    run_cli(tmp_path, short)
    sims = oracle_runs(short)
    assert len(sims) == 1 and len(sims[0].dumps) == 3 and sims[0].parameters.n_steps >= 3
    for k, psi in sims[0].dumps:
        assert rel(load_dump(tmp_path, "gaussian-overdensity-512-mft", k), psi) < 1e-11, k


def test_streams_in_waves(tmp_path):
    # --wave: the 3 seed streams + mean-field run through the device two at a time give the dumps of the run that
    # holds all four at once (the reference runs them one after the other anyway, main.rs:43-79)
    cfg = dict(TOPHAT, sampling=dict(seeds="1 to 3", scheme="Husimi"))
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(); two.mkdir()
    run_cli(one, cfg)
    r = run_cli(two, cfg, "--wave", "2")
    assert "Finished all streams" in r.stdout
    for n in ["spherical-tophat-stream00001", "spherical-tophat-stream00002", "spherical-tophat-stream00003", "spherical-tophat"]:
        for k in range(6):
            assert rel(load_dump(two, n, k), load_dump(one, n, k)) < 1e-12, (n, k)
    run_cli(two, cfg, "--wave", "2", "--synthesize", expect=1)


def test_more_streams_than_the_device_holds_run_in_waves(tmp_path):
    # 512^3 fp32 x 400 streams is 400 GiB of state: msm_create reports MSM_ERR_NOMEM and the driver halves the wave
    # until the device holds it (--test: initial conditions and sampling of every wave, no evolution)
    cfg = dict(TOPHAT, size=512, sampling=dict(seeds="1 to 399", scheme="Husimi"))
    r = run_cli(tmp_path, cfg, "--test", "--precision", "f32")
    assert "running in waves of" in r.stdout


def test_communicator_driver_path_with_one_rank(tmp_path):
    # --comm-file at world size 1 (what a one-GPU box can form): the id file hand-off, the per-iteration flag exchange
    # and the all-reduced synthesizer run through RCCL and must reproduce the plain run's dumps and products exactly
    cfg = dict(TOPHAT, sampling=dict(seeds="1 to 2", scheme="Husimi"))
    plain, comm = tmp_path / "plain", tmp_path / "comm"
    plain.mkdir(); comm.mkdir()
    run_cli(plain, cfg, "--synthesize")
    run_cli(comm, cfg, "--synthesize", "--comm-file", str(comm / "id"))
    assert not (comm / "id").exists()                       # spent once every rank has joined
    for n in ["spherical-tophat-stream00001", "spherical-tophat-stream00002", "spherical-tophat"]:
        for k in range(6):
            assert np.array_equal(load_dump(comm, n, k), load_dump(plain, n, k)), (n, k)
    for k in range(6):
        for field in ("psi", "psi2", "psik", "psik2"):
            assert np.array_equal(load_dump(comm, "spherical-tophat-combined", k, field),
                                  load_dump(plain, "spherical-tophat-combined", k, field)), (k, field)
    with open(comm / "sim-data" / "spherical-tophat-combined" / "Qx_real", "rb") as f, \
         open(plain / "sim-data" / "spherical-tophat-combined" / "Qx_real", "rb") as g:
        assert np.array_equal(np.load(f), np.load(g))
    # aliasing with a communicator: every rank (here: the one) leaves through the collective exit, status 101
    N = 8
    idx = np.indices((N, N, N)).sum(axis=0)
    psi = o.normalize(((-1.0) ** idx).astype(np.complex128), 30.0 / N, 3)
    np.savez(tmp_path / "nyq.npz", real=psi.real, imag=psi.imag)
    bad = dict(TOPHAT, size=N, ics=dict(type="UserSpecified", path=str(tmp_path / "nyq.npz")),
               sampling=dict(seeds="1", scheme="Husimi"))
    r = run_cli(tmp_path, bad, "--comm-file", str(tmp_path / "id2"), expect=101)
    assert "simulation aliased" in r.stderr
    # argument checks
    run_cli(tmp_path, cfg, "--synthesize", "--rank", "0", "--world", "2", expect=2)             # needs --comm-file
    run_cli(tmp_path, cfg, "--resume", "--synthesize", expect=2)


def test_single_precision_expanding_run(tmp_path):
    # --precision f32 --expanding: the mode of the largest benchmark configuration through the process surface, against
    # the oracle run in single precision; dumps are then <f4 planes (utils/io.rs:63-68 with T = f32)
    cfg = dict(TOPHAT, sim_name="f32cosmo", size=32, final_sim_time=0.6, num_data_dumps=3, cfl=0.05,
               sampling=dict(seeds="3", scheme="Husimi"),
               cosmology=dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01))
    run_cli(tmp_path, cfg, "--expanding", "--precision", "f32")
    sims = []
    for params in o.simulation_iter(cfg, expanding=True, dtype=np.float32):
        sim = o.SimulationObject(params)
        sim.dump()
        while sim.not_finished():
            sim.update()
        sims.append(sim)
    assert [s.parameters.sim_name for s in sims] == ["f32cosmo-stream00003", "f32cosmo"]
    for sim in sims:
        assert len(sim.dumps) == 4 and sim.parameters.n_steps >= 6
        for k, psi in sim.dumps:
            base = tmp_path / "sim-data" / sim.parameters.sim_name / f"psi_{k:05d}"
            re, im = np.load(open(str(base) + "_real", "rb")), np.load(open(str(base) + "_imag", "rb"))
            assert re.dtype == np.float32 and re.shape == (32, 32, 32, 1)
            assert rel((re + 1j * im)[..., 0], psi) < 2e-4, (sim.parameters.sim_name, k)
