"""CPU: the C restatement (oracle/sp_cpu.c, the timed CPU-port baseline) against the
numpy oracle and the committed golden vectors.  Tolerances as for the HIP path:
1e-11 * max|psi| after 10 steps, dt relative 1e-12, identical dump flags."""
import os
import subprocess

import numpy as np
import pytest

import sp_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.exists(os.path.join(ROOT, "oracle", "libsp_cpu.so")):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
import sp_cpu  # noqa: E402

TOPHAT = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11,
              hbar_=0.05, k2_cutoff=0.95, alias_threshold=0.02)
PLANE = dict(axis_length=60.0, final_sim_time=2000.0, cfl=0.1, num_data_dumps=64, total_mass=3e16,
             hbar_=0.01, k2_cutoff=0.95, alias_threshold=0.001)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


@pytest.mark.parametrize("N", [8, 32])
def test_fft_and_potential(N):
    rng = np.random.default_rng(N)
    psi = o.normalize((rng.standard_normal((N,) * 3) + 1j * rng.standard_normal((N,) * 3)) + 2.0, 30.0 / N, 3)
    sim = sp_cpu.CpuSim(N, **TOPHAT)
    sim.set_psi(psi)
    sim.fft(-1)
    assert rel(sim.get_psi(), o.forward(psi)) < 1e-13
    sim.fft(+1)
    assert rel(sim.get_psi(), psi) < 1e-13
    toml = dict(TOPHAT, sim_name="t", dims=3, size=N, ics=dict(type="none"))
    ref = o.SimulationObject(list(o.simulation_iter(toml))[-1], psi=psi)
    want = ref.calculate_potential()
    mx = sim.potential()
    assert rel(sim.get_phi(), want) < 1e-13 and abs(mx / np.max(np.abs(want)) - 1) < 1e-13


@pytest.mark.parametrize("name", ["tophat_n8_f64", "tophat_n16_f64", "planewave_n16_mft_f64",
                                  "planewave_n16_wigner1_f64", "tophat_n32_f64"])
def test_golden_vectors(name, golden_dir):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    base = TOPHAT if name.startswith("tophat") else PLANE
    N = g["psi0"].shape[0]
    sim = sp_cpu.CpuSim(N, **base)
    sim.set_psi(g["psi0"])
    for i in range(1, 11):
        info = sim.update()
        dt, dump, alias, maxphi, time = g["scalars"][i - 1]
        assert abs(info["dt"] / dt - 1) <= 1e-12 and info["dump"] == bool(dump)
        assert abs(info["time"] / time - 1) <= 1e-12 and abs(info["max_phi"] / maxphi - 1) <= 1e-12
        assert abs(info["alias_mass"] - alias) <= max(1e-4 * alias, 1e-20)
        if f"psi_{i}" in g:
            assert rel(sim.get_psi(), g[f"psi_{i}"]) < 1e-11
            sim.potential()
            assert rel(sim.get_phi(), g[f"phi_{i}"]) < 1e-11


def test_full_run_golden(golden_dir):
    # examples/spherical-tophat.toml at its own size 16, mean-field stream, all 200 dumps
    g = np.load(os.path.join(golden_dir, "tophat_n16_run200_f64.npz"))
    sim = sp_cpu.CpuSim(16, **TOPHAT)
    sim.set_psi(g["dump_0"])
    steps, dumps, t = 0, 0, 0.0
    while t < TOPHAT["final_sim_time"]:
        info = sim.update()
        steps += 1
        t = info["time"]
        if info["dump"]:
            dumps += 1
            if dumps == 1:
                assert rel(sim.get_psi(), g["dump_1"]) < 1e-11
    assert dumps + 1 == int(g["n_dumps"]) == 201 and steps == int(g["n_steps"]) and t == float(g["final_time"])
    assert rel(sim.get_psi(), g["dump_200"]) < 1e-10


def test_dump_schedule():
    N = 8
    cfgd = dict(TOPHAT, final_sim_time=2.0, num_data_dumps=10)
    toml = dict(cfgd, sim_name="t", dims=3, size=N, ics=dict(type="SphericalTophat", radius=5.0, slope=50.0, delta=100.0))
    ref = o.SimulationObject(list(o.simulation_iter(toml))[-1])
    sim = sp_cpu.CpuSim(N, **cfgd)
    sim.set_psi(ref.psi)
    for k in range(10):
        a, b = ref.update(), sim.update()
        assert b["dump"] and a["dump"] and b["time"] == float(a["time"]) == (k + 1) * 2.0 / 10
