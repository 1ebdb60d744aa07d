"""GPU: the in-situ synthesizer products against the numpy restatement of the reference's
post-processor applied to the same streams' downloaded dumps (synthesizer/src/main.rs:63-187).
Tolerance: 1e-12 relative to each field's maximum (fp64), 1e-5 (fp32 streams, fp64 sums)."""
import os

import numpy as np
import pytest

import sp_oracle as o
from msm_amd import Evolver
from msm_amd.synth import FIELDS, InSituSynthesizer

pytestmark = pytest.mark.gpu

PLANE = dict(axis_length=60.0, final_sim_time=2e-5, cfl=0.1, num_data_dumps=2, total_mass=3e16,
             hbar_=0.01, ntot=1e10, k2_cutoff=0.95, alias_threshold=0.001)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def load_complex(base):
    with open(base + "_real", "rb") as f:
        re = np.load(f)
    with open(base + "_imag", "rb") as f:
        im = np.load(f)
    assert re.dtype == np.float64 and re.ndim == 4
    return re + 1j * im


@pytest.mark.parametrize("precision,tol", [("f64", 1e-12), ("f32", 2e-5)])
def test_products_per_dump_small(tmp_path, precision, tol):
    N, ns = 16, 3
    psi0 = o.plane_wave_ics(N)
    cdt = np.complex128 if precision == "f64" else np.complex64
    with Evolver(size=N, n_streams=ns, precision=precision, **PLANE) as ev:
        ev.set_psi(0, psi0.astype(cdt))
        for s in range(1, ns):
            ev.copy_psi(s, 0)
        for s in range(ns):
            ev.sample(s, "Wigner", 11 + s)
        dv = ev.derived()["dx"] ** 3
        syn = InSituSynthesizer(ev, "pw", ns, PLANE["num_data_dumps"], out_root=str(tmp_path))
        want = {}
        held = {0: [ev.get_psi(s).astype(np.complex128) for s in range(ns)]}
        for s in range(ns):
            syn.add(s, 0)                                   # dump 0 = initial conditions (main.rs:61)
        syn.finish(0)
        seen = {s: 0 for s in range(ns)}
        for _ in range(200):
            ev.update()
            st = ev.status()
            for s in range(ns):
                if st[s].dump and st[s].current_dumps > seen[s]:
                    seen[s] = k = st[s].current_dumps
                    held.setdefault(k, []).append(ev.get_psi(s).astype(np.complex128))
                    syn.add(s, k)
                    if syn.ready(k, ns):
                        syn.finish(k)
            if not any(r.active for r in st):
                break
        syn.close()
        assert sorted(held) == [0, 1, 2] and all(len(v) == ns for v in held.values())
        qx = load_complex(os.path.join(str(tmp_path), "sim-data", "pw-combined", "Qx"))
        assert qx.shape == (3, 1, 1, 1)
        for k, psis in held.items():
            want = o.reference_synthesis(psis, dv)
            for f in FIELDS:
                got = load_complex(os.path.join(str(tmp_path), "sim-data", "pw-combined", f"{f}_{k:05d}"))
                assert got.shape == (N, N, N, 1)
                assert rel(got[..., 0], want[f]) < tol, (k, f)
            assert abs(qx[k, 0, 0, 0] - want["Qx"]) <= tol * max(abs(want["Qx"]), np.sum(np.abs(psis[0]) ** 2) * dv)


def test_products_two_pass_engine_256():
    """The 256^3 engine keeps psi_k transposed on the device; the products must not notice."""
    N, ns = 256, 2
    rng = np.random.default_rng(3)
    with Evolver(size=N, n_streams=ns, **dict(PLANE, final_sim_time=2000.0)) as ev:
        for s in range(ns):
            psi = 1.0 + 0.05 * (rng.standard_normal((N, N, N)) + 1j * rng.standard_normal((N, N, N)))
            ev.set_psi(s, o.normalize(psi, 60.0 / N, 3))
        ev.update()
        assert ev.derived()["transform_passes_mode"] == 2.0
        psis = [ev.get_psi(s) for s in range(ns)]
        for s in range(ns):
            ev.synth_accumulate(s, 7)
        want = o.reference_synthesis(psis, ev.derived()["dx"] ** 3)
        for f in FIELDS:
            assert rel(ev.synth_fetch(7, f), want[f]) < 1e-12, f
        qx = ev.synth_qx(7)
        assert abs(qx - want["Qx"]) <= 1e-12 * np.sum(np.abs(psis[0]) ** 2) * ev.derived()["dx"] ** 3
        ev.synth_release(7)
        with pytest.raises(Exception):
            ev.synth_fetch(7, "psi")


_DIST_SCRIPT = r"""
import os, sys
import numpy as np
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import sp_oracle as o
from msm_amd import Evolver
from msm_amd.comm import LibraryComm
from msm_amd.synth import FIELDS, InSituSynthesizer
N, ns = 32, 2
base = dict(axis_length=60.0, final_sim_time=2000.0, cfl=0.1, num_data_dumps=4, total_mass=3e16, hbar_=0.01,
            ntot=1e10, k2_cutoff=0.95, alias_threshold=0.001)
with Evolver(size=N, n_streams=ns, **base) as ev:
    ev.set_psi(0, o.plane_wave_ics(N)); ev.copy_psi(1, 0)
    for s in range(ns):
        ev.sample(s, "Wigner", 5 + s)
    ev.update()
    psis = [ev.get_psi(s) for s in range(ns)]
    comm = LibraryComm(ev, 0, 1, lambda cid: cid)      # a communicator of one: RCCL inside the library
    assert comm.allreduce([2.5, -1.0], 2) == [2.5, -1.0]
    syn = InSituSynthesizer(ev, "d", ns, 4, out_root=sys.argv[2], comm=comm)
    for s in range(ns):
        syn.add(s, 1)
    syn.finish(1)                 # msm_synth_allreduce: the device sums and the stream count over RCCL, in place
    syn.close()
want = o.reference_synthesis(psis, (60.0 / N) ** 3)
for f in FIELDS:
    b = os.path.join(sys.argv[2], "sim-data", "d-combined", f"{f}_00001")
    got = np.load(open(b + "_real", "rb"))[..., 0] + 1j * np.load(open(b + "_imag", "rb"))[..., 0]
    assert np.max(np.abs(got - want[f])) < 1e-12 * np.max(np.abs(want[f])), f
print("SYNTH_DIST_OK")
"""


def test_reduction_path_over_rccl(tmp_path):
    """finish() with the library's communicator: the device sums go through RCCL in place (msm_synth_allreduce)."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "synth_dist.py"
    script.write_text(_DIST_SCRIPT)
    r = subprocess.run([sys.executable, str(script), root, str(tmp_path), str(port)], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "SYNTH_DIST_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_multi_process_driver_matches_cli(tmp_path):
    """python -m msm_amd.run on two ranks (3 seed streams split 2 + 1, mean-field run on rank 0, gloo over the one
    GPU of the test box): dumps equal the single-process msm-simulator's, the -combined/ products equal the
    reference synthesizer's arithmetic on those dumps."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    toml = tmp_path / "run.toml"
    toml.write_text("""axis_length = 30.0
final_sim_time = 1.0
cfl = 0.5
num_data_dumps = 5
total_mass = 1e11
ntot = 1e6
hbar_ = 0.05
sim_name = "spherical-tophat"
k2_cutoff = 0.95
alias_threshold = 0.02
dims = 3
size = 16

[ics]
type = "SphericalTophat"
radius = 5.0
slope = 50.0
delta = 100.0

[sampling]
seeds = "1 to 3"
scheme = "Husimi"
""")
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(); two.mkdir()
    r = subprocess.run([os.path.join(root, "msm_amd", "bin", "msm-simulator"), "--toml", str(toml)], cwd=one, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "msm_amd.run", "--toml", str(toml), "--synthesize", "--backend", "gloo", "--share-gpu"],
                       cwd=two, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    names = [f"spherical-tophat-stream{k:05d}" for k in (1, 2, 3)]
    for n in names + ["spherical-tophat"]:
        for k in range(6):
            a = load_complex(str(two / "sim-data" / n / f"psi_{k:05d}"))
            b = load_complex(str(one / "sim-data" / n / f"psi_{k:05d}"))
            assert rel(a, b) < 1e-12, (n, k)
    dv = (30.0 / 16) ** 3
    qx = load_complex(str(two / "sim-data" / "spherical-tophat-combined" / "Qx"))
    assert qx.shape == (6, 1, 1, 1)
    for k in range(6):
        want = o.reference_synthesis([load_complex(str(one / "sim-data" / n / f"psi_{k:05d}"))[..., 0] for n in names], dv)
        for f in FIELDS:
            got = load_complex(str(two / "sim-data" / "spherical-tophat-combined" / f"{f}_{k:05d}"))[..., 0]
            assert rel(got, want[f]) < 1e-11, (k, f)
        assert abs(qx[k, 0, 0, 0] - want["Qx"]) < 1e-11
