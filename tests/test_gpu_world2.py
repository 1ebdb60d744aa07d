"""GPU: the multi-rank paths at world size 2 on ONE GPU.

RCCL refuses two ranks on one device, so these tests bind the library to tests/fake_rccl (MSM_RCCL_LIB), a test double
that carries the same five entry points over shared memory between processes.  What runs for real is everything on
this side of the RCCL boundary -- the order and number of the collective calls of libmsm_hip.so, bench.py's N > 1 path
and msm-simulator's --comm-file loop -- against single-context runs of the same streams; a mismatch in that order ends
in the double's timeout instead of a silent hang.  The RCCL calls themselves are exercised at world size 1
(tests/test_gpu_comm.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import sp_oracle as o
from msm_amd import Evolver

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
BIN = os.path.join(ROOT, "msm_amd", "bin", "msm-simulator")


def fake_env(**extra):
    assert os.path.exists(FAKE), "build tests/fake_rccl first (python -c 'import __graft_entry__ as g; g.build()')"
    return dict(os.environ, MSM_RCCL_LIB=FAKE, **extra)


def run_ranks(cmds, env, cwd, timeout=300):
    procs = [subprocess.Popen(c, env=env, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for c in cmds]
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, so, se))
    return outs


def test_library_collectives_two_ranks(tmp_path):
    """5 streams as 3 + 2 over two contexts in two processes: mean density after every step, the synthesizer's sums,
    small all-reduces in between -- against one context that holds all five."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import world2_worker as w
    N, total, steps = 32, 5, 4
    worker = os.path.join(ROOT, "tests", "world2_worker.py")
    outs = run_ranks([[sys.executable, worker, str(r), "2", str(N), str(total), str(steps), str(tmp_path)] for r in range(2)],
                     fake_env(), str(tmp_path))
    for rc, so, se in outs:
        assert rc == 0, (rc, so[-1500:], se[-1500:])
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    with Evolver(size=N, n_streams=total, batch_streams=2, **w.TOPHAT) as ev:
        for s in range(total):
            ev.set_psi(s, w.stream_psi(N, s))
        for i in range(steps):
            ev.update()
            want = sum(np.abs(ev.get_psi(s)) ** 2 for s in range(total)) / total
            for r in (r0, r1):                                        # both ranks hold the reduced product
                assert np.allclose(r["means"][i], want, rtol=1e-11, atol=0), i
        assert np.array_equal(r0["means"], r1["means"])
        psi = np.concatenate([r0["psi"], r1["psi"]])
        # a stream does not care which context holds it -- to rounding: its Poisson solve shares a complex transform
        # with whichever stream it is paired with
        for s in range(total):
            want = ev.get_psi(s)
            assert np.max(np.abs(psi[s] - want)) < 1e-12 * np.max(np.abs(want)), s
        assert np.allclose(np.concatenate([r0["dt"], r1["dt"]]), np.array([r.dt for r in ev.status()]), rtol=1e-12, atol=0)
        for s in range(total):
            ev.synth_accumulate(s, 3)
        for f in ("psi", "psi2", "psik2"):
            assert np.allclose(r0["synth_" + f], ev.synth_fetch(3, f), rtol=1e-12, atol=1e-14), f
            assert np.array_equal(r0["synth_" + f], r1["synth_" + f])
        assert int(r0["count3"]) == total and int(r1["count3"]) == total
        assert int(r0["count7"]) == 3 and int(r1["count7"]) == 3        # rank 1 contributed zeros to slot 7
        assert abs(complex(r0["qx"]) - ev.synth_qx(3)) < 1e-12 * abs(ev.synth_qx(3)) + 1e-15


def test_bench_two_ranks_share_the_gpu(tmp_path):
    """bench.py --gpus 2 self-launched with both ranks on GPU 0 (MSM_BENCH_SHARE_GPU): communicator id through the
    store, density all-reduce every step, the ranks' agreement on steps and aliasing, one JSON line from rank 0."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--streams", "6", "--batch", "2",
                        "--steps", "3", "--warmup", "1"], env=fake_env(MSM_BENCH_SHARE_GPU="1"), cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["density_allreduce"] is True and d["config"]["streams_per_gpu"] == 3
    assert d["value"] > 0 and "extra_configs" not in d and "cpu_baseline" not in d
    assert r.stderr.count("bench.py rank ") == 2 and "fake_rccl" in r.stderr     # one diagnostic line per rank


def test_bench_real_rccl_world_1_prints_one_line(tmp_path):
    """bench.py's multi-rank path through the REAL librccl with a communicator of one (MSM_BENCH_FORCE_DIST): RCCL writes
    a version banner to the process's stdout when a communicator is formed; rank 0's stdout must still carry exactly
    one line, the JSON."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "64", "--streams", "4", "--batch", "2", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline"], env=dict(os.environ, MSM_BENCH_FORCE_DIST="1"), cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["density_allreduce"] is True and d["value"] > 0
    assert "RCCL from librccl" in r.stderr


def test_bench_two_ranks_under_torchrun(tmp_path):
    """The same through the launcher the round-end driver uses (python -m torch.distributed.run ... bench.py --gpus N):
    the agent hosts the store, every rank is its client."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--streams", "5",
                        "--batch", "2", "--steps", "2", "--warmup", "1"], env=fake_env(MSM_BENCH_SHARE_GPU="1"), cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["streams_per_gpu"] == 3 and d["value"] > 0      # ragged: 3 + 2


def _toml(tmp_path, name, **over):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_cli import TOPHAT, toml_text
    cfg = dict(TOPHAT, sim_name=name, final_sim_time=0.6, num_data_dumps=3, sampling=dict(seeds="1 to 3", scheme="Husimi"), **over)
    path = tmp_path / f"{name}.toml"
    path.write_text(toml_text(cfg))
    return cfg, path


def _load(tmp_path, sim, what, k, parts=("_real", "_imag")):
    arrs = [np.load(open(tmp_path / "sim-data" / sim / f"{what}_{k:05d}{p}", "rb")) for p in parts]
    return arrs[0][..., 0] + 1j * arrs[1][..., 0]


def test_cli_mean_density_world_1(tmp_path):
    """msm-simulator --mean-density through a communicator of one (real RCCL): the per-step product written at the
    steps that land every stream on a dump equals the synthesizer's psi2 of that dump (synthesizer/src/lib.rs:217-229)
    from the same run; the mean-field run still writes its dumps, from a context of its own."""
    cfg, path = _toml(tmp_path, "md1")
    r = subprocess.run([BIN, "--toml", str(path), "--synthesize", "--mean-density", "--comm-file", str(tmp_path / "c.id")],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    for k in (1, 2, 3):
        with open(tmp_path / "sim-data" / "md1-combined" / f"mean_density_{k:05d}", "rb") as f:
            got = np.load(f)
        assert got.shape == (16, 16, 16, 1) and got.dtype == np.float64
        want = _load(tmp_path, "md1-combined", "psi2", k).real
        assert np.allclose(got[..., 0], want, rtol=1e-12, atol=0), k
    assert (tmp_path / "sim-data" / "md1" / "psi_00003_real").exists()          # the mean-field run
    sims = o.simulation_iter(cfg)
    mft = o.SimulationObject(list(sims)[-1])
    while mft.not_finished():
        mft.update()
    assert np.max(np.abs(_load(tmp_path, "md1", "psi", 3) - mft.psi)) < 1e-11 * np.max(np.abs(mft.psi))


@pytest.mark.parametrize("expanding", [False, True], ids=["static", "expanding"])
def test_cli_two_ranks_match_one(tmp_path, expanding):
    """msm-simulator --rank r --world 2 --comm-file ... --synthesize --mean-density on one GPU against the same TOML
    run by one process: identical dumps, ensemble products equal to rounding of the sums' order (static and expanding
    boxes: the expanding loop keeps its host scale-factor state per rank)."""
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir()
    two.mkdir()
    over = dict(cosmology=dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01)) if expanding else {}
    cfg, path = _toml(tmp_path, "w2", **over)
    flags = ["--expanding"] if expanding else []
    r = subprocess.run([BIN, "--toml", str(path), "--synthesize", "--mean-density", "--comm-file", str(one / "c.id"), *flags],
                       cwd=one, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    outs = run_ranks([[BIN, "--toml", str(path), "--synthesize", "--mean-density", "--rank", str(rk), "--world", "2",
                       "--comm-file", str(two / "c.id"), *flags] for rk in range(2)], fake_env(), str(two))
    for rc, so, se in outs:
        assert rc == 0, (rc, so[-1500:], se[-1500:])
    for sim in ("w2", "w2-stream00001", "w2-stream00002", "w2-stream00003"):
        for k in range(4):
            assert np.array_equal(_load(one, sim, "psi", k), _load(two, sim, "psi", k)), (sim, k)
    for k in range(4):
        for f in ("psi", "psi2", "psik", "psik2"):
            a, b = _load(one, "w2-combined", f, k), _load(two, "w2-combined", f, k)
            assert np.max(np.abs(a - b)) <= 1e-13 * np.max(np.abs(a)), (f, k)
    for k in (1, 2, 3):
        a = np.load(open(one / "sim-data" / "w2-combined" / f"mean_density_{k:05d}", "rb"))
        b = np.load(open(two / "sim-data" / "w2-combined" / f"mean_density_{k:05d}", "rb"))
        assert np.allclose(a, b, rtol=1e-13, atol=0), k
    qa = np.load(open(one / "sim-data" / "w2-combined" / "Qx_real", "rb"))
    qb = np.load(open(two / "sim-data" / "w2-combined" / "Qx_real", "rb"))
    assert np.allclose(qa, qb, rtol=1e-10, atol=1e-16)


def test_cli_failing_rank_ends_the_other(tmp_path):
    """Rank 1 cannot create its context (a device that does not exist).  It leaves its note beside the communicator
    file; rank 0 -- by then waiting for it inside the communicator's formation -- is ended by its watchdog.  Both exit
    non-zero within seconds instead of rank 0 waiting for ever."""
    cfg, path = _toml(tmp_path, "fail")
    cmds = [[BIN, "--toml", str(path), "--rank", "0", "--world", "2", "--comm-file", str(tmp_path / "c.id")],
            [BIN, "--toml", str(path), "--rank", "1", "--world", "2", "--comm-file", str(tmp_path / "c.id"), "--device", "97"]]
    outs = run_ranks(cmds, fake_env(), str(tmp_path), timeout=120)
    assert outs[1][0] == 1 and "msm_create failed" in outs[1][2]
    assert outs[0][0] == 1 and "another rank failed" in outs[0][2] and "rank 1" in outs[0][2]


@pytest.mark.parametrize("expanding", [False, True], ids=["static", "expanding"])
def test_python_driver_two_ranks_through_the_library_communicator(tmp_path, expanding):
    """python -m msm_amd.run under torch.distributed.run on two ranks sharing GPU 0: flags and the synthesizer's sums go
    through msm_comm_* (bound to the stand-in), only the communicator id through the launcher's store; dumps and
    -combined/ products equal msm-simulator's single-process run of the same TOML, static and expanding."""
    import socket
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir()
    two.mkdir()
    over = dict(cosmology=dict(omega_matter_now=0.7, omega_radiation_now=0.0, h=0.7, z0=1.0, max_dloga=0.01)) if expanding else {}
    cfg, path = _toml(tmp_path, "py2", **over)
    flags = ["--expanding"] if expanding else []
    r = subprocess.run([BIN, "--toml", str(path), "--synthesize", *flags], cwd=one, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = fake_env(PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "msm_amd.run", "--toml", str(path), "--synthesize", "--backend", "rccl",
                        "--share-gpu", *flags], cwd=two, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    for sim in ("py2", "py2-stream00001", "py2-stream00002", "py2-stream00003"):
        for k in range(4):
            a, b = _load(one, sim, "psi", k), _load(two, sim, "psi", k)
            assert np.max(np.abs(a - b)) <= 1e-12 * np.max(np.abs(a)), (sim, k)
    for k in range(4):
        for f in ("psi", "psi2", "psik", "psik2"):
            a, b = _load(one, "py2-combined", f, k), _load(two, "py2-combined", f, k)
            assert np.max(np.abs(a - b)) <= 1e-12 * np.max(np.abs(a)), (f, k)
