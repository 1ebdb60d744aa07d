"""CPU: multi-rank path (stream partition + ensemble-density all-reduce) with gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest

from msm_amd.sharding import local_seeds, partition_streams


def test_partition_covers_every_stream_once():
    for n, world in [(256, 8), (256, 1), (5, 2), (3, 4), (64, 3), (0, 2), (7, 7)]:
        blocks = [partition_streams(n, world, r) for r in range(world)]
        owned = [s for first, count in blocks for s in range(first, first + count)]
        assert owned == list(range(n)), (n, world)
        assert max(c for _, c in blocks) == -(-n // world) if n else True
    assert [partition_streams(256, 8, r) for r in (0, 7)] == [(0, 32), (224, 32)]
    assert local_seeds(list(range(1, 6)), 2, 1) == [4, 5]
    with pytest.raises(ValueError):
        partition_streams(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_streams, N, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    import torch
    import torch.distributed as dist
    import sp_oracle as o
    from msm_amd.sharding import allreduce_density, local_seeds

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base = o.plane_wave_ics(N)
    dens = torch.zeros(N ** 3, dtype=torch.float64)
    for seed in local_seeds(list(range(1, n_streams + 1)), world, rank):
        psi = o.sample_quantum_perturbation(base, 60.0 / N, 3, 3e16, 3e6, "Wigner", seed)
        dens += torch.from_numpy((np.abs(psi) ** 2).reshape(-1))
    work = allreduce_density(dist, dens, async_op=True)
    work.wait()
    if rank == 0:
        np.save(out, dens.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_density_allreduce_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    import sp_oracle as o

    n_streams, N = 5, 8          # ragged: 3 + 2 streams
    out = str(tmp_path / "dens.npy")
    mp.spawn(_worker, args=(2, _free_port(), n_streams, N, out), nprocs=2, join=True)
    got = np.load(out)
    base = o.plane_wave_ics(N)
    want = np.zeros(N ** 3)
    for seed in range(1, n_streams + 1):
        want += (np.abs(o.sample_quantum_perturbation(base, 60.0 / N, 3, 3e16, 3e6, "Wigner", seed)) ** 2).reshape(-1)
    assert np.allclose(got, want, rtol=1e-14, atol=0)


class _FakeSynthEvolver:
    """Host stand-in with the Evolver's synth_* surface, numpy sums (driver logic only)."""

    def __init__(self, psis, N):
        self.psis, self.shape, self.slots = psis, (N, N, N), {}

    def synth_accumulate(self, stream, slot):
        n3 = int(np.prod(self.shape))
        sums, cnt = self.slots.setdefault(slot, [np.zeros(4 * n3, dtype=np.complex128), 0])
        p = self.psis[stream].reshape(-1)
        k = np.fft.fftn(self.psis[stream]).reshape(-1)
        sums[0:n3] += p
        sums[n3:2 * n3] += np.abs(p) ** 2
        sums[2 * n3:3 * n3] += k
        sums[3 * n3:] += np.abs(k) ** 2
        self.slots[slot][1] = cnt + 1

    def synth_device(self, slot):
        return self.slots[slot][0], self.slots[slot][1]

    def synth_set_count(self, slot, count):
        self.slots[slot][1] = count

    def synth_fetch(self, slot, field):
        from msm_amd.synth import FIELDS
        n3, i = int(np.prod(self.shape)), FIELDS.index(field)
        return (self.slots[slot][0][i * n3:(i + 1) * n3] / self.slots[slot][1]).reshape(self.shape)

    def synth_qx(self, slot):
        n3, (sums, cnt) = int(np.prod(self.shape)), self.slots[slot]
        return complex(np.sum(sums[n3:2 * n3] / cnt - np.abs(sums[:n3] / cnt) ** 2) * 0.5)

    def synth_release(self, slot):
        del self.slots[slot]


def _synth_worker(rank, world, port, n_streams, N, root_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    from msm_amd.sharding import partition_streams
    from msm_amd.synth import InSituSynthesizer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = partition_streams(n_streams, world, rank)
    rng = np.random.default_rng(0)
    allpsi = [rng.standard_normal((N, N, N)) + 1j * rng.standard_normal((N, N, N)) for _ in range(n_streams)]
    ev = _FakeSynthEvolver(allpsi[first:first + count], N)
    from msm_amd.comm import GlooComm
    syn = InSituSynthesizer(ev, "run", n_streams, 1, out_root=root_dir,
                            comm=GlooComm(dist, to_tensor=lambda arr, n: torch.from_numpy(arr.view(np.float64))))
    for dump in (0, 1):
        for s in range(count):
            syn.add(s, dump)
        assert syn.ready(dump, count)
        syn.finish(dump)
    syn.close()
    dist.barrier()
    dist.destroy_process_group()


def test_in_situ_synthesizer_two_ranks(tmp_path):
    """The synthesizer driver over two ranks (3 + 2 streams): sums and counts are all-reduced,
    rank 0 writes the reference's `-combined/` layout."""
    import torch.multiprocessing as mp
    import sp_oracle as o

    n_streams, N = 5, 8
    mp.spawn(_synth_worker, args=(2, _free_port(), n_streams, N, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(0)
    allpsi = [rng.standard_normal((N, N, N)) + 1j * rng.standard_normal((N, N, N)) for _ in range(n_streams)]
    want = o.reference_synthesis(allpsi, 0.5)
    base = tmp_path / "sim-data" / "run-combined"
    for f in ("psi", "psi2", "psik", "psik2"):
        for dump in (0, 1):
            re = np.load(open(base / f"{f}_{dump:05d}_real", "rb"))
            im = np.load(open(base / f"{f}_{dump:05d}_imag", "rb"))
            assert re.shape == (N, N, N, 1) and re.dtype == np.float64
            assert np.allclose(re[..., 0] + 1j * im[..., 0], want[f], rtol=1e-12, atol=1e-12)
    qx = np.load(open(base / "Qx_real", "rb"))
    assert qx.shape == (2, 1, 1, 1) and np.allclose(qx[:, 0, 0, 0], want["Qx"].real, rtol=1e-12)


class _FakeStatus:
    def __init__(self):
        self.active, self.aliased, self.dump, self.current_dumps, self.alias_mass, self.n_steps = 1, 0, 0, 0, 0.0, 0


class _FakeStepper:
    """Host stand-in for the Evolver in driver-logic tests: every update is a dump step; the stream sampled with
    `alias_seed` reports aliasing in its second step (as a device context would, simulation_object.rs:607-617)."""
    alias_seed = 2

    def __init__(self, *, size, n_streams, dims=3, num_data_dumps=1, **_):
        self.shape, self.n_streams, self.nd = (size,) * dims, n_streams, num_data_dumps
        self.st = [_FakeStatus() for _ in range(n_streams)]
        self.psi = [np.zeros(self.shape, complex) for _ in range(n_streams)]
        self.seeds = [None] * n_streams

    def derived(self):
        return {"dx": 1.0}

    def set_psi(self, s, psi):
        self.psi[s] = np.array(psi)

    def copy_psi(self, dst, src):
        self.psi[dst] = self.psi[src].copy()

    def sample(self, s, scheme, seed):
        self.seeds[s] = seed

    def update(self):
        for s, r in enumerate(self.st):
            if not r.active:
                r.dump = 0
                continue
            r.n_steps += 1
            if self.seeds[s] == self.alias_seed and r.n_steps == 2:
                r.aliased, r.active, r.alias_mass, r.dump = 1, 0, 0.5, 0
                continue
            r.dump, r.current_dumps = 1, r.current_dumps + 1
            r.active = 1 if r.current_dumps < self.nd else 0

    def status(self):
        return self.st

    def get_psi(self, s):
        return self.psi[s]

    def close(self):
        pass


def _abort_worker(rank, world, port, toml, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import msm_amd.evolver
    import msm_amd.run
    msm_amd.evolver.Evolver = _FakeStepper
    rc = msm_amd.run.main(["--toml", toml, "--backend", "gloo", "--out", out_dir])
    with open(os.path.join(out_dir, f"rc{rank}"), "w") as f:
        f.write(str(rc))


@pytest.mark.timeout(180)
def test_alias_on_one_rank_ends_every_rank(tmp_path):
    """Two ranks, seeds 1 | 2 (+ the mean-field run on rank 0), six dumps: the stream on rank 1 aliases in its second
    step.  Both ranks must leave the stepping loop in the same iteration and return the reference's panic status
    instead of rank 0 running on into a collective that rank 1 never enters."""
    import torch.multiprocessing as mp
    toml = tmp_path / "run.toml"
    toml.write_text('axis_length = 30.0\nfinal_sim_time = 40.0\ncfl = 0.5\nnum_data_dumps = 6\ntotal_mass = 1e11\n'
                    'hbar_ = 0.05\nsim_name = "abort"\nk2_cutoff = 0.95\nalias_threshold = 0.02\ndims = 3\nsize = 8\n'
                    '[ics]\ntype = "SphericalTophat"\nradius = 5.0\ndelta = 100.0\nslope = 50.0\n'
                    '[sampling]\nscheme = "Wigner"\nseeds = "1, 2"\n')
    mp.spawn(_abort_worker, args=(2, _free_port(), str(toml), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "rc0").read_text() == "101" and (tmp_path / "rc1").read_text() == "101"
    # rank 0's streams wrote dumps 0..2 at most (the abort is seen one step after the aliasing), never all six
    written = sorted(p.name for p in (tmp_path / "sim-data" / "abort-stream00001").iterdir())
    assert "psi_00001_real" in written and "psi_00006_real" not in written
