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
