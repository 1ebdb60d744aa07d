"""GPU: the library's own RCCL path (msm_comm_*, msm_mean_density*, msm_synth_allreduce) at world size 1 -- one GPU is
what the test box has; the partition logic for more ranks is covered on the CPU (tests/test_sharding.py).  At one
rank an all-reduce is the identity, so the products must equal what the downloads give."""
import numpy as np
import pytest

import sp_oracle as o
from msm_amd import Evolver, MsmError

pytestmark = pytest.mark.gpu

TOPHAT = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11,
              hbar_=0.05, k2_cutoff=0.95, alias_threshold=0.02)


def _fill(ev, ns, N, dims, seed):
    rng = np.random.default_rng(seed)
    shape = (N,) * dims
    for s in range(ns):
        psi = o.normalize(1.0 + 0.05 * (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)), 30.0 / N, dims)
        ev.set_psi(s, psi)


@pytest.mark.parametrize("N,dims,ns,batch", [(32, 3, 5, 2), (256, 3, 3, 2), (48, 2, 3, 2)])
def test_mean_density_each_step_world_1(N, dims, ns, batch):
    cid = Evolver.comm_unique_id()
    assert len(cid) == 128 and any(cid)
    with Evolver(size=N, dims=dims, n_streams=ns, batch_streams=batch, **dict(TOPHAT, cfl=0.05)) as ev:
        _fill(ev, ns, N, dims, N)
        with pytest.raises(MsmError):
            ev.mean_density_begin(ns)                     # no communicator yet
        ev.comm_init(cid, 0, 1)
        with pytest.raises(MsmError):
            ev.comm_init(cid, 0, 1)                       # twice
        assert ev.comm_allreduce([1.5, -2.0, 7.0], Evolver.COMM_SUM) == [1.5, -2.0, 7.0]
        assert ev.comm_allreduce([3.0], Evolver.COMM_MAX) == [3.0]
        ev.mean_density_begin(ns)
        with pytest.raises(MsmError):
            ev.mean_density()                             # no step yet
        with pytest.raises(MsmError):
            ev.step_density(None)                         # the density target is taken
        for i in range(3):                                # both buffers and a reuse
            ev.update()
            got = ev.mean_density()
            want = sum(np.abs(ev.get_psi(s)) ** 2 for s in range(ns)) / ns
            assert np.allclose(got, want, rtol=1e-11, atol=0), (i, float(np.abs(got - want).max()))
        assert ev.mean_density(to_host=False)


def test_synth_allreduce_world_1():
    N, ns = 16, 3
    with Evolver(size=N, n_streams=ns, **TOPHAT) as ev:
        _fill(ev, ns, N, 3, 5)
        ev.comm_init(Evolver.comm_unique_id(), 0, 1)
        for s in range(ns):
            ev.synth_accumulate(s, 2)
        before = {f: ev.synth_fetch(2, f) for f in ev.SYNTH_FIELDS}
        qx = ev.synth_qx(2)
        ev.synth_allreduce(2)
        assert ev.synth_device(2)[1] == ns
        for f in ev.SYNTH_FIELDS:
            assert np.array_equal(ev.synth_fetch(2, f), before[f])
        assert ev.synth_qx(2) == qx
        ev.synth_allreduce(4)                             # a slot nothing was added to: zeros, count 0
        assert ev.synth_device(4)[1] == 0
        ev.synth_release(2)
        ev.synth_release(4)
