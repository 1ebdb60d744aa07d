"""GPU: msm_step_density / msm_density_sum on every engine (fused 3-D power-of-two, unfused 1-D/2-D and
non-power-of-two, 256^3 two-pass), against the downloaded wavefunctions.  The targets are pinned host planes from
msm_host_alloc, which the GPU addresses directly, so no other device library is needed in this process."""
import ctypes as C

import numpy as np
import pytest

import sp_oracle as o
from msm_amd import Evolver

pytestmark = pytest.mark.gpu

TOPHAT = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11,
              hbar_=0.05, k2_cutoff=0.95, alias_threshold=0.02)


class PinnedF64:
    def __init__(self, ev, n):
        self.ev, self.p = ev, C.c_void_p()
        ev._check(ev._lib.msm_host_alloc(8 * n, C.byref(self.p)), "msm_host_alloc")
        self.a = np.frombuffer((C.c_char * (8 * n)).from_address(self.p.value), dtype=np.float64)
        self.a[:] = -1.0

    def free(self):
        self.ev._lib.msm_host_free(self.p)


@pytest.mark.parametrize("N,dims,ns,batch,precision", [(32, 3, 5, 2, "f64"), (96, 3, 3, 2, "f64"), (64, 2, 4, 4, "f64"),
                                                       (100, 1, 3, 2, "f64"), (256, 3, 3, 2, "f64"), (128, 3, 3, 2, "f32"),
                                                       (512, 3, 3, 2, "f32")])
def test_step_density_against_downloads(N, dims, ns, batch, precision):
    # final time two dump intervals away: every stream takes exactly two (dump-limited) steps and finishes in the
    # second one, whose density must still be part of the product
    base = dict(TOPHAT, final_sim_time=0.02, num_data_dumps=2, cfl=0.5)
    rng = np.random.default_rng(N + dims)
    shape = (N,) * dims
    # single precision: the 512^3 tiling sums the density from the |psi|^2 grids it writes
    tol = 1e-11 if precision == "f64" else 1e-5
    with Evolver(size=N, dims=dims, n_streams=ns, batch_streams=batch, precision=precision, **base) as ev:
        for s in range(ns):
            psi = o.normalize(1.0 + 0.05 * (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)), 30.0 / N, dims)
            ev.set_psi(s, psi.astype(ev.cplx))
        step, whole = PinnedF64(ev, N ** dims), PinnedF64(ev, N ** dims)
        try:
            ev.density_sum(whole.p.value)
            ev.sync()
            want0 = sum(np.abs(ev.get_psi(s).astype(np.complex128)) ** 2 for s in range(ns)).reshape(-1)
            assert np.allclose(whole.a, want0, rtol=1e-13 if precision == "f64" else 1e-6, atol=0)
            ev.step_density(step.p.value)
            for i in range(2):
                ev.update()
                ev.sync()
                st = ev.status()
                assert all(r.n_steps == i + 1 and r.dump == 1 for r in st)
                want = sum(np.abs(ev.get_psi(s).astype(np.complex128)) ** 2 for s in range(ns)).reshape(-1)
                assert np.allclose(step.a, want, rtol=tol, atol=0), (i, float(np.abs(step.a - want).max()))
            assert all(r.active == 0 for r in ev.status())
            ev.step_density(None)
            ev.density_sum(whole.p.value)
            ev.sync()
            assert np.allclose(whole.a, want, rtol=tol, atol=0)
            # a further step advances nobody: the product of a step without streams is zero
            ev.step_density(step.p.value)
            ev.update()
            ev.sync()
            assert np.all(step.a == 0.0)
            ev.step_density(None)
        finally:
            ev.sync()
            step.free()
            whole.free()
