"""GPU: msm_synth_qx_noalloc (the entry point that allocates nothing) against msm_synth_qx.

Both reduce the same slot; the no-alloc variant keeps its partial sums in the library's fixed per-GPU workspace
and writes {Re Qx, Im Qx} to a caller-owned, GPU-addressable buffer (pinned host memory here).  The bar is
bit-exact equality: the two paths add the same partial sums in the same order.  Two calls in a row share the
workspace, the second on a smaller grid, so stale contents of the first must not leak into it."""
import ctypes as C

import numpy as np
import pytest

import sp_oracle as o
from msm_amd import Evolver, MsmError

pytestmark = pytest.mark.gpu

TOPHAT = dict(axis_length=30.0, final_sim_time=40.0, cfl=0.5, num_data_dumps=200, total_mass=1e11,
              hbar_=0.05, k2_cutoff=0.95, alias_threshold=0.02)


def _qx_both_ways(ev, n_streams, N, dims, seed0, out_ptr, out):
    rng = np.random.default_rng(seed0)
    shape = (N,) * dims
    for s in range(n_streams):
        psi = o.normalize(1.0 + 0.3 * (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)), 30.0 / N, dims)
        ev.set_psi(s, psi.astype(ev.cplx))
        ev.synth_accumulate(s, 0)
    want = ev.synth_qx(0)
    out[:] = np.nan
    ev.synth_qx_noalloc(0, out_ptr)
    ev.sync()
    return want, complex(out[0], out[1])


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_qx_noalloc_matches_qx_two_calls_second_smaller(precision):
    # The library holds one workspace per GPU and precision (the double- and single-precision engines are separate
    # translation units), so both contexts of a pair have the same precision: the second call really finds the first
    # call's partial sums in the array it is about to use.
    with Evolver(size=64, n_streams=3, precision=precision, **TOPHAT) as big, \
            Evolver(size=12, n_streams=2, dims=2, precision=precision, **TOPHAT) as small:
        p = C.c_void_p()
        big._check(big._lib.msm_host_alloc(16, C.byref(p)), "msm_host_alloc")
        out = np.frombuffer((C.c_char * 16).from_address(p.value), dtype=np.float64)
        try:
            want1, got1 = _qx_both_ways(big, 3, 64, 3, 1, p.value, out)
            want2, got2 = _qx_both_ways(small, 2, 12, 2, 2, p.value, out)     # 144 cells: most partial sums are empty
            assert want1 == got1 and np.isfinite(got1.real) and abs(got1.real) > 0
            assert want2 == got2 and np.isfinite(got2.real) and abs(got2.real) > 0
            assert got1 != got2
            # and once more on the first context: the workspace now holds the small grid's partial sums
            big.synth_qx_noalloc(0, p.value)
            big.sync()
            assert complex(out[0], out[1]) == want1
            with pytest.raises(MsmError):
                big.synth_qx_noalloc(5, p.value)          # slot not open
            with pytest.raises(MsmError):
                big.synth_qx_noalloc(0, None)             # null output
        finally:
            big._lib.msm_host_free(p)
