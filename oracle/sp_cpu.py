"""ctypes handle on oracle/libsp_cpu.so (the C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(os.path.join(_HERE, "libsp_cpu.so"))
        l.sp_create.restype = C.c_void_p
        l.sp_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                C.c_double, C.c_double, C.c_double]
        l.sp_destroy.argtypes = [C.c_void_p]
        for n in ("sp_set_psi", "sp_get_psi", "sp_get_psik"):
            getattr(l, n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        l.sp_get_phi.argtypes = [C.c_void_p, C.c_void_p]
        l.sp_fft.argtypes = [C.c_void_p, C.c_int]
        l.sp_potential.restype = C.c_double
        l.sp_potential.argtypes = [C.c_void_p]
        l.sp_update.restype = C.c_int
        l.sp_update.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        l.sp_threads.restype = C.c_int
        _lib = l
    return _lib


class CpuSim:
    def __init__(self, N, *, axis_length, hbar_, total_mass, cfl, final_sim_time, num_data_dumps, k2_cutoff,
                 alias_threshold, time=0.0, **_ignored):
        self.l = lib()
        self.N = N
        self.h = self.l.sp_create(N, axis_length, hbar_, total_mass, cfl, final_sim_time, num_data_dumps,
                                  k2_cutoff, alias_threshold, time)
        if not self.h:
            raise ValueError("sp_create rejected the size (power of two >= 8)")

    def close(self):
        if self.h:
            self.l.sp_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_psi(self, psi):
        re = np.ascontiguousarray(psi.real, dtype=np.float64)
        im = np.ascontiguousarray(psi.imag, dtype=np.float64)
        self.l.sp_set_psi(self.h, re.ctypes.data, im.ctypes.data)

    def _get(self, fn):
        re = np.empty((self.N,) * 3)
        im = np.empty((self.N,) * 3)
        fn(self.h, re.ctypes.data, im.ctypes.data)
        return re + 1j * im

    def get_psi(self):
        return self._get(self.l.sp_get_psi)

    def get_psik(self):
        return self._get(self.l.sp_get_psik)

    def get_phi(self):
        phi = np.empty((self.N,) * 3)
        self.l.sp_get_phi(self.h, phi.ctypes.data)
        return phi

    def fft(self, sign):
        self.l.sp_fft(self.h, sign)

    def potential(self):
        return self.l.sp_potential(self.h)

    def update(self):
        out = (C.c_double * 5)()
        aliased = self.l.sp_update(self.h, out)
        return dict(dt=out[0], dump=bool(out[1]), alias_mass=out[2], max_phi=out[3], time=out[4],
                    aliased=bool(aliased))

    @property
    def threads(self):
        return self.l.sp_threads()
