"""Host side of the expanding-box update for a whole device context (ctypes over libmsm_host.so).

The reference advances its scale-factor solver inside `update` between the two half-kicks
(simulator/src/simulation_object.rs:726-760) and sizes the step against the next dump in supercomoving time
(:939-990).  The device library leaves that bookkeeping to its caller (msm_step_begin / msm_step_end);
`msm-simulator` does it with host/expanding_loop.hpp, and this class hands the same native code to Python callers
(the benchmark, tests): no Python arithmetic takes part in the step.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_host = None


def _lib():
    global _host
    if _host is None:
        path = os.path.join(_HERE, "libmsm_host.so")
        if not os.path.exists(path):
            raise OSError(f"{path} is missing: build it with `make -C msm_amd/csrc`")
        lib = C.CDLL(path)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        lib.msmh_last_error.restype = C.c_char_p
        lib.msmh_expanding_new.restype = C.c_void_p
        lib.msmh_expanding_new.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint]
        lib.msmh_expanding_free.argtypes = [C.c_void_p]
        lib.msmh_expanding_begin.argtypes = [C.c_void_p, ip, dp, dp, dp]
        lib.msmh_expanding_advance.argtypes = [C.c_void_p, dp, ip, dp, dp, dp]
        lib.msmh_expanding_commit.argtypes = [C.c_void_p, dp, dp]
        _host = lib
    return _host


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class ExpandingHost:
    """Scale-factor solvers, clocks and the tau(t) cache of `n_streams` streams in precision `precision`."""

    def __init__(self, cosmology: dict, n_streams: int, precision: str, final_sim_time: float, num_data_dumps: int,
                 time: float = 0.0):
        self._lib = _lib()
        self.n = n_streams
        md = cosmology.get("max_dloga")
        c = (C.c_double * 5)(cosmology["omega_matter_now"], cosmology["omega_radiation_now"], cosmology["h"],
                             cosmology["z0"], md or 0.0)
        self._h = self._lib.msmh_expanding_new(c, 1 if md else 0, 0 if precision == "f64" else 1, n_streams, time,
                                               final_sim_time, num_data_dumps)
        if not self._h:
            raise ValueError(self._lib.msmh_last_error().decode())
        f = lambda: np.zeros(n_streams, dtype=np.float64)   # noqa: E731
        self.a_now, self.tau_next, self.snapped = f(), f(), f()
        self.kick_a, self.new_time, self.new_tau = f(), f(), f()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.msmh_expanding_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def begin(self, current_dumps):
        d = np.ascontiguousarray(current_dumps, dtype=np.intc)
        if self._lib.msmh_expanding_begin(self._h, _ip(d), _dp(self.a_now), _dp(self.tau_next), _dp(self.snapped)) != 0:
            raise ValueError(self._lib.msmh_last_error().decode())
        return self.a_now, self.tau_next, self.snapped

    def advance(self, dtau, active):
        dt = np.ascontiguousarray(dtau, dtype=np.float64)
        act = np.ascontiguousarray(active, dtype=np.intc)
        if self._lib.msmh_expanding_advance(self._h, _dp(dt), _ip(act), _dp(self.kick_a), _dp(self.new_time),
                                            _dp(self.new_tau)) != 0:
            raise ValueError(self._lib.msmh_last_error().decode())
        return self.kick_a, self.new_time, self.new_tau

    def commit(self, time, tau):
        t = np.ascontiguousarray(time, dtype=np.float64)
        ta = np.ascontiguousarray(tau, dtype=np.float64)
        self._lib.msmh_expanding_commit(self._h, _dp(t), _dp(ta))

    def update(self, ev, prev_status=None):
        """One expanding update (simulation_object.rs:669-873) of every stream of `ev`; returns the new status list."""
        prev = prev_status if prev_status is not None else ev.status()
        self.begin([r.current_dumps for r in prev])
        ev.step_begin(self.a_now, self.tau_next)
        mid = ev.status()
        self.advance([r.dt for r in mid], [r.active for r in prev])
        ev.step_end(self.kick_a, self.new_time, self.new_tau, self.snapped)
        st = ev.status()
        self.commit([r.time for r in st], [r.tau for r in st])
        return st
