"""Host-side handle on one GPU's batch of streams (thin layer over the C ABI).

Names follow the reference's `SimulationObject` (simulator/src/simulation_object.rs):
`update` is one split-step update of every active stream, `calculate_potential`,
`get_timestep`-derived quantities come back in `status()`, `forward`/`inverse` are
the unitary FFT wrappers of simulator/src/utils/fft.rs.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib as L

HBAR = 1.757e-90  # common/src/constants.rs:5


class MsmError(RuntimeError):
    pass


def determine_pmass_hbar_(total_mass, ntot=None, particle_mass=None, hbar_=None):
    """common/src/parameters.rs:222-259: resolution order ntot -> particle_mass -> hbar_."""
    if ntot is not None:
        pm = total_mass / ntot
        hb = hbar_ if hbar_ is not None else HBAR / pm
    elif particle_mass is not None:
        pm = particle_mass
        hb = hbar_ if hbar_ is not None else HBAR / pm
    elif hbar_ is not None:
        hb = hbar_
        pm = HBAR / hb
    else:
        raise ValueError("You must specify the total mass and one of ntot, particle_mass or hbar_")
    return pm, hb


class Evolver:
    def __init__(self, *, size: int, n_streams: int, axis_length: float, final_sim_time: float, cfl: float,
                 num_data_dumps: int, total_mass: float, k2_cutoff: float, alias_threshold: float,
                 hbar_: Optional[float] = None, particle_mass: Optional[float] = None,
                 ntot: Optional[float] = None, time: float = 0.0, dims: int = 3, precision: str = "f64",
                 batch_streams: int = 0, cosmology: Optional[dict] = None, device: int = 0,
                 hip_stream: Optional[int] = None):
        self._lib = L.load()
        pm, hb = determine_pmass_hbar_(total_mass, ntot, particle_mass, hbar_)
        cfg = L.MsmConfig()
        cfg.struct_size = C.sizeof(L.MsmConfig)
        cfg.device, cfg.dims, cfg.size = device, dims, size
        cfg.precision = {"f64": L.MSM_F64, "f32": L.MSM_F32}[precision]
        cfg.n_streams, cfg.batch_streams = n_streams, batch_streams
        cfg.expanding = 1 if cosmology else 0
        cfg.axis_length, cfg.time, cfg.final_sim_time, cfg.cfl = axis_length, time, final_sim_time, cfl
        cfg.num_data_dumps = num_data_dumps
        cfg.total_mass, cfg.particle_mass, cfg.hbar_ = total_mass, pm, hb
        cfg.k2_cutoff, cfg.alias_threshold = k2_cutoff, alias_threshold
        if cosmology:
            cfg.omega_matter_now = cosmology["omega_matter_now"]
            cfg.omega_radiation_now = cosmology["omega_radiation_now"]
            cfg.h, cfg.z0 = cosmology["h"], cosmology["z0"]
            cfg.max_dloga = cosmology.get("max_dloga") or 0.0
        cfg.hip_stream = hip_stream
        self.cfg = cfg
        self.size, self.dims, self.n_streams = size, dims, n_streams
        self.real = np.float64 if precision == "f64" else np.float32
        self.cplx = np.complex128 if precision == "f64" else np.complex64
        self.shape = (size,) * dims
        self._ctx = C.c_void_p()
        rc = self._lib.msm_create(C.byref(cfg), C.byref(self._ctx))
        if rc != 0:
            raise MsmError(f"msm_create failed ({rc}): {self._lib.msm_last_error(None).decode()}")

    # -- plumbing ---------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise MsmError(f"{what} failed ({rc}): {self._lib.msm_last_error(self._ctx).decode()}")

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.msm_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def batch_streams(self) -> int:
        b = C.c_int32()
        self._check(self._lib.msm_get_batch(self._ctx, C.byref(b)), "msm_get_batch")
        return b.value

    def derived(self) -> dict:
        out = (C.c_double * 8)()
        self._check(self._lib.msm_get_derived(self._ctx, out), "msm_get_derived")
        keys = ["dx", "dk", "k2_max", "hbar_", "n_tot", "kinetic_dt", "comoving_boxsize", "transform_passes_mode"]
        return dict(zip(keys, list(out)))

    # -- initial conditions ----------------------------------------------
    def set_psi(self, stream: int, psi: np.ndarray):
        if psi.shape != self.shape:
            raise ValueError(f"psi has shape {psi.shape}, expected {self.shape}")
        re = np.ascontiguousarray(psi.real, dtype=self.real)
        im = np.ascontiguousarray(psi.imag, dtype=self.real)
        self._check(self._lib.msm_set_psi(self._ctx, stream, re.ctypes.data, im.ctypes.data), "msm_set_psi")

    def copy_psi(self, dst: int, src: int):
        self._check(self._lib.msm_copy_psi(self._ctx, dst, src), "msm_copy_psi")

    def sample(self, stream: int, scheme: str, seed: int):
        code = {"Poisson": L.MSM_POISSON, "Wigner": L.MSM_WIGNER, "Husimi": L.MSM_HUSIMI}[scheme]
        self._check(self._lib.msm_sample(self._ctx, stream, code, seed), "msm_sample")

    # -- operators --------------------------------------------------------
    def forward(self, stream: int):
        self._check(self._lib.msm_fft(self._ctx, stream, L.MSM_FORWARD), "msm_fft")

    def inverse(self, stream: int):
        self._check(self._lib.msm_fft(self._ctx, stream, L.MSM_INVERSE), "msm_fft")

    def calculate_potential(self, first: int = 0, count: Optional[int] = None):
        count = self.n_streams - first if count is None else count
        self._check(self._lib.msm_potential(self._ctx, first, count), "msm_potential")

    def update(self):
        self._check(self._lib.msm_step(self._ctx), "msm_step")

    def step_begin(self, a_now, tau_to_next_dump):
        a = np.ascontiguousarray(a_now, dtype=np.float64)
        t = np.ascontiguousarray(tau_to_next_dump, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self._check(self._lib.msm_step_begin(self._ctx, a.ctypes.data_as(dp), t.ctypes.data_as(dp)),
                    "msm_step_begin")

    def step_end(self, kick_a, new_time, new_tau, snapped_tau):
        arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (kick_a, new_time, new_tau, snapped_tau)]
        dp = C.POINTER(C.c_double)
        self._check(self._lib.msm_step_end(self._ctx, *[x.ctypes.data_as(dp) for x in arrs]), "msm_step_end")

    def sync(self):
        self._check(self._lib.msm_sync(self._ctx), "msm_sync")

    def status(self):
        buf = (L.MsmStreamStatus * self.n_streams)()
        self._check(self._lib.msm_get_status(self._ctx, buf, self.n_streams), "msm_get_status")
        return list(buf)

    # -- downloads ----------------------------------------------------------
    def _get_c(self, fn, name, stream):
        re = np.empty(self.shape, dtype=self.real)
        im = np.empty(self.shape, dtype=self.real)
        self._check(fn(self._ctx, stream, re.ctypes.data, im.ctypes.data), name)
        return (re + 1j * im).astype(self.cplx)

    def get_psi(self, stream: int) -> np.ndarray:
        return self._get_c(self._lib.msm_get_psi, "msm_get_psi", stream)

    def get_psik(self, stream: int) -> np.ndarray:
        return self._get_c(self._lib.msm_get_psik, "msm_get_psik", stream)

    def get_potential(self, stream: int) -> np.ndarray:
        phi = np.empty(self.shape, dtype=self.real)
        self._check(self._lib.msm_get_potential(self._ctx, stream, phi.ctypes.data), "msm_get_potential")
        return phi

    def density_sum(self, device_ptr: int):
        self._check(self._lib.msm_density_sum(self._ctx, device_ptr), "msm_density_sum")

    def step_density(self, device_ptr):
        self._check(self._lib.msm_step_density(self._ctx, device_ptr), "msm_step_density")

    # -- dump pipeline / restart ---------------------------------------------
    def pinned_planes(self):
        """Two pinned host planes (re, im) for get_psi_async; released with free_pinned."""
        out = []
        for _ in range(2):
            p = C.c_void_p()
            nbytes = int(np.prod(self.shape)) * np.dtype(self.real).itemsize
            self._check(self._lib.msm_host_alloc(nbytes, C.byref(p)), "msm_host_alloc")
            buf = (C.c_char * nbytes).from_address(p.value)
            arr = np.frombuffer(buf, dtype=self.real).reshape(self.shape)
            out.append((arr, p.value))
        return out

    def free_pinned(self, planes):
        for _, ptr in planes:
            self._lib.msm_host_free(ptr)

    def get_psi_async(self, stream: int, planes) -> int:
        t = C.c_int32()
        self._check(self._lib.msm_get_psi_async(self._ctx, stream, planes[0][1], planes[1][1], C.byref(t)), "msm_get_psi_async")
        return t.value

    def download_wait(self, ticket: int):
        self._check(self._lib.msm_download_wait(self._ctx, ticket), "msm_download_wait")

    def set_clock(self, stream: int, time: float, tau: float, current_dumps: int):
        self._check(self._lib.msm_set_clock(self._ctx, stream, time, tau, current_dumps), "msm_set_clock")

    # -- in-situ synthesizer (synthesizer/src/main.rs:63-93,161-187) -------
    SYNTH_FIELDS = ("psi", "psi2", "psik", "psik2")

    def synth_accumulate(self, stream: int, slot: int):
        self._check(self._lib.msm_synth_accumulate(self._ctx, stream, slot), "msm_synth_accumulate")

    def synth_device(self, slot: int):
        """(device pointer of the 4 * N^dims complex128 sums, streams accumulated)"""
        ptr, cnt = C.c_void_p(), C.c_int64()
        self._check(self._lib.msm_synth_device(self._ctx, slot, C.byref(ptr), C.byref(cnt)), "msm_synth_device")
        return ptr.value, cnt.value

    def synth_set_count(self, slot: int, count: int):
        self._check(self._lib.msm_synth_set_count(self._ctx, slot, count), "msm_synth_set_count")

    def synth_fetch(self, slot: int, field: str) -> np.ndarray:
        re = np.empty(self.shape, dtype=np.float64)
        im = np.empty(self.shape, dtype=np.float64)
        self._check(self._lib.msm_synth_fetch(self._ctx, slot, self.SYNTH_FIELDS.index(field), re.ctypes.data,
                                              im.ctypes.data), "msm_synth_fetch")
        return re + 1j * im

    def synth_qx(self, slot: int) -> complex:
        re, im = C.c_double(), C.c_double()
        self._check(self._lib.msm_synth_qx(self._ctx, slot, C.byref(re), C.byref(im)), "msm_synth_qx")
        return complex(re.value, im.value)

    def synth_qx_noalloc(self, slot: int, device_out2: int):
        """Qx into two doubles at a GPU-addressable address; asynchronous, allocates nothing."""
        self._check(self._lib.msm_synth_qx_noalloc(self._ctx, slot, device_out2), "msm_synth_qx_noalloc")

    def synth_release(self, slot: int):
        self._check(self._lib.msm_synth_release(self._ctx, slot), "msm_synth_release")

    # -- several GPUs (RCCL inside the library) -------------------------------
    COMM_SUM, COMM_MAX, COMM_MIN = 0, 2, 3

    @staticmethod
    def comm_unique_id() -> bytes:
        """128 bytes one rank creates and hands to every other rank (msm_comm_unique_id)."""
        lib = L.load()
        buf = C.create_string_buffer(128)
        rc = lib.msm_comm_unique_id(buf)
        if rc != 0:
            raise MsmError(f"msm_comm_unique_id failed ({rc}): {lib.msm_last_error(None).decode()}")
        return buf.raw

    @staticmethod
    def comm_origin() -> str:
        """Which shared object RCCL was bound from (or why none could be)."""
        return L.load().msm_comm_origin().decode()

    def comm_init(self, comm_id: bytes, rank: int, world: int):
        if len(comm_id) != 128:
            raise ValueError("the communicator id is 128 bytes")
        self._check(self._lib.msm_comm_init(self._ctx, comm_id, rank, world), "msm_comm_init")

    def comm_allreduce(self, values, op: int = 0):
        arr = (C.c_double * len(values))(*[float(v) for v in values])
        self._check(self._lib.msm_comm_allreduce_f64(self._ctx, arr, len(values), op), "msm_comm_allreduce_f64")
        return list(arr)

    def mean_density_begin(self, total_streams: int):
        self._check(self._lib.msm_mean_density_begin(self._ctx, total_streams), "msm_mean_density_begin")

    def mean_density(self, to_host: bool = True):
        """<|psi|^2> over all ranks' streams after the most recent step: numpy array, or the device pointer."""
        dev = C.c_void_p()
        if not to_host:
            self._check(self._lib.msm_mean_density(self._ctx, None, C.byref(dev)), "msm_mean_density")
            return dev.value
        out = np.empty(self.shape, dtype=np.float64)
        self._check(self._lib.msm_mean_density(self._ctx, out.ctypes.data, C.byref(dev)), "msm_mean_density")
        return out

    def synth_allreduce(self, slot: int):
        self._check(self._lib.msm_synth_allreduce(self._ctx, slot), "msm_synth_allreduce")

    def timer_start(self):
        self._check(self._lib.msm_timer_start(self._ctx), "msm_timer_start")

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        self._check(self._lib.msm_timer_stop_ms(self._ctx, C.byref(ms)), "msm_timer_stop_ms")
        return ms.value

    def profile(self, enable: bool):
        self._check(self._lib.msm_profile(self._ctx, 1 if enable else 0), "msm_profile")

    def profile_report(self) -> dict:
        """{pass kind: (total ms, launches)} since the last report."""
        ms = (C.c_double * 16)()
        cnt = (C.c_int64 * 16)()
        n = C.c_int32()
        self._check(self._lib.msm_profile_report(self._ctx, 16, ms, cnt, C.byref(n)), "msm_profile_report")
        return {self._lib.msm_profile_kind_name(k).decode(): (ms[k], cnt[k]) for k in range(n.value)}
