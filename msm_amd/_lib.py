"""ctypes binding of libmsm_hip.so (the C ABI declared in include/msm.h).

The library is the product; there is no fallback.  Importing this module
without the built shared object raises, so that nothing silently runs on a
CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSM_HIP_LIB selects another build of the same library (kernel-variant experiments)
LIB_PATH = os.environ.get("MSM_HIP_LIB") or os.path.join(_HERE, "libmsm_hip.so")

MSM_F64, MSM_F32 = 0, 1
MSM_POISSON, MSM_WIGNER, MSM_HUSIMI = 0, 1, 2
MSM_FORWARD, MSM_INVERSE = -1, 1


class MsmConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("dims", C.c_int32), ("size", C.c_int32),
        ("precision", C.c_int32), ("n_streams", C.c_int32), ("batch_streams", C.c_int32),
        ("expanding", C.c_int32),
        ("axis_length", C.c_double), ("time", C.c_double), ("final_sim_time", C.c_double),
        ("cfl", C.c_double),
        ("num_data_dumps", C.c_uint32), ("reserved0", C.c_uint32),
        ("total_mass", C.c_double), ("particle_mass", C.c_double), ("hbar_", C.c_double),
        ("k2_cutoff", C.c_double), ("alias_threshold", C.c_double),
        ("omega_matter_now", C.c_double), ("omega_radiation_now", C.c_double), ("h", C.c_double),
        ("z0", C.c_double), ("max_dloga", C.c_double),
        ("hip_stream", C.c_void_p),
    ]


class MsmStreamStatus(C.Structure):
    _fields_ = [
        ("time", C.c_double), ("dt", C.c_double), ("max_phi", C.c_double), ("alias_mass", C.c_double),
        ("tau", C.c_double), ("n_steps", C.c_int64), ("current_dumps", C.c_int32), ("dump", C.c_int32),
        ("active", C.c_int32), ("aliased", C.c_int32),
    ]


# every symbol include/msm.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "msm_create": (C.c_int32, [C.POINTER(MsmConfig), C.POINTER(_P)]),
    "msm_destroy": (C.c_int32, [_P]),
    "msm_last_error": (C.c_char_p, [_P]),
    "msm_get_derived": (C.c_int32, [_P, C.POINTER(C.c_double)]),
    "msm_set_psi": (C.c_int32, [_P, C.c_int32, _P, _P]),
    "msm_copy_psi": (C.c_int32, [_P, C.c_int32, C.c_int32]),
    "msm_sample": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_uint64]),
    "msm_fft": (C.c_int32, [_P, C.c_int32, C.c_int32]),
    "msm_potential": (C.c_int32, [_P, C.c_int32, C.c_int32]),
    "msm_get_batch": (C.c_int32, [_P, C.POINTER(C.c_int32)]),
    "msm_step": (C.c_int32, [_P]),
    "msm_step_begin": (C.c_int32, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "msm_step_end": (C.c_int32, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "msm_sync": (C.c_int32, [_P]),
    "msm_get_status": (C.c_int32, [_P, C.POINTER(MsmStreamStatus), C.c_int32]),
    "msm_get_psi": (C.c_int32, [_P, C.c_int32, _P, _P]),
    "msm_get_psik": (C.c_int32, [_P, C.c_int32, _P, _P]),
    "msm_get_potential": (C.c_int32, [_P, C.c_int32, _P]),
    "msm_density_sum": (C.c_int32, [_P, _P]),
    "msm_step_density": (C.c_int32, [_P, _P]),
    "msm_get_psi_async": (C.c_int32, [_P, C.c_int32, _P, _P, C.POINTER(C.c_int32)]),
    "msm_download_wait": (C.c_int32, [_P, C.c_int32]),
    "msm_host_alloc": (C.c_int32, [C.c_size_t, C.POINTER(_P)]),
    "msm_host_free": (C.c_int32, [_P]),
    "msm_set_clock": (C.c_int32, [_P, C.c_int32, C.c_double, C.c_double, C.c_int32]),
    "msm_synth_accumulate": (C.c_int32, [_P, C.c_int32, C.c_int32]),
    "msm_synth_device": (C.c_int32, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "msm_synth_set_count": (C.c_int32, [_P, C.c_int32, C.c_int64]),
    "msm_synth_fetch": (C.c_int32, [_P, C.c_int32, C.c_int32, _P, _P]),
    "msm_synth_qx": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "msm_synth_qx_noalloc": (C.c_int32, [_P, C.c_int32, _P]),
    "msm_synth_release": (C.c_int32, [_P, C.c_int32]),
    "msm_comm_unique_id": (C.c_int32, [_P]),
    "msm_comm_origin": (C.c_char_p, []),
    "msm_comm_init": (C.c_int32, [_P, _P, C.c_int32, C.c_int32]),
    "msm_comm_allreduce_f64": (C.c_int32, [_P, C.POINTER(C.c_double), C.c_int32, C.c_int32]),
    "msm_mean_density_begin": (C.c_int32, [_P, C.c_int64]),
    "msm_mean_density": (C.c_int32, [_P, _P, C.POINTER(_P)]),
    "msm_synth_allreduce": (C.c_int32, [_P, C.c_int32]),
    "msm_timer_start": (C.c_int32, [_P]),
    "msm_timer_stop_ms": (C.c_int32, [_P, C.POINTER(C.c_float)]),
    "msm_profile": (C.c_int32, [_P, C.c_int32]),
    "msm_profile_report": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int32)]),
    "msm_profile_kind_name": (C.c_char_p, [C.c_int32]),
}

_lib = None


def load() -> C.CDLL:
    """Load libmsm_hip.so and type every entry point; raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is not built; run `make -C msm_amd/csrc` (or __graft_entry__.build()). "
                "There is no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def kernel_source_hash() -> str:
    """sha256 over the device sources (msm_amd/csrc/*.hip, *.inc, *.hpp, sorted by name): ties a committed profile
    summary to the kernels it was taken from.  Empty when the sources are not present beside the library."""
    import glob
    import hashlib
    src = os.path.join(_HERE, "csrc")
    names = sorted(n for pat in ("*.hip", "*.inc", "*.hpp") for n in glob.glob(os.path.join(src, pat)))
    if not names:
        return ""
    h = hashlib.sha256()
    for n in names:
        h.update(os.path.basename(n).encode() + b"\0")
        with open(n, "rb") as f:
            h.update(f.read())
    return h.hexdigest()
