"""ctypes access to libmsm_host.so: the native host side of the reference's process surface (TOML schema of
common/src/parameters.rs, stream expansion of simulator/src/utils/io.rs:115-246, IC builders of simulator/src/ics.rs).
No GPU dependency."""
import ctypes as C
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libmsm_host.so")
        if not os.path.exists(path):
            raise OSError(f"{path} is missing: build it with `make -C msm_amd/csrc`")
        h = C.CDLL(path)
        h.msmh_last_error.restype = C.c_char_p
        h.msmh_describe_toml.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        h.msmh_build_ics.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        _lib = h
    return _lib


def read_toml(path, expanding=False) -> dict:
    """The parsed configuration plus the list of simulations it expands to (`streams`: N seeds, then the mean-field run)."""
    buf = C.create_string_buffer(1 << 22)
    if lib().msmh_describe_toml(str(path).encode(), int(expanding), buf, len(buf)) != 0:
        raise ValueError(lib().msmh_last_error().decode())
    return json.loads(buf.value.decode())


def build_ics(path, size, dims, dx, precision="f64", expanding=False) -> np.ndarray:
    """psi_0 named by the TOML's [ics] table, built by the native IC builders (for ColdGaussKSpace: the k-space field,
    which the device turns into psi_0 with one forward transform, ics.rs:425)."""
    real = np.float64 if precision == "f64" else np.float32
    re = np.zeros((size,) * dims, dtype=real)
    im = np.zeros((size,) * dims, dtype=real)
    if lib().msmh_build_ics(str(path).encode(), int(expanding), 0 if precision == "f64" else 1, float(dx),
                            re.ctypes.data, im.ctypes.data) != 0:
        raise ValueError(lib().msmh_last_error().decode())
    return re + 1j * im
