"""msm_amd: MI355X-native split-step Schroedinger-Poisson evolver (host-side Python binding).

All compute lives in libmsm_hip.so (HIP, gfx950) behind the C ABI of include/msm.h.
"""
from ._lib import LIB_PATH, SYMBOLS, load  # noqa: F401
from .evolver import Evolver, MsmError, determine_pmass_hbar_  # noqa: F401
