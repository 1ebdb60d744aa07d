"""CPU restatement (numpy) of the reference's split-step Schroedinger-Poisson path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (msm_amd/, the C-ABI
library, the CLI) may import or execute this file; it exists so that tests,
the smoke check and the benchmark's CPU-baseline leg can check the HIP path
against an independent statement of the reference algorithm.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  Array convention: a grid is a numpy array of shape
(N,)*dims in C order; element [a][b][c] is the reference's device flat index
a*N^2 + b*N + c (simulator/src/utils/io.rs:60-68 writes the device buffer
verbatim with shape (N,N,N,1)).  Every operator on the path is symmetric under
axis permutation, so only load/dump and ColdGauss care about that mapping.

Pinned against the reference's own known answers in tests/test_oracle.py
(k-grid, k^2 grid, FFT round trip/unitarity, normalize, seeds grammar, the
shipped planeWave3d npz).  Two pieces are "parity unpinned" because the
reference delegates them to third-party code that is not in its tree and has no
test touching it: the Philox/Box-Muller normal stream of ArrayFire 3.8.0
(simulator/src/ics.rs:492,563-575) and the scale-factor integrator of the
`cosmology` 0.2.0 crate (simulator/src/expanding.rs:81-116).
"""
from __future__ import annotations

import math
import os
import re
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

# common/src/constants.rs:2,5,9
POIS_CONST = 4.0 * math.pi * 4.49e-12
HBAR = 1.757e-90
LITTLE_H_TO_BIG_H = 1.022e-4
DEFAULT_MAX_DLOGA = 1e-3  # simulator/src/expanding.rs:27


# ----------------------------------------------------------------------------
# k-grid and FFT wrappers (simulator/src/utils/fft.rs)
# ----------------------------------------------------------------------------
def get_kgrid(dx, size: int, dtype=np.float64) -> np.ndarray:
    """utils/fft.rs:100-120: k_i = i/(N dx) for i < N/2 else (i-N)/(N dx)."""
    assert size % 2 == 0
    T = np.dtype(dtype).type
    dx = T(dx)
    out = np.empty(size, dtype=dtype)
    for i in range(size):
        ii = i if i < size // 2 else i - size
        out[i] = T(ii) / (T(size) * dx)
    return out


def spec_grid(dx, dims: int, size: int, dtype=np.float64) -> np.ndarray:
    """utils/fft.rs:123-161: ((kx^2 [+ ky^2] [+ kz^2]) * (2 pi)^2, summed in
    axis order 0,1,2 starting from a zero array, then scaled."""
    T = np.dtype(dtype).type
    k2 = get_kgrid(dx, size, dtype)
    k2 = (k2 * k2).astype(dtype)
    arr = np.zeros((size,) * dims, dtype=dtype)
    for ax in range(dims):
        shape = [1] * dims
        # reference axis i is the i-th FASTEST axis (ArrayFire dim i)
        shape[dims - 1 - ax] = size
        arr = arr + k2.reshape(shape)
    two_pi_sq = T(T(2.0 * math.pi) ** T(2.0))
    return (arr * two_pi_sq).astype(dtype)


def _norm_factor(dims: int, size: int) -> float:
    # utils/fft.rs:17,44,70,90
    return 1.0 / float(size) ** (dims / 2.0)


# Threads for the transforms of large grids (tests at 256^3 / 512^3 set it): > 1 switches from numpy's
# pocketfft to scipy's, the same algorithm with a `workers` argument.  0 keeps numpy.
FFT_WORKERS = 0


def _fftn(a: np.ndarray, backward: bool) -> np.ndarray:
    if FFT_WORKERS > 1:
        import scipy.fft as sf
        return sf.ifftn(a, norm="forward", workers=FFT_WORKERS) if backward else sf.fftn(a, workers=FFT_WORKERS)
    return np.fft.ifftn(a, norm="forward") if backward else np.fft.fftn(a)


def forward(a: np.ndarray) -> np.ndarray:
    """utils/fft.rs:6-31,60-78: unitary C2C FFT over all axes (scale applied
    after an unnormalised transform, forward sign e^{-2 pi i kn/N})."""
    dims, size = a.ndim, a.shape[0]
    out = _fftn(a, False) * a.real.dtype.type(_norm_factor(dims, size))
    return out.astype(a.dtype)


def inverse(a: np.ndarray) -> np.ndarray:
    """utils/fft.rs:33-58,80-98: unitary inverse (unnormalised backward
    transform times 1/N^{d/2})."""
    dims, size = a.ndim, a.shape[0]
    out = _fftn(a, True) * a.real.dtype.type(_norm_factor(dims, size))
    return out.astype(a.dtype)


# ----------------------------------------------------------------------------
# grid utils (simulator/src/utils/grid.rs)
# ----------------------------------------------------------------------------
def normalize(grid: np.ndarray, dx, dims: int) -> np.ndarray:
    """utils/grid.rs:11-33: grid *= sqrt(dx^-dims / sum |grid|^2)."""
    T = grid.real.dtype.type
    norm = T(np.sum((grid * np.conj(grid)).real))
    return (grid * T(np.sqrt(T(dx) ** T(-dims) / norm))).astype(grid.dtype)


def check_norm(grid: np.ndarray, dx, dims: int) -> bool:
    """utils/grid.rs:35-64."""
    T = grid.real.dtype.type
    s = np.sum((grid * np.conj(grid)).real)
    return abs(float(s * T(dx) ** T(dims)) - 1.0) < 1e-4


# ----------------------------------------------------------------------------
# config layer (common/src/parameters.rs, common/src/ics.rs)
# ----------------------------------------------------------------------------
def parse_seeds(s: str) -> List[int]:
    """common/src/parameters.rs:148-202: `a..=b`, `a to b`, or comma list
    (optionally bracketed)."""
    if re.search(r"\d+..=\d+", s):
        a, b = (int(x) for x in s.split("..="))
        return list(range(a, b + 1))
    if re.search(r"\d+ to \d+", s):
        a, b = (int(x) for x in s.split(" to "))
        return list(range(a, b + 1))
    found = re.findall(r"\d+", s)
    if found:
        return [int(x) for x in found]
    raise ValueError(
        "seeds did not match expected patterns: low..=high, low to high, [s1, s2, s3]"
    )


@dataclass
class CosmologyParameters:
    """common/src/parameters.rs:71-86."""
    omega_matter_now: float
    omega_radiation_now: float
    h: float
    z0: float
    max_dloga: Optional[float] = None


def determine_pmass_hbar_(toml: dict) -> Tuple[float, float]:
    """common/src/parameters.rs:222-259: ntot -> particle_mass -> hbar_."""
    if toml.get("ntot") is not None:
        pm = toml["total_mass"] / toml["ntot"]
        hb = toml["hbar_"] if toml.get("hbar_") is not None else HBAR / pm
    elif toml.get("particle_mass") is not None:
        pm = toml["particle_mass"]
        hb = toml["hbar_"] if toml.get("hbar_") is not None else HBAR / pm
    elif toml.get("hbar_") is not None:
        hb = toml["hbar_"]
        pm = HBAR / hb
    else:
        raise ValueError("You must specify the total mass and one of ntot, particle_mass, hbar_")
    return pm, hb


def get_supercomoving_boxsize(hbar_: float, cosmo: CosmologyParameters, axis_length: float) -> float:
    """common/src/parameters.rs:204-220."""
    a0 = 1.0 / (1.0 + cosmo.z0)
    comoving = axis_length / a0
    return math.sqrt(
        math.sqrt(1.5 * cosmo.omega_matter_now * (LITTLE_H_TO_BIG_H * cosmo.h) ** 2) / hbar_
    ) * comoving


# ----------------------------------------------------------------------------
# scale factor (simulator/src/expanding.rs + the un-vendored `cosmology` crate)
# ----------------------------------------------------------------------------
class ScaleFactorSolver:
    """Flat LCDM a(t), t in Myr, a(0) = 1/(1+z0).

    PARITY UNPINNED: the reference calls cosmology 0.2.0's
    `ScaleFactor::{new, step_forward, get_a, get_dadt, get_time}`
    (simulator/src/expanding.rs:81-116); that crate is not vendored and no
    reference test exercises it.  This restates the published physics
    (da/dt = a H0 sqrt(Om a^-3 + Or a^-4 + Ode)) with classical RK4 and
    sub-steps bounded by max_dloga; validated against the Einstein-de Sitter
    closed form in tests.
    """

    def __init__(self, cosmo: CosmologyParameters):
        assert cosmo.omega_matter_now + cosmo.omega_radiation_now <= 1.0
        assert cosmo.z0 >= 0.0
        self.om = cosmo.omega_matter_now
        self.orad = cosmo.omega_radiation_now
        self.ode = 1.0 - self.om - self.orad
        self.h0 = cosmo.h * LITTLE_H_TO_BIG_H
        self.max_dloga = cosmo.max_dloga if cosmo.max_dloga is not None else DEFAULT_MAX_DLOGA
        self.a = 1.0 / (1.0 + cosmo.z0)
        self.t = 0.0

    def clone(self) -> "ScaleFactorSolver":
        c = object.__new__(ScaleFactorSolver)
        c.__dict__.update(self.__dict__)
        return c

    def _dadt(self, a: float) -> float:
        return a * self.h0 * math.sqrt(self.om / a ** 3 + self.orad / a ** 4 + self.ode)

    def get_a(self) -> float:
        return self.a

    def get_dadt(self) -> float:
        return self._dadt(self.a)

    def get_time(self) -> float:
        return self.t

    def step(self, dt: float) -> float:
        if dt != 0.0:
            n = int(math.ceil(abs(dt) * self._dadt(self.a) / self.a / self.max_dloga))
            n = max(n, 1)
            h = dt / n
            a = self.a
            for _ in range(n):
                k1 = self._dadt(a)
                k2 = self._dadt(a + 0.5 * h * k1)
                k3 = self._dadt(a + 0.5 * h * k2)
                k4 = self._dadt(a + h * k3)
                a = a + h * (k1 + 2.0 * k2 + 2.0 * k3 + k4) / 6.0
            self.a = a
            self.t += dt
        return self.a


def rk4(f, tn: float, yn: float, h: float) -> float:
    """simulator/src/utils/mod.rs:14-43."""
    k1 = f(tn, yn)
    k2 = f(tn + h / 2.0, yn + h * k1 / 2.0)
    k3 = f(tn + h / 2.0, yn + h * k2 / 2.0)
    k4 = f(tn + h, yn + h * k3)
    return yn + h * (k1 + 2.0 * k2 + 2.0 * k3 + k4) / 6.0


def get_tau(target_time: float, cosmo: CosmologyParameters) -> float:
    """simulator/src/simulation_object.rs:1408-1453."""
    solver = ScaleFactorSolver(cosmo)
    pref = math.sqrt(1.5 * cosmo.omega_matter_now * (LITTLE_H_TO_BIG_H * cosmo.h) ** 2)

    def dtau_dt(t, _tau):
        a = solver.step(t - solver.get_time())
        return pref / a ** 2

    tau, time = 0.0, 0.0
    while time < target_time:
        dt = target_time / 1000.0
        if cosmo.max_dloga is not None:
            dt = min(dt, solver.get_a() / solver.get_dadt() * cosmo.max_dloga)
        dt = min(dt, target_time - time)
        tau = rk4(dtau_dt, time, tau, dt)
        time += dt
    return tau


# ----------------------------------------------------------------------------
# Philox4x32-10 counter RNG + Box-Muller (build's own normal stream)
# ----------------------------------------------------------------------------
_PH_M0, _PH_M1 = 0xD2511F53, 0xCD9E8D57
_PH_W0, _PH_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr: np.ndarray, key: Tuple[int, int]) -> np.ndarray:
    """Philox4x32-10 (Salmon et al., SC'11).  ctr: (n,4) uint32 -> (n,4) uint32."""
    c = ctr.astype(np.uint64)
    c0, c1, c2, c3 = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    k0, k1 = np.uint64(key[0] & 0xFFFFFFFF), np.uint64(key[1] & 0xFFFFFFFF)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(_PH_M0) * c0
        p1 = np.uint64(_PH_M1) * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & mask, lo1, (hi0 ^ c3 ^ k1) & mask, lo0
        k0 = (k0 + np.uint64(_PH_W0)) & mask
        k1 = (k1 + np.uint64(_PH_W1)) & mask
    return np.stack([c0, c1, c2, c3], axis=1).astype(np.uint32)


def complex_normals(n: int, seed: int) -> np.ndarray:
    """n complex samples, real and imaginary parts independent N(0,1).

    Element e uses Philox counter (e_lo, e_hi, 0, 0) under key (seed_lo,
    seed_hi); words (r0,r1) and (r2,r3) give two 53-bit uniforms in (0,1) and
    Box-Muller maps them to the (re, im) pair.  Always evaluated in float64.
    PARITY UNPINNED with respect to ArrayFire's random_normal stream.
    """
    e = np.arange(n, dtype=np.uint64)
    ctr = np.zeros((n, 4), dtype=np.uint32)
    ctr[:, 0] = (e & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (e >> np.uint64(32)).astype(np.uint32)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).astype(np.uint64)
    b1 = (r[:, 0] | (r[:, 1] << np.uint64(32))) >> np.uint64(11)
    b2 = (r[:, 2] | (r[:, 3] << np.uint64(32))) >> np.uint64(11)
    u1 = (b1.astype(np.float64) + 0.5) * 2.0 ** -53
    u2 = (b2.astype(np.float64) + 0.5) * 2.0 ** -53
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * math.pi * u2
    return rad * np.cos(ang) + 1j * rad * np.sin(ang)


def _uniform_pairs(e: np.ndarray, seed: int, j: int, lane: int):
    """Two 53-bit uniforms in (0,1) from the Philox block with counter (e_lo, e_hi, j, lane)."""
    ctr = np.zeros((e.size, 4), dtype=np.uint32)
    ctr[:, 0] = (e & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (e >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2] = np.uint32(j)
    ctr[:, 3] = np.uint32(lane)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).astype(np.uint64)
    b1 = (r[:, 0] | (r[:, 1] << np.uint64(32))) >> np.uint64(11)
    b2 = (r[:, 2] | (r[:, 3] << np.uint64(32))) >> np.uint64(11)
    return (b1.astype(np.float64) + 0.5) * 2.0 ** -53, (b2.astype(np.float64) + 0.5) * 2.0 ** -53


def uniforms(n: int, seed: int) -> np.ndarray:
    """n uniforms in (0,1): element e takes the first 53-bit uniform of Philox counter (e_lo, e_hi, 0, 2).
    PARITY UNPINNED with respect to ArrayFire's random_uniform stream."""
    return _uniform_pairs(np.arange(n, dtype=np.uint64), seed, 0, 2)[0]


def poisson_counts(lam: np.ndarray, seed: int) -> np.ndarray:
    """One Poisson(lam[e]) draw per element from the element's own Philox stream (counter
    (e_lo, e_hi, j, 1), j = 0, 1, ...): Knuth's product method for lam < 10, Hoermann's PTRS
    transformed rejection otherwise.  The reference draws from rand::thread_rng (ics.rs:497-520),
    which is not reproducible; only the distribution carries over."""
    lam = np.asarray(lam, dtype=np.float64).reshape(-1)
    n = lam.size
    e_all = np.arange(n, dtype=np.uint64)
    out = np.zeros(n, dtype=np.float64)
    small = lam < 10.0
    # Knuth
    idx = np.nonzero(small)[0]
    if idx.size:
        limit = np.exp(-lam[idx])
        p = np.ones(idx.size)
        k = np.zeros(idx.size)
        live = np.ones(idx.size, dtype=bool)
        j = 0
        while live.any():
            u1, u2 = _uniform_pairs(e_all[idx], seed, j, 1)
            for u in (u1, u2):
                p = np.where(live, p * u, p)
                acc = live & (p > limit)
                k = np.where(acc, k + 1, k)
                live = acc
            j += 1
        out[idx] = k
    # PTRS
    idx = np.nonzero(~small)[0]
    if idx.size:
        l = lam[idx]
        slam, loglam = np.sqrt(l), np.log(l)
        b = 0.931 + 2.53 * slam
        a = -0.059 + 0.02483 * b
        invalpha = 1.1239 + 1.1328 / (b - 3.4)
        vr = 0.9277 - 3.6224 / (b - 2.0)
        res = np.zeros(idx.size)
        live = np.ones(idx.size, dtype=bool)
        j = 0
        from scipy.special import gammaln
        while live.any():
            u1, u2 = _uniform_pairs(e_all[idx], seed, j, 1)
            U = u1 - 0.5
            V = u2
            us = 0.5 - np.abs(U)
            k = np.floor((2.0 * a / us + b) * U + l + 0.43)
            fast = (us >= 0.07) & (V <= vr)
            rejected = (k < 0) | ((us < 0.013) & (V > us))
            with np.errstate(invalid="ignore", divide="ignore"):
                slow = (np.log(V) + np.log(invalpha) - np.log(a / (us * us) + b)) <= (-l + k * loglam - gammaln(k + 1.0))
            acc = live & (fast | (~rejected & slow))
            res = np.where(acc, k, res)
            live = live & ~acc
            j += 1
        out[idx] = res
    return out


# ----------------------------------------------------------------------------
# initial conditions (simulator/src/ics.rs)
# ----------------------------------------------------------------------------
def spherical_tophat(axis_length, dx_param, dims, size, radius, delta, slope, dtype=np.float64):
    """ics.rs:165-280: sqrt(1 + delta / (1 + exp(slope (r/R - 1)))) at cell
    centres of the PROPER box, normalised with the parameter dx."""
    T = np.dtype(dtype).type
    L = T(axis_length)
    dx = L / T(size)
    x = (np.arange(size) * 2 + 1).astype(dtype) * dx / T(2.0)
    c = x - L / T(2.0)
    if dims == 3:
        r2 = c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2
    elif dims == 2:
        r2 = c[:, None] ** 2 + c[None, :] ** 2 + T(0.0)
    else:
        r2 = c ** 2
    r = np.sqrt(r2.astype(dtype))
    with np.errstate(over="ignore"):  # exp overflows to inf far outside the tophat; 1/inf = 0 as in the reference
        ramp = T(1.0) / (T(1.0) + np.exp(T(slope) * (r / T(radius) - T(1.0))))
    psi = np.sqrt(T(1.0) + T(delta) * ramp).astype(dtype)
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    return normalize(psi.astype(cdt), dx_param, dims)


def cold_gauss(mean, std, dx, dims, size, dtype=np.float64):
    """ics.rs:24-162: product of per-axis real Gaussians; each factor and the
    product are normalised with (dx, dims).  mean[0]/std[0] belong to the
    FASTEST axis (the last numpy axis)."""
    T = np.dtype(dtype).type
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    x = (np.arange(size) * 2 + 1).astype(dtype) * T(dx) / T(2.0)
    psi = None
    for ax in range(dims):
        g = np.exp(T(-0.5) * ((x - T(mean[ax])) / T(std[ax])) ** T(2.0)).astype(cdt)
        g = normalize(g, dx, dims)
        shape = [1] * dims
        shape[dims - 1 - ax] = size
        g = g.reshape(shape)
        psi = g if psi is None else (psi * g).astype(cdt)
    psi = np.ascontiguousarray(np.broadcast_to(psi, (size,) * dims)).astype(cdt)
    return normalize(psi, dx, dims)


def cold_gauss_kspace_k(mean, std, dx, dims, size, seed=None, dtype=np.float64):
    """ics.rs:282-424: separable Gaussians in k (get_kgrid values), each factor and the product
    normalised with (dk = dx, dims), times exp(2 pi i u) with u uniform from the Philox engine seeded
    with `seed` (default 0).  mean[0]/std[0] belong to the fastest axis."""
    T = np.dtype(dtype).type
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    k = get_kgrid(dx, size, dtype)
    psi = None
    for ax in range(dims):
        g = np.exp(T(-0.5) * ((k - T(mean[ax])) / T(std[ax])) ** T(2.0)).astype(cdt)
        g = normalize(g, dx, dims)
        shape = [1] * dims
        shape[dims - 1 - ax] = size
        g = g.reshape(shape)
        psi = g if psi is None else (psi * g).astype(cdt)
    psi = np.ascontiguousarray(np.broadcast_to(psi, (size,) * dims)).astype(cdt)
    psi = normalize(psi, dx, dims)
    u = uniforms(size ** dims, 0 if seed is None else int(seed)).reshape(psi.shape).astype(dtype)
    ph = T(2.0 * math.pi) * u
    return (psi.real * np.cos(ph) + 1j * (psi.real * np.sin(ph))).astype(cdt)


def cold_gauss_kspace(mean, std, dx, dims, size, seed=None, dtype=np.float64):
    """ics.rs:282-431: the k-space field above, then the FORWARD unitary transform (the reference's
    k -> x step is forward_inplace, :425)."""
    return forward(cold_gauss_kspace_k(mean, std, dx, dims, size, seed, dtype))


def user_specified_ics(path: str, dims: int, size: int, dtype=np.float64):
    """ics.rs:650-730: npz members real/imag, cubic, no renormalisation."""
    with np.load(path) as d:
        re_, im_ = d["real"], d["imag"]
    if re_.ndim != dims:
        raise ValueError(
            f"Dimensions of user-provided data ({re_.ndim}) do not match the dimensions specified in the toml ({dims})"
        )
    if any(s != re_.shape[0] for s in re_.shape):
        raise ValueError("Only uniform grids are supported at this time")
    if re_.shape[0] != size:
        raise ValueError(
            f"Grid size of user-provided data ({re_.shape[0]}) does not match the size specified in the toml ({size})"
        )
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    return (re_.astype(dtype) + 1j * im_.astype(dtype)).astype(cdt)


def sample_quantum_perturbation(psi, dx, dims, total_mass, particle_mass, scheme: str, seed: int):
    """ics.rs:434-648 (Wigner :560-602, Husimi :604-646).

    psi_count = psi sqrt(dV); noise = (N1 + i N2)/(2 sqrt n) [Wigner] or
    /(sqrt 2 sqrt n) [Husimi]; psi = (psi_count + noise)/sqrt(dV).
    """
    T = psi.real.dtype.type
    n = total_mass / particle_mass
    sqrt_n = T(math.sqrt(n))
    sqrt_dv = T(np.sqrt(T(dx) ** T(dims)))
    psi_count = psi * sqrt_dv
    z = complex_normals(psi.size, seed).reshape(psi.shape).astype(psi.dtype)
    if scheme == "Wigner":
        div = sqrt_n * T(2.0)
    elif scheme == "Husimi":
        div = sqrt_n * T(np.sqrt(T(2.0)))
    elif scheme == "Poisson":
        # ics.rs:495-558: count ~ Poisson(|psi|^2 dV n), amplitude sqrt(count / n), original phase
        lam = (np.abs(psi).astype(np.float64) ** 2 * float(T(dx)) ** dims) * n
        a = poisson_counts(lam, seed).reshape(psi.shape)
        amp = np.sqrt(a / n).astype(T)
        mag = np.abs(psi)
        unit = np.where(mag > 0, psi / np.where(mag > 0, mag, 1), 1.0).astype(psi.dtype)
        return ((amp * unit) / sqrt_dv).astype(psi.dtype)
    else:
        raise ValueError(scheme)
    samples = z / div
    return ((psi_count + samples) / sqrt_dv).astype(psi.dtype)


def plane_wave_ics(N: int, L=60.0, a_ini=0.01, A=(10.0, 10.0, 10.0), hbar_=0.01, H0=6.9e-5):
    """Restatement of sim.py:124-183 (`coords`, `calc_n`, `calc_phi`, `getPsi`)
    at grid size N.  Keeps sim.py's quirks: zq is built with A[1], Qz reuses
    Qy_, and the normalisation uses dx (not dx^3)."""
    D = a_ini
    dx = L / N
    q = np.linspace(-L / 2.0, L / 2.0, N)
    x = np.linspace(-L / 2.0, L / 2.0, N)
    ones = np.ones(N)
    xq = q - D * (L / np.pi / 2.0) * A[0] * np.sin(2 * q * np.pi / L)
    yq = q - D * (L / np.pi / 2.0) * A[1] * np.sin(2 * q * np.pi / L)
    Qx_ = np.interp(x, xq, q)
    Qy_ = np.interp(x, yq, q)
    Qx = np.einsum("i,j,k->ijk", Qx_, ones, ones)
    Qy = np.einsum("i,j,k->ijk", ones, Qy_, ones)
    Qz = np.einsum("i,j,k->ijk", ones, ones, Qy_)
    Q = [Qx, Qy, Qz]
    n = 1.0
    for i in range(3):
        n = n * (1.0 / (1.0 - D * A[i] * np.cos(2 * np.pi * Q[i] / L)))
    H = H0 / a_ini ** 3
    factor = a_ini ** 2 * D * 1.0 * H
    phi = 0.0
    for i in range(3):
        phi = phi + factor * (
            A[i] * L ** 2 / (2 * np.pi) ** 2 * np.cos(Q[i] * 2 * np.pi / L)
            + 0.5 * D * (A[i] * L / (2 * np.pi) * np.sin(Q[i] * 2 * np.pi / L)) ** 2
        )
    psi = np.sqrt(n) * np.exp(1j * phi / hbar_)
    mtot = np.sum(np.abs(psi) ** 2) * dx
    return psi / np.sqrt(mtot)


# ----------------------------------------------------------------------------
# the simulation object (simulator/src/simulation_object.rs)
# ----------------------------------------------------------------------------
@dataclass
class SimulationParameters:
    """simulation_object.rs:67-140 / `new` :223-315."""
    axis_length: float
    time: float
    final_sim_time: float
    cfl: float
    num_data_dumps: int
    total_mass: float
    particle_mass: float
    sim_name: str
    k2_cutoff: float
    alias_threshold: float
    hbar_: Optional[float]
    dims: int
    size: int
    output_potential: bool = False
    cosmo: Optional[CosmologyParameters] = None   # not None <=> feature "expanding"
    sampling: Optional[Tuple[int, str]] = None    # (seed, scheme)
    ics: Optional[dict] = None
    dtype: type = np.float64
    # derived
    dx: float = field(init=False)
    dk: float = field(init=False)
    k2_max: float = field(init=False)
    current_dumps: int = field(init=False, default=0)
    n_steps: int = field(init=False, default=0)

    def __post_init__(self):
        T = np.dtype(self.dtype).type
        self.T = T
        self.hbar_ = T(self.hbar_ if self.hbar_ is not None else HBAR / self.particle_mass)
        self.axis_length = T(self.axis_length)
        self.time = T(self.time)
        self.final_sim_time = T(self.final_sim_time)
        self.cfl = T(self.cfl)
        if self.cosmo is not None:
            self.tau = T(get_tau(float(self.time), self.cosmo))
            self.final_sim_tau = T(get_tau(float(self.final_sim_time), self.cosmo))
            self.comoving_boxsize = T(
                get_supercomoving_boxsize(float(self.hbar_), self.cosmo, float(self.axis_length))
            )
            self.dx = self.comoving_boxsize / T(self.size)
        else:
            self.dx = self.axis_length / T(self.size)
        self.dk = self.dx
        self.n_tot = self.total_mass / self.particle_mass
        self.spec_grid = spec_grid(self.dx, self.dims, self.size, self.dtype)
        self.k2_max = T(self.spec_grid.max())

    @property
    def expanding(self) -> bool:
        return self.cosmo is not None


class AliasError(RuntimeError):
    """The reference panics (simulation_object.rs:607-617)."""


class SimulationObject:
    """simulation_object.rs:145-184; `update` :475-661 (static) / :669-873
    (expanding)."""

    def __init__(self, params: SimulationParameters, psi: Optional[np.ndarray] = None):
        p = self.parameters = params
        cdt = np.complex128 if p.dtype == np.float64 else np.complex64
        if psi is None:
            psi = build_ics(p)
            if p.sampling is not None:
                seed, scheme = p.sampling
                psi = sample_quantum_perturbation(
                    psi, p.dx, p.dims, p.total_mass, p.particle_mass, scheme, seed
                )
        self.psi = psi.astype(cdt)
        self.psik = self.psi.copy()          # :206 "initialized with incorrect values"
        self.phi = self.psi.real.astype(p.dtype)
        self.dumps: List[Tuple[int, np.ndarray]] = []
        self.last = {}
        if p.expanding:
            self.scale_factor_solver = ScaleFactorSolver(p.cosmo)

    # -- :1031-1063
    def calculate_density(self) -> np.ndarray:
        p = self.parameters
        T = p.T
        if p.expanding:
            factor = T(
                p.total_mass * POIS_CONST
                * (2.0 / (3.0 * (p.cosmo.h * LITTLE_H_TO_BIG_H) ** 2 * p.cosmo.omega_matter_now)) ** 0.25
            ) / T(p.hbar_ ** T(p.dims / 2.0))
        else:
            factor = T(p.total_mass)
        return (factor * (self.psi * np.conj(self.psi)).real).astype(p.dtype)

    # -- :1066-1110
    def calculate_potential(self) -> np.ndarray:
        p = self.parameters
        T = p.T
        cdt = self.psi.dtype
        rho_k = forward(self.calculate_density().astype(cdt))
        c = T(-1.0) if p.expanding else T(-POIS_CONST)
        with np.errstate(divide="ignore", invalid="ignore"):
            phik = (c * rho_k) / p.spec_grid.astype(cdt)
        phik[np.isnan(phik)] = 0.0
        self.phi = inverse(phik.astype(cdt)).real.astype(p.dtype)
        return self.phi

    # -- :878-934 / :939-990
    def get_timestep(self) -> Tuple[bool, float]:
        p = self.parameters
        T = p.T
        potential_max = T(np.max(np.abs(self.phi)))
        with np.errstate(divide="ignore"):
            if p.expanding:
                kinetic = p.cfl * T(2.0) * p.comoving_boxsize / np.sqrt(p.k2_max)
                potential = p.cfl * T(2.0 * math.pi) / (T(2.0 * self.scale_factor_solver.get_a()) * potential_max)
            else:
                kinetic = p.cfl * T(2.0) * p.axis_length / np.sqrt(p.k2_max) / p.hbar_
                potential = p.cfl * T(2.0 * math.pi) * p.hbar_ / (T(2.0) * potential_max)
        ttnd = T(p.current_dumps + 1) * p.final_sim_time / T(p.num_data_dumps) - p.time
        if p.expanding:
            ttnd = T(get_tau(float(p.time + ttnd), p.cosmo)) - p.tau
        dt = min(min(kinetic, potential), ttnd)
        self.last.update(kinetic_dt=kinetic, potential_dt=potential, potential_max=potential_max)
        return bool(dt == ttnd), T(dt)

    # -- :1249-1293
    def check_alias(self) -> Tuple[float, bool]:
        p = self.parameters
        T = p.T
        a2 = (self.psik * np.conj(self.psik)).real.astype(p.dtype)
        over = p.spec_grid > p.k2_max * T(p.k2_cutoff)
        p_mass = T(np.sum(np.where(over, a2, T(0.0)))) * T(p.dk) ** T(p.dims)
        return p_mass, bool(p_mass > T(p.alias_threshold))

    def not_finished(self) -> bool:
        return bool(self.parameters.time < self.parameters.final_sim_time)

    def dump(self):
        self.dumps.append((self.parameters.current_dumps, self.psi.copy()))

    def _calculate_dt_from_dtau(self, dtau: float) -> float:
        """:1344-1388."""
        p = self.parameters
        solver = self.scale_factor_solver.clone()
        pref = math.sqrt(1.5 * p.cosmo.omega_matter_now * (LITTLE_H_TO_BIG_H * p.cosmo.h) ** 2)

        def dt_dtau(_tau, t):
            a = solver.step(t - solver.get_time())
            return 1.0 / (pref / a ** 2)

        return rk4(dt_dtau, float(p.tau), float(p.time), float(dtau)) - float(p.time)

    def update(self) -> dict:
        p = self.parameters
        T = p.T
        cdt = self.psi.dtype
        if p.n_steps == 0:
            self.psik = forward(self.psi)                                   # :477-479
        self.calculate_potential()                                          # :497
        dump, dt = self.get_timestep()                                      # :500
        if p.expanding:
            kphase = (T(-dt / T(4.0))) * p.spec_grid                        # :699-708
        else:
            kphase = (T(-dt / T(4.0) * p.hbar_)) * p.spec_grid              # :504-516
        k_evolution = np.exp(1j * kphase).astype(cdt)
        self.psik = (self.psik * k_evolution).astype(cdt)
        self.psi = inverse(self.psik)                                       # :523
        self.calculate_potential()                                          # :530
        if p.expanding:
            for _ in range(2):                                              # :726-760
                a = T(self.scale_factor_solver.get_a())
                r = np.exp(1j * (T(-dt / T(2.0) * a) * self.phi)).astype(cdt)
                self.psi = (self.psi * r).astype(cdt)
                dt_myr = T(self._calculate_dt_from_dtau(dt / T(2.0)))
                self.scale_factor_solver.step(float(dt_myr))
                p.time = T(p.time + dt_myr)
                p.tau = T(p.tau + dt / T(2.0))
        else:
            r = np.exp(1j * (T(-dt / p.hbar_) * self.phi)).astype(cdt)      # :535-545
            self.psi = (self.psi * r).astype(cdt)
        self.psik = forward(self.psi)                                       # :552
        self.psik = (self.psik * k_evolution).astype(cdt)                   # :562-574
        self.psi = inverse(self.psik)                                       # :581
        if not p.expanding:
            p.time = T(p.time + dt)                                         # :590
        p_mass, aliased = self.check_alias()                                # :607
        info = dict(dt=dt, dump=dump, alias_mass=p_mass, aliased=aliased,
                    max_phi=self.last["potential_max"])
        if aliased:
            raise AliasError(
                f"simulation aliased Fourier aliasing occurred; p_mass is {p_mass} and the threshold "
                f"was set at {p.alias_threshold} with k2_cutoff = {p.k2_cutoff}"
            )
        if dump:                                                            # :620-631
            p.current_dumps += 1
            self.dump()
            p.time = T(p.current_dumps) * p.final_sim_time / T(p.num_data_dumps)
            if p.expanding:
                p.tau = T(get_tau(float(p.time), p.cosmo))
        p.n_steps += 1
        info["time"] = p.time
        return info


def build_ics(p: SimulationParameters) -> np.ndarray:
    """simulation_object.rs:406-430 dispatch."""
    ics = p.ics
    kind = ics["type"]
    if kind == "UserSpecified":
        return user_specified_ics(ics["path"], p.dims, p.size, p.dtype)
    if kind == "ColdGauss":
        return cold_gauss(ics["mean"], ics["std"], p.dx, p.dims, p.size, p.dtype)
    if kind == "ColdGaussKSpace":
        return cold_gauss_kspace(ics["mean"], ics["std"], p.dx, p.dims, p.size, ics.get("phase_seed"), p.dtype)
    if kind == "SphericalTophat":
        return spherical_tophat(p.axis_length, p.dx, p.dims, p.size,
                                ics["radius"], ics["delta"], ics["slope"], p.dtype)
    raise NotImplementedError(kind)


def simulation_iter(toml: dict, expanding: bool = False, dtype=np.float64):
    """simulator/src/utils/io.rs:115-246: one parameter set per seed named
    `{sim}-stream{seed:05}`, then one mean-field set named `{sim}`."""
    pm, hb = determine_pmass_hbar_(toml)
    cosmo = None
    if expanding:
        c = toml["cosmology"]
        cosmo = CosmologyParameters(c["omega_matter_now"], c["omega_radiation_now"], c["h"],
                                    c["z0"], c.get("max_dloga"))
    samp = toml.get("sampling")
    seeds = parse_seeds(samp["seeds"]) if samp else []
    names = [(f"{toml['sim_name']}-stream{s:05d}", (s, samp["scheme"])) for s in seeds]
    names.append((toml["sim_name"], None))
    for name, sampling in names:
        yield SimulationParameters(
            axis_length=toml["axis_length"], time=toml.get("time", 0.0) or 0.0,
            final_sim_time=toml["final_sim_time"], cfl=toml["cfl"],
            num_data_dumps=toml["num_data_dumps"], total_mass=toml["total_mass"],
            particle_mass=pm, sim_name=name, k2_cutoff=toml["k2_cutoff"],
            alias_threshold=toml["alias_threshold"], hbar_=hb, dims=toml["dims"],
            size=toml["size"], output_potential=toml.get("output_potential", False),
            cosmo=cosmo, sampling=sampling, ics=toml["ics"], dtype=dtype,
        )


def write_dump(dirpath: str, k: int, psi: np.ndarray):
    """simulation_object.rs:1149-1164 + utils/io.rs:34-88: two NPY files named
    psi_{k:05}_real / _imag (no extension), shape (N,N,N,1) padded with ones."""
    os.makedirs(dirpath, exist_ok=True)
    shape = tuple(list(psi.shape) + [1] * (4 - psi.ndim))
    for part, arr in (("real", psi.real), ("imag", psi.imag)):
        with open(os.path.join(dirpath, f"psi_{k:05d}_{part}"), "wb") as f:
            np.save(f, np.ascontiguousarray(arr).reshape(shape))


# ---------------------------------------------------------------------------
# synthesizer (synthesizer/src/main.rs:63-93,161-187; lib.rs:190-300)
# ---------------------------------------------------------------------------
def reference_synthesis(dumps_by_stream, dv: float):
    """analyze_sims + the Qx post-combine for ONE dump index: dumps_by_stream = list of psi arrays
    (one per stream, the mean-field run excluded).  The transform is ndrustfft's ndfft along every
    axis, i.e. un-normalised."""
    n = len(dumps_by_stream)
    psi = sum(dumps_by_stream) / n
    psi2 = sum(np.abs(p) ** 2 for p in dumps_by_stream) / n
    ks = [np.fft.fftn(p) for p in dumps_by_stream]
    psik = sum(ks) / n
    psik2 = sum(np.abs(k) ** 2 for k in ks) / n
    qx = np.sum(psi2 - np.abs(psi) ** 2) * dv
    return dict(psi=psi, psi2=psi2.astype(np.complex128), psik=psik, psik2=psik2.astype(np.complex128), Qx=complex(qx))
