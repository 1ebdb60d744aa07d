"""Plane-wave (Zel'dovich) initial conditions of the reference's Python pre-processor.

Restates `coords`, `calc_n`, `calc_phi` and `getPsi` of sim.py:124-183 for an
arbitrary grid size N (sim.py hard-codes N = 16 and writes
initial_conditions/planeWave3d_e10_sym.npz, which this reproduces at N = 16).
sim.py's own quirks are kept because they define the shipped fixture: the z
displacement uses A[1], Qz reuses the y map, and the norm uses dx rather than
dx^3 (sim.py:140,152,176).
"""
import numpy as np


def plane_wave_psi(N: int, L: float = 60.0, a_ini: float = 0.01, A=(10.0, 10.0, 10.0),
                   hbar_: float = 0.01, H0: float = 6.9e-5) -> np.ndarray:
    D = a_ini
    dx = L / N
    q = np.linspace(-L / 2.0, L / 2.0, N)
    x = np.linspace(-L / 2.0, L / 2.0, N)
    # Lagrangian -> Eulerian map per axis, inverted by interpolation (sim.py:133-147)
    disp = D * (L / np.pi / 2.0) * np.sin(2 * q * np.pi / L)
    Qx_ = np.interp(x, q - disp * A[0], q)
    Qy_ = np.interp(x, q - disp * A[1], q)
    Q = [Qx_[:, None, None], Qy_[None, :, None], Qy_[None, None, :]]
    # density (sim.py:156-163) and velocity potential (sim.py:165-178)
    dens = np.ones((N, N, N))
    for i in range(3):
        dens = dens * (1.0 / (1.0 - D * A[i] * np.cos(2 * np.pi * Q[i] / L)))
    H = H0 / a_ini ** 3
    factor = a_ini ** 2 * D * 1.0 * H
    phase = np.zeros((N, N, N))
    for i in range(3):
        phase = phase + factor * (A[i] * L ** 2 / (2 * np.pi) ** 2 * np.cos(Q[i] * 2 * np.pi / L)
                                  + .5 * D * (A[i] * L / (2 * np.pi) * np.sin(Q[i] * 2 * np.pi / L)) ** 2)
    psi = np.sqrt(dens) * np.exp(1j * phase / hbar_)
    mtot = np.sum(np.abs(psi) ** 2) * dx
    return psi / np.sqrt(mtot)
