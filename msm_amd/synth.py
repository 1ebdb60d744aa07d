"""In-situ synthesizer: the reference's post-processor (synthesizer/src/main.rs:63-187,
lib.rs:105-342,351-609) run while the streams are still on the device.

The reference re-reads every stream's every dump from disk, transforms it with an
un-normalised FFT and averages psi, |psi|^2, psik, |psik|^2 over streams per dump, then
derives Qx = sum(<|psi|^2> - |<psi>|^2) dv per dump.  Here a stream adds itself to its dump's
running sums (msm_synth_accumulate) the moment it reaches the dump; when every stream of every
rank has done so the sums are all-reduced (msm_amd.comm: RCCL inside the library, or gloo in CPU
rehearsals), averaged and written in the reference's layout:

    sim-data/{sim_name}-combined/{psi,psi2,psik,psik2}_{dump:05}_{real,imag}   NPY (N,N,N,1) f64
    sim-data/{sim_name}-combined/Qx_{real,imag}                                NPY (n_dumps+1,1,1,1)

The mean-field run is not a stream and is not included (the reference globs "-stream*",
lib.rs:190-193).
"""
import os
from typing import Dict

import numpy as np

FIELDS = ("psi", "psi2", "psik", "psik2")


def write_complex(path_base: str, arr: np.ndarray):
    """dump_complex (synthesizer/src/lib.rs:72-102): two NPY files without extension, 4-D."""
    os.makedirs(os.path.dirname(path_base), exist_ok=True)
    a4 = arr.reshape(arr.shape + (1,) * (4 - arr.ndim))
    for suffix, part in (("_real", a4.real), ("_imag", a4.imag)):
        with open(path_base + suffix, "wb") as f:
            np.save(f, np.ascontiguousarray(part, dtype=np.float64))


class InSituSynthesizer:
    def __init__(self, ev, sim_name: str, n_streams_global: int, n_dumps: int, out_root: str = ".",
                 comm=None, write_rank: int = 0, n_cells=None):
        """ev: msm_amd.Evolver (or an object with the same synth_* methods).
        n_streams_global: streams over all ranks; a dump is finalised when that many have arrived.
        comm: a msm_amd.comm carrier for multi-rank runs (LibraryComm on GPUs), None for one rank.
        n_cells: cells per grid, needed only by a rank that holds no streams at all (ev is None; gloo rehearsals)."""
        from .comm import NoComm
        self.ev, self.comm = ev, comm if comm is not None else NoComm()
        self.base = os.path.join(out_root, "sim-data", f"{sim_name}-combined")
        self.n_global, self.n_dumps = n_streams_global, n_dumps
        self.rank = self.comm.rank
        self.write_rank = write_rank
        self.local_counts: Dict[int, int] = {}
        self.qx: Dict[int, complex] = {}
        self.done = set()
        self.n_cells = n_cells

    def add(self, stream: int, dump: int):
        """Stream `stream` (local index) has just reached dump `dump`."""
        self.ev.synth_accumulate(stream, dump)
        self.local_counts[dump] = self.local_counts.get(dump, 0) + 1

    def ready(self, dump: int, n_local_streams: int) -> bool:
        return self.local_counts.get(dump, 0) == n_local_streams and dump not in self.done

    def finish(self, dump: int):
        """All local streams have contributed to `dump`: reduce over ranks, average, write, free.
        Collective: every rank must call it for the same dump in the same order (a rank without seed streams too:
        it contributes zeros)."""
        ev = self.ev
        count = self.comm.synth_allreduce(ev, dump, self.n_cells)
        if count != self.n_global:
            raise RuntimeError(f"dump {dump}: {count} streams accumulated, expected {self.n_global}")
        if self.rank == self.write_rank and ev is not None:
            for f in FIELDS:
                write_complex(os.path.join(self.base, f"{f}_{dump:05d}"), ev.synth_fetch(dump, f))
            self.qx[dump] = ev.synth_qx(dump)
        try:
            ev.synth_release(dump)
        except Exception:
            pass                                            # a rank that never opened the slot (or holds no context)
        self.done.add(dump)
        self.local_counts.pop(dump, None)

    def close(self):
        """Writes the Qx series (post_combine, lib.rs:351-609), ordered by dump."""
        if self.rank == self.write_rank and self.qx:
            series = np.array([self.qx[k] for k in sorted(self.qx)], dtype=np.complex128)
            write_complex(os.path.join(self.base, "Qx"), series.reshape(-1, 1, 1, 1))
