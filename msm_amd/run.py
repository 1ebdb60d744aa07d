"""Multi-GPU run of one reference TOML from Python: one process per GPU, streams block-partitioned, the in-situ
synthesizer reduced over the ranks.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        -m msm_amd.run --toml run.toml [--synthesize] [--expanding] [--precision f32] [--backend rccl|gloo]
    python -m msm_amd.run --toml run.toml                       (single GPU, no launcher)

Same process surface as `msm-simulator` (simulator/src/main.rs:9-79): TOML and initial conditions are read by the native
host library (libmsm_host.so), every rank writes the dumps of its own streams in the reference's layout, rank 0 also runs
the mean-field stream and writes the `-combined/` products.  The seed streams are independent (main.rs:43-79): per
step the ranks exchange one pair of flags, per dump index the synthesizer's sums.  With `--backend rccl` (default) both
go through the library's own RCCL communicator (msm_comm_*), the launcher's store only hands the communicator id from
rank 0 to the others -- the same path `msm-simulator --comm-file` and bench.py take; `--backend gloo` carries them over
torch.distributed on the CPU (rehearsals without GPUs, or several ranks sharing one GPU).  `--expanding` stands for the
reference's cargo feature: the scale-factor bookkeeping between msm_step_begin and msm_step_end is the native host
loop (msm_amd.expanding).
"""
import argparse
import datetime
import os
import sys

import numpy as np

from . import comm as mc
from . import host as mh
from .sharding import partition_streams
from .synth import InSituSynthesizer


def write_dump(root, sim_name, k, psi):
    """complex_array_to_disk (utils/io.rs:34-88): psi_{k:05}_real/_imag, NPY, 4-D."""
    d = os.path.join(root, "sim-data", sim_name)
    os.makedirs(d, exist_ok=True)
    a4 = psi.reshape(psi.shape + (1,) * (4 - psi.ndim))
    for suffix, part in (("_real", a4.real), ("_imag", a4.imag)):
        with open(os.path.join(d, f"psi_{k:05d}{suffix}"), "wb") as f:
            np.save(f, np.ascontiguousarray(part))


def store_exchange(rank, world):
    """Hands rank 0's communicator id to the others through the launcher's key-value store (TCPStore on
    MASTER_ADDR:MASTER_PORT; under torchrun the agent hosts it)."""
    from torch.distributed import TCPStore
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
    store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]), world,
                     is_master=(rank == 0 and not agent_store), timeout=datetime.timedelta(seconds=300), wait_for_workers=False)
    key = "msm_run_comm_id/" + os.environ.get("TORCHELASTIC_RUN_ID", "self") + "/" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")

    def exchange(cid):
        if cid is not None:
            store.set(key, cid)
        return bytes(store.get(key))
    return exchange


def main(argv=None):
    ap = argparse.ArgumentParser(prog="msm_amd.run")
    ap.add_argument("--toml", "-t", required=True)
    ap.add_argument("--synthesize", action="store_true")
    ap.add_argument("--expanding", action="store_true")
    ap.add_argument("--precision", choices=("f64", "f32"), default="f64")
    ap.add_argument("--backend", choices=("rccl", "nccl", "gloo"), default="rccl")
    ap.add_argument("--out", default=".")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on device 0 (one-GPU boxes; needs gloo or a stand-in for RCCL)")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 and args.backend == "gloo":
        import torch.distributed as dist
        dist.init_process_group("gloo")
    elif world > 1:
        import torch                    # noqa: F401  before the Evolver: libmsm_hip.so binds to the HIP runtime torch loaded
    from .evolver import Evolver

    t = mh.read_toml(args.toml, args.expanding)
    specs = t["streams"]
    seeds, mft = specs[:-1], specs[-1]
    first, count = partition_streams(len(seeds), world, rank)
    mine = seeds[first:first + count] + ([mft] if rank == 0 else [])
    n_local = len(mine)
    N, dims = t["size"], t["dims"]
    nd = t["num_data_dumps"]
    if world > 1 and args.backend != "gloo" and count == 0:
        raise SystemExit("more ranks than seed streams: every rank of the communicator needs a device context")

    ev = xhost = None
    if n_local:
        device = 0 if (args.share_gpu or world == 1) else local_rank
        ev = Evolver(size=N, n_streams=n_local, dims=dims, precision=args.precision, batch_streams=args.batch,
                     device=device, axis_length=t["axis_length"], final_sim_time=t["final_sim_time"],
                     cfl=t["cfl"], num_data_dumps=nd, total_mass=t["total_mass"], k2_cutoff=t["k2_cutoff"],
                     alias_threshold=t["alias_threshold"], particle_mass=t["resolved_particle_mass"],
                     hbar_=t["resolved_hbar_"], time=t["time"] or 0.0, cosmology=t["cosmology"] if args.expanding else None)
        ev.set_psi(0, mh.build_ics(args.toml, N, dims, ev.derived()["dx"], args.precision, args.expanding))
        if t["ics"]["type"] == "ColdGaussKSpace":
            ev.forward(0)               # k space -> real space is the reference's forward_inplace (ics.rs:425)
        for s in range(1, n_local):
            ev.copy_psi(s, 0)
        for s, sp in enumerate(mine):
            if sp["sampled"]:
                ev.sample(s, t["sampling"]["scheme"], sp["seed"])
        if args.expanding:
            from .expanding import ExpandingHost
            xhost = ExpandingHost(t["cosmology"], n_local, args.precision, t["final_sim_time"], nd, t["time"] or 0.0)

    if world == 1:
        comm = mc.NoComm()
    elif dist is not None:
        comm = mc.GlooComm(dist)
    else:
        comm = mc.LibraryComm(ev, rank, world, store_exchange(rank, world))

    n_seed_local = count
    syn = None
    if args.synthesize and seeds:
        syn = InSituSynthesizer(ev, t["sim_name"], len(seeds), nd, out_root=args.out, comm=comm, n_cells=N ** dims)
    finished_dumps = set()

    def on_dump(s, k):
        write_dump(args.out, mine[s]["sim_name"], k, ev.get_psi(s))
        if syn is not None and mine[s]["sampled"]:
            syn.add(s, k)

    def flush_synth():
        # every rank finalises dump k once all of ITS seed streams have reached it; ranks agree on the order
        if syn is None:
            return
        for k in range(nd + 1):
            if k in finished_dumps:
                continue
            mine_ready = (n_seed_local == 0) or syn.local_counts.get(k, 0) == n_seed_local
            if comm.allreduce([1.0 if mine_ready else 0.0], mc.MIN)[0] < 0.5:
                break
            syn.finish(k)               # a rank without seed streams contributes zeros
            finished_dumps.add(k)

    for s in range(n_local):
        on_dump(s, 0)                   # main.rs:61
    flush_synth()
    seen = [0] * n_local
    alias_msg = None
    abort = False
    st = ev.status() if ev is not None else []
    while True:
        # One collective per step carries both flags: "some stream is still running" and "some stream aliased".
        # Every rank therefore leaves the loop in the same iteration; a rank that raised on its own would leave
        # the others blocked in the next all-reduce.
        active = any(r.active for r in st)
        go, ab = comm.allreduce([1.0 if active else 0.0, 1.0 if alias_msg else 0.0], mc.MAX)
        abort = ab > 0.5
        if abort or go < 0.5:
            break
        if active:
            if xhost is not None:
                st = xhost.update(ev, st)
            else:
                ev.update()
                st = ev.status()
            for s in range(n_local):
                if st[s].aliased and alias_msg is None:
                    # the reference panics here (simulation_object.rs:607-617)
                    alias_msg = (f"simulation aliased Fourier aliasing occurred; p_mass is {st[s].alias_mass} and the "
                                 f"threshold was set at {t['alias_threshold']} with k2_cutoff = {t['k2_cutoff']} "
                                 f"({mine[s]['sim_name']})")
                if st[s].dump and st[s].current_dumps > seen[s] and not st[s].aliased:
                    seen[s] = st[s].current_dumps
                    on_dump(s, seen[s])
        flush_synth()       # entered by every rank, the aliased one included: its collectives stay matched
    if syn is not None and not abort:
        syn.close()
    comm.barrier()
    if ev is not None:
        ev.close()
    if dist is not None:
        dist.destroy_process_group()
    if abort:
        # every rank ends with the reference's non-zero panic status; the one that saw the aliasing says where
        print(f"thread 'main' panicked at '{alias_msg}'" if alias_msg else
              f"rank {rank}: stopping, a stream on another rank aliased", file=sys.stderr)
        return 101
    return 0


if __name__ == "__main__":
    sys.exit(main())
