"""Multi-GPU run of one reference TOML: one process per GPU, streams block-partitioned, the
in-situ synthesizer reduced over the ranks.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        -m msm_amd.run --toml run.toml [--synthesize] [--precision f32] [--backend nccl|gloo]
    python -m msm_amd.run --toml run.toml                       (single GPU, no process group)

Same process surface as `msm-simulator` (simulator/src/main.rs:9-79) for static boxes: TOML and
initial conditions are read by the native host library (libmsm_host.so), every rank writes the
dumps of its own streams in the reference's layout, rank 0 also runs the mean-field stream and
writes the `-combined/` products.  The seed streams are independent (main.rs:43-79): the only
exchange is the synthesizer's all-reduce per dump index.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

from .sharding import partition_streams
from .synth import InSituSynthesizer

_HERE = os.path.dirname(os.path.abspath(__file__))


def _host():
    lib = C.CDLL(os.path.join(_HERE, "libmsm_host.so"))
    lib.msmh_last_error.restype = C.c_char_p
    lib.msmh_build_ics.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    return lib


def read_toml(host, path):
    buf = C.create_string_buffer(1 << 20)
    if host.msmh_describe_toml(str(path).encode(), 0, buf, len(buf)) != 0:
        raise ValueError(host.msmh_last_error().decode())
    return json.loads(buf.value.decode())


def write_dump(root, sim_name, k, psi):
    """complex_array_to_disk (utils/io.rs:34-88): psi_{k:05}_real/_imag, NPY, 4-D."""
    d = os.path.join(root, "sim-data", sim_name)
    os.makedirs(d, exist_ok=True)
    a4 = psi.reshape(psi.shape + (1,) * (4 - psi.ndim))
    for suffix, part in (("_real", a4.real), ("_imag", a4.imag)):
        with open(os.path.join(d, f"psi_{k:05d}{suffix}"), "wb") as f:
            np.save(f, np.ascontiguousarray(part))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="msm_amd.run")
    ap.add_argument("--toml", "-t", required=True)
    ap.add_argument("--synthesize", action="store_true")
    ap.add_argument("--precision", choices=("f64", "f32"), default="f64")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl")
    ap.add_argument("--out", default=".")
    ap.add_argument("--batch", type=int, default=0)
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch                    # before the Evolver: libmsm_hip.so binds to the HIP runtime torch loaded
        import torch.distributed as dist
        if args.backend == "nccl":      # gloo runs (CPU rehearsals of the driver) select no device
            ndev = max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank % ndev)
        dist.init_process_group(args.backend)
    from .evolver import Evolver

    host = _host()
    t = read_toml(host, args.toml)
    # a [cosmology] table is ignored, as by the reference built without its "expanding" feature; expanding boxes run
    # through msm-simulator --expanding
    specs = t["streams"]
    seeds, mft = specs[:-1], specs[-1]
    first, count = partition_streams(len(seeds), world, rank)
    mine = seeds[first:first + count] + ([mft] if rank == 0 else [])
    n_local = len(mine)
    real = np.float64 if args.precision == "f64" else np.float32
    N, dims = t["size"], t["dims"]
    nd = t["num_data_dumps"]

    ev = None
    if n_local:
        ndev = 1
        if world > 1:
            import torch
            ndev = max(torch.cuda.device_count(), 1)
        ev = Evolver(size=N, n_streams=n_local, dims=dims, precision=args.precision, batch_streams=args.batch,
                     device=local_rank % ndev, axis_length=t["axis_length"], final_sim_time=t["final_sim_time"],
                     cfl=t["cfl"], num_data_dumps=nd, total_mass=t["total_mass"], k2_cutoff=t["k2_cutoff"],
                     alias_threshold=t["alias_threshold"], particle_mass=t["resolved_particle_mass"],
                     hbar_=t["resolved_hbar_"], time=t["time"] or 0.0)
        re = np.zeros((N,) * dims, dtype=real)
        im = np.zeros((N,) * dims, dtype=real)
        if host.msmh_build_ics(str(args.toml).encode(), 0, 0 if real == np.float64 else 1, ev.derived()["dx"],
                               re.ctypes.data, im.ctypes.data) != 0:
            raise SystemExit(host.msmh_last_error().decode())
        ev.set_psi(0, re + 1j * im)
        if t["ics"]["type"] == "ColdGaussKSpace":
            ev.forward(0)               # k space -> real space is the reference's forward_inplace (ics.rs:425)
        for s in range(1, n_local):
            ev.copy_psi(s, 0)
        for s, sp in enumerate(mine):
            if sp["sampled"]:
                ev.sample(s, t["sampling"]["scheme"], sp["seed"])

    n_seed_local = count
    syn = None
    if args.synthesize and seeds:
        syn = InSituSynthesizer(ev, t["sim_name"], len(seeds), nd, out_root=args.out, dist=dist)
    finished_dumps = set()

    def coll_device():
        import torch
        return "cuda" if torch.cuda.is_available() else "cpu"

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
            ready = mine_ready
            if dist is not None:
                import torch
                flag = torch.tensor([1 if mine_ready else 0], dtype=torch.int32, device=coll_device())
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ready = bool(flag.item())
            if not ready:
                break
            if n_seed_local == 0:
                ev_slotless(k)
            else:
                syn.finish(k)
            finished_dumps.add(k)

    def ev_slotless(k):
        # no local seed streams: contribute zeros to the reduction
        import torch
        n_doubles = 2 * 4 * int(N ** dims)
        z = torch.zeros(n_doubles, dtype=torch.float64, device=coll_device())
        dist.all_reduce(z)
        c = torch.zeros(1, dtype=torch.int64, device=coll_device())
        dist.all_reduce(c)

    for s in range(n_local):
        on_dump(s, 0)                   # main.rs:61
    flush_synth()
    seen = [0] * n_local
    alias_msg = None
    while True:
        # One collective per step carries both flags: "some stream is still running" and "some stream aliased".
        # Every rank therefore leaves the loop in the same iteration; a rank that raised on its own would leave
        # the others blocked in the next all-reduce.
        active = False
        if ev is not None:
            st = ev.status()
            active = any(r.active for r in st)
        go, abort = active, alias_msg is not None
        if dist is not None:
            import torch
            flag = torch.tensor([1 if active else 0, 1 if alias_msg else 0], dtype=torch.int32, device=coll_device())
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            go, abort = bool(flag[0].item()), bool(flag[1].item())
        if abort or not go:
            break
        if active:
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
    if ev is not None:
        ev.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if abort:
        # every rank ends with the reference's non-zero panic status; the one that saw the aliasing says where
        print(f"thread 'main' panicked at '{alias_msg}'" if alias_msg else
              f"rank {rank}: stopping, a stream on another rank aliased", file=sys.stderr)
        return 101
    return 0


if __name__ == "__main__":
    sys.exit(main())
