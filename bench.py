#!/usr/bin/env python3
"""Headline benchmark: stream-cell updates/s of the split-step update at 256^3 x 256 streams fp64.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one SimulationObject::update of every stream.  The 256 streams are
block-partitioned over the N ranks (one process per GPU, no data-path exchange
between streams); with N > 1 every step also sums |psi|^2 over the local
streams and all-reduces that mean-density grid over RCCL inside the library
(msm_mean_density_begin), overlapped with the next step.  Inputs are synthetic:
the reference's plane-wave ICs (sim.py) regenerated at N = 256 plus Wigner
noise per stream seed, resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant pass kernel,
timed with HIP events inside the timed region; `cpu_baseline` is the C/OpenMP
port of the reference algorithm (oracle/sp_cpu.c) timed on this host on one stream.

`roofline.achieved` is the kernel's own algorithmic bytes -- every input it must read once and every output it
must write once, per stream-cell (KINDS*: second entry) -- over its measured launch time; `traffic` is what the
PMC counters saw it move.  The whole-step figures `step_frac*` use the reference-level model of 224 B per update
(7 transforms, one read and one write each); `model224_*` is that model's share attributed to the kernel.

Without a launcher (`WORLD_SIZE` unset) `--gpus N` with N > 1 starts its own N rank processes, before anything
in this process has touched a GPU.
"""
import argparse
import datetime
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import msm_amd  # noqa: E402
from msm_amd.ics import plane_wave_psi  # noqa: E402
from msm_amd.sharding import partition_streams  # noqa: E402

PLANE = dict(axis_length=60.0, final_sim_time=2000.0, cfl=0.1, num_data_dumps=64, total_mass=3e16,
             hbar_=0.01, ntot=1e10, k2_cutoff=0.95, alias_threshold=0.001)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
HBM_COPY_GBS = 6290.0          # best streaming copy measured on the part (achievable ceiling)
B_MOVED_STEP_TWO_PASS = 192.0  # bytes the two-pass engine really moves per update (sum over its eight launches)
B_MOVED_STEP_GENERIC = 368.0   # the three-pass engine
B_ALG_STEP = 224.0             # algorithmic bytes per stream-cell update, fp64: 7 transforms x (16 B in + 16 B out)
B_ALG_PASS = 32.0 / 3.0        # one axis pass is a third of a transform under the same model
# pass kind -> (axis passes it performs per stream-cell, bytes it reads + writes per stream-cell)
KINDS = {
    "psi_inv_z": (1, 32.0), "psi_inv_y": (1, 32.0), "psi_invx_rho": (2, 24.0),   # 16 psi in + 8 W out
    "psi_fwd_x": (2, 40.0),                                                      # 8 W in + 16 in + 16 out
    "psi_fwd_y": (1, 32.0), "psi_fwdz_invz": (2, 48.0),
    "w_fwd_y": (1, 16.0), "poisson_z": (2, 16.0), "w_inv_y": (1, 16.0), "phi_inv_x": (1, 8.0),
    "rho_fwd_x": (1, 24.0), "plain": (1, 32.0),
}
# The 256^3 engine does a 3-D transform in two passes (A: x + outer half of y | B: inner half of y + z), so a
# launch carries 1.5 axis passes per transform leg it performs.  Between steps a stream's state is the pass-A image of
# psi, so each kinetic half-step is one round trip through pass B and psi_k is never stored; the kinds keep their names:
#   psi_inv_z      first half-drift: state -> B -> x D -> B^-1 (the B half of F^-1[D psi_k])   T in, T out
#   psi_invx_rho   pass A^-1 of that inverse + rho + pass A of F[rho]                          T in, pair-packed T out
#   poisson_z      pass B of the Poisson solve, there and back                                 in place
#   psi_fwd_x      pass A^-1 of phi, kick, pass A of F[psi]                                    W in, T in, T out
#   psi_fwdz_invz  pass B of F[psi], second half-drift, aliased mass, pass B^-1 -> new state   T in, T out
#   phi_inv_x      pass A^-1 of phi for max|phi|                                               W in
# Bytes moved per update: 32 + 24 + 16 + 40 + 32 + 24 + 16 + 8 = 192.
KINDS_TWO_PASS = {
    "psi_inv_z": (1.5, 32.0), "psi_invx_rho": (3, 24.0), "poisson_z": (3, 16.0), "psi_fwd_x": (3, 40.0),
    "psi_fwdz_invz": (3, 32.0), "phi_inv_x": (1.5, 8.0), "rho_fwd_x": (1.5, 24.0), "plain": (1.5, 32.0),
}


# pass kind -> kernel name in the committed PMC summaries (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very
# command at 32 and at 128 streams per launch, corrected as profiles/summarize_pmc.py documents)
PMC_SUMMARY = {32: os.path.join(ROOT, "profiles", "r02_pmc_traffic_32streams.json"),
               128: os.path.join(ROOT, "profiles", "r02_pmc_traffic_128streams.json")}
PMC_KERNEL = {"psi_inv_z": "k_psi_rt<0>", "psi_invx_rho": "k_invx_rho3<0>", "poisson_z": "k_poisson_b",
              "psi_fwd_x": "k_kick3", "psi_fwdz_invz": "k_psi_rt<1>", "phi_inv_x": "k_phi_max3"}


def measured_traffic(kind, two_pass, cells_per_launch, n3):
    """HBM bytes per launch of the dominant kernel from the PMC counters, when the committed summary covers this
    launch shape (two-pass engine, 32 or 128 x 256^3 cells per launch); None otherwise."""
    per_launch = cells_per_launch // n3
    if not two_pass or cells_per_launch != per_launch * n3 or per_launch not in PMC_SUMMARY or n3 != 256 ** 3 \
            or kind not in PMC_KERNEL:
        return None
    try:
        with open(PMC_SUMMARY[per_launch]) as f:
            return int(json.load(f)[PMC_KERNEL[kind]]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(size, seconds_cap=30.0):
    """C port of the reference step (oracle/sp_cpu.c, OpenMP) on one stream of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sp_cpu          # checker / baseline leg only
    import sp_oracle as o  # ICs for that leg

    base = {k: v for k, v in PLANE.items() if k != "ntot"}
    sim = sp_cpu.CpuSim(size, **base)
    psi = o.sample_quantum_perturbation(o.plane_wave_ics(size), 60.0 / size, 3, PLANE["total_mass"],
                                        PLANE["total_mass"] / PLANE["ntot"], "Wigner", 1)
    sim.set_psi(psi)
    sim.update()  # warm-up; includes the one-off psi_k = F psi
    t0 = time.perf_counter()
    steps = 0
    while steps < 20 and time.perf_counter() - t0 < seconds_cap / 2:
        sim.update()
        steps += 1
    dt = time.perf_counter() - t0
    return dict(value=size ** 3 * steps / dt, unit="stream-cell updates/s", cores=sim.threads, kind="port",
                sample=f"{size}^3 fp64, 1 stream, {steps} timed steps of the C/OpenMP restatement of the "
                       f"reference algorithm (not ArrayFire), {sim.threads} threads",
                seconds=round(dt, 2))


def self_launch(args):
    """Start one fresh child process per rank (never a re-exec of a process that has used the GPU: this parent
    has not touched one) and hand back the worst exit status; rank 0's JSON line passes through on stdout."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for r, pr in enumerate(procs):
        code = pr.wait()
        if code != 0:
            print(f"bench.py: rank {r} exited with status {code}", file=sys.stderr)
            rc = rc or code
            for other in procs:          # a dead rank leaves the others waiting in a collective: end them
                if other.poll() is None:
                    other.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--streams", type=int, default=256, help="total streams over all GPUs")
    ap.add_argument("--batch", type=int, default=128, help="streams per kernel launch (sub-batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    force_dist = os.environ.get("MSM_BENCH_FORCE_DIST") == "1"   # rehearses the RCCL path on one GPU
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist):
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count must equal --gpus")
    N = args.size
    n3 = N ** 3

    store = None
    use_dist = world > 1 or force_dist
    if use_dist:
        # The launcher's key-value store (torch.distributed's TCPStore on MASTER_ADDR:MASTER_PORT) only hands the
        # communicator id from rank 0 to the others.  Everything else -- the density grids, the barriers around the
        # timed region, the maximum over the ranks' clocks -- goes through the library's own RCCL communicator.
        import torch            # before the first Evolver: libmsm_hip.so then binds to the HIP runtime torch loaded
        from torch.distributed import TCPStore
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: this node shows {torch.cuda.device_count()} GPU(s), --gpus {args.gpus} needs "
                             f"one per rank")
        # under torchrun the agent hosts the store and every worker is a client; self-launched, rank 0 hosts it
        agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
        store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]), world,
                         is_master=(rank == 0 and not agent_store), timeout=datetime.timedelta(seconds=300),
                         wait_for_workers=False)

    # block partition of the seed list 1..streams (simulator/src/utils/io.rs:183-213 order)
    per = (args.streams + world - 1) // world
    first, n_local = partition_streams(args.streams, world, rank)
    seeds = [first + 1 + i for i in range(n_local)]

    ev = msm_amd.Evolver(size=N, n_streams=n_local, batch_streams=args.batch, device=local_rank, **PLANE)
    if use_dist:
        key = "msm_bench_comm_id/" + os.environ.get("TORCHELASTIC_RUN_ID", "self") + "/" + \
              os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
        if rank == 0:
            store.set(key, msm_amd.Evolver.comm_unique_id())
        ev.comm_init(bytes(store.get(key)), rank, world)        # RCCL over xGMI, one rank per GPU
        ev.mean_density_begin(args.streams)                     # every step now all-reduces sum_s |psi_s|^2
    psi0 = plane_wave_psi(N)
    ev.set_psi(0, psi0)
    for s in range(1, n_local):
        ev.copy_psi(s, 0)
    for s, seed in enumerate(seeds):
        ev.sample(s, "Wigner", seed)       # stochastic stream resampling, one seed per stream
    del psi0

    taken = [0]

    def one_step(i):
        taken[0] += 1
        ev.update()     # with a communicator: the step's density sum is all-reduced on a second stream, beside step i + 1
        ev.sync()       # per-step status (dt, dump flags) on the host

    def barrier():
        ev.sync()
        if use_dist:
            if taken[0]:
                ev.mean_density(to_host=False)                  # the last step's all-reduce has finished
            ev.comm_allreduce([0.0])                            # barrier over the ranks

    for i in range(args.warmup):
        one_step(i)
    barrier()
    ev.profile(True)
    ev.timer_start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    ev_ms = ev.timer_stop_ms()
    prof = ev.profile_report()
    ev.profile(False)
    st = ev.status()
    steps_done = min(s.n_steps for s in st) - args.warmup if n_local else args.steps
    aliased = sum(s.aliased for s in st)

    if use_dist:
        elapsed = ev.comm_allreduce([elapsed], msm_amd.Evolver.COMM_MAX)[0]     # the slowest rank's clock

    if rank == 0:
        if steps_done != args.steps or aliased:
            raise SystemExit(f"invalid run: {steps_done} of {args.steps} steps done, {aliased} streams aliased")
        value = n3 * args.streams * args.steps / elapsed
        # dominant kernel = the pass kind with the largest total time
        prof = {k: v for k, v in prof.items() if v[1] > 0}
        kind, (kms, kcnt) = max(prof.items(), key=lambda kv: kv[1][0])
        two_pass = ev.derived().get("transform_passes_mode", 0.0) >= 2.0
        npass, moved = (KINDS_TWO_PASS if two_pass else KINDS)[kind]
        cells_per_launch = n3 * min(args.batch, n_local)           # stream-cells one launch processes
        avg_s = kms / kcnt / 1e3
        # the kernel's own algorithmic bytes: each of its inputs read once, each of its outputs written once
        achieved = moved * cells_per_launch / avg_s / 1e9
        model224 = npass * B_ALG_PASS * cells_per_launch / avg_s / 1e9
        traffic = measured_traffic(kind, two_pass, cells_per_launch, n3)             # PMC counters, or null
        b_moved_step = B_MOVED_STEP_TWO_PASS if two_pass else B_MOVED_STEP_GENERIC
        roofline = dict(bound="hbm", kernel=kind, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                        traffic_gbs=round(traffic / avg_s / 1e9, 1) if traffic else None,
                        achievable_peak=HBM_COPY_GBS, achievable_frac=round(achieved / HBM_COPY_GBS, 4),
                        alg_bytes_per_launch=moved * cells_per_launch,
                        avg_launch_ms=round(kms / kcnt, 4), launches=int(kcnt),
                        # share of the reference-level 224-B model attributed to this kernel (not an HBM rate)
                        model224_bytes_per_launch=npass * B_ALG_PASS * cells_per_launch,
                        model224_gbs=round(model224, 1), model224_frac=round(model224 / HBM_PEAK_GBS, 4),
                        # whole step: modelled (224 B per update: 7 transforms x 32 B) and really moved
                        step_alg_gbs=round(value * B_ALG_STEP / world / 1e9, 1),
                        step_frac=round(value * B_ALG_STEP / world / 1e9 / HBM_PEAK_GBS, 4),
                        step_frac_is="modelled: 224 B per update / 8 TB/s",
                        step_moved_gbs=round(value * b_moved_step / world / 1e9, 1),
                        step_moved_frac=round(value * b_moved_step / world / 1e9 / HBM_PEAK_GBS, 4),
                        # stricter model: the four Poisson transforms counted as real-data transforms (168 B per update)
                        step_frac_real_poisson=round(value * 168.0 / world / 1e9 / HBM_PEAK_GBS, 4),
                        per_kind_ms={k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()},
                        passes_per_transform=2 if two_pass else 3)
        out = {
            "metric": "stream-cell updates/sec (N^3*N_streams/step) at 256^3x256; % HBM roofline",
            "value": value, "unit": "stream-cell updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{N}^3 fp64 x {args.streams} streams, plane-wave ICs + Wigner noise "
                                   f"(tomls/planeWave3d_e10_sym.toml, size override), static box",
                       "streams_per_gpu": per, "batch_streams": args.batch,
                       "density_allreduce": use_dist, "event_ms_per_step": ev_ms / args.steps},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:          # a reported baseline of the single-GPU line only
            out["cpu_baseline"] = cpu_baseline(N)
        print(json.dumps(out), flush=True)

    ev.close()


if __name__ == "__main__":
    main()
