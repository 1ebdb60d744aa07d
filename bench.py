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

With one GPU the same line also carries `extra_configs`: the other single-GPU configurations of the benchmark list,
each run after the headline context has been closed, with its own timed steps and HIP-event pass timings --
  c1_512_f64_mft                      examples/gaussian-overdensity-mft.toml as shipped (size 512, fp64, the mean-field
                                      stream alone): the reference's own example at the precision its binary runs
  c2_128_1stream / c2_128_64streams   examples/spherical-tophat.toml at size 128, fp64, static box
  c3_256_64streams                    tomls/planeWave3d_e10_sym.toml at size 256, 64 Wigner streams in sub-batches of 32
  c5_512_f32_expanding_64streams      examples/spherical-tophat-cosmo.toml at size 512, fp32, expanding box, the 64
                                      streams one GPU holds of the 512; the scale-factor bookkeeping between
                                      msm_step_begin and msm_step_end is the native host loop msm-simulator uses
`--configs headline` leaves them out; `--extra-budget-s` bounds their wall time.

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
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import msm_amd  # noqa: E402
from msm_amd._lib import kernel_source_hash  # noqa: E402
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
# command at 32, 128 and 256 streams per launch, corrected as profiles/summarize_pmc.py documents)
PMC_SUMMARY = {n: os.path.join(ROOT, "profiles", f"r03_pmc_traffic_{n}streams.json") for n in (32, 128, 256)}
PMC_KERNEL = {"psi_inv_z": "k_psi_rt<0>", "psi_invx_rho": "k_invx_rho3<0>", "poisson_z": "k_poisson_b",
              "psi_fwd_x": "k_kick3", "psi_fwdz_invz": "k_psi_rt<1>", "phi_inv_x": "k_phi_max3"}


def measured_traffic(kind, two_pass, cells_per_launch, n3):
    """(HBM bytes per launch of the dominant kernel from the PMC counters, note).  The bytes are reported only when a
    committed summary covers this launch shape (two-pass engine, 32, 128 or 256 x 256^3 cells per launch) AND was taken
    from the device sources this run's library was built from (the summary's `_meta.kernel_source_hash`)."""
    per_launch = cells_per_launch // n3
    if not two_pass or cells_per_launch != per_launch * n3 or per_launch not in PMC_SUMMARY or n3 != 256 ** 3 \
            or kind not in PMC_KERNEL:
        return None, "no PMC summary covers this launch shape"
    try:
        with open(PMC_SUMMARY[per_launch]) as f:
            summary = json.load(f)
        profiled = summary.get("_meta", {}).get("kernel_source_hash")
        here = kernel_source_hash()
        if not profiled or profiled != here:
            return None, (f"{os.path.basename(PMC_SUMMARY[per_launch])} was taken from other device sources "
                          f"(hash {str(profiled)[:12]}, this tree {here[:12]}): profile again")
        return int(summary[PMC_KERNEL[kind]]["hbm_bytes_per_launch"]), os.path.basename(PMC_SUMMARY[per_launch])
    except (OSError, KeyError, ValueError) as e:
        return None, f"PMC summary unreadable: {e}"


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
    while steps < 60 and time.perf_counter() - t0 < seconds_cap / 2:      # about 15 s of CPU work
        sim.update()
        steps += 1
    dt = time.perf_counter() - t0
    return dict(value=size ** 3 * steps / dt, unit="stream-cell updates/s", cores=sim.threads, kind="port",
                sample=f"{size}^3 fp64, 1 stream, {steps} timed steps of the C/OpenMP restatement of the "
                       f"reference algorithm (not ArrayFire), {sim.threads} threads",
                seconds=round(dt, 2))


# ---------------------------------------------------------------------------
# The other single-GPU configurations of the benchmark list, on the same clock
# ---------------------------------------------------------------------------
TOPHAT_TOML = """axis_length = 30
final_sim_time = 40
cfl = 0.5
num_data_dumps = 200
total_mass = 1e11
hbar_ = 0.05
sim_name = "{name}"
k2_cutoff = 0.95
alias_threshold = 0.02
dims = 3
size = {size}

[ics]
type = "SphericalTophat"
radius = 5.0
slope = 50
delta = 100
{sampling}{cosmology}"""                 # examples/spherical-tophat.toml and spherical-tophat-cosmo.toml, size overridden
TOPHAT_COSMOLOGY = """
[cosmology]
omega_matter_now = 0.7
omega_radiation_now = 0.0
h = 0.7
z0 = 1.0
max_dloga = 0.01
"""
# bytes one launch of a pass kind reads + writes per stream-cell, by engine (double precision; halve for single):
#   two-pass tilings of 128^3 and 256^3 = KINDS_TWO_PASS, generic engine = KINDS, and the 512^3 single-precision
#   tiling, where |psi|^2 and phi cross memory as real grids (given in single-precision bytes)
#   psi_invx_rho there = the first stream of a pair (8 in, 4 out) + the second, which packs the pair (8 + 4 in, 8 out per
#   pair): 16 per stream-cell; phi_inv_x = the pass that stores phi for the kick (4 in, 4 out) and the one that only
#   takes max|phi| (4 in): 6 on average
KINDS_TILING_512_F32 = {"psi_inv_z": 16.0, "psi_invx_rho": 16.0, "poisson_z": 8.0, "phi_inv_x": 6.0, "psi_fwd_x": 20.0,
                        "psi_fwdz_invz": 16.0, "rho_fwd_x": 12.0, "plain": 16.0}
MOVED_512_F32 = 112.0     # 16 + 16 + 8 + 8 + 20 + 16 + 16 + 8 + 4


def run_config(name, workload, toml_text, size, streams, batch, precision, expanding, scheme, steps, warmup,
               psi0=None, params=None):
    """One configuration from its TOML: context, ICs, `warmup` + `steps` timed updates with the host reading the
    status after every step (as the reference's loop does), then a second, shorter leg with HIP events around every
    pass launch for the per-kind times.  Returns the entry of `extra_configs`."""
    from msm_amd import host as mh
    from msm_amd.expanding import ExpandingHost

    t_enter = time.perf_counter()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "run.toml")
        with open(path, "w") as f:
            f.write(toml_text)
        t = mh.read_toml(path, expanding)
        kw = dict(axis_length=t["axis_length"], final_sim_time=t["final_sim_time"], cfl=t["cfl"],
                  num_data_dumps=t["num_data_dumps"], total_mass=t["total_mass"], k2_cutoff=t["k2_cutoff"],
                  alias_threshold=t["alias_threshold"], particle_mass=t["resolved_particle_mass"],
                  hbar_=t["resolved_hbar_"])
        if params:
            kw.update(params)
        ev = msm_amd.Evolver(size=size, n_streams=streams, batch_streams=batch, precision=precision,
                             cosmology=t["cosmology"] if expanding else None, **kw)
        if psi0 is None:
            psi0 = mh.build_ics(path, size, 3, ev.derived()["dx"], precision, expanding)
    try:
        ev.set_psi(0, psi0)
        del psi0
        for s in range(1, streams):
            ev.copy_psi(s, 0)
        if scheme:
            for s in range(streams):
                ev.sample(s, scheme, s + 1)          # seeds 1..streams, one per stream
        xhost = ExpandingHost(t["cosmology"], streams, precision, t["final_sim_time"], t["num_data_dumps"]) if expanding else None
        state = {"st": ev.status()}

        def one_step():
            if xhost is None:
                ev.update()
                state["st"] = ev.status()           # synchronises: dt, dump flags, activity on the host
            else:
                state["st"] = xhost.update(ev, state["st"])

        for _ in range(warmup):
            one_step()
        ev.sync()
        n0 = min(r.n_steps for r in state["st"])
        ev.timer_start()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        ev.sync()
        wall = time.perf_counter() - t0
        ev_ms = ev.timer_stop_ms()
        st = state["st"]
        done = min(r.n_steps for r in st) - n0
        aliased = sum(r.aliased for r in st)
        if done != steps or aliased:
            return dict(workload=workload, error=f"{done} of {steps} steps done, {aliased} streams aliased")
        # second leg: per-kind pass times (event records switch off the graph replay of small grids, hence not in the timed leg)
        psteps = max(2, min(steps, 10))
        ev.profile(True)
        for _ in range(psteps):
            one_step()
        prof = {k: v for k, v in ev.profile_report().items() if v[1] > 0}
        ev.profile(False)
        st = state["st"]
        mode = ev.derived()["transform_passes_mode"]
    finally:
        ev.close()
    n3 = size ** 3
    f = 1.0 if precision == "f64" else 0.5
    if mode >= 2.0 and size == 512:
        table, moved_step, engine = KINDS_TILING_512_F32, MOVED_512_F32, "two passes per transform, register-resident 2^15-point tiles"
    elif mode >= 2.0:
        table = {k: v[1] * f for k, v in KINDS_TWO_PASS.items()}
        moved_step, engine = B_MOVED_STEP_TWO_PASS * f, "two passes per transform"
    else:
        table = {k: v[1] * f for k, v in KINDS.items()}
        moved_step, engine = B_MOVED_STEP_GENERIC * f, "three passes per transform"
    alg_step = B_ALG_STEP * f
    ups = n3 * streams * steps / wall
    kind, (kms, kcnt) = max(prof.items(), key=lambda kv: kv[1][0])
    cells = n3 * min(batch or streams, streams)
    k_gbs = table[kind] * cells / (kms / kcnt / 1e3) / 1e9
    return dict(workload=workload, engine=engine, dtype=precision, streams=streams, batch_streams=min(batch or streams, streams),
                steps=steps, warmup=warmup, ms_per_step=round(wall / steps * 1e3, 4),
                event_ms_per_step=round(ev_ms / steps, 4), updates_per_s=ups,
                step_frac=round(ups * alg_step / 1e9 / HBM_PEAK_GBS, 4),
                step_frac_is=f"modelled: {alg_step:g} B per update / 8 TB/s",
                moved_frac=round(ups * moved_step / 1e9 / HBM_PEAK_GBS, 4), moved_bytes_per_update=moved_step,
                dominant=dict(kernel=kind, avg_launch_ms=round(kms / kcnt, 4), launches_per_step=round(kcnt / psteps, 2),
                              bytes_per_stream_cell=table[kind], gbs=round(k_gbs, 1), frac=round(k_gbs / HBM_PEAK_GBS, 4)),
                per_kind_ms={k: round(v[0] / v[1], 4) for k, v in prof.items()},
                dumps_crossed=int(max(r.current_dumps for r in st)),
                wall_s=round(time.perf_counter() - t_enter, 1))


def extra_configs(wanted, budget_s):
    sampled = lambda n, scheme: f'\n[sampling]\nseeds = "1 to {n}"\nscheme = "{scheme}"\n'   # noqa: E731
    plane_toml = ('axis_length = 60.0\nfinal_sim_time = 2000\ncfl = 0.1\nnum_data_dumps = 64\ntotal_mass = 3e16\nhbar_ = 0.01\n'
                  'ntot = 1e10\nsim_name = "planeWave3d_e10_sym"\nk2_cutoff = 0.95\nalias_threshold = 0.001\ndims = 3\nsize = 256\n'
                  '[ics]\ntype = "UserSpecified"\npath = "regenerated-at-size-256"\n' + sampled(64, "Wigner"))
    gauss_toml = ('axis_length = 30\nfinal_sim_time = 400\ncfl = 0.02\nnum_data_dumps = 200\ntotal_mass = 1e10\nhbar_ = 0.02\n'
                  'sim_name = "gaussian-overdensity-512-mft"\nk2_cutoff = 0.95\nalias_threshold = 0.02\ndims = 3\nsize = 512\n'
                  '[ics]\ntype = "ColdGauss"\nmean = [15, 15, 15]\nstd = [10, 10, 10]\n')
    specs = {
        "c1_512_f64_mft": lambda: run_config(
            "c1_512_f64_mft", "examples/gaussian-overdensity-mft.toml as shipped: size 512, fp64, mean-field stream only, static box",
            gauss_toml, 512, 1, 0, "f64", False, None, steps=20, warmup=2),
        "c2_128_1stream": lambda: run_config(
            "c2_128_1stream", "examples/spherical-tophat.toml, size 128, fp64, mean-field stream only, static box",
            TOPHAT_TOML.format(name="spherical-tophat", size=128, sampling="", cosmology=""), 128, 1, 0, "f64", False, None,
            steps=150, warmup=10),          # (every step of this TOML lands on one of its 200 dumps: the run ends after 200)
        "c2_128_64streams": lambda: run_config(
            "c2_128_64streams", "examples/spherical-tophat.toml, size 128, fp64, 64 Husimi streams, static box",
            TOPHAT_TOML.format(name="spherical-tophat", size=128, sampling=sampled(64, "Husimi"), cosmology=""), 128, 64, 64, "f64",
            False, "Husimi", steps=50, warmup=5),
        "c3_256_64streams": lambda: run_config(
            "c3_256_64streams", "tomls/planeWave3d_e10_sym.toml, size 256, fp64, 64 Wigner streams in sub-batches of 32, static box",
            plane_toml, 256, 64, 32, "f64", False, "Wigner", steps=15, warmup=2, psi0=plane_wave_psi(256)),
        "c5_512_f32_expanding_64streams": lambda: run_config(
            "c5_512_f32_expanding_64streams",
            "examples/spherical-tophat-cosmo.toml, size 512, fp32, expanding box, 64 Husimi streams in one launch per kernel "
            "(one GPU's share of 512), host scale-factor loop between msm_step_begin and msm_step_end",
            TOPHAT_TOML.format(name="spherical-tophat-cosmo", size=512, sampling=sampled(64, "Husimi"), cosmology=TOPHAT_COSMOLOGY),
            512, 64, 64, "f32", True, "Husimi", steps=10, warmup=2),
    }
    out, t0 = {}, time.perf_counter()
    for name, fn in specs.items():
        if wanted and name.split("_")[0] not in wanted and name not in wanted:
            continue
        if time.perf_counter() - t0 > budget_s:
            out[name] = dict(skipped=f"the --extra-budget-s of {budget_s:g} s was spent by the configurations before it")
            continue
        try:
            out[name] = fn()
        except Exception as e:          # a configuration that cannot run must not take the headline line with it
            out[name] = dict(error=f"{type(e).__name__}: {e}")
    return out



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
    ap.add_argument("--batch", type=int, default=256, help="streams per kernel launch (sub-batch); halved while the batch "
                                                            "buffers do not fit the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--configs", default="all", help="all | comma list of headline, c1, c2, c3, c5 (extras run with one GPU only)")
    ap.add_argument("--extra-budget-s", type=float, default=200.0, help="wall-time bound of the extra configurations")
    args = ap.parse_args()

    force_dist = os.environ.get("MSM_BENCH_FORCE_DIST") == "1"   # rehearses the RCCL path on one GPU
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist):
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rank 0's stdout carries ONE line, the JSON.  Libraries write there too (RCCL prints a version banner when a
    # communicator is formed), so from here on file descriptor 1 is stderr, and the result goes to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count must equal --gpus")
    N = args.size
    n3 = N ** 3
    wanted = None if args.configs == "all" else {c.strip() for c in args.configs.split(",") if c.strip()}

    store = None
    use_dist = world > 1 or force_dist
    if use_dist:
        # The launcher's key-value store (torch.distributed's TCPStore on MASTER_ADDR:MASTER_PORT) only hands the
        # communicator id from rank 0 to the others.  Everything else -- the density grids, the barriers around the
        # timed region, the maximum over the ranks' clocks -- goes through the library's own RCCL communicator.
        import torch            # before the first Evolver: libmsm_hip.so then binds to the HIP runtime torch loaded
        from torch.distributed import TCPStore
        if os.environ.get("MSM_BENCH_SHARE_GPU") == "1":     # rehearsal on a one-GPU box (with a stand-in for RCCL,
            local_rank = local_rank % max(torch.cuda.device_count(), 1)   # which refuses two ranks on one device)
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

    # All local streams in one launch per kernel when the batch buffers fit beside the states (256 streams: 64 GiB of
    # states + 128 GiB of batch buffers); otherwise halve the sub-batch.
    args.batch = max(1, min(args.batch, n_local))
    while True:
        try:
            ev = msm_amd.Evolver(size=N, n_streams=n_local, batch_streams=args.batch, device=local_rank, **PLANE)
            break
        except msm_amd.MsmError as e:
            if not any(k in str(e).lower() for k in ("device memory", "out of memory")) or args.batch <= 32:
                raise
            args.batch //= 2
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
        # one line per rank on stderr, then the ranks agree on the run's validity: every rank must have taken exactly
        # the timed steps with no stream aliased, or all of them exit non-zero together
        print(f"bench.py rank {rank}/{world}: device {local_rank}, {n_local} streams (seeds {seeds[0] if seeds else '-'}.."
              f"{seeds[-1] if seeds else '-'}), {steps_done} timed steps, {aliased} aliased, {elapsed / max(args.steps, 1) * 1e3:.2f} "
              f"ms/step here, RCCL from {msm_amd.Evolver.comm_origin()}", file=sys.stderr, flush=True)
        worst = ev.comm_allreduce([elapsed, -float(steps_done), float(steps_done), float(aliased)], msm_amd.Evolver.COMM_MAX)
        elapsed = worst[0]                                                      # the slowest rank's clock
        if -worst[1] != args.steps or worst[2] != args.steps or worst[3] > 0:
            ev.close()
            raise SystemExit(f"invalid run (rank {rank}): steps per rank between {int(-worst[1])} and {int(worst[2])} of "
                             f"{args.steps}, {int(worst[3])} streams aliased on the worst rank")

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
        traffic, traffic_src = measured_traffic(kind, two_pass, cells_per_launch, n3)   # PMC counters, or null
        b_moved_step = B_MOVED_STEP_TWO_PASS if two_pass else B_MOVED_STEP_GENERIC
        roofline = dict(bound="hbm", kernel=kind, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src,
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
                        # every pass kind's own algorithmic bytes over its average launch time (GB/s), and launches per step
                        per_kind_gbs={k: round((KINDS_TWO_PASS if two_pass else KINDS)[k][1] * cells_per_launch
                                               / (v[0] / max(v[1], 1) / 1e3) / 1e9, 1) for k, v in prof.items()},
                        per_kind_launches_per_step={k: round(v[1] / args.steps, 2) for k, v in prof.items()},
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
        ev.close()
        if world == 1 and not use_dist and (wanted is None or wanted - {"headline"}):
            out["extra_configs"] = extra_configs(None if wanted is None else wanted - {"headline"}, args.extra_budget_s)
        if not args.no_cpu_baseline and world == 1:          # a reported baseline of the single-GPU line only
            out["cpu_baseline"] = cpu_baseline(N)
        os.write(result_fd, (json.dumps(out) + "\n").encode())

    ev.close()


if __name__ == "__main__":
    main()
