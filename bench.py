#!/usr/bin/env python3
"""Benchmark of the FDTD hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reps R] [--grid n] [--variant v]

One "step" is one time step of the propagator (section0 stencil sweep + section1 source
scatter, reference openacc.cpp:148-205) on fields that are already resident in HBM.

N = 1 runs the headline configuration of the benchmark: 512^3 fp32, one Ricker source, the kernels of
Kernel_CUDA_Optimized.  With the defaults W = 5 and K = 45 the timed window is steps 5..49 of a
50-step run, i.e. exactly the reference's timer window (openacc.cpp:5,148).

N > 1 runs the slab-decomposed propagator, one rank per GPU, ghost planes exchanged with the two
neighbours as RCCL point-to-point messages.  Default: the multi-GPU configuration of the benchmark, a
2048^3 grid split into N x-slabs (strong scaling); --weak gives every rank --grid planes instead.
`python bench.py --gpus N` starts its own N ranks (child processes, started before this process
touches the GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it
is one of the ranks.  After that run rank 0, when it sees N GPUs, measures the same workload through the
library's one-process multi-device session (ghost planes pushed by the copy engines) in a process of its
own; the line carries both runs and `value` is the faster one (peer_leg below says why).

Protocol: the reference driver repeats every configuration 5 times and reports mean and population
standard deviation (main.cpp:99-111,388).  Here a repetition zeroes the wavefield, does W untimed
steps and times exactly K steps between barrier + synchronize on both sides (maximum over ranks).
`value` and `ms_per_step` are the mean over the repetitions; `reps` carries the spread.
Rank 0 prints one JSON line.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak
ALGO_BYTES_PER_UPDATE = 16.0    # fp32: read u[t0], u[t1], m; write u[t2]; 4 B each
ALGO_BYTES_PER_UPDATE_F16 = 10.0  # fp16 storage: u levels 2 B each, m stays 4 B
MULTI_GPU_GRID = 2048           # the multi-GPU configuration: 2048^3 across the GPUs of a node


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=45)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=5, help="repetitions (main.cpp:388); mean and population stddev are reported")
    ap.add_argument("--grid", type=int, default=0,
                    help="cells per axis; default 512 for one GPU, 2048 (split along x) for several")
    ap.add_argument("--variant", default="auto",
                    choices=["auto", "flat", "naive", "march", "tile3d", "vec", "march2"],
                    help="auto: two steps per sweep where the shape allows, else the marching kernel")
    ap.add_argument("--storage", default="f32", choices=["f32", "f16"],
                    help="f16: wavefield levels stored as binary16 in HBM, fp32 arithmetic (config 4; N=1 only)")
    ap.add_argument("--global-x", type=int, default=0,
                    help="N>1: total planes along x split over the ranks (default: --grid, i.e. the cube is cut into N slabs)")
    ap.add_argument("--weak", action="store_true",
                    help="N>1: every rank owns --grid planes of a (grid*N) x grid x grid domain (weak scaling)")
    ap.add_argument("--self-exchange", action="store_true",
                    help="N=1 rehearsal of the N>1 schedule: the rank is its own two neighbours, RCCL messages "
                         "included (measures what the exchange costs next to the interior sweep; not the headline)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "peer", "both"],
                    help="ghost exchange of the slab runs: rccl = one process per GPU, RCCL point-to-point messages "
                         "(default); peer = one process, one issuing thread per slab, hipMemcpyPeerAsync pushes on the "
                         "copy engines (the library's fdtd_hip_multigpu_* session); both = one run of each, the peer "
                         "run's figures under `exchange_peer`")
    ap.add_argument("--slabs", type=int, default=0,
                    help="--transport peer: number of slabs (default --gpus); more slabs than GPUs puts several slabs "
                         "on a device (a rehearsal of the N-GPU schedule on fewer GPUs)")
    ap.add_argument("--no-breakdown", action="store_true",
                    help="--transport peer: skip the two extra runs (no exchange; no exchange and no interior / "
                         "boundary split) that price the exchange and the schedule")
    ap.add_argument("--velocity", default="constant", choices=["constant", "layered"],
                    help="N=1: m = 1.5 everywhere (the reference's benchmark input, main.cpp:352) or five horizontal "
                         "layers of velocity 1.5 .. 4.5 (m = 1/v^2): same kernels, heterogeneous medium")
    ap.add_argument("--space-order", type=int, default=4, choices=[4, 6, 8, 10, 12],
                    help="space order of the stencil (main.cpp:27-32: arrays padded by that many cells); N = 1 only, "
                         "orders above 4 run one step per sweep through the library's time loop")
    ap.add_argument("--no-traffic", action="store_true",
                    help="N=1: skip the two counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around a short run of "
                         "this same command in child processes) that measure roofline.traffic")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=50, help="oracle steps for the CPU baseline sample")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# N > 1 without a launcher: this process starts the ranks and relays rank 0's line.  Nothing here
# touches torch or HIP: the children are fresh interpreters.
# ---------------------------------------------------------------------------------------------
def spawn_ranks(args):
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    failed = None
    line = None
    try:
        out0, _ = None, None
        pending = set(range(n))
        while pending and failed is None:
            for r in sorted(pending):
                if r == 0:
                    try:
                        out0, _ = procs[0].communicate(timeout=0.2)
                    except subprocess.TimeoutExpired:
                        continue
                    rc = procs[0].returncode
                else:
                    rc = procs[r].poll()
                    if rc is None:
                        continue
                pending.discard(r)
                if rc != 0:
                    failed = (r, rc)
                    break
            time.sleep(0.05)
        if out0 is not None:
            for ln in out0.decode(errors="replace").splitlines():
                if ln.startswith("{"):
                    line = ln
                else:
                    print(ln, file=sys.stderr)
    finally:
        for p in procs:            # the ranks this process started, by handle
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    if failed is not None:
        print(f"bench.py: rank {failed[0]} exited with code {failed[1]}", file=sys.stderr)
        return 1
    if line is None:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    print(line)
    return 0


def library_source_hash():
    """Hash of the kernel sources the loaded library was built from (what a traffic measurement is tied to)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "accelerated-3d-acoustic-fdtd-kernel_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def measure_traffic(kernel_substr, launches_expected):
    """HBM-side bytes per launch of the dominant kernel, measured now: two child processes run a short version of this
    very command under `rocprofv3 --pmc`, one counter per pass (FETCH_SIZE, then WRITE_SIZE: the counter blocks do not
    hold both), and the per-dispatch values of the kernel are averaged.  gfx950 counts a 128-byte read request as 64
    bytes, so FETCH_SIZE is doubled; WRITE_SIZE is exact.  Returns a dict or None (no profiler, a pass failed)."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3")
    if prof is None:
        return None
    # already under a profiler (its settings would be inherited by the child runs and combined with the counter
    # collection): leave the measurement to whoever is profiling
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or \
            "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return None
    argv = [a for a in sys.argv[1:] if a not in ("--no-cpu-baseline",)]
    # strip the options this function sets itself
    clean, skip = [], 0
    for a in argv:
        if skip:
            skip -= 1
            continue
        if a in ("--steps", "--warmup", "--reps", "--cpu-steps"):
            skip = 1
            continue
        if a.split("=")[0] in ("--steps", "--warmup", "--reps", "--cpu-steps"):
            continue
        clean.append(a)
    child = [sys.executable, os.path.abspath(__file__)] + clean + ["--steps", "8", "--warmup", "2", "--reps", "1",
                                                                  "--no-cpu-baseline", "--no-traffic"]
    out = {}
    tmp = tempfile.mkdtemp(prefix="fdtd_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                p = subprocess.run(["timeout", "-s", "KILL", "150", prof, "--pmc", counter, "--output-format", "csv", "-d", d,
                                    "--"] + child, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                   timeout=170)
            except (subprocess.TimeoutExpired, OSError):
                return None
            files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
            if p.returncode != 0 or not files:
                return None
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                    if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
            if not vals:
                return None
            out[counter + "_KB_per_launch"] = sum(vals) / len(vals)
            out[counter + "_launches"] = len(vals)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["hbm_read_bytes_per_launch"] = 2 * 1024 * out["FETCH_SIZE_KB_per_launch"]
    out["hbm_write_bytes_per_launch"] = 1024 * out["WRITE_SIZE_KB_per_launch"]
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
    out["method"] = ("measured in this run: two child runs of this command (8 timed steps) under rocprofv3 --pmc, FETCH_SIZE "
                     "and WRITE_SIZE in separate passes, averaged over the kernel's dispatches; FETCH_SIZE doubled (gfx950 "
                     "counts a 128-byte request as 64 bytes)")
    return out


def cpu_baseline(pkg, n, steps):
    """Time the CPU oracle (test infrastructure; never on the product path) on a bounded sample of
    the same workload: n^3, `steps` steps of the Ricker run, its own section timers, on all the
    host threads this job may use; and once more on one thread on a smaller grid."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    # OMP_NUM_THREADS is the job's CPU share on the GPU box (os.cpu_count() reports the whole machine there)
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    lib = ob.oracle()
    set_threads = getattr(lib, "oracle_set_threads", None)

    def sample(n_, steps_, threads):
        if set_threads is not None:
            set_threads(threads)
        u, m, src, coords = pkg.standard_inputs(n_, n_, n_, steps_, 1)
        prof = ob.run_oracle(m, src, coords, u)
        secs = prof.section0 + prof.section1
        return n_ ** 3 * (steps_ - pkg.WARMUP_STEPS) / secs / 1e6, secs

    rate, secs = sample(n, steps, cores)
    out = {
        "value": round(rate, 1),
        "unit": "Mcells/s",
        "cores": cores,
        "host_cpus_visible": os.cpu_count(),
        "kind": "port",
        "sample": f"{n}^3 fp32 Ricker run, {steps} steps, steps {pkg.WARMUP_STEPS}..{steps - 1} timed "
                  f"({secs:.2f} s), OpenMP over x,y on {cores} threads (the job's CPU share); bit-identical "
                  f"restatement of openacc.cpp:159-196, standing in for `nvc++ -acc=multicore` (not installed)",
    }
    if set_threads is not None:
        n1 = min(n, 256)
        steps1 = min(steps, 25)
        rate1, secs1 = sample(n1, steps1, 1)
        set_threads(cores)
        out["one_thread"] = {"value": round(rate1, 1), "unit": "Mcells/s", "cores": 1,
                             "sample": f"{n1}^3, {steps1} steps, {secs1:.2f} s timed"}
    return out


def stats(xs):
    """Mean and population standard deviation (main.cpp:99-111)."""
    mean = sum(xs) / len(xs)
    var = sum((x - mean) ** 2 for x in xs) / len(xs)
    return mean, var ** 0.5


def roofline_basic(algo_bytes, launch_cells, kernel_ms, updates_per_launch, kernel_desc):
    achieved = algo_bytes * launch_cells / (kernel_ms * 1e-3) / 1e9
    kbytes = 10.0 if updates_per_launch == 2 else 16.0
    return {
        "bound": "hbm",
        "kernel": kernel_desc,
        "cell_updates_per_launch": launch_cells,
        "achieved": round(achieved, 1),
        "achieved_basis": f"{algo_bytes:g} B per cell update (algorithmic, single-step model) x cell updates per launch / kernel time",
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": None,
        "avg_kernel_ms": round(kernel_ms, 4),
        "compulsory_bytes_per_launch": kbytes * launch_cells,
        "compulsory_GBps": round(kbytes * launch_cells / (kernel_ms * 1e-3) / 1e9, 1),
    }, kbytes


def peer_main(args):
    """One process, the library's multi-device session: slabs on the devices, one issuing thread per slab, ghost
    planes pushed by the copy engines.  A step is a time step of the whole domain."""
    # three streams per slab; with HIP's default of 4 hardware queues per process two chains of one slab can land
    # on one queue and serialise (must be set before the HIP runtime initialises)
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("FDTD_HW_QUEUES", "8")
    import torch
    from __graft_entry__ import load_package

    pkg = load_package()
    ngpu_visible = torch.cuda.device_count()
    if ngpu_visible < 1:
        raise SystemExit("bench.py: no GPU visible")
    slabs = args.slabs or args.gpus
    if os.environ.get("FDTD_MULTIGPU_DEVICES"):
        devices = [int(t) for t in os.environ["FDTD_MULTIGPU_DEVICES"].split(",") if t != ""]
        slabs = len(devices)
    else:
        ndev = min(args.gpus, ngpu_visible)
        per = (slabs + ndev - 1) // ndev
        devices = [min(i // per, ndev - 1) for i in range(slabs)]     # neighbouring slabs share a device
    n = args.grid or (512 if (args.gpus == 1 and slabs == 1) else MULTI_GPU_GRID)
    NX = n * slabs if args.weak else (args.global_x or n)
    K, W, R = args.steps, args.warmup, max(1, args.reps)
    total = K + W
    g = pkg.Geom(NX + 8, n + 8, n + 8, 0, NX - 1, 0, n - 1, 0, n - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)

    def run(flags, reps):
        times, st, mx = [], None, 0.0
        with pkg.MultiGpuSession(g, devices, flags) as s:
            for rep in range(reps):
                s.fill_standard_inputs(total + 1, 1)
                if W:
                    s.advance(0, W - 1)
                s.stats(reset=True)
                t0 = time.perf_counter()
                s.advance(W, total - 1)          # returns when every device has finished
                times.append(time.perf_counter() - t0)
                st = s.stats()
            mx, bad = s.max_abs()
            assert bad == 0 and mx > 0.0, "field is empty or not finite"
        return times, st, mx

    elapsed_reps, st, _ = run(0, R)
    mean_s, sd_s = stats(elapsed_reps)
    cells_per_step = float(NX) * n * n
    mcells = cells_per_step * K / mean_s / 1e6
    rates = [cells_per_step * K / e / 1e6 for e in elapsed_reps]
    ms_step = mean_s / K * 1e3
    exchange = {
        "transport": "hipMemcpyPeerAsync pushes on the copy engines, one process",
        "issuing": "one host thread per slab" if st["thread_per_slab"] else "the calling thread for every slab",
        "ghost_bytes_per_side_per_step": 4 * (n + 8) * (n + 8) * 4 / 2,
        "host_issue_us_per_pair": round(st["issue_us_per_op"], 1),
        "host_wait_for_neighbour_thread_us_per_pair": round(st["wait_us_per_op"], 1),
        "device_us_per_pair": round(st["wall_us_per_op"], 1),
        "interior_boundary_split": bool(st["split"]),
        "outer_thirds_first": bool(st["outer_thirds_first"]),
    }
    if not args.no_breakdown and slabs > 1:
        # what the exchange costs: the same schedule without pushes and without waits for the neighbours (wrong
        # results, right timing); what the schedule costs: one whole-slab call per pair and slab on top of that
        t_noex, _, _ = run(pkg.abi.MULTIGPU_NO_EXCHANGE, 1 if R < 3 else 2)
        t_whole, _, _ = run(pkg.abi.MULTIGPU_NO_EXCHANGE | pkg.abi.MULTIGPU_NO_OVERLAP, 1 if R < 3 else 2)
        ms_noex, ms_whole = min(t_noex) / K * 1e3, min(t_whole) / K * 1e3
        t_one, st_one, _ = run(pkg.abi.MULTIGPU_ONE_THREAD, 1 if R < 3 else 2)
        exchange.update({
            "exposed_ms_per_step": round(max(0.0, ms_step - ms_noex), 4),
            "sweep_only_ms_per_step": round(ms_whole, 4),
            "same_schedule_without_exchange_ms_per_step": round(ms_noex, 4),
            "schedule_overhead_ms_per_step": round(ms_noex - ms_whole, 4),
            "one_issuing_thread": {"ms_per_step": round(min(t_one) / K * 1e3, 4),
                                   "host_issue_us_per_pair": round(st_one["issue_us_per_op"], 1)},
        })
    per_dev = max(devices.count(d) for d in set(devices))
    own_cells = cells_per_step / slabs
    pair_ms = st["wall_us_per_op"] * 1e-3 / per_dev      # one slab's pair, when the slabs of a device run one after the other
    roof, kbytes = roofline_basic(ALGO_BYTES_PER_UPDATE, own_cells * 2, pair_ms, 2,
                                  "all launches of one pair of steps on one slab (interior sweep, boundary zones, seam "
                                  "passes; wall time per pair / slabs per device)")
    out = {
        "metric": "Mcells/s", "value": round(mcells, 1), "unit": "Mcells/s", "n_gpus": args.gpus, "steps": K, "warmup": W,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True,
        "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "effective_GBps_at_algorithmic_bytes": round(mcells * ALGO_BYTES_PER_UPDATE / 1e3, 1),
        "reps": {"n": R, "protocol": "each repetition: inputs made on the devices, warm-up steps untimed, then the timed steps",
                 "ms_per_step_mean": round(ms_step, 4), "ms_per_step_stddev": round(sd_s / K * 1e3, 4),
                 "Mcells_per_s": [round(r, 1) for r in rates], "Mcells_per_s_stddev": round(stats(rates)[1], 1)},
        "config": {"workload": f"{NX}x{n}x{n} fp32 as {slabs} x-slabs on {len(set(devices))} GPU(s) (devices {devices}), two steps "
                               f"per sweep and exchange (4 ghost planes of u(t+2) per side) pushed with hipMemcpyPeerAsync, "
                               f"{total} steps ({W} untimed)"
                               + (" [several slabs per GPU: schedule rehearsal]" if per_dev > 1 else ""),
                   "variant": "auto", "algorithmic_bytes_per_cell_update": ALGO_BYTES_PER_UPDATE,
                   "kernel_bytes_per_cell_update": kbytes, "ranks": 1, "slabs": slabs},
        "roofline": roof,
        "exchange": exchange,
        "library_source_hash": library_source_hash(),
    }
    print(json.dumps(out))
    return 0


def peer_leg(args, out, world):
    """After the ranks' run over RCCL, rank 0 measures the same workload through the library's multi-device session
    (one process, ghost planes pushed by the copy engines) in a process of its own, and the line carries both.

    Why both: RCCL's send/receive kernel needs compute units of its own (64 workgroups of 132 registers per lane; it
    cannot share a CU with a workgroup of the two-step sweep), and a sweep launch of 256 workgroups that finds some CUs
    taken needs a second round for the displaced ones.  With messages to itself on one GPU that costs 3 % of a step;
    between GPUs the kernel lives as long as the transfer takes on the link.  The copy engines take no CUs.  `value` is
    the faster of the two runs; `config.transport` says which, `ranks_over_rccl` / `one_process_peer_copies` hold both.
    FDTD_BENCH_PEER_LEG=0: the RCCL run only; =force: also when this process sees fewer GPUs than ranks (rehearsal)."""
    mode = os.environ.get("FDTD_BENCH_PEER_LEG", "1")
    rccl = {"value": out["value"], "ms_per_step": out["ms_per_step"], "transport": out["exchange"]["transport"],
            "Mcells_per_s": out["reps"]["Mcells_per_s"]}
    out["ranks_over_rccl"] = rccl
    out["config"]["transport"] = "rccl"
    if mode == "0":
        return
    import torch
    if mode != "force" and torch.cuda.device_count() < world:
        out["one_process_peer_copies"] = {"skipped": f"this process sees {torch.cuda.device_count()} GPU(s), the run has {world} ranks"}
        return
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
                        "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FDTD_BENCH_BACKEND")
           and not k.startswith("TORCHELASTIC_")}
    argv = [sys.executable, os.path.abspath(__file__), "--transport", "peer", "--gpus", str(world), "--steps", str(args.steps),
            "--warmup", str(args.warmup), "--reps", str(args.reps), "--no-breakdown"]
    if args.grid:
        argv += ["--grid", str(args.grid)]
    if args.global_x:
        argv += ["--global-x", str(args.global_x)]
    if args.weak:
        argv += ["--weak"]
    try:
        p = subprocess.run(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=float(os.environ.get("FDTD_BENCH_PEER_LEG_TIMEOUT", "240")))
        lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            tail = p.stderr.decode(errors="replace").strip().splitlines()[-3:]
            out["one_process_peer_copies"] = {"failed": f"exit code {p.returncode}", "stderr_tail": tail}
            return
        peer = json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        out["one_process_peer_copies"] = {"failed": "time limit"}
        return
    out["one_process_peer_copies"] = {"value": peer["value"], "ms_per_step": peer["ms_per_step"],
                                      "Mcells_per_s": peer["reps"]["Mcells_per_s"], "workload": peer["config"]["workload"],
                                      "exchange": peer.get("exchange", {})}
    if peer["value"] > out["value"]:
        for k in ("value", "ms_per_step", "effective_GBps_at_algorithmic_bytes", "reps", "roofline", "exchange"):
            if k in peer:
                out[k] = peer[k]
        out["config"]["workload"] = peer["config"]["workload"]
        out["config"]["transport"] = "peer copies (one process)"
        for k in ("ranks", "slabs"):
            if k in peer["config"]:
                out["config"][k] = peer["config"][k]


def run_both(args):
    """--transport both: the RCCL run, then the peer run in a process of its own; one line, the peer run's figures
    under `exchange_peer`."""
    def child(transport):
        argv = [a for a in sys.argv[1:]]
        for i, a in enumerate(argv):
            if a == "--transport":
                argv[i + 1] = transport
            elif a.startswith("--transport="):
                argv[i] = "--transport=" + transport
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, stdout=subprocess.PIPE)
        lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            raise SystemExit(f"bench.py: the --transport {transport} run failed (exit code {p.returncode})")
        return json.loads(lines[-1])
    rccl = child("rccl")
    peer = child("peer")
    rccl["exchange_peer"] = dict(peer.get("exchange", {}), value=peer["value"], ms_per_step=peer["ms_per_step"],
                                 workload=peer["config"]["workload"])
    print(json.dumps(rccl))
    return 0


def main():
    args = parse_args()
    if args.transport == "both":
        sys.exit(run_both(args))
    if args.transport == "peer":
        if int(os.environ.get("RANK", "0")) != 0:
            return          # under a launcher: rank 0 drives every device, the other ranks have nothing to do
        sys.exit(peer_main(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    slabs = world > 1 or args.self_exchange
    if slabs:
        # The slab schedule keeps three device queues busy at once (interior chain, boundary chain,
        # RCCL).  With HIP's default of 4 hardware queues per process the boundary stream was seen to
        # land on the interior stream's queue, which serialises the two chains; 8 keeps them apart.
        # Must be in the environment before the HIP runtime initialises, i.e. before torch is imported.
        os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("FDTD_HW_QUEUES", "8")

    import numpy as np
    import torch
    from __graft_entry__ import load_package

    pkg = load_package()
    L = pkg.lib()

    # one GPU per rank; the modulo only matters when the N>1 path is rehearsed on a 1-GPU box
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    n = args.grid or (512 if world == 1 else MULTI_GPU_GRID)
    global_x = 0
    if world > 1 and not args.weak:
        global_x = args.global_x or n
    elif args.global_x:
        global_x = args.global_x
    K, W, R = args.steps, args.warmup, max(1, args.reps)
    total_steps = K + W
    variant = pkg.VARIANTS[args.variant]
    h16 = args.storage == "f16"
    algo_bytes = ALGO_BYTES_PER_UPDATE_F16 if (h16 and not slabs) else ALGO_BYTES_PER_UPDATE
    elapsed_reps = []
    kernel_ms_reps = []
    exchange = None

    if slabs:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # "nccl" is RCCL on ROCm.  FDTD_BENCH_BACKEND=gloo stages the ghosts through the host
        # (for rehearsing this path with several ranks on one GPU; never a performance number).
        backend = os.environ.get("FDTD_BENCH_BACKEND", "nccl")
        # RCCL prints a version banner on stdout when its first communicator comes up; this program's
        # stdout carries one JSON line and nothing else, so file descriptor 1 points at stderr until the
        # untimed steps are through.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if backend == "nccl":
            pkg.slab.init_rccl(dev)
        else:
            dist.init_process_group(backend)
        ranks_seen = dist.get_world_size()
        slab = pkg.slab
        # two steps per exchange (fused sweep, 4-plane ghosts) where the slab's cross-section allows it
        gq = pkg.Geom(n + 8, n + 8, n + 8, 0, n - 1, 0, n - 1, 0, n - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
        pairs = args.storage == "f32" and bool(L.fdtd_hip_two_steps_fuses(C.byref(gq), 1, variant))
        maker = slab.standard_pair_benchmark if pairs else slab.SlabPropagator.standard_benchmark
        prop = maker(n_per_rank=n, ny=n, nz=n, nt=total_steps + 1, rank=rank, world=world, device=dev,
                     variant=variant, global_x=global_x, self_exchange=args.self_exchange and world == 1)

        def advance(t_from, t_to):
            t = t_from
            if pairs:
                while t + 1 < t_to:
                    prop.pair(t)
                    t += 2
                if t < t_to:
                    prop.single(t)
            else:
                for t in range(t_from, t_to):
                    prop.step(t)

        for rep in range(R):
            if rep:
                prop.reset()
            advance(0, W)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            if rep == 0:
                sys.stdout.flush()
                os.dup2(saved_stdout, 1)
                os.close(saved_stdout)
            t_start = time.perf_counter()
            advance(W, total_steps)
            t_issue = time.perf_counter() - t_start
            prop.stepper.finish()       # the last step's ghosts have arrived
            t_fin = time.perf_counter() - t_start
            dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t_start
            if os.environ.get("FDTD_BENCH_DEBUG"):
                print(f"[rank {rank}] rep {rep}: issue {t_issue*1e3:.3f} ms, finished {t_fin*1e3:.3f} ms, "
                      f"after barrier {elapsed*1e3:.3f} ms", file=sys.stderr)
            tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed_reps.append(float(tmax.item()))
        kernel_ms = prop.mean_interior_kernel_ms()
        kernel_ms_reps = [kernel_ms]
        cells_per_step = float(prop.layout.NX) * n * n
        updates_per_launch = 2 if pairs else 1
        launch_cells = prop.interior_cells() * updates_per_launch
        # what the exchange costs: per-step time of this run against the kernel time of the interior range
        # scaled to the whole slab (the time the same cells take without any exchange)
        nx_loc = prop.layout.nx_loc(rank)
        own_cells = float(nx_loc) * n * n
        ms_step = stats(elapsed_reps)[0] / K * 1e3
        exchange = {
            "ghost_bytes_per_side_per_step": (4 if pairs else 2) * (n + 8) * (n + 8) * 4 / (2 if pairs else 1),
            "transport": "RCCL point-to-point" if backend == "nccl" else "gloo, host staged (rehearsal)",
        }
        if kernel_ms == kernel_ms and launch_cells > 0:      # (NaN: too few steps for a timed interior launch)
            sweep_ms_per_step = kernel_ms / updates_per_launch * own_cells / (launch_cells / updates_per_launch)
            exchange["exposed_ms_per_step"] = round(max(0.0, ms_step - sweep_ms_per_step), 4)
            exchange["sweep_only_ms_per_step"] = round(sweep_ms_per_step, 4)
        dist.destroy_process_group()
    else:
        ranks_seen = 1
        order = args.space_order
        if order != 4:
            if h16 or args.variant != "auto":
                raise SystemExit("bench.py: --space-order above 4 goes with fp32 storage and --variant auto")
            assert L.fdtd_hip_set_space_order(order) == 0
        u, m, src, coords = pkg.standard_inputs(n, n, n, total_steps, 1, halo=order)
        if args.velocity == "layered":
            v = 1.5 + 0.75 * np.minimum(4, (np.arange(m.shape[2]) * 5) // m.shape[2])
            m[:] = (1.0 / (v * v)).astype(np.float32)[None, None, :]
        u_d = torch.from_numpy(u).to(dev)
        m_d = torch.from_numpy(m).to(dev)
        s_d = torch.from_numpy(src).to(dev)
        c_d = torch.from_numpy(coords).to(dev)
        del u, m
        npad = n + 2 * order
        g = pkg.Geom(npad, npad, npad, 0, n - 1, 0, n - 1, 0, n - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
        stream = torch.cuda.current_stream()
        sh = C.c_void_p(stream.cuda_stream)
        if h16:
            u_d = u_d.to(torch.float16)

        def stencil(t0, t1, t2):
            if h16:
                return L.fdtd_hip_stencil_step_h16(C.byref(g), m_d.data_ptr(), u_d.data_ptr(), t0, t1, t2, sh)
            return L.fdtd_hip_stencil_step(C.byref(g), m_d.data_ptr(), u_d.data_ptr(), t0, t1, t2, variant, sh)

        def inject(t2, t):
            fn = L.fdtd_hip_source_inject_h16 if h16 else L.fdtd_hip_source_inject
            return fn(C.byref(g), m_d.data_ptr(), s_d.data_ptr(), c_d.data_ptr(), u_d.data_ptr(), t2, t, 0, 0, 1, 3, sh)

        # two steps per sweep where the library takes the fused sweep for this box
        fused = (not h16) and order == 4 and K >= 2 and bool(L.fdtd_hip_two_steps_fuses(C.byref(g), 1, variant))
        updates_per_launch = 2 if fused else 1
        use_events = os.environ.get("FDTD_BENCH_EVENTS", "1") != "0"
        # every n-th two-step call carries the two events that take the start and stop times of its dispatch (each
        # pair of events costs a few microseconds on the stream: 6 % of a pair at 256^3, 0.7 % at 512^3)
        ev_stride = max(1, int(os.environ.get("FDTD_BENCH_EVENT_STRIDE", "4")))
        if fused:
            # Two steps per sweep: four level buffers rotate (the three ring slots and one more).  All
            # halos are zero and the source sits inside the grid, so no shell has to follow the ring.
            extra = torch.zeros_like(u_d[0])
            P = [u_d[0], u_d[1], u_d[2], extra]
            margs = (m_d.data_ptr(), s_d.data_ptr(), c_d.data_ptr())
            n_pairs = K // 2
            # Every two-step call hands the library two events.  Where the call is a single sweep launch
            # (512-wide rows) they receive the start and stop times of that dispatch itself, which is the
            # kernel duration a profiler reports; otherwise they are recorded on the stream around all the
            # launches of the call (sweep launches, seam pass, scatter).
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_pairs)]
            for a, b in ev:  # torch creates the HIP event at the first record; the library re-records it
                a.record(stream)
                b.record(stream)
            evh = [(C.c_void_p(a.cuda_event), C.c_void_p(b.cuda_event)) for a, b in ev]

        for rep in range(R):
            if rep:
                u_d.zero_()
                if fused:
                    extra.zero_()
            if fused:
                cur, prev, third, spare = 0, 2, 1, 3          # time 0: t0 = 0, t1 = 2, t2 = 1
                # The untimed steps take the same kernels as the timed ones (as the operators' time loop does):
                # one single step if their number is odd, pairs for the rest.  With single steps only, the
                # process's first launch of the two-step sweep would fall into the timed window (about 1 ms).
                t = 0
                if W % 2:
                    assert L.fdtd_hip_one_step(C.byref(g), *margs, P[cur].data_ptr(), P[prev].data_ptr(),
                                               P[third].data_ptr(), t, 0, 0, 1, 3, variant, sh) == 0
                    cur, prev, third = third, cur, prev
                    t += 1
                while t < W:
                    assert L.fdtd_hip_two_steps_ex(C.byref(g), *margs, P[cur].data_ptr(), P[prev].data_ptr(),
                                                   P[third].data_ptr(), P[spare].data_ptr(), t, 0, 0, 1, 3, variant, 1,
                                                   None, None, sh) == 0
                    cur, prev, third, spare = spare, third, cur, prev
                    t += 2
                torch.cuda.synchronize()
                t_start = time.perf_counter()
                t = W
                for i in range(n_pairs):
                    # The source is inside the grid (flag 1: no halo-corner launches).
                    hb, he = evh[i] if (use_events and i % ev_stride == 0) else (None, None)
                    rc = L.fdtd_hip_two_steps_ex(C.byref(g), *margs, P[cur].data_ptr(), P[prev].data_ptr(),
                                                 P[third].data_ptr(), P[spare].data_ptr(), t, 0, 0, 1, 3, variant, 1,
                                                 hb, he, sh)
                    assert rc == 0, rc
                    cur, prev, third, spare = spare, third, cur, prev
                    t += 2
                if t < total_steps:
                    assert L.fdtd_hip_one_step(C.byref(g), *margs, P[cur].data_ptr(), P[prev].data_ptr(),
                                               P[third].data_ptr(), t, 0, 0, 1, 3, variant, sh) == 0
                    cur, prev, third = third, cur, prev
                torch.cuda.synchronize()
                elapsed = time.perf_counter() - t_start
                if use_events:
                    kernel_ms_reps.append(float(np.mean([a.elapsed_time(b) for a, b in ev[::ev_stride]])))
                else:   # experiment: no events at all, the launch period stands in for the kernel time
                    kernel_ms_reps.append(elapsed / max(1, n_pairs) * 1e3)
                newest = P[cur]
            elif args.variant == "auto" and not h16 and os.environ.get("FDTD_BENCH_TIME_LOOP", "1") != "0":
                # Boxes the library does not fuse (below 16 M cells): its own time loop, which runs them with one
                # launch per step (sweep and scatter in one kernel).  No per-kernel events: the time per step,
                # launch gap included, stands in for the kernel time.
                tl_args = (C.byref(g), m_d.data_ptr(), s_d.data_ptr(), c_d.data_ptr(), u_d.data_ptr(), 0, 0, 1, 3)
                assert L.fdtd_hip_time_loop(*tl_args, 0, W - 1, variant, sh, None) == 0
                torch.cuda.synchronize()
                t_start = time.perf_counter()
                assert L.fdtd_hip_time_loop(*tl_args, W, total_steps - 1, variant, sh, None) == 0
                torch.cuda.synchronize()
                elapsed = time.perf_counter() - t_start
                kernel_ms_reps.append(elapsed / K * 1e3)
                newest = u_d[total_steps % 3]
            else:
                for t in range(W):
                    assert stencil(t % 3, (t + 2) % 3, (t + 1) % 3) == 0
                    assert inject((t + 1) % 3, t) == 0
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
                torch.cuda.synchronize()
                t_start = time.perf_counter()
                for i, t in enumerate(range(W, total_steps)):
                    t0, t1, t2 = t % 3, (t + 2) % 3, (t + 1) % 3
                    ev[i][0].record(stream)
                    rc = stencil(t0, t1, t2)
                    ev[i][1].record(stream)
                    rc |= inject(t2, t)
                    assert rc == 0, rc
                torch.cuda.synchronize()
                elapsed = time.perf_counter() - t_start
                kernel_ms_reps.append(float(np.mean([a.elapsed_time(b) for a, b in ev])))
                newest = u_d[total_steps % 3]
            elapsed_reps.append(elapsed)
            max_u = float(newest.float().abs().max().item())
            assert np.isfinite(max_u) and max_u > 0.0, "field is empty or not finite"
        kernel_ms = float(np.mean(kernel_ms_reps))
        cells_per_step = float(n) ** 3
        launch_cells = cells_per_step * updates_per_launch

    if rank != 0:
        return

    mean_s, sd_s = stats(elapsed_reps)
    mcells = cells_per_step * K / mean_s / 1e6
    rates = [cells_per_step * K / e / 1e6 for e in elapsed_reps]
    # roofline.achieved: ALGORITHMIC bytes per launch over the kernel's duration: 16 B per cell update of
    # the single-step traffic model (10 B with fp16 storage) times the cell updates one launch performs.
    # The two-step sweep performs two updates per cell and launch while it moves only five level-sized
    # streams for them (10 B per update, compulsory), so this figure is an equivalent rate, not the
    # bytes on the HBM pins; those are `traffic` / `physical_GBps` (counters) and `compulsory_GBps`.
    achieved = algo_bytes * launch_cells / (kernel_ms * 1e-3) / 1e9
    kernel_bytes_per_update = 10.0 if (updates_per_launch == 2 or (h16 and not slabs)) else 16.0
    compulsory = kernel_bytes_per_update * launch_cells
    roof = {
        "bound": "hbm",
        "kernel": ("stencil_march2_kernel: two-step sweep, two cell updates and both source scatters per launch"
                   if updates_per_launch == 2 else "stencil sweep (section0)"),
        "cell_updates_per_launch": launch_cells,
        "achieved": round(achieved, 1),
        "achieved_basis": f"{algo_bytes:g} B per cell update (algorithmic, single-step model) x cell updates per launch / kernel time",
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "frac_basis": "achieved / peak, with `achieved` as defined by achieved_basis (algorithmic bytes); the bytes that "
                      "actually cross the fabric are `traffic`, their rate `physical_GBps` and its share of the peak "
                      "`physical_frac`",
        "traffic": None,
        "avg_kernel_ms": round(kernel_ms, 4),
        "compulsory_bytes_per_launch": compulsory,
        "compulsory_GBps": round(compulsory / (kernel_ms * 1e-3) / 1e9, 1),
    }
    # HBM bytes per launch from the PMC counters: measured in separate profiler passes
    # (tools/profile_bench.sh) and only valid for the kernel sources and workload they were taken on.
    live = None
    if not slabs and not args.no_traffic and not h16 and args.space_order == 4 and args.variant == "auto":
        live = measure_traffic("stencil_march2" if updates_per_launch == 2 else "stencil_march_lds", K)
    if live is not None:
        roof["traffic"] = live["hbm_bytes_per_launch"]
        roof["traffic_source"] = live
        roof["physical_GBps"] = round(roof["traffic"] / (kernel_ms * 1e-3) / 1e9, 1)
        roof["physical_frac"] = round(roof["physical_GBps"] / HBM_PEAK_GBS, 4)
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if live is None and os.path.exists(tpath) and not slabs:
        try:
            tj = json.load(open(tpath))
            want = "stencil_march2" if updates_per_launch == 2 else "stencil_march_lds"
            same = (tj.get("kernel") == want and tj.get("library_source_hash") == library_source_hash()
                    and tj.get("grid") == n and tj.get("storage", "f32") == args.storage
                    and tj.get("variant", "auto") == args.variant)
            if same:
                roof["traffic"] = tj.get("hbm_bytes_per_launch")
                roof["traffic_source"] = {"file": "profiles/traffic_latest.json", "kernel": tj.get("kernel"),
                                          "library_source_hash": tj.get("library_source_hash"),
                                          "method": tj.get("note")}
                roof["physical_GBps"] = round(roof["traffic"] / (kernel_ms * 1e-3) / 1e9, 1)
                roof["physical_frac"] = round(roof["physical_GBps"] / HBM_PEAK_GBS, 4)
            else:
                roof["traffic_source"] = "profiles/traffic_latest.json was measured on other kernel sources or another workload"
        except Exception:
            pass
    if not slabs:
        so = args.space_order
        wl = (f"{n}^3 fp32 acoustic FDTD, {so}th-order star + leapfrog + 1 Ricker source, "
              f"{total_steps} steps ({W} untimed)"
              + (", five velocity layers 1.5 .. 4.5" if args.velocity == "layered" else ""))
    else:
        NX = int(cells_per_step / n / n)
        wl = (f"{NX}x{n}x{n} fp32 as {world} x-slabs, "
              + ("two steps per sweep and exchange (4 ghost planes of u(t+2) per side)" if updates_per_launch == 2
                 else "2-plane ghost exchange per step")
              + (" over RCCL" if os.environ.get("FDTD_BENCH_BACKEND", "nccl") == "nccl"
                 else " over gloo (host-staged rehearsal)")
              + f", {total_steps} steps ({W} untimed)")
    out = {
        "metric": "Mcells/s",
        "value": round(mcells, 1),
        "unit": "Mcells/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(mean_s / K * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong" if (world > 1 and global_x) else "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.storage == "f32" else "f32 arithmetic, f16 storage",
        "data": "synthetic",
        "effective_GBps_at_algorithmic_bytes": round(mcells * algo_bytes / 1e3, 1),
        "reps": {
            "n": R,
            "protocol": "each repetition: zeroed field, warm-up steps untimed, then the timed steps (main.cpp:388-399)",
            "ms_per_step_mean": round(mean_s / K * 1e3, 4),
            "ms_per_step_stddev": round(sd_s / K * 1e3, 4),
            "Mcells_per_s": [round(r, 1) for r in rates],
            "Mcells_per_s_stddev": round(stats(rates)[1], 1),
        },
        "config": {
            "workload": wl,
            "variant": args.variant,
            "algorithmic_bytes_per_cell_update": algo_bytes,
            "kernel_bytes_per_cell_update": kernel_bytes_per_update,
            "ranks": ranks_seen,
        },
        "roofline": roof,
    }
    if exchange is not None:
        out["exchange"] = exchange
    if slabs and world == 1:
        out["config"]["workload"] += " [one rank, own neighbour on both sides: schedule rehearsal]"
    if slabs and world > 1 and not h16:
        prop = stepper = tmax = None      # this rank's arrays go before another process allocates its slabs on the device
        try:
            peer_leg(args, out, world)
        except Exception as e:            # the second leg is extra evidence: it never takes the ranks' line with it
            out["one_process_peer_copies"] = {"failed": f"{type(e).__name__}: {e}"}
    if world == 1 and not slabs and not args.no_cpu_baseline and args.space_order == 4:
        out["cpu_baseline"] = cpu_baseline(pkg, n, args.cpu_steps)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
