#!/usr/bin/env python3
"""Benchmark of the FDTD hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid n] [--variant v]

One "step" is one time step of the propagator (section0 stencil sweep + section1 source
scatter, reference openacc.cpp:148-205) on fields that are already resident in HBM.
N = 1 runs BASELINE.json's headline configuration: 512^3 fp32, one Ricker source, the
LDS-free x-marching kernel (Kernel_CUDA_Optimized's path).  With the defaults W = 5 and
K = 45 the timed window is steps 5..49 of a 50-step run, i.e. exactly the reference's timer
window (openacc.cpp:5,148).  N > 1 (launched by torch.distributed.run, one rank per GPU)
runs the slab-decomposed propagator: every rank owns a 512-plane slab of a (512*N) x 512 x 512
grid and exchanges 2 ghost planes per side and step with its neighbours over RCCL (weak
scaling).  Rank 0 prints one JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The slab schedule keeps three device queues busy at once (interior chain, boundary chain, RCCL).
# With HIP's default of 4 hardware queues per process the boundary stream was seen to land on the
# interior stream's queue, which serialises the two chains; 8 keeps them apart.  Must be in the
# environment before the HIP runtime initialises, i.e. before torch is imported.
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("FDTD_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
ALGO_BYTES_PER_CELL = 16.0     # read u[t0], u[t1], m; write u[t2]; 4 B each
ALGO_BYTES_PER_CELL_F16 = 10.0  # fp16 storage: u levels 2 B each, m stays 4 B


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=45)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=512, help="cells per axis (per-rank slab thickness for N>1)")
    ap.add_argument("--variant", default="auto",
                    choices=["auto", "flat", "naive", "march", "tile3d", "vec", "march2"],
                    help="auto: two steps per sweep where the shape allows (512-wide z), else the marching kernel")
    ap.add_argument("--storage", default="f32", choices=["f32", "f16"],
                    help="f16: wavefield levels stored as binary16 in HBM, fp32 arithmetic (config 4; N=1 only)")
    ap.add_argument("--global-x", type=int, default=0,
                    help="N>1: total planes along x split over the ranks (strong scaling), instead of --grid planes per rank")
    ap.add_argument("--self-exchange", action="store_true",
                    help="N=1 rehearsal of the N>1 schedule: the rank is its own two neighbours, RCCL messages "
                         "included (measures what the exchange costs next to the interior sweep; not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=50, help="oracle steps for the CPU baseline sample")
    return ap.parse_args()


def cpu_baseline(pkg, n, steps):
    """Time the CPU oracle (test infrastructure; never on the product path) on a bounded sample
    of the same workload: n^3, `steps` steps of the Ricker run, its own section timers."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    u, m, src, coords = pkg.standard_inputs(n, n, n, steps, 1)
    prof = ob.run_oracle(m, src, coords, u)
    timed = steps - pkg.WARMUP_STEPS
    secs = prof.section0 + prof.section1
    return {
        "value": round(n ** 3 * timed / secs / 1e6, 1),
        "unit": "Mcells/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n}^3 fp32 Ricker run, {steps} steps, steps {pkg.WARMUP_STEPS}..{steps - 1} timed "
                  f"({secs:.2f} s), OpenMP over x,y; bit-identical restatement of openacc.cpp:159-196",
    }


def main():
    args = parse_args()
    import numpy as np
    import torch
    from __graft_entry__ import load_package

    pkg = load_package()
    L = pkg.lib()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 through: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    # one GPU per rank; the modulo only matters when the N>1 path is rehearsed on a 1-GPU box
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    n = args.grid
    K, W = args.steps, args.warmup
    total_steps = K + W
    variant = pkg.VARIANTS[args.variant]

    slabs = world > 1 or args.self_exchange
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
        slab = pkg.slab
        # two steps per exchange (fused sweep, 4-plane ghosts) where the slab's cross-section allows it
        gq = pkg.Geom(n + 8, n + 8, n + 8, 0, n - 1, 0, n - 1, 0, n - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
        pairs = args.storage == "f32" and bool(L.fdtd_hip_two_steps_fuses(C.byref(gq), 1, variant))
        maker = slab.standard_pair_benchmark if pairs else slab.SlabPropagator.standard_benchmark
        prop = maker(n_per_rank=n, ny=n, nz=n, nt=total_steps + 1, rank=rank, world=world, device=dev,
                     variant=variant, global_x=args.global_x, self_exchange=args.self_exchange and world == 1)

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

        advance(0, W)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
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
            print(f"[rank {rank}] issue {t_issue*1e3:.3f} ms, finished {t_fin*1e3:.3f} ms, "
                  f"after barrier {elapsed*1e3:.3f} ms", file=sys.stderr)
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        kernel_ms = prop.mean_interior_kernel_ms()
        cells_per_step = float(prop.layout.NX) * n * n
        updates_per_launch = 2 if pairs else 1
        launch_cells = prop.interior_cells() * updates_per_launch
        dist.destroy_process_group()
    else:
        u, m, src, coords = pkg.standard_inputs(n, n, n, total_steps, 1)
        u_d = torch.from_numpy(u).to(dev)
        m_d = torch.from_numpy(m).to(dev)
        s_d = torch.from_numpy(src).to(dev)
        c_d = torch.from_numpy(coords).to(dev)
        del u, m
        npad = n + 2 * pkg.HALO
        g = pkg.Geom(npad, npad, npad, 0, n - 1, 0, n - 1, 0, n - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
        stream = torch.cuda.current_stream()
        sh = C.c_void_p(stream.cuda_stream)
        h16 = args.storage == "f16"
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
        fused = (not h16) and K >= 2 and bool(L.fdtd_hip_two_steps_fuses(C.byref(g), 1, variant))
        updates_per_launch = 1
        if fused:
            # Two steps per sweep: four level buffers rotate (the three ring slots and one more).  All
            # halos are zero and the source sits inside the grid, so no shell has to follow the ring.
            updates_per_launch = 2
            extra = torch.zeros_like(u_d[0])
            P = [u_d[0], u_d[1], u_d[2], extra]
            cur, prev, third, spare = 0, 2, 1, 3          # time 0: t0 = 0, t1 = 2, t2 = 1
            margs = (m_d.data_ptr(), s_d.data_ptr(), c_d.data_ptr())
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
            n_pairs = K // 2
            # Every two-step call hands the library two events.  Where the call is a single sweep launch
            # (512-wide rows) they receive the start and stop times of that dispatch itself, which is the
            # kernel duration a profiler reports; otherwise they are recorded on the stream around all the
            # launches of the call (sweep launches, seam pass, scatter).
            stride = 1
            ev = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for i in range(0, n_pairs, stride)}
            for a, b in ev.values():  # torch creates the HIP event at the first record; the library re-records it
                a.record(stream)
                b.record(stream)
            evh = {i: (C.c_void_p(a.cuda_event), C.c_void_p(b.cuda_event)) for i, (a, b) in ev.items()}
            torch.cuda.synchronize()
            t_start = time.perf_counter()
            t = W
            for i in range(n_pairs):
                # The source is inside the grid (flag 1: no halo-corner launches).
                h = evh.get(i, (None, None)) if os.environ.get("FDTD_BENCH_EVENTS", "1") != "0" else (None, None)
                rc = L.fdtd_hip_two_steps_ex(C.byref(g), *margs, P[cur].data_ptr(), P[prev].data_ptr(),
                                             P[third].data_ptr(), P[spare].data_ptr(), t, 0, 0, 1, 3, variant, 1,
                                             h[0], h[1], sh)
                assert rc == 0, rc
                cur, prev, third, spare = spare, third, cur, prev
                t += 2
            if t < total_steps:
                assert L.fdtd_hip_one_step(C.byref(g), *margs, P[cur].data_ptr(), P[prev].data_ptr(),
                                           P[third].data_ptr(), t, 0, 0, 1, 3, variant, sh) == 0
                cur, prev, third = third, cur, prev
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t_start
            if os.environ.get("FDTD_BENCH_EVENTS", "1") != "0":
                kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev.values()]))
            else:   # experiment: no events at all, the launch period stands in for the kernel time
                kernel_ms = elapsed / max(1, n_pairs) * 1e3
            u_d = torch.stack([P[cur], P[prev], P[third]])
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
            kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        cells_per_step = float(n) ** 3
        launch_cells = cells_per_step * updates_per_launch
        max_u = float(u_d.float().abs().max().item())
        assert np.isfinite(max_u) and max_u > 0.0, "field is empty or not finite"

    if rank != 0:
        return

    mcells = cells_per_step * K / elapsed / 1e6
    if not slabs and args.storage == "f16":
        globals()["ALGO_BYTES_PER_CELL"] = ALGO_BYTES_PER_CELL_F16
    achieved = ALGO_BYTES_PER_CELL * launch_cells / (kernel_ms * 1e-3) / 1e9  # GB/s, dominant kernel
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath) and not slabs and n % 512 == 0 and args.storage == "f32" and args.variant == "auto":
        try:
            tj = json.load(open(tpath))
            want = "stencil_march2" if updates_per_launch == 2 else "stencil_march_lds"
            traffic = tj.get("hbm_bytes_per_launch") if tj.get("kernel", "stencil_march_lds") == want else None
        except Exception:
            traffic = None
    out = {
        "metric": "Mcells/s",
        "value": round(mcells, 1),
        "unit": "Mcells/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong" if (world > 1 and args.global_x) else "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.storage == "f32" else "f32 arithmetic, f16 storage",
        "data": "synthetic",
        "achieved_hbm_GBps_whole_step": round(mcells * ALGO_BYTES_PER_CELL / 1e3, 1),
        "config": {
            "workload": (f"{n}^3 fp32 acoustic FDTD, 4th-order star + leapfrog + 1 Ricker source, "
                         f"{total_steps} steps ({W} untimed)" if not slabs else
                         f"{int(cells_per_step / n / n)}x{n}x{n} fp32 as {world} x-slabs, "
                         + ("two steps per sweep and exchange (4 ghost planes of u(t+2) per side)" if updates_per_launch == 2
                            else "2-plane ghost exchange per step")
                         + (" over RCCL" if os.environ.get("FDTD_BENCH_BACKEND", "nccl") == "nccl"
                            else " over gloo (host-staged rehearsal)")
                         + f", {total_steps} steps ({W} untimed)"),
            "variant": args.variant,
            "bytes_per_cell_update": ALGO_BYTES_PER_CELL,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": ("stencil_march2_kernel: two-step sweep, two cell updates and both source scatters per launch"
                       if updates_per_launch == 2 else "stencil sweep (section0)"),
            "cell_updates_per_launch": launch_cells,
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "avg_kernel_ms": round(kernel_ms, 4),
        },
    }
    if slabs and world == 1:
        out["config"]["workload"] += " [one rank, own neighbour on both sides: schedule rehearsal]"
    if world == 1 and not slabs and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, n, args.cpu_steps)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
