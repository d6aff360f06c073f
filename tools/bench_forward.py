#!/usr/bin/env python3
"""Rate of the damped update of the forward operator on an n^3 box: two steps per sweep
(fdtd_hip_two_steps_damped), the single-step marching sweep (fdtd_hip_stencil_step_damped) and, with
FDTD_ORDER_MARCH=0, the one-cell-per-thread kernel.

    python tools/bench_forward.py [--grid 512] [--steps 20]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--inputs", default="benchmark", choices=["benchmark", "random"],
                    help="benchmark: zero field, m = 1.5, one Ricker source, a 40-cell quadratic sponge on every face "
                         "(what a propagator runs); random: random field and damping everywhere (kernel time depends on "
                         "the data through power and clock)")
    args = ap.parse_args()
    import torch
    pkg = load_package()
    lib = pkg.abi.lib()
    n = args.grid
    shape = (n + 8,) * 3
    u = torch.zeros((3,) + shape, dtype=torch.float32, device="cuda")
    m = torch.full(shape, 1.5, dtype=torch.float32, device="cuda")
    nt = args.steps + 16
    src = torch.from_numpy(pkg.fill_ricker(nt, 1)).cuda()
    coords = torch.from_numpy(pkg.fill_source_coords(1, n, n, n)).cuda()
    if args.inputs == "random":
        u[0].normal_()
        u[2].copy_(u[0])
        damp = torch.rand(shape, dtype=torch.float32, device="cuda")
        g = pkg.Geom(*shape, 0, n - 1, 0, n - 1, 0, n - 1, 1e-3, 10.0, 10.0, 10.0, 0.0, 0.0, 0.0)
    else:
        i = torch.arange(n, device="cuda")
        dist = torch.minimum(i, n - 1 - i).float()
        ramp = torch.where(dist < 40, ((40 - dist) / 40) ** 2, torch.zeros_like(dist)) * 40.0
        full = torch.zeros(n + 8, device="cuda")
        full[4:-4] = ramp
        damp = torch.maximum(torch.maximum(full[:, None, None], full[None, :, None]), full[None, None, :]).contiguous()
        g = pkg.Geom(*shape, 0, n - 1, 0, n - 1, 0, n - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
    # two steps per sweep: four level buffers rotate
    extra = torch.zeros(shape, dtype=torch.float32, device="cuda")
    P = [u[0], u[1], u[2], extra]
    cur, prev, third, spare = 0, 2, 1, 3

    def pair(t):
        nonlocal cur, prev, third, spare
        rc = lib.fdtd_hip_two_steps_damped(C.byref(g), m.data_ptr(), damp.data_ptr(), src.data_ptr(), coords.data_ptr(),
                                           P[cur].data_ptr(), P[prev].data_ptr(), P[third].data_ptr(), P[spare].data_ptr(),
                                           t, 0, 0, 1, 3, pkg.VARIANTS["auto"], 1, None)
        assert rc == 0, rc
        cur, prev, third, spare = spare, third, cur, prev
    for t in range(0, 6, 2):
        pair(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    npairs = max(1, args.steps // 2)
    e0.record()
    for i in range(npairs):
        pair(6 + 2 * i)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (2 * npairs)
    gcells = n ** 3 / ms / 1e6
    print(f"inputs: {args.inputs}")
    print(f"damped sweep  grid {n}^3  two steps per sweep  {ms:8.3f} ms/step  {gcells:7.1f} Gcells/s  "
          f"{gcells * 20 / 1000:6.2f} TB/s at 20 B/cell (the single-step model), {gcells * 12 / 1000:6.2f} TB/s at the "
          f"12 B/cell the sweep moves", flush=True)
    for mode in ("1", "0"):
        os.environ["FDTD_ORDER_MARCH"] = mode

        def sweep(t):
            rc = lib.fdtd_hip_stencil_step_damped(C.byref(g), m.data_ptr(), damp.data_ptr(), u.data_ptr(), t % 3, (t + 2) % 3,
                                                  (t + 1) % 3, None)
            assert rc == 0, rc
        for t in range(3):
            sweep(t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(args.steps):
            sweep(t)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        gcells = n ** 3 / ms / 1e6
        print(f"damped sweep  grid {n}^3  {'march' if mode != '0' else 'cell '}  {ms:8.3f} ms/sweep  {gcells:7.1f} Gcells/s  "
              f"{gcells * 20 / 1000:6.2f} TB/s at 20 B/cell", flush=True)


if __name__ == "__main__":
    main()
