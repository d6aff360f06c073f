#!/usr/bin/env python3
"""Rate of the damped sweep of the forward operator (fdtd_hip_stencil_step_damped) on an n^3 box:
the marching sweep and, with FDTD_ORDER_MARCH=0, the one-cell-per-thread kernel.

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
    args = ap.parse_args()
    import torch
    pkg = load_package()
    lib = pkg.abi.lib()
    n = args.grid
    shape = (n + 8,) * 3
    u = torch.zeros((3,) + shape, dtype=torch.float32, device="cuda")
    u[0].normal_()
    u[2].copy_(u[0])
    m = torch.full(shape, 1.5, dtype=torch.float32, device="cuda")
    damp = torch.rand(shape, dtype=torch.float32, device="cuda")
    g = pkg.Geom(*shape, 0, n - 1, 0, n - 1, 0, n - 1, 1e-3, 10.0, 10.0, 10.0, 0.0, 0.0, 0.0)
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
