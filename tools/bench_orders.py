#!/usr/bin/env python3
"""Sweep rate of the space orders 4..12 (fdtd_hip_stencil_step_order): the marching sweep and the
one-cell-per-thread sweep, HIP events around `steps` sweeps of an n^3 box padded by the order.

    python tools/bench_orders.py [--grid 512] [--steps 20] [--orders 4,6,8,10,12]
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
    ap.add_argument("--shape", default="", help="nx,ny,nz instead of a cube")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--orders", default="4,6,8,10,12")
    ap.add_argument("--modes", default="1,0", help="FDTD_ORDER_MARCH values to time")
    args = ap.parse_args()
    import torch
    pkg = load_package()
    lib = pkg.abi.lib()
    n = args.grid
    dims = tuple(int(v) for v in args.shape.split(",")) if args.shape else (n, n, n)
    for order in [int(v) for v in args.orders.split(",")]:
        H = order
        shape = tuple(v + 2 * H for v in dims)
        u = torch.zeros((3,) + shape, dtype=torch.float32, device="cuda")
        u[0].normal_()
        u[2].copy_(u[0])
        m = torch.full(shape, 1.5, dtype=torch.float32, device="cuda")
        g = pkg.Geom(*shape, 0, dims[0] - 1, 0, dims[1] - 1, 0, dims[2] - 1, 1e-3, 10.0, 10.0, 10.0, 0.0, 0.0, 0.0)
        for mode in args.modes.split(","):
            os.environ["FDTD_ORDER_MARCH"] = mode
            def sweep(t):
                rc = lib.fdtd_hip_stencil_step_order(C.byref(g), m.data_ptr(), u.data_ptr(), t % 3, (t + 2) % 3, (t + 1) % 3,
                                                     order, None)
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
            gcells = dims[0] * dims[1] * dims[2] / ms / 1e6
            print(f"order {order:2d}  grid {dims[0]}x{dims[1]}x{dims[2]}  {'march' if mode != '0' else 'cell '}  {ms:8.3f} ms/sweep  {gcells:7.1f} Gcells/s  "
                  f"{gcells * 16 / 1000:6.2f} TB/s at 16 B/cell", flush=True)
        del u, m


if __name__ == "__main__":
    main()
