#!/usr/bin/env python3
"""Config 5 at its full size and step count on one GPU: the 2048^3 grid, 200 steps, as 8 slabs (the 8-GPU schedule with
every push a device-to-device copy) and undivided; the 64-bit word sums of the three newest levels and max |u| must be
the same.  A third run doubles the source: the update is linear and a factor of two is exact in binary32, so max |u|
must double exactly.  Nothing of the field leaves the device.

    python tools/config5_checksums.py [--grid 2048] [--steps 200] [--slabs 8]
"""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--slabs", type=int, default=8)
    args = ap.parse_args()
    pkg = load_package()
    n = args.grid
    g = pkg.Geom(n + 8, n + 8, n + 8, 0, n - 1, 0, n - 1, 0, n - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
    out = []
    for slabs in (args.slabs, 1):
        with pkg.MultiGpuSession(g, [0] * slabs) as s:
            s.fill_standard_inputs(args.steps + 1, 1)
            s.advance(0, 4)
            t0 = time.perf_counter()
            s.advance(5, args.steps - 1)
            dt = time.perf_counter() - t0
            sums, (mx, bad) = s.checksum(), s.max_abs()
        rate = n ** 3 * (args.steps - 5) / dt / 1e9
        print(f"{n}^3, {args.steps} steps, {slabs} slab(s) on one GPU: {dt / (args.steps - 5) * 1e3:7.3f} ms/step {rate:6.1f} Gcells/s  "
              f"max|u| {mx:.9e}  non-finite {bad}  word sums {[hex(v) for v in sums]}", flush=True)
        out.append((sums, mx, bad))
    same = out[0] == out[1]
    print("same field on both decompositions:", same)
    # linearity: twice the source, twice the field (exact: every operation of the update commutes with a factor of 2)
    with pkg.MultiGpuSession(g, [0] * args.slabs) as s:
        s.fill_standard_inputs(args.steps + 1, 1)
        src = pkg.fill_ricker(args.steps + 1, 1) * 2.0
        s.set_sources(src.astype("float32"), pkg.fill_source_coords(1, n, n, n))
        s.advance(0, args.steps - 1)
        mx2, bad2 = s.max_abs()
    linear = bad2 == 0 and mx2 == 2.0 * out[0][1]
    print(f"source doubled: max|u| {mx2:.9e} = 2 x {out[0][1]:.9e}: {linear}")
    sys.exit(0 if same and linear and out[0][2] == 0 and out[0][1] > 0 else 1)


if __name__ == "__main__":
    main()
