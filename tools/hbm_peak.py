#!/usr/bin/env python3
"""What this GPU's HBM delivers to plain streaming kernels, as context for the roofline fractions:
copy (1 read + 1 write stream), add (2 + 1), the stream mix of a single-step sweep (3 reads + 1 write,
torch.addcmul) and of the two-step sweep (3 reads + 2 writes, two launches of 1.5 + 1), on 1 GiB arrays.

    python tools/hbm_peak.py
"""
import torch


def rate(fn, nbytes, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12


def main():
    n = 1 << 28   # 1 GiB of fp32 per array
    a, b, c, d = (torch.rand(n, device="cuda") for _ in range(4))
    gib = n * 4
    print(f"copy      (1R + 1W): {rate(lambda: d.copy_(a), 2 * gib):5.2f} TB/s")
    print(f"add       (2R + 1W): {rate(lambda: torch.add(a, b, out=d), 3 * gib):5.2f} TB/s")
    print(f"addcmul   (3R + 1W): {rate(lambda: torch.addcmul(a, b, c, out=d), 4 * gib):5.2f} TB/s")
    print(f"fill      (1W)     : {rate(lambda: d.fill_(1.0), gib):5.2f} TB/s")
    print(f"sum       (1R)     : {rate(lambda: a.sum(), gib):5.2f} TB/s")


if __name__ == "__main__":
    main()
