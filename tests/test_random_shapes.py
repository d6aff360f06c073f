"""Randomised differential test: the three operators against the CPU oracle on boxes nobody picked by hand.

Shapes, sub-ranges, grid spacings, velocity fields, source counts and the first time index are drawn from a
seeded generator; the size thresholds of the automatic choice are lowered so that the small boxes reach the
kernels that large grids take (marching, two steps per sweep with masked tiles, packed rows).  Bit for bit,
as everywhere for fp32 (up to 128 sources the scatter is evaluated in the reference's order)."""
import os

import numpy as np
import pytest

import oracle_binding as ob

VARIANTS = ["auto", "march2", "march2-128", "march2-mixed", "march", "rows", "vec", "naive"]


def draw_case(seed):
    rng = np.random.default_rng(1000 + seed)
    zs = [int(rng.integers(5, 40)), 4 * int(rng.integers(2, 20)), 4 * int(rng.integers(30, 70)), 256 + 4 * int(rng.integers(0, 12))]
    n = (int(rng.integers(5, 40)), int(rng.integers(5, 44)), zs[int(rng.integers(0, 4))])
    shape = tuple(v + 8 for v in n)
    u = np.zeros((3,) + shape, dtype=np.float32)
    u[0] = rng.uniform(-1, 1, size=shape).astype(np.float32)
    u[1] = rng.uniform(-1, 1, size=shape).astype(np.float32)      # halos differ per level: they follow the ring
    u[2] = u[0] + rng.uniform(-0.01, 0.01, size=shape).astype(np.float32)
    m = rng.uniform(0.5, 3, size=shape).astype(np.float32)
    h = tuple(float(v) for v in rng.uniform(0.5, 2.0, size=3).astype(np.float32))
    if rng.integers(0, 3) == 0:          # a sub-range (z range a multiple of 4 cells half of the time)
        lo = [int(rng.integers(0, max(1, v // 3))) for v in n]
        hi = [int(rng.integers(max(l, v - 1 - v // 3), v)) for l, v in zip(lo, n)]
        if rng.integers(0, 2) == 0:
            lo[2] -= lo[2] % 4
            hi[2] = max(lo[2] + 3, hi[2] - (hi[2] - lo[2] + 1) % 4)
        bounds = (lo[0], hi[0], lo[1], hi[1], lo[2], min(hi[2], n[2] - 1))
    else:
        bounds = (0, n[0] - 1, 0, n[1] - 1, 0, n[2] - 1)
    nsrc = [0, 1, 3, 9, 20][int(rng.integers(0, 5))]
    time_m = int(rng.integers(0, 4)) if rng.integers(0, 4) == 1 else 0
    nt = time_m + int(rng.integers(6, 15))
    src = coords = None
    if nsrc:
        src = rng.uniform(-50, 50, size=(nt, nsrc)).astype(np.float32)
        ext = (np.array(n) - 1) * np.array(h)
        coords = (rng.uniform(-0.02, 1.02, size=(nsrc, 3)) * ext).astype(np.float32)   # some corners fall outside
        coords[0] = ((n[0] - 1) * h[0] * 0.999, 0.0, (n[2] - 1) * h[2] * 0.5)
    kw = dict(h=h, dt=float(rng.uniform(5e-4, 2e-3)), bounds=bounds, time_m=time_m, time_M=nt - 1,
              o=tuple(float(v) for v in rng.uniform(-1, 1, size=3).astype(np.float32)))
    if nsrc:
        coords += np.array(kw["o"], dtype=np.float32)
    return u, m, src, coords, kw, VARIANTS[int(rng.integers(0, len(VARIANTS)))]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("FDTD_FUZZ_SEEDS", "48"))))   # more seeds on request
def test_operators_on_random_boxes(pkg, hiplib, monkeypatch, seed):
    u0, m, src, coords, kw, variant = draw_case(seed)
    for k in ("FDTD_MARCH_MIN_CELLS", "FDTD_FUSE2_MIN_CELLS", "FDTD_FUSE2_MIN_FILL"):
        monkeypatch.setenv(k, "0")
    monkeypatch.setenv("FDTD_FUSE2_MIN_PLANES", "1")
    if variant.startswith("march2"):
        monkeypatch.setenv("FDTD_VARIANT", "march2")
        if variant.endswith("128"):
            monkeypatch.setenv("FDTD_FUSE2_TILE", "128")
        elif variant.endswith("mixed"):
            monkeypatch.setenv("FDTD_FUSE2_MIXED", "2")
    elif variant != "auto":
        monkeypatch.setenv("FDTD_VARIANT", variant)
    u_ref = u0.copy()
    ob.run_oracle(m, src, coords, u_ref, **kw)
    assert np.isfinite(u_ref).all()
    for op in ("Kernel_OpenACC", "Kernel_CUDA", "Kernel_CUDA_Optimized"):
        u = u0.copy()
        rc, _ = pkg.run_operator(getattr(hiplib, op), m, src, coords, u, **kw)
        assert rc == 0, (seed, op)
        assert ob.bits_equal(u, u_ref), (seed, variant, op, u0.shape, kw["bounds"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("FDTD_FUZZ_SEEDS", "24"))))
def test_multi_device_operator_on_random_boxes(pkg, hiplib, monkeypatch, seed):
    """Kernel_HIP_MultiGPU with two to four slabs on the one device: whole-x boxes of 16 to 70 planes, sources
    anywhere (on the cuts between slabs too), both forms of the boundary zones."""
    rng = np.random.default_rng(5000 + seed)
    nslab = int(rng.integers(2, 5))
    monkeypatch.setenv("FDTD_MULTIGPU_DEVICES", ",".join(["0"] * nslab))
    monkeypatch.setenv("FDTD_SLAB_ZONE_FUSED", str(int(rng.integers(0, 2))))
    for k in ("FDTD_MARCH_MIN_CELLS", "FDTD_FUSE2_MIN_CELLS", "FDTD_FUSE2_MIN_FILL"):
        monkeypatch.setenv(k, "0")
    nz = [4 * int(rng.integers(3, 40)), 256, 512, int(rng.integers(9, 50))][int(rng.integers(0, 4))]
    n = (int(rng.integers(16, 71)), int(rng.integers(5, 40)), nz)
    shape = tuple(v + 8 for v in n)
    u0 = np.zeros((3,) + shape, dtype=np.float32)
    u0[0] = rng.uniform(-1, 1, size=shape).astype(np.float32)
    u0[1] = rng.uniform(-1, 1, size=shape).astype(np.float32)
    u0[2] = u0[0] + rng.uniform(-0.01, 0.01, size=shape).astype(np.float32)
    m = rng.uniform(0.5, 3, size=shape).astype(np.float32)
    h = (0.8, 1.0, 1.25)
    nsrc = [0, 1, 4, 11][int(rng.integers(0, 4))]
    time_m = int(rng.integers(0, 3))
    nt = time_m + int(rng.integers(5, 14))
    src = coords = None
    if nsrc:
        src = rng.uniform(-50, 50, size=(nt, nsrc)).astype(np.float32)
        ext = (np.array(n) - 1) * np.array(h)
        coords = (rng.uniform(0.0, 1.0, size=(nsrc, 3)) * ext).astype(np.float32)
        coords[0, 0] = (n[0] // nslab - 0.5) * h[0]          # corners on either side of the first cut
    b = (0, n[0] - 1, int(rng.integers(0, 3)), n[1] - 1 - int(rng.integers(0, 3)), 0, n[2] - 1)
    kw = dict(h=h, dt=1e-3, bounds=b, time_m=time_m, time_M=nt - 1)
    u_ref = u0.copy()
    ob.run_oracle(m, src, coords, u_ref, **kw)
    u = u0.copy()
    rc, _ = pkg.run_operator(hiplib.Kernel_HIP_MultiGPU, m, src, coords, u, deviceid=-1, **kw)
    assert rc == 0
    assert ob.bits_equal(u, u_ref), (seed, nslab, n, b, nsrc)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("FDTD_FUZZ_SEEDS", "30"))))
def test_space_orders_and_forward_operator_on_random_boxes(pkg, hiplib, monkeypatch, seed):
    """Orders 4 to 12 with and without sponge and receivers, through the marching sweep (forced, short x-chunks)
    or the one-cell-per-thread sweep, on random boxes and sub-ranges, against the generic restatements."""
    rng = np.random.default_rng(9000 + seed)
    order = int(rng.choice([4, 6, 8, 10, 12]))
    march = bool(rng.integers(0, 2))
    forward = bool(rng.integers(0, 2))
    monkeypatch.setenv("FDTD_ORDER_MARCH", "2" if march else "0")
    monkeypatch.setenv("FDTD_ORDER_XCHUNK", str(int(rng.integers(4, 30))))
    if rng.integers(0, 2):
        monkeypatch.setenv("FDTD_ORDER_LPR", "32")
    H = order
    n = (int(rng.integers(5, 40)), int(rng.integers(5, 44)), [int(rng.integers(5, 60)), int(rng.integers(200, 330))][int(rng.integers(0, 2))])
    shape = tuple(v + 2 * H for v in n)
    u0 = np.zeros((3,) + shape, dtype=np.float32)
    u0[0] = rng.uniform(-1, 1, size=shape).astype(np.float32)
    u0[1] = rng.uniform(-1, 1, size=shape).astype(np.float32)
    u0[2] = u0[0] + rng.uniform(-0.01, 0.01, size=shape).astype(np.float32)
    m = rng.uniform(0.5, 3, size=shape).astype(np.float32)
    damp = rng.uniform(0, 40, size=shape).astype(np.float32)
    h = tuple(float(v) for v in rng.uniform(0.5, 2.0, size=3).astype(np.float32))
    if rng.integers(0, 3) == 0:
        lo = [int(rng.integers(0, max(1, v // 3))) for v in n]
        hi = [int(rng.integers(max(l, v - 1 - v // 3), v)) for l, v in zip(lo, n)]
        b = (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])
    else:
        b = (0, n[0] - 1, 0, n[1] - 1, 0, n[2] - 1)
    nsrc, nrec = int(rng.integers(0, 4)), int(rng.integers(1, 6))
    nt = int(rng.integers(6, 11))
    ext = (np.array(n) - 1) * np.array(h)
    src = coords = None
    if nsrc:
        src = rng.uniform(-50, 50, size=(nt, nsrc)).astype(np.float32)
        coords = (rng.uniform(-0.02, 1.02, size=(nsrc, 3)) * ext).astype(np.float32)
    rec = np.full((nt, nrec), np.nan, dtype=np.float32)
    rcoords = (rng.uniform(-0.02, 1.02, size=(nrec, 3)) * ext).astype(np.float32)
    kw = dict(h=h, dt=1e-3, bounds=b, time_M=nt - 1)
    assert hiplib.fdtd_hip_set_space_order(order) == 0
    try:
        if forward:
            u_ref, rec_ref = u0.copy(), rec.copy()
            assert pkg.run_forward(ob.forward_oracle_order(order), damp, m, rec_ref, rcoords, src, coords, u_ref, **kw)[0] == 0
            u = u0.copy()
            assert pkg.run_forward(hiplib.Kernel_HIP_Forward, damp, m, rec, rcoords, src, coords, u, **kw)[0] == 0
            assert ob.bits_equal(rec, rec_ref), (seed, order, march)
        else:
            u_ref = u0.copy()
            ob.run_oracle_order(order, m, src, coords, u_ref, **kw)
            u = u0.copy()
            assert pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u, **kw)[0] == 0
    finally:
        assert hiplib.fdtd_hip_set_space_order(4) == 0
    assert ob.bits_equal(u, u_ref), (seed, order, march, forward, n, b)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("FDTD_FUZZ_SEEDS", "16"))))
def test_fp16_storage_on_random_boxes(pkg, hiplib, monkeypatch, seed):
    """Kernel_HIP_FP16Storage (binary16 wavefield in memory, fp32 arithmetic) on random whole boxes, through the
    marching kernel or the one-cell-per-thread kernel, bit for bit against the oracle stepped with a
    round-to-binary16 after every store (inputs pre-rounded so that the upload is lossless)."""
    rng = np.random.default_rng(13000 + seed)
    monkeypatch.setenv("FDTD_MARCH_MIN_CELLS", "0" if rng.integers(0, 2) else "2000000000")
    nz = [int(rng.integers(5, 70)), 8 * int(rng.integers(4, 40)), 4 * int(rng.integers(30, 80))][int(rng.integers(0, 3))]
    n = (int(rng.integers(5, 30)), int(rng.integers(5, 40)), nz)
    shape = tuple(v + 8 for v in n)
    u = np.zeros((3,) + shape, dtype=np.float32)
    u[0] = rng.uniform(-1, 1, size=shape)
    u[2] = u[0] + rng.uniform(-0.01, 0.01, size=shape)
    u = u.astype(np.float16).astype(np.float32)
    m = rng.uniform(0.5, 3, size=shape).astype(np.float32)
    h = tuple(float(v) for v in rng.uniform(0.5, 2.0, size=3).astype(np.float32))
    nsrc = [0, 1, 5][int(rng.integers(0, 3))]
    nt = int(rng.integers(6, 12))
    src = coords = None
    if nsrc:
        src = rng.uniform(-50, 50, size=(nt, nsrc)).astype(np.float32)
        coords = (rng.uniform(0.0, 1.0, size=(nsrc, 3)) * (np.array(n) - 1) * np.array(h)).astype(np.float32)
    ue = u.copy()
    bounds = (0, n[0] - 1, 0, n[1] - 1, 0, n[2] - 1)
    hh = tuple(np.float32(v) for v in h)
    for t in range(nt):
        t0, t1, t2 = t % 3, (t + 2) % 3, (t + 1) % 3
        ob.oracle_stencil_step(m, ue, bounds, np.float32(1e-3), hh, t0, t1, t2)
        ue[t2] = ue[t2].astype(np.float16).astype(np.float32)
        if nsrc:
            ob.oracle_source_inject(m, src, coords, ue, bounds, hh, (0.0, 0.0, 0.0), (0, nsrc - 1), t, t2)
            ue[t2] = ue[t2].astype(np.float16).astype(np.float32)
    rc, _ = pkg.run_operator(hiplib.Kernel_HIP_FP16Storage, m, src, coords, u, h=h, time_M=nt - 1)
    assert rc == 0
    assert ob.bits_equal(u, ue), (seed, n, nsrc)
