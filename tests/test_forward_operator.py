"""Kernel_HIP_Forward: the update with an absorbing sponge, the source scatter and receivers.

PARITY UNPINNED: the reference has no sponge and no receivers, so nothing of its own can be compared.  What is
checked: the HIP operator against the C restatement of the same form bit for bit; that the form degenerates to
the reference's update when damp = 0 (same values as Kernel_CUDA_Optimized, which is fixture-checked); that a
receiver on a grid node reads that node; and, on the CPU, that the sponge does what it is for."""
import numpy as np
import pytest

import oracle_binding as ob


def sponge(shape, nbl, peak, halo=4):
    """Quadratic ramp over the outer nbl interior cells of every face (0 in the middle), padded shape."""
    d = np.zeros(shape, dtype=np.float32)
    for ax in range(3):
        n = shape[ax] - 2 * halo
        i = np.arange(n)
        dist = np.minimum(i, n - 1 - i)
        ramp = np.where(dist < nbl, ((nbl - dist) / nbl) ** 2, 0.0) * peak
        full = np.zeros(shape[ax])
        full[halo:-halo] = ramp
        sh = [1, 1, 1]
        sh[ax] = shape[ax]
        d = np.maximum(d, full.reshape(sh).astype(np.float32))
    return np.ascontiguousarray(np.broadcast_to(d, shape), dtype=np.float32)


def random_case(n, nt, nsrc, nrec, seed, halo=4):
    rng = np.random.default_rng(seed)
    shape = tuple(v + 2 * halo for v in n)
    u = np.zeros((3,) + shape, dtype=np.float32)
    u[0] = rng.uniform(-1, 1, size=shape).astype(np.float32)
    u[1] = rng.uniform(-1, 1, size=shape).astype(np.float32)
    u[2] = u[0] + rng.uniform(-0.01, 0.01, size=shape).astype(np.float32)
    m = rng.uniform(1, 3, size=shape).astype(np.float32)
    h = (0.8, 1.0, 1.25)
    ext = (np.array(n) - 1) * np.array(h)
    src = rng.uniform(-50, 50, size=(nt, nsrc)).astype(np.float32)
    coords = (rng.uniform(0.05, 0.95, size=(nsrc, 3)) * ext).astype(np.float32)
    rec = np.full((nt, nrec), np.nan, dtype=np.float32)
    rcoords = (rng.uniform(0.0, 1.0, size=(nrec, 3)) * ext).astype(np.float32)
    rcoords[0] = ((n[0] - 1) * h[0] + 0.3 * h[0], 0.0, (n[2] - 1) * h[2])      # outermost cell: corners in the halo layer
    rcoords[1] = (5 * h[0], 7 * h[1], 4 * h[2])                               # on a grid node
    return u, m, src, coords, rec, rcoords, h


@pytest.mark.gpu
@pytest.mark.parametrize("n", [(20, 36, 44), (9, 8, 70)])
def test_forward_operator_matches_its_restatement(pkg, hiplib, n):
    nt = 11
    u, m, src, coords, rec, rcoords, h = random_case(n, nt, 2, 6, 3)
    damp = sponge(m.shape, 4, 40.0)
    u_ref, rec_ref = u.copy(), rec.copy()
    rc, _ = pkg.run_forward(ob.oracle().Oracle_Kernel_Forward, damp, m, rec_ref, rcoords, src, coords, u_ref, h=h)
    assert rc == 0
    rc, prof = pkg.run_forward(hiplib.Kernel_HIP_Forward, damp, m, rec, rcoords, src, coords, u, h=h)
    assert rc == 0 and prof.section0 > 0 and prof.section1 > 0
    assert ob.bits_equal(u, u_ref)
    assert ob.bits_equal(rec, rec_ref) and np.isfinite(rec).all()
    # without receivers and sources the same entry still steps the field
    u2, u2_ref = u.copy(), u.copy()
    assert pkg.run_forward(ob.oracle().Oracle_Kernel_Forward, damp, m, None, None, None, None, u2_ref, h=h, time_M=6)[0] == 0
    assert pkg.run_forward(hiplib.Kernel_HIP_Forward, damp, m, None, None, None, None, u2, h=h, time_M=6)[0] == 0
    assert ob.bits_equal(u2, u2_ref) and not ob.bits_equal(u2, u)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [(20, 36, 300), (37, 50, 131)])
def test_forward_operator_through_the_marching_sweep(pkg, hiplib, monkeypatch, n):
    """The damped update of large boxes runs through the marching sweep (FDTD_ORDER_MARCH=2 sends these small
    boxes there, in x-chunks of 16 planes): rows that end inside a z-tile, rows beyond the last y-tile."""
    monkeypatch.setenv("FDTD_ORDER_MARCH", "2")
    monkeypatch.setenv("FDTD_ORDER_XCHUNK", "16")
    nt = 8
    u, m, src, coords, rec, rcoords, h = random_case(n, nt, 2, 4, 5)
    damp = sponge(m.shape, 6, 40.0)
    u_ref, rec_ref = u.copy(), rec.copy()
    rc, _ = pkg.run_forward(ob.oracle().Oracle_Kernel_Forward, damp, m, rec_ref, rcoords, src, coords, u_ref, h=h)
    assert rc == 0
    rc, _ = pkg.run_forward(hiplib.Kernel_HIP_Forward, damp, m, rec, rcoords, src, coords, u, h=h)
    assert rc == 0
    assert ob.bits_equal(u, u_ref) and ob.bits_equal(rec, rec_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [(64, 512, 512), (96, 448, 384)])
def test_forward_operator_full_size_default_path(pkg, hiplib, n):
    """What ships by default, with no switch set: a box of 16 M cells, which the size rule sends to the large-box
    sweep of the damped update, a dozen steps with a sponge, sources and receivers against the restatement."""
    import os
    for name in ("FDTD_ORDER_MARCH", "FDTD_ORDER_XCHUNK", "FDTD_ORDER_LPR", "FDTD_ORDER_MARCH_MIN_CELLS",
                 "FDTD_FUSE2_MIN_CELLS", "FDTD_FUSE2"):
        assert name not in os.environ
    nt = 12
    u, m, src, coords, rec, rcoords, h = random_case(n, nt, 3, 5, 41)
    damp = sponge(m.shape, 12, 40.0)
    u_ref, rec_ref = u.copy(), rec.copy()
    rc, _ = pkg.run_forward(ob.oracle().Oracle_Kernel_Forward, damp, m, rec_ref, rcoords, src, coords, u_ref, h=h)
    assert rc == 0
    rc, prof = pkg.run_forward(hiplib.Kernel_HIP_Forward, damp, m, rec, rcoords, src, coords, u, h=h)
    assert rc == 0 and prof.section0 > 0
    assert hiplib.fdtd_hip_forward_last_call_fused() == 1      # two damped steps per sweep
    assert ob.bits_equal(u, u_ref) and ob.bits_equal(rec, rec_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("n,nsrc,nt", [((34, 16, 512), 2, 11), ((36, 32, 256), 3, 10), ((34, 16, 1024), 3, 9),
                                       ((40, 24, 384), 2, 8), ((33, 20, 640), 1, 12), ((34, 40, 128), 12, 7),
                                       ((36, 13, 516), 2, 9)])
def test_forward_operator_two_steps_per_sweep(pkg, hiplib, monkeypatch, n, nsrc, nt):
    """The damped two-step sweep on small boxes (switches let it take them): whole-row tiles, two tiles per workgroup,
    rows cut into z-tiles (seam pass), mixed and masked tilings, more than 8 sources (repair pass), odd step counts,
    different halos per ring level, a source in the outermost cell, receivers in the halo layer."""
    monkeypatch.setenv("FDTD_FUSE2_MIN_CELLS", "0")
    monkeypatch.setenv("FDTD_FUSE2_MIN_FILL", "0")
    u, m, src, coords, rec, rcoords, h = random_case(n, nt, nsrc, 5, 60 + n[2])
    coords[0] = ((n[0] - 1) * h[0] + 0.4 * h[0], 3.3 * h[1], 9.7 * h[2])     # outermost cell: corners in the halo layer
    damp = sponge(m.shape, 5, 40.0)
    u_ref, rec_ref = u.copy(), rec.copy()
    rc, _ = pkg.run_forward(ob.oracle().Oracle_Kernel_Forward, damp, m, rec_ref, rcoords, src, coords, u_ref, h=h)
    assert rc == 0
    before = hiplib.fdtd_hip_fused_launch_count()
    rc, prof = pkg.run_forward(hiplib.Kernel_HIP_Forward, damp, m, rec, rcoords, src, coords, u, h=h)
    assert rc == 0 and prof.section0 > 0
    assert hiplib.fdtd_hip_forward_last_call_fused() == 1 and hiplib.fdtd_hip_fused_launch_count() > before
    assert ob.bits_equal(u, u_ref)
    assert ob.bits_equal(rec, rec_ref) and np.isfinite(rec).all()


@pytest.mark.gpu
def test_two_step_forward_without_sponge_is_the_reference_update(pkg, hiplib, monkeypatch):
    """damp = 0 through the damped two-step sweep gives the values of Kernel_CUDA_Optimized (fixture-checked)."""
    monkeypatch.setenv("FDTD_FUSE2_MIN_CELLS", "0")
    n, nt = (40, 16, 512), 13
    u, m, src, coords, rec, rcoords, h = random_case(n, nt, 3, 3, 8)
    damp = np.zeros_like(m)
    u_plain = u.copy()
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u_plain, h=h)
    assert rc == 0
    rc, _ = pkg.run_forward(hiplib.Kernel_HIP_Forward, damp, m, rec, rcoords, src, coords, u, h=h)
    assert rc == 0 and hiplib.fdtd_hip_forward_last_call_fused() == 1
    assert np.array_equal(u, u_plain)          # same values (a negative zero may come out positive)


def test_generic_restatement_at_order_4_is_the_order_4_restatement(pkg):
    """CPU: Oracle_Kernel_ForwardOrder(order = 4) gives the bits of Oracle_Kernel_Forward."""
    n, nt = (14, 12, 20), 9
    u, m, src, coords, rec, rcoords, h = random_case(n, nt, 2, 4, 11)
    damp = sponge(m.shape, 4, 40.0)
    u_a, rec_a, u_b, rec_b = u.copy(), rec.copy(), u.copy(), rec.copy()
    assert pkg.run_forward(ob.oracle().Oracle_Kernel_Forward, damp, m, rec_a, rcoords, src, coords, u_a, h=h)[0] == 0
    assert pkg.run_forward(ob.forward_oracle_order(4), damp, m, rec_b, rcoords, src, coords, u_b, h=h)[0] == 0
    assert ob.bits_equal(u_a, u_b) and ob.bits_equal(rec_a, rec_b)
    assert pkg.run_forward(ob.forward_oracle_order(5), damp, m, rec_b, rcoords, src, coords, u_b, h=h)[0] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("order", [6, 8, 12])
@pytest.mark.parametrize("march", [False, True])
def test_forward_operator_at_space_orders(pkg, hiplib, monkeypatch, order, march):
    """Sponge, sources and receivers on arrays padded by the space order, through the one-cell-per-thread sweep
    and through the marching sweep, against the generic restatement."""
    monkeypatch.setenv("FDTD_ORDER_MARCH", "2" if march else "0")
    monkeypatch.setenv("FDTD_ORDER_XCHUNK", "16")
    n, nt = (24, 30, 140), 8
    u, m, src, coords, rec, rcoords, h = random_case(n, nt, 2, 4, 20 + order, halo=order)
    damp = sponge(m.shape, 6, 40.0, halo=order)
    b = (0, n[0] - 1, 0, n[1] - 1, 0, n[2] - 1)
    u_ref, rec_ref = u.copy(), rec.copy()
    assert pkg.run_forward(ob.forward_oracle_order(order), damp, m, rec_ref, rcoords, src, coords, u_ref, h=h, bounds=b)[0] == 0
    assert hiplib.fdtd_hip_set_space_order(order) == 0
    try:
        rc, _ = pkg.run_forward(hiplib.Kernel_HIP_Forward, damp, m, rec, rcoords, src, coords, u, h=h, bounds=b)
    finally:
        assert hiplib.fdtd_hip_set_space_order(4) == 0
    assert rc == 0
    assert ob.bits_equal(u, u_ref) and ob.bits_equal(rec, rec_ref) and np.isfinite(rec).all()


@pytest.mark.gpu
def test_forward_operator_without_sponge_is_the_reference_update(pkg, hiplib):
    n, nt = (20, 36, 44), 12
    u, m, src, coords, rec, rcoords, h = random_case(n, nt, 2, 3, 4)
    damp = np.zeros_like(m)
    u_plain = u.copy()
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u_plain, h=h)
    assert rc == 0
    # receivers see the field of the step they are sampled in: record it with one-step runs of the plain operator
    rc, _ = pkg.run_forward(hiplib.Kernel_HIP_Forward, damp, m, rec, rcoords, src, coords, u, h=h)
    assert rc == 0
    assert np.array_equal(u, u_plain)          # same values (a negative zero may come out positive)
    # the receiver on node (5, 7, 4) reads u(time) there: at time 0 that is level 0 of the input
    u0, _, _, _, _, _, _ = random_case(n, nt, 2, 3, 4)
    assert rec[0, 1] == u0[0, 5 + 4, 7 + 4, 4 + 4]


def test_sponge_absorbs_and_receivers_record(pkg):
    """CPU, restatement only: a pulse released in the middle of a 40^3 box reaches the faces within 120 steps
    (v dt / h = 0.4).  With the fixed frame of the reference it is reflected and its energy stays in the box; with
    a 10-cell sponge most of it is gone.  A receiver next to the pulse records a signal either way."""
    n, nt = (40, 40, 40), 120
    shape = tuple(v + 8 for v in n)
    ax = np.arange(shape[0]) - 4 - (n[0] - 1) / 2
    r2 = ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2
    pulse = np.exp(-r2 / 8.0).astype(np.float32)
    pulse[:4] = pulse[-4:] = 0
    pulse[:, :4] = pulse[:, -4:] = 0
    pulse[:, :, :4] = pulse[:, :, -4:] = 0
    m = np.ones(shape, dtype=np.float32)
    rcoords = np.array([[25.0, 19.5, 19.5]], dtype=np.float32)
    energy = {}
    for name, peak in (("frame", 0.0), ("sponge", 0.35)):
        u = np.zeros((3,) + shape, dtype=np.float32)
        u[0] = u[2] = pulse
        rec = np.zeros((nt, 1), dtype=np.float32)
        damp = sponge(shape, 10, peak)
        rc, _ = pkg.run_forward(ob.oracle().Oracle_Kernel_Forward, damp, m, rec, rcoords, None, None, u, dt=0.4, h=(1.0, 1.0, 1.0),
                                time_M=nt - 1)
        assert rc == 0 and np.isfinite(u).all()
        energy[name] = float((u[nt % 3].astype(np.float64) ** 2).sum())
        assert np.abs(rec).max() > 1e-3
    assert energy["sponge"] < 0.2 * energy["frame"], energy
