"""fdtd_hip_multigpu_*: the slab decomposition as a session that owns its slabs on the devices, through the C ABI.
One GPU is enough: a device named several times carries several slabs, so every issuing thread, counter, peer
copy, event and stream of the schedule runs.  Bar: the oracle's bits (the fixtures' for the operator built on
the session)."""
import ctypes as C

import numpy as np
import pytest

import cases
import oracle_binding as ob
from test_oracle import check_against_golden

pytestmark = pytest.mark.gpu


def _geom(pkg, n, dt=1e-3, h=(0.1, 0.1, 0.1), o=(0.0, 0.0, 0.0)):
    return pkg.Geom(n[0] + 8, n[1] + 8, n[2] + 8, 0, n[0] - 1, 0, n[1] - 1, 0, n[2] - 1, dt, *h, *o)


def _oracle_to(pkg, m, src, coords, u0, time_M):
    u = u0.copy()
    ob.run_oracle(m, src, coords, u, time_M=time_M)
    return u


@pytest.mark.parametrize("threads", ["1", "0"])
@pytest.mark.parametrize("name", ["fused_z512", "fused_z512_edge_source", "ricker_n32", "correctness_n32",
                                  "ricker27_n32", "fused_z1024", "time_offset"])
def test_operator_on_session_both_issue_modes(pkg, hiplib, name, threads, monkeypatch):
    """The operator is create + upload + advance + gather; one issuing thread per slab and the calling thread alone
    give the fixture's bits."""
    monkeypatch.setenv("FDTD_MULTIGPU_DEVICES", "0,0,0")
    monkeypatch.setenv("FDTD_MULTIGPU_THREADS", threads)
    c = cases.build(name, pkg)
    rc, _ = pkg.run_operator(hiplib.Kernel_HIP_MultiGPU, c["m"], c["src"], c["coords"], c["u"], deviceid=-1, **c["kw"])
    assert rc == 0
    check_against_golden(name, c["u"], exact=True)


@pytest.mark.parametrize("flags", [0, 1, 2])
@pytest.mark.parametrize("n,slabs", [((72, 64, 512), 3), ((40, 24, 64), 4)])
def test_session_advances_in_pieces(pkg, hiplib, n, slabs, flags):
    """Device-made inputs, several advance calls of odd and even length with a gather in between (which puts the
    levels back into their ring slots and must leave the session able to go on), against the oracle run to the
    same time index.  Flags: threads, calling thread only, no interior / boundary split."""
    nt = 23
    u0, m, src, coords = pkg.standard_inputs(*n, nt, 2)
    with pkg.MultiGpuSession(_geom(pkg, n), [0] * slabs, flags) as s:
        s.fill_standard_inputs(nt, 2)
        got = np.empty_like(u0)
        t = 0
        for upto in (4, 5, 12, 19, 22):          # steps t .. upto
            s.advance(t, upto)
            t = upto + 1
            s.gather(got)
            ref = _oracle_to(pkg, m, src, coords, u0, upto)
            assert ob.bits_equal(got, ref), f"after step {upto}"
        mx, bad = s.max_abs()
        assert bad == 0 and mx == np.abs(got[t % 3][4:-4]).max()
        st = s.stats()
        assert st["slabs"] == slabs and st["pairs"] + st["single_steps"] > 0
        assert st["thread_per_slab"] == (0 if flags & 1 else 1)
        # time indices must follow each other
        assert hiplib.fdtd_hip_multigpu_advance(s.h, t + 1, t + 2, None) != 0


def test_session_upload_time_offset_and_halos(pkg, hiplib):
    """Host arrays with different halos per ring level and a source in the outermost cell (shells follow the ring),
    starting at a time index that is not a multiple of 3; two sessions' worth of calls equal one oracle run."""
    n, nt = (48, 40, 256), 17
    u0, m, src, coords = pkg.standard_inputs(*n, nt, 3)
    rng = np.random.default_rng(3)
    u0[:] = rng.uniform(-1, 1, size=u0.shape).astype(np.float32)
    m[:] = rng.uniform(1, 3, size=m.shape).astype(np.float32)
    coords[0] = (15.5 * 0.1, 2.0, 12.8)               # corners in planes 15 | 16: the cut between slabs 0 and 1
    coords[1] = (47 * 0.1 + 0.03, 0.0, 25.5 * 0.1)    # outermost cell: corners in the halo layer
    time_m = 4
    ref = u0.copy()
    ob.run_oracle(m, src, coords, ref, time_m=time_m, time_M=nt - 1)
    with pkg.MultiGpuSession(_geom(pkg, n), [0, 0, 0]) as s:
        s.set_sources(src, coords)
        s.upload(u0, m, time_m)
        s.advance(time_m, 9)
        s.advance(10, nt - 1)
        got = np.empty_like(u0)
        s.gather(got)
        assert s.stats()["shells_follow"] == 1
    assert ob.bits_equal(got, ref)


@pytest.mark.parametrize("n,slabs,zone_fused", [((72, 64, 512), 3, "0"), ((72, 64, 512), 3, "1"), ((40, 24, 64), 3, "0")])
def test_session_with_damping(pkg, hiplib, n, slabs, zone_fused, monkeypatch):
    """The damped update through the session: two damped steps per sweep on the interior of each slab, boundary zones
    through scratch or through the sweep itself, single damped steps at the end of odd counts (small box: everything
    single-sweep pairs); against the forward operator's restatement (no reference arithmetic: parity unpinned)."""
    from test_forward_operator import sponge
    monkeypatch.setenv("FDTD_SLAB_ZONE_FUSED", zone_fused)
    nt = 13
    u0, m, src, coords = pkg.standard_inputs(*n, nt, 2)
    rng = np.random.default_rng(17)
    u0[:] = rng.uniform(-1, 1, size=u0.shape).astype(np.float32)
    m[:] = rng.uniform(1, 3, size=m.shape).astype(np.float32)
    damp = sponge(m.shape, 6, 40.0)
    ref = u0.copy()
    rc, _ = pkg.run_forward(ob.oracle().Oracle_Kernel_Forward, damp, m, None, None, src, coords, ref)
    assert rc == 0
    with pkg.MultiGpuSession(_geom(pkg, n), [0] * slabs) as s:
        s.set_sources(src, coords)
        s.set_damping(damp)
        s.upload(u0, m, 0)
        s.advance(0, 6)
        s.advance(7, nt - 1)
        got = np.empty_like(u0)
        s.gather(got)
        assert ob.bits_equal(got, ref)
        # without the damping field the same session runs the plain update again
        s.set_damping(None)
        s.upload(u0, m, 0)
        s.advance(0, nt - 1)
        s.gather(got)
    plain = u0.copy()
    ob.run_oracle(m, src, coords, plain)
    assert ob.bits_equal(got, plain)


def test_checksum_is_the_same_for_every_decomposition(pkg, hiplib):
    """Full-width slabs (512 x 512 cross-section, the benchmark inputs made on the devices): 1, 2 and 4 slabs give the
    same 64-bit word sums of all three ring slots and the same max |u|, and the sum is the host's sum of the gathered
    field — the check the full-size runs rely on (profiles/r03_config5_checksums_one_gpu.txt)."""
    n, nt = (128, 512, 512), 24
    sums, maxes = [], []
    for slabs in (1, 2, 4):
        with pkg.MultiGpuSession(_geom(pkg, n), [0] * slabs) as s:
            s.fill_standard_inputs(nt + 1, 1)
            s.advance(0, nt - 1)
            sums.append(s.checksum())
            maxes.append(s.max_abs())
            if slabs == 4:
                got = np.empty((3, n[0] + 8, n[1] + 8, n[2] + 8), dtype=np.float32)
                s.gather(got)
    assert sums[0] == sums[1] == sums[2] and maxes[0] == maxes[1] == maxes[2] and maxes[0][0] > 0
    host = [int(got[slot][4:-4].view(np.uint32).astype(np.uint64).sum()) & 0xFFFFFFFFFFFFFFFF for slot in range(3)]
    assert host == sums[2]


def test_session_rejects_bad_arguments(pkg, hiplib):
    n = (32, 16, 64)
    g = _geom(pkg, n)
    h = C.c_void_p()
    dev = (C.c_int * 8)(*([0] * 8))
    assert hiplib.fdtd_hip_multigpu_create(C.byref(g), 5, dev, 0, C.byref(h)) != 0     # fewer than 8 planes per slab
    g2 = _geom(pkg, n)
    g2.x_m = 1
    assert hiplib.fdtd_hip_multigpu_create(C.byref(g2), 2, dev, 0, C.byref(h)) != 0    # not the whole x extent
    bad = (C.c_int * 2)(0, 99)
    assert hiplib.fdtd_hip_multigpu_create(C.byref(g), 2, bad, 0, C.byref(h)) != 0     # no such device
    assert hiplib.fdtd_hip_multigpu_create(C.byref(g), 2, dev, 0, C.byref(h)) == 0
    assert hiplib.fdtd_hip_multigpu_advance(h, 0, 3, None) != 0                        # no fields yet
    assert hiplib.fdtd_hip_multigpu_fill_standard_inputs(h, 6, 1) == 0
    assert hiplib.fdtd_hip_multigpu_advance(h, 0, 6, None) != 0                        # beyond the source samples
    assert hiplib.fdtd_hip_multigpu_advance(h, 0, 5, None) == 0
    assert hiplib.fdtd_hip_multigpu_destroy(h) == 0
