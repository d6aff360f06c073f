"""Kernel_HIP_MultiGPU (one host thread, one slab per device, ghost planes pushed with peer copies) and
the device copies that operator calls with devicerm = 0 leave behind, through the C ABI.  One GPU is
enough: FDTD_MULTIGPU_DEVICES names device 0 several times, so it carries several slabs and every
peer copy, event and stream of the multi-device schedule is exercised.  Bar: the fixtures' bits."""
import ctypes as C
import os

import numpy as np
import pytest

import cases
import oracle_binding as ob
from test_oracle import check_against_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
@pytest.mark.parametrize("name", list(cases.CASES))
def test_multigpu_operator_matches_golden(pkg, hiplib, name, devices, monkeypatch):
    """Every fixture case: whole-x boxes of at least 8 planes per slab are cut into slabs (two steps per
    exchange; interior and boundary chains where the fused sweep takes the cross-section, one chain
    otherwise), the others run on one device; the bytes are the fixture's either way."""
    monkeypatch.setenv("FDTD_MULTIGPU_DEVICES", devices)
    c = cases.build(name, pkg)
    rc, prof = pkg.run_operator(hiplib.Kernel_HIP_MultiGPU, c["m"], c["src"], c["coords"], c["u"], deviceid=-1, **c["kw"])
    assert rc == 0
    check_against_golden(name, c["u"], exact=True)
    n = c["u"].shape[1:]
    b = c["kw"].get("bounds") or (0, n[0] - 9, 0, n[1] - 9, 0, n[2] - 9)
    time_M = c["kw"].get("time_M", None if c["src"] is None else c["src"].shape[0] - 1)
    n_steps = time_M - c["kw"].get("time_m", 0) + 1
    nslab = len(devices.split(","))
    want = min(nslab, (n[0] - 8) // 8)
    whole = b[0] == 0 and b[1] == n[0] - 9 and b[3] >= b[2] and b[5] >= b[4] and n_steps > 0
    assert hiplib.fdtd_hip_multigpu_last_slab_count() == (want if (whole and want >= 2) else 1), name
    assert (prof.section0 > 0) == (n_steps > pkg.WARMUP_STEPS)


@pytest.mark.parametrize("zone_fused", ["0", "1"])
def test_multigpu_operator_large_cross_section(pkg, hiplib, zone_fused, monkeypatch):
    """Both boundary-zone forms (4-plane zones through scratch, outer thirds through the fused sweep first)
    against the oracle on a box big enough for three slabs of 24 planes at a 64 x 512 cross-section, with a
    source on a cut between two slabs and one in the outermost cell."""
    monkeypatch.setenv("FDTD_MULTIGPU_DEVICES", "0,0,0")
    monkeypatch.setenv("FDTD_SLAB_ZONE_FUSED", zone_fused)
    n, nt = (72, 64, 512), 15
    u, m, src, coords = pkg.standard_inputs(*n, nt, 3)
    coords[0] = (23.5 * 0.1, 3.1, 25.55)        # corners in planes 23 | 24: the cut between slabs 0 and 1
    coords[1] = (71 * 0.1 + 0.03, 0.0, 51.1)    # outermost cell: corners in the halo layer
    rng = np.random.default_rng(5)
    u[:2] = rng.uniform(-1, 1, size=u[:2].shape).astype(np.float32)   # different halos per level
    m[:] = rng.uniform(1, 3, size=m.shape).astype(np.float32)
    u_ref = u.copy()
    ob.run_oracle(m, src, coords, u_ref)
    before = hiplib.fdtd_hip_fused_launch_count()
    rc, prof = pkg.run_operator(hiplib.Kernel_HIP_MultiGPU, m, src, coords, u, deviceid=-1)
    assert rc == 0 and hiplib.fdtd_hip_multigpu_last_slab_count() == 3
    assert hiplib.fdtd_hip_fused_launch_count() > before
    assert ob.bits_equal(u, u_ref)
    assert prof.section0 > 0


def _uploads(hiplib):
    up = (C.c_int * 4)()
    hiplib.fdtd_hip_last_call_uploads(C.byref(up))
    return list(up)


def test_device_copies_persist_with_devicerm_0(pkg, hiplib, monkeypatch):
    """openacc.cpp:208-212: u is copied out and released at the end of every call, m / src / src_coords are
    deleted only `if(devicerm)`.  Default mode: after a devicerm = 0 call of Kernel_OpenACC the next call uploads
    u (the host's u is the state) but not m; the CUDA-named operators ignore devicerm like cuda.cu:321.  Opt-in
    modes: FDTD_RESIDENT_ALL_OPERATORS extends this to every operator, FDTD_RESIDENT_KEEP_U keeps u as well."""
    n, nt = (64, 64, 512), 40
    u_full, m, src, coords = pkg.standard_inputs(*n, nt, 2)
    rng = np.random.default_rng(9)
    m[:] = rng.uniform(1, 3, size=m.shape).astype(np.float32)
    u0 = u_full.copy()
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u_full)
    assert rc == 0
    half = dict(time_m=0, time_M=nt // 2 - 1)
    rest = dict(time_m=nt // 2, time_M=nt - 1)
    try:
        # --- default: the reference's semantics, operator by operator ---
        u = u0.copy()
        rc, _ = pkg.run_operator(hiplib.Kernel_OpenACC, m, src, coords, u, devicerm=0, **half)
        assert rc == 0 and _uploads(hiplib) == [1, 1, 1, 1]
        rc, _ = pkg.run_operator(hiplib.Kernel_OpenACC, m, src, coords, u, devicerm=1, **rest)
        assert rc == 0 and _uploads(hiplib) == [1, 0, 1, 1]     # u from the host again, m was present
        assert ob.bits_equal(u, u_full)
        rc, _ = pkg.run_operator(hiplib.Kernel_OpenACC, m, src, coords, u0.copy(), devicerm=1, **half)
        assert rc == 0 and _uploads(hiplib) == [1, 1, 1, 1]     # devicerm = 1 dropped the copies
        # a host edit of u between two calls is seen (the reference re-uploads u)
        u = u0.copy()
        rc, _ = pkg.run_operator(hiplib.Kernel_OpenACC, m, src, coords, u, devicerm=0, **half)
        assert rc == 0
        u[:] = u0
        rc, _ = pkg.run_operator(hiplib.Kernel_OpenACC, m, src, coords, u, devicerm=1, time_m=0, time_M=nt - 1)
        assert rc == 0 and ob.bits_equal(u, u_full)
        # the CUDA-named operators ignore devicerm (cuda.cu:321): everything goes up on every call
        u = u0.copy()
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u, devicerm=0, **half)
        assert rc == 0
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u, devicerm=0, **rest)
        assert rc == 0 and _uploads(hiplib) == [1, 1, 1, 1] and ob.bits_equal(u, u_full)

        # --- every operator, u kept too: two half-length calls equal one full call, nothing but the sources goes up ---
        monkeypatch.setenv("FDTD_RESIDENT_ALL_OPERATORS", "1")
        monkeypatch.setenv("FDTD_RESIDENT_KEEP_U", "1")
        u = u0.copy()
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u, devicerm=0, **half)
        assert rc == 0
        mid = u.copy()
        u[:] = np.nan        # the host copy is not read again: the device copy is the state
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u, devicerm=1, **rest)
        assert rc == 0 and _uploads(hiplib) == [0, 0, 1, 1]
        assert ob.bits_equal(u, u_full) and not ob.bits_equal(mid, u_full)
        # devicerm = 1 dropped the copies: the next call starts from the host arrays again
        u = u0.copy()
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u)
        assert rc == 0 and _uploads(hiplib) == [1, 1, 1, 1] and ob.bits_equal(u, u_full)
        # another u array with the same m: u goes up, m stays
        monkeypatch.delenv("FDTD_RESIDENT_KEEP_U")
        u = u0.copy()
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u, devicerm=0, **half)
        assert rc == 0
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u, devicerm=0, **rest)
        assert rc == 0 and _uploads(hiplib) == [1, 0, 1, 1] and ob.bits_equal(u, u_full)
    finally:
        hiplib.fdtd_hip_release_device_copies()


def test_resident_m_is_revalidated(pkg, hiplib):
    """A host array freed and allocated again at the same address with other values (or edited in place) must
    not compute on the stale device copy: the entry's content tag of m differs and m is uploaded again."""
    n, nt = (24, 20, 64), 12
    u0, m, src, coords = pkg.standard_inputs(*n, nt, 2)
    rng = np.random.default_rng(21)
    try:
        m[:] = rng.uniform(1, 3, size=m.shape).astype(np.float32)
        u = u0.copy()
        rc, _ = pkg.run_operator(hiplib.Kernel_OpenACC, m, src, coords, u, devicerm=0)
        assert rc == 0
        addr = m.ctypes.data
        # "free and reallocate": same address, same shape, other values
        m[:] = rng.uniform(1, 3, size=m.shape).astype(np.float32)
        assert m.ctypes.data == addr
        u = u0.copy()
        u_ref = u0.copy()
        ob.run_oracle(m, src, coords, u_ref)
        rc, _ = pkg.run_operator(hiplib.Kernel_OpenACC, m, src, coords, u, devicerm=0)
        assert rc == 0 and _uploads(hiplib)[1] == 1
        assert ob.bits_equal(u, u_ref)
        # unchanged m: present, not uploaded, same bits
        u = u0.copy()
        rc, _ = pkg.run_operator(hiplib.Kernel_OpenACC, m, src, coords, u, devicerm=0)
        assert rc == 0 and _uploads(hiplib)[1] == 0 and ob.bits_equal(u, u_ref)
        # same address, another shape (a view of the same buffer): the old entry is dropped
        n2 = (20, 24, 64)
        u2, m2, src2, coords2 = pkg.standard_inputs(*n2, nt, 2)
        m_view = m.reshape(-1)[:m2.size].reshape(m2.shape)
        m_view[:] = rng.uniform(1, 3, size=m2.shape).astype(np.float32)
        assert m_view.ctypes.data == addr
        u2_ref = u2.copy()
        ob.run_oracle(m_view, src2, coords2, u2_ref)
        rc, _ = pkg.run_operator(hiplib.Kernel_OpenACC, m_view, src2, coords2, u2, devicerm=0)
        assert rc == 0 and _uploads(hiplib)[1] == 1 and ob.bits_equal(u2, u2_ref)
    finally:
        hiplib.fdtd_hip_release_device_copies()
