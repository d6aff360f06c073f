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


def test_device_copies_persist_with_devicerm_0(pkg, hiplib):
    """openacc.cpp:208-212 deletes the device copies only `if(devicerm)`.  Two half-length calls, the first
    with devicerm = 0, equal one full call bit for bit, and the second call uploads neither u nor m."""
    n, nt = (64, 64, 512), 40
    u_full, m, src, coords = pkg.standard_inputs(*n, nt, 2)
    rng = np.random.default_rng(9)
    m[:] = rng.uniform(1, 3, size=m.shape).astype(np.float32)
    u_half = u_full.copy()
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u_full)
    assert rc == 0
    ph = (C.c_double * 4)()
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u_half, time_m=0, time_M=nt // 2 - 1, devicerm=0)
    assert rc == 0
    hiplib.fdtd_hip_last_call_phases(C.byref(ph))
    first_h2d = ph[1]
    mid = u_half.copy()
    u_half[:] = np.nan        # the host copy is not read again: the device copy is the state
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u_half, time_m=nt // 2, time_M=nt - 1, devicerm=1)
    assert rc == 0
    hiplib.fdtd_hip_last_call_phases(C.byref(ph))
    assert ob.bits_equal(u_half, u_full)
    assert not ob.bits_equal(mid, u_full)
    assert ph[1] < 0.25 * first_h2d, (ph[1], first_h2d)      # only src and coords went up
    # devicerm = 1 dropped the copies: the next call starts from the host arrays again
    u_again = np.zeros_like(u_full)
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u_again)
    assert rc == 0 and ob.bits_equal(u_again, u_full)
    # FDTD_RESIDENT_REUPLOAD_U=1: the host's u wins over the device copy
    os.environ["FDTD_RESIDENT_REUPLOAD_U"] = "1"
    try:
        u2 = np.zeros_like(u_full)
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u2, time_m=0, time_M=nt // 2 - 1, devicerm=0)
        assert rc == 0
        u2[:] = 0.0
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u2, time_m=0, time_M=nt - 1, devicerm=1)
        assert rc == 0 and ob.bits_equal(u2, u_full)
    finally:
        del os.environ["FDTD_RESIDENT_REUPLOAD_U"]
        hiplib.fdtd_hip_release_device_copies()
