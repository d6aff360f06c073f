"""The allocation-free form of fdtd_hip_two_steps_slab (fdtd_hip_two_steps_slab_noalloc): its work
space is a static device array inside the library instead of a caller buffer.  Called through the C
ABI and compared bit for bit with fdtd_hip_two_steps_slab on a caller-provided work space, twice in a
row — the second call on a smaller range, so that it finds the first call's contents in the array —,
with a check that the device's free memory does not change across a call; once more with a range the
static array cannot hold, which must be refused on the two-sweep path and accepted on the fused one; and
from two streams, whose calls the library orders."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

H = (0.9, 1.0, 1.1)
X_SHIFT = 40


def _zone_inputs(torch, shape, x_lo, x_hi, nsrc, seed):
    nx, ny, nz = shape
    gen = torch.Generator(device="cuda").manual_seed(seed)
    pshape = (nx + 8, ny + 8, nz + 8)
    lv = [torch.rand(pshape, device="cuda", generator=gen) * 2 - 1 for _ in range(4)]
    lv[1] = lv[0] + 0.01 * lv[1]
    m = torch.rand(pshape, device="cuda", generator=gen) * 2 + 1
    src = coords = None
    if nsrc:
        src = torch.rand((4, nsrc), device="cuda", generator=gen) * 100 - 50
        coords = torch.rand((nsrc, 3), device="cuda", generator=gen)
        coords[:, 0] = (X_SHIFT + x_lo + coords[:, 0] * (x_hi - x_lo)) * H[0]   # corners in and next to the range
        coords[:, 1] *= (ny - 1) * H[1]
        coords[:, 2] *= (nz - 1) * H[2]
    return pshape, lv, m, src, coords


def _zone_prepare(pkg, torch, shape, x_lo, x_hi, nsrc, seed):
    """Inputs and argument list (without work space and stream) of one two-step call on planes x_lo..x_hi of a
    slab of shape (nx, ny, nz) that sits in the middle of a longer domain."""
    nx, ny, nz = shape
    pshape, lv, m, src, coords = _zone_inputs(torch, shape, x_lo, x_hi, nsrc, seed)
    g = pkg.Geom(*pshape, x_lo, x_hi, 0, ny - 1, 0, nz - 1, 1e-3, *H, 0.0, 0.0, 0.0)
    cur, prev, n1, n2 = (t.clone() for t in lv)
    common = (C.byref(g), m.data_ptr(), src.data_ptr() if nsrc else None, coords.data_ptr() if nsrc else None,
              cur.data_ptr(), prev.data_ptr(), n1.data_ptr(), n2.data_ptr(), 1, 0, nsrc - 1, max(nsrc, 1), 3,
              pkg.VARIANTS["auto"], 1, X_SHIFT, nx + 80)
    return common, n1, n2, (g, lv, m, src, coords, cur, prev)


def _zone_call(pkg, hiplib, torch, shape, x_lo, x_hi, nsrc, seed, static):
    """The two-sweep path with its u(time+1) planes built in the static array or in a caller buffer."""
    nx, ny, nz = shape
    common, n1, n2, keep = _zone_prepare(pkg, torch, shape, x_lo, x_hi, nsrc, seed)
    torch.cuda.synchronize()
    if static:
        free_before = torch.cuda.mem_get_info()[0]        # hipMemGetInfo
        rc = hiplib.fdtd_hip_two_steps_slab_noalloc(*common, None)
        torch.cuda.synchronize()
        # after the first call on the device (symbol look-up, event) the entry point allocates nothing
        assert not _zone_call.warm or torch.cuda.mem_get_info()[0] == free_before
        _zone_call.warm = True
    else:
        ws = torch.full(((x_hi - x_lo + 5), ny + 8, nz + 8), float("nan"), device="cuda")
        rc = hiplib.fdtd_hip_two_steps_slab(*common, ws.data_ptr(), ws.numel(), None)
    torch.cuda.synchronize()
    return rc, n1, n2, keep


_zone_call.warm = False


def test_static_work_space_matches_caller_work_space(pkg, hiplib):
    import torch
    # first call: 6 planes of 72 x 264 padded cells; second call: a smaller range of a smaller slab
    for shape, x_lo, x_hi, nsrc, seed in (((20, 64, 256), 6, 11, 3, 11), ((12, 24, 128), 4, 7, 2, 12)):
        rc_s, s1, s2, keep = _zone_call(pkg, hiplib, torch, shape, x_lo, x_hi, nsrc, seed, static=True)
        lv = keep[1]
        rc_c, c1, c2, _ = _zone_call(pkg, hiplib, torch, shape, x_lo, x_hi, nsrc, seed, static=False)
        assert rc_s == 0 and rc_c == 0
        assert torch.equal(s1.view(torch.int32), c1.view(torch.int32))
        assert torch.equal(s2.view(torch.int32), c2.view(torch.int32))
        assert not torch.equal(s2, lv[3]) and not torch.equal(s1, lv[2])   # something was computed
        # nothing outside the range was stored (the range's neighbours are updated by other calls)
        assert torch.equal(s1[:4 + x_lo], lv[2][:4 + x_lo]) and torch.equal(s1[4 + x_hi + 1:], lv[2][4 + x_hi + 1:])
        assert torch.equal(s2[:4 + x_lo], lv[3][:4 + x_lo]) and torch.equal(s2[4 + x_hi + 1:], lv[3][4 + x_hi + 1:])


def test_static_work_space_refuses_oversized_range(pkg, hiplib):
    import torch
    nx, ny, nz = 40, 504, 504          # 28 + 4 planes of 512 x 512 floats: twice the static array
    pshape = (nx + 8, ny + 8, nz + 8)
    assert (28 + 4) * pshape[1] * pshape[2] > pkg.abi.STATIC_SCRATCH_FLOATS
    lv = [torch.zeros(pshape, device="cuda") for _ in range(4)]
    m = torch.full(pshape, 1.5, device="cuda")
    g = pkg.Geom(*pshape, 6, 33, 0, ny - 1, 0, nz - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)

    def call(variant):
        return hiplib.fdtd_hip_two_steps_slab_noalloc(C.byref(g), m.data_ptr(), None, None, lv[0].data_ptr(),
                                                      lv[1].data_ptr(), lv[2].data_ptr(), lv[3].data_ptr(), 0, 0, -1, 1,
                                                      3, pkg.VARIANTS[variant], 1, 40, 200, None)
    assert call("auto") == 1      # hipErrorInvalidValue: the two-sweep path would need 32 planes of work space
    torch.cuda.synchronize()
    assert not lv[2].any() and not lv[3].any()
    # the same range through the fused sweep needs no work space: accepted whatever its size
    before = hiplib.fdtd_hip_fused_launch_count()
    lv[0][4 + 20, 4 + 100, 4 + 100] = 1.0
    assert call("march2") == 0
    torch.cuda.synchronize()
    assert hiplib.fdtd_hip_fused_launch_count() > before
    assert lv[2].any() and lv[3].any()


def test_static_work_space_orders_calls_on_different_streams(pkg, hiplib):
    """Two calls on two streams share the static array: the second is made to wait for the first's last use of
    it, so both give the bits of the caller-buffer form."""
    import torch
    shape, x_lo, x_hi = (20, 64, 256), 6, 11
    want = [_zone_call(pkg, hiplib, torch, shape, x_lo, x_hi, 2, seed, static=False) for seed in (31, 32)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    prepared = [_zone_prepare(pkg, torch, shape, x_lo, x_hi, 2, seed) for seed in (31, 32)]
    torch.cuda.synchronize()
    rcs = [hiplib.fdtd_hip_two_steps_slab_noalloc(*p[0], C.c_void_p(st.cuda_stream)) for p, st in zip(prepared, streams)]
    torch.cuda.synchronize()
    for rc_s, (_, s1, s2, _), (rc_c, c1, c2, _) in zip(rcs, prepared, want):
        assert rc_s == 0 and rc_c == 0
        assert torch.equal(s1.view(torch.int32), c1.view(torch.int32))
        assert torch.equal(s2.view(torch.int32), c2.view(torch.int32))
