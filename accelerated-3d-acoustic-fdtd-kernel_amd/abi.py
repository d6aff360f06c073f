"""ctypes binding of libfdtd_hip.so — the structs and entry points of include/fdtd_abi.h.

Struct layouts follow the reference driver (main.cpp:35-50); the operator signature is
main.cpp:53-72 (max before min in every pair, all ranges inclusive).
"""
import ctypes as C
import os

import numpy as np

HALO = 4            # main.cpp:32, cuda.cu:55
WARMUP_STEPS = 5    # openacc.cpp:5
STATIC_SCRATCH_FLOATS = 4 << 20   # FDTD_STATIC_SCRATCH_FLOATS of fdtd_abi.h
MULTIGPU_ONE_THREAD, MULTIGPU_NO_OVERLAP, MULTIGPU_NO_EXCHANGE = 1, 2, 4   # flags of fdtd_hip_multigpu_create
VARIANTS = {"flat": 0, "naive": 1, "march": 2, "tile3d": 3, "vec": 4, "march2": 5, "rows": 6, "auto": -1}

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # FDTD_HIP_LIB: another build of the library (A/B measurements)
    return os.environ.get("FDTD_HIP_LIB") or os.path.join(_HERE, "libfdtd_hip.so")


class Dataobj(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("size", C.POINTER(C.c_int)),
        ("nbytes", C.c_ulong),
        ("npsize", C.c_void_p),
        ("dsize", C.c_void_p),
        ("hsize", C.c_void_p),
        ("hofs", C.c_void_p),
        ("oofs", C.c_void_p),
        ("dmap", C.c_void_p),
    ]


class Profiler(C.Structure):
    _fields_ = [("section0", C.c_double), ("section1", C.c_double)]


class Geom(C.Structure):
    _fields_ = [
        ("nxp", C.c_int), ("nyp", C.c_int), ("nzp", C.c_int),
        ("x_m", C.c_int), ("x_M", C.c_int), ("y_m", C.c_int), ("y_M", C.c_int),
        ("z_m", C.c_int), ("z_M", C.c_int),
        ("dt", C.c_float), ("h_x", C.c_float), ("h_y", C.c_float), ("h_z", C.c_float),
        ("o_x", C.c_float), ("o_y", C.c_float), ("o_z", C.c_float),
    ]


class Parity(C.Structure):
    _fields_ = [
        ("max_abs", C.c_double), ("max_rel", C.c_double), ("l2", C.c_double),
        ("nan_count", C.c_long), ("inf_count", C.c_long),
    ]


OPERATOR_ARGTYPES = (
    [C.POINTER(Dataobj)] * 4 + [C.c_int] * 6 + [C.c_float] * 7 + [C.c_int] * 6
    + [C.POINTER(Profiler)]
)

# Kernel_HIP_Forward: damp, m, rec, rec_coords, src, src_coords, u; bounds; dt, h, o; p_rec, p_src, time; device; timers
FORWARD_ARGTYPES = ([C.POINTER(Dataobj)] * 7 + [C.c_int] * 6 + [C.c_float] * 7 + [C.c_int] * 8
                    + [C.POINTER(Profiler)])

_lib = None


def lib():
    """Load libfdtd_hip.so (once).  Raises if it has not been built: there is no other path."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own HIP runtime.  If it is going to be used in this process (device
    # memory, streams, torch.distributed) it has to be loaded before libfdtd_hip.so pulls in the
    # system one, otherwise torch finds the GPU already claimed by a second runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `make -C {_HERE}` (hipcc --offload-arch=gfx950); "
            "this engine has no CPU fallback")
    L = C.CDLL(path)
    for name in ("Kernel_OpenACC", "Kernel_CUDA", "Kernel_CUDA_Optimized", "Kernel_HIP_FP16Storage",
                 "Kernel_HIP_MultiGPU"):
        fn = getattr(L, name)
        fn.argtypes = OPERATOR_ARGTYPES
        fn.restype = C.c_int
    L.Kernel_HIP_Forward.argtypes = FORWARD_ARGTYPES
    L.Kernel_HIP_Forward.restype = C.c_int
    L.Kernel_HIP_ForwardOrder.argtypes = FORWARD_ARGTYPES + [C.c_int]
    L.Kernel_HIP_ForwardOrder.restype = C.c_int
    L.Kernel_HIP_SpaceOrder.argtypes = OPERATOR_ARGTYPES + [C.c_int]
    L.Kernel_HIP_SpaceOrder.restype = C.c_int
    L.fdtd_hip_time_loop_order.argtypes = [C.POINTER(Geom), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.POINTER(Profiler)]
    L.fdtd_hip_time_loop_order.restype = C.c_int
    L.fdtd_hip_stencil_step_damped.argtypes = [C.POINTER(Geom), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                               C.c_int, C.c_void_p]
    L.fdtd_hip_stencil_step_damped.restype = C.c_int
    L.fdtd_hip_receiver_sample.argtypes = [C.POINTER(Geom), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.fdtd_hip_receiver_sample.restype = C.c_int
    L.fdtd_hip_stencil_step_damped_order.argtypes = [C.POINTER(Geom), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                     C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.fdtd_hip_stencil_step_damped_order.restype = C.c_int
    L.fdtd_hip_receiver_sample_order.argtypes = [C.POINTER(Geom), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.fdtd_hip_receiver_sample_order.restype = C.c_int
    L.fdtd_hip_set_space_order.argtypes = [C.c_int]
    L.fdtd_hip_set_space_order.restype = C.c_int
    L.fdtd_hip_get_space_order.argtypes = []
    L.fdtd_hip_get_space_order.restype = C.c_int
    L.fdtd_hip_stencil_step_order.argtypes = [C.POINTER(Geom), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_void_p]
    L.fdtd_hip_stencil_step_order.restype = C.c_int
    L.fdtd_hip_source_inject_order.argtypes = [C.POINTER(Geom), C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int, C.c_void_p]
    L.fdtd_hip_source_inject_order.restype = C.c_int
    L.fdtd_hip_source_inject_slab_order.argtypes = L.fdtd_hip_source_inject_order.argtypes[:-2] + [
        C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.fdtd_hip_source_inject_slab_order.restype = C.c_int
    L.fdtd_hip_multigpu_last_slab_count.argtypes = []
    L.fdtd_hip_multigpu_last_slab_count.restype = C.c_int
    L.fdtd_hip_multigpu_create.argtypes = [C.POINTER(Geom), C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    L.fdtd_hip_multigpu_destroy.argtypes = [C.c_void_p]
    L.fdtd_hip_multigpu_set_sources.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                C.c_int, C.c_int]
    L.fdtd_hip_multigpu_fill_standard_inputs.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.fdtd_hip_multigpu_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.fdtd_hip_multigpu_set_damping.argtypes = [C.c_void_p, C.c_void_p]
    L.fdtd_hip_multigpu_set_damping.restype = C.c_int
    L.fdtd_hip_multigpu_advance.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Profiler)]
    L.fdtd_hip_multigpu_gather.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.fdtd_hip_multigpu_max_abs.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_long)]
    L.fdtd_hip_multigpu_checksum.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]
    L.fdtd_hip_multigpu_checksum.restype = C.c_int
    L.fdtd_hip_multigpu_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double * 10), C.c_int]
    for name in ("create", "destroy", "set_sources", "fill_standard_inputs", "upload", "advance", "gather", "max_abs",
                 "stats"):
        getattr(L, "fdtd_hip_multigpu_" + name).restype = C.c_int
    L.fdtd_hip_release_device_copies.argtypes = []
    L.fdtd_hip_release_device_copies.restype = None
    L.fdtd_hip_last_call_phases.argtypes = [C.POINTER(C.c_double * 4)]
    L.fdtd_hip_last_call_phases.restype = None
    L.fdtd_hip_last_call_uploads.argtypes = [C.POINTER(C.c_int * 4)]
    L.fdtd_hip_last_call_uploads.restype = None
    L.fdtd_hip_stencil_step.argtypes = [C.POINTER(Geom), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p]
    L.fdtd_hip_stencil_step.restype = C.c_int
    L.fdtd_hip_source_inject.argtypes = [C.POINTER(Geom), C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p]
    L.fdtd_hip_source_inject.restype = C.c_int
    L.fdtd_hip_source_inject_slab.argtypes = L.fdtd_hip_source_inject.argtypes[:-1] + [
        C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.fdtd_hip_source_inject_slab.restype = C.c_int
    L.fdtd_hip_two_steps.argtypes = [C.POINTER(Geom)] + [C.c_void_p] * 7 + [C.c_int] * 6 + [C.c_void_p]
    L.fdtd_hip_two_steps.restype = C.c_int
    L.fdtd_hip_two_steps_damped.argtypes = [C.POINTER(Geom)] + [C.c_void_p] * 8 + [C.c_int] * 7 + [C.c_void_p]
    L.fdtd_hip_two_steps_damped.restype = C.c_int
    L.fdtd_hip_forward_last_call_fused.argtypes = []
    L.fdtd_hip_forward_last_call_fused.restype = C.c_int
    L.fdtd_hip_two_steps_ex.argtypes = [C.POINTER(Geom)] + [C.c_void_p] * 7 + [C.c_int] * 7 + [C.c_void_p] * 3
    L.fdtd_hip_two_steps_ex.restype = C.c_int
    L.fdtd_hip_two_steps_slab.argtypes = ([C.POINTER(Geom)] + [C.c_void_p] * 7 + [C.c_int] * 9
                                          + [C.c_void_p, C.c_long, C.c_void_p])
    L.fdtd_hip_two_steps_slab.restype = C.c_int
    L.fdtd_hip_two_steps_slab_noalloc.argtypes = ([C.POINTER(Geom)] + [C.c_void_p] * 7 + [C.c_int] * 9
                                                  + [C.c_void_p])
    L.fdtd_hip_two_steps_slab_noalloc.restype = C.c_int
    L.fdtd_hip_one_step_slab.argtypes = [C.POINTER(Geom)] + [C.c_void_p] * 6 + [C.c_int] * 8 + [C.c_void_p]
    L.fdtd_hip_one_step_slab.restype = C.c_int
    L.fdtd_hip_fused_launch_count.argtypes = []
    L.fdtd_hip_fused_launch_count.restype = C.c_long
    L.fdtd_hip_two_steps_fuses.argtypes = [C.POINTER(Geom), C.c_int, C.c_int]
    L.fdtd_hip_two_steps_fuses.restype = C.c_int
    L.fdtd_hip_two_steps_tiling.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                            C.POINTER(C.c_double)]
    L.fdtd_hip_two_steps_tiling.restype = C.c_long
    L.fdtd_hip_small_step_count.argtypes = []
    L.fdtd_hip_small_step_count.restype = C.c_long
    L.fdtd_hip_one_step.argtypes = [C.POINTER(Geom)] + [C.c_void_p] * 6 + [C.c_int] * 6 + [C.c_void_p]
    L.fdtd_hip_one_step.restype = C.c_int
    L.fdtd_hip_copy_shell.argtypes = [C.POINTER(Geom), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.fdtd_hip_copy_shell.restype = C.c_int
    L.fdtd_hip_time_loop.argtypes = [C.POINTER(Geom), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.POINTER(Profiler)]
    L.fdtd_hip_time_loop.restype = C.c_int
    L.fdtd_hip_stencil_step_h16.argtypes = [C.POINTER(Geom), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                            C.c_int, C.c_void_p]
    L.fdtd_hip_stencil_step_h16.restype = C.c_int
    L.fdtd_hip_source_inject_h16.argtypes = L.fdtd_hip_source_inject.argtypes
    L.fdtd_hip_source_inject_h16.restype = C.c_int
    L.fdtd_hip_convert_f32_to_h16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.fdtd_hip_convert_f32_to_h16.restype = C.c_int
    L.fdtd_hip_convert_h16_to_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.fdtd_hip_convert_h16_to_f32.restype = C.c_int
    L.fdtd_fill_ricker.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
    L.fdtd_fill_ricker.restype = None
    L.fdtd_fill_source_coords.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.fdtd_fill_source_coords.restype = None
    L.fdtd_fill_correctness_inputs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.fdtd_fill_correctness_inputs.restype = None
    L.fdtd_compare_fields.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Parity)]
    L.fdtd_compare_fields.restype = None
    for name in ("run_openacc", "run_cuda", "run_cuda_optimized"):
        fn = getattr(L, name)
        fn.argtypes = [C.c_int] * 5 + [C.c_void_p, C.POINTER(C.c_float), C.POINTER(Profiler)]
        fn.restype = C.c_int
    L.FDTD_SetRuntimeConfig.argtypes = [C.c_int, C.c_int, C.c_int]
    L.FDTD_SetRuntimeConfig.restype = None
    L.fdtd_hip_version.argtypes = []
    L.fdtd_hip_version.restype = C.c_char_p
    _lib = L
    return L


def make_dataobj(arr, sizes=None):
    """Wrap a C-contiguous float32 array (or None) the way main.cpp:114-126 does.

    Returns (Dataobj, keepalive); keep `keepalive` referenced while the struct is in use.
    """
    if arr is None:
        sz = (C.c_int * len(sizes))(*sizes)
        return Dataobj(None, sz, 0, None, None, None, None, None, None), (sz,)
    assert arr.dtype == np.float32 and arr.flags["C_CONTIGUOUS"]
    shape = tuple(sizes) if sizes is not None else arr.shape
    sz = (C.c_int * len(shape))(*shape)
    d = Dataobj(arr.ctypes.data, sz, arr.nbytes, None, None, None, None, None, None)
    return d, (sz, arr)


def run_operator(fn, m, src, coords, u, *, bounds=None, dt=1e-3, h=(0.1, 0.1, 0.1),
                 o=(0.0, 0.0, 0.0), p_src=None, time_m=0, time_M=None, deviceid=0, devicerm=1,
                 src_sizes=None, coord_sizes=None):
    """Call an operator with the 24-argument signature of main.cpp:53-72.

    `u` is float32 [3][nxp][nyp][nzp] (updated in place), `m` is [nxp][nyp][nzp];
    `src` is [T][nsrc] and `coords` is [nsrc][3], or both None for a run without
    sources (main.cpp:537-545 passes sizes {0,1} and {2,0} then).
    `bounds` = (x_m, x_M, y_m, y_M, z_m, z_M), default the whole interior.
    Returns (return code, Profiler).
    """
    nxp, nyp, nzp = u.shape[1:]
    if bounds is None:
        bounds = (0, nxp - 2 * HALO - 1, 0, nyp - 2 * HALO - 1, 0, nzp - 2 * HALO - 1)
    x_m, x_M, y_m, y_M, z_m, z_M = bounds
    keep = []
    md, k = make_dataobj(m); keep.append(k)
    ud, k = make_dataobj(u); keep.append(k)
    if src is None:
        sd, k = make_dataobj(None, src_sizes or (0, 1)); keep.append(k)
        cd, k = make_dataobj(None, coord_sizes or (2, 0)); keep.append(k)
        if p_src is None:
            p_src = (0, -1)
    else:
        sd, k = make_dataobj(src, src_sizes); keep.append(k)
        cd, k = make_dataobj(coords, coord_sizes); keep.append(k)
        if p_src is None:
            p_src = (0, src.shape[1] - 1)
        if time_M is None:
            time_M = src.shape[0] - 1
    if time_M is None:
        raise ValueError("time_M is required when there are no sources")
    prof = Profiler(0.0, 0.0)
    rc = fn(C.byref(md), C.byref(sd), C.byref(cd), C.byref(ud),
            x_M, x_m, y_M, y_m, z_M, z_m, dt, h[0], h[1], h[2], o[0], o[1], o[2],
            p_src[1], p_src[0], time_M, time_m, deviceid, devicerm, C.byref(prof))
    del keep
    return rc, prof


def run_forward(fn, damp, m, rec, rec_coords, src, coords, u, *, bounds=None, dt=1e-3, h=(0.1, 0.1, 0.1),
                o=(0.0, 0.0, 0.0), time_m=0, time_M=None, deviceid=0):
    """Call a forward operator (Kernel_HIP_Forward or the oracle's) on numpy arrays; `rec` is [T][nrec] (written),
    `rec_coords` [nrec][3], or both None."""
    nxp, nyp, nzp = u.shape[1:]
    if bounds is None:
        bounds = (0, nxp - 2 * HALO - 1, 0, nyp - 2 * HALO - 1, 0, nzp - 2 * HALO - 1)
    x_m, x_M, y_m, y_M, z_m, z_M = bounds
    keep = []

    def wrap(arr, empty_sizes):
        d, k = make_dataobj(arr, None if arr is not None else empty_sizes)
        keep.append(k)
        return d
    dd, md, ud = wrap(damp, None), wrap(m, None), wrap(u, None)
    rd, rcd = wrap(rec, (0, 1)), wrap(rec_coords, (2, 0))
    sd, scd = wrap(src, (0, 1)), wrap(coords, (2, 0))
    p_rec = (0, rec.shape[1] - 1) if rec is not None else (0, -1)
    p_src = (0, src.shape[1] - 1) if src is not None else (0, -1)
    if time_M is None:
        time_M = (src if src is not None else rec).shape[0] - 1
    prof = Profiler(0.0, 0.0)
    rc = fn(C.byref(dd), C.byref(md), C.byref(rd), C.byref(rcd), C.byref(sd), C.byref(scd), C.byref(ud),
            x_M, x_m, y_M, y_m, z_M, z_m, dt, h[0], h[1], h[2], o[0], o[1], o[2],
            p_rec[1], p_rec[0], p_src[1], p_src[0], time_M, time_m, deviceid, 1, C.byref(prof))
    del keep
    return rc, prof


# ----- input generators of the reference driver (implemented in the library) -----

def fill_ricker(T, S, dt=1e-3):
    src = np.empty((T, max(S, 1)), dtype=np.float32)
    lib().fdtd_fill_ricker(src.ctypes.data, T, max(S, 1), dt)
    return src


def fill_source_coords(S, nx, ny, nz):
    coords = np.empty((max(S, 1), 3), dtype=np.float32)
    lib().fdtd_fill_source_coords(coords.ctypes.data, max(S, 1), nx, ny, nz)
    return coords


def fill_correctness_inputs(n):
    """Inputs of the driver's parity run on an n^3 grid (main.cpp:525-532)."""
    npad = n + 2 * HALO
    u = np.empty((3, npad, npad, npad), dtype=np.float32)
    m = np.empty((npad, npad, npad), dtype=np.float32)
    lib().fdtd_fill_correctness_inputs(u.ctypes.data, m.ctypes.data, npad ** 3)
    return u, m


def standard_inputs(nx, ny, nz, nt, nsrc=1, halo=HALO):
    """Inputs of the driver's benchmark run (main.cpp:345-356): u=0, m=1.5, Ricker sources; `halo` cells of
    padding per side (HALO = STENCIL_ORDER, main.cpp:32)."""
    shape = (nx + 2 * halo, ny + 2 * halo, nz + 2 * halo)
    u = np.zeros((3,) + shape, dtype=np.float32)
    m = np.full(shape, 1.5, dtype=np.float32)
    src = fill_ricker(nt, nsrc)
    coords = fill_source_coords(nsrc, nx, ny, nz)
    return u, m, src, coords


def compare_fields(test, ref):
    """max_abs, max_rel, L2 and NaN/Inf counts exactly as main.cpp:573-593 computes them."""
    test = np.ascontiguousarray(test, dtype=np.float32)
    ref = np.ascontiguousarray(ref, dtype=np.float32)
    assert test.size == ref.size
    p = Parity()
    lib().fdtd_compare_fields(test.ctypes.data, ref.ctypes.data, test.size, C.byref(p))
    return p


class MultiGpuSession:
    """fdtd_hip_multigpu_* of fdtd_abi.h: the slabs of one domain on several devices (or several slabs on one),
    owned by the library between calls."""

    def __init__(self, geom, devices, flags=0):
        self.L = lib()
        self.geom = geom
        self.devices = list(devices)
        self.h = C.c_void_p()
        arr = (C.c_int * len(self.devices))(*self.devices)
        rc = self.L.fdtd_hip_multigpu_create(C.byref(geom), len(self.devices), arr, flags, C.byref(self.h))
        if rc:
            raise RuntimeError(f"fdtd_hip_multigpu_create returned {rc}")

    def _check(self, rc, what):
        if rc:
            raise RuntimeError(f"fdtd_hip_multigpu_{what} returned {rc}")

    def set_sources(self, src, coords, p_src=None):
        if src is None:
            return self._check(self.L.fdtd_hip_multigpu_set_sources(self.h, None, 0, 1, None, 0, 3, 0, -1), "set_sources")
        p_src = p_src or (0, src.shape[1] - 1)
        self._check(self.L.fdtd_hip_multigpu_set_sources(self.h, src.ctypes.data, src.shape[0], src.shape[1],
                                                         coords.ctypes.data, coords.shape[0], coords.shape[1],
                                                         p_src[0], p_src[1]), "set_sources")

    def fill_standard_inputs(self, nt, nsrc=1):
        self._check(self.L.fdtd_hip_multigpu_fill_standard_inputs(self.h, nt, nsrc), "fill_standard_inputs")

    def upload(self, u, m, time_m=0):
        self._check(self.L.fdtd_hip_multigpu_upload(self.h, u.ctypes.data, None if m is None else m.ctypes.data, time_m),
                    "upload")

    def set_damping(self, damp):
        self._check(self.L.fdtd_hip_multigpu_set_damping(self.h, None if damp is None else damp.ctypes.data), "set_damping")

    def advance(self, time_m, time_M):
        prof = Profiler(0.0, 0.0)
        self._check(self.L.fdtd_hip_multigpu_advance(self.h, time_m, time_M, C.byref(prof)), "advance")
        return prof.section0

    def gather(self, u):
        """All three ring slots into u[3][nxp][nyp][nzp]."""
        for slot in range(3):
            self._check(self.L.fdtd_hip_multigpu_gather(self.h, slot, u[slot].ctypes.data), "gather")

    def max_abs(self):
        mx, bad = C.c_float(0.0), C.c_long(0)
        self._check(self.L.fdtd_hip_multigpu_max_abs(self.h, C.byref(mx), C.byref(bad)), "max_abs")
        return mx.value, bad.value

    def checksum(self):
        """64-bit word sums of the three ring slots."""
        out = []
        for slot in range(3):
            v = C.c_ulonglong(0)
            self._check(self.L.fdtd_hip_multigpu_checksum(self.h, slot, C.byref(v)), "checksum")
            out.append(v.value)
        return out

    def stats(self, reset=False):
        out = (C.c_double * 10)()
        self._check(self.L.fdtd_hip_multigpu_stats(self.h, C.byref(out), 1 if reset else 0), "stats")
        keys = ("slabs", "pairs", "single_steps", "issue_us_per_op", "wait_us_per_op", "wall_us_per_op", "split",
                "outer_thirds_first", "thread_per_slab", "shells_follow")
        return dict(zip(keys, list(out)))

    def close(self):
        if self.h:
            self.L.fdtd_hip_multigpu_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
