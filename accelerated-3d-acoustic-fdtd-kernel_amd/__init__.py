"""MI355X-native 3D acoustic FDTD engine: Python host mirror of the C ABI.

The compute lives in ``libfdtd_hip.so`` (hand-written gfx950 kernels behind the
Devito-style operator ABI of the reference, main.cpp:53-72).  This package only
binds that library with ctypes and adds the multi-GPU slab propagator; there is no
CPU fallback — importing :mod:`abi` raises if the library is missing.
"""
from . import abi  # noqa: F401
from . import slab  # noqa: F401
from .abi import (  # noqa: F401
    HALO,
    WARMUP_STEPS,
    Dataobj,
    Geom,
    MultiGpuSession,
    Parity,
    Profiler,
    VARIANTS,
    compare_fields,
    fill_correctness_inputs,
    fill_ricker,
    fill_source_coords,
    lib,
    lib_path,
    make_dataobj,
    run_forward,
    run_operator,
    standard_inputs,
)
