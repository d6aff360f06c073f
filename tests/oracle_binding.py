"""ctypes binding of the CPU oracle (oracle/libfdtd_oracle.so) and of the compiled
reference translation unit (oracle/_ref/libref_openacc.so) — test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from __graft_entry__ import load_package

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libfdtd_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_openacc.so")

_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.run(["make", "-C", ORACLE_DIR, "libfdtd_oracle.so"], check=True)
        pkg = load_package()
        L = C.CDLL(ORACLE_SO)
        L.Oracle_Kernel_OpenACC.argtypes = pkg.abi.OPERATOR_ARGTYPES
        L.Oracle_Kernel_OpenACC.restype = C.c_int
        L.oracle_stencil_step.argtypes = ([C.c_void_p, C.c_void_p] + [C.c_long] * 3 + [C.c_int] * 6
                                          + [C.c_float] * 4 + [C.c_int] * 3)
        L.oracle_stencil_step.restype = None
        L.oracle_source_inject.argtypes = ([C.c_void_p] * 4 + [C.c_long] * 3 + [C.c_int] * 6
                                           + [C.c_float] * 6 + [C.c_int] * 3 + [C.c_long] * 2
                                           + [C.c_int])
        L.oracle_source_inject.restype = None
        L.oracle_source_inject_slab.argtypes = ([C.c_void_p] * 4 + [C.c_long] * 3 + [C.c_int] * 4
                                                + [C.c_float] * 6 + [C.c_int] * 3 + [C.c_long] * 2
                                                + [C.c_int] * 4)
        L.oracle_source_inject_slab.restype = None
        L.oracle_source_inject_slab_mode.argtypes = L.oracle_source_inject_slab.argtypes + [C.c_int] * 3
        L.oracle_source_inject_slab_mode.restype = None
        L.oracle_stencil_step_order.argtypes = ([C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_long] * 3
                                                + [C.c_int] * 6 + [C.c_float] * 4 + [C.c_int] * 3)
        L.oracle_stencil_step_order.restype = None
        L.oracle_source_inject_slab_order.argtypes = ([C.c_int] + [C.c_void_p] * 4 + [C.c_long] * 3 + [C.c_int] * 4
                                                      + [C.c_float] * 6 + [C.c_int] * 3 + [C.c_long] * 2
                                                      + [C.c_int] * 4)
        L.oracle_source_inject_slab_order.restype = None
        L.Oracle_Kernel_SpaceOrder.argtypes = pkg.abi.OPERATOR_ARGTYPES + [C.c_int]
        L.Oracle_Kernel_SpaceOrder.restype = C.c_int
        L.Oracle_Kernel_Forward.argtypes = pkg.abi.FORWARD_ARGTYPES
        L.Oracle_Kernel_Forward.restype = C.c_int
        L.Oracle_Kernel_ForwardOrder.argtypes = pkg.abi.FORWARD_ARGTYPES + [C.c_int]
        L.Oracle_Kernel_ForwardOrder.restype = C.c_int
        L.oracle_compare_fields.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t,
                                            C.POINTER(pkg.Parity)]
        L.oracle_compare_fields.restype = None
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The reference's own openacc.cpp compiled for the host (serial)."""
    global _ref
    if _ref is None:
        pkg = load_package()
        L = C.CDLL(REF_SO)
        L.Kernel_OpenACC.argtypes = pkg.abi.OPERATOR_ARGTYPES
        L.Kernel_OpenACC.restype = C.c_int
        _ref = L
    return _ref


def run_oracle(m, src, coords, u, **kw):
    """Oracle_Kernel_OpenACC with the same keyword interface as pkg.run_operator."""
    pkg = load_package()
    rc, prof = pkg.run_operator(oracle().Oracle_Kernel_OpenACC, m, src, coords, u, **kw)
    assert rc == 0
    return prof


def run_oracle_order(order, m, src, coords, u, **kw):
    """The oracle for space order 4, 6, 8, 10 or 12 (arrays padded by `order` cells per side); keyword interface of
    pkg.run_operator except that `bounds` is required when the padding is not 4."""
    pkg = load_package()
    fn = oracle().Oracle_Kernel_SpaceOrder

    def call(*args):
        return fn(*args, order)
    rc, prof = pkg.run_operator(call, m, src, coords, u, **kw)
    assert rc == 0
    return prof


def forward_oracle_order(order):
    """Oracle_Kernel_ForwardOrder with the order bound: a callable with the argument list of Kernel_HIP_Forward."""
    fn = oracle().Oracle_Kernel_ForwardOrder

    def call(*args):
        return fn(*args, order)
    return call


def run_ref(m, src, coords, u, **kw):
    pkg = load_package()
    rc, prof = pkg.run_operator(ref().Kernel_OpenACC, m, src, coords, u, **kw)
    assert rc == 0
    return prof


def oracle_stencil_step(m, u, bounds, dt, h, t0, t1, t2):
    nxp, nyp, nzp = u.shape[1:]
    oracle().oracle_stencil_step(m.ctypes.data, u.ctypes.data, nxp, nyp, nzp, *bounds, dt, *h,
                                 t0, t1, t2)


def oracle_source_inject(m, src, coords, u, bounds, h, o, p_src, time, t2):
    nxp, nyp, nzp = u.shape[1:]
    oracle().oracle_source_inject(m.ctypes.data, src.ctypes.data, coords.ctypes.data, u.ctypes.data,
                                  nxp, nyp, nzp, *bounds, *h, *o, p_src[0], p_src[1], time,
                                  src.shape[1], coords.shape[1], t2)


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32),
                          np.ascontiguousarray(b).view(np.uint32))
