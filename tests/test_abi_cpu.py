"""CPU tests of the boundary: the library loads, exports every symbol the header declares,
keeps the reference's struct layout, and its host-side helpers (input generators, parity
metrics, argument checks) behave like the reference driver's.  No GPU compute is called."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fdtd_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:int|void|const char \*)\s*\**\s*([A-Za-z_]\w*)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_every_declared_symbol_is_exported(hiplib):
    names = declared_symbols()
    for must in ("Kernel_OpenACC", "Kernel_CUDA", "Kernel_CUDA_Optimized", "fdtd_hip_stencil_step",
                 "fdtd_hip_source_inject", "fdtd_hip_time_loop", "run_cuda_optimized",
                 "FDTD_SetRuntimeConfig"):
        assert must in names
    for n in names:
        assert hasattr(hiplib, n), f"libfdtd_hip.so does not export {n}"


def test_header_is_plain_c(tmp_path):
    """include/fdtd_abi.h is the boundary: it must compile as C11 and as C++17 on its own, at the default space order
    and with the padding macro set the way the reference driver sets STENCIL_ORDER (main.cpp:27-32)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    prog = tmp_path / "use_abi.c"
    prog.write_text('#include "fdtd_abi.h"\n'
                    'int halo(void) { struct fdtd_geom g = {0}; struct profiler t = {0, 0}; (void)g; (void)t; return FDTD_HALO; }\n')
    inc = os.path.join(ROOT, "include")
    for extra in ([], ["-DFDTD_STENCIL_ORDER=8"]):
        subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-c", "-I", inc, str(prog), "-o",
                        str(tmp_path / "a.o")] + extra, check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-x", "c++", "-c", "-I", inc, str(prog), "-o",
                    str(tmp_path / "b.o")], check=True)


def test_struct_layout_matches_reference(pkg):
    # main.cpp:35-50: nine pointer-sized fields, two doubles
    assert C.sizeof(pkg.Dataobj) == 72
    assert pkg.Dataobj.data.offset == 0 and pkg.Dataobj.size.offset == 8 and pkg.Dataobj.nbytes.offset == 16
    assert C.sizeof(pkg.Profiler) == 16


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg.abi, "_lib", None)
    monkeypatch.setattr(pkg.abi, "lib_path", lambda: "/nonexistent/libfdtd_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.abi.lib()


@pytest.mark.parametrize("n,nsrc,variant,expect", [
    ((512, 512, 512), 1, "auto", 1),      # full tiles of 8 rows x 512 cells
    ((1024, 1024, 1024), 1, "auto", 1),   # two z-tiles
    ((768, 768, 768), 1, "auto", 1),      # three z-tiles of 256
    ((500, 500, 500), 1, "auto", 1),      # masked tiles, 97 % full
    ((640, 640, 640), 1, "auto", 1),      # 83 %
    ((384, 384, 384), 1, "auto", 1),      # 75 %
        ((256, 256, 256), 1, "auto", 1),      # rows of 256 cells
    ((260, 260, 260), 1, "auto", 1),      # tiles half empty
    ((1024, 256, 68), 1, "auto", 0),      # 27 %: left to the marching kernel ...
    ((1024, 256, 68), 1, "march2", 1),    # ... unless asked for
        ((128, 128, 512), 1, "auto", 0),      # fewer than 16 M cells
    ((64, 512, 512), 1, "auto", 1),
    ((20, 512, 512), 1, "auto", 0),       # fewer than 32 planes
    ((20, 512, 512), 1, "march2", 1),
    ((64, 64, 516), 1, "march2", 1),      # a row ending on the first float4 of a wave
    ((64, 64, 514), 1, "march2", 0),      # rows that are not whole float4s
    ((64, 7, 512), 1, "march2", 1),       # fewer than 8 rows: one masked tile
    ((512, 512, 512), 27, "auto", 1),     # more than 8 sources: fused as well (scatter and repair passes behind the sweep)
    ((512, 512, 512), 8, "auto", 1),
    ((512, 512, 512), 1, "march", 0),     # a single-step variant by name
])
def test_which_boxes_take_the_fused_sweep(pkg, hiplib, n, nsrc, variant, expect):
    """fdtd_hip_two_steps_fuses: the host-side rule for which boxes fdtd_hip_two_steps runs as one fused
    sweep (pure host logic, no device call)."""
    g = pkg.Geom(n[0] + 8, n[1] + 8, n[2] + 8, 0, n[0] - 1, 0, n[1] - 1, 0, n[2] - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
    assert hiplib.fdtd_hip_two_steps_fuses(C.byref(g), nsrc, pkg.VARIANTS[variant]) == expect


def test_fused_sweep_switch_and_unpadded_arrays(pkg, hiplib, monkeypatch):
    n = (512, 512, 512)
    g = pkg.Geom(n[0] + 8, n[1] + 8, n[2] + 8, 0, n[0] - 1, 0, n[1] - 1, 0, n[2] - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
    # arrays with less than 4 cells of padding behind the box: the sweep would read beyond them
    tight = pkg.Geom(n[0] + 8, n[1] + 8, n[2] + 4, 0, n[0] - 1, 0, n[1] - 1, 0, n[2] - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
    assert hiplib.fdtd_hip_two_steps_fuses(C.byref(tight), 1, pkg.VARIANTS["auto"]) == 0
    monkeypatch.setenv("FDTD_FUSE2", "0")
    assert hiplib.fdtd_hip_two_steps_fuses(C.byref(g), 1, pkg.VARIANTS["auto"]) == 0


def test_ricker_generator(pkg):
    src = pkg.fill_ricker(50, 3)
    assert src.shape == (50, 3) and (src[:, 0] == src[:, 2]).all()
    f0, dt = np.float32(10.0), np.float32(1e-3)
    for t in (0, 7, 49):
        ts = np.float32(np.float32(t) * dt) - np.float32(1.0) / f0
        a = np.float32(math.pi) * np.float32(math.pi) * f0 * f0 * ts * ts
        want = (1.0 - 2.0 * float(a)) * math.exp(-float(a))
        assert float(src[t, 0]) == pytest.approx(want, rel=2e-6)


def test_source_coords_generator(pkg):
    n = 32
    c = pkg.fill_source_coords(30, n, n, n)
    L = np.float32(n - 1) * np.float32(0.1)
    assert np.allclose(c[0], [0.25 * L] * 3, rtol=1e-6)
    assert np.allclose(c[1], [0.25 * L, 0.25 * L, 0.5 * L], rtol=1e-6)   # z varies fastest
    assert np.allclose(c[26], [0.75 * L] * 3, rtol=1e-6)
    assert np.allclose(c[27:], 0.5 * L, rtol=1e-6)                        # extras at the centre
    # the benchmark source sits at grid coordinate 0.25*(n-1) = 7.75
    assert np.float32(c[0, 0]) / np.float32(0.1) == pytest.approx(7.75, rel=1e-6)


def test_correctness_inputs(pkg):
    u, m = pkg.fill_correctness_inputs(8)
    assert (m == 1.5).all() and (u[2] == 0).all() and np.array_equal(u[0], u[1])
    flat = u[0].reshape(-1)
    for i in (0, 5, 4095):
        want = math.sin(float(np.float32(np.float32(i) * np.float32(0.001)))) * 10.0 + 100.0
        assert float(flat[i]) == pytest.approx(want, rel=1e-6)


def test_parity_metrics_match_oracle_copy(pkg):
    rng = np.random.default_rng(5)
    ref = rng.normal(size=10000).astype(np.float32) * 100
    test = ref + rng.normal(size=10000).astype(np.float32) * 1e-3
    test[17] = np.nan
    test[99] = np.inf
    ref[5] = 0.0
    a = pkg.compare_fields(test, ref)
    b = pkg.Parity()
    ob.oracle().oracle_compare_fields(test.ctypes.data, ref.ctypes.data, test.size, C.byref(b))
    assert (a.max_abs, a.max_rel, a.l2, a.nan_count, a.inf_count) == \
           (b.max_abs, b.max_rel, b.l2, b.nan_count, b.inf_count)
    assert a.nan_count == 1 and a.inf_count == 1
    ok = np.isfinite(test)
    d = np.abs(test[ok] - ref[ok]).astype(np.float64)
    assert a.max_abs == pytest.approx(d.max())
    assert a.l2 == pytest.approx(np.sqrt((d ** 2).sum() / ((ref[ok].astype(np.float64) ** 2).sum() + 1e-30)), rel=1e-6)


def test_operator_rejects_bad_shapes_before_touching_the_gpu(pkg, hiplib):
    """Shape errors return hipErrorInvalidValue (1), like cuda_optimized.cu:346-347."""
    u = np.zeros((3, 12, 12, 12), np.float32)
    m_bad = np.zeros((12, 12, 11), np.float32)
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m_bad, None, None, u, time_M=3)
    assert rc == 1
    u2 = np.zeros((2, 12, 12, 12), np.float32)
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA, np.zeros((12, 12, 12), np.float32), None, None, u2, time_M=3)
    assert rc == 1
    # sources requested but no source table
    m = np.ones((12, 12, 12), np.float32)
    rc, _ = pkg.run_operator(hiplib.Kernel_OpenACC, m, None, None, u, time_M=3, p_src=(0, 0))
    assert rc == 1


def test_no_vector_write_directly_behind_a_wide_buffer_store(tmp_path):
    """A 16-byte buffer store whose data registers are overwritten by the very next instruction lost data on MI355X when
    the store's scalar-offset field held a register (the compiler leaves no wait state there).  The two-step sweep is the
    only source with buffer stores; its stores keep that field at zero, and its assembly must not contain such a pair."""
    import re
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "accelerated-3d-acoustic-fdtd-kernel_amd", "csrc")
    for name in os.listdir(csrc):
        text = open(os.path.join(csrc, name)).read()
        if name != "fdtd_march2.hip":
            assert "raw_buffer_store" not in text, name
        else:
            # every store goes through st4, which passes an immediate zero as the scalar offset
            calls = re.findall(r"__builtin_amdgcn_raw_buffer_store_b128\(([^;]*)\);", text)
            assert len(calls) == 1 and re.search(r",\s*0,\s*NT \? 2 : 0$", calls[0].strip()), calls
    hipcc = shutil.which("hipcc")
    if hipcc is None:
        pytest.skip("no hipcc")
    asm = tmp_path / "march2.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc",
                    "-S", "--cuda-device-only", "-o", str(asm), os.path.join(csrc, "fdtd_march2.hip")], check=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_store_hazard.py"), str(asm)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
