"""GPU parity tests: the HIP operators, called through the C ABI, against the CPU oracle and the
golden fixtures.  fp32 bar: bit-exact (same IEEE operations in the same order as the reference's
serial host code); the north-star tolerance L2 < 1e-4 (main.cpp:593) is asserted as well.
Only the atomic form of the scatter (more than 128 sources, or forced) with coincident targets
depends on arrival order: L2 < 1e-6 there."""
import ctypes as C
import os

import numpy as np
import pytest

import cases
import oracle_binding as ob
from test_oracle import check_against_golden

pytestmark = pytest.mark.gpu

OPERATORS = ["Kernel_OpenACC", "Kernel_CUDA", "Kernel_CUDA_Optimized"]
L2_GATE = 1e-4  # north-star parity gate, L2 as main.cpp:593 defines it


@pytest.mark.parametrize("op", OPERATORS)
@pytest.mark.parametrize("name", list(cases.CASES))
def test_operator_matches_golden(pkg, hiplib, op, name, monkeypatch):
    # the fast operator picks its kernel by problem size; force the marching kernel here so that
    # every case whose shape allows it runs through it (others fall back to the naive kernel)
    monkeypatch.setenv("FDTD_MARCH_MIN_CELLS", "0")
    monkeypatch.setenv("FDTD_FUSE2_MIN_CELLS", "0")      # ... and the two-step sweep on the small fused_z* shapes
    if name.startswith("fused_z256"):
        # rows of 256 cells take the two-steps-per-sweep kernel only on explicit request
        monkeypatch.setenv("FDTD_VARIANT", "march2")
    c = cases.build(name, pkg)
    fused_before = hiplib.fdtd_hip_fused_launch_count()
    rc, prof = pkg.run_operator(getattr(hiplib, op), c["m"], c["src"], c["coords"], c["u"], **c["kw"])
    assert rc == 0
    # the fused_z* cases are there for the two-steps-per-sweep kernel: make sure it is what ran
    took_fused = hiplib.fdtd_hip_fused_launch_count() > fused_before
    assert took_fused == (name.startswith("fused_z") and op == "Kernel_CUDA_Optimized")
    # up to 128 sources the scatter is ordered like the reference's serial loop, so even coincident
    # scatter targets (ricker30_n24_overlap) are bit-exact
    check_against_golden(name, c["u"], exact=True)
    time_M = c["kw"].get("time_M", None if c["src"] is None else c["src"].shape[0] - 1)
    n_steps = time_M - c["kw"].get("time_m", 0) + 1
    assert (prof.section0 > 0) == (n_steps > pkg.WARMUP_STEPS)  # warm-up steps are not timed


@pytest.mark.parametrize("variant,ry,nw", [("flat", 0, 0), ("naive", 0, 0), ("vec", 0, 0), ("tile3d", 0, 0), ("rows", 0, 0),
                                           ("march", 2, 8), ("march", 4, 4), ("march", 2, 4), ("march", 1, 8), ("march", 1, 16)])
@pytest.mark.parametrize("nt", ["0", "1"])
def test_variants_and_tile_shapes(pkg, hiplib, variant, ry, nw, nt, monkeypatch):
    """Every stencil variant (and every rows-per-wave x waves-per-workgroup shape of the marching
    kernel, with and without non-temporal hints) gives the oracle's bits on shapes that exercise
    partial tiles in y and z, sub-ranges and full tiles."""
    if nt == "0" and variant in ("flat", "naive", "tile3d", "rows"):
        pytest.skip("no non-temporal path in this variant")
    monkeypatch.setenv("FDTD_VARIANT", variant)
    monkeypatch.setenv("FDTD_NT", nt)
    if variant == "march":
        monkeypatch.setenv("FDTD_MARCH_RY", str(ry))
        monkeypatch.setenv("FDTD_MARCH_NW", str(nw))
    for name in ("tile_edges_100", "noncubic_aniso", "subrange_aligned", "odd_sizes", "full_tiles_256"):
        c = cases.build(name, pkg)
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, c["m"], c["src"], c["coords"], c["u"], **c["kw"])
        assert rc == 0
        check_against_golden(name, c["u"], exact=True)


@pytest.mark.parametrize("lpr", ["16", "32", "64"])
def test_march_row_group_width(pkg, hiplib, lpr, monkeypatch):
    """All lane layouts of the marching kernel (a wave spanning one 256-cell row group, two 128-cell
    or four 64-cell row groups side by side) on every fixture case, fp32 storage."""
    monkeypatch.setenv("FDTD_VARIANT", "march")
    monkeypatch.setenv("FDTD_MARCH_LPR", lpr)
    for name in cases.CASES:
        c = cases.build(name, pkg)
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, c["m"], c["src"], c["coords"], c["u"], **c["kw"])
        assert rc == 0
        check_against_golden(name, c["u"], exact=True)


@pytest.mark.parametrize("xchunk", ["1", "3", "8", "1000"])
def test_march_x_chunking(pkg, hiplib, xchunk, monkeypatch):
    monkeypatch.setenv("FDTD_VARIANT", "march")
    monkeypatch.setenv("FDTD_XCHUNK", xchunk)
    c = cases.build("noncubic_aniso", pkg)
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, c["m"], c["src"], c["coords"], c["u"], **c["kw"])
    assert rc == 0
    check_against_golden("noncubic_aniso", c["u"], exact=True)


@pytest.mark.parametrize("n", [128, 256])
def test_reference_sweep_sizes_correctness_inputs(pkg, hiplib, n):
    """The driver's parity run (main.cpp:511-652) at the sizes too large for fixtures: HIP vs the
    oracle computed here, all three operators, metrics as the driver prints them."""
    u0, m = pkg.fill_correctness_inputs(n)
    u_ref = u0.copy()
    ob.run_oracle(m, None, None, u_ref, time_M=49, h=(1.0, 1.0, 1.0))
    for op in OPERATORS:
        u = u0.copy()
        rc, _ = pkg.run_operator(getattr(hiplib, op), m, None, None, u, time_M=49, h=(1.0, 1.0, 1.0))
        assert rc == 0
        p = pkg.compare_fields(u, u_ref)
        assert p.nan_count == 0 and p.inf_count == 0
        assert p.l2 < L2_GATE
        assert p.max_abs < 1e-4          # the driver's own PASS criterion (main.cpp:603-604)
        assert ob.bits_equal(u, u_ref)   # and in fact not a bit differs


def test_reference_sweep_512(pkg, hiplib):
    """512^3, 50 steps, both input sets, fast path only (the full-size case of the sweep)."""
    n = 512
    u, m = pkg.fill_correctness_inputs(n)
    u_ref = u.copy()
    ob.run_oracle(m, None, None, u_ref, time_M=49, h=(1.0, 1.0, 1.0))
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, None, None, u, time_M=49, h=(1.0, 1.0, 1.0))
    assert rc == 0
    p = pkg.compare_fields(u, u_ref)
    assert p.l2 < L2_GATE and p.max_abs < 1e-4 and p.nan_count == 0
    assert ob.bits_equal(u, u_ref)
    del u, u_ref
    u, m, src, coords = pkg.standard_inputs(n, n, n, 50, 1)
    u_ref = u.copy()
    ob.run_oracle(m, src, coords, u_ref)
    rc, prof = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u)
    assert rc == 0 and prof.section0 > 0 and prof.section1 > 0
    assert ob.bits_equal(u, u_ref)
    assert f"{float(np.abs(u).max()):.1f}" == "0.1"   # main.cpp:481


def test_size_independent_properties(pkg, hiplib):
    """Size-independent properties at the benchmark size, no oracle needed:
    zero field + no sources stays zero (main.cpp:483-486); the update is linear in (u, src)
    for power-of-two scalings (exact in binary floating point away from the denormal range)."""
    n = 256  # property checks do not need 512^3; keep host time low
    u, m, src, coords = pkg.standard_inputs(n, n, n, 30, 1)
    z = np.zeros_like(u)
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, None, None, z, time_M=29)
    assert rc == 0 and not z.any()
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u)
    assert rc == 0
    u4 = np.zeros_like(u)
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src * np.float32(4.0), coords, u4)
    assert rc == 0
    big = np.abs(u) > 1e-20       # away from the denormal range scaling by 4 is exact
    assert np.array_equal(u4[big], (u * np.float32(4.0))[big])


@pytest.mark.parametrize("op", OPERATORS)
def test_impulse_mirror_symmetry(pkg, hiplib, op):
    """Single-source symmetry check (suggested, not implemented, in DEBUG.md:500-512): a source
    half-way between the two central nodes of every axis feeds 8 equal corners, and the update
    (openacc.cpp:159-164) only ever adds the two members of a mirror pair to each other, so the
    field stays bit-for-bit mirror symmetric about the centre in x, y and z.  192^3 takes the
    packed-rows kernel in Kernel_CUDA_Optimized."""
    n, nt = 192, 24
    h = 0.125                                       # power of two: (c - o)/h is exact
    u, m, src, _ = pkg.standard_inputs(n, n, n, nt, 1)
    coords = np.full((1, 3), (n / 2 - 0.5) * h, dtype=np.float32)
    rc, _ = pkg.run_operator(getattr(hiplib, op), m, src, coords, u, h=(h, h, h))
    assert rc == 0
    f = u[:, 4:-4, 4:-4, 4:-4]
    assert np.abs(f).max() > 0
    for axis in (1, 2, 3):
        assert np.array_equal(f, np.flip(f, axis=axis)), f"not mirror symmetric along axis {axis}"


@pytest.mark.parametrize("op", OPERATORS)
def test_atomic_scatter_path(pkg, hiplib, op, monkeypatch):
    """The atomic scatter: forced on the fixture cases (exact where no two corners share a cell,
    L2 < 1e-6 where they do), and taken by itself for 300 random sources."""
    monkeypatch.setenv("FDTD_INJECT_ATOMIC", "1")
    for name in ("ricker27_n32", "noncubic_aniso", "ricker30_n24_overlap"):
        c = cases.build(name, pkg)
        rc, _ = pkg.run_operator(getattr(hiplib, op), c["m"], c["src"], c["coords"], c["u"], **c["kw"])
        assert rc == 0
        if name in cases.ORDER_DEPENDENT:
            check_against_golden(name, c["u"], exact=False, l2_tol=1e-6)
        else:
            check_against_golden(name, c["u"], exact=True)
    monkeypatch.delenv("FDTD_INJECT_ATOMIC")
    rng = np.random.default_rng(77)
    n, nt, nsrc = 40, 8, 300
    u, m, _, _ = pkg.standard_inputs(n, n, n, nt, 1)
    src = rng.uniform(-5, 5, size=(nt, nsrc)).astype(np.float32)
    coords = (rng.uniform(0.0, 1.0, size=(nsrc, 3)) * (n - 1) * 0.1).astype(np.float32)
    u_ref = u.copy()
    ob.run_oracle(m, src, coords, u_ref)
    rc, _ = pkg.run_operator(getattr(hiplib, op), m, src, coords, u)
    assert rc == 0
    p = pkg.compare_fields(u, u_ref)
    assert p.nan_count == 0 and p.l2 < 1e-6


def test_determinism(pkg, hiplib):
    """Repeated runs return identical bytes (the reference's stability criterion,
    STABILITY_FIX_LOG.md:18,130)."""
    crcs = set()
    for _ in range(5):
        c = cases.build("ricker27_n32", pkg)
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, c["m"], c["src"], c["coords"], c["u"], **c["kw"])
        assert rc == 0
        crcs.add(tuple(cases.digest(c["u"])["crc32"]))
    assert len(crcs) == 1


def test_device_resident_time_loop_matches_operator(pkg, hiplib):
    """fdtd_hip_time_loop on device buffers == the host-buffer operator."""
    import torch
    c = cases.build("noncubic_aniso", pkg)
    u_d = torch.from_numpy(c["u"]).cuda()
    m_d = torch.from_numpy(c["m"]).cuda()
    s_d = torch.from_numpy(c["src"]).cuda()
    c_d = torch.from_numpy(c["coords"]).cuda()
    nxp, nyp, nzp = c["u"].shape[1:]
    h = c["kw"]["h"]
    g = pkg.Geom(nxp, nyp, nzp, 0, nxp - 9, 0, nyp - 9, 0, nzp - 9, 1e-3, h[0], h[1], h[2], 0, 0, 0)
    nt = c["src"].shape[0]
    torch.cuda.synchronize()
    rc = hiplib.fdtd_hip_time_loop(C.byref(g), m_d.data_ptr(), s_d.data_ptr(), c_d.data_ptr(),
                                   u_d.data_ptr(), 0, c["src"].shape[1] - 1, c["src"].shape[1], 3,
                                   0, nt - 1, pkg.VARIANTS["auto"], None, None)
    assert rc == 0
    check_against_golden("noncubic_aniso", u_d.cpu().numpy(), exact=True)


@pytest.mark.parametrize("shape,nsrc,variant", [((70, 64, 512), 3, "auto"), ((20, 16, 512), 2, "march2"),
                                                ((48, 64, 256), 2, "march2"), ((300, 256, 256), 1, "march2"),
                                                ((24, 8, 512), 0, "march2"), ((520, 512, 512), 1, "auto"),
                                                ((20, 16, 1024), 3, "march2"), ((70, 24, 1536), 2, "auto"),
                                                ((40, 32, 768), 2, "auto"), ((300, 1024, 1024), 2, "auto"),
                                                ((20, 20, 640), 2, "march2"), ((24, 13, 520), 2, "march2"),
                                                ((40, 100, 264), 3, "march2"), ((36, 9, 252), 2, "march2"),
                                                ((40, 44, 480), 2, "auto"), ((34, 500, 1000), 2, "auto"),
                                                ((40, 24, 512), 20, "auto"), ((36, 20, 1024), 11, "auto"),
                                                ((24, 8, 516), 2, "march2"), ((24, 8, 260), 2, "march2"),
                                                ((24, 16, 516), 3, "march2"), ((40, 100, 772), 2, "march2"),
                                                ((20, 24, 4), 1, "march2"), ((24, 5, 512), 2, "march2"),
                                                ((24, 1, 256), 1, "march2"), ((30, 3, 520), 2, "march2"),
                                                ((24, 2, 1028), 2, "march2"), ((30, 20, 516), 12, "march2"),
                                                ((24, 5, 260), 10, "march2"),
                                                ((24, 32, 384), 2, "march2"), ((24, 64, 640), 3, "march2"),
                                                ((24, 32, 768), 2, "march2"), ((20, 40, 896), 2, "march2"),
                                                ((24, 20, 388), 2, "march2"), ((24, 70, 1156), 12, "march2")])
@pytest.mark.parametrize("tile", ["", "128", "256", "mixed"])
def test_two_steps_fused_equals_two_single_steps(pkg, hiplib, shape, nsrc, variant, tile, monkeypatch):
    """fdtd_hip_two_steps, fused sweep against the two single sweeps it stands for (those are
    oracle-checked by the other tests), bit for bit, on random fields with different halos per level:
    one launch, forced fusion on a short x range, rows of 256 cells (fused on request only), no
    sources, an x range long enough to be cut into two launches (520 planes at 64 y-tiles), and rows
    of 1024, 1536 and 768 cells (two and three z-tiles of 512, three of 256; the cells around the cuts
    are redone by the seam pass); a full cross-section of 256 tiles in three launches; and cross-sections
    (384 = 256 + 128, 640 = 512 + 128, 768 = 512 + 256, 896 = 512 + 256 + 128, 388 and 1156 with a masked last tile: the
    mixed tiling, which the automatic choice takes for them), and cross-sections
    that do not fill their tiles (rows ending inside a z-tile, inside the first half of one, eight cells into
    one; fewer than 8 or 16 rows in the last y-tile, or in the whole box; rows that end on the first float4 of a wave, in the
    second wave of a 512-cell tile, in a z-tile of its own, in both tilings; rows of 4 cells), which run
    the masked variant of the kernel."""
    import torch
    monkeypatch.setenv("FDTD_FUSE2_MIN_CELLS", "0")   # the automatic choice would leave the small shapes alone
    if tile == "mixed":                               # rows cut into tiles of 512, then 256, then 128 cells
        monkeypatch.setenv("FDTD_FUSE2_MIXED", "2")
        monkeypatch.setenv("FDTD_FUSE2_MIN_FILL", "0")
    elif tile:                                        # every shape through tiles of 128 / 256 cells as well
        monkeypatch.setenv("FDTD_FUSE2_TILE", tile)
        monkeypatch.setenv("FDTD_FUSE2_MIN_FILL", "0")
    nx, ny, nz = shape
    gen = torch.Generator(device="cuda").manual_seed(1234)
    pshape = (nx + 8, ny + 8, nz + 8)
    lv = [torch.rand(pshape, device="cuda", generator=gen) * 2 - 1 for _ in range(4)]
    lv[1] = lv[0] + 0.01 * lv[1]
    m = torch.rand(pshape, device="cuda", generator=gen) * 2 + 1
    h = (0.9, 1.0, 1.1)
    g = pkg.Geom(*pshape, 0, nx - 1, 0, ny - 1, 0, nz - 1, 1e-3, *h, 0.0, 0.0, 0.0)
    src = coords = None
    if nsrc:
        src = (torch.rand((4, nsrc), device="cuda", generator=gen) * 100 - 50)
        ext = torch.tensor([(nx - 1) * h[0], (ny - 1) * h[1], (nz - 1) * h[2]], device="cuda")
        coords = (torch.rand((nsrc, 3), device="cuda", generator=gen) * 0.9 + 0.05) * ext
        coords[0] = torch.tensor([(nx - 1) * h[0] + 0.3 * h[0], 0.0, (nz - 1) * h[2]])   # corners in the halo layer
        if nz > 512 and nsrc > 1:
            coords[1, 2] = 511.5 * h[2]     # corners on either side of the cut between two z-tiles
        if nsrc > 8:                        # more sources than the sweep scatters itself
            coords[3] = coords[2] + torch.tensor([1.2 * h[0], 0.0, 0.6 * h[2]], device="cuda")   # overlapping stars
            coords[5] = coords[4]                                                                  # coincident corners
            coords[6] = torch.tensor([0.4 * h[0], 0.3 * h[1], 1.7 * h[2]], device="cuda")          # in the box's corner

    def run(var):
        cur, prev, n1, n2 = (t.clone() for t in lv)
        rc = hiplib.fdtd_hip_two_steps(C.byref(g), m.data_ptr(), src.data_ptr() if nsrc else None,
                                       coords.data_ptr() if nsrc else None, cur.data_ptr(), prev.data_ptr(),
                                       n1.data_ptr(), n2.data_ptr(), 1, 0, nsrc - 1, max(nsrc, 1), 3,
                                       pkg.VARIANTS[var], None)
        assert rc == 0
        torch.cuda.synchronize()
        return n1, n2

    before = hiplib.fdtd_hip_fused_launch_count()
    f1, f2 = run(variant)
    assert hiplib.fdtd_hip_fused_launch_count() > before
    s1, s2 = run("naive")
    assert torch.equal(f1.view(torch.int32), s1.view(torch.int32))
    assert torch.equal(f2.view(torch.int32), s2.view(torch.int32))
    assert not torch.equal(f2, lv[3])


@pytest.mark.parametrize("chunk_kb,threads", [("64", "1"), ("100", "3"), ("4096", "4")])
def test_operator_staged_transfers(pkg, hiplib, chunk_kb, threads, monkeypatch):
    """The operators move the caller's arrays through pinned bounce buffers when they are large;
    here every transfer is staged, in chunks that do not divide the arrays (64 KB and 100 KB chunks
    of a 2.8 MB field, the last one partial) and in a single chunk, with 1, 3 and 4 host threads."""
    monkeypatch.setenv("FDTD_STAGE_MIN_KB", "0")
    monkeypatch.setenv("FDTD_STAGE_CHUNK_KB", chunk_kb)
    monkeypatch.setenv("FDTD_STAGE_THREADS", threads)
    for name in ("noncubic_aniso", "ricker_n64"):
        c = cases.build(name, pkg)
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, c["m"], c["src"], c["coords"], c["u"], **c["kw"])
        assert rc == 0
        check_against_golden(name, c["u"], exact=True)


@pytest.mark.parametrize("lpr", ["16", "32", "64"])
def test_small_grid_packed_rows_kernel(pkg, hiplib, lpr, monkeypatch):
    """The one-launch-per-step loop with every lane packing of its float4-per-lane kernel (a wave spanning
    four rows of 64 cells, two of 128, one of 256) on every fixture case, and with the one-cell-per-thread
    kernel it replaces on shapes it does not take; the bytes are the fixture's."""
    monkeypatch.setenv("FDTD_ROWS_LPR", lpr)
    monkeypatch.setenv("FDTD_MARCH_MIN_CELLS", "2000000000")   # every case through the small-grid loop
    monkeypatch.setenv("FDTD_FUSE2", "0")
    for name in cases.CASES:
        c = cases.build(name, pkg)
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, c["m"], c["src"], c["coords"], c["u"], **c["kw"])
        assert rc == 0
        check_against_golden(name, c["u"], exact=True)
    monkeypatch.setenv("FDTD_SMALL_ROWS", "0")
    for name in ("ricker_n32", "noncubic_aniso", "fused_z512_ricker8"):
        c = cases.build(name, pkg)
        rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, c["m"], c["src"], c["coords"], c["u"], **c["kw"])
        assert rc == 0
        check_against_golden(name, c["u"], exact=True)


@pytest.mark.parametrize("name", list(cases.CASES))
def test_small_grid_time_loop(pkg, hiplib, name):
    """Default settings: grids below the marching kernel's threshold go through the one-launch-per-step
    loop (sweep and scatter in one kernel) unless a source corner lands in the halo layer or the fused
    two-step sweep takes the case; the bytes are the fixture's either way, timers cover the timed window."""
    c = cases.build(name, pkg)
    before = hiplib.fdtd_hip_small_step_count()
    rc, prof = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, c["m"], c["src"], c["coords"], c["u"], **c["kw"])
    assert rc == 0
    check_against_golden(name, c["u"], exact=True)
    kind, p = cases.CASES[name]
    time_M = c["kw"].get("time_M", None if c["src"] is None else c["src"].shape[0] - 1)
    n_steps = time_M - c["kw"].get("time_m", 0) + 1
    bounds = c["kw"].get("bounds")
    empty = n_steps <= 0 or (bounds is not None and bounds[1] < bounds[0])
    inside = True   # every source corner is a cell of the updated box
    if c["src"] is not None:
        n = c["u"].shape[1:]
        b = bounds if bounds is not None else (0, n[0] - 9, 0, n[1] - 9, 0, n[2] - 9)
        h = np.asarray(c["kw"].get("h", (0.1, 0.1, 0.1)), dtype=np.float32)
        lo, hi = c["kw"].get("p_src", (0, c["coords"].shape[0] - 1))
        pos = np.floor(c["coords"][lo:hi + 1].astype(np.float32) / h).astype(int)
        for d in range(3):
            inside &= bool(np.all(pos[:, d] >= b[2 * d]) and np.all(pos[:, d] + 1 <= b[2 * d + 1]))
    expected = inside and not (empty or p.get("nsrc", 0) > 16)   # (all fixture grids are below 16 M cells)
    assert hiplib.fdtd_hip_small_step_count() - before == (n_steps if expected else 0), name
    assert (prof.section0 > 0) == (n_steps > pkg.WARMUP_STEPS)


def test_two_steps_rejects_overlapping_levels(pkg, hiplib):
    """fdtd_hip_two_steps writes two levels while it reads two others: any overlap between a written
    level and another argument is refused (hipErrorInvalidValue = 1), nothing is launched."""
    import torch
    n = (20, 16, 512)
    pshape = tuple(v + 8 for v in n)
    buf = torch.zeros((5,) + pshape, device="cuda")
    m = torch.full(pshape, 1.5, device="cuda")
    g = pkg.Geom(*pshape, 0, n[0] - 1, 0, n[1] - 1, 0, n[2] - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
    P = [buf[i].data_ptr() for i in range(5)]

    def call(cur, prev, n1, n2):
        return hiplib.fdtd_hip_two_steps(C.byref(g), m.data_ptr(), None, None, cur, prev, n1, n2, 0, 0, -1, 1, 3,
                                         pkg.VARIANTS["auto"], None)

    assert call(P[0], P[1], P[2], P[3]) == 0
    assert call(P[0], P[1], P[2], P[2]) == 1      # both results in one buffer
    assert call(P[0], P[1], P[2], P[1]) == 1      # u(t+2) over u(t-1): the reference's ring, not possible here
    assert call(P[0], P[1], P[0], P[3]) == 1
    assert call(P[0], P[1], P[2], P[2] + 4096) == 1   # partial overlap
    assert call(P[0], P[1], m.data_ptr(), P[3]) == 1
    torch.cuda.synchronize()


@pytest.mark.parametrize("name", list(cases.CASES))
@pytest.mark.parametrize("tile", ["", "128", "mixed"])
def test_forced_fused_sweep_on_every_case(pkg, hiplib, name, tile, monkeypatch):
    """FDTD_VARIANT=march2: every fixture case whose box has rows of whole float4s goes
    through the two-step sweep (masked tiles for all but the fused_z* shapes; with 27 and 30 sources the
    scatter and repair passes behind the sweep), the others through the ordinary loop; the bytes are the
    fixture's either way."""
    monkeypatch.setenv("FDTD_VARIANT", "march2")
    if tile == "mixed":
        monkeypatch.setenv("FDTD_FUSE2_MIXED", "2")
    elif tile:
        monkeypatch.setenv("FDTD_FUSE2_TILE", tile)
    c = cases.build(name, pkg)
    before = hiplib.fdtd_hip_fused_launch_count()
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, c["m"], c["src"], c["coords"], c["u"], **c["kw"])
    assert rc == 0
    check_against_golden(name, c["u"], exact=True)
    n = c["u"].shape[1:]
    b = c["kw"].get("bounds") or (0, n[0] - 9, 0, n[1] - 9, 0, n[2] - 9)
    ex, ey, ez = b[1] - b[0] + 1, b[3] - b[2] + 1, b[5] - b[4] + 1
    time_M = c["kw"].get("time_M", None if c["src"] is None else c["src"].shape[0] - 1)
    n_steps = time_M - c["kw"].get("time_m", 0) + 1
    lo, hi = c["kw"].get("p_src", (0, (c["coords"].shape[0] if c["coords"] is not None else 0) - 1))
    fusable = (ex >= 1 and ey >= 1 and ez % 4 == 0 and b[4] % 4 == 0 and n_steps >= 4
               and b[5] + 8 <= n[2] - 1 and b[3] + 8 <= n[1] - 1 and b[1] + 8 <= n[0] - 1)
    assert (hiplib.fdtd_hip_fused_launch_count() > before) == fusable, name


def test_t_fuse_knob_requests_fused_sweep(pkg, hiplib):
    """FDTD_SetRuntimeConfig(use_tc, t_fuse, nfields), the driver's hook (main.cpp:84,272): with
    t_fuse = 2 the optimized operator fuses even an x range of 20 planes, with 1 it does not; both
    give the oracle's bits."""
    n = (20, 16, 512)
    u_ref, m, src, coords = pkg.standard_inputs(*n, 12, 1)
    ob.run_oracle(m, src, coords, u_ref)
    try:
        for t_fuse, expect in ((2, True), (1, False)):
            hiplib.FDTD_SetRuntimeConfig(0, t_fuse, 1)
            before = hiplib.fdtd_hip_fused_launch_count()
            u = np.zeros_like(u_ref)
            rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u)
            assert rc == 0
            assert (hiplib.fdtd_hip_fused_launch_count() > before) == expect
            assert ob.bits_equal(u, u_ref)
    finally:
        hiplib.FDTD_SetRuntimeConfig(0, 1, 1)


def test_fused_time_loop_switch(pkg, hiplib, monkeypatch):
    """The time loop with the two-step sweep, with the small-grid loop and with neither gives the same bytes as the oracle: Ricker run
    at 64 x 64 x 512, 50 steps (5 untimed single steps, 22 pairs, 1 single step), timers filled."""
    n = (64, 64, 512)
    u_ref, m, src, coords = pkg.standard_inputs(*n, 50, 1)
    ob.run_oracle(m, src, coords, u_ref)
    monkeypatch.setenv("FDTD_FUSE2_MIN_CELLS", "0")
    for fuse, small in (("1", "1"), ("0", "1"), ("0", "0")):
        monkeypatch.setenv("FDTD_FUSE2", fuse)
        monkeypatch.setenv("FDTD_SMALL_FUSED", small)   # 2 M cells: without the fused sweep, the small-grid loop
        u = np.zeros_like(u_ref)
        rc, prof = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u)
        assert rc == 0
        assert ob.bits_equal(u, u_ref), f"FDTD_FUSE2={fuse} FDTD_SMALL_FUSED={small}"
        assert prof.section0 > 0 and (prof.section1 > 0 or (fuse, small) == ("0", "1"))


def test_run_wrappers(pkg, hiplib):
    """run_openacc / run_cuda / run_cuda_optimized with the standard benchmark inputs."""
    n, nt = 32, 50
    want = ob  # noqa: F841
    u_ref, m, src, coords = pkg.standard_inputs(n, n, n, nt, 1)
    ob.run_oracle(m, src, coords, u_ref)
    for name in ("run_openacc", "run_cuda", "run_cuda_optimized"):
        out = np.empty_like(u_ref)
        mx = C.c_float(0)
        prof = pkg.Profiler(0, 0)
        rc = getattr(hiplib, name)(n, n, n, nt, 1, out.ctypes.data, C.byref(mx), C.byref(prof))
        assert rc == 0
        assert ob.bits_equal(out, u_ref)
        assert mx.value == float(np.abs(u_ref).max())
        assert prof.section0 > 0


def fp16_emulation(ob, u, m, src, coords, n, nt, h):
    """The oracle stepped with a round-to-binary16 after every store (stencil and scatter)."""
    ue = u.copy()
    bounds = (0, n[0] - 1, 0, n[1] - 1, 0, n[2] - 1)
    hh = tuple(np.float32(v) for v in h)
    for t in range(nt):
        t0, t1, t2 = t % 3, (t + 2) % 3, (t + 1) % 3
        ob.oracle_stencil_step(m, ue, bounds, np.float32(1e-3), hh, t0, t1, t2)
        ue[t2] = ue[t2].astype(np.float16).astype(np.float32)
        if src is not None:
            ob.oracle_source_inject(m, src, coords, ue, bounds, hh, (0.0, 0.0, 0.0), (0, src.shape[1] - 1), t, t2)
            ue[t2] = ue[t2].astype(np.float16).astype(np.float32)
    return ue


@pytest.mark.parametrize("shape,march,ry,nw", [((64, 64, 64), False, 0, 0),      # small grid: one-cell-per-thread kernel
                                               ((12, 40, 200), True, 1, 8),      # marching kernel, partial tiles (half-width groups)
                                               ((12, 40, 328), True, 1, 8),      # partial tiles, full-width groups
                                               ((10, 16, 512), True, 1, 8),      # marching kernel, full tiles
                                               ((9, 36, 264), True, 2, 4)])
def test_fp16_storage_mode(pkg, hiplib, shape, march, ry, nw, monkeypatch):
    """fp16 storage / fp32 compute on the Ricker benchmark inputs (|u| <= 0.12).  Checked two
    ways: bit-exact against the oracle stepped with a round-to-binary16 after every store, and
    the L2 distance to the fp32 oracle reported against its own tolerance (fp16 keeps 11
    significant bits; the leapfrog recursion feeds the rounding back every step)."""
    nt = 30
    monkeypatch.setenv("FDTD_MARCH_MIN_CELLS", "0" if march else "2000000000")
    if march:
        monkeypatch.setenv("FDTD_MARCH_RY", str(ry))
        monkeypatch.setenv("FDTD_MARCH_NW", str(nw))
    u, m, src, coords = pkg.standard_inputs(*shape, nt, 2)
    u32 = u.copy()
    ob.run_oracle(m, src, coords, u32)
    ue = fp16_emulation(ob, u, m, src, coords, shape, nt, (0.1, 0.1, 0.1))
    rc, prof = pkg.run_operator(hiplib.Kernel_HIP_FP16Storage, m, src, coords, u)
    assert rc == 0 and prof.section0 > 0
    assert ob.bits_equal(u, ue)
    p = pkg.compare_fields(u, u32)
    print(f"fp16-storage vs fp32 oracle {shape}: L2={p.l2:.3e} max_abs={p.max_abs:.3e}")
    assert p.nan_count == 0 and p.l2 < 5e-2


def test_fp16_storage_random_field(pkg, hiplib, monkeypatch):
    """fp16 storage on a non-trivial wavefield with varying m and anisotropic spacing: bit-exact
    against the emulation (values pre-rounded to binary16 so that the upload is lossless)."""
    monkeypatch.setenv("FDTD_MARCH_MIN_CELLS", "0")
    c = cases.build("tile_edges_100", pkg)          # 12 x 100 x 260: z extent 260 is not a multiple of 8
    c2 = cases.build("full_tiles_256", pkg)         # 11 x 32 x 256
    c3 = cases.build("odd_sizes", pkg)              # 17 x 33 x 65: a level has an odd number of cells, so the
                                                    # second level starts on a 2-byte boundary (scatter words)
    for case in (c, c2, c3):
        u = case["u"].astype(np.float16).astype(np.float32)
        nt = case["src"].shape[0]
        shape = tuple(d - 8 for d in u.shape[1:])
        ue = fp16_emulation(ob, u, case["m"], case["src"], case["coords"], shape, nt, case["kw"]["h"])
        rc, _ = pkg.run_operator(hiplib.Kernel_HIP_FP16Storage, case["m"], case["src"], case["coords"], u, **case["kw"])
        assert rc == 0
        assert ob.bits_equal(u, ue)


def test_fp16_storage_512_cubed(pkg, hiplib):
    """Config 4 at its full size: 512^3, fp16 storage, the Ricker benchmark inputs, a dozen steps, bit for
    bit against the oracle stepped with a round-to-binary16 after every store (the pinned fp32 oracle plus
    numpy's IEEE rounding; there is no reference kernel that time-steps fp16 state)."""
    n, nt = (512, 512, 512), 12
    u, m, src, coords = pkg.standard_inputs(*n, nt, 1)
    ue = fp16_emulation(ob, u, m, src, coords, n, nt, (0.1, 0.1, 0.1))
    rc, prof = pkg.run_operator(hiplib.Kernel_HIP_FP16Storage, m, src, coords, u)
    assert rc == 0 and prof.section0 > 0
    assert float(np.abs(u).max()) > 0
    assert ob.bits_equal(u, ue)


@pytest.mark.parametrize("shape,nsrc,what", [((32, 512, 1024), 3, "two z-tiles of 512 cells: seam pass"),
                                             ((36, 500, 1000), 2, "masked tiles in y and z"),
                                             ((40, 640, 640), 12, "more than 8 sources: scatter and repair passes"),
                                             ((48, 768, 768), 2, "tiles of 512 and 256 cells"),
                                             ((44, 640, 640), 3, "tiles of 512 and 128 cells"),
                                             ((112, 384, 384), 2, "tiles of 256 and 128 cells"),
                                             ((34, 2048, 2048), 2, "the cross-section of the 2048^3 configuration: four z-tiles of 512")])
def test_large_fused_shapes_against_oracle(pkg, hiplib, shape, nsrc, what):
    """The two-step sweep on cross-sections of the size the benchmarks use (16 M cells and more), compared with
    the CPU oracle directly rather than with another HIP kernel: random fields with different halos per level,
    varying m, anisotropic spacing, sources on a z-cut, in the outermost cell and (12 sources) overlapping."""
    nx, ny, nz = shape
    nt = 9
    rng = np.random.default_rng(2024)
    pshape = (nx + 8, ny + 8, nz + 8)
    u = np.zeros((3,) + pshape, dtype=np.float32)
    u[0] = rng.uniform(-1, 1, size=pshape).astype(np.float32)
    u[1] = u[0] + rng.uniform(-0.01, 0.01, size=pshape).astype(np.float32)
    u[2] = rng.uniform(-1, 1, size=pshape).astype(np.float32)
    m = rng.uniform(1, 3, size=pshape).astype(np.float32)
    h = (0.9, 1.0, 1.1)
    src = rng.uniform(-50, 50, size=(nt, nsrc)).astype(np.float32)
    ext = np.array([nx - 1, ny - 1, nz - 1]) * np.array(h)
    coords = (rng.uniform(0.05, 0.95, size=(nsrc, 3)) * ext).astype(np.float32)
    coords[0] = ((nx - 1) * h[0] + 0.3 * h[0], 0.0, (nz - 1) * h[2])        # outermost cell: halo-layer corners
    coords[1, 2] = np.float32(511.5 * h[2])                                    # corners at z = 511 | 512
    if nsrc > 8:
        coords[3] = coords[2] + np.array([1.2 * h[0], 0.0, 0.6 * h[2]], dtype=np.float32)   # overlapping stars
        coords[5] = coords[4]                                                                # coincident corners
    u_ref = u.copy()
    ob.run_oracle(m, src, coords, u_ref, h=h)
    before = hiplib.fdtd_hip_fused_launch_count()
    rc, _ = pkg.run_operator(hiplib.Kernel_CUDA_Optimized, m, src, coords, u, h=h)
    assert rc == 0
    assert hiplib.fdtd_hip_fused_launch_count() > before, what
    assert ob.bits_equal(u, u_ref), what
