"""Parity cases shared by the fixture generator and the tests.

Each case builds its inputs deterministically (no reference code involved): the driver's own
generators for the two published input sets (main.cpp:285-356 and 525-532), seeded numpy
for the edge cases the reference never exercises.  `build(case, pkg)` returns a dict with the
arrays and the keyword arguments of pkg.run_operator.
"""
import zlib

import numpy as np

HALO = 4


def _rng_field(rng, shape, lo, hi):
    return rng.uniform(lo, hi, size=shape).astype(np.float32)


CASES = {
    # name: (kind, params)
    "ricker_n32": ("ricker", dict(n=(32, 32, 32), nt=50, nsrc=1)),
    "ricker_n64": ("ricker", dict(n=(64, 64, 64), nt=50, nsrc=1)),
    "correctness_n32": ("correctness", dict(n=32, nt=50)),
    "correctness_n64": ("correctness", dict(n=64, nt=50)),
    "ricker27_n32": ("ricker", dict(n=(32, 32, 32), nt=20, nsrc=27)),
    # 30 sources: 27 ticks + 3 copies of the grid centre, i.e. coincident scatter targets
    "ricker30_n24_overlap": ("ricker", dict(n=(24, 24, 24), nt=12, nsrc=30)),
    "noncubic_aniso": ("random", dict(n=(20, 36, 44), nt=12, nsrc=2, h=(0.8, 1.0, 1.25), seed=11,
                                      edge_source=True)),
    "odd_sizes": ("random", dict(n=(17, 33, 65), nt=9, nsrc=3, h=(1.0, 1.0, 1.0), seed=12)),
    "tile_edges_100": ("random", dict(n=(12, 100, 260), nt=7, nsrc=1, h=(1.0, 0.9, 1.1), seed=13)),
    # every tile of the marching kernel is full here (z extent 256, y extent 32): its branch-free path
    "full_tiles_256": ("random", dict(n=(11, 32, 256), nt=7, nsrc=1, h=(1.0, 1.1, 0.9), seed=18)),
    "subrange_aligned": ("random", dict(n=(32, 32, 32), nt=10, nsrc=2, h=(1.0, 1.0, 1.0), seed=14,
                                        bounds=(3, 20, 2, 29, 4, 27))),
    "subrange_unaligned": ("random", dict(n=(32, 32, 32), nt=10, nsrc=2, h=(1.0, 1.0, 1.0), seed=15,
                                          bounds=(1, 30, 5, 26, 5, 26))),
    "time_offset": ("random", dict(n=(24, 24, 24), nt=19, nsrc=2, h=(1.0, 1.0, 1.0), seed=16,
                                   time_m=7)),
    # only sources 1..2 of 4 are active (p_src_m > 0, p_src_M < nsrc-1)
    "source_subset": ("random", dict(n=(16, 20, 24), nt=8, nsrc=4, h=(1.0, 1.0, 1.0), seed=19, p_src=(1, 2))),
    # empty update range in x (x_M < x_m): no cell is updated, the scatter test excludes every corner too
    "empty_range": ("random", dict(n=(12, 12, 16), nt=4, nsrc=1, h=(1.0, 1.0, 1.0), seed=20,
                                   bounds=(5, 4, 0, 11, 0, 15))),
    # time_M < time_m: nothing happens at all
    "zero_steps": ("random", dict(n=(12, 12, 16), nt=4, nsrc=1, h=(1.0, 1.0, 1.0), seed=21, time_m=3,
                                  time_M_override=2)),
    # z extent 512, y extent a multiple of 8, at least 32 planes: the two-steps-per-sweep kernel.
    # Random fields carry a different halo in each ring level, so the shells have to follow the ring;
    # the second case adds a source in the outermost cell (scatter into the halo layer) and a time offset;
    # the third has zero halos throughout (no shell copies) and the largest number of sources the fused sweep takes.
    "fused_z512": ("random", dict(n=(40, 16, 512), nt=11, nsrc=3, h=(1.0, 0.9, 1.1), seed=22)),
    "fused_z512_edge_source": ("random", dict(n=(36, 24, 512), nt=17, nsrc=2, h=(0.8, 1.0, 1.25), seed=23,
                                              edge_source=True, time_m=7)),
    "fused_z512_ricker8": ("ricker", dict(n=(34, 16, 512), nt=14, nsrc=8)),
    # the fused sweep on a box strictly inside the arrays: x 2..37, y 8..23, z 4..515 of 40 x 32 x 520
    "fused_z512_subrange": ("random", dict(n=(40, 32, 520), nt=9, nsrc=2, h=(1.0, 1.0, 1.0), seed=24,
                                           bounds=(2, 37, 8, 23, 4, 515))),
    # rows of 256 cells: the fused sweep carries two 8-row tiles per workgroup (y extent a multiple of 16)
    "fused_z256": ("random", dict(n=(36, 32, 256), nt=10, nsrc=3, h=(1.1, 1.0, 0.9), seed=25, edge_source=True)),
    # rows of 1024 cells: two z-tiles of 512 with a cut between cells 511 and 512; sources straddle the cut
    # (corners at 511 | 512) and sit right beside it (corners at 510, 511)
    "fused_z1024": ("random", dict(n=(34, 16, 1024), nt=10, nsrc=3, h=(1.0, 1.1, 0.9), seed=26, edge_source=True,
                                   seam_sources=True)),
    # a source above the last cell in z, half way along x: its upper corners land in the halo layer of planes
    # in the middle of the grid (of the interior range of a slab, where the shell copy and the scatter of those
    # corners are issued on different streams)
    "fused_z512_zedge_source": ("random", dict(n=(36, 24, 512), nt=13, nsrc=2, h=(1.0, 1.0, 1.0), seed=27,
                                               zedge_mid_source=True)),
    "warmup_only": ("ricker", dict(n=(16, 16, 16), nt=3, nsrc=1)),
    "no_sources_random": ("random", dict(n=(28, 20, 24), nt=8, nsrc=0, h=(1.0, 1.0, 1.0), seed=17)),
}

# cases whose scatter targets coincide: float atomics may land in any order on the GPU
ORDER_DEPENDENT = {"ricker30_n24_overlap"}


def build(name, pkg):
    kind, p = CASES[name]
    if kind == "ricker":
        nx, ny, nz = p["n"]
        u, m, src, coords = pkg.standard_inputs(nx, ny, nz, p["nt"], p["nsrc"])
        return dict(u=u, m=m, src=src, coords=coords, kw=dict())
    if kind == "correctness":
        u, m = pkg.fill_correctness_inputs(p["n"])
        return dict(u=u, m=m, src=None, coords=None,
                    kw=dict(time_M=p["nt"] - 1, h=(1.0, 1.0, 1.0)))
    # random: smooth-ish random wavefield on levels 0 and 1 (halos included), varying m
    rng = np.random.default_rng(p["seed"])
    nx, ny, nz = p["n"]
    shape = (nx + 2 * HALO, ny + 2 * HALO, nz + 2 * HALO)
    u = np.zeros((3,) + shape, dtype=np.float32)
    u[0] = _rng_field(rng, shape, -1.0, 1.0)
    u[1] = u[0] + _rng_field(rng, shape, -0.01, 0.01)
    u[2] = _rng_field(rng, shape, -1.0, 1.0)  # stale data in the level that gets overwritten
    m = _rng_field(rng, shape, 1.0, 3.0)
    h = p["h"]
    nt, nsrc = p["nt"], p["nsrc"]
    kw = dict(h=h, dt=1e-3)
    if "bounds" in p:
        kw["bounds"] = p["bounds"]
    if "time_m" in p:
        kw["time_m"] = p["time_m"]
    if "p_src" in p:
        kw["p_src"] = p["p_src"]
    src = coords = None
    if nsrc > 0:
        src = _rng_field(rng, (nt, nsrc), -50.0, 50.0)
        ext = np.array([nx - 1, ny - 1, nz - 1], dtype=np.float64) * np.array(h)
        coords = (rng.uniform(0.05, 0.95, size=(nsrc, 3)) * ext).astype(np.float32)
        if p.get("edge_source"):
            # a source in the outermost cell: its scatter reaches the first halo layer
            coords[0] = np.array([(nx - 1) * h[0] + 0.3 * h[0], 0.0, (nz - 1) * h[2]],
                                 dtype=np.float32)
        if p.get("zedge_mid_source"):
            coords[1] = np.array([0.5 * (nx - 1) * h[0] + 0.2 * h[0], 0.4 * (ny - 1) * h[1], (nz - 1) * h[2] + 0.3 * h[2]],
                                 dtype=np.float32)
        if p.get("seam_sources"):
            coords[1, 2] = np.float32(511.5 * h[2])
            coords[2, 2] = np.float32(510.25 * h[2])
        kw["time_M"] = nt - 1
    else:
        kw["time_M"] = nt - 1
    if "time_M_override" in p:
        kw["time_M"] = p["time_M_override"]
    return dict(u=u, m=m, src=src, coords=coords, kw=kw)


def digest(u):
    """Bit-level digest of a wavefield: CRC32 of each level + fp64 moments."""
    u = np.ascontiguousarray(u)
    d = {"crc32": [format(zlib.crc32(u[t].tobytes()), "08x") for t in range(3)]}
    f = u.astype(np.float64)
    d["sum"] = float(f.sum())
    d["sumsq"] = float((f * f).sum())
    d["maxabs"] = float(np.abs(f).max())
    return d


def sample_indices(size, count=4096, seed=1234):
    rng = np.random.default_rng(seed)
    return np.sort(rng.choice(size, size=min(count, size), replace=False))
