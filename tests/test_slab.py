"""Slab decomposition (multi-GPU path): world_size 2 and 3 over gloo.

CPU tests drive the real orchestration (region split, source ownership, ghost exchange, ring
levels) with a stepper built on the oracle and require the reassembled field to be bit-identical
to the single-domain golden result.  The GPU test runs the HIP kernels on two ranks that share
one GPU (ghosts staged through the host, since RCCL needs one device per rank)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cases
from test_oracle import check_against_golden

HERE = os.path.dirname(os.path.abspath(__file__))
HALO = 4


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_slabs(kind, name, world, tmp_path, pkg):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "slab_worker.py"), kind, name,
                                       str(tmp_path)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    c = cases.build(name, pkg)
    NX = c["u"].shape[1] - 2 * HALO
    layout = pkg.slab.SlabLayout(NX, world)
    out = np.full_like(c["u"], np.nan)
    for rank in range(world):
        loc = np.load(os.path.join(tmp_path, f"u_rank{rank}.npy"))
        layout.paste_owned(out, loc, rank, axis=1)
    return out


def run_order_slabs(kind, name, world, tmp_path, pkg):
    """A slab run of the one-step schedule at a space order above 4 and / or with the damping term, reassembled,
    and the same case through the single-domain restatement."""
    import oracle_binding as ob
    from slab_worker import order_case
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(free_port()), OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "slab_worker.py"), kind, name,
                                       str(tmp_path)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    order, u, m, damp, src, coords, kw = order_case(name)
    layout = pkg.slab.SlabLayout(u.shape[1] - 2 * order, world, order=order)
    out = np.full_like(u, np.nan)
    for rank in range(world):
        layout.paste_owned(out, np.load(os.path.join(tmp_path, f"u_rank{rank}.npy")), rank, axis=1)
    ref = u.copy()
    if damp is None:
        ob.run_oracle_order(order, m, src, coords, ref, **kw)
    else:
        rc, _ = pkg.run_forward(ob.forward_oracle_order(order), damp, m, None, None, src, coords, ref, h=kw["h"],
                                dt=kw["dt"], bounds=kw["bounds"], time_M=kw["time_M"])
        assert rc == 0
    return out, ref


# One step per exchange at the higher space orders (order / 2 ghost planes per side on arrays padded by the order) and
# with the damping term of the forward operator: no reference arithmetic exists for either (parity unpinned); the
# union of the slabs must be the single-domain restatement's bits.
@pytest.mark.parametrize("name,world", [("o6", 2), ("o8", 2), ("o12", 2), ("o8", 3), ("o4_damped", 2), ("o8_damped", 2),
                                        ("o12_damped", 3)])
def test_slab_orders_and_damping_orchestration_cpu(pkg, tmp_path, name, world):
    import oracle_binding as ob
    out, ref = run_order_slabs("oracle_order", name, world, tmp_path, pkg)
    assert ob.bits_equal(out, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("o6", 2), ("o8", 3), ("o12", 2), ("o4_damped", 2), ("o8_damped", 2)])
def test_slab_orders_and_damping_hip_ranks_one_gpu(pkg, tmp_path, name, world):
    """The same with the HIP kernels, several ranks sharing one GPU (ghosts staged through the host)."""
    import oracle_binding as ob
    out, ref = run_order_slabs("hip_order", name, world, tmp_path, pkg)
    assert ob.bits_equal(out, ref)


def test_layout_covers_every_plane_once(pkg):
    for NX, world in ((32, 2), (20, 3), (17, 4), (512, 8), (2048, 8)):
        lay = pkg.slab.SlabLayout(NX, world)
        assert sum(lay.counts) == NX and lay.offsets[0] == 0
        owned = []
        for r in range(world):
            lo, hi = lay.owned_corner_range(r)
            owned += list(range(lo, hi + 1))
        assert owned == list(range(-1, NX + 1))      # scatter targets incl. the two halo planes, once each
    with pytest.raises(ValueError):
        pkg.slab.SlabLayout(5, 4)


# sources sit at 1/4, 1/2, 3/4 of the x extent in the ricker cases, i.e. next to / on slab cuts
@pytest.mark.parametrize("name,world", [("ricker27_n32", 2), ("ricker27_n32", 3), ("noncubic_aniso", 2),
                                        ("time_offset", 3), ("ricker30_n24_overlap", 2),
                                        ("no_sources_random", 2)])
def test_slab_orchestration_cpu(pkg, tmp_path, name, world):
    u = run_slabs("oracle", name, world, tmp_path, pkg)
    check_against_golden(name, u, exact=True)


# Two steps per exchange: 4-plane ghosts, u(t+1) recomputed on two ghost planes, shells following the ring.
# noncubic_aniso has a source in the outermost cell (halo-layer corners), time_offset an odd start and
# an odd number of steps (pairs, then one single step), ricker27 sources on and next to the cuts.
@pytest.mark.parametrize("name,world", [("ricker27_n32", 2), ("ricker27_n32", 3), ("noncubic_aniso", 2),
                                        ("time_offset", 2), ("odd_sizes", 2), ("no_sources_random", 3)])
def test_slab_pairs_orchestration_cpu(pkg, tmp_path, name, world):
    u = run_slabs("oracle_pair", name, world, tmp_path, pkg)
    check_against_golden(name, u, exact=True)


@pytest.mark.parametrize("name,world", [("ricker27_n32", 2), ("noncubic_aniso", 2), ("time_offset", 2),
                                        ("no_sources_random", 2)])
def test_slab_pairs_edges_first_orchestration_cpu(pkg, tmp_path, name, world, monkeypatch):
    """The pair schedule of large cross-sections (outer thirds of the slab first, then the messages, then
    the middle third) over gloo with the oracle's sweeps: same bytes as the single-domain fixtures."""
    monkeypatch.setenv("FDTD_TEST_EDGES_FIRST", "1")
    u = run_slabs("oracle_pair", name, world, tmp_path, pkg)
    check_against_golden(name, u, exact=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("fused_z512", 2), ("fused_z512_edge_source", 2), ("fused_z512_ricker8", 3),
                                        ("ricker27_n32", 2), ("fused_z1024", 2), ("fused_z512_zedge_source", 2)])
def test_slab_pairs_hip_ranks_one_gpu(pkg, tmp_path, name, world):
    """The fused sweep inside the slab schedule (512-wide cases: boundary and interior calls on two
    streams) and the two-sweep fallback (32^3), ranks sharing one GPU."""
    u = run_slabs("hip_pair", name, world, tmp_path, pkg)
    check_against_golden(name, u, exact=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("ricker27_n32", 2), ("ricker30_n24_overlap", 2), ("noncubic_aniso", 2),
                                        ("odd_sizes", 2)])
def test_slab_pairs_forced_fused_hip_ranks_one_gpu(pkg, tmp_path, name, world, monkeypatch):
    """The fused sweep forced on slabs of small and odd shapes: 27 and 30 sources (one call per rank over
    its whole range, scatter and repair passes behind the sweep, ghost planes of u(t+1) kept), masked tiles
    with the interior / boundary split, and a row length that is not a multiple of 4 (two-sweep fallback)."""
    monkeypatch.setenv("FDTD_TEST_SLAB_VARIANT", "march2")
    u = run_slabs("hip_pair", name, world, tmp_path, pkg)
    check_against_golden(name, u, exact=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("ricker27_n32", 2), ("noncubic_aniso", 2), ("time_offset", 3),
                                        ("ricker30_n24_overlap", 2)])
def test_slab_hip_two_ranks_one_gpu(pkg, tmp_path, name, world):
    u = run_slabs("hip", name, world, tmp_path, pkg)
    check_against_golden(name, u, exact=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name,march", [("ricker27_n32", False), ("ricker27_n32", True), ("time_offset", True)])
def test_slab_rccl_schedule_one_gpu(pkg, tmp_path, name, march):
    """The RCCL path itself (batched isend/irecv on the second stream, boundary and ghost events)
    with one rank acting as its own two neighbours; bit-identical to the CPU orchestration."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if march:   # the marching kernel also on these small grids
        env.update(FDTD_MARCH_MIN_CELLS="0", FDTD_MARCH_MIN_PLANES="1")
    p = subprocess.run([sys.executable, os.path.join(HERE, "slab_worker.py"), "nccl_self", name, str(tmp_path)],
                       env=env, timeout=300)
    assert p.returncode == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,zone_fused", [("fused_z512", "0"), ("fused_z512_edge_source", "0"),
                                             ("fused_z512_ricker8", "0"), ("fused_z1024", "0"),
                                             ("fused_z512_edge_source", "1"), ("fused_z1024", "1"),
                                             ("fused_z512_zedge_source", "0"), ("fused_z512_zedge_source", "1")])
def test_slab_pairs_rccl_schedule_one_gpu(pkg, tmp_path, name, zone_fused):
    """Two steps per exchange with real RCCL messages (4 planes per side and pair) next to the fused
    interior sweep, one rank as its own two neighbours; bit-identical to the CPU orchestration.  The
    boundary zones go through the two small sweeps in scratch, or (as large cross-sections do by
    default) through the fused sweep themselves."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="8",
               FDTD_SLAB_ZONE_FUSED=zone_fused)
    p = subprocess.run([sys.executable, os.path.join(HERE, "slab_worker.py"), "nccl_self_pair", name, str(tmp_path)],
                       env=env, timeout=300)
    assert p.returncode == 0
