"""One rank of a slab-decomposed run (spawned by the slab tests).

usage: python slab_worker.py <stepper: oracle|hip> <case> <outdir>     (RANK/WORLD_SIZE/MASTER_* in env)
Every rank builds the same global case, cuts its slab (ghosts included), runs the SlabPropagator
and saves its local wavefield; the parent reassembles and compares with a single-domain run.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import cases  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

HALO = 4


class OracleStepper:
    """CPU stepper for the orchestration tests: the oracle's sweeps on the local numpy arrays."""

    def __init__(self):
        import oracle_binding as ob
        self.ob = ob

    def bind(self, prop):
        self.p = prop
        self.u = prop.u.numpy()
        self.m = prop.m.numpy()
        self.damp = None if getattr(prop, "damp", None) is None else prop.damp.numpy()
        self.src = prop.src.numpy() if prop.nsrc else None
        self.coords = prop.coords.numpy() if prop.nsrc else None

    def begin_step(self):
        pass

    def stencil(self, x_lo, x_hi, t0, t1, t2, boundary):
        if x_hi < x_lo:
            return
        nxp, nyp, nzp = self.u.shape[1:]
        H = self.p.halo
        if H != HALO or self.damp is not None:      # the sweeps of the higher orders / with the damping term
            self.ob.oracle().oracle_stencil_step_order(
                H, self.m.ctypes.data, None if self.damp is None else self.damp.ctypes.data, self.u.ctypes.data,
                nxp, nyp, nzp, x_lo, x_hi, 0, nyp - 2 * H - 1, 0, nzp - 2 * H - 1, float(np.float32(self.p.dt)),
                *[float(np.float32(v)) for v in self.p.h], t0, t1, t2)
            return
        self.ob.oracle_stencil_step(self.m, self.u, (x_lo, x_hi, 0, nyp - 9, 0, nzp - 9),
                                    np.float32(self.p.dt), tuple(np.float32(v) for v in self.p.h), t0, t1, t2)

    def inject(self, region, t2, time, boundary):
        p = self.p
        nxp, nyp, nzp = self.u.shape[1:]
        H = p.halo
        if H != HALO:
            self.ob.oracle().oracle_source_inject_slab_order(
                H, self.m.ctypes.data, self.src.ctypes.data, self.coords.ctypes.data, self.u.ctypes.data,
                nxp, nyp, nzp, 0, nyp - 2 * H - 1, 0, nzp - 2 * H - 1, *[float(v) for v in p.h], *[float(v) for v in p.o],
                0, p.nsrc - 1, time, self.src.shape[1], self.coords.shape[1], t2, p.x_off, region[0], region[1])
            return
        self.ob.oracle().oracle_source_inject_slab(
            self.m.ctypes.data, self.src.ctypes.data, self.coords.ctypes.data, self.u.ctypes.data,
            nxp, nyp, nzp, 0, nyp - 9, 0, nzp - 9, *[float(v) for v in p.h], *[float(v) for v in p.o],
            0, p.nsrc - 1, time, self.src.shape[1], self.coords.shape[1], t2, p.x_off, region[0], region[1])

    def boundary_done(self):
        pass

    def interior_done(self):
        pass

    def finish(self):
        pass


class OraclePairStepper:
    """CPU stand-in for the two-steps-per-exchange calls: what fdtd_hip_two_steps_slab and
    fdtd_hip_one_step_slab are specified to do, with the oracle's sweeps on one [4][...] array."""

    def __init__(self, U):
        import oracle_binding as ob
        self.ob, self.U = ob, U.numpy()
        # FDTD_TEST_EDGES_FIRST=1: the schedule of large cross-sections (outer thirds, messages, middle third)
        self.zone_fused = os.environ.get("FDTD_TEST_EDGES_FIRST") == "1"

    def interior_after_boundary(self):
        pass

    def pairs_can_overlap(self, prop):
        return True          # sequential here, so the three ranges of the two-chain schedule are exercised

    def bind_pairs(self, prop):
        self.p = prop
        self.m = prop.m.numpy()
        self.src = prop.src.numpy() if prop.nsrc else None
        self.coords = prop.coords.numpy() if prop.nsrc else None

    def begin_step(self):
        pass

    def _stencil(self, lo, hi, t0, t1, t2):
        nyp, nzp = self.U.shape[2:]
        self.ob.oracle_stencil_step(self.m, self.U, (lo, hi, 0, nyp - 9, 0, nzp - 9), np.float32(self.p.dt),
                                    tuple(np.float32(v) for v in self.p.h), t0, t1, t2)

    def _inject(self, lvl, time, g_lo, g_hi, mode):
        p = self.p
        if not p.nsrc or g_hi < g_lo:
            return
        nxp, nyp, nzp = self.U.shape[1:]
        self.ob.oracle().oracle_source_inject_slab_mode(
            self.m.ctypes.data, self.src.ctypes.data, self.coords.ctypes.data, self.U.ctypes.data,
            nxp, nyp, nzp, 0, nyp - 9, 0, nzp - 9, *[float(v) for v in p.h], *[float(v) for v in p.o],
            0, p.nsrc - 1, time, self.src.shape[1], self.coords.shape[1], lvl, p.x_off, g_lo, g_hi, mode,
            0, p.layout.NX - 1)

    def _own(self, lo, hi):
        p = self.p
        g_lo, g_hi = lo + p.x_off, hi + p.x_off
        return (-1 if g_lo == 0 else g_lo), (p.layout.NX if g_hi == p.layout.NX - 1 else g_hi)

    def two_steps(self, lo, hi, cur, prev, n1, n2, time, boundary):
        if hi < lo:
            return
        p = self.p
        d_lo, d_hi = -p.x_off, p.layout.NX - 1 - p.x_off
        e_lo, e_hi = max(lo - 2, d_lo), min(hi + 2, d_hi)
        own = self._own(lo, hi)
        self._inject(n1, time, *own, 1)
        self._inject(n2, time + 1, *own, 1)
        self._stencil(e_lo, e_hi, cur, prev, n1)
        self._inject(n1, time, e_lo + p.x_off, e_hi + p.x_off, 2)
        self._stencil(lo, hi, n1, cur, n2)
        self._inject(n2, time + 1, lo + p.x_off, hi + p.x_off, 2)

    def one_step(self, lo, hi, cur, prev, nxt, time, boundary):
        if hi < lo:
            return
        self._stencil(lo, hi, cur, prev, nxt)
        self._inject(nxt, time, *self._own(lo, hi), 0)

    def copy_shell(self, dst, src):
        nx = self.p.nx
        a, b = self.U[dst], self.U[src]
        box = (slice(HALO, HALO + nx), slice(HALO, -HALO), slice(HALO, -HALO))
        grown = (slice(HALO - 2, HALO + nx + 2), slice(HALO - 2, -HALO + 2), slice(HALO - 2, -HALO + 2))
        keep = a[box].copy()
        a[grown] = b[grown]
        a[box] = keep

    def interior_done(self):
        pass

    def boundary_done(self):
        pass

    def finish(self):
        pass


def pair_run(kind, name, outdir):
    """One rank of a two-steps-per-exchange run (kind: oracle_pair over gloo on the CPU, hip_pair
    with the HIP kernels, ghosts staged through the host)."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # rendezvous through a file in the test's directory, which all ranks of a run share (no TCP port to collide on)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "rendezvous"), rank=rank,
                            world_size=world)
    pkg = load_package()
    c = cases.build(name, pkg)
    kw = c["kw"]
    h, dt = kw.get("h", (0.1, 0.1, 0.1)), kw.get("dt", 1e-3)
    NX = c["u"].shape[1] - 2 * HALO
    layout = pkg.slab.SlabLayout(NX, world)
    time_m = kw.get("time_m", 0)
    time_M = kw.get("time_M", None if c["src"] is None else c["src"].shape[0] - 1)
    u_loc = layout.cut(c["u"], rank, axis=1)
    U = torch.zeros((4,) + u_loc.shape[1:], dtype=torch.float32)
    U[:3] = torch.from_numpy(np.ascontiguousarray(u_loc))
    m_loc = torch.from_numpy(np.ascontiguousarray(layout.cut(c["m"], rank, axis=0)))
    dev = "cpu" if kind == "oracle_pair" else torch.device("cuda", 0)
    U, m_loc = U.to(dev), m_loc.to(dev)
    src_t = None if c["src"] is None else torch.from_numpy(c["src"]).to(dev)
    coords_t = None if c["coords"] is None else torch.from_numpy(c["coords"]).to(dev)
    # FDTD_TEST_SLAB_VARIANT: kernel variant of the HIP stepper (march2 = fused sweep whatever the shape)
    hip_variant = pkg.VARIANTS[os.environ.get("FDTD_TEST_SLAB_VARIANT", "auto")]
    stepper = OraclePairStepper(U) if kind == "oracle_pair" else pkg.slab.HipStepper(dev, variant=hip_variant)
    comm = pkg.slab.TorchComm(dist, rank, world, stepper)
    prop = pkg.slab.SlabPairPropagator(layout, rank, [U[i] for i in range(4)], m_loc, src_t, coords_t, dt=dt, h=h,
                                       o=(0.0, 0.0, 0.0), stepper=stepper, comm=comm, time0=time_m)
    prop.run(time_m, time_M)
    if kind == "hip_pair" and os.environ.get("FDTD_TEST_SLAB_VARIANT") == "march2" and time_M > time_m:
        # the forced variant must have taken the fused sweep wherever rows are whole float4s
        nz = U.shape[3] - 2 * HALO
        took = pkg.lib().fdtd_hip_fused_launch_count() > 0
        assert took == (nz % 4 == 0), (name, nz, took)
    # back into ring order; the shell travels with the level, padding further out stays with the slot
    out = torch.from_numpy(np.ascontiguousarray(u_loc)).clone()
    Uc = U.cpu()
    T = time_M
    nx = prop.nx
    grown = (slice(HALO - 2, HALO + nx + 2), slice(HALO - 2, -HALO + 2), slice(HALO - 2, -HALO + 2))
    for role, slot in zip(prop.ring(), ((T + 1) % 3, T % 3, (T + 2) % 3)):
        out[slot][grown] = Uc[role][grown]
    np.save(os.path.join(outdir, f"u_rank{rank}.npy"), out.numpy())
    dist.barrier()
    dist.destroy_process_group()


class SelfCopyComm:
    """What a rank that is its own left and right neighbour receives: its own boundary planes."""
    has_neighbours = True

    def exchange_planes(self, prop, items):
        for lvl, n in items:
            prop.recv_left(lvl, n).copy_(prop.send_left(lvl, n).clone())
            prop.recv_right(lvl, n).copy_(prop.send_right(lvl, n).clone())

    def exchange(self, prop, lvl):
        prop.recv_left(lvl).copy_(prop.send_left(lvl).clone())
        prop.recv_right(lvl).copy_(prop.send_right(lvl).clone())


def self_neighbour_run(name, outdir):
    """One process, one GPU, backend "nccl": the rank names itself as both neighbours, so every
    step runs the real RCCL messages on the second stream behind the boundary event.  Checked
    bit for bit against the same orchestration on the CPU (oracle sweeps, plain copies)."""
    pkg = load_package()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # one process: rendezvous through a file in the test's own directory (no TCP port to collide on)
    pkg.slab.init_rccl(dev, 0, 1, init_method="file://" + os.path.join(outdir, "rendezvous"))
    c = cases.build(name, pkg)
    kw = c["kw"]
    h, dt = kw.get("h", (0.1, 0.1, 0.1)), kw.get("dt", 1e-3)
    NX = c["u"].shape[1] - 2 * HALO
    time_m = kw.get("time_m", 0)
    time_M = kw.get("time_M", None if c["src"] is None else c["src"].shape[0] - 1)
    layout = pkg.slab.SlabLayout(NX, 1)
    results = []
    for kind in ("oracle", "hip"):
        d = "cpu" if kind == "oracle" else dev
        u_t, m_t = torch.from_numpy(c["u"].copy()).to(d), torch.from_numpy(c["m"]).to(d)
        src_t = None if c["src"] is None else torch.from_numpy(c["src"]).to(d)
        coords_t = None if c["coords"] is None else torch.from_numpy(c["coords"]).to(d)
        if kind == "oracle":
            stepper, comm = OracleStepper(), SelfCopyComm()
        else:
            stepper = pkg.slab.HipStepper(dev, variant=pkg.VARIANTS["auto"])
            comm = pkg.slab.TorchComm(dist, 0, 1, stepper, left=0, right=0)
            assert comm.backend == "nccl"
        prop = pkg.slab.SlabPropagator(layout, 0, u_t, m_t, src_t, coords_t, dt=dt, h=h, o=(0.0, 0.0, 0.0),
                                       stepper=stepper, comm=comm)
        assert prop.overlap
        prop.run(time_m, time_M)
        results.append(u_t.cpu().numpy())
    dist.destroy_process_group()
    a, b = (r.view(np.uint32) for r in results)
    nbad = int((a != b).sum())
    ghosts_used = not np.array_equal(results[0][:, 2:4], c["u"][:, 2:4])
    print(f"self-neighbour nccl run: {nbad} differing words, ghosts changed: {ghosts_used}")
    np.save(os.path.join(outdir, "u_rank0.npy"), results[1])
    sys.exit(0 if nbad == 0 and ghosts_used else 1)


def self_neighbour_pair_run(name, outdir):
    """The two-steps-per-exchange schedule with real RCCL messages on one GPU (the rank is its own
    two neighbours), against the same orchestration on the CPU."""
    pkg = load_package()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # one process: rendezvous through a file in the test's own directory (no TCP port to collide on)
    pkg.slab.init_rccl(dev, 0, 1, init_method="file://" + os.path.join(outdir, "rendezvous"))
    c = cases.build(name, pkg)
    kw = c["kw"]
    h, dt = kw.get("h", (0.1, 0.1, 0.1)), kw.get("dt", 1e-3)
    NX = c["u"].shape[1] - 2 * HALO
    time_m = kw.get("time_m", 0)
    time_M = kw.get("time_M", None if c["src"] is None else c["src"].shape[0] - 1)
    layout = pkg.slab.SlabLayout(NX, 1)
    results = []
    for kind in ("oracle", "hip"):
        d = "cpu" if kind == "oracle" else dev
        U = torch.zeros((4,) + c["u"].shape[1:], dtype=torch.float32)
        U[:3] = torch.from_numpy(c["u"])
        U, m_t = U.to(d), torch.from_numpy(c["m"]).to(d)
        src_t = None if c["src"] is None else torch.from_numpy(c["src"]).to(d)
        coords_t = None if c["coords"] is None else torch.from_numpy(c["coords"]).to(d)
        if kind == "oracle":
            stepper, comm = OraclePairStepper(U), SelfCopyComm()
        else:
            stepper = pkg.slab.HipStepper(dev, variant=pkg.VARIANTS["auto"])
            comm = pkg.slab.TorchComm(dist, 0, 1, stepper, left=0, right=0)
        prop = pkg.slab.SlabPairPropagator(layout, 0, [U[i] for i in range(4)], m_t, src_t, coords_t, dt=dt, h=h,
                                           o=(0.0, 0.0, 0.0), stepper=stepper, comm=comm, time0=time_m)
        assert prop.overlap
        prop.run(time_m, time_M)
        results.append(torch.stack([U[i] for i in prop.ring()]).cpu().numpy())
    dist.destroy_process_group()
    a, b = (r.view(np.uint32) for r in results)
    nbad = int((a != b).sum())
    print(f"self-neighbour nccl pair run: {nbad} differing words")
    if nbad:   # where: level and index ranges of the differing words
        idx = np.argwhere(a != b)
        for lvl in np.unique(idx[:, 0]):
            sel = idx[idx[:, 0] == lvl]
            print(f"  level {lvl}: {len(sel)} words, x {sel[:, 1].min()}..{sel[:, 1].max()} "
                  f"y {sel[:, 2].min()}..{sel[:, 2].max()} z {sel[:, 3].min()}..{sel[:, 3].max()} (padded indices)")
    sys.exit(0 if nbad == 0 else 1)


def order_case(name):
    """Cases of the one-step schedule beyond order 4 / with damping: "o<order>" or "o<order>_damped" — a random
    field on arrays padded by the order, varying m, three sources (one in the outermost cell), a sponge."""
    from test_forward_operator import sponge
    from test_space_orders import padded_case
    order = int(name[1:].split("_")[0])
    n, nt = (26, 18, 40), 9
    u, m, src, coords, kw = padded_case(order, n, nt, 3, 300 + order, None, True)
    coords[1, 0] = 12.5 * kw["h"][0]      # corners in planes 12 | 13: the cut between two slabs of 13 planes
    damp = sponge(m.shape, 5, 40.0, halo=order) if name.endswith("_damped") else None
    return order, u, m, damp, src, coords, kw


def order_run(kind, name, outdir):
    """One rank of a one-step-per-exchange run at a space order above 4 and / or with the damping term."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "rendezvous"), rank=rank,
                            world_size=world)
    pkg = load_package()
    order, u, m, damp, src, coords, kw = order_case(name)
    NX = u.shape[1] - 2 * order
    layout = pkg.slab.SlabLayout(NX, world, order=order)
    dev = "cpu" if kind == "oracle_order" else torch.device("cuda", 0)
    cut = lambda a, axis: torch.from_numpy(np.ascontiguousarray(layout.cut(a, rank, axis=axis))).to(dev)
    u_t, m_t = cut(u, 1), cut(m, 0)
    damp_t = None if damp is None else cut(damp, 0)
    stepper = OracleStepper() if kind == "oracle_order" else pkg.slab.HipStepper(dev)
    comm = pkg.slab.TorchComm(dist, rank, world, stepper)
    prop = pkg.slab.SlabPropagator(layout, rank, u_t, m_t, torch.from_numpy(src).to(dev), torch.from_numpy(coords).to(dev),
                                   dt=kw["dt"], h=kw["h"], o=(0.0, 0.0, 0.0), stepper=stepper, comm=comm, damp=damp_t)
    assert prop.ghost == order // 2 and prop.overlap == (prop.nx >= order + 1)
    prop.run(0, kw["time_M"])
    np.save(os.path.join(outdir, f"u_rank{rank}.npy"), u_t.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def main():
    kind, name, outdir = sys.argv[1:4]
    if kind in ("oracle_order", "hip_order"):
        return order_run(kind, name, outdir)
    if kind == "nccl_self_pair":
        return self_neighbour_pair_run(name, outdir)
    if kind == "nccl_self":
        return self_neighbour_run(name, outdir)
    if kind in ("oracle_pair", "hip_pair"):
        return pair_run(kind, name, outdir)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # rendezvous through a file in the test's directory, which all ranks of a run share (no TCP port to collide on)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "rendezvous"), rank=rank,
                            world_size=world)
    pkg = load_package()
    c = cases.build(name, pkg)
    kw = c["kw"]
    h = kw.get("h", (0.1, 0.1, 0.1))
    dt = kw.get("dt", 1e-3)
    NX = c["u"].shape[1] - 2 * HALO
    layout = pkg.slab.SlabLayout(NX, world)
    u_loc = np.ascontiguousarray(layout.cut(c["u"], rank, axis=1))
    m_loc = np.ascontiguousarray(layout.cut(c["m"], rank, axis=0))
    time_m = kw.get("time_m", 0)
    time_M = kw.get("time_M", None if c["src"] is None else c["src"].shape[0] - 1)
    if kind == "oracle":
        dev = "cpu"
        stepper = OracleStepper()
    else:
        dev = torch.device("cuda", 0)
        stepper = pkg.slab.HipStepper(dev, variant=pkg.VARIANTS["auto"])
    u_t = torch.from_numpy(u_loc).to(dev)
    m_t = torch.from_numpy(m_loc).to(dev)
    src_t = None if c["src"] is None else torch.from_numpy(c["src"]).to(dev)
    coords_t = None if c["coords"] is None else torch.from_numpy(c["coords"]).to(dev)
    comm = pkg.slab.TorchComm(dist, rank, world, stepper)
    prop = pkg.slab.SlabPropagator(layout, rank, u_t, m_t, src_t, coords_t, dt=dt, h=h, o=(0.0, 0.0, 0.0),
                                   stepper=stepper, comm=comm)
    prop.run(time_m, time_M)
    np.save(os.path.join(outdir, f"u_rank{rank}.npy"), u_t.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
