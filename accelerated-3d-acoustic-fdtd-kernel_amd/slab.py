"""Multi-GPU propagation: the grid is cut into slabs along x, the slowest storage axis, one slab
per process/GPU (torch.distributed; backend "nccl" is RCCL over xGMI).

The reference is single-device (no counterpart); the arithmetic per cell is unchanged
(openacc.cpp:159-164, 185-196), so the union of the slabs is bit-identical to a single-domain
run.  What is new is the data movement:

* storage: rank r owns global planes [x_off, x_off + nx_loc) and keeps them in a local array
  with the usual 4-plane padding on both sides.  The inner two padding planes are the ghosts
  (copies of the neighbour's outermost owned planes); on the first/last rank they are the
  physical halo.  Every ghost block is one contiguous span of 2*nyp*nzp floats, so no pack or
  unpack kernels exist.  (Cutting along the contiguous z axis instead would turn each ghost
  "plane" into a stride-nzp gather of 2-float pieces.)
* per step, boundary chain: (1) update the two outermost owned planes on each side, (2) scatter
  the sources that land there, (3) hand exactly those planes of the new level to the
  neighbours — two point-to-point messages per neighbour, no collective.  Interior chain:
  (4) update the interior planes, (5) scatter the remaining sources.  The two chains are
  independent within a step (disjoint planes of the new level) and run on two streams; across
  steps the boundary chain needs the previous interior update and the previous messages, the
  interior chain needs the previous boundary update, never the messages.
* a source corner is applied by the rank that owns its global x index (first/last rank also
  own the physical halo plane next to them, openacc.cpp:192), so each corner is added once.

`stepper` supplies the kernels (HipStepper: libfdtd_hip.so on the GPU; the tests drive the same
orchestration with a CPU stepper built on the oracle).
"""
import ctypes as C
import os

import numpy as np

from . import abi

HALO = abi.HALO
GHOST = 2  # radius of the stencil = planes exchanged per side


class SlabLayout:
    """Index arithmetic of the x-slab decomposition of NX interior planes over `world` ranks."""

    def __init__(self, NX, world, order=2 * GHOST):
        """`order`: space order of the stencil; the arrays are padded by `order` cells per side (main.cpp:32) and
        order / 2 planes are exchanged per side and step."""
        if world < 1 or NX < world:
            raise ValueError("need at least one plane per rank")
        base, rem = divmod(NX, world)
        self.NX, self.world = NX, world
        self.halo, self.ghost = order, order // 2
        self.counts = [base + (1 if r < rem else 0) for r in range(world)]
        self.offsets = [sum(self.counts[:r]) for r in range(world)]
        if world > 1 and min(self.counts) < self.ghost:
            raise ValueError("slabs thinner than the stencil radius are not supported")

    def nx_loc(self, rank):
        return self.counts[rank]

    def x_off(self, rank):
        return self.offsets[rank]

    def owned_corner_range(self, rank):
        """Global x indices of scatter targets this rank applies (inclusive)."""
        lo = self.offsets[rank] if rank > 0 else -1
        hi = self.offsets[rank] + self.counts[rank] - 1 if rank < self.world - 1 else self.NX
        return lo, hi

    def cut(self, global_padded, rank, axis):
        """Local padded view [x_off, x_off + nx_loc + 8) of a globally padded array along `axis`."""
        sl = [slice(None)] * global_padded.ndim
        sl[axis] = slice(self.offsets[rank], self.offsets[rank] + self.counts[rank] + 2 * self.halo)
        return global_padded[tuple(sl)]

    def paste_owned(self, global_padded, local_padded, rank, axis):
        """Write a rank's owned planes (plus the physical halo on the first/last rank) back."""
        lo, hi = self.halo, self.halo + self.counts[rank]
        if rank == 0:
            lo = 0
        if rank == self.world - 1:
            hi = self.counts[rank] + 2 * self.halo
        src = [slice(None)] * local_padded.ndim
        dst = [slice(None)] * global_padded.ndim
        src[axis] = slice(lo, hi)
        dst[axis] = slice(self.offsets[rank] + lo, self.offsets[rank] + hi)
        global_padded[tuple(dst)] = local_padded[tuple(src)]


def source_floor_planes(coords, o_x, h_x):
    """floor((-o_x + c_x)/h_x) per source, in binary32 like openacc.cpp:185."""
    c = np.asarray(coords, dtype=np.float32)[:, 0]
    g = (np.float32(-o_x) + c) / np.float32(h_x)
    return np.floor(g).astype(np.int64)


class SlabPropagator:
    """Time stepping of one slab, one step per exchange.  `u` is the local [3][nx_loc+2H][nyp][nzp] tensor, `m`
    the local [nx_loc+2H][nyp][nzp] tensor (both torch tensors on the stepper's device, ghosts filled), H = the
    layout's padding = the space order (4 by default; 6 .. 12: the sweeps of the higher space orders, order / 2
    ghost planes per side).  `damp` (shape of m, or None): the damped update of the forward operator; it has no
    reference arithmetic, like the higher orders (parity unpinned)."""

    def __init__(self, layout, rank, u, m, src, coords, *, dt, h, o, stepper, comm, damp=None):
        self.layout, self.rank = layout, rank
        self.u, self.m, self.damp = u, m, damp
        self.halo, self.ghost, self.order = layout.halo, layout.ghost, layout.halo
        self.nx = layout.nx_loc(rank)
        self.x_off = layout.x_off(rank)
        self.dt, self.h, self.o = dt, h, o
        self.stepper, self.comm = stepper, comm
        self.nsrc = 0 if src is None else int(src.shape[1])
        self.src, self.coords = src, coords
        lo, hi = layout.owned_corner_range(rank)
        # regions of owned scatter targets: left boundary, right boundary, interior (global x)
        nx, xo = self.nx, self.x_off
        exchanging = layout.world > 1 or getattr(comm, "has_neighbours", False)
        G = self.ghost
        self.overlap = exchanging and nx >= 2 * G + 1
        if self.overlap:
            self.regions = {
                "left": (lo, xo + G - 1),
                "right": (xo + nx - G, hi),
                "interior": (xo + G, xo + nx - G - 1),
            }
        else:
            self.regions = {"all": (lo, hi)}
        # which regions can receive a corner at all (decided once, on the host)
        self.active = {k: False for k in self.regions}
        if self.nsrc > 0:
            coords_host = coords.detach().cpu().numpy() if hasattr(coords, "detach") else np.asarray(coords)
            pos = source_floor_planes(coords_host, o[0], h[0])
            for k, (a, b) in self.regions.items():
                self.active[k] = bool(((pos + 1 >= a) & (pos <= b)).any())
        stepper.bind(self)

    # ---- one time step (ring indices as openacc.cpp:148) ----
    def step(self, time):
        t0, t1, t2 = time % 3, (time + 2) % 3, (time + 1) % 3
        st, nx, G = self.stepper, self.nx, self.ghost
        st.begin_step()
        if self.overlap:
            # the long chain is issued first so that it never queues behind the short one
            st.stencil(G, nx - G - 1, t0, t1, t2, boundary=False)
            if self.active["interior"]:
                st.inject(self.regions["interior"], t2, time, boundary=False)
            st.interior_done()
            st.stencil(0, G - 1, t0, t1, t2, boundary=True)
            st.stencil(nx - G, nx - 1, t0, t1, t2, boundary=True)
            for k in ("left", "right"):
                if self.active[k]:
                    st.inject(self.regions[k], t2, time, boundary=True)
            st.boundary_done()
            self.comm.exchange(self, t2)
        else:
            st.stencil(0, nx - 1, t0, t1, t2, boundary=True)
            if self.active["all"]:
                st.inject(self.regions["all"], t2, time, boundary=True)
            st.boundary_done()
            self.comm.exchange(self, t2)

    def run(self, time_m, time_M):
        for time in range(time_m, time_M + 1):
            self.step(time)
        self.stepper.finish()

    def reset(self):
        """Zero wavefield, ready to start again at time 0 (a benchmark repetition, main.cpp:389)."""
        self.stepper.finish()
        self.u.zero_()

    # planes of level `lvl` sent to / received from the neighbours (contiguous tensor views)
    def send_left(self, lvl):
        return self.u[lvl, self.halo:self.halo + self.ghost]

    def send_right(self, lvl):
        return self.u[lvl, self.halo + self.nx - self.ghost:self.halo + self.nx]

    def recv_left(self, lvl):
        return self.u[lvl, self.halo - self.ghost:self.halo]

    def recv_right(self, lvl):
        return self.u[lvl, self.halo + self.nx:self.halo + self.nx + self.ghost]

    # ---- benchmark helpers ----
    def interior_cells(self):
        nyp, nzp = self.u.shape[2], self.u.shape[3]
        planes = self.nx - 2 * self.ghost if self.overlap else self.nx
        return float(planes) * (nyp - 2 * self.halo) * (nzp - 2 * self.halo)

    def mean_interior_kernel_ms(self):
        return self.stepper.mean_interior_kernel_ms()

    @classmethod
    def standard_benchmark(cls, n_per_rank, ny, nz, nt, rank, world, device, variant=-1, global_x=0,
                           self_exchange=False):
        """Weak-scaling benchmark state: global (n_per_rank*world) x ny x nz grid, u = 0, m = 1.5,
        h = 0.1, dt = 1e-3, one Ricker source at a quarter of the global extents
        (main.cpp:285-356 applied to the global grid), built directly on the device.
        `self_exchange` (world 1 only): the rank is its own left and right neighbour, so one GPU
        runs the whole multi-GPU schedule including the RCCL messages."""
        import torch
        import torch.distributed as dist
        NX = global_x if global_x else n_per_rank * world   # strong scaling: fixed global extent
        layout = SlabLayout(NX, world)
        nx = layout.nx_loc(rank)
        shape = (nx + 2 * HALO, ny + 2 * HALO, nz + 2 * HALO)
        u = torch.zeros((3,) + shape, dtype=torch.float32, device=device)
        m = torch.full(shape, 1.5, dtype=torch.float32, device=device)
        src = torch.from_numpy(abi.fill_ricker(nt, 1)).to(device)
        coords = torch.from_numpy(abi.fill_source_coords(1, NX, ny, nz)).to(device)
        stepper = HipStepper(device, variant=variant, time_interior=True)
        if self_exchange:
            comm = TorchComm(dist, rank, world, stepper, left=rank, right=rank)
        else:
            comm = TorchComm(dist, rank, world, stepper)
        return cls(layout, rank, u, m, src, coords, dt=1e-3, h=(0.1, 0.1, 0.1), o=(0.0, 0.0, 0.0),
                   stepper=stepper, comm=comm)


PAIR_GHOST = 4  # planes handed over per side when two steps are taken per exchange


class SlabPairPropagator:
    """Time stepping of one slab, two steps per exchange (fdtd_hip_two_steps_slab).

    `levels` are four tensors [nx_loc+8][nyp][nzp]: the three ring slots as they stand at `time0`
    (slot i holds what level i of a [3][...] array holds) and one spare.  The 4-plane padding in
    x is the ghost zone: a pair of steps reads u(t) four planes and u(t-1) two planes beyond the
    owned range and recomputes u(t+1) on two ghost planes, bit for bit as the neighbour does (and
    keeps them: the next pair reads them as its u(t-1)).  After a pair the outermost 4 owned planes
    of u(t+2) go to each neighbour; after a single step (odd counts) those of the new level.  The schedule is the
    two-chain one of SlabPropagator with a boundary zone of 4 planes.

    In the reference's ring a level inherits the halo of the slot it overwrites (u(t+2) that of
    u(t-1)); with four rotating buffers that is kept by copying the shell of `prev` into the
    buffer that receives u(t+2) before every pair, unless `uniform_halos` promises that all slots
    carry the same halo values and no source corner can land in the halo layer."""

    def __init__(self, layout, rank, levels, m, src, coords, *, dt, h, o, stepper, comm, time0=0,
                 uniform_halos=False):
        self.layout, self.rank = layout, rank
        self.levels, self.m = list(levels), m
        self.nx = layout.nx_loc(rank)
        self.x_off = layout.x_off(rank)
        self.dt, self.h, self.o = dt, h, o
        self.stepper, self.comm = stepper, comm
        self.nsrc = 0 if src is None else int(src.shape[1])
        self.src, self.coords = src, coords
        self.uniform_halos = uniform_halos
        if layout.world > 1 and min(layout.counts) < PAIR_GHOST:
            raise ValueError("slabs thinner than 4 planes cannot take two steps per exchange")
        self.cur, self.prev, self.third, self.spare = time0 % 3, (time0 + 2) % 3, (time0 + 1) % 3, 3
        exchanging = layout.world > 1 or getattr(comm, "has_neighbours", False)
        self.overlap = exchanging and self.nx >= 2 * PAIR_GHOST + 1 and stepper.pairs_can_overlap(self)
        stepper.bind_pairs(self)
        self.edges_first = self.overlap and getattr(stepper, "zone_fused", False) and self.nx >= 3 * PAIR_GHOST

    def _ranges(self):
        nx, gw = self.nx, PAIR_GHOST
        if self.overlap and self.edges_first:
            gw = max(PAIR_GHOST, min(nx // 3, 96))
        if self.overlap:
            return [(gw, nx - gw - 1, False), (0, gw - 1, True), (nx - gw, nx - 1, True)]
        return [(0, nx - 1, True)]

    def pair(self, time):
        """Steps `time` and `time + 1`."""
        st = self.stepper
        st.begin_step()
        if not self.uniform_halos:
            st.copy_shell(self.spare, self.prev)
        ranges = self._ranges()
        if self.overlap and self.edges_first:
            # Large cross-sections: the outer thirds of the slab (ordinary x-chunks of the fused sweep,
            # no extra lead-in planes) first, the messages as soon as those are done, the middle third
            # behind them; a sweep of several rounds of workgroups leaves RCCL's kernel room by itself.
            for lo, hi, boundary in ranges[1:]:
                st.two_steps(lo, hi, self.cur, self.prev, self.third, self.spare, time, boundary)
            st.boundary_done()
            self.comm.exchange_planes(self, [(self.spare, PAIR_GHOST)])
            st.interior_after_boundary()
            lo, hi, boundary = ranges[0]
            st.two_steps(lo, hi, self.cur, self.prev, self.third, self.spare, time, boundary)
            st.interior_done()
        else:
            for lo, hi, boundary in ranges:
                st.two_steps(lo, hi, self.cur, self.prev, self.third, self.spare, time, boundary)
                if not boundary:
                    st.interior_done()
            st.boundary_done()
            # u(t+1) needs no message: its two ghost planes were computed here, as the neighbour computes them
            self.comm.exchange_planes(self, [(self.spare, PAIR_GHOST)])
        self.cur, self.prev, self.third, self.spare = self.spare, self.third, self.cur, self.prev

    def single(self, time):
        """One step; the new level is handed over 4 planes deep so that a pair can follow."""
        st = self.stepper
        st.begin_step()
        for lo, hi, boundary in self._ranges():
            st.one_step(lo, hi, self.cur, self.prev, self.third, time, boundary)
            if not boundary:
                st.interior_done()
        st.boundary_done()
        self.comm.exchange_planes(self, [(self.third, PAIR_GHOST)])
        self.cur, self.prev, self.third = self.third, self.cur, self.prev

    def run(self, time_m, time_M):
        time = time_m
        while time + 1 <= time_M:
            self.pair(time)
            time += 2
        if time <= time_M:
            self.single(time)
        self.stepper.finish()

    def reset(self, time0=0):
        """Zero wavefield in all four buffers, roles as at `time0` (a benchmark repetition, main.cpp:389)."""
        self.stepper.finish()
        for lv in self.levels:
            lv.zero_()
        self.cur, self.prev, self.third, self.spare = time0 % 3, (time0 + 2) % 3, (time0 + 1) % 3, 3

    def ring(self):
        """The three newest levels in ring order: element i is what slot i of a [3][...] array holds
        after the last step (roles: cur = u(T+1), prev = u(T), third = u(T-1))."""
        return self.cur, self.prev, self.third

    # planes of a level sent to / received from the neighbours (contiguous views)
    def send_left(self, lvl, n=GHOST):
        return self.levels[lvl][HALO:HALO + n]

    def send_right(self, lvl, n=GHOST):
        return self.levels[lvl][HALO + self.nx - n:HALO + self.nx]

    def recv_left(self, lvl, n=GHOST):
        return self.levels[lvl][HALO - n:HALO]

    def recv_right(self, lvl, n=GHOST):
        return self.levels[lvl][HALO + self.nx:HALO + self.nx + n]

    def interior_cells(self):
        nyp, nzp = self.levels[0].shape[1], self.levels[0].shape[2]
        lo, hi, _ = self._ranges()[0]
        return float(hi - lo + 1) * (nyp - 2 * HALO) * (nzp - 2 * HALO)

    def mean_interior_kernel_ms(self):
        return self.stepper.mean_interior_kernel_ms()


def init_rccl(device, rank=None, world=None, init_method=None):
    """Process group for the ghost messages: backend "nccl" (= RCCL), communicator created eagerly
    on `device`.  RCCL's stream keeps the default priority: with a high-priority stream the
    first exchange was seen to never complete on an MI355X (device-side wait that no kernel
    resolves), and the messages are hidden behind the interior sweep either way."""
    import torch.distributed as dist
    kw = {} if rank is None else {"rank": rank, "world_size": world}
    if init_method is not None:      # e.g. "file:///path": rendezvous without a TCP port
        kw["init_method"] = init_method
    dist.init_process_group("nccl", device_id=device, **kw)
    return dist


def standard_pair_benchmark(n_per_rank, ny, nz, nt, rank, world, device, variant=-1, global_x=0,
                            self_exchange=False):
    """The weak-scaling benchmark state of SlabPropagator.standard_benchmark for the two-steps-per-
    exchange schedule: four zero level buffers, m = 1.5, one Ricker source inside the grid (zero
    halos everywhere and no halo-layer corner: uniform_halos)."""
    import torch
    import torch.distributed as dist
    NX = global_x if global_x else n_per_rank * world
    layout = SlabLayout(NX, world)
    nx = layout.nx_loc(rank)
    shape = (nx + 2 * HALO, ny + 2 * HALO, nz + 2 * HALO)
    levels = [torch.zeros(shape, dtype=torch.float32, device=device) for _ in range(4)]
    m = torch.full(shape, 1.5, dtype=torch.float32, device=device)
    src = torch.from_numpy(abi.fill_ricker(nt, 1)).to(device)
    coords = torch.from_numpy(abi.fill_source_coords(1, NX, ny, nz)).to(device)
    stepper = HipStepper(device, variant=variant, time_interior=True)
    if self_exchange:
        comm = TorchComm(dist, rank, world, stepper, left=rank, right=rank)
    else:
        comm = TorchComm(dist, rank, world, stepper)
    return SlabPairPropagator(layout, rank, levels, m, src, coords, dt=1e-3, h=(0.1, 0.1, 0.1), o=(0.0, 0.0, 0.0),
                              stepper=stepper, comm=comm, time0=0, uniform_halos=True)


class HipStepper:
    """Kernels from libfdtd_hip.so on two HIP streams.

    `edge` carries the boundary chain of a step (outermost planes, their sources, then the ghost
    messages, which the communication library runs on its own stream behind that point);
    `inner` carries the interior chain.  Per step t:
        edge : wait interior(t-1), wait messages(t-1); boundary update; scatter; record boundary(t)
        inner: wait boundary(t-1);                     interior update; scatter; record interior(t)
    Every wait names an event recorded earlier in host order, and these waits cover the
    write-after-read hazards of the 3-level ring as well (a level is rewritten three steps after
    it was written; its readers of steps t-1 and t-2 precede the awaited events on their streams).
    A slab too thin to split (or a single rank) runs everything on `inner`."""

    TIMING_STRIDE = 4   # every 4th interior launch is bracketed by timing events

    def __init__(self, device, variant=-1, time_interior=False):
        import torch
        self.torch = torch
        self.lib = abi.lib()
        self.device = device
        self.variant = variant
        self.inner = torch.cuda.current_stream(device)
        self.edge = self.inner
        self.ev_boundary = None     # last recorded boundary / interior events
        self.ev_interior = None
        self.pending = []           # message requests the next boundary update waits for
        # Two steps per exchange only: the interior sweep also waits for the previous messages.  It
        # does not need them, but RCCL's waves do not fit next to the fused sweep's (264 + 2 x 208
        # registers per SIMD lane of 512), so messages issued while a sweep runs complete when it drains; if
        # the next sweep started before that, the boundary zone would fall a whole sweep behind and
        # stay there (its result is what the sweep after needs).
        self.hold_interior = False
        self.time_interior = time_interior
        self.order, self.damp_ptr = 2 * GHOST, None
        self.timing = []
        self._launches = 0
        self._geoms = {}

    def bind(self, prop):
        self.p = prop
        nxp, nyp, nzp = prop.u.shape[1:]
        H = prop.halo
        self.geom = abi.Geom(nxp, nyp, nzp, 0, prop.nx - 1, 0, nyp - 2 * H - 1, 0, nzp - 2 * H - 1,
                             prop.dt, prop.h[0], prop.h[1], prop.h[2], prop.o[0], prop.o[1], prop.o[2])
        self.u_ptr, self.m_ptr = prop.u.data_ptr(), prop.m.data_ptr()
        self.order = prop.order
        self.damp_ptr = None if prop.damp is None else prop.damp.data_ptr()
        self.src_ptr = prop.src.data_ptr() if prop.nsrc else None
        self.coords_ptr = prop.coords.data_ptr() if prop.nsrc else None
        if prop.overlap:
            self.edge = self.torch.cuda.Stream(self.device)
            self.edge.wait_stream(self.inner)       # the fields were produced on the current stream
        self.split = prop.overlap
        self._sh = {True: C.c_void_p(self.edge.cuda_stream), False: C.c_void_p(self.inner.cuda_stream)}

    def begin_step(self):
        if self.split:
            if self.ev_interior is not None:
                self.edge.wait_event(self.ev_interior)
            if self.ev_boundary is not None:
                self.inner.wait_event(self.ev_boundary)
        # the boundary update reads the ghosts delivered by the previous step's messages
        if self.pending:
            with self.torch.cuda.stream(self.edge):
                for req in self.pending:
                    req.wait()          # stream-side wait; the host does not block
            if self.split and self.hold_interior:
                with self.torch.cuda.stream(self.inner):
                    for req in self.pending:
                        req.wait()
            self.pending = []

    def stencil(self, x_lo, x_hi, t0, t1, t2, boundary):
        if x_hi < x_lo:
            return
        g = self._geoms.get((x_lo, x_hi))
        if g is None:
            g = abi.Geom.from_buffer_copy(self.geom)
            g.x_m, g.x_M = x_lo, x_hi
            self._geoms[(x_lo, x_hi)] = g
        stream = self.edge if boundary else self.inner
        timed = self.time_interior and (not boundary or not self.split)
        if timed:
            self._launches += 1
            timed = self._launches % self.TIMING_STRIDE == 0
        if timed:
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        if self.damp_ptr is not None:      # the damped update (every space order)
            rc = self.lib.fdtd_hip_stencil_step_damped_order(C.byref(g), self.m_ptr, self.damp_ptr, self.u_ptr, t0, t1, t2,
                                                             self.order, self._sh[boundary])
        elif self.order != 2 * GHOST:      # space orders 6 .. 12
            rc = self.lib.fdtd_hip_stencil_step_order(C.byref(g), self.m_ptr, self.u_ptr, t0, t1, t2, self.order,
                                                      self._sh[boundary])
        else:
            rc = self.lib.fdtd_hip_stencil_step(C.byref(g), self.m_ptr, self.u_ptr, t0, t1, t2, self.variant,
                                                self._sh[boundary])
        if rc:
            raise RuntimeError(f"the stencil sweep failed with hipError {rc}")
        if timed:
            e1.record(stream)
            self.timing.append((e0, e1))

    def inject(self, region, t2, time, boundary):
        p = self.p
        if self.order != 2 * GHOST:
            rc = self.lib.fdtd_hip_source_inject_slab_order(
                C.byref(self.geom), self.m_ptr, self.src_ptr, self.coords_ptr, self.u_ptr, t2, time,
                0, p.nsrc - 1, p.src.shape[1], p.coords.shape[1], p.x_off, region[0], region[1], self.order,
                self._sh[boundary])
        else:
            rc = self.lib.fdtd_hip_source_inject_slab(
                C.byref(self.geom), self.m_ptr, self.src_ptr, self.coords_ptr, self.u_ptr, t2, time,
                0, p.nsrc - 1, p.src.shape[1], p.coords.shape[1], p.x_off, region[0], region[1], self._sh[boundary])
        if rc:
            raise RuntimeError(f"the slab scatter failed with hipError {rc}")

    # ---- two steps per exchange (SlabPairPropagator) ----
    def pairs_can_overlap(self, prop):
        """The boundary and interior calls may run side by side only when they are the fused sweep,
        which stores nothing outside its range; the two-sweep fallback rewrites u(t+1) on two planes
        of the neighbouring range, so it keeps to one chain."""
        import os
        if os.environ.get("FDTD_SLAB_OVERLAP", "1") == "0":
            return False
        if self.variant not in (-1, abi.VARIANTS["march2"]) or prop.nsrc > 8:
            return False      # (more than 8 sources: the fused sweep needs the planes beside its range in memory)
        # the interior range of the split slab, as the library will see it
        nxp, nyp, nzp = prop.levels[0].shape
        g = abi.Geom(nxp, nyp, nzp, PAIR_GHOST, prop.nx - PAIR_GHOST - 1, 0, nyp - 2 * HALO - 1, 0, nzp - 2 * HALO - 1,
                     prop.dt, prop.h[0], prop.h[1], prop.h[2], prop.o[0], prop.o[1], prop.o[2])
        return bool(self.lib.fdtd_hip_two_steps_fuses(C.byref(g), prop.nsrc, abi.VARIANTS["march2"]))

    def bind_pairs(self, prop):
        self.p = prop
        nxp, nyp, nzp = prop.levels[0].shape
        self.geom = abi.Geom(nxp, nyp, nzp, 0, prop.nx - 1, 0, nyp - 2 * HALO - 1, 0, nzp - 2 * HALO - 1,
                             prop.dt, prop.h[0], prop.h[1], prop.h[2], prop.o[0], prop.o[1], prop.o[2])
        self.lv_ptr = [t.data_ptr() for t in prop.levels]
        self.m_ptr = prop.m.data_ptr()
        self.src_ptr = prop.src.data_ptr() if prop.nsrc else None
        self.coords_ptr = prop.coords.data_ptr() if prop.nsrc else None
        self.src_stride = prop.src.shape[1] if prop.nsrc else 1
        self.coord_stride = prop.coords.shape[1] if prop.nsrc else 3
        if prop.overlap:
            self.edge = self.torch.cuda.Stream(self.device)
            self.edge.wait_stream(self.inner)
        self.split = prop.overlap
        self._sh = {True: C.c_void_p(self.edge.cuda_stream), False: C.c_void_p(self.inner.cuda_stream)}
        self.flags = 1 if prop.uniform_halos else 0
        # The message kernel cannot be placed beside the interior sweep; a shorter last x-chunk ends
        # early and hands it CUs for the tail of the sweep (FDTD_SLAB_TAIL_PLANES, 0 = equal chunks).
        # 24 planes are about 115 us at a 512 x 512 cross-section; the sweep as a whole is no slower.
        self.tail_planes = max(0, min(255, int(os.environ.get("FDTD_SLAB_TAIL_PLANES", "24"))))
        self.hold_interior = True
        # The boundary zones are updated next to the interior.  Small cross-sections: two small sweeps
        # with their u(t+1) planes built in scratch, which run in the registers and LDS the interior
        # sweep leaves free.  From 256 tiles of the fused sweep on (1024 x 1024 cells) those small
        # sweeps are starved for so long that the fused sweep itself, which stores nothing outside
        # its range either, is the cheaper way (FDTD_SLAB_ZONE_FUSED=0/1 overrides).
        tiles = ((nyp - 2 * HALO + 7) // 8) * ((nzp - 2 * HALO + 511) // 512)
        zf = os.environ.get("FDTD_SLAB_ZONE_FUSED")
        self.zone_fused = prop.overlap and (tiles >= 256 if zf is None else zf != "0")
        self.scratch = None
        if prop.overlap and not self.zone_fused:
            self.scratch = self.torch.empty((PAIR_GHOST + 4, nyp, nzp), dtype=self.torch.float32, device=self.device)

    def _range_geom(self, x_lo, x_hi):
        g = self._geoms.get((x_lo, x_hi))
        if g is None:
            g = abi.Geom.from_buffer_copy(self.geom)
            g.x_m, g.x_M = x_lo, x_hi
            self._geoms[(x_lo, x_hi)] = g
        return g

    def _timed(self, boundary, stream):
        timed = self.time_interior and (not boundary or not self.split)
        if timed:
            self._launches += 1
            timed = self._launches % self.TIMING_STRIDE == 0
        if not timed:
            return None
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        return e0, e1

    def two_steps(self, x_lo, x_hi, cur, prev, n1, n2, time, boundary):
        if x_hi < x_lo:
            return
        p, stream = self.p, (self.edge if boundary else self.inner)
        ev = self._timed(boundary, stream)
        # Side by side with the boundary zones the interior has to be the fused sweep whatever its
        # length (it stores nothing outside its range); the zones themselves go through scratch.
        variant = abi.VARIANTS["march2"] if (self.split and (not boundary or self.zone_fused)) else self.variant
        rc = self.lib.fdtd_hip_two_steps_slab(
            C.byref(self._range_geom(x_lo, x_hi)), self.m_ptr, self.src_ptr, self.coords_ptr, self.lv_ptr[cur],
            self.lv_ptr[prev], self.lv_ptr[n1], self.lv_ptr[n2], time, 0, p.nsrc - 1, self.src_stride,
            self.coord_stride, variant,
            self.flags | (4 if x_lo == 0 else 0) | (8 if x_hi == p.nx - 1 else 0)
            | ((self.tail_planes << 8) if (self.split and not boundary) else 0),
            p.x_off, p.layout.NX,
            self.scratch.data_ptr() if (boundary and self.scratch is not None) else None,
            self.scratch.numel() if (boundary and self.scratch is not None) else 0, self._sh[boundary])
        if rc:
            raise RuntimeError(f"fdtd_hip_two_steps_slab failed with hipError {rc}")
        if ev:
            ev[1].record(stream)
            self.timing.append(ev)

    def one_step(self, x_lo, x_hi, cur, prev, nxt, time, boundary):
        if x_hi < x_lo:
            return
        p = self.p
        variant = -1 if self.variant == abi.VARIANTS["march2"] else self.variant
        rc = self.lib.fdtd_hip_one_step_slab(
            C.byref(self._range_geom(x_lo, x_hi)), self.m_ptr, self.src_ptr, self.coords_ptr, self.lv_ptr[cur],
            self.lv_ptr[prev], self.lv_ptr[nxt], time, 0, p.nsrc - 1, self.src_stride, self.coord_stride,
            variant, p.x_off, p.layout.NX, self._sh[boundary])
        if rc:
            raise RuntimeError(f"fdtd_hip_one_step_slab failed with hipError {rc}")

    def copy_shell(self, dst, src):
        # in the boundary chain: its event orders the copy before the next pair's interior reads
        rc = self.lib.fdtd_hip_copy_shell(C.byref(self.geom), self.lv_ptr[dst], self.lv_ptr[src], 0, self._sh[True])
        if rc:
            raise RuntimeError(f"fdtd_hip_copy_shell failed with hipError {rc}")
        if self.split:
            # the interior call adds the source corners that land in the halo layer of its own planes
            # to this very shell: not before the copy has been through
            ev = self.torch.cuda.Event()
            ev.record(self.edge)
            self.inner.wait_event(ev)

    def boundary_done(self):
        if self.split:
            self.ev_boundary = self.torch.cuda.Event()
            self.ev_boundary.record(self.edge)

    def interior_done(self):
        self.ev_interior = self.torch.cuda.Event()
        self.ev_interior.record(self.inner)

    def interior_after_boundary(self):
        """Orders the interior call of this pair behind its boundary calls (same pair)."""
        if self.split and self.ev_boundary is not None:
            self.inner.wait_event(self.ev_boundary)

    def finish(self):
        self.begin_step()
        self.torch.cuda.synchronize(self.device)

    def mean_interior_kernel_ms(self):
        self.torch.cuda.synchronize(self.device)
        if not self.timing:
            return float("nan")
        # skip the untimed warm-up launches at the front of the list if there are enough samples
        ev = self.timing[len(self.timing) // 4:]
        return float(np.mean([a.elapsed_time(b) for a, b in ev]))


class TorchComm:
    """Ghost exchange with torch.distributed point-to-point messages (RCCL on GPUs).

    With the "nccl" backend the batch is issued from the stepper's boundary stream right after
    the boundary work: RCCL's stream waits for that point and no further, so the messages run
    next to the interior update, and the requests are handed to the stepper, whose next boundary
    update waits for them on the device.  With "gloo" (CPU tests, or several ranks
    sharing one GPU) device tensors are staged through the host."""

    _DEFAULT = object()

    def __init__(self, dist, rank, world, stepper=None, left=_DEFAULT, right=_DEFAULT):
        """`left` / `right`: peer ranks of the two ghost blocks (None = physical boundary).  The
        default is the chain r-1 / r+1; a rank may name itself (RCCL matches a send to self with
        the receive from self of the same batch, in order), which lets one process on one GPU
        drive the full RCCL schedule."""
        self.dist, self.rank, self.world, self.stepper = dist, rank, world, stepper
        self.left = (rank - 1 if rank > 0 else None) if left is self._DEFAULT else left
        self.right = (rank + 1 if rank < world - 1 else None) if right is self._DEFAULT else right
        self.has_neighbours = self.left is not None or self.right is not None
        self.backend = dist.get_backend() if self.has_neighbours else None
        self._ops = {}

    def exchange_planes(self, prop, items):
        """items: [(level index, planes per side), ...] — every listed block goes to both neighbours,
        in list order (both sides list the same order, which is what matches sends to receives)."""
        if not self.has_neighbours:
            return
        dist, st = self.dist, self.stepper
        on_gpu = prop.levels[0].is_cuda
        if on_gpu and self.backend == "nccl":
            key = tuple(items)
            ops = self._ops.get(key)
            if ops is None:
                ops = []
                for peer, snd, rcv in ((self.left, prop.send_left, prop.recv_left),
                                       (self.right, prop.send_right, prop.recv_right)):
                    if peer is None:
                        continue
                    for lvl, n in items:
                        ops.append(dist.P2POp(dist.isend, snd(lvl, n), peer))
                        ops.append(dist.P2POp(dist.irecv, rcv(lvl, n), peer))
                self._ops[key] = ops
            with st.torch.cuda.stream(st.edge):
                st.pending = list(dist.batch_isend_irecv(ops))
            return
        if on_gpu:
            st.torch.cuda.synchronize(st.device)
        stage = (lambda t: t.cpu().contiguous()) if on_gpu else (lambda t: t.contiguous())
        reqs, recvs = [], []
        for peer, snd, rcv in ((self.left, prop.send_left, prop.recv_left),
                               (self.right, prop.send_right, prop.recv_right)):
            if peer is None:
                continue
            for k, (lvl, n) in enumerate(items):
                reqs.append(dist.isend(stage(snd(lvl, n)), peer, tag=k))
                buf = stage(rcv(lvl, n))
                reqs.append(dist.irecv(buf, peer, tag=k))
                recvs.append((rcv(lvl, n), buf))
        for q in reqs:
            q.wait()
        for dst, buf in recvs:
            dst.copy_(buf)

    def exchange(self, prop, lvl):
        if not self.has_neighbours:
            return
        dist = self.dist
        st = self.stepper
        on_gpu = prop.u.is_cuda
        if on_gpu and self.backend == "nccl":
            torch = st.torch
            ops = self._ops.get(lvl)
            if ops is None:
                ops = []
                if self.left is not None:
                    ops.append(dist.P2POp(dist.isend, prop.send_left(lvl), self.left))
                    ops.append(dist.P2POp(dist.irecv, prop.recv_left(lvl), self.left))
                if self.right is not None:
                    ops.append(dist.P2POp(dist.isend, prop.send_right(lvl), self.right))
                    ops.append(dist.P2POp(dist.irecv, prop.recv_right(lvl), self.right))
                self._ops[lvl] = ops
            with torch.cuda.stream(st.edge):
                st.pending = list(dist.batch_isend_irecv(ops))
            return
        # host-staged path
        if on_gpu:
            st.torch.cuda.synchronize(st.device)
        stage = (lambda t: t.cpu().contiguous()) if on_gpu else (lambda t: t.contiguous())
        reqs, recvs = [], []
        if self.left is not None:
            reqs.append(dist.isend(stage(prop.send_left(lvl)), self.left))
            buf = stage(prop.recv_left(lvl))
            reqs.append(dist.irecv(buf, self.left))
            recvs.append((prop.recv_left(lvl), buf))
        if self.right is not None:
            reqs.append(dist.isend(stage(prop.send_right(lvl)), self.right))
            buf = stage(prop.recv_right(lvl))
            reqs.append(dist.irecv(buf, self.right))
            recvs.append((prop.recv_right(lvl), buf))
        for q in reqs:
            q.wait()
        for dst, buf in recvs:
            dst.copy_(buf)
