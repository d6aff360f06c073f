// fdtd_multigpu.hip — the operator of main.cpp:53-72 on several GPUs of one node, in one process.
// The reference is single-device (no counterpart); the arithmetic per cell is that of
// openacc.cpp:159-164,185-196, and the union of the slabs is bit for bit what one device computes.
//
// Two layers:
//   fdtd_hip_multigpu_*   a session that owns the slabs on the devices: create, fill or upload, advance,
//                         gather, destroy.  Nothing of the wavefield lives in host memory between calls,
//                         so a 2048^3 grid (139 GB of caller arrays) needs no host copy at all.
//   Kernel_HIP_MultiGPU   the 24-argument operator on host arrays: a thin caller of the session.
//
// Decomposition: the x range (the slowest storage axis) is cut into one slab per device.  A device
// holds its planes with the ABI's 4-plane padding on both sides; on the side of a neighbour those
// four planes are ghost planes.  Every ghost block is one contiguous span of whole planes, so there
// are no pack or unpack kernels.
//
// Schedule: two time steps per exchange (fdtd_two_steps_impl, slab form).  Per pair and device:
//   edge  stream: wait for the previous pair's interior sweep and for the ghost planes the neighbours
//                 pushed; [shell copy]; boundary zones of the slab; record "boundary done"
//   inner stream: wait for the previous pair's boundary zones; interior sweep; record "interior done"
//   copy  stream: wait for "boundary done" and for each neighbour's "ready to receive"; push the
//                 outermost 4 owned planes of u(t+2) into the neighbour's ghost planes with
//                 hipMemcpyPeerAsync — a copy-engine transfer over xGMI that occupies no compute unit,
//                 so it runs next to the interior sweep without competing for registers or LDS.
// u(t+1) needs no message: its two ghost planes are recomputed locally, bit for bit as their owner
// computes them.  Large cross-sections (from 256 tiles of the fused sweep on) update the outer
// thirds of the slab first and the middle third behind them instead of 4-plane zones in scratch.
//
// Who issues: one host thread per slab (default).  A pair is 8 to 15 launches, 6 to 8 event calls and up
// to two peer copies per device — 40 to 80 us of host time — so a single issuing thread sets the pace as
// soon as 8 devices finish a pair in less than that sum (cross-sections of 512 x 512: 0.6 ms per pair),
// while at 2048 x 2048 (5 ms per pair) it does not matter.  The threads order themselves through two
// counters per slab: "ready to receive recorded for operation k" and "pushes recorded for operation k".
// A stream wait must name an event that has been recorded already (a wait on an event that was never
// recorded passes at once), so a thread reads its neighbour's counter before it makes a stream wait for
// the neighbour's event; the data dependencies themselves keep an event from being recorded again
// before its last waiter has been issued (a slab cannot record "ready" for pair k+1 before both
// neighbours have issued their pushes of pair k, which follow their waits on "ready" of pair k).
// FDTD_MULTIGPU_THREADS=0 issues everything from the calling thread, device after device.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fdtd_device.h"
#include "fdtd_host.h"

extern "C" void fdtd_fill_ricker(float *src, int T, int S, float dt);
extern "C" void fdtd_fill_source_coords(float *coords, int S, int nx, int ny, int nz);

namespace {

constexpr int PAIR_GHOST = 4;   // planes pushed per side and pair

__global__ __launch_bounds__(256) void fill_value_kernel(float *p, size_t n, float v) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// max |v| over n floats as the bit pattern of a non-negative float (ordered like unsigned integers), and
// the number of NaN / Inf values; out[0] = max bits, out[1] = count
__global__ __launch_bounds__(256) void max_abs_kernel(const float *p, size_t n, unsigned *out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned best = 0, bad = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned b = __float_as_uint(p[i]) & 0x7fffffffu;
    if (b >= 0x7f800000u) ++bad;
    else best = b > best ? b : best;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned other = (unsigned)__shfl_down((int)best, o);
    best = other > best ? other : best;
    bad += (unsigned)__shfl_down((int)bad, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out, best);
    if (bad) atomicAdd(out + 1, bad);
  }
}

// sum of n 32-bit words as a 64-bit integer (wraps; independent of the order of the additions, hence of how the
// domain is cut into slabs): a checksum that two runs of the same inputs can be compared by
__global__ __launch_bounds__(256) void word_sum_kernel(const unsigned *p, size_t n, unsigned long long *out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += p[i];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

struct Dev {
  int id = 0;                 // HIP device
  int nx = 0, off = 0;        // owned planes and the first of them (interior index)
  int left = -1, right = -1;  // neighbour slabs (indices into the slab list), -1 = physical boundary
  float *lv[4] = {nullptr, nullptr, nullptr, nullptr};
  float *m = nullptr, *src = nullptr, *coords = nullptr, *scratch = nullptr;
  float *damp = nullptr;        // damping field of the slab (fdtd_hip_multigpu_set_damping), else null
  unsigned *reduce = nullptr;   // two words for max_abs_kernel
  long scratch_floats = 0;
  size_t src_floats = 0, coords_floats = 0;
  hipStream_t inner = nullptr, edge = nullptr, copy = nullptr;
  hipEvent_t ev_interior = nullptr, ev_boundary = nullptr, ev_ready = nullptr, ev_shell = nullptr;
  hipEvent_t ev_push_left = nullptr, ev_push_right = nullptr;   // this slab's pushes to its neighbours
  bool interior_recorded = false, boundary_recorded = false;
  // operations (pairs or single steps, counted over the life of the session) for which this slab has
  // recorded "ready to receive" / its pushes
  std::atomic<long> ready_seq{0}, push_seq{0};
  fdtd_geom g;                // local geometry: x range 0 .. nx-1 of nx+8 planes
  double issue_ns = 0.0, wait_ns = 0.0;   // host time of this slab's issuing thread: HIP calls / waiting for a neighbour's counter
};

// one operation of the time loop, with the roles of the four level buffers while it runs
struct Op {
  bool pair;
  int time;
  int cur, prev, third, spare;
  long k;   // 1, 2, ... over the life of the session
};

double now_ns() {
  using ns = std::chrono::duration<double, std::nano>;
  return ns(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int mod3(int t) { return ((t % 3) + 3) % 3; }

// Devices of an operator call: FDTD_MULTIGPU_DEVICES (comma-separated HIP device numbers; a device may be
// named more than once, it then carries several slabs), else FDTD_NGPUS devices (default: all with
// deviceid = -1, one otherwise) starting at deviceid.
int pick_devices(int deviceid, std::vector<int> *out) {
  int count = 0;
  FDTD_HIP_CHECK(hipGetDeviceCount(&count));
  if (count <= 0) return (int)hipErrorNoDevice;
  if (const char *list = getenv("FDTD_MULTIGPU_DEVICES")) {
    std::string s(list);
    size_t at = 0;
    while (at < s.size()) {
      const size_t comma = s.find(',', at);
      const std::string tok = s.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
      if (!tok.empty()) {
        const int id = atoi(tok.c_str());
        if (id < 0 || id >= count) return (int)hipErrorInvalidDevice;
        out->push_back(id);
      }
      if (comma == std::string::npos) break;
      at = comma + 1;
    }
    if (!out->empty()) return 0;
  }
  const char *e = getenv("FDTD_NGPUS");
  int n = e ? atoi(e) : (deviceid == -1 ? count : 1);
  if (n < 1) n = 1;
  if (n > count) n = count;
  const int first = deviceid == -1 ? 0 : deviceid;
  if (first < 0 || first >= count) return (int)hipErrorInvalidDevice;
  for (int i = 0; i < n; ++i) out->push_back((first + i) % count);
  return 0;
}

int g_multigpu_last_slabs = 0;

}  // namespace

// The session.  All members are touched by the thread that calls the fdtd_hip_multigpu_* functions
// (calls on one session are sequential); the issuing threads read them while a job runs.
struct fdtd_multigpu {
  std::vector<Dev> d;
  fdtd_geom G;                // the whole domain (x range 0 .. NX-1)
  int NX = 0;
  size_t plane = 0;
  int flags = 0;
  bool exchange = true, want_split = true, threaded = true;
  // sources
  int nsrc = 0, p_src_m = 0, p_src_M = -1, src_stride = 1, coord_stride = 3, src_steps = 0;
  bool sources_inside = true;
  // schedule, decided by prepare() once sources and fields are known
  bool prepared = false, follow = false, shells_identical = true, split = false, edges_first = false;
  // ring
  int cur = 0, prev = 2, third = 1, spare = 3, t_now = 0;
  bool fields_set = false;
  long seq = 0;               // operations issued so far
  long pairs_timed = 0, singles_timed = 0;
  double advance_wall_ns = 0.0;
  // issuing threads
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  long job_gen = 0;
  int done = 0;
  bool quit = false;
  std::function<int(Dev &)> job;
  std::vector<int> job_rc;
  std::atomic<bool> abort_flag{false};
  bool poisoned = false;

  // ---- ranges of a slab: interior first (split), then the two boundary zones ----
  struct Range { int lo, hi; bool boundary; };
  std::vector<Range> ranges(const Dev &v) const {
    if (!split) return {{0, v.nx - 1, true}};
    int gw = PAIR_GHOST;
    if (edges_first) gw = v.nx / 3 < 96 ? v.nx / 3 : 96;
    if (gw < PAIR_GHOST) gw = PAIR_GHOST;
    return {{gw, v.nx - gw - 1, false}, {0, gw - 1, true}, {v.nx - gw, v.nx - 1, true}};
  }

  bool wait_seq(const std::atomic<long> &a, long want, Dev &v) {
    if (a.load(std::memory_order_acquire) >= want) return true;
    const double t0 = now_ns();
    int spins = 0;
    while (a.load(std::memory_order_acquire) < want) {
      if (abort_flag.load(std::memory_order_relaxed)) return false;
      if (++spins > 200) std::this_thread::yield();
    }
    v.wait_ns += now_ns() - t0;
    return true;
  }

  int begin(Dev &v, const Op &op) {
    if (split) {
      if (v.interior_recorded) FDTD_HIP_CHECK(hipStreamWaitEvent(v.edge, v.ev_interior, 0));
      if (v.boundary_recorded) FDTD_HIP_CHECK(hipStreamWaitEvent(v.inner, v.ev_boundary, 0));
    }
    if (!exchange || op.k < 2) return 0;
    // the ghost planes the neighbours pushed behind the previous operation
    if (v.left >= 0) {
      Dev &nb = d[(size_t)v.left];
      if (!wait_seq(nb.push_seq, op.k - 1, v)) return (int)hipErrorUnknown;
      FDTD_HIP_CHECK(hipStreamWaitEvent(v.edge, nb.ev_push_right, 0));
    }
    if (v.right >= 0) {
      Dev &nb = d[(size_t)v.right];
      if (!wait_seq(nb.push_seq, op.k - 1, v)) return (int)hipErrorUnknown;
      FDTD_HIP_CHECK(hipStreamWaitEvent(v.edge, nb.ev_push_left, 0));
    }
    return 0;
  }

  int two_steps(Dev &v, const Range &r, const Op &op) {
    hipStream_t st = r.boundary ? v.edge : v.inner;
    fdtd_geom g = v.g;
    g.x_m = r.lo; g.x_M = r.hi;
    // side by side with other ranges a call must store nothing outside its range: the fused sweep does
    // not (forced whatever the length of the range), the two-sweep path needs scratch for that
    const bool fused_call = split && (!r.boundary || edges_first);
    const int variant = fused_call ? FDTD_VARIANT_MARCH2 : FDTD_VARIANT_AUTO;
    const int fl = (sources_inside ? FDTD_TWO_STEPS_SOURCES_INSIDE : 0) | (r.lo == 0 ? FDTD_TWO_STEPS_KEEP_LOW : 0) |
                   (r.hi == v.nx - 1 ? FDTD_TWO_STEPS_KEEP_HIGH : 0);
    const bool use_scratch = split && r.boundary && !edges_first;
    return fdtd_two_steps_impl(&g, v.m, v.src, v.coords, v.lv[op.cur], v.lv[op.prev], v.lv[op.third], v.lv[op.spare],
                               op.time, nsrc ? p_src_m : 0, nsrc ? p_src_M : -1, src_stride, coord_stride, variant, fl,
                               v.off, NX, st, nullptr, nullptr, use_scratch ? v.scratch : nullptr,
                               use_scratch ? v.scratch_floats : 0, v.damp);
  }

  // everything of one operation that a slab issues before its pushes
  int compute(Dev &v, const Op &op) {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    int rc = begin(v, op);
    if (rc) return rc;
    if (op.pair && follow) {
      rc = fdtd_hip_copy_shell(&v.g, v.lv[op.spare], v.lv[op.prev], 0, v.edge);
      if (rc) return rc;
      if (split) {   // the interior call scatters halo-layer corners of its own planes into this very shell
        FDTD_HIP_CHECK(hipEventRecord(v.ev_shell, v.edge));
        FDTD_HIP_CHECK(hipStreamWaitEvent(v.inner, v.ev_shell, 0));
      }
    }
    FDTD_HIP_CHECK(hipEventRecord(v.ev_ready, v.edge));   // from here on the neighbours may write the ghost planes
    v.ready_seq.store(op.k, std::memory_order_release);
    const std::vector<Range> rs = ranges(v);
    if (!op.pair) {
      // one step; the new level is handed over 4 planes deep so that a pair can follow
      for (const Range &r : rs) {
        fdtd_geom g = v.g;
        g.x_m = r.lo; g.x_M = r.hi;
        hipStream_t st = r.boundary ? v.edge : v.inner;
        if (v.damp) {   // the damped sweep, then the scatter of the planes this range owns (as fdtd_hip_one_step_slab)
          rc = fdtd_stencil_levels_damped(&g, v.m, v.damp, v.lv[op.cur], v.lv[op.prev], v.lv[op.third], FDTD_STENCIL_ORDER, st);
          if (!rc && nsrc) {
            const int own_lo = r.lo + v.off == 0 ? -1 : r.lo + v.off, own_hi = r.hi + v.off == NX - 1 ? NX : r.hi + v.off;
            rc = fdtd_inject_level(&g, v.m, v.src, v.coords, v.lv[op.third], op.time, p_src_m, p_src_M, src_stride,
                                   coord_stride, v.off, own_lo, own_hi, 0, 0, 0, st);
          }
        } else {
          rc = fdtd_hip_one_step_slab(&g, v.m, v.src, v.coords, v.lv[op.cur], v.lv[op.prev], v.lv[op.third], op.time,
                                      nsrc ? p_src_m : 0, nsrc ? p_src_M : -1, src_stride, coord_stride,
                                      FDTD_VARIANT_AUTO, v.off, NX, st);
        }
        if (rc) return rc;
        if (!r.boundary) {
          FDTD_HIP_CHECK(hipEventRecord(v.ev_interior, v.inner));
          v.interior_recorded = true;
        }
      }
      FDTD_HIP_CHECK(hipEventRecord(v.ev_boundary, v.edge));
      v.boundary_recorded = true;
    } else if (split && edges_first) {
      for (size_t i = 1; i < rs.size(); ++i)
        if ((rc = two_steps(v, rs[i], op))) return rc;
      FDTD_HIP_CHECK(hipEventRecord(v.ev_boundary, v.edge));
      v.boundary_recorded = true;
      FDTD_HIP_CHECK(hipStreamWaitEvent(v.inner, v.ev_boundary, 0));
      if ((rc = two_steps(v, rs[0], op))) return rc;
      FDTD_HIP_CHECK(hipEventRecord(v.ev_interior, v.inner));
      v.interior_recorded = true;
    } else {
      for (const Range &r : rs) {
        if ((rc = two_steps(v, r, op))) return rc;
        if (!r.boundary) {
          FDTD_HIP_CHECK(hipEventRecord(v.ev_interior, v.inner));
          v.interior_recorded = true;
        }
      }
      FDTD_HIP_CHECK(hipEventRecord(v.ev_boundary, v.edge));
      v.boundary_recorded = true;
    }
    return 0;
  }

  // pushes the outermost 4 owned planes of the level the operation produced last into the neighbours' ghost planes
  int push(Dev &v, const Op &op) {
    const int lvl = op.pair ? op.spare : op.third;
    if (exchange && (v.left >= 0 || v.right >= 0)) {
      FDTD_HIP_CHECK(hipSetDevice(v.id));
      FDTD_HIP_CHECK(hipStreamWaitEvent(v.copy, v.ev_boundary, 0));
      const size_t bytes = PAIR_GHOST * plane * sizeof(float);
      if (v.left >= 0) {
        Dev &nb = d[(size_t)v.left];
        if (!wait_seq(nb.ready_seq, op.k, v)) return (int)hipErrorUnknown;
        FDTD_HIP_CHECK(hipStreamWaitEvent(v.copy, nb.ev_ready, 0));
        FDTD_HIP_CHECK(hipMemcpyPeerAsync(nb.lv[lvl] + (size_t)(FDTD_HALO + nb.nx) * plane, nb.id,
                                          v.lv[lvl] + (size_t)FDTD_HALO * plane, v.id, bytes, v.copy));
        FDTD_HIP_CHECK(hipEventRecord(v.ev_push_left, v.copy));
      }
      if (v.right >= 0) {
        Dev &nb = d[(size_t)v.right];
        if (!wait_seq(nb.ready_seq, op.k, v)) return (int)hipErrorUnknown;
        FDTD_HIP_CHECK(hipStreamWaitEvent(v.copy, nb.ev_ready, 0));
        FDTD_HIP_CHECK(hipMemcpyPeerAsync(nb.lv[lvl], nb.id, v.lv[lvl] + (size_t)(FDTD_HALO + v.nx - PAIR_GHOST) * plane,
                                          v.id, bytes, v.copy));
        FDTD_HIP_CHECK(hipEventRecord(v.ev_push_right, v.copy));
      }
    }
    v.push_seq.store(op.k, std::memory_order_release);
    return 0;
  }

  int sync_slab(Dev &v) {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    FDTD_HIP_CHECK(hipStreamSynchronize(v.inner));
    FDTD_HIP_CHECK(hipStreamSynchronize(v.edge));
    FDTD_HIP_CHECK(hipStreamSynchronize(v.copy));
    return 0;
  }

  // one slab's share of a list of operations, issued by its own thread
  int run_ops(Dev &v, const std::vector<Op> &ops) {
    int rc = 0;
    for (const Op &op : ops) {
      const double t0 = now_ns(), w0 = v.wait_ns;
      rc = compute(v, op);
      if (!rc) rc = push(v, op);
      v.issue_ns += (now_ns() - t0) - (v.wait_ns - w0);
      if (rc) break;
    }
    if (rc) abort_flag.store(true);
    const int rc2 = sync_slab(v);
    return rc ? rc : rc2;
  }

  // runs fn once per slab: on the issuing threads, or slab after slab on the calling thread
  int for_each_slab(const std::function<int(Dev &)> &fn, bool parallel) {
    if (!parallel || workers.empty()) {
      int first = 0;
      for (Dev &v : d) {
        const int rc = fn(v);
        if (rc && !first) first = rc;
      }
      return first;
    }
    std::unique_lock<std::mutex> lk(mu);
    job = fn;
    done = 0;
    ++job_gen;
    cv_job.notify_all();
    cv_done.wait(lk, [&] { return done == (int)d.size(); });
    for (int rc : job_rc)
      if (rc) return rc;
    return 0;
  }

  void worker_main(size_t i) {
    long seen = 0;
    for (;;) {
      std::function<int(Dev &)> fn;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return quit || job_gen > seen; });
        if (quit) return;
        seen = job_gen;
        fn = job;
      }
      const int rc = fn(d[i]);
      std::lock_guard<std::mutex> lk(mu);
      job_rc[i] = rc;
      if (++done == (int)d.size()) cv_done.notify_all();
    }
  }

  bool devices_distinct() const {
    for (size_t i = 0; i < d.size(); ++i)
      for (size_t j = i + 1; j < d.size(); ++j)
        if (d[i].id == d[j].id) return false;
    return true;
  }

  // Decides the schedule for the sources and fields at hand and allocates what it needs.
  int prepare() {
    if (prepared) return 0;
    follow = !sources_inside || !shells_identical;
    // Can interior and boundary zones run side by side?  Only as the fused sweep, which stores nothing
    // outside its range; with more than 8 sources the sweep needs the planes beside its range in memory.
    split = want_split && nsrc <= 8;
    int min_nx = d[0].nx;
    for (const Dev &v : d) {
      min_nx = v.nx < min_nx ? v.nx : min_nx;
      fdtd_geom gi = v.g;
      gi.x_m = PAIR_GHOST; gi.x_M = v.nx - PAIR_GHOST - 1;
      split = split && v.nx >= 2 * PAIR_GHOST + 1 && fdtd_hip_two_steps_fuses(&gi, nsrc, FDTD_VARIANT_MARCH2);
    }
    const int ey = G.y_M - G.y_m + 1, ez = G.z_M - G.z_m + 1;
    const long tiles = (long)((ey + 7) / 8) * ((ez + 511) / 512);
    const char *zf = getenv("FDTD_SLAB_ZONE_FUSED");
    edges_first = split && min_nx >= 3 * PAIR_GHOST && (zf ? atoi(zf) != 0 : tiles >= 256);
    if (split && !edges_first)
      for (Dev &v : d) {
        if (v.scratch) continue;
        FDTD_HIP_CHECK(hipSetDevice(v.id));
        v.scratch_floats = (long)(PAIR_GHOST + 4) * (long)plane;
        FDTD_HIP_CHECK(hipMalloc((void **)&v.scratch, (size_t)v.scratch_floats * sizeof(float)));
      }
    prepared = true;
    return 0;
  }

  // Puts the three newest levels into the ring slots the reference's ring leaves them in (slot = time % 3)
  // and makes lv[3] the spare buffer again.  A level moves with its box and its shell; on the side of a
  // neighbour the shell is taken four planes deep, so that the ghost planes move with it.
  int settle() {
    const int tgt_of[3] = {mod3(t_now), mod3(t_now + 2), mod3(t_now + 1)};   // cur, prev, third
    if (cur == tgt_of[0] && prev == tgt_of[1] && third == tgt_of[2]) return 0;
    for (Dev &v : d) {
      FDTD_HIP_CHECK(hipSetDevice(v.id));
      fdtd_geom g = v.g;
      if (v.left >= 0) g.x_m = -2;
      if (v.right >= 0) g.x_M = v.nx + 1;
      const int pos[3] = {cur, prev, third};
      const int rc = fdtd_ring_settle(&g, v.lv, pos, spare, tgt_of, v.edge);
      if (rc) return rc;
      FDTD_HIP_CHECK(hipStreamSynchronize(v.edge));
    }
    cur = tgt_of[0]; prev = tgt_of[1]; third = tgt_of[2]; spare = 3;
    return 0;
  }

  void release() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv_job.notify_all();
    for (std::thread &t : workers) t.join();
    workers.clear();
    for (Dev &v : d) {
      if (hipSetDevice(v.id) != hipSuccess) continue;
      for (hipStream_t s : {v.inner, v.edge, v.copy})
        if (s) (void)hipStreamSynchronize(s);
      for (float *p : v.lv)
        if (p) (void)hipFree(p);
      for (void *p : {(void *)v.m, (void *)v.src, (void *)v.coords, (void *)v.scratch, (void *)v.reduce, (void *)v.damp})
        if (p) (void)hipFree(p);
      for (hipStream_t s : {v.inner, v.edge, v.copy})
        if (s) (void)hipStreamDestroy(s);
      for (hipEvent_t e : {v.ev_interior, v.ev_boundary, v.ev_ready, v.ev_shell, v.ev_push_left, v.ev_push_right})
        if (e) (void)hipEventDestroy(e);
    }
    (void)hipGetLastError();
  }
};

namespace {

// the calling thread's current device is put back on every way out of an entry point
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() { (void)hipGetDevice(&dev); }
  ~DeviceGuard() {
    if (dev >= 0) (void)hipSetDevice(dev);
  }
};

int set_sources_impl(fdtd_multigpu *h, const float *src, int src_steps, int src_stride, const float *coords,
                     int coord_rows, int coord_stride, int p_src_m, int p_src_M) {
  const bool has_src = p_src_M >= p_src_m && src && coords && (long)src_steps * src_stride > 0;
  h->nsrc = has_src ? p_src_M - p_src_m + 1 : 0;
  h->p_src_m = p_src_m; h->p_src_M = p_src_M;
  h->src_stride = has_src ? src_stride : 1;
  h->coord_stride = has_src ? coord_stride : 3;
  h->src_steps = has_src ? src_steps : 0;
  h->sources_inside = true;
  h->prepared = false;
  if (!has_src) return 0;
  if (p_src_m < 0 || p_src_M >= src_stride || p_src_M >= coord_rows || coord_stride < 3) return (int)hipErrorInvalidValue;
  h->sources_inside = fdtd_host_sources_inside(coords, coord_stride, p_src_m, p_src_M, h->G);
  const size_t n_src = (size_t)src_steps * src_stride, n_coords = (size_t)coord_rows * coord_stride;
  for (Dev &v : h->d) {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    if (v.src_floats < n_src) {
      if (v.src) FDTD_HIP_CHECK(hipFree(v.src));
      v.src = nullptr;
      FDTD_HIP_CHECK(hipMalloc((void **)&v.src, n_src * sizeof(float)));
      v.src_floats = n_src;
    }
    if (v.coords_floats < n_coords) {
      if (v.coords) FDTD_HIP_CHECK(hipFree(v.coords));
      v.coords = nullptr;
      FDTD_HIP_CHECK(hipMalloc((void **)&v.coords, n_coords * sizeof(float)));
      v.coords_floats = n_coords;
    }
    FDTD_HIP_CHECK(hipMemcpy(v.src, src, n_src * sizeof(float), hipMemcpyHostToDevice));
    FDTD_HIP_CHECK(hipMemcpy(v.coords, coords, n_coords * sizeof(float), hipMemcpyHostToDevice));
  }
  return 0;
}

}  // namespace

extern "C" int fdtd_hip_multigpu_create(const struct fdtd_geom *geom, int ndev, const int *devices, int flags,
                                        fdtd_multigpu **out) {
  if (!geom || !devices || !out || ndev < 1) return (int)hipErrorInvalidValue;
  *out = nullptr;
  const fdtd_geom &G = *geom;
  const int NX = G.nxp - 2 * FDTD_HALO;
  if (NX < 1 || G.nyp < 2 * FDTD_HALO + 1 || G.nzp < 2 * FDTD_HALO + 1) return (int)hipErrorInvalidValue;
  // the decomposition covers the whole x extent of the arrays; slabs of at least 8 planes
  if (G.x_m != 0 || G.x_M != NX - 1 || G.y_M < G.y_m || G.z_M < G.z_m) return (int)hipErrorInvalidValue;
  if (G.y_m < 0 || G.y_M > G.nyp - 2 * FDTD_HALO - 1 || G.z_m < 0 || G.z_M > G.nzp - 2 * FDTD_HALO - 1)
    return (int)hipErrorInvalidValue;
  if (ndev > 1 && NX / 8 < ndev) return (int)hipErrorInvalidValue;
  int count = 0;
  FDTD_HIP_CHECK(hipGetDeviceCount(&count));
  for (int i = 0; i < ndev; ++i)
    if (devices[i] < 0 || devices[i] >= count) return (int)hipErrorInvalidDevice;

  DeviceGuard guard;
  fdtd_multigpu *h = new fdtd_multigpu();
  h->G = G; h->NX = NX; h->plane = (size_t)G.nyp * G.nzp;
  h->flags = flags;
  h->exchange = !(flags & FDTD_MULTIGPU_NO_EXCHANGE);
  h->want_split = !(flags & FDTD_MULTIGPU_NO_OVERLAP) &&
                  (!getenv("FDTD_SLAB_OVERLAP") || atoi(getenv("FDTD_SLAB_OVERLAP")) != 0);
  const char *te = getenv("FDTD_MULTIGPU_THREADS");
  h->threaded = ndev > 1 && !(flags & FDTD_MULTIGPU_ONE_THREAD) && !(te && atoi(te) == 0);
  h->d = std::vector<Dev>((size_t)ndev);
  const int base = NX / ndev, rem = NX % ndev;
  int off = 0;
  for (int r = 0; r < ndev; ++r) {
    Dev &v = h->d[(size_t)r];
    v.id = devices[r];
    v.nx = base + (r < rem ? 1 : 0);
    v.off = off;
    off += v.nx;
    v.left = r > 0 ? r - 1 : -1;
    v.right = r < ndev - 1 ? r + 1 : -1;
    v.g = G;
    v.g.nxp = v.nx + 2 * FDTD_HALO;
    v.g.x_m = 0; v.g.x_M = v.nx - 1;
  }
  auto fail = [&](int rc) {
    h->release();
    delete h;
    return rc;
  };
  for (Dev &v : h->d) {
    hipError_t e = hipSetDevice(v.id);
    if (e != hipSuccess) return fail((int)e);
    for (int nb : {v.left, v.right}) {
      if (nb < 0 || h->d[(size_t)nb].id == v.id) continue;
      int can = 0;
      if ((e = hipDeviceCanAccessPeer(&can, v.id, h->d[(size_t)nb].id)) != hipSuccess) return fail((int)e);
      if (can) {
        e = hipDeviceEnablePeerAccess(h->d[(size_t)nb].id, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail((int)e);
        (void)hipGetLastError();
      }   // without peer access hipMemcpyPeerAsync stages through the host: slower, same bytes
    }
    const size_t lvol = (size_t)(v.nx + 2 * FDTD_HALO) * h->plane;
    for (int l = 0; l < 4; ++l)
      if ((e = hipMalloc((void **)&v.lv[l], lvol * sizeof(float))) != hipSuccess) return fail((int)e);
    if ((e = hipMalloc((void **)&v.m, lvol * sizeof(float))) != hipSuccess) return fail((int)e);
    if ((e = hipMalloc((void **)&v.reduce, 2 * sizeof(unsigned))) != hipSuccess) return fail((int)e);
    for (hipStream_t *s : {&v.inner, &v.edge, &v.copy})
      if ((e = hipStreamCreateWithFlags(s, hipStreamNonBlocking)) != hipSuccess) return fail((int)e);
    for (hipEvent_t *ev : {&v.ev_interior, &v.ev_boundary, &v.ev_ready, &v.ev_shell, &v.ev_push_left, &v.ev_push_right})
      if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return fail((int)e);
  }
  if (h->threaded) {
    h->job_rc.assign((size_t)ndev, 0);
    for (size_t i = 0; i < (size_t)ndev; ++i) h->workers.emplace_back([h, i] { h->worker_main(i); });
  }
  *out = h;
  return 0;
}

extern "C" int fdtd_hip_multigpu_destroy(fdtd_multigpu *h) {
  if (!h) return 0;
  DeviceGuard guard;
  h->release();
  delete h;
  return 0;
}

extern "C" int fdtd_hip_multigpu_set_sources(fdtd_multigpu *h, const float *src, int src_steps, int src_stride,
                                             const float *src_coords, int coord_rows, int coord_stride, int p_src_m,
                                             int p_src_M) {
  if (!h) return (int)hipErrorInvalidValue;
  DeviceGuard guard;
  return set_sources_impl(h, src, src_steps, src_stride, src_coords, coord_rows, coord_stride, p_src_m, p_src_M);
}

// The damping field of the forward operator (fdtd_forward.hip; host array of the padded shape of m), cut into
// the slabs: every later advance runs the damped update (two damped steps per sweep where the sweep is fused).
// NULL takes it away again.  No reference arithmetic exists for the damped update: parity unpinned.
extern "C" int fdtd_hip_multigpu_set_damping(fdtd_multigpu *h, const float *damp) {
  if (!h) return (int)hipErrorInvalidValue;
  DeviceGuard guard;
  const size_t plane = h->plane;
  auto up = [&](Dev &v) -> int {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    const size_t lvol = (size_t)(v.nx + 2 * FDTD_HALO) * plane;
    if (!damp) {
      if (v.damp) FDTD_HIP_CHECK(hipFree(v.damp));
      v.damp = nullptr;
      return 0;
    }
    if (!v.damp) FDTD_HIP_CHECK(hipMalloc((void **)&v.damp, lvol * sizeof(float)));
    return staged_copier_for_current_device().to_device(v.damp, damp + (size_t)v.off * plane, lvol * sizeof(float));
  };
  return h->for_each_slab(up, h->devices_distinct());
}

// The benchmark inputs of main.cpp:345-356 made on the devices: u = 0, m = 1.5 over the padded volume,
// nsrc Ricker sources (main.cpp:290-298) at the driver's positions (main.cpp:301-325), time 0.
extern "C" int fdtd_hip_multigpu_fill_standard_inputs(fdtd_multigpu *h, int nt, int nsrc) {
  if (!h || nt < 1 || nsrc < 0) return (int)hipErrorInvalidValue;
  DeviceGuard guard;
  const fdtd_geom &G = h->G;
  int rc = 0;
  if (nsrc > 0) {
    std::vector<float> src((size_t)nt * nsrc), coords((size_t)nsrc * 3);
    fdtd_fill_ricker(src.data(), nt, nsrc, G.dt);
    fdtd_fill_source_coords(coords.data(), nsrc, h->NX, G.nyp - 2 * FDTD_HALO, G.nzp - 2 * FDTD_HALO);
    rc = set_sources_impl(h, src.data(), nt, nsrc, coords.data(), nsrc, 3, 0, nsrc - 1);
  } else {
    rc = set_sources_impl(h, nullptr, 0, 1, nullptr, 0, 3, 0, -1);
  }
  if (rc) return rc;
  for (Dev &v : h->d) {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    const size_t lvol = (size_t)(v.nx + 2 * FDTD_HALO) * h->plane;
    for (int l = 0; l < 4; ++l) FDTD_HIP_CHECK(hipMemsetAsync(v.lv[l], 0, lvol * sizeof(float), v.edge));
    hipLaunchKernelGGL(fill_value_kernel, dim3(2048), dim3(256), 0, v.edge, v.m, lvol, 1.5f);
    FDTD_HIP_CHECK(hipGetLastError());
  }
  for (Dev &v : h->d) {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    FDTD_HIP_CHECK(hipStreamSynchronize(v.edge));
    v.interior_recorded = v.boundary_recorded = false;
  }
  h->shells_identical = true;
  h->prepared = false;
  h->cur = 0; h->prev = 2; h->third = 1; h->spare = 3; h->t_now = 0;
  h->fields_set = true;
  return 0;
}

// Host arrays in: u = [3][nxp][nyp][nzp] holding the ring of the reference at time index time_m (slot
// time_m % 3 is the current level, openacc.cpp:148), m = [nxp][nyp][nzp].  m may be NULL: the velocity
// model of an earlier upload or fill stays.
extern "C" int fdtd_hip_multigpu_upload(fdtd_multigpu *h, const float *u, const float *m, int time_m) {
  if (!h || !u) return (int)hipErrorInvalidValue;
  DeviceGuard guard;
  const size_t plane = h->plane, vol = (size_t)h->G.nxp * plane;
  h->shells_identical = fdtd_host_shells_identical(u, vol, h->G);
  auto up = [&](Dev &v) -> int {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    const size_t lvol = (size_t)(v.nx + 2 * FDTD_HALO) * plane;
    StagedCopier &copier = staged_copier_for_current_device();
    for (int l = 0; l < 3; ++l) {
      const int rc = copier.to_device(v.lv[l], u + (size_t)l * vol + (size_t)v.off * plane, lvol * sizeof(float));
      if (rc) return rc;
    }
    if (m) {
      const int rc = copier.to_device(v.m, m + (size_t)v.off * plane, lvol * sizeof(float));
      if (rc) return rc;
    }
    // the fourth buffer starts as a copy of a ring slot: same padding beyond the shell as the caller's arrays
    FDTD_HIP_CHECK(hipMemcpy(v.lv[3], v.lv[0], lvol * sizeof(float), hipMemcpyDeviceToDevice));
    v.interior_recorded = v.boundary_recorded = false;
    return 0;
  };
  const int rc = h->for_each_slab(up, h->devices_distinct());
  if (rc) return rc;
  h->prepared = false;
  h->cur = mod3(time_m); h->prev = mod3(time_m + 2); h->third = mod3(time_m + 1); h->spare = 3;
  h->t_now = time_m;
  h->fields_set = true;
  return 0;
}

// Steps time_m .. time_M (inclusive, like the operators' arguments); time_m must be the time index the
// session stands at.  Pairs of steps, and a single step at the end of an odd count.  Returns after every
// device has finished; timers->section0 receives the wall time of the call (host clock from the first
// launch to the last device's completion), section1 nothing: the scatter is part of the sweep.
extern "C" int fdtd_hip_multigpu_advance(fdtd_multigpu *h, int time_m, int time_M, struct profiler *timers) {
  if (!h || !h->fields_set || h->poisoned) return (int)hipErrorInvalidValue;
  if (time_M < time_m) return 0;
  if (time_m != h->t_now) return (int)hipErrorInvalidValue;
  if (h->nsrc && (time_m < 0 || time_M >= h->src_steps)) return (int)hipErrorInvalidValue;
  DeviceGuard guard;
  int rc = h->prepare();
  if (rc) return rc;
  std::vector<Op> ops;
  int cur = h->cur, prev = h->prev, third = h->third, spare = h->spare;
  long pairs = 0, singles = 0;
  for (int t = time_m; t <= time_M;) {
    const bool pair = t + 1 <= time_M;
    ops.push_back(Op{pair, t, cur, prev, third, spare, ++h->seq});
    if (pair) {
      const int ncur = spare, nprev = third, nthird = cur, nspare = prev;
      cur = ncur; prev = nprev; third = nthird; spare = nspare;
      t += 2;
      ++pairs;
    } else {
      const int ncur = third, nprev = cur, nthird = prev;
      cur = ncur; prev = nprev; third = nthird;
      t += 1;
      ++singles;
    }
  }
  const double t0 = now_ns();
  if (h->threaded) {
    rc = h->for_each_slab([h, &ops](Dev &v) { return h->run_ops(v, ops); }, true);
  } else {
    for (const Op &op : ops) {
      const double i0 = now_ns();
      for (Dev &v : h->d)
        if ((rc = h->compute(v, op))) break;
      if (!rc)
        for (Dev &v : h->d)
          if ((rc = h->push(v, op))) break;
      h->d[0].issue_ns += now_ns() - i0;   // one thread issues for every slab
      if (rc) break;
    }
    for (Dev &v : h->d) {
      const int rc2 = h->sync_slab(v);
      if (!rc) rc = rc2;
    }
  }
  const double wall = now_ns() - t0;
  if (rc) {
    h->poisoned = true;
    return rc;
  }
  h->cur = cur; h->prev = prev; h->third = third; h->spare = spare;
  h->t_now = time_M + 1;
  h->pairs_timed += pairs; h->singles_timed += singles;
  h->advance_wall_ns += wall;
  if (timers) timers->section0 += wall * 1e-9;
  return 0;
}

// One level of the wavefield into a host array of nxp*nyp*nzp floats: ring_slot 0 .. 2 is the slot of the
// reference's ring (slot t % 3 holds u(t) for the time index t the session stands at, slot (t+2) % 3 the
// level before it, slot (t+1) % 3 the one before that).  Every slab writes its owned planes, the first
// and the last slab the padding planes beyond the domain as well.
extern "C" int fdtd_hip_multigpu_gather(fdtd_multigpu *h, int ring_slot, float *host_level) {
  if (!h || !host_level || ring_slot < 0 || ring_slot > 2 || !h->fields_set || h->poisoned)
    return (int)hipErrorInvalidValue;
  DeviceGuard guard;
  int rc = h->settle();
  if (rc) return rc;
  const size_t plane = h->plane;
  auto down = [&](Dev &v) -> int {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    const int lo = v.left < 0 ? 0 : FDTD_HALO, hi = v.right < 0 ? v.nx + 2 * FDTD_HALO : v.nx + FDTD_HALO;
    StagedCopier &copier = staged_copier_for_current_device();
    return copier.to_host(host_level + (size_t)(v.off + lo) * plane, v.lv[ring_slot] + (size_t)lo * plane,
                          (size_t)(hi - lo) * plane * sizeof(float));
  };
  return h->for_each_slab(down, h->devices_distinct());
}

// max |u| over the newest level (owned planes of every slab) and the number of NaN / Inf values in it,
// reduced on the devices: a sanity value that needs no host copy of the field.
extern "C" int fdtd_hip_multigpu_max_abs(fdtd_multigpu *h, float *max_abs, long *nonfinite) {
  if (!h || !h->fields_set || h->poisoned) return (int)hipErrorInvalidValue;
  DeviceGuard guard;
  unsigned best = 0;
  long bad = 0;
  for (Dev &v : h->d) {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    FDTD_HIP_CHECK(hipMemsetAsync(v.reduce, 0, 2 * sizeof(unsigned), v.edge));
    hipLaunchKernelGGL(max_abs_kernel, dim3(2048), dim3(256), 0, v.edge, v.lv[h->cur] + (size_t)FDTD_HALO * h->plane,
                       (size_t)v.nx * h->plane, v.reduce);
    FDTD_HIP_CHECK(hipGetLastError());
    unsigned out[2] = {0, 0};
    FDTD_HIP_CHECK(hipMemcpyAsync(out, v.reduce, sizeof(out), hipMemcpyDeviceToHost, v.edge));
    FDTD_HIP_CHECK(hipStreamSynchronize(v.edge));
    best = out[0] > best ? out[0] : best;
    bad += out[1];
  }
  if (max_abs) memcpy(max_abs, &best, sizeof(float));
  if (nonfinite) *nonfinite = bad;
  return 0;
}

// Sum of all 32-bit words of one ring slot's level (owned planes of every slab, padding in y and z included) as a
// 64-bit integer: the same for every decomposition of the same run, so a run on 8 slabs can be compared with the
// undivided one without either field leaving the devices.
extern "C" int fdtd_hip_multigpu_checksum(fdtd_multigpu *h, int ring_slot, unsigned long long *sum) {
  if (!h || !sum || ring_slot < 0 || ring_slot > 2 || !h->fields_set || h->poisoned) return (int)hipErrorInvalidValue;
  DeviceGuard guard;
  const int rc = h->settle();
  if (rc) return rc;
  unsigned long long total = 0;
  for (Dev &v : h->d) {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    unsigned long long *acc = (unsigned long long *)v.reduce;
    FDTD_HIP_CHECK(hipMemsetAsync(acc, 0, sizeof(unsigned long long), v.edge));
    hipLaunchKernelGGL(word_sum_kernel, dim3(2048), dim3(256), 0, v.edge,
                       (const unsigned *)(v.lv[ring_slot] + (size_t)FDTD_HALO * h->plane), (size_t)v.nx * h->plane, acc);
    FDTD_HIP_CHECK(hipGetLastError());
    unsigned long long out = 0;
    FDTD_HIP_CHECK(hipMemcpyAsync(&out, acc, sizeof(out), hipMemcpyDeviceToHost, v.edge));
    FDTD_HIP_CHECK(hipStreamSynchronize(v.edge));
    total += out;
  }
  *sum = total;
  return 0;
}

// out[0] slabs, [1] pairs and [2] single steps advanced so far, [3] host time the busiest issuing thread
// spent in HIP calls per operation (us), [4] host time it spent waiting for a neighbour thread per
// operation (us), [5] wall time per operation (us), [6] 1: interior and boundary zones on two streams,
// [7] 1: outer thirds first (large cross-sections), [8] 1: one issuing thread per slab, [9] 1: shells
// follow the ring.  reset != 0 clears the counters behind the read.
extern "C" int fdtd_hip_multigpu_stats(fdtd_multigpu *h, double out[10], int reset) {
  if (!h || !out) return (int)hipErrorInvalidValue;
  const double ops = (double)(h->pairs_timed + h->singles_timed);
  double issue = 0.0, wait = 0.0;
  for (const Dev &v : h->d) {
    issue = v.issue_ns > issue ? v.issue_ns : issue;
    wait = v.wait_ns > wait ? v.wait_ns : wait;
  }
  out[0] = (double)h->d.size();
  out[1] = (double)h->pairs_timed;
  out[2] = (double)h->singles_timed;
  out[3] = ops > 0 ? issue / ops * 1e-3 : 0.0;
  out[4] = ops > 0 ? wait / ops * 1e-3 : 0.0;
  out[5] = ops > 0 ? h->advance_wall_ns / ops * 1e-3 : 0.0;
  out[6] = h->split; out[7] = h->edges_first; out[8] = h->threaded; out[9] = h->follow;
  if (reset) {
    h->pairs_timed = h->singles_timed = 0;
    h->advance_wall_ns = 0.0;
    for (Dev &v : h->d) v.issue_ns = v.wait_ns = 0.0;
  }
  return 0;
}

extern "C" int Kernel_HIP_MultiGPU(FDTD_OPERATOR_ARGS) {
  int rc = fdtd_check_operator_args(m_vec, src_vec, src_coords_vec, u_vec, p_src_M, p_src_m);
  if (rc) return rc;
  std::vector<int> devices;
  rc = pick_devices(deviceid, &devices);
  if (rc) return rc;

  fdtd_geom G;
  G.nxp = u_vec->size[1]; G.nyp = u_vec->size[2]; G.nzp = u_vec->size[3];
  G.x_m = x_m; G.x_M = x_M; G.y_m = y_m; G.y_M = y_M; G.z_m = z_m; G.z_M = z_M;
  G.dt = dt; G.h_x = h_x; G.h_y = h_y; G.h_z = h_z; G.o_x = o_x; G.o_y = o_y; G.o_z = o_z;
  const int NX = G.nxp - 2 * FDTD_HALO;
  const size_t vol = (size_t)G.nxp * G.nyp * G.nzp;
  const bool has_src = p_src_M >= p_src_m && (long)src_vec->size[0] * src_vec->size[1] > 0;   // openacc.cpp:173
  if (has_src && (time_M >= src_vec->size[0] || time_m < 0)) return (int)hipErrorInvalidValue;

  // Slabs of at least 8 planes; the decomposition covers the whole x extent of the arrays.  Anything
  // else (a sub-range in x, a handful of planes, no time step at all) is one device's work.
  int nslab = (int)devices.size();
  if (NX / 8 < nslab) nslab = NX / 8;
  // (the slab kernels are written for space order 4; other orders run on one device)
  const bool whole_x = x_m == 0 && x_M == NX - 1 && y_M >= y_m && z_M >= z_m && time_M >= time_m &&
                       fdtd_hip_get_space_order() == FDTD_STENCIL_ORDER;
  g_multigpu_last_slabs = (nslab >= 2 && whole_x) ? nslab : 1;
  if (nslab < 2 || !whole_x)
    return Kernel_CUDA_Optimized(m_vec, src_vec, src_coords_vec, u_vec, x_M, x_m, y_M, y_m, z_M, z_m, dt, h_x, h_y, h_z,
                                 o_x, o_y, o_z, p_src_M, p_src_m, time_M, time_m, devices[0], devicerm, timers);

  fdtd_multigpu *h = nullptr;
  rc = fdtd_hip_multigpu_create(&G, nslab, devices.data(), 0, &h);
  if (rc) return rc;
  struct Closer {
    fdtd_multigpu *h;
    ~Closer() { fdtd_hip_multigpu_destroy(h); }
  } closer{h};
  if (has_src)
    rc = fdtd_hip_multigpu_set_sources(h, (const float *)src_vec->data, src_vec->size[0], src_vec->size[1],
                                       (const float *)src_coords_vec->data, src_coords_vec->size[0],
                                       src_coords_vec->size[1], p_src_m, p_src_M);
  if (rc) return rc;
  rc = fdtd_hip_multigpu_upload(h, (const float *)u_vec->data, (const float *)m_vec->data, time_m);
  if (rc) return rc;
  // the untimed steps end where the timed window starts (openacc.cpp:90-93,148): no pair straddles it
  const int first_timed = time_m + FDTD_WARMUP_STEPS;
  const int warm_M = time_M < first_timed - 1 ? time_M : first_timed - 1;
  rc = fdtd_hip_multigpu_advance(h, time_m, warm_M, nullptr);
  if (rc) return rc;
  if (time_M >= first_timed) rc = fdtd_hip_multigpu_advance(h, first_timed, time_M, timers);
  if (rc) return rc;
  // u(T+1), u(T), u(T-1) go to the slots the reference's ring leaves them in
  float *hu_out = (float *)u_vec->data;
  for (int l = 0; l < 3 && !rc; ++l) rc = fdtd_hip_multigpu_gather(h, l, hu_out + (size_t)l * vol);
  return rc;
}

// Number of slabs the last Kernel_HIP_MultiGPU call ran on (1 = it was handed to a single device).
extern "C" int fdtd_hip_multigpu_last_slab_count(void) { return g_multigpu_last_slabs; }
