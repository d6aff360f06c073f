// fdtd_multigpu.hip — Kernel_HIP_MultiGPU: the operator of main.cpp:53-72 on several GPUs of one node,
// driven by one host thread.  The reference is single-device (no counterpart); the arithmetic per cell
// is that of openacc.cpp:159-164,185-196, and the union of the slabs is bit for bit what one device
// computes.
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
// computes them.  All events belong to this process, so every wait names an event that was recorded
// earlier in host order.  Large cross-sections (from 256 tiles of the fused sweep on) update the outer
// thirds of the slab first and the middle third behind them instead of 4-plane zones in scratch.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "fdtd_device.h"
#include "fdtd_host.h"

namespace {

constexpr int PAIR_GHOST = 4;   // planes pushed per side and pair

struct Dev {
  int id = 0;                 // HIP device
  int nx = 0, off = 0;        // owned planes and the first of them (interior index)
  int left = -1, right = -1;  // neighbour slabs (indices into the slab list), -1 = physical boundary
  float *lv[4] = {nullptr, nullptr, nullptr, nullptr};
  float *m = nullptr, *src = nullptr, *coords = nullptr, *scratch = nullptr;
  long scratch_floats = 0;
  hipStream_t inner = nullptr, edge = nullptr, copy = nullptr;
  hipEvent_t ev_interior = nullptr, ev_boundary = nullptr, ev_ready = nullptr, ev_shell = nullptr;
  hipEvent_t ev_push_left = nullptr, ev_push_right = nullptr;   // this slab's pushes to its neighbours
  bool interior_recorded = false, boundary_recorded = false, pushed = false;
  fdtd_geom g;                // local geometry: x range 0 .. nx-1 of nx+8 planes
};

struct Slabs {
  std::vector<Dev> d;
  ~Slabs() {
    for (Dev &v : d) {
      if (hipSetDevice(v.id) != hipSuccess) continue;
      (void)hipDeviceSynchronize();
      for (float *p : v.lv)
        if (p) (void)hipFree(p);
      if (v.m) (void)hipFree(v.m);
      if (v.src) (void)hipFree(v.src);
      if (v.coords) (void)hipFree(v.coords);
      if (v.scratch) (void)hipFree(v.scratch);
      for (hipStream_t s : {v.inner, v.edge, v.copy})
        if (s) (void)hipStreamDestroy(s);
      for (hipEvent_t e : {v.ev_interior, v.ev_boundary, v.ev_ready, v.ev_shell, v.ev_push_left, v.ev_push_right})
        if (e) (void)hipEventDestroy(e);
    }
  }
};

// Devices of the run: FDTD_MULTIGPU_DEVICES (comma-separated HIP device numbers; a device may be
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

// Do the three ring levels of the host array carry bit-identical shells (the two layers of cells around
// the updated box)?  Then no shell has to follow the ring through the four rotating buffers.
bool host_shells_identical(const float *u, size_t vol, const fdtd_geom &g) {
  const long sx = (long)g.nyp * g.nzp, sy = g.nzp;
  const int ez = g.z_M - g.z_m + 1;
  for (int x = g.x_m - 2; x <= g.x_M + 2; ++x)
    for (int y = g.y_m - 2; y <= g.y_M + 2; ++y) {
      const bool whole = x < g.x_m || x > g.x_M || y < g.y_m || y > g.y_M;
      const long row = (long)(x + FDTD_HALO) * sx + (long)(y + FDTD_HALO) * sy + FDTD_HALO + g.z_m - 2;
      for (int l = 1; l < 3; ++l) {
        const float *a = u + row, *b = u + (size_t)l * vol + row;
        if (whole) {
          if (memcmp(a, b, (size_t)(ez + 4) * sizeof(float))) return false;
        } else if (memcmp(a, b, 2 * sizeof(float)) || memcmp(a + ez + 2, b + ez + 2, 2 * sizeof(float))) {
          return false;
        }
      }
    }
  return true;
}

double now_ms() {
  using ms = std::chrono::duration<double, std::milli>;
  return ms(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Schedule {
  Slabs &S;
  const int global_nx;
  const int nsrc, p_src_m, p_src_M, src_stride, coord_stride;
  const bool follow;          // shells follow the ring (copied from prev into spare before every pair)
  const bool split;           // interior and boundary chains on two streams
  const bool edges_first;     // outer thirds through the fused sweep instead of 4-plane zones in scratch
  const int flags0;           // FDTD_TWO_STEPS_SOURCES_INSIDE or 0
  int cur, prev, third, spare;

  // ranges of a slab: interior first (split), then the two boundary zones
  struct Range { int lo, hi; bool boundary; };
  std::vector<Range> ranges(const Dev &v) const {
    if (!split) return {{0, v.nx - 1, true}};
    int gw = PAIR_GHOST;
    if (edges_first) gw = v.nx / 3 < 96 ? v.nx / 3 : 96;
    if (gw < PAIR_GHOST) gw = PAIR_GHOST;
    return {{gw, v.nx - gw - 1, false}, {0, gw - 1, true}, {v.nx - gw, v.nx - 1, true}};
  }

  int begin(Dev &v) {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    if (split) {
      if (v.interior_recorded) FDTD_HIP_CHECK(hipStreamWaitEvent(v.edge, v.ev_interior, 0));
      if (v.boundary_recorded) FDTD_HIP_CHECK(hipStreamWaitEvent(v.inner, v.ev_boundary, 0));
    }
    // the ghost planes the neighbours pushed behind the previous pair or step
    if (v.left >= 0 && S.d[v.left].pushed) FDTD_HIP_CHECK(hipStreamWaitEvent(v.edge, S.d[v.left].ev_push_right, 0));
    if (v.right >= 0 && S.d[v.right].pushed) FDTD_HIP_CHECK(hipStreamWaitEvent(v.edge, S.d[v.right].ev_push_left, 0));
    return 0;
  }

  int two_steps(Dev &v, const Range &r, int time) {
    hipStream_t st = r.boundary ? v.edge : v.inner;
    fdtd_geom g = v.g;
    g.x_m = r.lo; g.x_M = r.hi;
    // side by side with other ranges a call must store nothing outside its range: the fused sweep does
    // not (forced whatever the length of the range), the two-sweep path needs scratch for that
    const bool fused_call = split && (!r.boundary || edges_first);
    const int variant = fused_call ? FDTD_VARIANT_MARCH2 : FDTD_VARIANT_AUTO;
    const int flags = flags0 | (r.lo == 0 ? FDTD_TWO_STEPS_KEEP_LOW : 0) | (r.hi == v.nx - 1 ? FDTD_TWO_STEPS_KEEP_HIGH : 0);
    const bool use_scratch = split && r.boundary && !edges_first;
    return fdtd_two_steps_impl(&g, v.m, v.src, v.coords, v.lv[cur], v.lv[prev], v.lv[third], v.lv[spare], time,
                               nsrc ? p_src_m : 0, nsrc ? p_src_M : -1, src_stride, coord_stride, variant, flags, v.off,
                               global_nx, st, nullptr, nullptr, use_scratch ? v.scratch : nullptr,
                               use_scratch ? v.scratch_floats : 0);
  }

  // pushes the outermost 4 owned planes of level `lvl` into the neighbours' ghost planes
  int push(Dev &v, int lvl) {
    if (v.left < 0 && v.right < 0) return 0;
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    FDTD_HIP_CHECK(hipStreamWaitEvent(v.copy, v.ev_boundary, 0));
    const size_t plane = (size_t)v.g.nyp * v.g.nzp, bytes = PAIR_GHOST * plane * sizeof(float);
    if (v.left >= 0) {
      Dev &nb = S.d[v.left];
      FDTD_HIP_CHECK(hipStreamWaitEvent(v.copy, nb.ev_ready, 0));
      FDTD_HIP_CHECK(hipMemcpyPeerAsync(nb.lv[lvl] + (size_t)(FDTD_HALO + nb.nx) * plane, nb.id,
                                        v.lv[lvl] + (size_t)FDTD_HALO * plane, v.id, bytes, v.copy));
      FDTD_HIP_CHECK(hipEventRecord(v.ev_push_left, v.copy));
    }
    if (v.right >= 0) {
      Dev &nb = S.d[v.right];
      FDTD_HIP_CHECK(hipStreamWaitEvent(v.copy, nb.ev_ready, 0));
      FDTD_HIP_CHECK(hipMemcpyPeerAsync(nb.lv[lvl], nb.id, v.lv[lvl] + (size_t)(FDTD_HALO + v.nx - PAIR_GHOST) * plane,
                                        v.id, bytes, v.copy));
      FDTD_HIP_CHECK(hipEventRecord(v.ev_push_right, v.copy));
    }
    v.pushed = true;
    return 0;
  }

  int pair(int time) {
    for (Dev &v : S.d) {
      int rc = begin(v);
      if (rc) return rc;
      if (follow) {
        rc = fdtd_hip_copy_shell(&v.g, v.lv[spare], v.lv[prev], 0, v.edge);
        if (rc) return rc;
        if (split) {   // the interior call scatters halo-layer corners of its own planes into this very shell
          FDTD_HIP_CHECK(hipEventRecord(v.ev_shell, v.edge));
          FDTD_HIP_CHECK(hipStreamWaitEvent(v.inner, v.ev_shell, 0));
        }
      }
      FDTD_HIP_CHECK(hipEventRecord(v.ev_ready, v.edge));   // from here on the neighbours may write the ghost planes
      const std::vector<Range> rs = ranges(v);
      if (split && edges_first) {
        for (size_t i = 1; i < rs.size(); ++i)
          if ((rc = two_steps(v, rs[i], time))) return rc;
        FDTD_HIP_CHECK(hipEventRecord(v.ev_boundary, v.edge));
        v.boundary_recorded = true;
        FDTD_HIP_CHECK(hipStreamWaitEvent(v.inner, v.ev_boundary, 0));
        if ((rc = two_steps(v, rs[0], time))) return rc;
        FDTD_HIP_CHECK(hipEventRecord(v.ev_interior, v.inner));
        v.interior_recorded = true;
      } else {
        for (const Range &r : rs) {
          if ((rc = two_steps(v, r, time))) return rc;
          if (!r.boundary) {
            FDTD_HIP_CHECK(hipEventRecord(v.ev_interior, v.inner));
            v.interior_recorded = true;
          }
        }
        FDTD_HIP_CHECK(hipEventRecord(v.ev_boundary, v.edge));
        v.boundary_recorded = true;
      }
    }
    for (Dev &v : S.d) {
      const int rc = push(v, spare);
      if (rc) return rc;
    }
    const int ncur = spare, nprev = third, nthird = cur, nspare = prev;
    cur = ncur; prev = nprev; third = nthird; spare = nspare;
    return 0;
  }

  // one step; the new level is handed over 4 planes deep so that a pair can follow
  int single(int time) {
    for (Dev &v : S.d) {
      int rc = begin(v);
      if (rc) return rc;
      FDTD_HIP_CHECK(hipEventRecord(v.ev_ready, v.edge));
      for (const Range &r : ranges(v)) {
        fdtd_geom g = v.g;
        g.x_m = r.lo; g.x_M = r.hi;
        rc = fdtd_hip_one_step_slab(&g, v.m, v.src, v.coords, v.lv[cur], v.lv[prev], v.lv[third], time,
                                    nsrc ? p_src_m : 0, nsrc ? p_src_M : -1, src_stride, coord_stride, FDTD_VARIANT_AUTO,
                                    v.off, global_nx, r.boundary ? v.edge : v.inner);
        if (rc) return rc;
        if (!r.boundary) {
          FDTD_HIP_CHECK(hipEventRecord(v.ev_interior, v.inner));
          v.interior_recorded = true;
        }
      }
      FDTD_HIP_CHECK(hipEventRecord(v.ev_boundary, v.edge));
      v.boundary_recorded = true;
    }
    for (Dev &v : S.d) {
      const int rc = push(v, third);
      if (rc) return rc;
    }
    const int ncur = third, nprev = cur, nthird = prev;
    cur = ncur; prev = nprev; third = nthird;
    return 0;
  }

  int sync_all() {
    for (Dev &v : S.d) {
      FDTD_HIP_CHECK(hipSetDevice(v.id));
      FDTD_HIP_CHECK(hipDeviceSynchronize());
    }
    return 0;
  }
};

int g_multigpu_last_slabs = 0;

}  // namespace

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
  const size_t plane = (size_t)G.nyp * G.nzp, vol = (size_t)G.nxp * plane;
  const bool has_src = p_src_M >= p_src_m && (long)src_vec->size[0] * src_vec->size[1] > 0;   // openacc.cpp:173
  if (has_src && (time_M >= src_vec->size[0] || time_m < 0)) return (int)hipErrorInvalidValue;
  const int nsrc = has_src ? p_src_M - p_src_m + 1 : 0;

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

  // the calling thread's current device is put back on every way out
  struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { (void)hipGetDevice(&dev); }
    ~DeviceGuard() {
      if (dev >= 0) (void)hipSetDevice(dev);
    }
  } device_guard;
  const float *hu = (const float *)u_vec->data, *hm = (const float *)m_vec->data;
  // shells follow the ring unless the three levels start with identical shells and no source corner can
  // reach the halo layer (positions as openacc.cpp:185-187 computes them)
  bool sources_inside = true;
  if (has_src) {
    const float *hc = (const float *)src_coords_vec->data;
    const int cs = src_coords_vec->size[1];
    for (int p = p_src_m; p <= p_src_M && sources_inside; ++p) {
      const int posx = (int)floorf((-o_x + hc[(size_t)p * cs + 0]) / h_x);
      const int posy = (int)floorf((-o_y + hc[(size_t)p * cs + 1]) / h_y);
      const int posz = (int)floorf((-o_z + hc[(size_t)p * cs + 2]) / h_z);
      sources_inside = !(posx < x_m || posx + 1 > x_M || posy < y_m || posy + 1 > y_M || posz < z_m || posz + 1 > z_M);
    }
  }
  const bool follow = !sources_inside || !host_shells_identical(hu, vol, G);

  Slabs S;
  S.d.resize((size_t)nslab);
  const int base = NX / nslab, rem = NX % nslab;
  int off = 0;
  for (int r = 0; r < nslab; ++r) {
    Dev &v = S.d[(size_t)r];
    v.id = devices[(size_t)r];
    v.nx = base + (r < rem ? 1 : 0);
    v.off = off;
    off += v.nx;
    v.left = r > 0 ? r - 1 : -1;
    v.right = r < nslab - 1 ? r + 1 : -1;
    v.g = G;
    v.g.nxp = v.nx + 2 * FDTD_HALO;
    v.g.x_m = 0; v.g.x_M = v.nx - 1;
  }
  // can interior and boundary zones run side by side?  Only as the fused sweep, which stores nothing
  // outside its range; with more than 8 sources the sweep needs the planes beside its range in memory.
  bool split = nsrc <= 8 && (!getenv("FDTD_SLAB_OVERLAP") || atoi(getenv("FDTD_SLAB_OVERLAP")) != 0);
  int min_nx = S.d[0].nx;
  for (const Dev &v : S.d) {
    min_nx = v.nx < min_nx ? v.nx : min_nx;
    fdtd_geom gi = v.g;
    gi.x_m = PAIR_GHOST; gi.x_M = v.nx - PAIR_GHOST - 1;
    split = split && v.nx >= 2 * PAIR_GHOST + 1 && fdtd_hip_two_steps_fuses(&gi, nsrc, FDTD_VARIANT_MARCH2);
  }
  const int ey = y_M - y_m + 1, ez = z_M - z_m + 1;
  const long tiles = (long)((ey + 7) / 8) * ((ez + 511) / 512);
  const char *zf = getenv("FDTD_SLAB_ZONE_FUSED");
  const bool edges_first = split && min_nx >= 3 * PAIR_GHOST && (zf ? atoi(zf) != 0 : tiles >= 256);

  // ---- allocate and upload ----
  const size_t n_src = has_src ? (size_t)src_vec->size[0] * src_vec->size[1] : 0;
  const size_t n_coords = has_src ? (size_t)src_coords_vec->size[0] * src_coords_vec->size[1] : 0;
  for (Dev &v : S.d) {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    for (int nb : {v.left, v.right}) {
      if (nb < 0 || S.d[(size_t)nb].id == v.id) continue;
      int can = 0;
      FDTD_HIP_CHECK(hipDeviceCanAccessPeer(&can, v.id, S.d[(size_t)nb].id));
      if (can) {
        const hipError_t e = hipDeviceEnablePeerAccess(S.d[(size_t)nb].id, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return (int)e;
        (void)hipGetLastError();
      }   // without peer access hipMemcpyPeerAsync stages through the host: slower, same bytes
    }
    const size_t lvol = (size_t)(v.nx + 2 * FDTD_HALO) * plane;
    for (int l = 0; l < 4; ++l) FDTD_HIP_CHECK(hipMalloc((void **)&v.lv[l], lvol * sizeof(float)));
    FDTD_HIP_CHECK(hipMalloc((void **)&v.m, lvol * sizeof(float)));
    if (has_src) {
      FDTD_HIP_CHECK(hipMalloc((void **)&v.src, n_src * sizeof(float)));
      FDTD_HIP_CHECK(hipMalloc((void **)&v.coords, n_coords * sizeof(float)));
      FDTD_HIP_CHECK(hipMemcpy(v.src, src_vec->data, n_src * sizeof(float), hipMemcpyHostToDevice));
      FDTD_HIP_CHECK(hipMemcpy(v.coords, src_coords_vec->data, n_coords * sizeof(float), hipMemcpyHostToDevice));
    }
    if (split && !edges_first) {
      v.scratch_floats = (long)(PAIR_GHOST + 4) * (long)plane;
      FDTD_HIP_CHECK(hipMalloc((void **)&v.scratch, (size_t)v.scratch_floats * sizeof(float)));
    }
    FDTD_HIP_CHECK(hipStreamCreateWithFlags(&v.inner, hipStreamNonBlocking));
    FDTD_HIP_CHECK(hipStreamCreateWithFlags(&v.edge, hipStreamNonBlocking));
    FDTD_HIP_CHECK(hipStreamCreateWithFlags(&v.copy, hipStreamNonBlocking));
    for (hipEvent_t *e : {&v.ev_interior, &v.ev_boundary, &v.ev_ready, &v.ev_shell, &v.ev_push_left, &v.ev_push_right})
      FDTD_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    StagedCopier &copier = staged_copier_for_current_device();
    for (int l = 0; l < 3; ++l) {
      rc = copier.to_device(v.lv[l], hu + (size_t)l * vol + (size_t)v.off * plane, lvol * sizeof(float));
      if (rc) return rc;
    }
    rc = copier.to_device(v.m, hm + (size_t)v.off * plane, lvol * sizeof(float));
    if (rc) return rc;
    // the fourth buffer starts as a copy of a ring slot: same padding beyond the shell as the caller's arrays
    FDTD_HIP_CHECK(hipMemcpy(v.lv[3], v.lv[0], lvol * sizeof(float), hipMemcpyDeviceToDevice));
  }

  auto mod3 = [](int t) { return ((t % 3) + 3) % 3; };
  Schedule sch{S, NX, nsrc, p_src_m, p_src_M, has_src ? src_vec->size[1] : 1, has_src ? src_coords_vec->size[1] : 3,
               follow, split, edges_first, sources_inside ? FDTD_TWO_STEPS_SOURCES_INSIDE : 0,
               mod3(time_m), mod3(time_m + 2), mod3(time_m + 1), 3};
  if (!follow)   // one shell for everybody
    for (Dev &v : S.d) {
      FDTD_HIP_CHECK(hipSetDevice(v.id));
      rc = fdtd_hip_copy_shell(&v.g, v.lv[3], v.lv[0], 0, v.edge);
      if (rc) return rc;
    }
  rc = sch.sync_all();
  if (rc) return rc;

  // ---- time loop: single steps where a pair would straddle the start of the timed window or the end ----
  const int first_timed = time_m + FDTD_WARMUP_STEPS;
  const bool timed = timers && time_M >= first_timed;
  double t_window = 0.0;
  int time = time_m;
  while (time <= time_M) {
    if (timed && time == first_timed) {
      rc = sch.sync_all();
      if (rc) return rc;
      t_window = now_ms();
    }
    const bool pair = time + 1 <= time_M && !(timed && time < first_timed && time + 1 >= first_timed);
    rc = pair ? sch.pair(time) : sch.single(time);
    if (rc) return rc;
    time += pair ? 2 : 1;
  }
  rc = sch.sync_all();
  if (rc) return rc;
  if (timed) timers->section0 += (now_ms() - t_window) * 1e-3;   // whole window, all devices; the scatter has no time of its own

  // ---- ring order and download: u(T+1), u(T), u(T-1) go to the slots the reference's ring leaves them in ----
  float *hu_out = (float *)u_vec->data;
  for (Dev &v : S.d) {
    FDTD_HIP_CHECK(hipSetDevice(v.id));
    int pos[3] = {sch.cur, sch.prev, sch.third};
    const int tgt[3] = {mod3(time_M + 1), mod3(time_M), mod3(time_M + 2)};
    int free_pos = sch.spare;
    for (int it = 0; it < 8; ++it) {
      int mover = -1;
      for (int r = 0; r < 3; ++r)
        if (pos[r] != tgt[r] && tgt[r] == free_pos) mover = r;
      if (mover < 0)
        for (int r = 0; r < 3; ++r)
          if (pos[r] != tgt[r]) mover = r;
      if (mover < 0) break;
      rc = fdtd_hip_copy_shell(&v.g, v.lv[free_pos], v.lv[pos[mover]], 1, v.edge);
      if (rc) return rc;
      const int old = pos[mover];
      pos[mover] = free_pos;
      free_pos = old;
    }
    FDTD_HIP_CHECK(hipStreamSynchronize(v.edge));
    // owned planes, plus the physical padding on the first and the last slab
    const int lo = v.left < 0 ? 0 : FDTD_HALO, hi = v.right < 0 ? v.nx + 2 * FDTD_HALO : v.nx + FDTD_HALO;
    StagedCopier &copier = staged_copier_for_current_device();
    for (int l = 0; l < 3; ++l) {
      rc = copier.to_host(hu_out + (size_t)l * vol + (size_t)(v.off + lo) * plane, v.lv[l] + (size_t)lo * plane,
                          (size_t)(hi - lo) * plane * sizeof(float));
      if (rc) return rc;
    }
  }
  return 0;
}

// Number of slabs the last Kernel_HIP_MultiGPU call ran on (1 = it was handed to a single device).
extern "C" int fdtd_hip_multigpu_last_slab_count(void) { return g_multigpu_last_slabs; }
