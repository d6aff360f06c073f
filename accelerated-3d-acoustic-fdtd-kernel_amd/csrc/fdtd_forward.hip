// fdtd_forward.hip — the steps either side of the hot path in a Devito-style propagator: an absorbing
// sponge (damping field) in the update and receivers that sample the wavefield (Kernel_HIP_Forward).
//
// The reference has neither: its halo is a fixed frame and nothing reads the field during the run
// (main.cpp:352 `m = 1.5`, openacc.cpp:148-205 has two sections only).  There is therefore no reference
// arithmetic for this file: parity unpinned.  The forms are chosen so that they degenerate to the
// reference's: the damped update is openacc.cpp:159-164 with a first-order damping term,
//     m (u+ - 2u + u-)/dt^2 + damp (u+ - u)/dt = lap(u)
//     u+ = dt^2 ( lap - (-2 r1 u + r1 u-) m + r0 damp u ) / (m + dt damp),        r0 = 1/dt, r1 = 1/dt^2
// (damp = 0: the reference expression), and the receiver interpolation is the transpose of the scatter of
// openacc.cpp:185-196 (same positions, weights, loop order and bounds test; a corner outside the bounds
// contributes nothing).  Receivers sample the level the step started from, u(time), as a generated
// Devito operator does.  The damped update is a variant of the sweeps of fdtd_orders.hip (marching for large
// boxes, one cell per thread for small ones; every space order); one receiver per thread.
#include <math.h>

#include <vector>

#include "fdtd_device.h"
#include "fdtd_host.h"

#pragma clang fp contract(off)

namespace {

struct ReceiverArgs {
  const float *u, *coords;
  float *rec_row;    // rec + time * rec_stride
  long sx, sy;
  int x_m, x_M, y_m, y_M, z_m, z_M;
  float h_x, h_y, h_z, o_x, o_y, o_z;
  int p_rec_m, p_rec_M, coord_stride;
  int halo;          // cells of padding per side of the arrays (the space order)
};

__global__ __launch_bounds__(256) void receiver_kernel(ReceiverArgs a) {
  const int p = a.p_rec_m + blockIdx.x * blockDim.x + threadIdx.x;
  if (p > a.p_rec_M) return;
  const float gx = (-a.o_x + a.coords[(long)p * a.coord_stride + 0]) / a.h_x;
  const float gy = (-a.o_y + a.coords[(long)p * a.coord_stride + 1]) / a.h_y;
  const float gz = (-a.o_z + a.coords[(long)p * a.coord_stride + 2]) / a.h_z;
  const int posx = (int)floorf(gx), posy = (int)floorf(gy), posz = (int)floorf(gz);
  const float px = -floorf(gx) + gx, py = -floorf(gy) + gy, pz = -floorf(gz) + gz;
  float sum = 0.0F;
  for (int rx = 0; rx <= 1; ++rx)
    for (int ry = 0; ry <= 1; ++ry)
      for (int rz = 0; rz <= 1; ++rz)
        if (rx + posx >= a.x_m - 1 && ry + posy >= a.y_m - 1 && rz + posz >= a.z_m - 1 && rx + posx <= a.x_M + 1 &&
            ry + posy <= a.y_M + 1 && rz + posz <= a.z_M + 1)
          sum += (rx * px + (1 - rx) * (1 - px)) * (ry * py + (1 - ry) * (1 - py)) * (rz * pz + (1 - rz) * (1 - pz)) *
                 a.u[(long)(rx + posx + a.halo) * a.sx + (long)(ry + posy + a.halo) * a.sy + (rz + posz + a.halo)];
  a.rec_row[p] = sum;
}

int g_forward_last_fused = 0;   // did the last Kernel_HIP_Forward call take two steps per sweep?

struct Buffers {
  std::vector<void *> p;
  ~Buffers() {
    for (void *q : p)
      if (q) (void)hipFree(q);
  }
  int alloc(void **out, size_t bytes) {
    FDTD_HIP_CHECK(hipMalloc(out, bytes ? bytes : 4));
    p.push_back(*out);
    return 0;
  }
};

}  // namespace

// One damped sweep of level t0 into level t2 (damp has the padded shape of m), arrays padded by `order` cells.
extern "C" int fdtd_hip_stencil_step_damped_order(const struct fdtd_geom *g, const float *m, const float *damp, float *u,
                                                  int t0, int t1, int t2, int order, void *stream) {
  if (!g || !m || !damp || !u || t0 < 0 || t0 > 2 || t1 < 0 || t1 > 2 || t2 < 0 || t2 > 2) return (int)hipErrorInvalidValue;
  const long vol = (long)g->nxp * g->nyp * g->nzp;
  return fdtd_stencil_levels_damped(g, m, damp, u + t0 * vol, u + t1 * vol, u + t2 * vol, order, stream);
}
extern "C" int fdtd_hip_stencil_step_damped(const struct fdtd_geom *g, const float *m, const float *damp, float *u,
                                            int t0, int t1, int t2, void *stream) {
  return fdtd_hip_stencil_step_damped_order(g, m, damp, u, t0, t1, t2, FDTD_STENCIL_ORDER, stream);
}

// rec[time][p] = trilinear interpolation of level t of u at rec_coords[p], p = p_rec_m .. p_rec_M.
extern "C" int fdtd_hip_receiver_sample_order(const struct fdtd_geom *g, const float *u, int t, const float *rec_coords,
                                              float *rec, int time, int p_rec_m, int p_rec_M, int rec_stride,
                                              int coord_stride, int order, void *stream) {
  if (p_rec_M < p_rec_m) return 0;
  if (!g || !u || !rec_coords || !rec || t < 0 || t > 2 || rec_stride < 1 || coord_stride < 3 || time < 0 || order < 4 ||
      order > 12 || order % 2)
    return (int)hipErrorInvalidValue;
  ReceiverArgs a;
  a.u = u + (long)t * g->nxp * g->nyp * g->nzp;
  a.coords = rec_coords;
  a.rec_row = rec + (long)time * rec_stride;
  a.sx = (long)g->nyp * g->nzp; a.sy = g->nzp;
  a.x_m = g->x_m; a.x_M = g->x_M; a.y_m = g->y_m; a.y_M = g->y_M; a.z_m = g->z_m; a.z_M = g->z_M;
  a.h_x = g->h_x; a.h_y = g->h_y; a.h_z = g->h_z; a.o_x = g->o_x; a.o_y = g->o_y; a.o_z = g->o_z;
  a.p_rec_m = p_rec_m; a.p_rec_M = p_rec_M; a.coord_stride = coord_stride;
  a.halo = order;
  const int n = p_rec_M - p_rec_m + 1;
  hipLaunchKernelGGL(receiver_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
extern "C" int fdtd_hip_receiver_sample(const struct fdtd_geom *g, const float *u, int t, const float *rec_coords,
                                        float *rec, int time, int p_rec_m, int p_rec_M, int rec_stride,
                                        int coord_stride, void *stream) {
  return fdtd_hip_receiver_sample_order(g, u, t, rec_coords, rec, time, p_rec_m, p_rec_M, rec_stride, coord_stride,
                                        FDTD_STENCIL_ORDER, stream);
}

// The forward operator of a Devito-style acoustic propagator: damped update, source scatter, receiver
// sampling, per time step.  Host arrays in, u and rec out; argument conventions of the operator ABI
// (max before min, inclusive ranges, timers accumulated over steps time_m+5 .. time_M; section0 = update,
// section1 = scatter and receivers).
extern "C" int Kernel_HIP_Forward(struct dataobj *damp_vec, struct dataobj *m_vec, struct dataobj *rec_vec,
                                  struct dataobj *rec_coords_vec, struct dataobj *src_vec,
                                  struct dataobj *src_coords_vec, struct dataobj *u_vec, const int x_M, const int x_m,
                                  const int y_M, const int y_m, const int z_M, const int z_m, const float dt,
                                  const float h_x, const float h_y, const float h_z, const float o_x, const float o_y,
                                  const float o_z, const int p_rec_M, const int p_rec_m, const int p_src_M,
                                  const int p_src_m, const int time_M, const int time_m, const int deviceid,
                                  const int devicerm, struct profiler *timers) {
  (void)devicerm;
  if (!damp_vec || !rec_vec || !rec_coords_vec) return (int)hipErrorInvalidValue;
  int rc = fdtd_check_operator_args(m_vec, src_vec, src_coords_vec, u_vec, p_src_M, p_src_m);
  if (rc) return rc;
  const int order = fdtd_hip_get_space_order();   // arrays padded by that many cells (main.cpp:32)
  if (!damp_vec->data || !damp_vec->size) return (int)hipErrorInvalidValue;
  for (int d = 0; d < 3; ++d)
    if (damp_vec->size[d] != m_vec->size[d]) return (int)hipErrorInvalidValue;
  const bool has_rec = p_rec_M >= p_rec_m;
  if (has_rec) {
    if (!rec_vec->data || !rec_vec->size || !rec_coords_vec->data || !rec_coords_vec->size) return (int)hipErrorInvalidValue;
    if (p_rec_m < 0 || p_rec_M >= rec_vec->size[1] || p_rec_M >= rec_coords_vec->size[0] || rec_coords_vec->size[1] < 3 ||
        time_m < 0 || time_M >= rec_vec->size[0])
      return (int)hipErrorInvalidValue;
  }
  const bool has_src = p_src_M >= p_src_m && (long)src_vec->size[0] * src_vec->size[1] > 0;
  if (has_src && (time_M >= src_vec->size[0] || time_m < 0)) return (int)hipErrorInvalidValue;
  if (deviceid != -1) FDTD_HIP_CHECK(hipSetDevice(deviceid));

  fdtd_geom g;
  g.nxp = u_vec->size[1]; g.nyp = u_vec->size[2]; g.nzp = u_vec->size[3];
  g.x_m = x_m; g.x_M = x_M; g.y_m = y_m; g.y_M = y_M; g.z_m = z_m; g.z_M = z_M;
  g.dt = dt; g.h_x = h_x; g.h_y = h_y; g.h_z = h_z; g.o_x = o_x; g.o_y = o_y; g.o_z = o_z;
  const size_t vol = (size_t)g.nxp * g.nyp * g.nzp;
  const size_t n_src = has_src ? (size_t)src_vec->size[0] * src_vec->size[1] : 0;
  const size_t n_sc = has_src ? (size_t)src_coords_vec->size[0] * src_coords_vec->size[1] : 0;
  const size_t n_rec = has_rec ? (size_t)rec_vec->size[0] * rec_vec->size[1] : 0;
  const size_t n_rc = has_rec ? (size_t)rec_coords_vec->size[0] * rec_coords_vec->size[1] : 0;

  Buffers buf;
  float *d_u, *d_m, *d_damp, *d_src = nullptr, *d_sc = nullptr, *d_rec = nullptr, *d_rc = nullptr;
  if ((rc = buf.alloc((void **)&d_u, 3 * vol * sizeof(float)))) return rc;
  if ((rc = buf.alloc((void **)&d_m, vol * sizeof(float)))) return rc;
  if ((rc = buf.alloc((void **)&d_damp, vol * sizeof(float)))) return rc;
  StagedCopier &copier = staged_copier_for_current_device();
  if ((rc = copier.to_device(d_u, u_vec->data, 3 * vol * sizeof(float)))) return rc;
  if ((rc = copier.to_device(d_m, m_vec->data, vol * sizeof(float)))) return rc;
  if ((rc = copier.to_device(d_damp, damp_vec->data, vol * sizeof(float)))) return rc;
  if (has_src) {
    if ((rc = buf.alloc((void **)&d_src, n_src * sizeof(float)))) return rc;
    if ((rc = buf.alloc((void **)&d_sc, n_sc * sizeof(float)))) return rc;
    FDTD_HIP_CHECK(hipMemcpy(d_src, src_vec->data, n_src * sizeof(float), hipMemcpyHostToDevice));
    FDTD_HIP_CHECK(hipMemcpy(d_sc, src_coords_vec->data, n_sc * sizeof(float), hipMemcpyHostToDevice));
  }
  if (has_rec) {
    if ((rc = buf.alloc((void **)&d_rec, n_rec * sizeof(float)))) return rc;
    if ((rc = buf.alloc((void **)&d_rc, n_rc * sizeof(float)))) return rc;
    FDTD_HIP_CHECK(hipMemcpy(d_rec, rec_vec->data, n_rec * sizeof(float), hipMemcpyHostToDevice));
    FDTD_HIP_CHECK(hipMemcpy(d_rc, rec_coords_vec->data, n_rc * sizeof(float), hipMemcpyHostToDevice));
  }

  const int first_timed = time_m + FDTD_WARMUP_STEPS;
  const int n_timed = (timers && time_M >= first_timed) ? (time_M - first_timed + 1) : 0;
  std::vector<hipEvent_t> ev((size_t)n_timed * 3, nullptr);
  struct EvGuard {
    std::vector<hipEvent_t> &e;
    ~EvGuard() {
      for (hipEvent_t x : e)
        if (x) (void)hipEventDestroy(x);
    }
  } guard{ev};
  for (hipEvent_t &e : ev) FDTD_HIP_CHECK(hipEventCreate(&e));
  hipStream_t s = nullptr;
  const int src_stride = has_src ? src_vec->size[1] : 1, sc_stride = has_src ? src_coords_vec->size[1] : 3;
  const int rec_stride = has_rec ? rec_vec->size[1] : 1, rc_stride = has_rec ? rec_coords_vec->size[1] : 3;
  auto mod3 = [](int t) { return ((t % 3) + 3) % 3; };
  int n_units = 0;   // timed units: steps, or pairs of steps

  // Order 4, boxes the two-step sweep is made for (16 M cells and more, rows of whole float4s): two damped
  // steps per sweep (fdtd_march2.hip with the damping field as a sixth stream: 12 instead of 20 bytes per
  // cell update).  Four level buffers rotate as in the operators' time loop: a level written over the slot
  // of u(t-1) in the reference's ring inherits that slot's halo, so the shell of prev is copied into the
  // buffer that receives u(t+2) before every pair unless all shells are the same and no source corner
  // reaches the halo layer.  Receivers sample u(time) and u(time+1) behind the sweep: u(time+1) is
  // complete in memory then, first scatter included.
  const char *fe = getenv("FDTD_FUSE2");
  const bool fuse = order == FDTD_STENCIL_ORDER && time_M - time_m + 1 >= 4 && (!fe || atoi(fe) != 0) &&
                    fdtd_march2_preferred(&g) &&
                    fdtd_march2_eligible(&g, d_m, d_u, d_u + vol, d_u + 2 * vol, d_damp, has_src ? p_src_M - p_src_m + 1 : 0);
  g_forward_last_fused = fuse ? 1 : 0;
  if (fuse) {
    float *d_extra = nullptr;
    if ((rc = buf.alloc((void **)&d_extra, vol * sizeof(float)))) return rc;
    // the fourth buffer starts as a copy of a ring slot: same padding beyond the shell as the caller's arrays
    FDTD_HIP_CHECK(hipMemcpyAsync(d_extra, d_u, vol * sizeof(float), hipMemcpyDeviceToDevice, s));
    float *P[4] = {d_u, d_u + vol, d_u + 2 * vol, d_extra};
    const bool sources_inside =
        !has_src || fdtd_host_sources_inside((const float *)src_coords_vec->data, sc_stride, p_src_m, p_src_M, g);
    const bool follow = !sources_inside || !fdtd_host_shells_identical((const float *)u_vec->data, vol, g);
    int cur = mod3(time_m), prev = mod3(time_m + 2), third = mod3(time_m + 1), spare = 3;
    auto sample = [&](const float *level, int time) {
      return has_rec ? fdtd_hip_receiver_sample_order(&g, level, 0, d_rc, d_rec, time, p_rec_m, p_rec_M, rec_stride,
                                                      rc_stride, order, s)
                     : 0;
    };
    int time = time_m;
    while (time <= time_M) {
      const bool timed = n_timed > 0 && time >= first_timed;
      // a pair must not straddle the start of the timed window
      const bool pair = time + 1 <= time_M && !(n_timed > 0 && time < first_timed && time + 1 >= first_timed);
      const int k = timed ? n_units : -1;
      if (k >= 0) FDTD_HIP_CHECK(hipEventRecord(ev[(size_t)k * 3], s));
      if (pair) {
        if (follow && (rc = fdtd_hip_copy_shell(&g, P[spare], P[prev], 0, s))) return rc;
        rc = fdtd_two_steps_impl(&g, d_m, d_src, d_sc, P[cur], P[prev], P[third], P[spare], time, has_src ? p_src_m : 0,
                                 has_src ? p_src_M : -1, src_stride, sc_stride, FDTD_VARIANT_AUTO, sources_inside ? 1 : 0,
                                 0, 0, s, nullptr, nullptr, nullptr, 0, d_damp);
        if (rc) return rc;
        if (k >= 0) FDTD_HIP_CHECK(hipEventRecord(ev[(size_t)k * 3 + 1], s));
        if ((rc = sample(P[cur], time)) || (rc = sample(P[third], time + 1))) return rc;
        const int ncur = spare, nprev = third, nthird = cur, nspare = prev;
        cur = ncur; prev = nprev; third = nthird; spare = nspare;
        time += 2;
      } else {
        if ((rc = fdtd_stencil_levels_damped(&g, d_m, d_damp, P[cur], P[prev], P[third], order, s))) return rc;
        if (k >= 0) FDTD_HIP_CHECK(hipEventRecord(ev[(size_t)k * 3 + 1], s));
        if (has_src && (rc = fdtd_inject_level(&g, d_m, d_src, d_sc, P[third], time, p_src_m, p_src_M, src_stride,
                                               sc_stride, 0, g.x_m - 1, g.x_M + 1, 0, 0, 0, s)))
          return rc;
        if ((rc = sample(P[cur], time))) return rc;
        const int ncur = third, nprev = cur, nthird = prev;
        cur = ncur; prev = nprev; third = nthird;
        time += 1;
      }
      if (k >= 0) FDTD_HIP_CHECK(hipEventRecord(ev[(size_t)k * 3 + 2], s));
      if (timed) ++n_units;
    }
    // u(T+1), u(T), u(T-1) go where the reference's ring leaves them
    const int pos[3] = {cur, prev, third};
    const int tgt[3] = {mod3(time_M + 1), mod3(time_M), mod3(time_M + 2)};
    if ((rc = fdtd_ring_settle(&g, P, pos, spare, tgt, s))) return rc;
  } else {
    for (int time = time_m; time <= time_M; ++time) {
      const int t0 = mod3(time), t1 = mod3(time + 2), t2 = mod3(time + 1);
      const int k = (n_timed > 0 && time >= first_timed) ? time - first_timed : -1;
      if (k >= 0) FDTD_HIP_CHECK(hipEventRecord(ev[(size_t)k * 3], s));
      if ((rc = fdtd_hip_stencil_step_damped_order(&g, d_m, d_damp, d_u, t0, t1, t2, order, s))) return rc;
      if (k >= 0) FDTD_HIP_CHECK(hipEventRecord(ev[(size_t)k * 3 + 1], s));
      if (has_src && (rc = fdtd_hip_source_inject_order(&g, d_m, d_src, d_sc, d_u, t2, time, p_src_m, p_src_M, src_stride,
                                                        sc_stride, order, s)))
        return rc;
      if (has_rec &&
          (rc = fdtd_hip_receiver_sample_order(&g, d_u, t0, d_rc, d_rec, time, p_rec_m, p_rec_M, rec_stride, rc_stride, order, s)))
        return rc;
      if (k >= 0) FDTD_HIP_CHECK(hipEventRecord(ev[(size_t)k * 3 + 2], s));
    }
    n_units = n_timed;
  }
  FDTD_HIP_CHECK(hipStreamSynchronize(s));
  for (int k = 0; k < n_units; ++k) {
    float a = 0.f, b = 0.f;
    FDTD_HIP_CHECK(hipEventElapsedTime(&a, ev[(size_t)k * 3], ev[(size_t)k * 3 + 1]));
    FDTD_HIP_CHECK(hipEventElapsedTime(&b, ev[(size_t)k * 3 + 1], ev[(size_t)k * 3 + 2]));
    timers->section0 += a * 1e-3;
    timers->section1 += b * 1e-3;
  }
  if ((rc = copier.to_host(u_vec->data, d_u, 3 * vol * sizeof(float)))) return rc;
  if (has_rec) FDTD_HIP_CHECK(hipMemcpy(rec_vec->data, d_rec, n_rec * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

// 1 when the last Kernel_HIP_Forward call ran two steps per sweep, 0 when it ran one sweep per step.
extern "C" int fdtd_hip_forward_last_call_fused(void) { return g_forward_last_fused; }

// Kernel_HIP_Forward at a space order given with the call (this call, this thread; the process-wide setting is
// neither read nor changed).
extern "C" int Kernel_HIP_ForwardOrder(struct dataobj *damp_vec, struct dataobj *m_vec, struct dataobj *rec_vec,
                                       struct dataobj *rec_coords_vec, struct dataobj *src_vec,
                                       struct dataobj *src_coords_vec, struct dataobj *u_vec, const int x_M,
                                       const int x_m, const int y_M, const int y_m, const int z_M, const int z_m,
                                       const float dt, const float h_x, const float h_y, const float h_z, const float o_x,
                                       const float o_y, const float o_z, const int p_rec_M, const int p_rec_m,
                                       const int p_src_M, const int p_src_m, const int time_M, const int time_m,
                                       const int deviceid, const int devicerm, struct profiler *timers, int space_order) {
  if (!fdtd_space_order_supported(space_order)) return (int)hipErrorInvalidValue;
  OrderScope scope(space_order);
  return Kernel_HIP_Forward(damp_vec, m_vec, rec_vec, rec_coords_vec, src_vec, src_coords_vec, u_vec, x_M, x_m, y_M, y_m,
                            z_M, z_m, dt, h_x, h_y, h_z, o_x, o_y, o_z, p_rec_M, p_rec_m, p_src_M, p_src_m, time_M, time_m,
                            deviceid, devicerm, timers);
}
