// fdtd_device.h — shared declarations of the gfx950 kernels and their launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fdtd_abi.h"

// FDTD_HALO / FDTD_RADIUS in this library's sources mean the order-4 layout its order-4 kernels are written
// for; the higher orders take their padding as an argument.
static_assert(FDTD_STENCIL_ORDER == 4 && FDTD_HALO == 4 && FDTD_RADIUS == 2,
              "build the library without -DFDTD_STENCIL_ORDER: the space order is a run-time choice here");

// Everything a stencil sweep needs, passed by value in the kernarg segment (scalar
// loads, no per-thread cost).  Level pointers already include the ring offset.
struct StencilArgs {
  const float *m;
  const float *u0;  // level t0 (current)
  const float *u1;  // level t1 (previous)
  float *u2;        // level t2 (next, written)
  long sx, sy;      // element strides of one x / one y step (nyp*nzp, nzp)
  int nxp, nyp, nzp;
  int x_m, x_M, y_m, y_M, z_m, z_M;
  float dt2;        // dt*dt, rounded once on the host like the reference does
  float r1, r2, r3, r4;
};

// Source scatter arguments.
struct InjectArgs {
  const float *m;
  const float *src;
  const float *coords;
  float *u2;
  long sx, sy;
  int x_m, x_M, y_m, y_M, z_m, z_M;
  float h_x, h_y, h_z, o_x, o_y, o_z;
  int p_src_m, p_src_M;
  int time;
  int src_stride, coord_stride;
  // Slab decomposition along x: the arrays hold global planes x_shift-4 .. ; a corner
  // is applied iff its GLOBAL x index lies in [x_lo, x_hi].  Single-domain calls use
  // x_shift = 0, x_lo = x_m - 1, x_hi = x_M + 1 (openacc.cpp:192).
  int x_shift, x_lo, x_hi;
  int target_mode;  // 0: all admitted corners; 1: only those landing outside the domain's box; 2: only inside
  int gx_m, gx_M;   // planes of the domain's box in the coordinates of the source positions (target_mode != 0)
  int halo;         // cells of padding per side of the arrays (FDTD_HALO; the space order for the higher orders)
};

#define FDTD_HIP_CHECK(expr)                  \
  do {                                        \
    hipError_t _e = (expr);                   \
    if (_e != hipSuccess) return (int)_e;     \
  } while (0)

// Launchers on explicit level pointers (defined in fdtd_kernels.hip), used by the time loops that
// rotate more than three level buffers.
int fdtd_stencil_levels(const struct fdtd_geom *g, const float *m, const float *u0, const float *u1,
                        float *u2, int variant, void *stream);
int fdtd_inject_level(const struct fdtd_geom *g, const float *m, const float *src,
                      const float *src_coords, float *level, int time, int p_src_m, int p_src_M,
                      int src_stride, int coord_stride, int x_shift, int x_lo, int x_hi, int target_mode,
                      int gx_m, int gx_M, void *stream);

// The scatter on arrays padded by `halo` cells per side (fdtd_inject_level is the case halo = FDTD_HALO).
int fdtd_inject_level_halo(const struct fdtd_geom *g, const float *m, const float *src,
                           const float *src_coords, float *level, int time, int p_src_m, int p_src_M,
                           int src_stride, int coord_stride, int x_shift, int x_lo, int x_hi, int target_mode,
                           int gx_m, int gx_M, int halo, void *stream);
// fdtd_orders.hip: one sweep of space order 4 .. 12 on arrays padded by `order` cells per side.
int fdtd_stencil_levels_order(const struct fdtd_geom *g, const float *m, const float *u0, const float *u1,
                              float *u2, int order, void *stream);

// fdtd_orders.hip: the same sweep with the damping term of fdtd_forward.hip (damp: padded shape of m).
int fdtd_stencil_levels_damped(const struct fdtd_geom *g, const float *m, const float *damp, const float *u0,
                               const float *u1, float *u2, int order, void *stream);

extern "C" int fdtd_hip_get_space_order(void);
// Space order of the call in progress on this thread: set (returns the value before), 0 = none, in which case
// fdtd_hip_get_space_order reports the process-wide setting.  OrderScope sets it for a scope.
int fdtd_space_order_for_this_call(int order);
bool fdtd_space_order_supported(int order);
struct OrderScope {
  int before;
  explicit OrderScope(int order) : before(fdtd_space_order_for_this_call(order)) {}
  ~OrderScope() { fdtd_space_order_for_this_call(before); }
};

// Sweep and scatter in one launch for small grids (fdtd_kernels.hip).
struct FdtdSmallSource {
  int x, y, z;          // cell of the source position (openacc.cpp:185-187)
  float px, py, pz;     // fractional parts
  float mc;             // m at that cell
};
constexpr int FDTD_SMALL_MAX_SOURCES = 16;
int fdtd_small_prepare(const struct fdtd_geom *g, const float *m, const float *src_coords, int p_src_m, int p_src_M,
                       int coord_stride, FdtdSmallSource *table, void *stream);
int fdtd_small_step(const struct fdtd_geom *g, const float *m, const float *u0, const float *u1, float *u2,
                    const FdtdSmallSource *table, const float *src_row, int nsrc, void *stream);
bool fdtd_auto_is_naive(const struct fdtd_geom *g, const float *m, const float *u);

// fdtd_march2.hip: can the two-steps-per-sweep kernel take this problem?
bool fdtd_march2_preferred(const struct fdtd_geom *g);
bool fdtd_march2_eligible(const struct fdtd_geom *g, const float *m, const float *b, const float *a_prev,
                          const float *c, const float *d, int nsrc);
int fdtd_shells_differ(const struct fdtd_geom *g, const float *a, const float *b, int *flag, void *stream);
int fdtd_two_steps_impl(const struct fdtd_geom *g, const float *m, const float *src,
                        const float *src_coords, const float *cur, const float *prev, float *next1,
                        float *next2, int time, int p_src_m, int p_src_M, int src_stride, int coord_stride,
                        int variant, int flags, int x_shift, int global_nx, void *stream, hipEvent_t ev_begin,
                        hipEvent_t ev_end, float *scratch, long scratch_floats, const float *damp = nullptr);
