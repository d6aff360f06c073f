/*
 * fdtd_oracle.c — CPU restatement of the reference's OpenACC operator.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check and the
 * benchmark's cpu_baseline leg load this library; the shipped engine
 * (libfdtd_hip.so) never calls into it and has no CPU fallback.
 *
 * What it restates (all citations are into the reference tree):
 *   openacc.cpp:84-87    r1 = 1/dt^2, r2..r4 = 1/h^2, computed in float on the host
 *   openacc.cpp:90-93    ring indices t0 = time%3, t1 = (time+2)%3, t2 = (time+1)%3
 *   openacc.cpp:90-145   the first 5 steps are real but untimed
 *   openacc.cpp:159-164  13-point star Laplacian + leapfrog update ("section0")
 *   openacc.cpp:173      guard of the source section
 *   openacc.cpp:185-196  trilinear 8-corner source scatter ("section1")
 *
 * Arithmetic contract: every operation is an IEEE-754 binary32 operation in the
 * order and association of the reference expression; build with -ffp-contract=off
 * and without -ffast-math so that no FMA contraction or re-association happens.
 * Under that contract a serial run is bit-identical to the reference translation
 * unit compiled with g++ -O2 (checked by tests/test_oracle.py when the reference build
 * under oracle/_ref/ is present).
 *
 * Pinning: the reference ships no golden vectors for this path.  The restatement is
 * pinned by (a) bit-comparison with the compiled reference (oracle/_ref, built by
 * oracle/Makefile from the sources where they lie) and (b) the fixtures under
 * tests/golden/ generated from that compiled reference by
 * tests/golden/make_golden.py.
 *
 * The x loop is parallelised with OpenMP when built with -fopenmp.  Cells of one
 * sweep are independent (reads from levels t0/t1, writes to level t2), so the
 * thread count does not change any bit of the result.  The source scatter stays
 * serial: it is O(8*nsrc) and its += order (p_src outermost, then x, y, z corner)
 * is the order a serial run of the reference uses.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stddef.h>
#include <sys/time.h>

#include "../include/fdtd_abi.h"

#ifdef _OPENMP
#include <omp.h>
#endif

/* Number of OpenMP threads of the following sweeps (the benchmark's CPU baseline times the
 * oracle on all threads of the job and on one). */
void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

static double now_s(void) {
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return (double)tv.tv_sec + (double)tv.tv_usec / 1000000.0;
}

/* openacc.cpp:159-164.  One interior sweep: u[t2] <- f(u[t0], u[t1], m). */
static void oracle_section0(const float *restrict m, const float *restrict u0,
                            const float *restrict u1, float *restrict u2,
                            long nyp, long nzp, int x_m, int x_M, int y_m, int y_M,
                            int z_m, int z_M, float dt, float r1, float r2,
                            float r3, float r4) {
  const long sx = nyp * nzp; /* stride of one x step */
  const long sy = nzp;       /* stride of one y step */
#pragma omp parallel for collapse(2) schedule(static)
  for (int x = x_m; x <= x_M; x += 1) {
    for (int y = y_m; y <= y_M; y += 1) {
      const long row = (long)(x + 4) * sx + (long)(y + 4) * sy + 4;
      for (int z = z_m; z <= z_M; z += 1) {
        const long c = row + z;
        const float uc = u0[c];
        const float r5 = -2.50F * uc;
        const float dxx = r5 + (-8.33333333e-2F) * (u0[c - 2 * sx] + u0[c + 2 * sx]) +
                          1.333333330F * (u0[c - sx] + u0[c + sx]);
        const float dyy = r5 + (-8.33333333e-2F) * (u0[c - 2 * sy] + u0[c + 2 * sy]) +
                          1.333333330F * (u0[c - sy] + u0[c + sy]);
        const float dzz = r5 + (-8.33333333e-2F) * (u0[c - 2] + u0[c + 2]) +
                          1.333333330F * (u0[c - 1] + u0[c + 1]);
        const float mc = m[c];
        u2[c] = dt * dt *
                (r2 * dxx + r3 * dyy + r4 * dzz - (-2.0F * r1 * uc + r1 * u1[c]) * mc) /
                mc;
      }
    }
  }
}

/* openacc.cpp:185-196.  Serial, in the loop order of the reference. */
static void oracle_section1(const float *restrict m, const float *restrict src,
                            const float *restrict coords, float *restrict u2,
                            long nyp, long nzp, int x_m, int x_M, int y_m, int y_M,
                            int z_m, int z_M, float h_x, float h_y, float h_z,
                            float o_x, float o_y, float o_z, int p_src_m,
                            int p_src_M, int time, long src_stride,
                            long coord_stride, int x_shift, int x_lo, int x_hi, int mode,
                            int gx_m, int gx_M) {
  /* x_shift/x_lo/x_hi: slab form used by the decomposition tests (arrays hold global planes
   * x_shift-4.., a corner is applied iff its global x lies in [x_lo, x_hi]); the reference
   * itself is the case x_shift = 0, x_lo = x_m-1, x_hi = x_M+1.  mode 1 / 2 keep only the corners
   * that land outside / inside the box [gx_m,gx_M] x [y_m,y_M] x [z_m,z_M] (tests of the
   * two-steps-per-call schedule, which treats the two kinds at different moments); 0 keeps all. */
  const long sx = nyp * nzp, sy = nzp;
  for (int p = p_src_m; p <= p_src_M; p += 1) {
    for (int rx = 0; rx <= 1; rx += 1) {
      for (int ry = 0; ry <= 1; ry += 1) {
        for (int rz = 0; rz <= 1; rz += 1) {
          const float gx = (-o_x + coords[p * coord_stride + 0]) / h_x;
          const float gy = (-o_y + coords[p * coord_stride + 1]) / h_y;
          const float gz = (-o_z + coords[p * coord_stride + 2]) / h_z;
          const int posx = (int)floorf(gx);
          const int posy = (int)floorf(gy);
          const int posz = (int)floorf(gz);
          const float px = -floorf(gx) + gx;
          const float py = -floorf(gy) + gy;
          const float pz = -floorf(gz) + gz;
          (void)x_m;
          (void)x_M;
          if (rx + posx >= x_lo && ry + posy >= y_m - 1 && rz + posz >= z_m - 1 &&
              rx + posx <= x_hi && ry + posy <= y_M + 1 && rz + posz <= z_M + 1) {
            const int inside = rx + posx >= gx_m && rx + posx <= gx_M && ry + posy >= y_m &&
                               ry + posy <= y_M && rz + posz >= z_m && rz + posz <= z_M;
            if ((mode == 1 && inside) || (mode == 2 && !inside)) continue;
            const float r0 = 1.0e-2F * (rx * px + (1 - rx) * (1 - px)) *
                             (ry * py + (1 - ry) * (1 - py)) *
                             (rz * pz + (1 - rz) * (1 - pz)) *
                             src[(long)time * src_stride + p] /
                             m[(long)(posx - x_shift + 4) * sx + (long)(posy + 4) * sy + (posz + 4)];
            u2[(long)(rx + posx - x_shift + 4) * sx + (long)(ry + posy + 4) * sy +
               (rz + posz + 4)] += r0;
          }
        }
      }
    }
  }
}

/* The operator, same signature and semantics as Kernel_OpenACC (openacc.cpp:61). */
int Oracle_Kernel_OpenACC(FDTD_OPERATOR_ARGS) {
  (void)deviceid;
  (void)devicerm;
  const long nxp = u_vec->size[1], nyp = u_vec->size[2], nzp = u_vec->size[3];
  const long vol = nxp * nyp * nzp;
  float *u = (float *)u_vec->data;
  const float *m = (const float *)m_vec->data;
  const float *src = (const float *)src_vec->data;
  const float *coords = (const float *)src_coords_vec->data;
  const long src_stride = src_vec->size[1];
  const long coord_stride = src_coords_vec->size[1];

  const float r1 = 1.0F / (dt * dt);
  const float r2 = 1.0F / (h_x * h_x);
  const float r3 = 1.0F / (h_y * h_y);
  const float r4 = 1.0F / (h_z * h_z);
  const int has_src =
      (src_vec->size[0] * src_vec->size[1] > 0 && p_src_M - p_src_m + 1 > 0);

  for (int time = time_m; time <= time_M; time += 1) {
    const int t0 = time % 3, t1 = (time + 2) % 3, t2 = (time + 1) % 3;
    const int timed = (time >= time_m + FDTD_WARMUP_STEPS) && timers;
    double a = timed ? now_s() : 0.0;
    oracle_section0(m, u + t0 * vol, u + t1 * vol, u + t2 * vol, nyp, nzp, x_m, x_M,
                    y_m, y_M, z_m, z_M, dt, r1, r2, r3, r4);
    double b = timed ? now_s() : 0.0;
    if (has_src)
      oracle_section1(m, src, coords, u + t2 * vol, nyp, nzp, x_m, x_M, y_m, y_M, z_m,
                      z_M, h_x, h_y, h_z, o_x, o_y, o_z, p_src_m, p_src_M, time,
                      src_stride, coord_stride, 0, x_m - 1, x_M + 1, 0, 0, 0);
    if (timed) {
      double c = now_s();
      timers->section0 += b - a;
      timers->section1 += c - b;
    }
  }
  return 0;
}

/* Single sweeps on caller-chosen levels; used by the slab-decomposition tests, which
 * need a stepper that can be driven one time step at a time. */
void oracle_stencil_step(const float *m, float *u, long nxp, long nyp, long nzp,
                         int x_m, int x_M, int y_m, int y_M, int z_m, int z_M,
                         float dt, float h_x, float h_y, float h_z, int t0, int t1,
                         int t2) {
  const long vol = nxp * nyp * nzp;
  const float r1 = 1.0F / (dt * dt);
  const float r2 = 1.0F / (h_x * h_x);
  const float r3 = 1.0F / (h_y * h_y);
  const float r4 = 1.0F / (h_z * h_z);
  oracle_section0(m, u + t0 * vol, u + t1 * vol, u + t2 * vol, nyp, nzp, x_m, x_M,
                  y_m, y_M, z_m, z_M, dt, r1, r2, r3, r4);
}

void oracle_source_inject(const float *m, const float *src, const float *coords,
                          float *u, long nxp, long nyp, long nzp, int x_m, int x_M,
                          int y_m, int y_M, int z_m, int z_M, float h_x, float h_y,
                          float h_z, float o_x, float o_y, float o_z, int p_src_m,
                          int p_src_M, int time, long src_stride, long coord_stride,
                          int t2) {
  const long vol = nxp * nyp * nzp;
  oracle_section1(m, src, coords, u + t2 * vol, nyp, nzp, x_m, x_M, y_m, y_M, z_m,
                  z_M, h_x, h_y, h_z, o_x, o_y, o_z, p_src_m, p_src_M, time,
                  src_stride, coord_stride, 0, x_m - 1, x_M + 1, 0, 0, 0);
}

void oracle_source_inject_slab(const float *m, const float *src, const float *coords,
                               float *u, long nxp, long nyp, long nzp, int y_m, int y_M,
                               int z_m, int z_M, float h_x, float h_y, float h_z, float o_x,
                               float o_y, float o_z, int p_src_m, int p_src_M, int time,
                               long src_stride, long coord_stride, int t2, int x_shift,
                               int x_lo, int x_hi) {
  const long vol = nxp * nyp * nzp;
  oracle_section1(m, src, coords, u + t2 * vol, nyp, nzp, 0, 0, y_m, y_M, z_m, z_M, h_x,
                  h_y, h_z, o_x, o_y, o_z, p_src_m, p_src_M, time, src_stride, coord_stride,
                  x_shift, x_lo, x_hi, 0, 0, 0);
}

/* The slab scatter restricted to corners outside (mode 1) or inside (mode 2) the domain's box. */
void oracle_source_inject_slab_mode(const float *m, const float *src, const float *coords,
                                    float *u, long nxp, long nyp, long nzp, int y_m, int y_M,
                                    int z_m, int z_M, float h_x, float h_y, float h_z, float o_x,
                                    float o_y, float o_z, int p_src_m, int p_src_M, int time,
                                    long src_stride, long coord_stride, int t2, int x_shift,
                                    int x_lo, int x_hi, int mode, int gx_m, int gx_M) {
  const long vol = nxp * nyp * nzp;
  oracle_section1(m, src, coords, u + t2 * vol, nyp, nzp, 0, 0, y_m, y_M, z_m, z_M, h_x,
                  h_y, h_z, o_x, o_y, o_z, p_src_m, p_src_M, time, src_stride, coord_stride,
                  x_shift, x_lo, x_hi, mode, gx_m, gx_M);
}

/* main.cpp:573-593, the driver's parity metrics, restated independently of the
 * product's own copy so the two can be checked against each other. */
void oracle_compare_fields(const float *test, const float *ref, size_t n,
                           struct fdtd_parity *out) {
  double max_abs = 0.0, max_rel = 0.0, l2d = 0.0, l2r = 0.0;
  long nan_count = 0, inf_count = 0;
  for (size_t i = 0; i < n; ++i) {
    if (isnan(test[i])) { nan_count++; continue; }
    if (isinf(test[i])) { inf_count++; continue; }
    const double diff = fabs((double)(test[i] - ref[i]));
    const double abs_ref = fabs((double)ref[i]);
    if (diff > max_abs) max_abs = diff;
    if (abs_ref > 1e-10 && diff / abs_ref > max_rel) max_rel = diff / abs_ref;
    l2d += diff * diff;
    l2r += (double)(ref[i] * ref[i]);
  }
  out->max_abs = max_abs;
  out->max_rel = max_rel;
  out->l2 = sqrt(l2d / (l2r + 1e-30));
  out->nan_count = nan_count;
  out->inf_count = inf_count;
}

/* ---------------------------------------------------------------------------------------------
 * Other space orders (6, 8, 10, 12).  The reference driver is parameterised on STENCIL_ORDER
 * ("Can be 4, 6, 8, 10, 12") with HALO = STENCIL_ORDER cells of padding per side
 * (main.cpp:27-32,129-136), but every reference kernel hard-codes order 4 (openacc.cpp:102-107,
 * cuda.cu:55), so there is NO reference arithmetic for the other orders: PARITY UNPINNED for them.  What is restated here is the order-4 expression
 * (openacc.cpp:159-164) with its pattern continued: the central second-difference weights of the
 * order as 9-digit float literals, the way the order-4 ones are printed (-2.5, 1.33333333,
 * -8.33333333e-2), each axis summed from the farthest pair of neighbours to the nearest, padding
 * offset = order.  With order 4 this function produces the bits of oracle_section0 (checked by the
 * tests on every fixture), which pins the generic form itself.
 * ------------------------------------------------------------------------------------------- */
static const float k_order_weights[5][7] = {
    /* order 4 */ {-2.50F, 1.333333330F, -8.33333333e-2F, 0.0F, 0.0F, 0.0F, 0.0F},
    /* order 6 */ {-2.72222222F, 1.5F, -1.5e-1F, 1.11111111e-2F, 0.0F, 0.0F, 0.0F},
    /* order 8 */ {-2.84722222F, 1.6F, -2.0e-1F, 2.53968254e-2F, -1.78571429e-3F, 0.0F, 0.0F},
    /* order 10: -5269/1800, 5/3, -5/21, 5/126, -5/1008, 1/3150 */
    {-2.92722222F, 1.66666667F, -2.38095238e-1F, 3.96825397e-2F, -4.96031746e-3F, 3.17460317e-4F, 0.0F},
    /* order 12: -5369/1800, 12/7, -15/56, 10/189, -1/112, 2/1925, -1/16632 */
    {-2.98277778F, 1.71428571F, -2.67857143e-1F, 5.29100529e-2F, -8.92857143e-3F, 1.03896104e-3F,
     -6.01250601e-5F},
};

static void oracle_section0_order(int order, const float *restrict m, const float *restrict u0,
                                  const float *restrict u1, float *restrict u2, long nyp, long nzp,
                                  int x_m, int x_M, int y_m, int y_M, int z_m, int z_M, float dt,
                                  float r1, float r2, float r3, float r4) {
  const int R = order / 2, H = order;
  const float *w = k_order_weights[R - 2];
  const long sx = nyp * nzp, sy = nzp;
#pragma omp parallel for collapse(2) schedule(static)
  for (int x = x_m; x <= x_M; x += 1) {
    for (int y = y_m; y <= y_M; y += 1) {
      const long row = (long)(x + H) * sx + (long)(y + H) * sy + H;
      for (int z = z_m; z <= z_M; z += 1) {
        const long c = row + z;
        const float uc = u0[c];
        const float r5 = w[0] * uc;
        float dxx = r5, dyy = r5, dzz = r5;
        for (int k = R; k >= 1; --k) {
          dxx = dxx + w[k] * (u0[c - k * sx] + u0[c + k * sx]);
          dyy = dyy + w[k] * (u0[c - k * sy] + u0[c + k * sy]);
          dzz = dzz + w[k] * (u0[c - k] + u0[c + k]);
        }
        const float mc = m[c];
        u2[c] = dt * dt *
                (r2 * dxx + r3 * dyy + r4 * dzz - (-2.0F * r1 * uc + r1 * u1[c]) * mc) /
                mc;
      }
    }
  }
}

/* The scatter of openacc.cpp:185-196 on arrays padded by H cells per side. */
static void oracle_section1_order(int H, const float *restrict m, const float *restrict src,
                                  const float *restrict coords, float *restrict u2, long nyp,
                                  long nzp, int x_m, int x_M, int y_m, int y_M, int z_m, int z_M,
                                  float h_x, float h_y, float h_z, float o_x, float o_y, float o_z,
                                  int p_src_m, int p_src_M, int time, long src_stride,
                                  long coord_stride, int x_shift, int x_lo, int x_hi) {
  /* x_shift / x_lo / x_hi: the slab form of oracle_section1 (arrays hold global planes x_shift-H ..,
   * a corner is applied iff its global x lies in [x_lo, x_hi]); a single domain is the case
   * x_shift = 0, x_lo = x_m - 1, x_hi = x_M + 1. */
  const long sx = nyp * nzp, sy = nzp;
  (void)x_m; (void)x_M;
  for (int p = p_src_m; p <= p_src_M; p += 1)
    for (int rx = 0; rx <= 1; rx += 1)
      for (int ry = 0; ry <= 1; ry += 1)
        for (int rz = 0; rz <= 1; rz += 1) {
          const float gx = (-o_x + coords[p * coord_stride + 0]) / h_x;
          const float gy = (-o_y + coords[p * coord_stride + 1]) / h_y;
          const float gz = (-o_z + coords[p * coord_stride + 2]) / h_z;
          const int posx = (int)floorf(gx), posy = (int)floorf(gy), posz = (int)floorf(gz);
          const float px = -floorf(gx) + gx, py = -floorf(gy) + gy, pz = -floorf(gz) + gz;
          if (rx + posx >= x_lo && ry + posy >= y_m - 1 && rz + posz >= z_m - 1 &&
              rx + posx <= x_hi && ry + posy <= y_M + 1 && rz + posz <= z_M + 1) {
            const float r0 = 1.0e-2F * (rx * px + (1 - rx) * (1 - px)) *
                             (ry * py + (1 - ry) * (1 - py)) * (rz * pz + (1 - rz) * (1 - pz)) *
                             src[(long)time * src_stride + p] /
                             m[(long)(posx - x_shift + H) * sx + (long)(posy + H) * sy + (posz + H)];
            u2[(long)(rx + posx - x_shift + H) * sx + (long)(ry + posy + H) * sy + (rz + posz + H)] += r0;
          }
        }
}

/* The operator for space order 4, 6, 8, 10 or 12 on arrays padded by `order` cells per side. */
int Oracle_Kernel_SpaceOrder(FDTD_OPERATOR_ARGS, int order) {
  (void)deviceid;
  (void)devicerm;
  if (order < 4 || order > 12 || order % 2) return 1;
  const long nxp = u_vec->size[1], nyp = u_vec->size[2], nzp = u_vec->size[3];
  const long vol = nxp * nyp * nzp;
  float *u = (float *)u_vec->data;
  const float *m = (const float *)m_vec->data;
  const float *src = (const float *)src_vec->data;
  const float *coords = (const float *)src_coords_vec->data;
  const long src_stride = src_vec->size[1], coord_stride = src_coords_vec->size[1];
  const float r1 = 1.0F / (dt * dt), r2 = 1.0F / (h_x * h_x), r3 = 1.0F / (h_y * h_y),
              r4 = 1.0F / (h_z * h_z);
  const int has_src = (src_vec->size[0] * src_vec->size[1] > 0 && p_src_M - p_src_m + 1 > 0);
  for (int time = time_m; time <= time_M; time += 1) {
    const int t0 = time % 3, t1 = (time + 2) % 3, t2 = (time + 1) % 3;
    const int timed = (time >= time_m + FDTD_WARMUP_STEPS) && timers;
    const double a = timed ? now_s() : 0.0;
    oracle_section0_order(order, m, u + t0 * vol, u + t1 * vol, u + t2 * vol, nyp, nzp, x_m, x_M,
                          y_m, y_M, z_m, z_M, dt, r1, r2, r3, r4);
    const double b = timed ? now_s() : 0.0;
    if (has_src)
      oracle_section1_order(order, m, src, coords, u + t2 * vol, nyp, nzp, x_m, x_M, y_m, y_M, z_m,
                            z_M, h_x, h_y, h_z, o_x, o_y, o_z, p_src_m, p_src_M, time, src_stride,
                            coord_stride, 0, x_m - 1, x_M + 1);
    if (timed) {
      timers->section0 += b - a;
      timers->section1 += now_s() - b;
    }
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Absorbing sponge and receivers (the steps either side of the hot path in a Devito-style propagator).
 * The reference has neither (fixed halo frame, nothing samples the field): PARITY UNPINNED.  Restated
 * here is the form the HIP operator Kernel_HIP_Forward evaluates, so that the two can be compared bit
 * for bit: the update of openacc.cpp:159-164 with a first-order damping term,
 *     u+ = dt^2 ( lap - (-2 r1 u + r1 u-) m + r0 damp u ) / (m + dt damp),     r0 = 1/dt,
 * and the receiver interpolation as the transpose of the scatter of openacc.cpp:185-196 (same
 * positions, weights, bounds test and loop order), sampling the level the step starts from.
 * ------------------------------------------------------------------------------------------- */
int Oracle_Kernel_Forward(struct dataobj *damp_vec, struct dataobj *m_vec, struct dataobj *rec_vec,
                          struct dataobj *rec_coords_vec, struct dataobj *src_vec,
                          struct dataobj *src_coords_vec, struct dataobj *u_vec, const int x_M,
                          const int x_m, const int y_M, const int y_m, const int z_M, const int z_m,
                          const float dt, const float h_x, const float h_y, const float h_z,
                          const float o_x, const float o_y, const float o_z, const int p_rec_M,
                          const int p_rec_m, const int p_src_M, const int p_src_m, const int time_M,
                          const int time_m, const int deviceid, const int devicerm,
                          struct profiler *timers) {
  (void)deviceid; (void)devicerm; (void)timers;
  const long nxp = u_vec->size[1], nyp = u_vec->size[2], nzp = u_vec->size[3];
  const long vol = nxp * nyp * nzp, sx = nyp * nzp, sy = nzp;
  float *u = (float *)u_vec->data;
  const float *m = (const float *)m_vec->data, *damp = (const float *)damp_vec->data;
  const float *src = (const float *)src_vec->data, *scoords = (const float *)src_coords_vec->data;
  float *rec = (float *)rec_vec->data;
  const float *rcoords = (const float *)rec_coords_vec->data;
  const float r0 = 1.0F / dt, r1 = 1.0F / (dt * dt), r2 = 1.0F / (h_x * h_x), r3 = 1.0F / (h_y * h_y),
              r4 = 1.0F / (h_z * h_z);
  const int has_src = (src_vec->size[0] * src_vec->size[1] > 0 && p_src_M - p_src_m + 1 > 0);
  for (int time = time_m; time <= time_M; time += 1) {
    const int t0 = time % 3, t1 = (time + 2) % 3, t2 = (time + 1) % 3;
    const float *u0 = u + t0 * vol, *u1 = u + t1 * vol;
    float *u2 = u + t2 * vol;
#pragma omp parallel for collapse(2) schedule(static)
    for (int x = x_m; x <= x_M; x += 1)
      for (int y = y_m; y <= y_M; y += 1)
        for (int z = z_m; z <= z_M; z += 1) {
          const long c = (long)(x + 4) * sx + (long)(y + 4) * sy + (z + 4);
          const float uc = u0[c];
          const float r5 = -2.50F * uc;
          const float dxx = r5 + (-8.33333333e-2F) * (u0[c - 2 * sx] + u0[c + 2 * sx]) +
                            1.333333330F * (u0[c - sx] + u0[c + sx]);
          const float dyy = r5 + (-8.33333333e-2F) * (u0[c - 2 * sy] + u0[c + 2 * sy]) +
                            1.333333330F * (u0[c - sy] + u0[c + sy]);
          const float dzz = r5 + (-8.33333333e-2F) * (u0[c - 2] + u0[c + 2]) +
                            1.333333330F * (u0[c - 1] + u0[c + 1]);
          const float mc = m[c], dc = damp[c];
          u2[c] = dt * dt *
                  (r2 * dxx + r3 * dyy + r4 * dzz - (-2.0F * r1 * uc + r1 * u1[c]) * mc + r0 * dc * uc) /
                  (mc + dt * dc);
        }
    if (has_src)
      oracle_section1(m, src, scoords, u2, nyp, nzp, x_m, x_M, y_m, y_M, z_m, z_M, h_x, h_y, h_z, o_x,
                      o_y, o_z, p_src_m, p_src_M, time, src_vec->size[1], src_coords_vec->size[1], 0,
                      x_m - 1, x_M + 1, 0, 0, 0);
    for (int p = p_rec_m; p <= p_rec_M; p += 1) {
      const long cs = rec_coords_vec->size[1];
      const float gx = (-o_x + rcoords[p * cs + 0]) / h_x, gy = (-o_y + rcoords[p * cs + 1]) / h_y,
                  gz = (-o_z + rcoords[p * cs + 2]) / h_z;
      const int posx = (int)floorf(gx), posy = (int)floorf(gy), posz = (int)floorf(gz);
      const float px = -floorf(gx) + gx, py = -floorf(gy) + gy, pz = -floorf(gz) + gz;
      float sum = 0.0F;
      for (int rx = 0; rx <= 1; rx += 1)
        for (int ry = 0; ry <= 1; ry += 1)
          for (int rz = 0; rz <= 1; rz += 1)
            if (rx + posx >= x_m - 1 && ry + posy >= y_m - 1 && rz + posz >= z_m - 1 &&
                rx + posx <= x_M + 1 && ry + posy <= y_M + 1 && rz + posz <= z_M + 1)
              sum += (rx * px + (1 - rx) * (1 - px)) * (ry * py + (1 - ry) * (1 - py)) *
                     (rz * pz + (1 - rz) * (1 - pz)) *
                     u0[(long)(rx + posx + 4) * sx + (long)(ry + posy + 4) * sy + (rz + posz + 4)];
      rec[(long)time * rec_vec->size[1] + p] = sum;
    }
  }
  return 0;
}

/* oracle_section0_order with the damping term of the forward operator. */
static void oracle_section0_order_damped(int order, const float *restrict m, const float *restrict damp,
                                         const float *restrict u0, const float *restrict u1,
                                         float *restrict u2, long nyp, long nzp, int x_m, int x_M, int y_m,
                                         int y_M, int z_m, int z_M, float dt, float r0, float r1, float r2,
                                         float r3, float r4) {
  const int R = order / 2, H = order;
  const float *w = k_order_weights[R - 2];
  const long sx = nyp * nzp, sy = nzp;
#pragma omp parallel for collapse(2) schedule(static)
  for (int x = x_m; x <= x_M; x += 1)
    for (int y = y_m; y <= y_M; y += 1)
      for (int z = z_m; z <= z_M; z += 1) {
        const long c = (long)(x + H) * sx + (long)(y + H) * sy + (z + H);
        const float uc = u0[c];
        const float r5 = w[0] * uc;
        float dxx = r5, dyy = r5, dzz = r5;
        for (int k = R; k >= 1; --k) {
          dxx = dxx + w[k] * (u0[c - k * sx] + u0[c + k * sx]);
          dyy = dyy + w[k] * (u0[c - k * sy] + u0[c + k * sy]);
          dzz = dzz + w[k] * (u0[c - k] + u0[c + k]);
        }
        const float mc = m[c], dc = damp[c];
        u2[c] = dt * dt *
                (r2 * dxx + r3 * dyy + r4 * dzz - (-2.0F * r1 * uc + r1 * u1[c]) * mc + r0 * dc * uc) /
                (mc + dt * dc);
      }
}

/* Single sweeps at any space order on caller-chosen levels (damp may be NULL: no damping term), and the
 * slab form of the scatter on arrays padded by H cells; used by the slab-decomposition tests. */
void oracle_stencil_step_order(int order, const float *m, const float *damp, float *u, long nxp, long nyp,
                               long nzp, int x_m, int x_M, int y_m, int y_M, int z_m, int z_M, float dt,
                               float h_x, float h_y, float h_z, int t0, int t1, int t2) {
  const long vol = nxp * nyp * nzp;
  const float r0 = 1.0F / dt, r1 = 1.0F / (dt * dt), r2 = 1.0F / (h_x * h_x), r3 = 1.0F / (h_y * h_y),
              r4 = 1.0F / (h_z * h_z);
  if (damp)
    oracle_section0_order_damped(order, m, damp, u + t0 * vol, u + t1 * vol, u + t2 * vol, nyp, nzp, x_m, x_M,
                                 y_m, y_M, z_m, z_M, dt, r0, r1, r2, r3, r4);
  else
    oracle_section0_order(order, m, u + t0 * vol, u + t1 * vol, u + t2 * vol, nyp, nzp, x_m, x_M, y_m, y_M,
                          z_m, z_M, dt, r1, r2, r3, r4);
}

void oracle_source_inject_slab_order(int H, const float *m, const float *src, const float *coords, float *u,
                                     long nxp, long nyp, long nzp, int y_m, int y_M, int z_m, int z_M,
                                     float h_x, float h_y, float h_z, float o_x, float o_y, float o_z,
                                     int p_src_m, int p_src_M, int time, long src_stride, long coord_stride,
                                     int t2, int x_shift, int x_lo, int x_hi) {
  const long vol = nxp * nyp * nzp;
  oracle_section1_order(H, m, src, coords, u + t2 * vol, nyp, nzp, 0, 0, y_m, y_M, z_m, z_M, h_x, h_y, h_z,
                        o_x, o_y, o_z, p_src_m, p_src_M, time, src_stride, coord_stride, x_shift, x_lo, x_hi);
}

/* The same operator for space order 4 .. 12 on arrays padded by `order` cells per side: the generic
 * sweep of oracle_section0_order with the damping term, the scatter and the receivers on the wider
 * padding.  With order 4 it gives the bits of Oracle_Kernel_Forward (checked by the tests). */
int Oracle_Kernel_ForwardOrder(struct dataobj *damp_vec, struct dataobj *m_vec, struct dataobj *rec_vec,
                               struct dataobj *rec_coords_vec, struct dataobj *src_vec,
                               struct dataobj *src_coords_vec, struct dataobj *u_vec, const int x_M,
                               const int x_m, const int y_M, const int y_m, const int z_M,
                               const int z_m, const float dt, const float h_x, const float h_y,
                               const float h_z, const float o_x, const float o_y, const float o_z,
                               const int p_rec_M, const int p_rec_m, const int p_src_M,
                               const int p_src_m, const int time_M, const int time_m,
                               const int deviceid, const int devicerm, struct profiler *timers,
                               int order) {
  (void)deviceid; (void)devicerm; (void)timers;
  if (order < 4 || order > 12 || order % 2) return 1;
  const int H = order;
  const long nxp = u_vec->size[1], nyp = u_vec->size[2], nzp = u_vec->size[3];
  const long vol = nxp * nyp * nzp, sx = nyp * nzp, sy = nzp;
  float *u = (float *)u_vec->data;
  const float *m = (const float *)m_vec->data, *damp = (const float *)damp_vec->data;
  const float *src = (const float *)src_vec->data, *scoords = (const float *)src_coords_vec->data;
  float *rec = (float *)rec_vec->data;
  const float *rcoords = (const float *)rec_coords_vec->data;
  const float r0 = 1.0F / dt, r1 = 1.0F / (dt * dt), r2 = 1.0F / (h_x * h_x), r3 = 1.0F / (h_y * h_y),
              r4 = 1.0F / (h_z * h_z);
  const int has_src = (src_vec->size[0] * src_vec->size[1] > 0 && p_src_M - p_src_m + 1 > 0);
  for (int time = time_m; time <= time_M; time += 1) {
    const int t0 = time % 3, t1 = (time + 2) % 3, t2 = (time + 1) % 3;
    const float *u0 = u + t0 * vol, *u1 = u + t1 * vol;
    float *u2 = u + t2 * vol;
    oracle_section0_order_damped(order, m, damp, u0, u1, u2, nyp, nzp, x_m, x_M, y_m, y_M, z_m, z_M, dt,
                                 r0, r1, r2, r3, r4);
    if (has_src)
      oracle_section1_order(H, m, src, scoords, u2, nyp, nzp, x_m, x_M, y_m, y_M, z_m, z_M, h_x, h_y,
                            h_z, o_x, o_y, o_z, p_src_m, p_src_M, time, src_vec->size[1],
                            src_coords_vec->size[1], 0, x_m - 1, x_M + 1);
    for (int p = p_rec_m; p <= p_rec_M; p += 1) {
      const long cs = rec_coords_vec->size[1];
      const float gx = (-o_x + rcoords[p * cs + 0]) / h_x, gy = (-o_y + rcoords[p * cs + 1]) / h_y,
                  gz = (-o_z + rcoords[p * cs + 2]) / h_z;
      const int posx = (int)floorf(gx), posy = (int)floorf(gy), posz = (int)floorf(gz);
      const float px = -floorf(gx) + gx, py = -floorf(gy) + gy, pz = -floorf(gz) + gz;
      float sum = 0.0F;
      for (int rx = 0; rx <= 1; rx += 1)
        for (int ry = 0; ry <= 1; ry += 1)
          for (int rz = 0; rz <= 1; rz += 1)
            if (rx + posx >= x_m - 1 && ry + posy >= y_m - 1 && rz + posz >= z_m - 1 &&
                rx + posx <= x_M + 1 && ry + posy <= y_M + 1 && rz + posz <= z_M + 1)
              sum += (rx * px + (1 - rx) * (1 - px)) * (ry * py + (1 - ry) * (1 - py)) *
                     (rz * pz + (1 - rz) * (1 - pz)) *
                     u0[(long)(rx + posx + H) * sx + (long)(ry + posy + H) * sy + (rz + posz + H)];
      rec[(long)time * rec_vec->size[1] + p] = sum;
    }
  }
  return 0;
}
