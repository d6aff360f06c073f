// fdtd_orders.hip — space orders 6, 8, 10 and 12 (and 4 through the same generic code).
//
// The reference driver is parameterised on STENCIL_ORDER ("Can be 4, 6, 8, 10, 12"), with
// HALO = STENCIL_ORDER cells of padding per side (main.cpp:27-32,129-136), but every reference
// kernel hard-codes order 4 (openacc.cpp:102-107, cuda.cu:55, cuda_optimized.cu:34): there is no
// reference arithmetic for the other orders, so their parity is unpinned.  What this file evaluates
// is the order-4 expression of openacc.cpp:159-164 with its pattern continued: the central
// second-difference weights of the order as 9-digit float literals (the way the order-4 ones are
// printed), each axis summed from the farthest pair of neighbours to the nearest, interior cell i
// at padded index i + order.  With order 4 it returns the bits of the order-4 kernels (tested on
// every fixture).
//
// Two kernels, same bits.  stencil_order_kernel: one thread per cell, z on the lanes, every operand
// from global memory (the shape of cuda.cu:61-108), for small boxes and boxes that reach into the
// padding.  stencil_order_march_kernel: a workgroup marches along x over a tile of rows; the 2R+1
// planes of the x-direction live in registers, the current plane goes through LDS for the y- and
// z-neighbours, so every value of u(t) is fetched from memory once per tile (plus the R rows on
// either side of it), whatever the radius.
#include "fdtd_device.h"

#pragma clang fp contract(off)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_R = 6;
template <int R> struct Weights;
template <> struct Weights<2> {
  static constexpr float w[MAX_R + 1] = {-2.50F, 1.333333330F, -8.33333333e-2F, 0.0F, 0.0F, 0.0F, 0.0F};
};
template <> struct Weights<3> {
  static constexpr float w[MAX_R + 1] = {-2.72222222F, 1.5F, -1.5e-1F, 1.11111111e-2F, 0.0F, 0.0F, 0.0F};
};
template <> struct Weights<4> {
  static constexpr float w[MAX_R + 1] = {-2.84722222F, 1.6F, -2.0e-1F, 2.53968254e-2F, -1.78571429e-3F, 0.0F, 0.0F};
};
template <> struct Weights<5> {
  static constexpr float w[MAX_R + 1] = {-2.92722222F, 1.66666667F,     -2.38095238e-1F, 3.96825397e-2F,
                                         -4.96031746e-3F, 3.17460317e-4F, 0.0F};
};
template <> struct Weights<6> {
  static constexpr float w[MAX_R + 1] = {-2.98277778F,    1.71428571F,    -2.67857143e-1F, 5.29100529e-2F,
                                         -8.92857143e-3F, 1.03896104e-3F, -6.01250601e-5F};
};

// DAMPED: the update with an absorbing sponge of fdtd_forward.hip (damping field of the shape of m):
//     u+ = dt^2 ( lap - (-2 r1 u + r1 u-) m + r0 damp u ) / (m + dt damp)
struct DampTerm {
  const float *damp;
  float dt, r0;   // r0 = 1 / dt
};

template <int R, bool DAMPED = false>
__global__ __launch_bounds__(256) void stencil_order_kernel(StencilArgs a, DampTerm dm) {
  constexpr int H = 2 * R;   // padding = space order
  const int z = a.z_m + blockIdx.x * 64 + threadIdx.x;
  const int y = a.y_m + blockIdx.y * 4 + threadIdx.y;
  const int x = a.x_m + blockIdx.z;
  if (z > a.z_M || y > a.y_M) return;
  const long sx = a.sx, sy = a.sy;
  const long c = (long)(x + H) * sx + (long)(y + H) * sy + (z + H);
  const float *__restrict__ u0 = a.u0;
  const float uc = u0[c];
  const float r5 = Weights<R>::w[0] * uc;
  float dxx = r5, dyy = r5, dzz = r5;
#pragma unroll
  for (int k = R; k >= 1; --k) {
    dxx = dxx + Weights<R>::w[k] * (u0[c - k * sx] + u0[c + k * sx]);
    dyy = dyy + Weights<R>::w[k] * (u0[c - k * sy] + u0[c + k * sy]);
    dzz = dzz + Weights<R>::w[k] * (u0[c - k] + u0[c + k]);
  }
  const float mc = a.m[c];
  const float neg2r1 = -2.0F * a.r1;
  if (DAMPED) {
    const float dc = dm.damp[c];
    a.u2[c] = a.dt2 * (a.r2 * dxx + a.r3 * dyy + a.r4 * dzz - (neg2r1 * uc + a.r1 * a.u1[c]) * mc + dm.r0 * dc * uc) /
              (mc + dm.dt * dc);
  } else {
    a.u2[c] = a.dt2 * (a.r2 * dxx + a.r3 * dyy + a.r4 * dzz - (neg2r1 * uc + a.r1 * a.u1[c]) * mc) / mc;
  }
}

// ---------------------------------------------------------------------------------
// The marching sweep.  Workgroup = 8 waves; tile = TY = 8 RW rows of 256 cells (4 cells per lane, RW rows
// per wave); x is cut into chunks, one workgroup per (tile, chunk).
//
// Registers of a lane, per row: ring[0..2R] = its 4 cells of the planes p-R .. p+R.  LDS: the plane p on
// the TY + 2R rows around the tile, 256 + 2 ZC cells each (ZC = 4 or 8 >= R cells beyond the tile on either
// side), double-buffered, one barrier per plane.  Per plane a lane issues the loads of the next iteration
// first (its cells of plane p+R+1, of u(t-1) and m of plane p+1, and its share of the rows and cells around
// the tile of plane p+1), then evaluates plane p, then hands plane p+1 to LDS.
//
// Cells beyond z_M or y_M: lanes and rows there work on clamped addresses and store nothing; the lanes
// within R cells behind z_M hold the padding the last cells need.  16-byte accesses are issued at 4-byte
// alignment (z_m is arbitrary), which global memory takes.
// ---------------------------------------------------------------------------------
// Loads of the coming plane are issued at raised wave priority (s_setprio; 0 = never): the wave of a SIMD that has
// reached them does not take turns with the other waves' arithmetic.  Order 8 at 512^3 316 against 312 Gcells/s, at 1024^3
// 343 against 337; order 12 at 1024^3 333.5 against 325.5.
#ifndef FDTD_ORDER_PRIO
#define FDTD_ORDER_PRIO 3
#endif
struct __attribute__((packed, aligned(4))) Quad { float v[4]; };
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // a float4 at 4-byte alignment
__device__ __forceinline__ f32x4 ldq(const float *p) {
  const Quad q = *(const Quad *)p;
  return f32x4{q.v[0], q.v[1], q.v[2], q.v[3]};
}
__device__ __forceinline__ int xcd_contiguous(int b, int nblocks) {   // workgroups b, b+8, .. share an XCD
  const int q = nblocks >> 3, r = nblocks & 7, xcd = b & 7, j = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// FDTD_WG_CLOCKS (a measurement build, `make clocks`): a time stamp per workgroup of the marching sweeps at its
// start and at its end, read back through fdtd_debug_wg_clocks_orders (tools/wg_clocks.py --order).
#ifdef FDTD_WG_CLOCKS
__device__ unsigned long long g_order_clock[3][4096];
struct WgClock {
  int tile;
  __device__ explicit WgClock(int t) : tile(t) {
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_order_clock[0][blockIdx.x] = wall_clock64();
  }
  __device__ ~WgClock() {
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
      g_order_clock[1][blockIdx.x] = wall_clock64();
      g_order_clock[2][blockIdx.x] = (unsigned long long)tile;
    }
  }
};
#define WG_CLOCK(t) WgClock wg_clock_(t)
#else
#define WG_CLOCK(t)
#endif

// LPR = lanes per row: 64 (tiles of 256 cells) or 32 (tiles of 128 cells: a wave then carries two rows per
// row slot and the tile has twice as many rows), for rows that fill multiples of 128 cells better.
template <int R, int RW, int D, bool NT, bool DAMPED = false, int LPR = 64>
__global__ __launch_bounds__(512) void stencil_order_march_kernel(StencilArgs a, int xchunk, int nyt, int nzt,
                                                                  DampTerm dm) {
  static_assert(D >= 1 && D <= R, "planes up to x_M + R + D must exist");
  static_assert(LPR == 64 || LPR == 32, "whole or half waves per row");
  constexpr int RPS = 64 / LPR;               // rows per row slot of a wave
  constexpr int TZ = 4 * LPR;                 // cells of a tile row
  constexpr int H = 2 * R, TY = 8 * RW * RPS, NROWS = TY + 2 * R;
  constexpr int ZF = R > 4 ? 2 : 1;           // float4s beyond the tile per side
  constexpr int ROWF4 = LPR + 2 * ZF;
  constexpr int NPIECE = 2 * R * LPR + TY * 2 * ZF;   // float4s around the tile, per plane
  constexpr int NPT = (NPIECE + 511) / 512;             // ... per thread
  __shared__ f32x4 plane[2][NROWS * ROWF4];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane % LPR, sub = lane / LPR;   // float4 of the row, row of the row slot
  const int t = xcd_contiguous(blockIdx.x, gridDim.x);
  WG_CLOCK(t);
  const int zt = t % nzt, yt = (t / nzt) % nyt, xc = t / (nzt * nyt);
  const int zb = a.z_m + zt * TZ, yb = a.y_m + yt * TY;
  const int x_begin = a.x_m + xc * xchunk, x_end = min(x_begin + xchunk - 1, a.x_M);
  if (x_begin > x_end) return;
  const int sy = (int)a.sy;
  const long sx = a.sx;
  // a column of 4 cells starting at z: its address is used as it is while z <= z_M + R (padding included)
  auto zaddr = [&](int z) { return (z <= a.z_M + R ? z : zb) + H; };
  auto yaddr = [&](int y) { return min(y, a.y_M + R) + H; };

  // own cells
  const int z0 = zb + 4 * col;
  const int nvalid = min(max(a.z_M + 1 - z0, 0), 4);
  int own[RW];          // offset inside a plane
  bool row_valid[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int y = yb + (wave * RW + r) * RPS + sub;
    row_valid[r] = y <= a.y_M;
    own[r] = yaddr(y) * sy + zaddr(z0);
  }
  // share of the cells around the tile
  int poff[NPT], pslot[NPT];
  bool pvalid[NPT];
#pragma unroll
  for (int q = 0; q < NPT; ++q) {
    const int i = (int)threadIdx.x + 512 * q;
    pvalid[q] = i < NPIECE;
    int y, z, slot;
    if (i < 2 * R * LPR) {           // rows below and above the tile
      const int hr = i / LPR, c = i % LPR;
      y = hr < R ? yb - R + hr : yb + TY + (hr - R);
      z = zb + 4 * c;
      slot = (hr < R ? hr : TY + hr) * ROWF4 + ZF + c;
    } else {                         // cells before and behind the tile, own rows
      const int j = i - 2 * R * LPR, row = (j / (2 * ZF)) % TY, k = j % (2 * ZF);
      y = yb + row;
      z = k < ZF ? zb - 4 * ZF + 4 * k : zb + TZ + 4 * (k - ZF);
      slot = (R + row) * ROWF4 + (k < ZF ? k : LPR + k);
    }
    poff[q] = yaddr(y) * sy + zaddr(z);
    pslot[q] = slot;
  }

  const float *u0p = a.u0 + (long)(x_begin + H) * sx;    // plane p of u(t)
  const float *u1p = a.u1 + (long)(x_begin + H) * sx;
  const float *mp = a.m + (long)(x_begin + H) * sx;
  float *u2p = a.u2 + (long)(x_begin + H) * sx;
  const float *dp = DAMPED ? dm.damp + (long)(x_begin + H) * sx : nullptr;

  // ring[k]: the lane's cells of plane p - R + k; uq[j], mq[j]: u(t-1) and m of plane p + j; aq[j]: the
  // thread's share of the cells around the tile, plane p + 1 + j.  Everything up to D planes ahead of what
  // the current plane needs is in registers or in flight.
  f32x4 ring[RW][2 * R + D], uq[RW][D], mq[RW][D], aq[NPT][D], dq[RW][DAMPED ? D : 1];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
#pragma unroll
    for (int k = 0; k < 2 * R + D; ++k) ring[r][k] = ldq(u0p + (long)(k - R) * sx + own[r]);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      uq[r][j] = ldq(u1p + (long)j * sx + own[r]);
      mq[r][j] = ldq(mp + (long)j * sx + own[r]);
      if (DAMPED) dq[r][j] = ldq(dp + (long)j * sx + own[r]);
    }
  }
#pragma unroll
  for (int q = 0; q < NPT; ++q)
#pragma unroll
    for (int j = 0; j + 1 < D; ++j) aq[q][j] = ldq(u0p + (long)(j + 1) * sx + poff[q]);
  {
    f32x4 *buf = plane[x_begin & 1];
#pragma unroll
    for (int r = 0; r < RW; ++r) buf[(R + (wave * RW + r) * RPS + sub) * ROWF4 + ZF + col] = ring[r][R];
#pragma unroll
    for (int q = 0; q < NPT; ++q)
      if (pvalid[q]) buf[pslot[q]] = ldq(u0p + poff[q]);
  }
  __syncthreads();

  const float neg2r1 = -2.0F * a.r1;
  for (int p = x_begin; p <= x_end; ++p) {
    const f32x4 *cur = plane[p & 1];
    f32x4 *nxt = plane[(p & 1) ^ 1];
    // loads for D planes ahead (the planes up to x_M + R + D exist in the arrays: D <= R)
    f32x4 rn[RW], un[RW], mn[RW], dn[RW];
#if FDTD_ORDER_PRIO
    __builtin_amdgcn_s_setprio(FDTD_ORDER_PRIO);
#endif
#pragma unroll
    for (int q = 0; q < NPT; ++q) aq[q][D - 1] = ldq(u0p + (long)D * sx + poff[q]);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      rn[r] = ldq(u0p + (long)(R + D) * sx + own[r]);
      un[r] = ldq(u1p + (long)D * sx + own[r]);
      mn[r] = ldq(mp + (long)D * sx + own[r]);
      if (DAMPED) dn[r] = ldq(dp + (long)D * sx + own[r]);
    }
#if FDTD_ORDER_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int row = R + (wave * RW + r) * RPS + sub;
      const f32x4 uc = ring[r][R];
      const f32x4 r5 = Weights<R>::w[0] * uc;
      f32x4 dxx = r5, dyy = r5, dzz = r5;
      // the row of plane p: ZF float4s before the lane's cells, the cells, ZF float4s behind
      float win[4 + 8 * ZF];
#pragma unroll
      for (int j = 0; j < ZF; ++j) {
        const f32x4 l = cur[row * ROWF4 + col + j], h = cur[row * ROWF4 + ZF + col + 1 + j];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          win[4 * j + c] = l[c];
          win[4 * ZF + 4 + 4 * j + c] = h[c];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) win[4 * ZF + c] = uc[c];
#pragma unroll
      for (int k = R; k >= 1; --k) {
        const float w = Weights<R>::w[k];
        dxx = dxx + w * (ring[r][R - k] + ring[r][R + k]);
        dyy = dyy + w * (cur[(row - k) * ROWF4 + ZF + col] + cur[(row + k) * ROWF4 + ZF + col]);
        const f32x4 zm = f32x4{win[4 * ZF - k], win[4 * ZF + 1 - k], win[4 * ZF + 2 - k], win[4 * ZF + 3 - k]};
        const f32x4 zp = f32x4{win[4 * ZF + k], win[4 * ZF + 1 + k], win[4 * ZF + 2 + k], win[4 * ZF + 3 + k]};
        dzz = dzz + w * (zm + zp);
      }
      const f32x4 mc = mq[r][0];
      f32x4 o;
      if (DAMPED) {
        const f32x4 dc = dq[r][0];
        const f32x4 num =
            a.dt2 * (a.r2 * dxx + a.r3 * dyy + a.r4 * dzz - (neg2r1 * uc + a.r1 * uq[r][0]) * mc + dm.r0 * dc * uc);
        const f32x4 den = mc + dm.dt * dc;
        o = f32x4{num.x / den.x, num.y / den.y, num.z / den.z, num.w / den.w};
      } else {
        const f32x4 num = a.dt2 * (a.r2 * dxx + a.r3 * dyy + a.r4 * dzz - (neg2r1 * uc + a.r1 * uq[r][0]) * mc);
        o = f32x4{num.x / mc.x, num.y / mc.y, num.z / mc.z, num.w / mc.w};
      }
      if (row_valid[r]) {
        float *dst = u2p + own[r];
        if (nvalid == 4) {
          Quad q4;
          q4.v[0] = o.x; q4.v[1] = o.y; q4.v[2] = o.z; q4.v[3] = o.w;
          if (NT) __builtin_nontemporal_store(o, (f32x4_a4 *)dst);
          else *(Quad *)dst = q4;
        } else {
          if (nvalid > 0) dst[0] = o.x;
          if (nvalid > 1) dst[1] = o.y;
          if (nvalid > 2) dst[2] = o.z;
        }
      }
    }
    // plane p+1 into the other buffer, rings forward
#pragma unroll
    for (int r = 0; r < RW; ++r) nxt[(R + (wave * RW + r) * RPS + sub) * ROWF4 + ZF + col] = ring[r][R + 1];
#pragma unroll
    for (int q = 0; q < NPT; ++q) {
      if (pvalid[q]) nxt[pslot[q]] = aq[q][0];
#pragma unroll
      for (int j = 0; j + 1 < D; ++j) aq[q][j] = aq[q][j + 1];
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
#pragma unroll
      for (int k = 0; k + 1 < 2 * R + D; ++k) ring[r][k] = ring[r][k + 1];
      ring[r][2 * R + D - 1] = rn[r];
#pragma unroll
      for (int j = 0; j + 1 < D; ++j) {
        uq[r][j] = uq[r][j + 1];
        mq[r][j] = mq[r][j + 1];
      }
      uq[r][D - 1] = un[r];
      mq[r][D - 1] = mn[r];
      if (DAMPED) {
#pragma unroll
        for (int j = 0; j + 1 < D; ++j) dq[r][j] = dq[r][j + 1];
        dq[r][D - 1] = dn[r];
      }
    }
    u0p += sx; u1p += sx; mp += sx; u2p += sx;
    if (DAMPED) dp += sx;
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------
// The marching sweep of radius 2, 3 and 4 in the shape of the order-4 marching kernel
// (stencil_march_lds_kernel, fdtd_kernels.hip), for rows of whole float4s (z_m and the z extent
// multiples of 4).  A wave owns RY = 2 rows x 256 cells (one float4 per lane) and marches along x;
// 8 waves are stacked in y.
//   x-neighbours: a register column of the 2R+1 planes p-R .. p+R of the owned rows.
//   z-neighbours: the float4 of the previous / next lane by DPP wavefront shifts (4 cells reach up to
//     R = 4); lane 0 / the last owned lane take them from a 16-byte edge load.
//   y-neighbours: the waves publish the centre plane of their rows to LDS (double-buffered, one
//     barrier per plane); the R rows either side of the 16-row tile are fetched by the waves at the
//     ends of the stack, at the plane step at which the neighbouring tile fetches them as its own
//     (p + R), and wait in a queue of R planes until their plane is the centre: one of the two
//     requests hits in L2.
// gfx950 retires loads and stores through one in-order counter, so an iteration first issues every
// load the next iteration needs into one of two register sets (the loop is unrolled twice, the sets
// alternate), then computes from the other set, then stores; no memory instruction sits under a
// divergent branch.  u(t-1), m, the damping field and the result carry non-temporal hints.
// One workgroup per CU, XCD-contiguous tile map, launches of whole rounds of workgroups (below).
// Same arithmetic and operand order as the kernels above: same bits.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_prev_lane(float edge, float v) {   // lane i <- lane i-1; lane 0 keeps `edge`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v),
                                                               0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_next_lane(float edge, float v) {   // lane i <- lane i+1; lane 63 keeps `edge`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v),
                                                               0x130, 0xf, 0xf, false));
}
template <bool NT>
__device__ __forceinline__ f32x4 ldq_stream(const float *p) {
  if constexpr (NT) return __builtin_nontemporal_load((const f32x4_a4 *)p);
  else return ldq(p);
}

template <bool DAMPED>
struct ColumnSet {            // what one plane step loads ahead
  f32x4 col[2];               // u(t), plane p+R (relative to the plane the set is current for)
  f32x4 prev[2], med[2];      // u(t-1), m of the current plane
  f32x4 dmp[DAMPED ? 2 : 1];  // damping field of the current plane
  f32x4 edge[2], edge_r[2];   // z-edge cells of the current plane: (z-4 .. z-1); full tiles: lane 63 holds (z+4 .. z+7)
  f32x4 halo;                 // one row outside the tile, plane p+R (loader waves)
};

template <int R, bool NT, bool FULL, bool DAMPED>
__global__ __launch_bounds__(512) void stencil_column_march_kernel(StencilArgs a, int xchunk, int nzt, int nyt,
                                                                   DampTerm dm) {
  static_assert(R >= 2 && R <= 4, "the neighbouring lane's float4 covers a radius of up to 4");
  constexpr int RY = 2, NW = 8, TY = RY * NW, H = 2 * R, TZ = 256;
  __shared__ f32x4 plane[2][TY + 2 * R][64];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = xcd_contiguous(blockIdx.x, gridDim.x);
  WG_CLOCK(t);
  const int zt = t % nzt, yt = (t / nzt) % nyt, xc = t / (nzt * nyt);
  const int yb = a.y_m + yt * TY, y0 = yb + wave * RY;
  const int zfirst = a.z_m + zt * TZ + lane * 4;
  // the z extent is a multiple of 4: a lane owns all of its cells or none.  Partial tiles: lanes behind the
  // last owned one repeat its address, and a second edge load gives the last owned lane its right-hand cells.
  const bool owned = FULL ? true : (zfirst <= a.z_M);
  const int zload = FULL ? zfirst : min(zfirst, a.z_M - 3);
  const bool last_owned = (lane == 63) || (!FULL && owned && zfirst + 4 > a.z_M);
  const int x_begin = a.x_m + xc * xchunk, x_end = min(x_begin + xchunk - 1, a.x_M);
  if (x_begin > x_end) return;
  const unsigned sy = (unsigned)a.sy;
  const long sx = a.sx;

  unsigned voff[RY], veoff[RY];   // lane offsets (cells) inside a plane; rows past y_M + R repeat a readable row
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    const int y = FULL ? (y0 + r) : min(y0 + r, a.y_M + R);
    voff[r] = (unsigned)(y + H) * sy + (unsigned)(zload + H);
    veoff[r] = (FULL && lane == 63) ? voff[r] + 4 : voff[r] - 4;   // (z+4 .. z+7) or (z-4 .. z-1)
  }
  // the waves at the ends of the stack also fetch one row outside the tile each
  const bool halo_loader = (wave < R) || (wave >= NW - R);
  const int halo_y = (wave < R) ? yb - R + wave
                                : ((wave >= NW - R) ? min(yb + TY + (wave - (NW - R)), a.y_M + R) : min(y0, a.y_M + R));
  const int halo_slot = (wave < R) ? wave : TY + R + (wave - (NW - R));
  const unsigned vhoff = (unsigned)(halo_y + H) * sy + (unsigned)(zload + H);

  const float *pu0 = a.u0 + (long)(x_begin + H) * sx;
  const float *pu1 = a.u1 + (long)(x_begin + H) * sx;
  const float *pm = a.m + (long)(x_begin + H) * sx;
  const float *pd = DAMPED ? dm.damp + (long)(x_begin + H) * sx : nullptr;
  float *pu2 = a.u2 + (long)(x_begin + H) * sx;

  f32x4 col[2 * R][RY];   // planes p-R .. p+R-1 of the owned rows
  f32x4 hq[R];            // the outside row at planes p .. p+R-1
  ColumnSet<DAMPED> sa, sb;
#pragma unroll
  for (int r = 0; r < RY; ++r) {
#pragma unroll
    for (int p = 0; p < 2 * R; ++p) col[p][r] = ldq(pu0 + (long)(p - R) * sx + voff[r]);
    sa.col[r] = ldq(pu0 + (long)R * sx + voff[r]);
    sa.prev[r] = ldq_stream<NT>(pu1 + voff[r]);
    sa.med[r] = ldq_stream<NT>(pm + voff[r]);
    if (DAMPED) sa.dmp[r] = ldq_stream<NT>(pd + voff[r]);
    sa.edge[r] = ldq(pu0 + veoff[r]);
    if (!FULL) sa.edge_r[r] = ldq(pu0 + voff[r] + 4);
  }
  sa.halo = f32x4{0, 0, 0, 0};
  sb.halo = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < R; ++j) hq[j] = f32x4{0, 0, 0, 0};
  if (halo_loader) {   // wave-uniform
#pragma unroll
    for (int j = 0; j < R; ++j) hq[j] = ldq(pu0 + (long)j * sx + vhoff);
    sa.halo = ldq(pu0 + (long)R * sx + vhoff);
  }
  const float neg2r1 = -2.0F * a.r1;

  // One plane.  `cur` was loaded during the previous iteration; `nxt` is filled now.
  auto iteration = [&](int x, int buf, const ColumnSet<DAMPED> &cur, ColumnSet<DAMPED> &nxt) __attribute__((always_inline)) {
    // ---- loads for plane x+1 (repeated on plane x at the end of the chunk) ----
    const long step = x < x_end ? sx : 0;
    const float *q0 = pu0 + step, *q1 = pu1 + step, *qm = pm + step, *qd = DAMPED ? pd + step : nullptr;
#if FDTD_ORDER_PRIO
    __builtin_amdgcn_s_setprio(FDTD_ORDER_PRIO);
#endif
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      nxt.col[r] = ldq(q0 + (long)R * sx + voff[r]);
      nxt.prev[r] = ldq_stream<NT>(q1 + voff[r]);
      nxt.med[r] = ldq_stream<NT>(qm + voff[r]);
      if (DAMPED) nxt.dmp[r] = ldq_stream<NT>(qd + voff[r]);
      nxt.edge[r] = ldq(q0 + veoff[r]);
      if (!FULL) nxt.edge_r[r] = ldq(q0 + voff[r] + 4);
    }
    if (halo_loader) nxt.halo = ldq(q0 + (long)R * sx + vhoff);
#if FDTD_ORDER_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif

    // ---- publish the centre plane of the owned rows (and the outside row) ----
#pragma unroll
    for (int r = 0; r < RY; ++r) plane[buf][wave * RY + r + R][lane] = col[R][r];
    if (halo_loader) plane[buf][halo_slot][lane] = hq[0];
    __syncthreads();
    f32x4 cen[RY + 2 * R];
#pragma unroll
    for (int j = 0; j < RY + 2 * R; ++j)
      cen[j] = (j >= R && j < R + RY) ? col[R][j - R] : plane[buf][wave * RY + j][lane];

    // ---- compute and store plane x ----
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      const f32x4 uc = cen[r + R];
      // the cells before and behind the lane's four
      f32x4 lo = f32x4{dpp_prev_lane(cur.edge[r].x, uc.x), dpp_prev_lane(cur.edge[r].y, uc.y),
                       dpp_prev_lane(cur.edge[r].z, uc.z), dpp_prev_lane(cur.edge[r].w, uc.w)};
      const f32x4 er = FULL ? cur.edge[r] : cur.edge_r[r];
      f32x4 hi = f32x4{dpp_next_lane(er.x, uc.x), dpp_next_lane(er.y, uc.y), dpp_next_lane(er.z, uc.z),
                       dpp_next_lane(er.w, uc.w)};
      if (!FULL) hi = last_owned ? er : hi;
      const float win[12] = {lo.x, lo.y, lo.z, lo.w, uc.x, uc.y, uc.z, uc.w, hi.x, hi.y, hi.z, hi.w};
      const f32x4 r5 = Weights<R>::w[0] * uc;
      f32x4 dxx = r5, dyy = r5, dzz = r5;
#pragma unroll
      for (int k = R; k >= 1; --k) {
        const float w = Weights<R>::w[k];
        const f32x4 xp = k == R ? cur.col[r] : col[R + k][r];
        dxx = dxx + w * (col[R - k][r] + xp);
        dyy = dyy + w * (cen[r + R - k] + cen[r + R + k]);
        const f32x4 zm = f32x4{win[4 - k], win[5 - k], win[6 - k], win[7 - k]};
        const f32x4 zp = f32x4{win[4 + k], win[5 + k], win[6 + k], win[7 + k]};
        dzz = dzz + w * (zm + zp);
      }
      const f32x4 mc = cur.med[r];
      f32x4 o;
      if (DAMPED) {
        const f32x4 dc = cur.dmp[r];
        const f32x4 num =
            a.dt2 * (a.r2 * dxx + a.r3 * dyy + a.r4 * dzz - (neg2r1 * uc + a.r1 * cur.prev[r]) * mc + dm.r0 * dc * uc);
        const f32x4 den = mc + dm.dt * dc;
        o = f32x4{num.x / den.x, num.y / den.y, num.z / den.z, num.w / den.w};
      } else {
        const f32x4 num = a.dt2 * (a.r2 * dxx + a.r3 * dyy + a.r4 * dzz - (neg2r1 * uc + a.r1 * cur.prev[r]) * mc);
        o = f32x4{num.x / mc.x, num.y / mc.y, num.z / mc.z, num.w / mc.w};
      }
      if (FULL || (owned && y0 + r <= a.y_M)) {
        if (NT) __builtin_nontemporal_store(o, (f32x4_a4 *)(pu2 + voff[r]));
        else *(f32x4_a4 *)(pu2 + voff[r]) = o;
      }
    }

    // ---- advance: shift the older planes, step the plane pointers ----
#pragma unroll
    for (int r = 0; r < RY; ++r) {
#pragma unroll
      for (int p = 0; p + 1 < 2 * R; ++p) col[p][r] = col[p + 1][r];
      col[2 * R - 1][r] = cur.col[r];
    }
#pragma unroll
    for (int j = 0; j + 1 < R; ++j) hq[j] = hq[j + 1];
    hq[R - 1] = cur.halo;
    pu0 += sx; pu1 += sx; pm += sx; pu2 += sx;
    if (DAMPED) pd += sx;
  };

  for (int x = x_begin; x <= x_end; x += 2) {
    iteration(x, 0, sa, sb);
    if (x + 1 > x_end) break;
    iteration(x + 1, 1, sb, sa);
  }
}

int g_space_order = 0;   // 0: not set yet (FDTD_STENCIL_ORDER, else 4)
// The order of the call in progress on this thread (the *_order / Kernel_HIP_SpaceOrder entry points): it
// takes precedence over the process-wide setting, so those entry points neither read nor change it.
thread_local int t_call_order = 0;

bool order_supported(int order) { return order >= 4 && order <= 2 * MAX_R && order % 2 == 0; }

int env_int_orders(const char *name, int fallback) {
  const char *v = getenv(name);
  return v ? atoi(v) : fallback;
}

// Workgroups of the marching sweep that one compute unit holds at a time (registers and LDS decide).
template <int R, int RW, int D, bool NT, bool DAMPED, int LPR>
int march_workgroups_per_cu() {
  static int n = 0;
  if (n == 0) {
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, stencil_order_march_kernel<R, RW, D, NT, DAMPED, LPR>, 512, 0) != hipSuccess ||
        v < 1)
      v = 1;
    n = v;
  }
  return n;
}

template <int R, int RW, int D, bool NT, bool DAMPED = false, int LPR = 64>
void launch_order_march(const StencilArgs &a, hipStream_t s, DampTerm dm = DampTerm{nullptr, 0.0F, 0.0F}) {
  constexpr int TY = 8 * RW * (64 / LPR), TZ = 4 * LPR;
  const int ex = a.x_M - a.x_m + 1, ey = a.y_M - a.y_m + 1, ez = a.z_M - a.z_m + 1;
  const int nyt = (ey + TY - 1) / TY, nzt = (ez + TZ - 1) / TZ;
  const long tiles = (long)nyt * nzt;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long slots = (long)cus * march_workgroups_per_cu<R, RW, D, NT, DAMPED, LPR>();
  // x-chunks: the device works through ceil(workgroups / slots) rounds of (chunk length + lead-in), where
  // the lead-in — the 2R planes a chunk loads before its first result, without arithmetic — counts as R
  // planes.  Take the number of chunks that minimises the product, with chunks of at most 128 planes (64
  // where a plane has 100 tiles or more): workgroups that march further drift apart and stop finding their
  // neighbours' rows in L2.  Where the tiles of a plane fill rounds of resident workgroups exactly, every round
  // is a launch of its own, so that the workgroups of a launch start together (1024^3 at order 8: 333 against
  // 298 Gcells/s in one launch); where they do not, one launch fills the partial rounds by itself (768^3, 144
  // tiles against 512 resident workgroups: 290 against 234 with a launch per round).
  const int lcap = env_int_orders("FDTD_ORDER_LCAP", tiles >= 100 ? 64 : 128);
  long nchunks = 1;
  double best = 0.0;
  bool have = false;
  for (long n = 1; n <= ex && n <= 4096; ++n) {
    const long len = (ex + n - 1) / n;
    if (n > 1 && len < 4) break;
    if (len > lcap && len > 4) continue;
    const long rounds = (tiles * ((ex + len - 1) / len) + slots - 1) / slots;
    const double cost = (double)rounds * (double)(len + R + 1);
    if (!have || cost < best) {
      best = cost;
      nchunks = n;
      have = true;
    }
  }
  const int forced = env_int_orders("FDTD_ORDER_XCHUNK", 0);
  const int xchunk = forced > 0 ? forced : (int)((ex + nchunks - 1) / nchunks);
  nchunks = (ex + xchunk - 1) / xchunk;
  long per_launch = nchunks;         // x-chunks per launch
  if (tiles <= slots && slots % tiles == 0 && !env_int_orders("FDTD_ORDER_ONE_LAUNCH", 0)) per_launch = slots / tiles;
  StencilArgs al = a;
  for (long c0 = 0; c0 < nchunks; c0 += per_launch) {
    const long nc = nchunks - c0 < per_launch ? nchunks - c0 : per_launch;
    al.x_m = a.x_m + (int)(c0 * xchunk);
    al.x_M = (int)(al.x_m + nc * xchunk - 1 < a.x_M ? al.x_m + nc * xchunk - 1 : a.x_M);
    hipLaunchKernelGGL((stencil_order_march_kernel<R, RW, D, NT, DAMPED, LPR>), dim3((unsigned)(tiles * nc)), dim3(512), 0, s,
                       al, xchunk, nyt, nzt, dm);
  }
}

// Launch shape of stencil_column_march_kernel: one workgroup per CU at a time, so a launch runs in rounds of
// 256 workgroups and is fastest when its workgroups advance plane by plane together; marches longer than
// ~128 planes drift apart, so long grids are cut into several launches (as for the order-4 marching kernel).
// Among the candidates the score prefers full rounds, then longer chunks (each chunk loads 2R planes before
// its first result).
template <int R, bool NT, bool DAMPED>
void launch_column_march(const StencilArgs &a0, hipStream_t s, DampTerm dm) {
  constexpr int TY = 16, TZ = 256;
  const int ex = a0.x_M - a0.x_m + 1, ey = a0.y_M - a0.y_m + 1, ez = a0.z_M - a0.z_m + 1;
  const int nzt = (ez + TZ - 1) / TZ, nyt = (ey + TY - 1) / TY;
  const long tiles = (long)nzt * nyt, cus = 256;
  const int lcap = env_int_orders("FDTD_ORDER_LCAP", tiles >= 100 ? 64 : 128);
  int xchunk = ex < lcap ? ex : lcap, chunks = 1;
  const int forced = env_int_orders("FDTD_ORDER_XCHUNK", 0);
  if (forced > 0) {
    xchunk = forced > ex ? ex : forced;
    chunks = (ex + xchunk - 1) / xchunk;
  } else {
    double best = -1.0;
    for (int c = 1; c <= 64; ++c) {
      int xc = (ex + c - 1) / c;
      if (xc > lcap) xc = lcap;
      if (xc < 8 && c > 1) break;
      const long blocks = (long)c * tiles, rounds = (blocks + cus - 1) / cus;
      const double fill = (double)blocks / (double)(rounds * cus);
      const double score = fill * (double)xc / (double)(xc + 2 * R + 2);
      if (score > best + 1e-9) { best = score; xchunk = xc; chunks = c; }
    }
    const long per = (long)chunks * xchunk, launches = (ex + per - 1) / per, planes = (ex + launches - 1) / launches;
    xchunk = (int)((planes + chunks - 1) / chunks);
    if (xchunk < 1) xchunk = 1;
  }
  const bool full = (ez % TZ == 0) && (ey % TY == 0);
  StencilArgs a = a0;
  const long per_launch = (long)chunks * xchunk;
  for (long x0 = a0.x_m; x0 <= a0.x_M; x0 += per_launch) {
    a.x_m = (int)x0;
    a.x_M = (int)((x0 + per_launch - 1 < a0.x_M) ? x0 + per_launch - 1 : a0.x_M);
    const int nxc = (a.x_M - a.x_m + xchunk) / xchunk;
    const dim3 grid((unsigned)(tiles * nxc));
    if (full) hipLaunchKernelGGL((stencil_column_march_kernel<R, NT, true, DAMPED>), grid, dim3(512), 0, s, a, xchunk, nzt, nyt, dm);
    else hipLaunchKernelGGL((stencil_column_march_kernel<R, NT, false, DAMPED>), grid, dim3(512), 0, s, a, xchunk, nzt, nyt, dm);
  }
}

// The marching sweep takes boxes that stay inside the interior of the arrays (its 16-byte accesses reach up
// to 8 cells beyond the box in z) and are large enough to fill the device; FDTD_ORDER_MARCH = 0 keeps
// every box on the one-cell-per-thread kernel, 2 sends every box that fits.
bool order_march_takes(const StencilArgs &a, int H) {
  const int mode = env_int_orders("FDTD_ORDER_MARCH", 1);
  if (mode == 0) return false;
  if (a.x_m < 0 || a.y_m < 0 || a.z_m < 0 || a.x_M > a.nxp - 2 * H - 1 || a.y_M > a.nyp - 2 * H - 1 ||
      a.z_M > a.nzp - 2 * H - 1)
    return false;
  if ((long)a.nyp * a.nzp >= (1L << 30)) return false;   // offsets inside a plane are ints
  // A lane's 16-byte access starts at most R cells behind z_M and must end inside the row (H cells of padding):
  // R + 3 <= H holds from order 6 on; at order 4 a row that reaches the end of the interior with a z extent
  // of 4k + 3 cells would touch the first float of the next row, so such boxes stay on the other kernel.
  if (H == 4 && a.z_M == a.nzp - 2 * H - 1 && (a.z_M - a.z_m + 1) % 4 == 3) return false;
  if (mode == 2) return true;
  const long ex = a.x_M - a.x_m + 1, ey = a.y_M - a.y_m + 1, ez = a.z_M - a.z_m + 1;
  // measured on MI355X: from 256^3 cells on the marching sweep is the faster one at every order; below, the
  // one-cell-per-thread sweep finds its operands in the caches
  // (rows from 96 cells on: 512 x 512 x 100 runs 251 against 129 Gcells/s at order 6, 1024 x 1024 x 96 256 against 120)
  return ez >= 96 && ex * ey * ez >= (long)env_int_orders("FDTD_ORDER_MARCH_MIN_CELLS", 12000000);
}

}  // namespace

#ifdef FDTD_WG_CLOCKS
extern "C" int fdtd_debug_wg_clocks_orders(unsigned long long *out) {   // 3 x 4096 values of the last launch
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_order_clock), sizeof(unsigned long long) * 3 * 4096);
}
#endif

extern "C" int fdtd_hip_set_space_order(int order) {
  if (!order_supported(order)) return (int)hipErrorInvalidValue;
  g_space_order = order;
  return 0;
}

int fdtd_space_order_for_this_call(int order) {
  const int before = t_call_order;
  t_call_order = order;
  return before;
}
bool fdtd_space_order_supported(int order) { return order_supported(order); }

extern "C" int fdtd_hip_get_space_order(void) {
  if (t_call_order) return t_call_order;
  if (g_space_order == 0) {
    const char *e = getenv("FDTD_STENCIL_ORDER");
    const int v = e ? atoi(e) : 4;
    g_space_order = order_supported(v) ? v : 4;
  }
  return g_space_order;
}

// One sweep of any supported order; damp != nullptr: with the damping term.
static int sweep_order(const struct fdtd_geom *g, const float *m, const float *damp, const float *u0, const float *u1,
                       float *u2, int order, void *stream) {
  if (!g || !m || !u0 || !u1 || !u2 || !order_supported(order)) return (int)hipErrorInvalidValue;
  const int R = order / 2, H = order;
  const int ex = g->x_M - g->x_m + 1, ey = g->y_M - g->y_m + 1, ez = g->z_M - g->z_m + 1;
  if (ex <= 0 || ey <= 0 || ez <= 0) return 0;
  // every stencil read (radius R around an updated cell) must stay inside the arrays
  if (g->x_m + H - R < 0 || g->x_M + H + R > g->nxp - 1 || g->y_m + H - R < 0 || g->y_M + H + R > g->nyp - 1 ||
      g->z_m + H - R < 0 || g->z_M + H + R > g->nzp - 1)
    return (int)hipErrorInvalidValue;
  StencilArgs a;
  a.m = m; a.u0 = u0; a.u1 = u1; a.u2 = u2;
  a.sx = (long)g->nyp * g->nzp; a.sy = g->nzp;
  a.nxp = g->nxp; a.nyp = g->nyp; a.nzp = g->nzp;
  a.x_m = g->x_m; a.x_M = g->x_M; a.y_m = g->y_m; a.y_M = g->y_M; a.z_m = g->z_m; a.z_M = g->z_M;
  a.dt2 = g->dt * g->dt;                 // openacc.cpp:160 `dt*dt*(...)`
  a.r1 = 1.0F / (g->dt * g->dt);         // openacc.cpp:84-87
  a.r2 = 1.0F / (g->h_x * g->h_x);
  a.r3 = 1.0F / (g->h_y * g->h_y);
  a.r4 = 1.0F / (g->h_z * g->h_z);
  const DampTerm dm{damp, g->dt, 1.0F / g->dt};
  hipStream_t s = (hipStream_t)stream;
  if (order_march_takes(a, H)) {
    // non-temporal stores: the level written is not read again before the next sweep has streamed the
    // whole field through the caches (FDTD_ORDER_NT = 0 keeps plain stores, for measurements)
    const int nt = env_int_orders("FDTD_ORDER_NT", 1);
    // rows that fill multiples of 128 cells better than multiples of 256 (384 = 3 x 128): half a wave per row
    const int lpr_auto = ((ez + 127) / 128) * 128 < ((ez + 255) / 256) * 256 ? 32 : 64;
    const int lpr = env_int_orders("FDTD_ORDER_LPR", lpr_auto) == 32 ? 32 : 64;
    // rows of whole float4s, radius up to 4, tiles of 256 cells: the register-column sweep (FDTD_ORDER_COLUMN=0:
    // the plane-in-LDS sweep below for every radius)
    if (R <= env_int_orders("FDTD_ORDER_COLUMN_MAX_R", 3) && R <= 4 && lpr == 64 && g->z_m % 4 == 0 && ez % 4 == 0 &&
        env_int_orders("FDTD_ORDER_COLUMN", 1) != 0) {
#define ORDER_COLUMN(R_)                                                            \
  case R_:                                                                          \
    if (damp) launch_column_march<R_, true, true>(a, s, dm);                        \
    else if (nt) launch_column_march<R_, true, false>(a, s, dm);                    \
    else launch_column_march<R_, false, false>(a, s, dm);                           \
    break;
      switch (R) {
        ORDER_COLUMN(2)
        ORDER_COLUMN(3)
        default:
        ORDER_COLUMN(4)
      }
#undef ORDER_COLUMN
      return (int)hipGetLastError();
    }
#define ORDER_MARCH(R_)                                                                    \
  case R_:                                                                                 \
    if (damp && lpr == 32) launch_order_march<R_, 2, 1, true, true, 32>(a, s, dm);         \
    else if (damp) launch_order_march<R_, 2, 1, true, true>(a, s, dm);                     \
    else if (lpr == 32) launch_order_march<R_, 2, 1, true, false, 32>(a, s);               \
    else if (nt) launch_order_march<R_, 2, 1, true>(a, s);                                 \
    else launch_order_march<R_, 2, 1, false>(a, s);                                        \
    break;
    switch (R) {
      ORDER_MARCH(2)
      ORDER_MARCH(3)
      ORDER_MARCH(4)
      ORDER_MARCH(5)
      default:
      ORDER_MARCH(6)
    }
#undef ORDER_MARCH
    return (int)hipGetLastError();
  }
  if (ex > 65535 || (ey + 3) / 4 > 65535) return (int)hipErrorInvalidValue;
  const dim3 grid((ez + 63) / 64, (ey + 3) / 4, ex), block(64, 4, 1);
#define ORDER_CELL(R_)                                                                           \
  case R_:                                                                                       \
    if (damp) hipLaunchKernelGGL((stencil_order_kernel<R_, true>), grid, block, 0, s, a, dm);    \
    else hipLaunchKernelGGL((stencil_order_kernel<R_, false>), grid, block, 0, s, a, dm);        \
    break;
  switch (R) {
    ORDER_CELL(2)
    ORDER_CELL(3)
    ORDER_CELL(4)
    ORDER_CELL(5)
    default:
    ORDER_CELL(6)
  }
#undef ORDER_CELL
  return (int)hipGetLastError();
}

int fdtd_stencil_levels_order(const struct fdtd_geom *g, const float *m, const float *u0, const float *u1,
                              float *u2, int order, void *stream) {
  return sweep_order(g, m, nullptr, u0, u1, u2, order, stream);
}

int fdtd_stencil_levels_damped(const struct fdtd_geom *g, const float *m, const float *damp, const float *u0,
                               const float *u1, float *u2, int order, void *stream) {
  if (!damp) return (int)hipErrorInvalidValue;
  return sweep_order(g, m, damp, u0, u1, u2, order, stream);
}

extern "C" int fdtd_hip_stencil_step_order(const struct fdtd_geom *g, const float *m, float *u, int t0, int t1,
                                           int t2, int order, void *stream) {
  if (!g || !u) return (int)hipErrorInvalidValue;
  if (t0 < 0 || t0 > 2 || t1 < 0 || t1 > 2 || t2 < 0 || t2 > 2) return (int)hipErrorInvalidValue;
  const long vol = (long)g->nxp * g->nyp * g->nzp;
  return fdtd_stencil_levels_order(g, m, u + (long)t0 * vol, u + (long)t1 * vol, u + (long)t2 * vol, order, stream);
}

extern "C" int fdtd_hip_source_inject_order(const struct fdtd_geom *g, const float *m, const float *src,
                                            const float *src_coords, float *u, int t2, int time, int p_src_m,
                                            int p_src_M, int src_stride, int coord_stride, int order, void *stream) {
  if (p_src_M < p_src_m) return 0;
  if (!g || !u || t2 < 0 || t2 > 2 || !order_supported(order)) return (int)hipErrorInvalidValue;
  const long vol = (long)g->nxp * g->nyp * g->nzp;
  return fdtd_inject_level_halo(g, m, src, src_coords, u + (long)t2 * vol, time, p_src_m, p_src_M, src_stride,
                                coord_stride, 0, g->x_m - 1, g->x_M + 1, 0, 0, 0, order, stream);
}

// Slab form of the scatter at any space order (arrays padded by `order` cells hold global planes x_shift-order ..;
// a corner is applied iff its global x index lies in [x_lo, x_hi]): fdtd_hip_source_inject_slab for the wider padding.
extern "C" int fdtd_hip_source_inject_slab_order(const struct fdtd_geom *g, const float *m, const float *src,
                                                 const float *src_coords, float *u, int t2, int time, int p_src_m,
                                                 int p_src_M, int src_stride, int coord_stride, int x_shift, int x_lo,
                                                 int x_hi, int order, void *stream) {
  if (p_src_M < p_src_m) return 0;
  if (!g || !u || t2 < 0 || t2 > 2 || !order_supported(order)) return (int)hipErrorInvalidValue;
  const long vol = (long)g->nxp * g->nyp * g->nzp;
  return fdtd_inject_level_halo(g, m, src, src_coords, u + (long)t2 * vol, time, p_src_m, p_src_M, src_stride,
                                coord_stride, x_shift, x_lo, x_hi, 0, 0, 0, order, stream);
}
