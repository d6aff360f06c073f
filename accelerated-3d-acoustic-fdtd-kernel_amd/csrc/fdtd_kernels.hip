// fdtd_kernels.hip — gfx950 (MI355X) device kernels of the acoustic FDTD hot path and
// their launchers (the device-resident ABI of include/fdtd_abi.h).
//
// section0: 13-point radius-2 star Laplacian + leapfrog update
//           (reference openacc.cpp:159-164, cuda.cu:61-108, cuda_optimized.cu:63-238)
// section1: trilinear 8-corner source scatter
//           (reference openacc.cpp:185-196, cuda.cu:112-170)
//
// Arithmetic contract: the update is evaluated in the operand order and association
// of openacc.cpp:159-164 with IEEE binary32 operations, a correctly rounded divide,
// denormals kept and no FMA contraction (this file is built with -ffp-contract=off),
// so every variant returns the bits a serial host run of the reference returns.
//
// The path is HBM-bound (16 B of compulsory traffic per cell update against ~45
// VALU operations), so the design work is in the memory system: 16-byte lane
// accesses, every wave streaming along x with the x-neighbours held in registers,
// no workgroup barriers, loads for plane x+1 in flight while plane x is computed,
// and a tile->workgroup map that keeps y-adjacent tiles on one XCD's L2.
#include "fdtd_device.h"

#pragma clang fp contract(off)

// Stencil coefficients (openacc.cpp:159-163; cuda.cu:51-53).  Scalar loads, once per wave.
__constant__ float c_m2 = -8.33333333e-2F;  // -1/12
__constant__ float c_m1 = 1.333333330F;     //  4/3
__constant__ float c_0 = -2.50F;            // -5/2

// ---------------------------------------------------------------------------------
// The cell update, shared by every variant.  Operand order follows openacc.cpp:159-164.
// ---------------------------------------------------------------------------------
struct Coef {
  float cm2, cm1, c0, dt2, r1, r2, r3, r4, neg2r1;
};

__device__ __forceinline__ Coef load_coef(const StencilArgs &a) {
  Coef k;
  k.cm2 = c_m2;
  k.cm1 = c_m1;
  k.c0 = c_0;
  k.dt2 = a.dt2;
  k.r1 = a.r1;
  k.r2 = a.r2;
  k.r3 = a.r3;
  k.r4 = a.r4;
  k.neg2r1 = -2.0F * a.r1;
  return k;
}

__device__ __forceinline__ float update_cell(const Coef &k, float uc, float xm2, float xm1,
                                             float xp1, float xp2, float ym2, float ym1,
                                             float yp1, float yp2, float zm2, float zm1,
                                             float zp1, float zp2, float up, float mc) {
  const float r5 = k.c0 * uc;
  const float dxx = r5 + k.cm2 * (xm2 + xp2) + k.cm1 * (xm1 + xp1);
  const float dyy = r5 + k.cm2 * (ym2 + yp2) + k.cm1 * (ym1 + yp1);
  const float dzz = r5 + k.cm2 * (zm2 + zp2) + k.cm1 * (zm1 + zp1);
  return k.dt2 * (k.r2 * dxx + k.r3 * dyy + k.r4 * dzz - (k.neg2r1 * uc + k.r1 * up) * mc) / mc;
}

__device__ __forceinline__ float update_from_global(const Coef &k, const StencilArgs &a, long c) {
  const float *__restrict__ u0 = a.u0;
  const long sx = a.sx, sy = a.sy;
  return update_cell(k, u0[c], u0[c - 2 * sx], u0[c - sx], u0[c + sx], u0[c + 2 * sx],
                     u0[c - 2 * sy], u0[c - sy], u0[c + sy], u0[c + 2 * sy], u0[c - 2],
                     u0[c - 1], u0[c + 1], u0[c + 2], a.u1[c], a.m[c]);
}

// ---------------------------------------------------------------------------------
// FLAT: the collapse(3) sweep as an OpenACC compiler lays it out — one flat index over
// (x,y,z), z fastest, one cell per thread (openacc.cpp:151-156).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stencil_flat_kernel(StencilArgs a) {
  const long ez = a.z_M - a.z_m + 1, ey = a.y_M - a.y_m + 1, ex = a.x_M - a.x_m + 1;
  const long total = ex * ey * ez;
  const Coef k = load_coef(a);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long z = i % ez, y = (i / ez) % ey, x = i / (ez * ey);
    const long c = (x + a.x_m + FDTD_HALO) * a.sx + (y + a.y_m + FDTD_HALO) * a.sy +
                   (z + a.z_m + FDTD_HALO);
    a.u2[c] = update_from_global(k, a, c);
  }
}

// ---------------------------------------------------------------------------------
// NAIVE: one thread per cell on a 3-D grid, every operand straight from global memory
// (cuda.cu:61-108) — but with z, the contiguous axis, on the lanes, so that a wave's
// 13 loads are 13 coalesced 256-byte rows.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stencil_naive_kernel(StencilArgs a) {
  const int z = a.z_m + blockIdx.x * 64 + threadIdx.x;
  const int y = a.y_m + blockIdx.y * 4 + threadIdx.y;
  const int x = a.x_m + blockIdx.z;
  if (z > a.z_M || y > a.y_M) return;
  const Coef k = load_coef(a);
  const long c = (long)(x + FDTD_HALO) * a.sx + (long)(y + FDTD_HALO) * a.sy + (z + FDTD_HALO);
  a.u2[c] = update_from_global(k, a, c);
}

// ---------------------------------------------------------------------------------
// MARCH: the fast path.
//
// A wave owns a tile of RY rows (y) by 256 cells (z, one float4 per lane) and marches
// along x.  Everything it needs lives in its own registers:
//   * the x-neighbours: a 5-plane register column (x-2..x+2) per owned row;
//   * the y-neighbours: the centre plane of the 2 rows above and below the tile is
//     loaded too, so rows y-2..y+RY+1 of the centre plane are all register resident;
//   * the z-neighbours: lanes hold consecutive float4s, so z-2,z-1 / z+4,z+5 are the
//     .z,.w / .x,.y of the previous / next lane, fetched with DPP wavefront shifts;
//     the first and last lane take them from an 8-byte edge load instead (it rides
//     in the `old` operand of the DPP move, which is what an unsourced lane keeps).
// There is no LDS and no barrier: each wave runs its own software pipeline, issuing
// the loads of iteration x+1 (u0 plane x+3 of its rows, halo rows and edges of plane
// x+1, u1 and m of plane x+1) before it computes plane x, so about 5 KB per owned row
// are in flight per wave at all times.
//
// Requirements checked by the launcher: nzp % 4 == 0, z_m % 4 == 0, extent in z a
// multiple of 4, 16-byte aligned level pointers, and z_M + 8 <= nzp - 1 so that the
// float4 behind the last owned one is inside the row.
// ---------------------------------------------------------------------------------
constexpr int MARCH_WAVES = 4;  // waves per workgroup (independent; grouped for locality)

__device__ __forceinline__ float dpp_from_prev_lane(float edge, float v) {
  // lane i <- lane i-1; lane 0 keeps `edge`
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                         __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_next_lane(float edge, float v) {
  // lane i <- lane i+1; lane 63 keeps `edge`
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                         __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// Workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the XCD).  Give each
// XCD one contiguous range of tile ids so that tiles sharing halo rows share an L2.
__device__ __forceinline__ int xcd_contiguous_tile(int b, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, xcd = b & 7, j = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <int RY>
__global__ __launch_bounds__(MARCH_WAVES * 64) void stencil_march_kernel(StencilArgs a, int xchunk,
                                                                         int nzt, int nyt) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int t = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int zt = t % nzt;
  const int yt = (t / nzt) % nyt;
  const int xc = t / (nzt * nyt);

  const int y0 = a.y_m + (yt * MARCH_WAVES + wave) * RY;  // first owned row
  if (y0 > a.y_M) return;                                  // wave-uniform, no barriers below
  const int zfirst = a.z_m + zt * 256 + lane * 4;          // first owned cell of this lane
  const bool store_z = zfirst <= a.z_M;
  // Lanes behind the last owned float4 load the float4 that follows it (it supplies
  // z+1,z+2 to the last owned lane); further lanes repeat that address.
  const int zload = min(zfirst, a.z_M + 1);
  const int x_begin = a.x_m + xc * xchunk;
  const int x_end = min(x_begin + xchunk - 1, a.x_M);

  const Coef k = load_coef(a);
  const float *__restrict__ u0 = a.u0;
  const float *__restrict__ u1 = a.u1;
  const float *__restrict__ mm = a.m;
  const long sx = a.sx, sy = a.sy;

  // Row offsets (elements, relative to a plane base) of rows y0-2 .. y0+RY+1; rows
  // past y_M+2 repeat the last readable row so that no load leaves the array.
  long rowoff[RY + 4];
#pragma unroll
  for (int r = 0; r < RY + 4; ++r) {
    const int y = min(y0 - 2 + r, a.y_M + 2);
    rowoff[r] = (long)(y + FDTD_HALO) * sy + (zload + FDTD_HALO);
  }
  // Edge loads: lane 0 fetches (z-2,z-1) of its float4, lane 63 fetches (z+4,z+5).
  const bool edge_lane = (lane == 0) || (lane == 63 && store_z);
  const long edge_shift = (lane == 0) ? -2 : 4;

  // Register column: col[p][r] holds plane (x-2+p) of owned row r.
  float4 col[5][RY];
  float4 halo[4];       // centre plane of rows y0-2, y0-1, y0+RY, y0+RY+1
  float2 edge[RY];      // centre plane z-edges of the owned rows (edge lanes only)
  float4 prev[RY], med[RY];  // u1 and m of the centre plane

  {
    const long base = (long)(x_begin + FDTD_HALO) * sx;
#pragma unroll
    for (int r = 0; r < RY; ++r) {
#pragma unroll
      for (int p = 0; p < 5; ++p)
        col[p][r] = *reinterpret_cast<const float4 *>(u0 + base + (p - 2) * sx + rowoff[r + 2]);
      prev[r] = *reinterpret_cast<const float4 *>(u1 + base + rowoff[r + 2]);
      med[r] = *reinterpret_cast<const float4 *>(mm + base + rowoff[r + 2]);
      edge[r] = make_float2(0.f, 0.f);
      if (edge_lane)
        edge[r] = *reinterpret_cast<const float2 *>(u0 + base + rowoff[r + 2] + edge_shift);
    }
    halo[0] = *reinterpret_cast<const float4 *>(u0 + base + rowoff[0]);
    halo[1] = *reinterpret_cast<const float4 *>(u0 + base + rowoff[1]);
    halo[2] = *reinterpret_cast<const float4 *>(u0 + base + rowoff[RY + 2]);
    halo[3] = *reinterpret_cast<const float4 *>(u0 + base + rowoff[RY + 3]);
  }

  for (int x = x_begin; x <= x_end; ++x) {
    const long base = (long)(x + FDTD_HALO) * sx;
    // ---- issue the loads of the next iteration (plane x+1 becomes the centre) ----
    // At the last plane they are repeated on plane x (in bounds, unused).
    const long nbase = (x < x_end) ? base + sx : base;
    float4 ncol[RY], nprev[RY], nmed[RY], nhalo[4];
    float2 nedge[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      ncol[r] = *reinterpret_cast<const float4 *>(u0 + nbase + 2 * sx + rowoff[r + 2]);
      nprev[r] = *reinterpret_cast<const float4 *>(u1 + nbase + rowoff[r + 2]);
      nmed[r] = *reinterpret_cast<const float4 *>(mm + nbase + rowoff[r + 2]);
      nedge[r] = make_float2(0.f, 0.f);
      if (edge_lane)
        nedge[r] = *reinterpret_cast<const float2 *>(u0 + nbase + rowoff[r + 2] + edge_shift);
    }
    nhalo[0] = *reinterpret_cast<const float4 *>(u0 + nbase + rowoff[0]);
    nhalo[1] = *reinterpret_cast<const float4 *>(u0 + nbase + rowoff[1]);
    nhalo[2] = *reinterpret_cast<const float4 *>(u0 + nbase + rowoff[RY + 2]);
    nhalo[3] = *reinterpret_cast<const float4 *>(u0 + nbase + rowoff[RY + 3]);

    // ---- compute plane x ----
    // centre plane of rows y0-2 .. y0+RY+1 (register aliases, no data movement)
    float4 cen[RY + 4];
    cen[0] = halo[0];
    cen[1] = halo[1];
#pragma unroll
    for (int r = 0; r < RY; ++r) cen[r + 2] = col[2][r];
    cen[RY + 2] = halo[2];
    cen[RY + 3] = halo[3];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      const float4 c = cen[r + 2];
      const float4 ym2 = cen[r], ym1 = cen[r + 1], yp1 = cen[r + 3], yp2 = cen[r + 4];
      // z neighbours across lanes
      const float lz = dpp_from_prev_lane(edge[r].x, c.z);  // z-2 of .x
      const float lw = dpp_from_prev_lane(edge[r].y, c.w);  // z-1 of .x
      const float rx = dpp_from_next_lane(edge[r].x, c.x);  // z+1 of .w
      const float ry = dpp_from_next_lane(edge[r].y, c.y);  // z+2 of .w
      const float4 xm2 = col[0][r], xm1 = col[1][r], xp1 = col[3][r], xp2 = col[4][r];
      const float4 up = prev[r], mc = med[r];
      float4 o;
      o.x = update_cell(k, c.x, xm2.x, xm1.x, xp1.x, xp2.x, ym2.x, ym1.x, yp1.x, yp2.x, lz, lw,
                        c.y, c.z, up.x, mc.x);
      o.y = update_cell(k, c.y, xm2.y, xm1.y, xp1.y, xp2.y, ym2.y, ym1.y, yp1.y, yp2.y, lw, c.x,
                        c.z, c.w, up.y, mc.y);
      o.z = update_cell(k, c.z, xm2.z, xm1.z, xp1.z, xp2.z, ym2.z, ym1.z, yp1.z, yp2.z, c.x, c.y,
                        c.w, rx, up.z, mc.z);
      o.w = update_cell(k, c.w, xm2.w, xm1.w, xp1.w, xp2.w, ym2.w, ym1.w, yp1.w, yp2.w, c.y, c.z,
                        rx, ry, up.w, mc.w);
      if (store_z && y0 + r <= a.y_M)
        *reinterpret_cast<float4 *>(a.u2 + base + rowoff[r + 2]) = o;
    }

    // ---- rotate the pipeline ----
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      col[0][r] = col[1][r];
      col[1][r] = col[2][r];
      col[2][r] = col[3][r];
      col[3][r] = col[4][r];
      col[4][r] = ncol[r];
      prev[r] = nprev[r];
      med[r] = nmed[r];
      edge[r] = nedge[r];
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) halo[h] = nhalo[h];
  }
}

// ---------------------------------------------------------------------------------
// section1: one thread per (source, corner).  Weights, bounds test and amplitude are
// the expressions of openacc.cpp:185-196; the += is a device-scope float atomic
// (cuda.cu:165).  Corners of one source never coincide, so with sources that do not
// share cells the result does not depend on the order the atomics land in.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void source_inject_kernel(InjectArgs a) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int nsrc = a.p_src_M - a.p_src_m + 1;
  if (tid >= nsrc * 8) return;
  const int p = a.p_src_m + (tid >> 3);
  const int rx = (tid >> 2) & 1, ry = (tid >> 1) & 1, rz = tid & 1;

  const float gx = (-a.o_x + a.coords[(long)p * a.coord_stride + 0]) / a.h_x;
  const float gy = (-a.o_y + a.coords[(long)p * a.coord_stride + 1]) / a.h_y;
  const float gz = (-a.o_z + a.coords[(long)p * a.coord_stride + 2]) / a.h_z;
  const int posx = (int)floorf(gx), posy = (int)floorf(gy), posz = (int)floorf(gz);
  const float px = -floorf(gx) + gx, py = -floorf(gy) + gy, pz = -floorf(gz) + gz;

  if (rx + posx >= a.x_m - 1 && ry + posy >= a.y_m - 1 && rz + posz >= a.z_m - 1 &&
      rx + posx <= a.x_M + 1 && ry + posy <= a.y_M + 1 && rz + posz <= a.z_M + 1) {
    const float r0 = 1.0e-2F * (rx * px + (1 - rx) * (1 - px)) * (ry * py + (1 - ry) * (1 - py)) *
                     (rz * pz + (1 - rz) * (1 - pz)) * a.src[(long)a.time * a.src_stride + p] /
                     a.m[(long)(posx + FDTD_HALO) * a.sx + (long)(posy + FDTD_HALO) * a.sy +
                         (posz + FDTD_HALO)];
    atomicAdd(a.u2 + (long)(rx + posx + FDTD_HALO) * a.sx + (long)(ry + posy + FDTD_HALO) * a.sy +
                  (rz + posz + FDTD_HALO),
              r0);
  }
}

// ---------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------
static int env_int(const char *name, int fallback) {
  const char *v = getenv(name);
  return v ? atoi(v) : fallback;
}

static bool geom_valid(const fdtd_geom *g) {
  if (!g || g->nxp <= 0 || g->nyp <= 0 || g->nzp <= 0) return false;
  // every stencil read (radius 2 around an updated cell) must stay inside the array
  if (g->x_m <= g->x_M && (g->x_m + FDTD_HALO - 2 < 0 || g->x_M + FDTD_HALO + 2 > g->nxp - 1)) return false;
  if (g->y_m <= g->y_M && (g->y_m + FDTD_HALO - 2 < 0 || g->y_M + FDTD_HALO + 2 > g->nyp - 1)) return false;
  if (g->z_m <= g->z_M && (g->z_m + FDTD_HALO - 2 < 0 || g->z_M + FDTD_HALO + 2 > g->nzp - 1)) return false;
  return true;
}

static bool march_eligible(const fdtd_geom *g, const float *m, const float *u) {
  const int ez = g->z_M - g->z_m + 1;
  return (g->nzp % 4 == 0) && (g->z_m % 4 == 0) && (ez % 4 == 0) &&
         (g->z_M + 8 <= g->nzp - 1) && (((uintptr_t)m | (uintptr_t)u) % 16 == 0) &&
         (((long)g->nxp * g->nyp * g->nzp) % 4 == 0);
}

static StencilArgs make_args(const fdtd_geom *g, const float *m, float *u, int t0, int t1, int t2) {
  StencilArgs a;
  const long vol = (long)g->nxp * g->nyp * g->nzp;
  a.m = m;
  a.u0 = u + (long)t0 * vol;
  a.u1 = u + (long)t1 * vol;
  a.u2 = u + (long)t2 * vol;
  a.sx = (long)g->nyp * g->nzp;
  a.sy = g->nzp;
  a.nxp = g->nxp; a.nyp = g->nyp; a.nzp = g->nzp;
  a.x_m = g->x_m; a.x_M = g->x_M; a.y_m = g->y_m; a.y_M = g->y_M; a.z_m = g->z_m; a.z_M = g->z_M;
  a.dt2 = g->dt * g->dt;                 // openacc.cpp:160 `dt*dt*(...)`
  a.r1 = 1.0F / (g->dt * g->dt);         // openacc.cpp:84-87
  a.r2 = 1.0F / (g->h_x * g->h_x);
  a.r3 = 1.0F / (g->h_y * g->h_y);
  a.r4 = 1.0F / (g->h_z * g->h_z);
  return a;
}

template <int RY>
static void launch_march(const StencilArgs &a, hipStream_t s) {
  const int ex = a.x_M - a.x_m + 1, ey = a.y_M - a.y_m + 1, ez = a.z_M - a.z_m + 1;
  const int nzt = (ez + 255) / 256;
  const int nyt = (ey + MARCH_WAVES * RY - 1) / (MARCH_WAVES * RY);
  // Choose the x-chunk so that the grid has at least ~4 workgroups per CU, but never
  // below 8 planes (each chunk re-reads 4 planes of u0 to prime its register column).
  int xchunk = env_int("FDTD_XCHUNK", 0);
  if (xchunk <= 0) {
    const long want_blocks = 1024;
    long chunks = (want_blocks + (long)nzt * nyt - 1) / ((long)nzt * nyt);
    if (chunks < 1) chunks = 1;
    xchunk = (int)((ex + chunks - 1) / chunks);
    if (xchunk < 8) xchunk = 8;
    if (xchunk > ex) xchunk = ex;
  }
  const int nxc = (ex + xchunk - 1) / xchunk;
  const long nblocks = (long)nzt * nyt * nxc;
  hipLaunchKernelGGL(stencil_march_kernel<RY>, dim3((unsigned)nblocks), dim3(MARCH_WAVES * 64), 0, s,
                     a, xchunk, nzt, nyt);
}

extern "C" int fdtd_hip_stencil_step(const struct fdtd_geom *g, const float *m, float *u, int t0,
                                     int t1, int t2, int variant, void *stream) {
  if (!geom_valid(g) || !m || !u) return (int)hipErrorInvalidValue;
  if (t0 < 0 || t0 > 2 || t1 < 0 || t1 > 2 || t2 < 0 || t2 > 2) return (int)hipErrorInvalidValue;
  const int ex = g->x_M - g->x_m + 1, ey = g->y_M - g->y_m + 1, ez = g->z_M - g->z_m + 1;
  if (ex <= 0 || ey <= 0 || ez <= 0) return 0;  // empty range: nothing to update
  hipStream_t s = (hipStream_t)stream;
  const StencilArgs a = make_args(g, m, u, t0, t1, t2);

  if (variant == FDTD_VARIANT_AUTO)
    variant = march_eligible(g, m, u) ? FDTD_VARIANT_MARCH : FDTD_VARIANT_NAIVE;
  if ((variant == FDTD_VARIANT_MARCH || variant == FDTD_VARIANT_TILE3D) && !march_eligible(g, m, u))
    variant = FDTD_VARIANT_NAIVE;

  switch (variant) {
    case FDTD_VARIANT_FLAT: {
      const long total = (long)ex * ey * ez;
      long blocks = (total + 255) / 256;
      if (blocks > 256L * 64) blocks = 256L * 64;  // grid-stride beyond 64 blocks per CU
      hipLaunchKernelGGL(stencil_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
      break;
    }
    case FDTD_VARIANT_NAIVE: {
      if (ex > 65535 || (ey + 3) / 4 > 65535) return (int)hipErrorInvalidValue;
      dim3 grid((ez + 63) / 64, (ey + 3) / 4, ex);
      hipLaunchKernelGGL(stencil_naive_kernel, grid, dim3(64, 4, 1), 0, s, a);
      break;
    }
    case FDTD_VARIANT_MARCH:
    case FDTD_VARIANT_TILE3D: {
      const int ry = env_int("FDTD_MARCH_RY", 4);
      if (ry == 2) launch_march<2>(a, s);
      else if (ry == 1) launch_march<1>(a, s);
      else launch_march<4>(a, s);
      break;
    }
    default:
      return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}

extern "C" int fdtd_hip_source_inject(const struct fdtd_geom *g, const float *m, const float *src,
                                      const float *src_coords, float *u, int t2, int time,
                                      int p_src_m, int p_src_M, int src_stride, int coord_stride,
                                      void *stream) {
  if (p_src_M < p_src_m) return 0;
  if (!g || !m || !src || !src_coords || !u || t2 < 0 || t2 > 2 || coord_stride < 3 ||
      src_stride < 1)
    return (int)hipErrorInvalidValue;
  InjectArgs a;
  const long vol = (long)g->nxp * g->nyp * g->nzp;
  a.m = m; a.src = src; a.coords = src_coords;
  a.u2 = u + (long)t2 * vol;
  a.sx = (long)g->nyp * g->nzp; a.sy = g->nzp;
  a.x_m = g->x_m; a.x_M = g->x_M; a.y_m = g->y_m; a.y_M = g->y_M; a.z_m = g->z_m; a.z_M = g->z_M;
  a.h_x = g->h_x; a.h_y = g->h_y; a.h_z = g->h_z; a.o_x = g->o_x; a.o_y = g->o_y; a.o_z = g->o_z;
  a.p_src_m = p_src_m; a.p_src_M = p_src_M; a.time = time;
  a.src_stride = src_stride; a.coord_stride = coord_stride;
  const long threads = (long)(p_src_M - p_src_m + 1) * 8;
  hipLaunchKernelGGL(source_inject_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------
// fp16-storage mode: the three wavefield levels live in HBM as IEEE binary16, all
// arithmetic is the fp32 update above, the result is rounded to nearest-even on store.
// 10 B of compulsory traffic per cell update (2+2+2 for u, 4 for m) instead of 16.
// (The reference's USE_FP32_ONLY=0 build keeps an fp16 mirror but still time-steps
// fp32 shadows, cuda_optimized.cu:301-302,482-499; here the stored state itself is fp16.)
// ---------------------------------------------------------------------------------
#include <hip/hip_fp16.h>

struct StencilArgsH16 {
  const float *m;
  const __half *u0;
  const __half *u1;
  __half *u2;
  StencilArgs f;  // geometry and coefficients (pointer members unused)
};

__global__ __launch_bounds__(256) void stencil_naive_h16_kernel(StencilArgsH16 h) {
  const StencilArgs &a = h.f;
  const int z = a.z_m + blockIdx.x * 64 + threadIdx.x;
  const int y = a.y_m + blockIdx.y * 4 + threadIdx.y;
  const int x = a.x_m + blockIdx.z;
  if (z > a.z_M || y > a.y_M) return;
  const Coef k = load_coef(a);
  const long sx = a.sx, sy = a.sy;
  const long c = (long)(x + FDTD_HALO) * sx + (long)(y + FDTD_HALO) * sy + (z + FDTD_HALO);
  const __half *__restrict__ u0 = h.u0;
#define H(i) __half2float(u0[i])
  const float v = update_cell(k, H(c), H(c - 2 * sx), H(c - sx), H(c + sx), H(c + 2 * sx),
                              H(c - 2 * sy), H(c - sy), H(c + sy), H(c + 2 * sy), H(c - 2),
                              H(c - 1), H(c + 1), H(c + 2), __half2float(h.u1[c]), h.m[c]);
#undef H
  h.u2[c] = __float2half_rn(v);
}

struct InjectArgsH16 {
  InjectArgs f;  // u2 member unused
  __half *u2;
};

// Same scatter as source_inject_kernel; the += on a binary16 element is a compare-and-
// swap on the aligned 32-bit word that holds it (two corners can share a word).
__global__ __launch_bounds__(256) void source_inject_h16_kernel(InjectArgsH16 h) {
  const InjectArgs &a = h.f;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int nsrc = a.p_src_M - a.p_src_m + 1;
  if (tid >= nsrc * 8) return;
  const int p = a.p_src_m + (tid >> 3);
  const int rx = (tid >> 2) & 1, ry = (tid >> 1) & 1, rz = tid & 1;
  const float gx = (-a.o_x + a.coords[(long)p * a.coord_stride + 0]) / a.h_x;
  const float gy = (-a.o_y + a.coords[(long)p * a.coord_stride + 1]) / a.h_y;
  const float gz = (-a.o_z + a.coords[(long)p * a.coord_stride + 2]) / a.h_z;
  const int posx = (int)floorf(gx), posy = (int)floorf(gy), posz = (int)floorf(gz);
  const float px = -floorf(gx) + gx, py = -floorf(gy) + gy, pz = -floorf(gz) + gz;
  if (rx + posx >= a.x_m - 1 && ry + posy >= a.y_m - 1 && rz + posz >= a.z_m - 1 &&
      rx + posx <= a.x_M + 1 && ry + posy <= a.y_M + 1 && rz + posz <= a.z_M + 1) {
    const float r0 = 1.0e-2F * (rx * px + (1 - rx) * (1 - px)) * (ry * py + (1 - ry) * (1 - py)) *
                     (rz * pz + (1 - rz) * (1 - pz)) * a.src[(long)a.time * a.src_stride + p] /
                     a.m[(long)(posx + FDTD_HALO) * a.sx + (long)(posy + FDTD_HALO) * a.sy +
                         (posz + FDTD_HALO)];
    const long c = (long)(rx + posx + FDTD_HALO) * a.sx + (long)(ry + posy + FDTD_HALO) * a.sy +
                   (rz + posz + FDTD_HALO);
    unsigned int *word = reinterpret_cast<unsigned int *>(h.u2) + (c >> 1);
    const int shift = (c & 1) ? 16 : 0;
    unsigned int seen = *word, assumed;
    do {
      assumed = seen;
      const unsigned short old_bits = (unsigned short)((assumed >> shift) & 0xffffu);
      const float sum = (float)__builtin_bit_cast(_Float16, old_bits) + r0;
      const unsigned short new_bits = __builtin_bit_cast(unsigned short, (_Float16)sum);
      const unsigned int next = (assumed & ~(0xffffu << shift)) | ((unsigned int)new_bits << shift);
      seen = atomicCAS(word, assumed, next);
    } while (seen != assumed);
  }
}

__global__ __launch_bounds__(256) void convert_f32_to_h16_kernel(const float *in, __half *out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = __float2half_rn(in[i]);
}
__global__ __launch_bounds__(256) void convert_h16_to_f32_kernel(const __half *in, float *out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = __half2float(in[i]);
}

extern "C" int fdtd_hip_stencil_step_h16(const struct fdtd_geom *g, const float *m, void *u_h16, int t0,
                                         int t1, int t2, void *stream) {
  if (!geom_valid(g) || !m || !u_h16) return (int)hipErrorInvalidValue;
  if (t0 < 0 || t0 > 2 || t1 < 0 || t1 > 2 || t2 < 0 || t2 > 2) return (int)hipErrorInvalidValue;
  const int ex = g->x_M - g->x_m + 1, ey = g->y_M - g->y_m + 1, ez = g->z_M - g->z_m + 1;
  if (ex <= 0 || ey <= 0 || ez <= 0) return 0;
  if (ex > 65535 || (ey + 3) / 4 > 65535) return (int)hipErrorInvalidValue;
  StencilArgsH16 h;
  h.f = make_args(g, m, nullptr, 0, 0, 0);
  const long vol = (long)g->nxp * g->nyp * g->nzp;
  __half *u = (__half *)u_h16;
  h.m = m;
  h.u0 = u + (long)t0 * vol;
  h.u1 = u + (long)t1 * vol;
  h.u2 = u + (long)t2 * vol;
  dim3 grid((ez + 63) / 64, (ey + 3) / 4, ex);
  hipLaunchKernelGGL(stencil_naive_h16_kernel, grid, dim3(64, 4, 1), 0, (hipStream_t)stream, h);
  return (int)hipGetLastError();
}

extern "C" int fdtd_hip_source_inject_h16(const struct fdtd_geom *g, const float *m, const float *src,
                                          const float *src_coords, void *u_h16, int t2, int time,
                                          int p_src_m, int p_src_M, int src_stride, int coord_stride,
                                          void *stream) {
  if (p_src_M < p_src_m) return 0;
  if (!g || !m || !src || !src_coords || !u_h16 || t2 < 0 || t2 > 2 || coord_stride < 3 ||
      src_stride < 1)
    return (int)hipErrorInvalidValue;
  InjectArgsH16 h;
  InjectArgs &a = h.f;
  const long vol = (long)g->nxp * g->nyp * g->nzp;
  a.m = m; a.src = src; a.coords = src_coords; a.u2 = nullptr;
  h.u2 = (__half *)u_h16 + (long)t2 * vol;
  a.sx = (long)g->nyp * g->nzp; a.sy = g->nzp;
  a.x_m = g->x_m; a.x_M = g->x_M; a.y_m = g->y_m; a.y_M = g->y_M; a.z_m = g->z_m; a.z_M = g->z_M;
  a.h_x = g->h_x; a.h_y = g->h_y; a.h_z = g->h_z; a.o_x = g->o_x; a.o_y = g->o_y; a.o_z = g->o_z;
  a.p_src_m = p_src_m; a.p_src_M = p_src_M; a.time = time;
  a.src_stride = src_stride; a.coord_stride = coord_stride;
  const long threads = (long)(p_src_M - p_src_m + 1) * 8;
  hipLaunchKernelGGL(source_inject_h16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, h);
  return (int)hipGetLastError();
}

extern "C" int fdtd_hip_convert_f32_to_h16(const float *in, void *out_h16, size_t n, void *stream) {
  if (n == 0) return 0;
  if (!in || !out_h16) return (int)hipErrorInvalidValue;
  size_t blocks = (n + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(convert_f32_to_h16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     in, (__half *)out_h16, n);
  return (int)hipGetLastError();
}
extern "C" int fdtd_hip_convert_h16_to_f32(const void *in_h16, float *out, size_t n, void *stream) {
  if (n == 0) return 0;
  if (!in_h16 || !out) return (int)hipErrorInvalidValue;
  size_t blocks = (n + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(convert_h16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const __half *)in_h16, out, n);
  return (int)hipGetLastError();
}
