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
// VALU operations), so the design work is in the memory system.  The default kernel
// (stencil_march_lds_kernel) uses 16-byte lane accesses, waves that stream along x
// with the x-neighbours held in registers, z-neighbours from adjacent lanes (DPP),
// y-neighbours handed between the waves of a workgroup through LDS (one barrier per
// plane), loads for plane x+1 in flight while plane x is computed, and a
// tile->workgroup map that keeps y-adjacent tiles on one XCD's L2 and marching in step.
// The other kernels (flat, naive, vec, tile3d) are the simpler shapes the reference
// and its north-star name; they serve small or unaligned grids and as measurement points.
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
// Building blocks of the marching kernel.
//
// A wave owns RY rows (y) by 256 cells (z, one float4 per lane) and marches along x:
//   * the x-neighbours live in a register column (planes x-2..x+2 of each owned row);
//   * the z-neighbours: lanes hold consecutive float4s, so z-2,z-1 / z+4,z+5 are the
//     .z,.w / .x,.y of the previous / next lane, fetched with DPP wavefront shifts; the
//     first and last lane take them from an 8-byte edge load instead (it rides in the
//     `old` operand of the DPP move, which is what an unsourced lane keeps);
//   * the y-neighbours come from the other waves of the workgroup through LDS.
// Requirements checked by the launcher: nzp % 4 == 0, z_m % 4 == 0, extent in z a
// multiple of 4, 16-byte aligned level pointers, and z_M + 8 <= nzp - 1 so that the
// float4 behind the last owned one is inside the row.
// ---------------------------------------------------------------------------------

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

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld_stream(const float *p) {
  if (NT) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const float4 *>(p);
}
template <bool NT>
__device__ __forceinline__ void st_stream(float *p, float4 v) {
  if (NT) {
    f32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4 *>(p));
  } else {
    *reinterpret_cast<float4 *>(p) = v;
  }
}

// ---------------------------------------------------------------------------------
// MARCH + LDS row exchange (the default fast path).
//
// Same register column along x as above, but the NW waves of a workgroup are stacked
// in y and hand each other the centre plane of their rows through LDS, so only the 4
// rows outside the NW*RY-row workgroup tile are fetched a second time (and those hit
// in L2 when the neighbouring workgroup marches in step).  One workgroup barrier per
// plane; the LDS plane is double buffered.
//
// Software pipeline.  gfx950 retires vector loads and stores through one in-order
// counter, so a wait for a load also waits for every older store.  The loop is
// therefore arranged so that nothing ever has to wait for the youngest operations:
//   * an iteration first issues every load the NEXT iteration needs (u0 plane x+3 of
//     the owned rows; u1, m, z-edges and the outside row of plane x+1) into one of two
//     register sets, then computes plane x from the other set, then stores;
//   * the two sets alternate by unrolling the loop twice (no register copies of
//     freshly loaded data), and the only copies are the 4 older planes of the column;
//   * no vector-memory instruction sits under a divergent branch (edge loads are
//     issued by every lane, full tiles store unconditionally), so the compiler's
//     counted waits are exact and the stores of plane x stay in flight while plane
//     x+1 is computed.
// Addresses are a wave-uniform plane pointer (scalar registers) plus one 32-bit lane
// offset per row.
// ---------------------------------------------------------------------------------
// Storage traits: how the wavefield levels are held in HBM.  A lane always moves 16 bytes of
// u per row and plane: 4 binary32 cells, or 8 binary16 cells (then a wave row spans 512 cells).
// m is binary32 in both modes.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct StoreF32 {
  static constexpr int CELLS = 4;
  static constexpr int ELEM = 4;
  typedef f32x4 raw_t;
  typedef f32x4 mem_t;   // a lane's 4 cells start 16 bytes into a 16-byte aligned row
  typedef f32x2 edge_t;  // 2 cells
  static __device__ __forceinline__ void unpack(raw_t v, float (&f)[CELLS]) {
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  static __device__ __forceinline__ raw_t pack(const float (&f)[CELLS]) {
    raw_t v = {f[0], f[1], f[2], f[3]};
    return v;
  }
  // cells z-2, z-1 (from the previous lane) and z+CELLS, z+CELLS+1 (from the next lane)
  // eL feeds the first lane (it has no previous lane), eR the last lane
  static __device__ __forceinline__ void neighbours(raw_t c, edge_t eL, edge_t eR, float &l0, float &l1,
                                                    float &r0, float &r1) {
    l0 = dpp_from_prev_lane(eL.x, c.z);
    l1 = dpp_from_prev_lane(eL.y, c.w);
    r0 = dpp_from_next_lane(eR.x, c.x);
    r1 = dpp_from_next_lane(eR.y, c.y);
  }
  static __device__ __forceinline__ void edge_values(edge_t e, float &e0, float &e1) { e0 = e.x; e1 = e.y; }
};

struct StoreF16 {
  static constexpr int CELLS = 8;
  static constexpr int ELEM = 2;
  typedef u32x4 raw_t;      // 8 packed binary16
  // In memory a lane's 8 cells start (z + 4) * 2 bytes into the row with z a multiple of 8: 8 bytes off a
  // 16-byte boundary.  The 16-byte accesses are therefore declared 8-byte aligned (gfx9 global accesses need
  // dword alignment only; the instruction is the same global_load_dwordx4).
  typedef u32x4 mem_t __attribute__((aligned(8)));
  typedef unsigned edge_t;  // 2 packed binary16
  static __device__ __forceinline__ float lo(unsigned w) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
  }
  static __device__ __forceinline__ float hi(unsigned w) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
  }
  static __device__ __forceinline__ unsigned pk(float a, float b) {  // round to nearest even
    return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)a) |
           ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)b) << 16);
  }
  static __device__ __forceinline__ void unpack(raw_t v, float (&f)[CELLS]) {
    f[0] = lo(v.x); f[1] = hi(v.x); f[2] = lo(v.y); f[3] = hi(v.y);
    f[4] = lo(v.z); f[5] = hi(v.z); f[6] = lo(v.w); f[7] = hi(v.w);
  }
  static __device__ __forceinline__ raw_t pack(const float (&f)[CELLS]) {
    raw_t v = {pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7])};
    return v;
  }
  static __device__ __forceinline__ void neighbours(raw_t c, edge_t eL, edge_t eR, float &l0, float &l1,
                                                    float &r0, float &r1) {
    const unsigned l = (unsigned)__builtin_amdgcn_update_dpp((int)eL, (int)c.w, 0x138, 0xf, 0xf, false);
    const unsigned r = (unsigned)__builtin_amdgcn_update_dpp((int)eR, (int)c.x, 0x130, 0xf, 0xf, false);
    l0 = lo(l); l1 = hi(l); r0 = lo(r); r1 = hi(r);
  }
  static __device__ __forceinline__ void edge_values(edge_t e, float &e0, float &e1) { e0 = lo(e); e1 = hi(e); }
};

template <typename ST, int RY>
struct MarchSet {
  typename ST::raw_t col[RY];   // u0, plane x+2 relative to the plane this set is "current" for
  typename ST::raw_t prev[RY];  // u1 of the current plane
  f32x4 med[RY][ST::CELLS / 4]; // m of the current plane
  typename ST::edge_t edge[RY]; // z-edge cells of the current plane: (z-2,z-1); full tiles: last lane holds (z+C,z+C+1)
  typename ST::edge_t edge_r[RY]; // partial tiles only: (z+CELLS, z+CELLS+1) for the last owned lane
  typename ST::raw_t halo;      // one row outside the workgroup tile, plane +2 like col (loader waves)
};

// T: the register type; M: the same vector as it lies in memory (possibly with a smaller alignment)
template <bool NT, typename T, typename M = T>
__device__ __forceinline__ T ld_raw(const char *p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const M *>(p));
  else return *reinterpret_cast<const M *>(p);
}
template <bool NT, typename T, typename M = T>
__device__ __forceinline__ void st_raw(char *p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<M *>(p));
  else *reinterpret_cast<M *>(p) = v;
}

// Level pointers are passed as bytes so that one kernel serves both storage types.
struct MarchArgs {
  const char *u0, *u1;
  char *u2;
  const char *m;
  StencilArgs f;  // geometry and coefficients (its pointer members are unused here)
};

// LPR = lanes per row.  With LPR = 64 a wave spans 64*CELLS cells of its RY rows.  With LPR = 32
// (16) the two halves (four quarters) of a wave work on different row groups of a tile half (a
// quarter) as wide, so that z extents of 1.5, 2.5, ... full-width tiles (384, 640; 320, 576) still
// fill every lane; each part then behaves like a narrow wave of its own ("virtual wave" vw below).
template <typename ST, int RY, int NW, bool NT, bool FULL, int LPR>
__global__ __launch_bounds__(NW * 64) void stencil_march_lds_kernel(MarchArgs ma, int xchunk, int nzt,
                                                                     int nyt) {
  constexpr int WR = 64 / LPR;          // row groups side by side in one wave
  constexpr int NWV = NW * WR;          // virtual waves per workgroup
  constexpr int TY = RY * NWV;
  constexpr int CELLS = ST::CELLS;
  constexpr int TZ = LPR * CELLS;
  constexpr bool EDGE2 = !FULL || WR > 1;  // separate left/right edge loads and explicit selects
  typedef typename ST::raw_t raw_t;
  typedef typename ST::edge_t edge_t;
  __shared__ raw_t plane[2][TY + 4][LPR];
  const StencilArgs &a = ma.f;

  const int lane = threadIdx.x & 63;
  const int zl = lane % LPR;            // position along z inside the row group
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int vw = (WR == 1) ? wave : wave * WR + lane / LPR;
  const int t = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int zt = t % nzt;
  const int yt = (t / nzt) % nyt;
  const int xc = t / (nzt * nyt);

  const int yb = a.y_m + yt * TY;  // first row of the workgroup tile
  const int y0 = yb + vw * RY;     // first owned row of this (virtual) wave
  const int zfirst = a.z_m + zt * TZ + zl * CELLS;
  // The z extent is a multiple of CELLS, so a lane owns all of its cells or none.  Full tiles:
  // one edge load per row serves both ends (lane 63 fetches z+CELLS.., every other lane z-2..).
  // Partial tiles: lanes behind the last owned one repeat its address, and a second edge load
  // gives the last owned lane its z+CELLS, z+CELLS+1 (it may also be the first lane).
  const bool owned = FULL ? true : (zfirst <= a.z_M);
  const int zload = FULL ? zfirst : min(zfirst, a.z_M - (CELLS - 1));
  const bool last_owned = (zl == LPR - 1) || (!FULL && owned && zfirst + CELLS > a.z_M);
  const int x_begin = a.x_m + xc * xchunk;
  const int x_end = min(x_begin + xchunk - 1, a.x_M);

  const Coef k = load_coef(a);
  const unsigned sy = (unsigned)a.sy;
  const long sxe = a.sx * ST::ELEM;  // bytes per plane of u
  const long sxm = a.sx * 4;         // bytes per plane of m

  // 32-bit lane offsets (in cells) inside a plane; rows past y_M+2 repeat a readable row
  unsigned voff[RY], veoff[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    const int y = FULL ? (y0 + r) : min(y0 + r, a.y_M + 2);
    voff[r] = (unsigned)(y + FDTD_HALO) * sy + (unsigned)(zload + FDTD_HALO);
    veoff[r] = (!EDGE2 && lane == 63) ? voff[r] + CELLS : voff[r] - 2;  // (z+CELLS, z+CELLS+1) or (z-2, z-1)
  }
  // the two (virtual) waves at each end of the stack also fetch one row outside the tile;
  // with WR > 1 that is a per-lane property, so then every lane issues the load (the others
  // re-read their own row) and only the LDS write is conditional
  const bool halo_loader = (vw < 2) || (vw >= NWV - 2);
  const int halo_y = (vw < 2) ? yb - 2 + vw
                              : ((vw >= NWV - 2) ? min(yb + TY + (vw - (NWV - 2)), a.y_M + 2)
                                                 : min(y0, a.y_M + 2));
  const int halo_slot = (vw < 2) ? vw : TY + 2 + (vw - (NWV - 2));
  const unsigned vhoff = (unsigned)(halo_y + FDTD_HALO) * sy + (unsigned)(zload + FDTD_HALO);
  const bool halo_load = (WR > 1) ? true : halo_loader;

  // wave-uniform plane pointers of the first plane
  const char *pu0 = ma.u0 + (long)(x_begin + FDTD_HALO) * sxe;
  const char *pu1 = ma.u1 + (long)(x_begin + FDTD_HALO) * sxe;
  const char *pm = ma.m + (long)(x_begin + FDTD_HALO) * sxm;
  char *pu2 = ma.u2 + (long)(x_begin + FDTD_HALO) * sxe;

#define LDU(p, off) ld_raw<false, raw_t, typename ST::mem_t>((p) + (size_t)(off) * ST::ELEM)
#define LDS_(p, off) ld_raw<NT, raw_t, typename ST::mem_t>((p) + (size_t)(off) * ST::ELEM)
#define LDE(p, off) ld_raw<false, edge_t>((p) + (size_t)(off) * ST::ELEM)
#define LDM(p, off, j) ld_raw<NT, f32x4>((p) + (size_t)(off) * 4 + 16 * (j))

  raw_t col[4][RY];  // planes x-2, x-1, x, x+1 of the owned rows
  MarchSet<ST, RY> sa, sb;
#pragma unroll
  for (int r = 0; r < RY; ++r) {
#pragma unroll
    for (int p = 0; p < 4; ++p) col[p][r] = LDU(pu0 + (p - 2) * sxe, voff[r]);
    sa.col[r] = LDU(pu0 + 2 * sxe, voff[r]);
    sa.prev[r] = LDS_(pu1, voff[r]);
#pragma unroll
    for (int j = 0; j < CELLS / 4; ++j) sa.med[r][j] = LDM(pm, voff[r], j);
    sa.edge[r] = LDE(pu0, veoff[r]);
    if (EDGE2) sa.edge_r[r] = LDE(pu0, voff[r] + CELLS);
  }
  // The outside row is fetched for plane x+2 (first iteration) / x+3 (later), i.e. at the same
  // plane step at which the neighbouring tile fetches it as one of its own rows, so that one of
  // the two requests hits in L2; it waits in a two-deep queue until its plane is the centre.
  sa.halo = raw_t{};
  sb.halo = raw_t{};
  raw_t hq0 = raw_t{}, hq1 = raw_t{};  // outside row at planes x, x+1
  if (halo_load) {
    hq0 = LDU(pu0, vhoff);
    hq1 = LDU(pu0 + sxe, vhoff);
    sa.halo = LDU(pu0 + 2 * sxe, vhoff);
  }

  // One plane.  `cur` was loaded during the previous iteration; `nxt` is filled now.
  auto iteration = [&](int x, int buf, const MarchSet<ST, RY> &cur, MarchSet<ST, RY> &nxt)
                       __attribute__((always_inline)) {
    // ---- loads for plane x+1 (repeated on plane x at the end of the chunk) ----
    const bool more = x < x_end;
    const char *q0 = pu0 + (more ? sxe : 0), *q1 = pu1 + (more ? sxe : 0), *qm = pm + (more ? sxm : 0);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      nxt.col[r] = LDU(q0 + 2 * sxe, voff[r]);
      nxt.prev[r] = LDS_(q1, voff[r]);
#pragma unroll
      for (int j = 0; j < CELLS / 4; ++j) nxt.med[r][j] = LDM(qm, voff[r], j);
      nxt.edge[r] = LDE(q0, veoff[r]);
      if (EDGE2) nxt.edge_r[r] = LDE(q0, voff[r] + CELLS);
    }
    if (halo_load) nxt.halo = LDU(q0 + 2 * sxe, vhoff);  // wave-uniform (WR == 1) or unconditional

    // ---- publish the centre plane of the owned rows (and the outside row) ----
#pragma unroll
    for (int r = 0; r < RY; ++r) plane[buf][vw * RY + r + 2][zl] = col[2][r];
    if (halo_loader) plane[buf][halo_slot][zl] = hq0;
    __syncthreads();
    raw_t cen[RY + 4];
    cen[0] = plane[buf][vw * RY + 0][zl];
    cen[1] = plane[buf][vw * RY + 1][zl];
#pragma unroll
    for (int r = 0; r < RY; ++r) cen[r + 2] = col[2][r];
    cen[RY + 2] = plane[buf][vw * RY + RY + 2][zl];
    cen[RY + 3] = plane[buf][vw * RY + RY + 3][zl];

    // ---- compute and store plane x ----
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      float c[CELLS], ym2[CELLS], ym1[CELLS], yp1[CELLS], yp2[CELLS];
      float xm2[CELLS], xm1[CELLS], xp1[CELLS], xp2[CELLS], up[CELLS], mc[CELLS], o[CELLS];
      ST::unpack(cen[r + 2], c);
      ST::unpack(cen[r], ym2);
      ST::unpack(cen[r + 1], ym1);
      ST::unpack(cen[r + 3], yp1);
      ST::unpack(cen[r + 4], yp2);
      ST::unpack(col[0][r], xm2);
      ST::unpack(col[1][r], xm1);
      ST::unpack(col[3][r], xp1);
      ST::unpack(cur.col[r], xp2);
      ST::unpack(cur.prev[r], up);
#pragma unroll
      for (int j = 0; j < CELLS / 4; ++j) {
        mc[4 * j + 0] = cur.med[r][j].x; mc[4 * j + 1] = cur.med[r][j].y;
        mc[4 * j + 2] = cur.med[r][j].z; mc[4 * j + 3] = cur.med[r][j].w;
      }
      // z-neighbours outside the lane: l0,l1 = z-2,z-1 of cell 0; r0,r1 = z+1,z+2 of the last cell
      float l0, l1, r0, r1;
      if (!EDGE2) {
        ST::neighbours(cen[r + 2], cur.edge[r], cur.edge[r], l0, l1, r0, r1);
      } else {
        ST::neighbours(cen[r + 2], cur.edge[r], cur.edge_r[r], l0, l1, r0, r1);
        float e0, e1;
        ST::edge_values(cur.edge_r[r], e0, e1);
        r0 = last_owned ? e0 : r0;
        r1 = last_owned ? e1 : r1;
        if (WR > 1) {  // the first lane of a row group must not take the other group's cells
          ST::edge_values(cur.edge[r], e0, e1);
          l0 = (zl == 0) ? e0 : l0;
          l1 = (zl == 0) ? e1 : l1;
        }
      }
      float zz[CELLS + 4];
      zz[0] = l0; zz[1] = l1; zz[CELLS + 2] = r0; zz[CELLS + 3] = r1;
#pragma unroll
      for (int i = 0; i < CELLS; ++i) zz[i + 2] = c[i];
#pragma unroll
      for (int i = 0; i < CELLS; ++i)
        o[i] = update_cell(k, c[i], xm2[i], xm1[i], xp1[i], xp2[i], ym2[i], ym1[i], yp1[i], yp2[i],
                           zz[i], zz[i + 1], zz[i + 3], zz[i + 4], up[i], mc[i]);
      if (FULL || (owned && y0 + r <= a.y_M))
        st_raw<NT, raw_t, typename ST::mem_t>(pu2 + (size_t)voff[r] * ST::ELEM, ST::pack(o));
    }

    // ---- advance: shift the four older planes, step the plane pointers ----
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      col[0][r] = col[1][r];
      col[1][r] = col[2][r];
      col[2][r] = col[3][r];
      col[3][r] = cur.col[r];
    }
    hq0 = hq1;
    hq1 = cur.halo;
    pu0 += sxe; pu1 += sxe; pm += sxm; pu2 += sxe;
  };

  for (int x = x_begin; x <= x_end; x += 2) {
    iteration(x, 0, sa, sb);
    if (x + 1 > x_end) break;
    iteration(x + 1, 1, sb, sa);
  }
#undef LDU
#undef LDS_
#undef LDE
#undef LDM
}

// ---------------------------------------------------------------------------------
// VEC: one-shot vector kernel.  One lane = one float4 (4 cells in z) of one (x, y) row,
// one wave = 256 contiguous cells, one workgroup = 4 adjacent rows; nothing loops.  All
// 9 float4 of u0 a lane needs in x and y (centre, x-2..x+2, y-2..y+2) are plain aligned
// 16-byte loads; the z-neighbours come from the adjacent lanes by DPP exactly as in the
// marching kernel.  Reuse is left to the caches, and the launch order is arranged so
// that they can deliver it:
//   * the 4 rows of a workgroup are each other's y-neighbours (same CU, same moment);
//   * workgroup b runs on XCD b % 8; each XCD is given a fixed contiguous range of
//     y-blocks and walks the planes x = x_m .. x_M in dispatch order, so the five u0
//     planes a plane needs were brought into that XCD's L2 by the planes just before
//     (about 1/8 of 5 planes = 0.7 MB at 512^3 against 4 MB of L2).
// The grid is ~130k short-lived workgroups at 512^3: latency is hidden by residency
// (8 waves per SIMD), not by a software pipeline.
// ---------------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(256) void stencil_vec_kernel(StencilArgs a, int nzt, int yb_per_xcd,
                                                          int nyb) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  const int per_plane = yb_per_xcd * nzt;
  const int xi = j / per_plane;
  const int rem = j - xi * per_plane;
  const int yb = xcd * yb_per_xcd + rem / nzt;
  const int zt = rem % nzt;
  if (yb >= nyb) return;
  const int y = a.y_m + yb * 4 + wave;
  if (y > a.y_M) return;
  const int x = a.x_m + xi;
  const int zfirst = a.z_m + zt * 256 + lane * 4;
  const bool store_z = zfirst <= a.z_M;
  const int zload = min(zfirst, a.z_M + 1);

  const Coef k = load_coef(a);
  const long sx = a.sx, sy = a.sy;
  // wave-uniform row base (scalar registers) + per-lane element offset
  const long row = (long)(x + FDTD_HALO) * sx + (long)(y + FDTD_HALO) * sy + FDTD_HALO;
  const float *__restrict__ p0 = a.u0 + row;
  const unsigned voff = (unsigned)zload * 4u;  // zero-extended lane offset: scalar base + vector offset
#define ROW4(base) (*reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + voff))
  float4 c = ROW4(p0);
  const float4 xm2 = ROW4(p0 - 2 * sx);
  const float4 xm1 = ROW4(p0 - sx);
  const float4 xp1 = ROW4(p0 + sx);
  const float4 xp2 = ROW4(p0 + 2 * sx);
  const float4 ym2 = ROW4(p0 - 2 * sy);
  const float4 ym1 = ROW4(p0 - sy);
  const float4 yp1 = ROW4(p0 + sy);
  const float4 yp2 = ROW4(p0 + 2 * sy);
#undef ROW4
  const float4 up = ld_stream<NT>(reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.u1 + row) + voff));
  const float4 mc = ld_stream<NT>(reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.m + row) + voff));
  float2 edge = make_float2(0.f, 0.f);
  if (lane == 0 || (lane == 63 && store_z))
    edge = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(p0 + (lane == 0 ? -2 : 4)) + voff);
  // keep the centre float4 as loaded (one 16-byte load), opaque to load splitting
  asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w));

  const float lz = dpp_from_prev_lane(edge.x, c.z);
  const float lw = dpp_from_prev_lane(edge.y, c.w);
  const float rx = dpp_from_next_lane(edge.x, c.x);
  const float ry = dpp_from_next_lane(edge.y, c.y);
  float4 o;
  o.x = update_cell(k, c.x, xm2.x, xm1.x, xp1.x, xp2.x, ym2.x, ym1.x, yp1.x, yp2.x, lz, lw, c.y,
                    c.z, up.x, mc.x);
  o.y = update_cell(k, c.y, xm2.y, xm1.y, xp1.y, xp2.y, ym2.y, ym1.y, yp1.y, yp2.y, lw, c.x, c.z,
                    c.w, up.y, mc.y);
  o.z = update_cell(k, c.z, xm2.z, xm1.z, xp1.z, xp2.z, ym2.z, ym1.z, yp1.z, yp2.z, c.x, c.y, c.w,
                    rx, up.z, mc.z);
  o.w = update_cell(k, c.w, xm2.w, xm1.w, xp1.w, xp2.w, ym2.w, ym1.w, yp1.w, yp2.w, c.y, c.z, rx,
                    ry, up.w, mc.w);
  if (store_z) st_stream<NT>(reinterpret_cast<float *>(reinterpret_cast<char *>(a.u2 + row) + voff), o);
}

// ---------------------------------------------------------------------------------
// TILE3D: the classic cubic LDS tile — an 8x8x8 workgroup (512 threads, one output cell
// each) stages the 12x12x12 box around its outputs (radius-2 halo on every face) in
// LDS and reads all 13 stencil points from there; u1 and m come straight from global
// memory.  One-shot, any shape.  Kept as a selectable variant and as a measurement
// point: an 8-cell z-extent means 32-byte global segments, so it moves many more cache
// lines per output than the marching kernel.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void stencil_tile3d_kernel(StencilArgs a, int ntz, int nty) {
  __shared__ float tile[12][12][13];  // z padded to 13 words: conflict-free y/x column reads
  const int tid = threadIdx.x;
  const int t = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int x0 = a.x_m + (t / (ntz * nty)) * 8;
  const int y0 = a.y_m + ((t / ntz) % nty) * 8;
  const int z0 = a.z_m + (t % ntz) * 8;
  const float *__restrict__ u0 = a.u0;
  for (int i = tid; i < 12 * 12 * 12; i += 512) {
    const int lz = i % 12, ly = (i / 12) % 12, lx = i / 144;
    // padded coordinates, clamped into the array for partial tiles at the far faces
    const int X = min(x0 - 2 + lx + FDTD_HALO, a.nxp - 1);
    const int Y = min(y0 - 2 + ly + FDTD_HALO, a.nyp - 1);
    const int Z = min(z0 - 2 + lz + FDTD_HALO, a.nzp - 1);
    tile[lx][ly][lz] = u0[(long)X * a.sx + (long)Y * a.sy + Z];
  }
  __syncthreads();
  const int lz = tid & 7, ly = (tid >> 3) & 7, lx = tid >> 6;
  const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
  if (x > a.x_M || y > a.y_M || z > a.z_M) return;
  const Coef k = load_coef(a);
  const long c = (long)(x + FDTD_HALO) * a.sx + (long)(y + FDTD_HALO) * a.sy + (z + FDTD_HALO);
  const int cx = lx + 2, cy = ly + 2, cz = lz + 2;
  a.u2[c] = update_cell(k, tile[cx][cy][cz], tile[cx - 2][cy][cz], tile[cx - 1][cy][cz],
                        tile[cx + 1][cy][cz], tile[cx + 2][cy][cz], tile[cx][cy - 2][cz],
                        tile[cx][cy - 1][cz], tile[cx][cy + 1][cz], tile[cx][cy + 2][cz],
                        tile[cx][cy][cz - 2], tile[cx][cy][cz - 1], tile[cx][cy][cz + 1],
                        tile[cx][cy][cz + 2], a.u1[c], a.m[c]);
}

// ---------------------------------------------------------------------------------
// section1, atomic form (more than 128 sources): one thread per (source, corner).  Weights,
// bounds test and amplitude are the expressions of openacc.cpp:185-196; the += is a
// device-scope float atomic (cuda.cu:165).  Corners of one source never coincide, so with
// sources that do not share cells the result does not depend on the order the atomics land in.
// ---------------------------------------------------------------------------------
// target_mode 1 keeps only the corners that land outside the updated box, i.e. in the first halo
// layer (openacc.cpp:192 admits one cell beyond every face); 2 keeps only those inside it.  The
// box is the whole domain's: gx_m..gx_M are its planes in the coordinates of the source positions.
__device__ __forceinline__ bool inject_target_selected(const InjectArgs &a, int x, int y, int z) {
  if (a.target_mode == 0) return true;
  const bool inside = x >= a.gx_m && x <= a.gx_M && y >= a.y_m && y <= a.y_M && z >= a.z_m && z <= a.z_M;
  return a.target_mode == 1 ? !inside : inside;
}

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

  if (rx + posx >= a.x_lo && ry + posy >= a.y_m - 1 && rz + posz >= a.z_m - 1 &&
      rx + posx <= a.x_hi && ry + posy <= a.y_M + 1 && rz + posz <= a.z_M + 1 &&
      inject_target_selected(a, rx + posx, ry + posy, rz + posz)) {
    const float r0 = 1.0e-2F * (rx * px + (1 - rx) * (1 - px)) * (ry * py + (1 - ry) * (1 - py)) *
                     (rz * pz + (1 - rz) * (1 - pz)) * a.src[(long)a.time * a.src_stride + p] /
                     a.m[(long)(posx - a.x_shift + a.halo) * a.sx + (long)(posy + a.halo) * a.sy +
                         (posz + a.halo)];
    atomicAdd(a.u2 + (long)(rx + posx - a.x_shift + a.halo) * a.sx + (long)(ry + posy + a.halo) * a.sy +
                  (rz + posz + a.halo),
              r0);
  }
}

// Ordered form of the scatter for up to 128 sources (1024 corners), one workgroup.
// Float addition is not associative, so when corners of different sources fall on the same
// cell the atomics above give a sum that depends on arrival order.  Here every corner is
// staged in LDS; the first corner (in the reference's loop order: source, then x, y, z
// corner) that targets a cell adds all corners of that cell to it one after the other in that
// order, with plain loads and stores — exactly the serial sequence of openacc.cpp:176-196.
__global__ __launch_bounds__(1024) void source_inject_ordered_kernel(InjectArgs a) {
  __shared__ long s_cell[1024];
  __shared__ float s_val[1024];
  const int tid = threadIdx.x;
  const int n = (a.p_src_M - a.p_src_m + 1) * 8;
  long cell = -1;
  float r0 = 0.f;
  if (tid < n) {
    const int p = a.p_src_m + (tid >> 3);
    const int rx = (tid >> 2) & 1, ry = (tid >> 1) & 1, rz = tid & 1;
    const float gx = (-a.o_x + a.coords[(long)p * a.coord_stride + 0]) / a.h_x;
    const float gy = (-a.o_y + a.coords[(long)p * a.coord_stride + 1]) / a.h_y;
    const float gz = (-a.o_z + a.coords[(long)p * a.coord_stride + 2]) / a.h_z;
    const int posx = (int)floorf(gx), posy = (int)floorf(gy), posz = (int)floorf(gz);
    const float px = -floorf(gx) + gx, py = -floorf(gy) + gy, pz = -floorf(gz) + gz;
    if (rx + posx >= a.x_lo && ry + posy >= a.y_m - 1 && rz + posz >= a.z_m - 1 &&
        rx + posx <= a.x_hi && ry + posy <= a.y_M + 1 && rz + posz <= a.z_M + 1 &&
        inject_target_selected(a, rx + posx, ry + posy, rz + posz)) {
      r0 = 1.0e-2F * (rx * px + (1 - rx) * (1 - px)) * (ry * py + (1 - ry) * (1 - py)) *
           (rz * pz + (1 - rz) * (1 - pz)) * a.src[(long)a.time * a.src_stride + p] /
           a.m[(long)(posx - a.x_shift + a.halo) * a.sx + (long)(posy + a.halo) * a.sy +
               (posz + a.halo)];
      cell = (long)(rx + posx - a.x_shift + a.halo) * a.sx + (long)(ry + posy + a.halo) * a.sy +
             (rz + posz + a.halo);
    }
  }
  s_cell[tid] = cell;
  s_val[tid] = r0;
  __syncthreads();
  if (cell < 0) return;
  for (int j = 0; j < tid; ++j)
    if (s_cell[j] == cell) return;  // an earlier corner owns this cell
  float v = a.u2[cell] + r0;
  for (int j = tid + 1; j < n; ++j)
    if (s_cell[j] == cell) v += s_val[j];
  a.u2[cell] = v;
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
         (((uintptr_t)m | (uintptr_t)u) % 16 == 0) && (((long)g->nxp * g->nyp * g->nzp) % 4 == 0);
}

static StencilArgs make_args(const fdtd_geom *g, const float *m, const float *u0, const float *u1, float *u2) {
  StencilArgs a;
  a.m = m;
  a.u0 = u0;
  a.u1 = u1;
  a.u2 = u2;
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

// Launch shape of the marching kernel: `chunks` x-chunks of `xchunk` planes per launch.
//
// The kernel keeps one workgroup per CU resident, so a launch runs in rounds of 256 workgroups,
// and it is fastest when the workgroups of a launch advance plane by plane together (tiles then
// find their neighbours' rows in L2).  Measured on MI355X: marches longer than ~128 planes drift
// apart (1024^3 in one launch: 294 Gcell/s; as 16 launches of 64 planes: 343), so long grids are
// cut into several launches.  Among the candidates the score prefers full rounds, then longer
// chunks (each chunk re-reads 4 planes of u0 to prime its register column).
struct MarchShape {
  int xchunk, chunks;
};
static MarchShape pick_march_shape(int ex, long tiles_per_plane) {
  MarchShape ms;
  const int forced = env_int("FDTD_XCHUNK", 0);
  if (forced > 0) {
    ms.xchunk = forced > ex ? ex : forced;
    ms.chunks = env_int("FDTD_CHUNKS_PER_LAUNCH", (ex + ms.xchunk - 1) / ms.xchunk);
    return ms;
  }
  const long cus = 256;
  // planes with many tiles (>= ~100) run several rounds per launch and prefer shorter marches
  const int lcap = env_int("FDTD_MARCH_LCAP", tiles_per_plane >= 100 ? 64 : 128);
  double best_score = -1.0;
  ms.xchunk = ex < lcap ? ex : lcap;
  ms.chunks = 1;
  for (int c = 1; c <= 64; ++c) {
    int xc = (ex + c - 1) / c;
    if (xc > lcap) xc = lcap;
    if (xc < 8 && c > 1) break;
    const long blocks = (long)c * tiles_per_plane;
    const long rounds = (blocks + cus - 1) / cus;
    const double fill = (double)blocks / (double)(rounds * cus);
    // each chunk re-reads 4 planes of u0 and fills/drains its pipeline: worth ~6 planes of marching
    const double score = fill * (double)xc / (double)(xc + 6);
    if (score > best_score + 1e-9) { best_score = score; ms.xchunk = xc; ms.chunks = c; }
  }
  // even out the launches
  const long per_launch = (long)ms.chunks * ms.xchunk;
  const long launches = (ex + per_launch - 1) / per_launch;
  const long planes = (ex + launches - 1) / launches;
  ms.xchunk = (int)((planes + ms.chunks - 1) / ms.chunks);
  if (ms.xchunk < 1) ms.xchunk = 1;
  return ms;
}

template <typename ST, int RY, int NW, bool NT, int LPR>
static void launch_march_shape(const StencilArgs &a0, const void *u0, const void *u1, void *u2, hipStream_t s) {
  constexpr int TZ = LPR * ST::CELLS;
  constexpr int TY = RY * NW * (64 / LPR);
  const int ex = a0.x_M - a0.x_m + 1, ey = a0.y_M - a0.y_m + 1, ez = a0.z_M - a0.z_m + 1;
  const int nzt = (ez + TZ - 1) / TZ;
  const int nyt = (ey + TY - 1) / TY;
  const MarchShape ms = pick_march_shape(ex, (long)nzt * nyt);
  const bool full = (ez % TZ == 0) && (ey % TY == 0);
  MarchArgs ma;
  ma.u0 = (const char *)u0; ma.u1 = (const char *)u1; ma.u2 = (char *)u2; ma.m = (const char *)a0.m;
  ma.f = a0;
  const long per_launch = (long)ms.chunks * ms.xchunk;
  for (long x0 = a0.x_m; x0 <= a0.x_M; x0 += per_launch) {
    ma.f.x_m = (int)x0;
    ma.f.x_M = (int)((x0 + per_launch - 1 < a0.x_M) ? x0 + per_launch - 1 : a0.x_M);
    const int nxc = (ma.f.x_M - ma.f.x_m + ms.xchunk) / ms.xchunk;
    const long nblocks = (long)nzt * nyt * nxc;
    if (full)
      hipLaunchKernelGGL((stencil_march_lds_kernel<ST, RY, NW, NT, true, LPR>), dim3((unsigned)nblocks),
                         dim3(NW * 64), 0, s, ma, ms.xchunk, nzt, nyt);
    else
      hipLaunchKernelGGL((stencil_march_lds_kernel<ST, RY, NW, NT, false, LPR>), dim3((unsigned)nblocks),
                         dim3(NW * 64), 0, s, ma, ms.xchunk, nzt, nyt);
  }
}

// Narrower row groups (LPR = 32, 16) are instantiated for the default shape of each storage type
// only: 512 -> 64 lanes per row, 384 or 640 -> 32, 320 or 576 -> 16.
template <typename ST, int RY, int NW, bool NT>
static void launch_march_lds(const StencilArgs &a, const void *u0, const void *u1, void *u2, hipStream_t s) {
  constexpr bool has_narrow = (ST::CELLS == 4 && RY == 2 && NW == 8) || (ST::CELLS == 8 && RY == 1 && NW == 8);
  if constexpr (has_narrow) {
    const int ez = a.z_M - a.z_m + 1;
    auto padded = [ez](int lpr) {
      const long tz = (long)lpr * ST::CELLS;
      return (ez + tz - 1) / tz * tz;
    };
    // 32 lanes per row as soon as it pads less than 64; 16 (more row groups per wave, shorter
    // contiguous segments) only when it saves another 10 % on an extent of at least one full tile
    int best = 64;
    if (padded(32) < padded(64)) best = 32;
    if (ez >= 64 * ST::CELLS && 10 * padded(16) <= 9 * padded(best)) best = 16;
    const int lpr = env_int("FDTD_MARCH_LPR", best);
    if (lpr == 32) { launch_march_shape<ST, RY, NW, NT, 32>(a, u0, u1, u2, s); return; }
    if (lpr == 16) { launch_march_shape<ST, RY, NW, NT, 16>(a, u0, u1, u2, s); return; }
  }
  launch_march_shape<ST, RY, NW, NT, 64>(a, u0, u1, u2, s);
}

template <typename ST>
static int launch_march_any(const StencilArgs &a, const void *u0, const void *u1, void *u2, hipStream_t s) {
  // rows per wave x waves per workgroup.  Measured fastest at 512^3: 2 x 8 for binary32 storage;
  // binary16 storage moves 8 cells per lane (twice the registers per row), so 1 x 8 there.
  const int ry = env_int("FDTD_MARCH_RY", ST::CELLS == 4 ? 2 : 1), nw = env_int("FDTD_MARCH_NW", 8);
  const bool nt = env_int("FDTD_NT", 1) != 0;  // non-temporal hints on u1, m, u2
#define FDTD_TRY(RY_, NW_)                                                              \
  if (ry == RY_ && nw == NW_) {                                                         \
    nt ? launch_march_lds<ST, RY_, NW_, true>(a, u0, u1, u2, s)                         \
       : launch_march_lds<ST, RY_, NW_, false>(a, u0, u1, u2, s);                       \
    return 0;                                                                           \
  }
  FDTD_TRY(2, 8)
  FDTD_TRY(1, 8)
  if constexpr (ST::CELLS == 4) {  // 16 waves leave 128 registers each: binary32 storage only
    FDTD_TRY(1, 16)
  }
  FDTD_TRY(4, 4)
  FDTD_TRY(2, 4)
#undef FDTD_TRY
  return (int)hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------
// Small grids: sweep and scatter in one launch.  Below a few million cells a step is two
// launches of a few microseconds each and the launches are what takes the time, so the naive
// sweep adds the source corners itself: the thread that owns a cell adds, after the update, the
// corner of every source whose 2 x 2 x 2 corner block contains the cell, in source order
// (openacc.cpp:176-196 visits sources in that order; one source has at most one corner per cell).
// Only for sources whose corners all lie inside the updated box (the caller checks): corners in
// the halo layer are not cells of any thread.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void small_prepare_kernel(InjectArgs a, FdtdSmallSource *tab) {
  const int i = threadIdx.x;
  if (i > a.p_src_M - a.p_src_m) return;
  const int p = a.p_src_m + i;
  const float gx = (-a.o_x + a.coords[(long)p * a.coord_stride + 0]) / a.h_x;
  const float gy = (-a.o_y + a.coords[(long)p * a.coord_stride + 1]) / a.h_y;
  const float gz = (-a.o_z + a.coords[(long)p * a.coord_stride + 2]) / a.h_z;
  FdtdSmallSource t;
  t.x = (int)floorf(gx); t.y = (int)floorf(gy); t.z = (int)floorf(gz);
  t.px = -floorf(gx) + gx; t.py = -floorf(gy) + gy; t.pz = -floorf(gz) + gz;
  t.mc = a.m[(long)(t.x + FDTD_HALO) * a.sx + (long)(t.y + FDTD_HALO) * a.sy + (t.z + FDTD_HALO)];
  tab[i] = t;
}

__global__ __launch_bounds__(256) void stencil_small_kernel(StencilArgs a, const FdtdSmallSource *__restrict__ tab,
                                                            const float *__restrict__ src_row, int nsrc) {
  const int z = a.z_m + blockIdx.x * 64 + threadIdx.x;
  const int y = a.y_m + blockIdx.y * 4 + threadIdx.y;
  const int x = a.x_m + blockIdx.z;
  if (z > a.z_M || y > a.y_M) return;
  const Coef k = load_coef(a);
  const long c = (long)(x + FDTD_HALO) * a.sx + (long)(y + FDTD_HALO) * a.sy + (z + FDTD_HALO);
  float v = update_from_global(k, a, c);
  for (int i = 0; i < nsrc; ++i) {   // wave-uniform table reads
    const FdtdSmallSource t = tab[i];
    const unsigned rx = (unsigned)(x - t.x), ry = (unsigned)(y - t.y), rz = (unsigned)(z - t.z);
    if (rx < 2u && ry < 2u && rz < 2u) {
      // openacc.cpp:194, evaluated left to right
      const float r0 = 1.0e-2F * ((int)rx * t.px + (1 - (int)rx) * (1 - t.px)) *
                       ((int)ry * t.py + (1 - (int)ry) * (1 - t.py)) * ((int)rz * t.pz + (1 - (int)rz) * (1 - t.pz)) *
                       src_row[i] / t.mc;
      v += r0;
    }
  }
  a.u2[c] = v;
}

// Small and medium grids, rows packed into waves: one lane = one float4 (4 cells in z) of one (x, y)
// row, a wave = 64 / LPR rows of LPR float4s each, a workgroup = 4 waves on consecutive rows of one
// plane; nothing loops.  With LPR = 32 (16) rows of 128 (64) cells fill every lane, where the
// one-cell-per-thread kernel needs four times the waves and fifteen scalar loads per cell.  The nine
// float4 of u0 a lane needs in x and y are aligned 16-byte loads, the z-neighbours come from the
// adjacent lanes (DPP) or, at the ends of a row, from 8-byte edge loads; lanes beyond the end of a
// row repeat its last float4 and store nothing.  The source corners are added by the lane that owns
// their cell, in source order, exactly as stencil_small_kernel does.
template <int LPR>
__global__ __launch_bounds__(256) void stencil_rows_kernel(StencilArgs a, const FdtdSmallSource *__restrict__ tab,
                                                           const float *__restrict__ src_row, int nsrc) {
  constexpr int RPW = 64 / LPR;   // rows per wave
  const int lane = threadIdx.x & 63;
  const int zl = lane % LPR;
  const int wave = threadIdx.x >> 6;
  const int y = a.y_m + ((int)blockIdx.y * 4 + wave) * RPW + lane / LPR;
  const int x = a.x_m + (int)blockIdx.z;
  const int zfirst = a.z_m + ((int)blockIdx.x * LPR + zl) * 4;
  const bool owned = zfirst <= a.z_M && y <= a.y_M;
  const int zload = min(zfirst, a.z_M - 3), yload = min(y, a.y_M);
  const bool first = zl == 0, last = zl == LPR - 1 || zfirst + 4 > a.z_M;
  const Coef k = load_coef(a);
  const long sx = a.sx, sy = a.sy;
  const long c0 = (long)(x + FDTD_HALO) * sx + (long)(yload + FDTD_HALO) * sy + (zload + FDTD_HALO);
  const float *__restrict__ u0 = a.u0;
#define ROW4(off) (*reinterpret_cast<const float4 *>(u0 + c0 + (off)))
  float4 c = ROW4(0);
  const float4 xm2 = ROW4(-2 * sx), xm1 = ROW4(-sx), xp1 = ROW4(sx), xp2 = ROW4(2 * sx);
  const float4 ym2 = ROW4(-2 * sy), ym1 = ROW4(-sy), yp1 = ROW4(sy), yp2 = ROW4(2 * sy);
#undef ROW4
  const float4 up = *reinterpret_cast<const float4 *>(a.u1 + c0);
  const float4 mc = *reinterpret_cast<const float4 *>(a.m + c0);
  // every lane issues both edge loads (no memory instruction under a divergent branch); only the
  // first and the last lane of a row use them
  const float2 eL = *reinterpret_cast<const float2 *>(u0 + c0 - 2);
  const float2 eR = *reinterpret_cast<const float2 *>(u0 + c0 + 4);
  asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w));
  float lz = dpp_from_prev_lane(eL.x, c.z), lw = dpp_from_prev_lane(eL.y, c.w);
  float rx = dpp_from_next_lane(eR.x, c.x), ry = dpp_from_next_lane(eR.y, c.y);
  lz = first ? eL.x : lz; lw = first ? eL.y : lw;
  rx = last ? eR.x : rx; ry = last ? eR.y : ry;
  float o[4];
  o[0] = update_cell(k, c.x, xm2.x, xm1.x, xp1.x, xp2.x, ym2.x, ym1.x, yp1.x, yp2.x, lz, lw, c.y, c.z, up.x, mc.x);
  o[1] = update_cell(k, c.y, xm2.y, xm1.y, xp1.y, xp2.y, ym2.y, ym1.y, yp1.y, yp2.y, lw, c.x, c.z, c.w, up.y, mc.y);
  o[2] = update_cell(k, c.z, xm2.z, xm1.z, xp1.z, xp2.z, ym2.z, ym1.z, yp1.z, yp2.z, c.x, c.y, c.w, rx, up.z, mc.z);
  o[3] = update_cell(k, c.w, xm2.w, xm1.w, xp1.w, xp2.w, ym2.w, ym1.w, yp1.w, yp2.w, c.y, c.z, rx, ry, up.w, mc.w);
  for (int i = 0; i < nsrc; ++i) {   // wave-uniform table reads
    const FdtdSmallSource t = tab[i];
    const unsigned dx = (unsigned)(x - t.x), dy = (unsigned)(y - t.y);
    if (dx < 2u && dy < 2u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned dz = (unsigned)(zfirst + j - t.z);
        if (dz < 2u) {
          // openacc.cpp:194, evaluated left to right
          const float r0 = 1.0e-2F * ((int)dx * t.px + (1 - (int)dx) * (1 - t.px)) *
                           ((int)dy * t.py + (1 - (int)dy) * (1 - t.py)) * ((int)dz * t.pz + (1 - (int)dz) * (1 - t.pz)) *
                           src_row[i] / t.mc;
          o[j] += r0;
        }
      }
    }
  }
  if (owned) *reinterpret_cast<float4 *>(a.u2 + c0) = make_float4(o[0], o[1], o[2], o[3]);
}

// Can the packed-rows kernel take this box?  Whole float4s in z, 16-byte aligned rows.
static bool rows_eligible(const fdtd_geom *g, const float *m, const float *u0, const float *u1, const float *u2) {
  const int ez = g->z_M - g->z_m + 1;
  return ez >= 4 && ez % 4 == 0 && g->nzp % 4 == 0 && g->z_m % 4 == 0 &&
         (((uintptr_t)m | (uintptr_t)u0 | (uintptr_t)u1 | (uintptr_t)u2) % 16) == 0 &&
         (((long)g->nyp * g->nzp) % 4) == 0;
}

static void launch_rows(const StencilArgs &a, const FdtdSmallSource *table, const float *src_row, int nsrc, hipStream_t s) {
  const int ex = a.x_M - a.x_m + 1, ey = a.y_M - a.y_m + 1, ez = a.z_M - a.z_m + 1;
  // Lanes per row.  Measured on MI355X (us per step, cubes): whole waves per row win from 144 cells on even
  // with idle lanes (144^3: 11.5 with 64 lanes per row, 15.5 with 32, 19.7 with 16; 192^3: 19.9 / 24.6 / 28.2),
  // half-waves at 128 and 96 (128^3: 9.1 / 6.9 / 8.8; 96^3: 7.1 / 5.7 / 6.0); below that a step is a launch.
  const int lpr = env_int("FDTD_ROWS_LPR", ez > 128 ? 64 : (ez > 64 ? 32 : 16));
  const int rows_per_wg = 4 * (64 / lpr);
  const dim3 grid((ez + 4 * lpr - 1) / (4 * lpr), (ey + rows_per_wg - 1) / rows_per_wg, ex), block(256);
  if (lpr == 16) hipLaunchKernelGGL(stencil_rows_kernel<16>, grid, block, 0, s, a, table, src_row, nsrc);
  else if (lpr == 32) hipLaunchKernelGGL(stencil_rows_kernel<32>, grid, block, 0, s, a, table, src_row, nsrc);
  else hipLaunchKernelGGL(stencil_rows_kernel<64>, grid, block, 0, s, a, table, src_row, nsrc);
}

static long g_small_steps = 0;

// Fills the table of the sources p_src_m .. p_src_M (positions, weights, m at the position).
int fdtd_small_prepare(const struct fdtd_geom *g, const float *m, const float *src_coords, int p_src_m, int p_src_M,
                       int coord_stride, FdtdSmallSource *table, void *stream) {
  if (!geom_valid(g) || !m || !src_coords || !table || p_src_M - p_src_m + 1 > FDTD_SMALL_MAX_SOURCES)
    return (int)hipErrorInvalidValue;
  InjectArgs a = {};
  a.m = m; a.coords = src_coords;
  a.sx = (long)g->nyp * g->nzp; a.sy = g->nzp;
  a.h_x = g->h_x; a.h_y = g->h_y; a.h_z = g->h_z; a.o_x = g->o_x; a.o_y = g->o_y; a.o_z = g->o_z;
  a.p_src_m = p_src_m; a.p_src_M = p_src_M; a.coord_stride = coord_stride;
  hipLaunchKernelGGL(small_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, table);
  return (int)hipGetLastError();
}

// One step: u2 = update(u0, u1, m) with the corners of the table's sources added, amplitudes src_row[0 .. nsrc).
int fdtd_small_step(const struct fdtd_geom *g, const float *m, const float *u0, const float *u1, float *u2,
                    const FdtdSmallSource *table, const float *src_row, int nsrc, void *stream) {
  if (!geom_valid(g) || !m || !u0 || !u1 || !u2 || (nsrc > 0 && (!table || !src_row))) return (int)hipErrorInvalidValue;
  const int ex = g->x_M - g->x_m + 1, ey = g->y_M - g->y_m + 1, ez = g->z_M - g->z_m + 1;
  if (ex <= 0 || ey <= 0 || ez <= 0) return 0;
  if (ex > 65535 || (ey + 3) / 4 > 65535) return (int)hipErrorInvalidValue;
  const StencilArgs a = make_args(g, m, u0, u1, u2);
  if (rows_eligible(g, m, u0, u1, u2) && env_int("FDTD_SMALL_ROWS", 1) != 0)
    launch_rows(a, table, src_row, nsrc, (hipStream_t)stream);
  else
    hipLaunchKernelGGL(stencil_small_kernel, dim3((ez + 63) / 64, (ey + 3) / 4, ex), dim3(64, 4, 1), 0,
                       (hipStream_t)stream, a, table, src_row, nsrc);
  __atomic_add_fetch(&g_small_steps, 1, __ATOMIC_RELAXED);
  return (int)hipGetLastError();
}

// Statistics: sweep-and-scatter launches issued by this process so far.
extern "C" long fdtd_hip_small_step_count(void) { return __atomic_load_n(&g_small_steps, __ATOMIC_RELAXED); }

// Does the automatic choice run this box with a one-shot kernel (packed rows, or one cell per thread)
// rather than the marching kernel?  Below FDTD_MARCH_MIN_CELLS (16 M) cells: measured per step, packed
// rows against marching: 192^3 19.9 against 30 us, 224^3 29.3 against 43.8, 240^3 41.7 against 45; at
// 256^3 the marching kernel and the two-step sweep are ahead (55 against 51 us).
bool fdtd_auto_is_naive(const struct fdtd_geom *g, const float *m, const float *u) {
  const int ex = g->x_M - g->x_m + 1, ey = g->y_M - g->y_m + 1, ez = g->z_M - g->z_m + 1;
  const long cells = (long)ex * ey * ez;
  return !(march_eligible(g, m, u) && cells >= env_int("FDTD_MARCH_MIN_CELLS", 16000000) &&
           ex >= env_int("FDTD_MARCH_MIN_PLANES", 8));
}

// One sweep on explicit level pointers (the levels need not be the three slots of one array).
int fdtd_stencil_levels(const struct fdtd_geom *g, const float *m, const float *u0, const float *u1,
                        float *u2, int variant, void *stream) {
  if (!geom_valid(g) || !m || !u0 || !u1 || !u2) return (int)hipErrorInvalidValue;
  const int ex = g->x_M - g->x_m + 1, ey = g->y_M - g->y_m + 1, ez = g->z_M - g->z_m + 1;
  if (ex <= 0 || ey <= 0 || ez <= 0) return 0;  // empty range: nothing to update
  hipStream_t s = (hipStream_t)stream;
  const StencilArgs a = make_args(g, m, u0, u1, u2);
  const bool eligible = march_eligible(g, m, u0) && march_eligible(g, u1, u2);
  if (variant == FDTD_VARIANT_MARCH2) variant = FDTD_VARIANT_AUTO;   // a single sweep cannot fuse

  if (variant == FDTD_VARIANT_AUTO) {
    // Measured crossover on MI355X: below 16 M cells the grid is served from the caches and a one-shot
    // kernel (no prologue, many workgroups) wins; from 256^3 on the marching kernel does.
    const long cells = (long)ex * ey * ez;
    // Ranges of fewer than 8 planes (the boundary planes of a slab) would spend more on priming
    // the register column than on the sweep: naive kernel as well.
    variant = (eligible && cells >= env_int("FDTD_MARCH_MIN_CELLS", 16000000) &&
               ex >= env_int("FDTD_MARCH_MIN_PLANES", 8))
                  ? FDTD_VARIANT_MARCH : FDTD_VARIANT_ROWS;
  }
  // one-shot sweeps: float4 lanes with rows packed into waves where the shape allows, else one cell per thread
  if (variant == FDTD_VARIANT_ROWS && (!rows_eligible(g, m, u0, u1, u2) || env_int("FDTD_SMALL_ROWS", 1) == 0))
    variant = FDTD_VARIANT_NAIVE;
  if ((variant == FDTD_VARIANT_MARCH || variant == FDTD_VARIANT_VEC) && !eligible)
    variant = FDTD_VARIANT_NAIVE;

  switch (variant) {
    case FDTD_VARIANT_FLAT: {
      const long total = (long)ex * ey * ez;
      long blocks = (total + 255) / 256;
      if (blocks > 256L * 64) blocks = 256L * 64;  // grid-stride beyond 64 blocks per CU
      hipLaunchKernelGGL(stencil_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
      break;
    }
    case FDTD_VARIANT_ROWS: {
      if (ex > 65535) return (int)hipErrorInvalidValue;
      launch_rows(a, nullptr, nullptr, 0, s);
      break;
    }
    case FDTD_VARIANT_NAIVE: {
      if (ex > 65535 || (ey + 3) / 4 > 65535) return (int)hipErrorInvalidValue;
      dim3 grid((ez + 63) / 64, (ey + 3) / 4, ex);
      hipLaunchKernelGGL(stencil_naive_kernel, grid, dim3(64, 4, 1), 0, s, a);
      break;
    }
    case FDTD_VARIANT_VEC: {
      const int nzt = (ez + 255) / 256;
      const int nyb = (ey + 3) / 4;
      const int yb_per_xcd = (nyb + 7) / 8;
      const long nblocks = 8L * yb_per_xcd * nzt * ex;
      if (nblocks > 0x7fffffffL) return (int)hipErrorInvalidValue;
      if (env_int("FDTD_NT", 1))
        hipLaunchKernelGGL((stencil_vec_kernel<true>), dim3((unsigned)nblocks), dim3(256), 0, s, a, nzt,
                           yb_per_xcd, nyb);
      else
        hipLaunchKernelGGL((stencil_vec_kernel<false>), dim3((unsigned)nblocks), dim3(256), 0, s, a, nzt,
                           yb_per_xcd, nyb);
      break;
    }
    case FDTD_VARIANT_TILE3D: {
      const int ntx = (ex + 7) / 8, nty = (ey + 7) / 8, ntz = (ez + 7) / 8;
      const long nblocks = (long)ntx * nty * ntz;
      if (nblocks > 0x7fffffffL) return (int)hipErrorInvalidValue;
      hipLaunchKernelGGL(stencil_tile3d_kernel, dim3((unsigned)nblocks), dim3(512), 0, s, a, ntz, nty);
      break;
    }
    case FDTD_VARIANT_MARCH: {
      const int rc = launch_march_any<StoreF32>(a, a.u0, a.u1, a.u2, s);
      if (rc) return rc;
      break;
    }
    default:
      return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}

extern "C" int fdtd_hip_stencil_step(const struct fdtd_geom *g, const float *m, float *u, int t0,
                                     int t1, int t2, int variant, void *stream) {
  if (!g || !u) return (int)hipErrorInvalidValue;
  if (t0 < 0 || t0 > 2 || t1 < 0 || t1 > 2 || t2 < 0 || t2 > 2) return (int)hipErrorInvalidValue;
  const long vol = (long)g->nxp * g->nyp * g->nzp;
  return fdtd_stencil_levels(g, m, u + (long)t0 * vol, u + (long)t1 * vol, u + (long)t2 * vol, variant, stream);
}

// The scatter on an explicit level pointer.  target_mode 0: every corner inside the reference's
// bounds; 1: only corners whose target cell lies outside the updated box (the first halo layer).
int fdtd_inject_level(const struct fdtd_geom *g, const float *m, const float *src,
                      const float *src_coords, float *level, int time, int p_src_m, int p_src_M,
                      int src_stride, int coord_stride, int x_shift, int x_lo, int x_hi, int target_mode,
                      int gx_m, int gx_M, void *stream) {
  return fdtd_inject_level_halo(g, m, src, src_coords, level, time, p_src_m, p_src_M, src_stride, coord_stride,
                                x_shift, x_lo, x_hi, target_mode, gx_m, gx_M, FDTD_HALO, stream);
}

int fdtd_inject_level_halo(const struct fdtd_geom *g, const float *m, const float *src,
                           const float *src_coords, float *level, int time, int p_src_m, int p_src_M,
                           int src_stride, int coord_stride, int x_shift, int x_lo, int x_hi, int target_mode,
                           int gx_m, int gx_M, int halo, void *stream) {
  if (p_src_M < p_src_m) return 0;
  if (!g || !m || !src || !src_coords || !level || coord_stride < 3 || src_stride < 1)
    return (int)hipErrorInvalidValue;
  if (x_hi < x_lo) return 0;
  InjectArgs a;
  a.m = m; a.src = src; a.coords = src_coords;
  a.u2 = level;
  a.target_mode = target_mode;
  a.gx_m = gx_m; a.gx_M = gx_M;
  a.halo = halo;
  a.sx = (long)g->nyp * g->nzp; a.sy = g->nzp;
  a.x_m = g->x_m; a.x_M = g->x_M; a.y_m = g->y_m; a.y_M = g->y_M; a.z_m = g->z_m; a.z_M = g->z_M;
  a.h_x = g->h_x; a.h_y = g->h_y; a.h_z = g->h_z; a.o_x = g->o_x; a.o_y = g->o_y; a.o_z = g->o_z;
  a.p_src_m = p_src_m; a.p_src_M = p_src_M; a.time = time;
  a.src_stride = src_stride; a.coord_stride = coord_stride;
  a.x_shift = x_shift; a.x_lo = x_lo; a.x_hi = x_hi;
  const long threads = (long)(p_src_M - p_src_m + 1) * 8;
  if (threads <= 1024 && env_int("FDTD_INJECT_ATOMIC", 0) == 0) {
    // one workgroup of just enough waves: a small workgroup still finds room on a CU that a
    // register-heavy sweep occupies, a 1024-thread one waits until the sweep drains
    const unsigned block = (unsigned)((threads + 63) / 64 * 64);
    hipLaunchKernelGGL(source_inject_ordered_kernel, dim3(1), dim3(block), 0, (hipStream_t)stream, a);
  }
  else
    hipLaunchKernelGGL(source_inject_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

static int launch_inject(const struct fdtd_geom *g, const float *m, const float *src,
                         const float *src_coords, float *u, int t2, int time, int p_src_m, int p_src_M,
                         int src_stride, int coord_stride, int x_shift, int x_lo, int x_hi, void *stream) {
  if (p_src_M < p_src_m) return 0;
  if (!g || !u || t2 < 0 || t2 > 2) return (int)hipErrorInvalidValue;
  const long vol = (long)g->nxp * g->nyp * g->nzp;
  return fdtd_inject_level(g, m, src, src_coords, u + (long)t2 * vol, time, p_src_m, p_src_M, src_stride,
                           coord_stride, x_shift, x_lo, x_hi, 0, 0, 0, stream);
}

extern "C" int fdtd_hip_source_inject(const struct fdtd_geom *g, const float *m, const float *src,
                                      const float *src_coords, float *u, int t2, int time,
                                      int p_src_m, int p_src_M, int src_stride, int coord_stride,
                                      void *stream) {
  if (!g) return (int)hipErrorInvalidValue;
  return launch_inject(g, m, src, src_coords, u, t2, time, p_src_m, p_src_M, src_stride, coord_stride,
                       0, g->x_m - 1, g->x_M + 1, stream);
}

extern "C" int fdtd_hip_source_inject_slab(const struct fdtd_geom *g, const float *m, const float *src,
                                           const float *src_coords, float *u, int t2, int time,
                                           int p_src_m, int p_src_M, int src_stride, int coord_stride,
                                           int x_shift, int x_lo, int x_hi, void *stream) {
  return launch_inject(g, m, src, src_coords, u, t2, time, p_src_m, p_src_M, src_stride, coord_stride,
                       x_shift, x_lo, x_hi, stream);
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
  InjectArgs f;     // u2 member unused
  __half *base;     // the [3][nxp][nyp][nzp] array: 4-byte aligned, whatever the parity of a level's size
  long level_off;   // t2 * nxp*nyp*nzp: first half of the level that is written
};

// Same scatter as source_inject_kernel; the += on a binary16 element is a compare-and-
// swap on the aligned 32-bit word that holds it (two corners can share a word).  Words are
// counted from the base of the whole array: a level of an odd number of cells starts on a
// 2-byte boundary, so words counted from the level would be misaligned.
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
  if (rx + posx >= a.x_lo && ry + posy >= a.y_m - 1 && rz + posz >= a.z_m - 1 &&
      rx + posx <= a.x_hi && ry + posy <= a.y_M + 1 && rz + posz <= a.z_M + 1) {
    const float r0 = 1.0e-2F * (rx * px + (1 - rx) * (1 - px)) * (ry * py + (1 - ry) * (1 - py)) *
                     (rz * pz + (1 - rz) * (1 - pz)) * a.src[(long)a.time * a.src_stride + p] /
                     a.m[(long)(posx - a.x_shift + FDTD_HALO) * a.sx + (long)(posy + FDTD_HALO) * a.sy +
                         (posz + FDTD_HALO)];
    const long c = (long)(rx + posx - a.x_shift + FDTD_HALO) * a.sx + (long)(ry + posy + FDTD_HALO) * a.sy +
                   (rz + posz + FDTD_HALO);
    const long idx = h.level_off + c;
    unsigned int *word = reinterpret_cast<unsigned int *>(h.base) + (idx >> 1);
    const int shift = (idx & 1) ? 16 : 0;
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
  h.f = make_args(g, m, nullptr, nullptr, nullptr);
  const long vol = (long)g->nxp * g->nyp * g->nzp;
  __half *u = (__half *)u_h16;
  h.m = m;
  h.u0 = u + (long)t0 * vol;
  h.u1 = u + (long)t1 * vol;
  h.u2 = u + (long)t2 * vol;
  // marching kernel when every lane can move 8 whole cells with aligned accesses
  const bool march_ok = (g->nzp % 4 == 0) && (g->z_m % 8 == 0) && (ez % 8 == 0) &&
                        (((uintptr_t)m | (uintptr_t)u_h16) % 16 == 0) && (vol % 8 == 0) &&
                        ((long)ex * ey * ez >= env_int("FDTD_MARCH_MIN_CELLS", 6000000)) &&
                        ex >= env_int("FDTD_MARCH_MIN_PLANES", 8);
  if (march_ok) {
    const int rc = launch_march_any<StoreF16>(h.f, h.u0, h.u1, h.u2, (hipStream_t)stream);
    if (rc) return rc;
    return (int)hipGetLastError();
  }
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
  if ((uintptr_t)u_h16 % 4) return (int)hipErrorInvalidValue;
  h.base = (__half *)u_h16;
  h.level_off = (long)t2 * vol;
  a.sx = (long)g->nyp * g->nzp; a.sy = g->nzp;
  a.x_m = g->x_m; a.x_M = g->x_M; a.y_m = g->y_m; a.y_M = g->y_M; a.z_m = g->z_m; a.z_M = g->z_M;
  a.h_x = g->h_x; a.h_y = g->h_y; a.h_z = g->h_z; a.o_x = g->o_x; a.o_y = g->o_y; a.o_z = g->o_z;
  a.p_src_m = p_src_m; a.p_src_M = p_src_M; a.time = time;
  a.src_stride = src_stride; a.coord_stride = coord_stride;
  a.x_shift = 0; a.x_lo = g->x_m - 1; a.x_hi = g->x_M + 1;
  a.target_mode = 0; a.gx_m = g->x_m; a.gx_M = g->x_M;
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
