// fdtd_march2.hip — two time steps per sweep (the reference's t_fuse knob, main.cpp:261,267, which
// no reference kernel implements).
//
// One launch advances the field by two leapfrog steps of openacc.cpp:159-164 and applies the
// source scatter of the first of them (openacc.cpp:185-196) in between, so the result is bit for
// bit what two single sweeps with a scatter after each give.  It reads B = u(t), A = u(t-1) and
// m, and writes C = u(t+1) and D = u(t+2): five level-sized streams for two updates of every
// cell, 10 B per cell update instead of 16.  D is a fourth buffer: it cannot overwrite A in
// place, because neighbouring workgroups still read A for the rows they compute redundantly.
//
// Tile: 8 output rows x 512 cells (the whole z extent) marching along x; 8 waves.  C is
// computed on 12 rows (2 redundant rows per side) from B on 16 rows; D on the 8 rows, two planes
// behind C.  Wave (q, h) owns z-half h of three row segments of 256 cells (one float4 per lane):
//   segment 0: an inner output row (2..5 of the tile): nobody else loads it, its A and m stream
//   segment 1: an outer output row (0, 1, 6, 7): the neighbouring tile recomputes C there
//   segment 2: a redundant row (-2, -1, 8, 9): C only, an outer output row of the neighbour
// x-neighbours: two 5-plane register rings (B for 3 segments, C for 2), advanced by renaming —
// the steady-state loop is unrolled 5 times.  y-neighbours: LDS (B centre planes double
// buffered, C planes in a ring of 3), one workgroup barrier per plane.  z-neighbours: DPP
// wavefront shifts inside a wave, LDS between the two halves of a row, 8-byte edge loads at the
// ends of the row.  Rows and planes of C outside the updated box are never computed: like a
// single sweep, the second step reads what the C level holds there in memory.
// Everything that comes from memory is reloaded for the next plane right after its last use,
// into the same registers (one register set, a whole iteration between a load and its use, exact
// counted waits).  Memory is addressed with buffer instructions: descriptor (scalar) + plane
// offset (scalar) + one loop-invariant 32-bit lane offset per row.
//
// Shapes: a workgroup carries one tile of 8 rows x 512 cells, two of 8 x 256 (one per wave parity) or four
// of 8 x 128 (one per half-wave).  A row is covered by z-tiles of one kind, or by a mix (tiles of 512 cells,
// then one of 256, then one of 128: stencil_march2_mixed_kernel), whichever the cost model of m2_tiling
// prefers; a cross-section that does not fill its tiles runs the MASK instantiations.  Rows must be whole
// float4s; anything else is run as two single sweeps by fdtd_hip_two_steps.  Up to 8 sources are scattered
// by the sweep itself, more by passes behind it (repair_kernel).
// Where a row has more than one z-tile, a tile cannot know its neighbour's C beyond the cut in time, so
// the four D cells around every cut (two on either side) come out wrong and are recomputed by seam_kernel
// after the sweep, from the C level in memory; the second scatter then runs as a launch of its own.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <type_traits>

#include "fdtd_device.h"

namespace {
// FDTD_WG_CLOCKS (a measurement build, `make clocks`): time stamps of every workgroup of stencil_march2_kernel, read back
// through fdtd_debug_wg_clocks (tools/wg_clocks.py).  Rows: 0 start, 1 end, 2 XCC id << 32 | tile, 3 .. 10 the last stamp
// taken in each 16-plane bucket of the steady planes, 11 prologue done, 12 lead-in done, 13 steady planes done.
#ifdef FDTD_WG_CLOCKS
__device__ unsigned long long g_wg_clock[14][1024];
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int M2_ROWS = 8;        // output rows per tile
constexpr int M2_ZCELLS = 512;    // z extent
constexpr int M2_MAX_SRC = 8;     // 8 corners each: one lane per corner
#ifndef FDTD_FUSE2_SHARE_ROWS
#define FDTD_FUSE2_SHARE_ROWS 1
#endif
#ifndef FDTD_FUSE2_PRIO
#define FDTD_FUSE2_PRIO 3
#endif
#ifndef FDTD_FUSE2_MIRROR_FIRST_TILE
#define FDTD_FUSE2_MIRROR_FIRST_TILE 1
#endif

struct Coef2 { float cm2, cm1, c0, dt2, r1, r2, r3, r4, neg2r1, dt, r0; };   // dt, r0 = 1/dt: damped form only

struct March2Args {
  const char *B, *A, *m;
  const char *damp;   // damping field of the shape of m (DAMP instantiations), else unused
  char *C, *D;
  long sx;            // cells per plane
  int sy;             // cells per row
  int x_m, x_M;       // planes of this launch (cut into x-chunks)
  int dx_m, dx_M;     // planes of the whole updated box: C outside of it is never computed
  int y_m, y_M, z_m;
  Coef2 k;
  // source scatter of the first step (time index `time`), applied to C between the two updates
  const float *src_row;   // src + time * src_stride: amplitudes of the first scatter (into C)
  const float *src_row2;  // src + (time + 1) * src_stride: second scatter (into D)
  const float *coords;
  const float *m_f;
  int nsrc, p_src_m, coord_stride;
  int z_M;
  int scatter_d;      // 0: the second scatter is left to the caller (rows cut into z-tiles)
  int keep_lo, keep_hi;   // store C on the two planes before x_m / behind x_M as well (inside the domain)
  int x_shift;        // source positions are in the coordinates of the whole domain: plane = position - x_shift
  float h_x, h_y, h_z, o_x, o_y, o_z;
};

__device__ __forceinline__ float dpp_prev(float edge, float v) {  // lane i <- lane i-1; lane 0 keeps `edge`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v),
                                                               0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_next(float edge, float v) {  // lane i <- lane i+1; lane 63 keeps `edge`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v),
                                                               0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ int xcd_contiguous_tile(int b, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, xcd = b & 7, j = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// Buffer access: aux 2 = non-temporal.  gfx9 raw buffer descriptor word 3: DATA_FORMAT = 32 bit.
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
template <bool NT>
__device__ __forceinline__ f32x4 ld4(rsrc_t r, int soff, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, NT ? 2 : 0));
}
__device__ __forceinline__ f32x2 ld2(rsrc_t r, int soff, unsigned voff) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, soff, 0));
}
// The scalar offset is folded into the lane offset and the instruction's own scalar-offset field stays zero.  With a
// register there, the compiler follows the gfx9 rule that a 16-byte buffer store needs no wait state before the next
// vector instruction overwrites its data registers, and schedules such a write directly behind the store.  On MI355X
// that lost data: in one build a ring move followed the store of u(t+2) and some lanes stored the moved values in
// the upper half of the float4 (the cells of a row's 13th to 16th lanes of every 16, run to run).  With an immediate
// zero the compiler inserts the wait state itself whenever a following instruction writes the store's registers.
template <bool NT>
__device__ __forceinline__ void st4(rsrc_t r, int soff, unsigned voff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)(voff + (unsigned)soff), 0, NT ? 2 : 0);
}

// One row segment, 4 cells per lane, evaluated on register pairs: a float4 lives in four
// consecutive registers, so its (x,y) and (z,w) halves are aligned pairs and every x/y-direction
// term maps onto packed fp32 instructions without shuffling; only the z-direction mixes the
// members of a pair.  Per cell: the operations, order and rounding of openacc.cpp:159-164.
__device__ __forceinline__ f32x2 lo2(f32x4 v) { return f32x2{v.x, v.y}; }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return f32x2{v.z, v.w}; }
// DAMP: the update with an absorbing sponge (fdtd_forward.hip; no reference arithmetic, parity unpinned):
//     u+ = dt^2 ( lap - (-2 r1 u + r1 u-) m + r0 damp u ) / (m + dt damp),   dc = damp at the cell
template <bool DAMP = false>
__device__ __forceinline__ f32x2 update_pair(const Coef2 &k, f32x2 uc, f32x2 xm2, f32x2 xm1, f32x2 xp1, f32x2 xp2,
                                             f32x2 ym2, f32x2 ym1, f32x2 yp1, f32x2 yp2, f32x2 zs2, f32x2 zs1, f32x2 up,
                                             f32x2 mc, f32x2 dc = f32x2{0, 0}) {
  // zs2 = u(z-2) + u(z+2), zs1 = u(z-1) + u(z+1)
  const f32x2 r5 = k.c0 * uc;
  const f32x2 dxx = r5 + k.cm2 * (xm2 + xp2) + k.cm1 * (xm1 + xp1);
  const f32x2 dyy = r5 + k.cm2 * (ym2 + yp2) + k.cm1 * (ym1 + yp1);
  const f32x2 dzz = r5 + k.cm2 * zs2 + k.cm1 * zs1;
  if constexpr (DAMP) {
    const f32x2 num = k.dt2 * (k.r2 * dxx + k.r3 * dyy + k.r4 * dzz - (k.neg2r1 * uc + k.r1 * up) * mc + k.r0 * dc * uc);
    const f32x2 den = mc + k.dt * dc;
    return f32x2{num.x / den.x, num.y / den.y};
  } else {
    const f32x2 num = k.dt2 * (k.r2 * dxx + k.r3 * dyy + k.r4 * dzz - (k.neg2r1 * uc + k.r1 * up) * mc);
    return f32x2{num.x / mc.x, num.y / mc.y};
  }
}
// edge_r: this lane's right neighbours are eR even though it is not lane 63 (the last lane of a row
// that ends inside the wave, or of a row part that is a half-wave).  edge_l: its left neighbours are eL
// even though it is not lane 0 (the first lane of the second half-wave).  offer: what this lane shows
// the lane before it as cells 0 and 1 (its own c.x, c.y, except in the lane behind a row that ends on
// the first lane of a row part).
template <bool DAMP = false>
__device__ __forceinline__ f32x4 update_seg(const Coef2 &k, f32x4 c, f32x4 xm2, f32x4 xm1, f32x4 xp1, f32x4 xp2,
                                            f32x4 ym2, f32x4 ym1, f32x4 yp1, f32x4 yp2, f32x2 eL, f32x2 eR, f32x4 up,
                                            f32x4 mc, bool edge_r, f32x2 offer, bool edge_l = false,
                                            f32x4 dc = f32x4{0, 0, 0, 0}) {
  f32x2 l = f32x2{dpp_prev(eL.x, c.z), dpp_prev(eL.y, c.w)};         // cells z-2, z-1 of cell 0
  f32x2 r = f32x2{dpp_next(eR.x, offer.x), dpp_next(eR.y, offer.y)}; // cells z+1, z+2 of cell 3
  l = edge_l ? eL : l;
  r = edge_r ? eR : r;
  const f32x2 cl = lo2(c), ch = hi2(c);
  // cells 0,1: z-2: (l.x, l.y)  z+2: (c.z, c.w)   z-1: (l.y, c.x)  z+1: (c.y, c.z)
  // cells 2,3: z-2: (c.x, c.y)  z+2: (r.x, r.y)   z-1: (c.y, c.z)  z+1: (c.w, r.x)
  const f32x2 zs2_lo = l + ch, zs2_hi = cl + r;
  const f32x2 zs1_lo = f32x2{l.y + c.y, c.x + c.z}, zs1_hi = f32x2{c.y + c.w, c.z + r.x};
  const f32x2 olo = update_pair<DAMP>(k, cl, lo2(xm2), lo2(xm1), lo2(xp1), lo2(xp2), lo2(ym2), lo2(ym1), lo2(yp1),
                                      lo2(yp2), zs2_lo, zs1_lo, lo2(up), lo2(mc), lo2(dc));
  const f32x2 ohi = update_pair<DAMP>(k, ch, hi2(xm2), hi2(xm1), hi2(xp1), hi2(xp2), hi2(ym2), hi2(ym1), hi2(yp1),
                                      hi2(yp2), zs2_hi, zs1_hi, hi2(up), hi2(mc), hi2(dc));
  return f32x4{olo.x, olo.y, ohi.x, ohi.y};
}

// ZH = 2: rows of 512 cells, the two waves (q, 0) and (q, 1) hold the two z-halves of the same rows.
// ZH = 1: rows of 256 cells; the workgroup then carries two independent 8-row tiles stacked in y,
// h selects the tile (each computes its own redundant rows; the LDS columns 0..63 / 64..127 are theirs).
// MASK: the last z-tile may end inside the tile (rows whose length is not a multiple of the tile
// width) and the last y-tile may have fewer than 8 rows inside the box.  Lanes beyond the end of
// the row and rows beyond y_M compute on clamped addresses and store nothing; the last lane inside
// the row takes its right neighbours from the edge loads (as lane 63 of a full row does), and a
// row beyond y_M contributes what the C level holds there in memory, like any other halo row.
// LPR = 32 (with ZH = 1): rows of 128 cells; every half-wave is a tile of its own, so the workgroup carries
// four independent 8-row tiles stacked in y (32 rows per workgroup).  Rows, and with them every row
// offset and every "row outside the box" condition, then differ between the two halves of a wave: row
// offsets move from the scalar to the per-lane part of the addresses, and the lanes 31 | 32 take their
// z-neighbours from the edge loads like the lanes 0 and 63 do.  For cross-sections whose rows fill
// multiples of 128 cells much better than multiples of 256 or 512 (384, 640).
//
// march2_tile is the work of one workgroup: the tile(s) of y-tile index yt whose rows start at cell zb, on
// x-chunk xc; cut_l / cut_r say whether another z-tile lies before / behind it (its outermost lanes then
// serve the seam pass).  The kernels below map workgroups to tiles.
typedef f32x4 (&BbufRef)[2][16][128];
typedef f32x4 (&CringRef)[3][12][128];
template <bool NT, int ZH, bool MASK, int LPR, bool DAMP, bool ALONE = true>
__device__ __forceinline__ void march2_tile(const March2Args &a, int xchunk, int zb, bool cut_l, bool cut_r, int yt,
                                            int xc, BbufRef Bbuf, CringRef Cring) {
  static_assert(LPR == 64 || (LPR == 32 && ZH == 1), "half-wave tiles exist for rows of 128 cells only");
  constexpr int TILE_CELLS = ZH == 2 ? M2_ZCELLS : LPR * 4;   // z extent of a tile
  constexpr int TILE_F4 = TILE_CELLS / 4;                       // float4s per tile row
  const int lane = threadIdx.x & 63;
  const int zl = LPR == 64 ? lane : (lane & 31);     // position inside the row part
  const int half = LPR == 64 ? 0 : (lane >> 5);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Which waves take which rows: the wave pair w = wave >> 1 takes row role q = w, except in the first y-tile of
  // the box when the tile is 512 cells wide, where the roles are mirrored (q = 3 - w).  In the first and in the last
  // y-tile two of the four roles carry a row outside the box (kept as memory holds it, not computed).  Measured on
  // MI355X with a time stamp per workgroup (FDTD_WG_CLOCKS, tools/wg_clocks.py): a workgroup whose older waves
  // (0 .. 3) have those roles takes 4-5 % longer per plane than every other workgroup of the launch, whichever end of
  // the box it sits at, and a launch lasts as long as its slowest workgroup (512^3: 586 -> 566 us per sweep).
  // Mirrored, the younger waves have them, as in the last y-tile.  Tiles of 256 and 128 cells showed no such tail.
  const bool flip = FDTD_FUSE2_MIRROR_FIRST_TILE && ZH == 2 && yt == 0;
  const int h = wave & 1, q = flip ? 3 - (wave >> 1) : (wave >> 1);
  const int yb = a.y_m + (ZH == 2 ? yt * M2_ROWS : (LPR == 64 ? (2 * yt + h) : (4 * yt + 2 * h + half)) * M2_ROWS);
  const int x_begin = a.x_m + xc * xchunk;
  const int x_end = min(x_begin + xchunk - 1, a.x_M);
  const Coef2 k = a.k;

  // rows of the three segments and their slots in the 12-row C stack (slot = row - yb + 2)
  const int yrow[3] = {yb + 2 + q, q < 2 ? yb + q : yb + 4 + q, q < 2 ? yb - 2 + q : yb + 6 + q};
  const int cslot[3] = {4 + q, q < 2 ? 2 + q : 6 + q, q < 2 ? q : 8 + q};
  const bool seg2_outside = yrow[2] < a.y_m || yrow[2] > a.y_M;
  // MASK: an output row may lie beyond y_M as well (never below y_m: tiles start at y_m)
  const bool out0 = MASK && yrow[0] > a.y_M, out1 = MASK && yrow[1] > a.y_M;
  // Half-wave tiles: these conditions differ between the two halves of a wave.  No memory instruction
  // may sit under a divergent branch (the counted waits would no longer be exact), so the loads they
  // guard are issued by the whole wave as soon as either half needs them, and selected per lane.
  const bool ld_seg2 = LPR == 64 ? seg2_outside : (__ballot(seg2_outside) != 0);
  const bool ld_out0 = LPR == 64 ? out0 : (MASK && __ballot(out0) != 0);
  const bool ld_out1 = LPR == 64 ? out1 : (MASK && __ballot(out1) != 0);
  const int yhalo = q < 2 ? yb - 4 + q : yb + 8 + q;   // B row outside the 12-row stack
  const int hslot = q < 2 ? q : 12 + q;
  // Two tiles of 256-cell rows stacked in a workgroup: rows 6, 7 of the lower tile are the redundant rows below the
  // upper one, and rows 0, 1 of the upper tile the redundant rows above the lower one.  The wave that owns such a
  // row as an outer output row writes its B and its C into the sibling tile's LDS slots as well, and the sibling's
  // wave does not load or compute that row (nor the B row two further out, which only it needed): 4 of the 8
  // waves carry two row segments instead of three.  FDTD_FUSE2_SHARE_ROWS = 0 at compile time keeps every tile
  // on its own.
  constexpr bool SHARE = ZH == 1 && LPR == 64 && FDTD_FUSE2_SHARE_ROWS;
  // (two copies of the checked form: see `checked` below.  The pairs of 256-cell tiles need the most registers; next to the
  // other tile kinds in the mixed kernel two copies of them spill, so there they keep one.)
  constexpr bool SPLIT_OK = ALONE || !(ZH == 1 && LPR == 64);
  const bool skip2 = SHARE && ((h == 0 && q >= 2) || (h == 1 && q < 2));   // wave-uniform
  const int dslot = h == 0 ? q - 2 : 10 + q;            // the sibling's C slot of this wave's outer output row
  const int dcol = h == 0 ? 64 + lane : lane;           // ... in the sibling's columns
  const int lc = h * 64 + lane;                         // this lane's column in the LDS rows
  const int lz = ZH == 2 ? lc : zl;                     // float4 index inside the row of cells
  const bool seam_lane = (lz == 0 && cut_l) || (lz == TILE_F4 - 1 && cut_r);   // next to a z-cut
  // MASK: float4s of this z-tile that are inside the row; lanes from nvalid on work on the address
  // of float4 `nvalid` (the four halo cells behind the row) and store nothing
  const int nvalid = MASK ? min(TILE_F4, (a.z_M + 1 - zb) >> 2) : TILE_F4;
  const bool lane_valid = !MASK || lz < nvalid;
  // The last lane inside the row takes its right neighbours from the edge loads.  When that lane
  // is lane 0 of its wave the edge loads are taken by its left neighbours; lane 1, which sits on
  // the halo cells behind the row, then shows it those: its B registers hold them anyway, and for
  // the C level it loads them from memory in place of its edge cells (offers_halo).
  const bool last_valid = MASK && lz == nvalid - 1 && zl != 0;
  const bool offers_halo = MASK && lz == nvalid && zl == 1;
  const bool edge_hi = (zl == LPR - 1 && lane_valid) || last_valid;
  // half-wave tiles: explicit selects where a lane shift would cross into the other tile
  const bool sel_l = LPR == 32 && zl == 0;
  const bool sel_r = last_valid || (LPR == 32 && zl == LPR - 1);
  // the row ends in this wave or before it: nothing to the right of the wave comes from LDS
  const bool row_ends_here = MASK && nvalid <= (ZH == 2 ? (h + 1) * 64 : 64);
  // Addresses: descriptor base + scalar offset (plane + row, wave-uniform) + one lane offset along z.
  // vz: the lane's 4 cells; vze: the 2 cells before them, or behind them in the last lane; vzc: the
  // same for the C array, where only the two cells before lane 0 / behind the last lane are ever
  // used, so the lanes in between repeat lane 0's address instead of touching the whole row.
  const unsigned vz0 = (unsigned)(zb + (lz - zl) * 4 + FDTD_HALO) * 4u;   // first lane of this row part
  const unsigned vz = (unsigned)(zb + (MASK ? min(lz, nvalid) : lz) * 4 + FDTD_HALO) * 4u;
  const unsigned vze = edge_hi ? vz + 16u : vz - 8u;
  const unsigned vzc = edge_hi ? vz + 16u : (offers_halo ? vz : vz0 - 8u);
  // byte offset of a segment's row inside a plane (MASK: rows clamped to the array): wave-uniform and
  // part of the scalar offset when a wave works on one tile; per lane, folded into the lane offsets,
  // when its halves work on different tiles (LPR = 32)
  int rowoff[3];
  unsigned vzs[3], vzes[3], vzcs[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int ro = ((MASK ? min(yrow[s], a.y_M + 4) : yrow[s]) + FDTD_HALO) * a.sy * 4;
    rowoff[s] = LPR == 64 ? ro : 0;
    vzs[s] = LPR == 64 ? vz : vz + (unsigned)ro;
    vzes[s] = LPR == 64 ? vze : vze + (unsigned)ro;
    vzcs[s] = LPR == 64 ? vzc : vzc + (unsigned)ro;
  }
  const int ro_h = ((MASK ? min(yhalo, a.y_M + 4) : yhalo) + FDTD_HALO) * a.sy * 4;
  const int rowoff_h = LPR == 64 ? ro_h : 0;
  const unsigned vzh = LPR == 64 ? vz : vz + (unsigned)ro_h;
  const int sxb = (int)(a.sx * 4);   // bytes per plane

  // descriptors are based at the lowest plane this workgroup touches, so that plane offsets stay
  // below 2^31 whatever the size of a level
  const int p0 = x_begin - 2, p_last = x_end + 2;
  const long base = (long)(p0 - 2 + FDTD_HALO) * a.sx * 4;
  const rsrc_t rB = make_rsrc(a.B + base, 0xffffffffu), rA = make_rsrc(a.A + base, 0xffffffffu);
  const rsrc_t rM = make_rsrc(a.m + base, 0xffffffffu);
  const rsrc_t rDm = make_rsrc((DAMP ? a.damp : a.m) + base, 0xffffffffu);   // damping field
  const rsrc_t rC = make_rsrc(a.C + base, 0xffffffffu), rD = make_rsrc(a.D + base, 0xffffffffu);
  int sp = 2 * sxb;   // byte offset of plane p

  // ---- source corners: lane l holds corner l (source l / 8, corner l % 8) with its value for the
  // scatter of the first step (into C) and of the second step (into D) ----
  bool cvalid = false;
  int cx = 0, cy = 0, cz = 0;
  float cval = 0.f, cval2 = 0.f;
  if (lane < a.nsrc * 8) {
    const int p = a.p_src_m + (lane >> 3);
    const int rx = (lane >> 2) & 1, ry = (lane >> 1) & 1, rz = lane & 1;
    const float gx = (-a.o_x + a.coords[(long)p * a.coord_stride + 0]) / a.h_x;
    const float gy = (-a.o_y + a.coords[(long)p * a.coord_stride + 1]) / a.h_y;
    const float gz = (-a.o_z + a.coords[(long)p * a.coord_stride + 2]) / a.h_z;
    const int posx = (int)floorf(gx), posy = (int)floorf(gy), posz = (int)floorf(gz);
    const float px = -floorf(gx) + gx, py = -floorf(gy) + gy, pz = -floorf(gz) + gz;
    cx = rx + posx - a.x_shift; cy = ry + posy; cz = rz + posz;
    // corners that land in the first halo layer are applied to memory before the launch; corners
    // further than 2 planes from this launch's range concern other launches (or other ranks)
    cvalid = cx >= a.dx_m && cx <= a.dx_M && cy >= a.y_m && cy <= a.y_M && cz >= a.z_m && cz <= a.z_M &&
             cx >= a.x_m - 2 && cx <= a.x_M + 2;
    if (cvalid) {
      // openacc.cpp:194 evaluates left to right: the weight product is common to both time steps
      const float w = 1.0e-2F * (rx * px + (1 - rx) * (1 - px)) * (ry * py + (1 - ry) * (1 - py)) *
                      (rz * pz + (1 - rz) * (1 - pz));
      const float mc = a.m_f[(long)(posx - a.x_shift + FDTD_HALO) * a.sx + (long)(posy + FDTD_HALO) * a.sy +
                             (posz + FDTD_HALO)];
      cval = w * a.src_row[p] / mc;
      cval2 = w * a.src_row2[p] / mc;
    }
  }
  // adds the corners of one scatter that fall on row `yr` of plane-mask `mask` to a row segment
  auto scatter_into = [&](f32x4 &v, unsigned long mask, float val, int yr) __attribute__((always_inline)) {
    while (mask) {   // wave-uniform; lane order = source order = the reference's loop order
      const int b = __builtin_ctzl(mask);
      mask &= mask - 1;
      const int sy_ = __builtin_amdgcn_readlane(cy, b), sz_ = __builtin_amdgcn_readlane(cz, b);
      const float sv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val), b));
      const int zi = sz_ - zb;
      const bool hit = sy_ == yr && (zi >> 2) == lz;
      const int comp = zi & 3;
      const f32x4 added = v + f32x4{sv, sv, sv, sv};
      v.x = (hit && comp == 0) ? added.x : v.x;
      v.y = (hit && comp == 1) ? added.y : v.y;
      v.z = (hit && comp == 2) ? added.z : v.z;
      v.w = (hit && comp == 3) ? added.w : v.w;
    }
  };

#define LD4(r, so, off) ld4<false>(r, so, off)
#define LD4S(r, so, off) ld4<NT>(r, so, off)
#define LD2(r, so, off) ld2(r, so, off)

  f32x4 bc[5][3];     // B planes p-2 .. p+2 (logical plane i lives in bc[(J+i)%5] at phase J)
  f32x4 cc[5][2];     // C planes p-4 .. p-1 and the new plane p (output rows)
  f32x4 mh[2][2];     // m planes p-2, p-1 (output rows)
  f32x4 av[3], mv[3]; // A, m of plane p
  f32x4 dv[DAMP ? 3 : 1], dh[2][DAMP ? 2 : 1];   // DAMP: the damping field of plane p, and of planes p-2, p-1 (output rows)
  f32x2 be[3];        // z-edge cells of B, plane p
  f32x4 bh, bh2;      // B row outside the stack, planes p+1 and p+2 (fetched when the neighbouring tile
                      // fetches the same row as one of its own: plane p+3)
  f32x2 ce[2];        // z-edge cells of the C array (halo of the domain), plane p-2
  f32x4 ch;           // C row of the redundant segment when that row is outside the domain, plane p
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (s == 2 && skip2) {   // the sibling tile supplies this row
#pragma unroll
      for (int i = 0; i < 5; ++i) bc[i][s] = f32x4{0, 0, 0, 0};
      av[s] = mv[s] = f32x4{0, 0, 0, 0};
      be[s] = f32x2{0, 0};
      if constexpr (DAMP) dv[s] = f32x4{0, 0, 0, 0};
      continue;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) bc[i][s] = LD4(rB, (sp + (i - 2) * sxb) + rowoff[s], vzs[s]);
    if (LPR == 64) {
      if (MASK && ((s == 0 && out0) || (s == 1 && out1))) av[s] = LD4(rC, sp + rowoff[s], vzs[s]);   // the C row itself
      else av[s] = s == 0 ? LD4S(rA, sp + rowoff[s], vzs[s]) : LD4(rA, sp + rowoff[s], vzs[s]);
    } else {
      av[s] = s == 0 ? LD4S(rA, sp + rowoff[s], vzs[s]) : LD4(rA, sp + rowoff[s], vzs[s]);
      if (MASK && ((s == 0 && ld_out0) || (s == 1 && ld_out1))) {
        const f32x4 crow = LD4(rC, sp + rowoff[s], vzs[s]);
        av[s] = (s == 0 ? out0 : out1) ? crow : av[s];
      }
    }
    mv[s] = s == 0 ? LD4S(rM, sp + rowoff[s], vzs[s]) : LD4(rM, sp + rowoff[s], vzs[s]);
    if constexpr (DAMP) dv[s] = s == 0 ? LD4S(rDm, sp + rowoff[s], vzs[s]) : LD4(rDm, sp + rowoff[s], vzs[s]);
    be[s] = LD2(rB, sp + rowoff[s], vzes[s]);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 5; ++i) cc[i][s] = f32x4{0, 0, 0, 0};
    mh[0][s] = mh[1][s] = f32x4{1, 1, 1, 1};
    if constexpr (DAMP) dh[0][s] = dh[1][s] = f32x4{0, 0, 0, 0};
    ce[s] = f32x2{0, 0};
  }
  bh = bh2 = ch = f32x4{0, 0, 0, 0};
  if (!skip2) {
    bh = LD4(rB, (sp + sxb) + rowoff_h, vzh);
    bh2 = LD4(rB, (sp + 2 * sxb) + rowoff_h, vzh);
    if (ld_seg2) ch = LD4(rC, sp + rowoff[2], vzs[2]);
  }
  {  // B(p0): own rows and the outside row
    const int buf = p0 & 1;
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (!(s == 2 && skip2)) Bbuf[buf][cslot[s] + 2][lc] = bc[2][s];
    if (skip2) Bbuf[buf][dslot + 2][dcol] = bc[2][1];
    else Bbuf[buf][hslot][lc] = LD4(rB, sp + rowoff_h, vzh);
  }
  __syncthreads();
  int ring_w = ((p0 % 3) + 3) % 3;        // ring slot written this iteration (plane p)
  int ring_r = (((p0 - 2) % 3) + 3) % 3;  // ring slot of plane p-2

  // One plane.  STEADY: plane inside the box and the chunk, second step active, no branch around
  // a memory instruction, ring phase J.  Otherwise (J = 0): every condition checked, rings
  // advanced by register moves.
  auto iteration = [&](auto phase, auto steady, auto inside, int p) __attribute__((always_inline)) {
    constexpr int J = decltype(phase)::value;
    constexpr bool STEADY = decltype(steady)::value;
#define PB(i) bc[(J + (i)) % 5]
#define PC(i) cc[(J + (i)) % 5]
    const int buf = p & 1;
    const bool more = STEADY ? true : p < p_last;
    const int sq = sp + (more ? sxb : 0);   // plane p+1 (repeats plane p at the very end)
    // inside: 1 = the plane lies inside the domain, 0 = it is one of the domain's halo planes (its u(t+1) is what memory
    // holds), 2 = decided here at run time
    constexpr int INSIDE = decltype(inside)::value;
    const bool plane_inside = STEADY || (INSIDE == 2 ? (p >= a.dx_m && p <= a.dx_M) : INSIDE == 1);
    const bool store_c = STEADY ? true
                                : (plane_inside && ((p >= x_begin && p <= x_end) || (a.keep_lo && p < a.x_m) ||
                                                    (a.keep_hi && p > a.x_M)));
    const int d = p - 2;
    const bool do_d = STEADY ? true : (d >= x_begin && d <= x_end);
    const unsigned long cmask = __ballot(cvalid && cx == p);       // corners of the first scatter in plane p
    const unsigned long dmask = __ballot(cvalid && a.scatter_d && cx == p - 2);   // second scatter, plane p-2

    auto compute = [&](auto seg) __attribute__((always_inline)) {
      constexpr int s = decltype(seg)::value;
      if constexpr (s == 2 && SHARE) {
        if (skip2) return;
      }
      const int r = cslot[s];   // the row's slot in Bbuf is r + 2
      // ---- C(p) ----
      f32x4 cnew;
      if (plane_inside) {
        const f32x4 ym2 = Bbuf[buf][r][lc], ym1 = Bbuf[buf][r + 1][lc];
        const f32x4 yp1 = Bbuf[buf][r + 3][lc], yp2 = Bbuf[buf][r + 4][lc];
        cnew = update_seg<DAMP>(k, PB(2)[s], PB(0)[s], PB(1)[s], PB(3)[s], PB(4)[s], ym2, ym1, yp1, yp2, be[s], be[s],
                                av[s], mv[s], sel_r, lo2(PB(2)[s]), sel_l, dv[DAMP ? s : 0]);
        if constexpr (s == 2) cnew = seg2_outside ? ch : cnew;
        if constexpr (MASK && s < 2) cnew = (s == 0 ? out0 : out1) ? av[s] : cnew;
        if (cmask) scatter_into(cnew, cmask, cval, yrow[s]);
      } else {   // halo plane of the domain: the intermediate level keeps what memory holds there
        cnew = LD4(rC, sp + rowoff[s], vzs[s]);
      }
      const bool row_out = s == 0 ? out0 : (s == 1 ? out1 : false);
      if constexpr (s < 2) {
        if (store_c && !row_out) {
          if (lane_valid) st4<NT>(rC, sp + rowoff[s], vzs[s], cnew);
        }
        // planes before and behind the chunk: their C exists only here, and seam_kernel needs it
        // next to the z-cuts (a chunk beside this one, or the owner of the plane, stores the same bits)
        else if (!STEADY && !store_c && !row_out && plane_inside && seam_lane) st4<false>(rC, sp + rowoff[s], vzs[s], cnew);
      }
      Cring[ring_w][r][lc] = cnew;
      if constexpr (s == 1 && SHARE) {
        if (skip2) Cring[ring_w][dslot][dcol] = cnew;   // the sibling tile's redundant row
      }
      // ---- D(p-2) ----
      if constexpr (s < 2) {
        if (do_d) {
          const f32x4 ym2 = Cring[ring_r][r - 2][lc], ym1 = Cring[ring_r][r - 1][lc];
          const f32x4 yp1 = Cring[ring_r][r + 1][lc], yp2 = Cring[ring_r][r + 2][lc];
          f32x2 eL = ce[s], eR = ce[s];
          if constexpr (ZH == 2) {   // the other z-half of the same row is in LDS
            const f32x4 nb = Cring[ring_r][r][h ? 63 : 64];
            if (h) eL = f32x2{nb.z, nb.w};
            else if (!row_ends_here) eR = f32x2{nb.x, nb.y};
          }
          f32x4 o = update_seg<DAMP>(k, PC(2)[s], PC(0)[s], PC(1)[s], PC(3)[s], cnew, ym2, ym1, yp1, yp2, eL, eR,
                                     PB(0)[s], mh[0][s], sel_r, offers_halo ? ce[s] : lo2(PC(2)[s]), sel_l,
                                     dh[0][DAMP ? s : 0]);
          if (dmask) scatter_into(o, dmask, cval2, yrow[s]);
          if (!row_out) {
            if (lane_valid) st4<NT>(rD, (sp - 2 * sxb) + rowoff[s], vzs[s], o);
          }
        }
        if (STEADY) {
          PC(4)[s] = cnew;
        } else {
          cc[0][s] = cc[1][s];
          cc[1][s] = cc[2][s];
          cc[2][s] = cc[3][s];
          cc[3][s] = cnew;
        }
        mh[0][s] = mh[1][s];
        mh[1][s] = mv[s];
        if constexpr (DAMP) {
          dh[0][s] = dh[1][s];
          dh[1][s] = dv[s];
        }
      }
    };
    // advance the B ring of a segment and reload what plane p+1 needs
    auto reload = [&](auto seg) __attribute__((always_inline)) {
      constexpr int s = decltype(seg)::value;
      if constexpr (s == 2 && SHARE) {
        if (skip2) return;
      }
      const int r = cslot[s];
      if constexpr (s < 2) ce[s] = LD2(rC, (sq - 2 * sxb) + rowoff[s], vzcs[s]);   // C array, plane (p+1)-2
      if (STEADY) {
        Bbuf[buf ^ 1][r + 2][lc] = PB(3)[s];           // B(p+1), centre plane of the next iteration
        if constexpr (s == 1 && SHARE) {
          if (skip2) Bbuf[buf ^ 1][dslot + 2][dcol] = PB(3)[s];
        }
        PB(0)[s] = LD4(rB, (sq + 2 * sxb) + rowoff[s], vzs[s]);     // B(p+3) replaces B(p-2)
      } else {
        bc[0][s] = bc[1][s];
        bc[1][s] = bc[2][s];
        bc[2][s] = bc[3][s];
        bc[3][s] = bc[4][s];
        Bbuf[buf ^ 1][r + 2][lc] = bc[2][s];
        if constexpr (s == 1 && SHARE) {
          if (skip2) Bbuf[buf ^ 1][dslot + 2][dcol] = bc[2][s];
        }
        bc[4][s] = LD4(rB, (sq + 2 * sxb) + rowoff[s], vzs[s]);
      }
      if (LPR == 64) {
        if (MASK && ((s == 0 && out0) || (s == 1 && out1))) av[s] = LD4(rC, sq + rowoff[s], vzs[s]);
        else av[s] = s == 0 ? LD4S(rA, sq + rowoff[s], vzs[s]) : LD4(rA, sq + rowoff[s], vzs[s]);
      } else {
        av[s] = s == 0 ? LD4S(rA, sq + rowoff[s], vzs[s]) : LD4(rA, sq + rowoff[s], vzs[s]);
        if (MASK && ((s == 0 && ld_out0) || (s == 1 && ld_out1))) {
          const f32x4 crow = LD4(rC, sq + rowoff[s], vzs[s]);
          av[s] = (s == 0 ? out0 : out1) ? crow : av[s];
        }
      }
      mv[s] = s == 0 ? LD4S(rM, sq + rowoff[s], vzs[s]) : LD4(rM, sq + rowoff[s], vzs[s]);
      if constexpr (DAMP) dv[s] = s == 0 ? LD4S(rDm, sq + rowoff[s], vzs[s]) : LD4(rDm, sq + rowoff[s], vzs[s]);
      be[s] = LD2(rB, sq + rowoff[s], vzes[s]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    // Outer and redundant rows first, and their reloads back to back: the neighbouring tile asks
    // for the same rows (as its redundant / outer rows) at the same point of its iteration, so
    // one of the two requests finds the line in L2.
    // A wave asks for the next plane's operands at raised priority (s_setprio FDTD_FUSE2_PRIO, 0 = never): of the two
    // waves of a SIMD the one that has reached its loads issues them at once instead of taking turns with the other
    // one's arithmetic (512^3 472 against 468.5 Gcells/s, 1024^3 470 against 465, 640^3 and 768^3 + 0.3 %).  Not in the
    // masked variants: 384^3 with its masked 512-cell tiles ran 336 against 341.
    constexpr bool PRIO = FDTD_FUSE2_PRIO != 0 && !MASK;
    compute(S1{});
    compute(S2{});
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(FDTD_FUSE2_PRIO);
    reload(S1{});
    reload(S2{});
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    compute(S0{});
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(FDTD_FUSE2_PRIO);
    reload(S0{});
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    if (!skip2) {
      Bbuf[buf ^ 1][hslot][lc] = bh;
      bh = bh2;
      bh2 = LD4(rB, (sq + 2 * sxb) + rowoff_h, vzh);
      if (ld_seg2) ch = LD4(rC, sq + rowoff[2], vzs[2]);
    }
    __syncthreads();
    ring_w = ring_w == 2 ? 0 : ring_w + 1;
    ring_r = ring_r == 2 ? 0 : ring_r + 1;
    sp += sxb;
#undef PB
#undef PC
  };
  using T = std::true_type;
  using F = std::false_type;
  int p = p0;
  // lead-in: two planes before the chunk, two planes without the second step
#ifdef FDTD_WG_CLOCKS
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg_clock[11][blockIdx.x] = wall_clock64();   // prologue done
#endif
  // The checked form comes in two copies, for planes inside the domain and for its halo planes, whose u(t+1) is loaded
  // instead of computed: with both in one copy every use of the row waits for a load that may just have been issued
  // (seven full waits per plane).  256^3 358 against 348 Gcells/s, 384^3 356 against 347.5, 768^3 446.6 against 443.7.
  // The damped variants keep one copy that decides at run time: two would not fit their registers.
  auto checked = [&](int pl) __attribute__((always_inline)) {
    using P0 = std::integral_constant<int, 0>;
    if constexpr (DAMP || !SPLIT_OK) iteration(P0{}, F{}, std::integral_constant<int, 2>{}, pl);
    else if (pl >= a.dx_m && pl <= a.dx_M) iteration(P0{}, F{}, std::integral_constant<int, 1>{}, pl);
    else iteration(P0{}, F{}, std::integral_constant<int, 0>{}, pl);
  };
  for (; p < x_begin + 2 && p <= p_last; ++p) checked(p);
#ifdef FDTD_WG_CLOCKS
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg_clock[12][blockIdx.x] = wall_clock64();   // lead-in done
#endif
  // steady state in groups of five planes, all of them inside [x_begin + 2, x_end - 1]
  for (; p + 4 <= x_end - 1; p += 5) {
    iteration(std::integral_constant<int, 0>{}, T{}, std::integral_constant<int, 1>{}, p);
    iteration(std::integral_constant<int, 1>{}, T{}, std::integral_constant<int, 1>{}, p + 1);
    iteration(std::integral_constant<int, 2>{}, T{}, std::integral_constant<int, 1>{}, p + 2);
    iteration(std::integral_constant<int, 3>{}, T{}, std::integral_constant<int, 1>{}, p + 3);
    iteration(std::integral_constant<int, 4>{}, T{}, std::integral_constant<int, 1>{}, p + 4);
#ifdef FDTD_WG_CLOCKS
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg_clock[3 + min((p - x_begin) / 16, 7)][blockIdx.x] = wall_clock64();
#endif
  }
#ifdef FDTD_WG_CLOCKS
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg_clock[13][blockIdx.x] = wall_clock64();   // steady planes done
#endif
  for (; p <= p_last; ++p) checked(p);
#undef LD4
#undef LD4S
#undef LD2
}

// Uniform tiling: every z-tile of the same kind; tile index = z-tile fastest (the tiles on either side of a
// z-cut read each other's edge cells and should share an L2), then y-tile, then x-chunk.
template <bool NT, int ZH, bool MASK, int LPR = 64, bool DAMP = false>
__global__ __launch_bounds__(512) void stencil_march2_kernel(March2Args a, int xchunk, int nyt, int nzt, int wg_base) {
  __shared__ f32x4 Bbuf[2][16][128];
  __shared__ f32x4 Cring[3][12][128];
  constexpr int TILE_CELLS = ZH == 2 ? M2_ZCELLS : LPR * 4;
  const int t = wg_base + xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int zt = t % nzt, yt = (t / nzt) % nyt, xc = t / (nyt * nzt);
#ifdef FDTD_WG_CLOCKS
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg_clock[0][blockIdx.x] = wall_clock64();
#endif
  march2_tile<NT, ZH, MASK, LPR, DAMP>(a, xchunk, a.z_m + zt * TILE_CELLS, zt > 0, zt < nzt - 1, yt, xc, Bbuf, Cring);
#ifdef FDTD_WG_CLOCKS
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    g_wg_clock[1][blockIdx.x] = wall_clock64();
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_wg_clock[2][blockIdx.x] = ((unsigned long long)xcc << 32) | (unsigned)t;
  }
#endif
}

// Mixed tiling: a row is covered by z-tiles of different kinds — tiles of 512 cells first, then one of 256,
// then one of 128 (the last one masked if the row ends inside it) — so that rows of 384 = 256 + 128,
// 640 = 512 + 128 or 768 = 512 + 256 cells fill every lane with a single cut or two.  Workgroups are
// dealt in bands of 32 rows: a band is 4 workgroups per 512-cell tile (8 rows each), 2 per 256-cell tile
// (two 8-row tiles each) and 1 per 128-cell tile (four), adjacent in tile index, so that everything that
// touches the same rows shares an XCD's L2.
struct MixedTiling {
  int ntiles;            // z-tiles of a row
  int kind[8];           // 0: 512 cells, 1: 256, 2: 128
  int zoff[8];           // first cell of the z-tile, from z_m
  int slots;             // workgroups per band
  int nbands;
};
template <bool NT, bool MASK, bool DAMP = false>
__global__ __launch_bounds__(512) void stencil_march2_mixed_kernel(March2Args a, int xchunk, MixedTiling mt, int wg_base) {
  __shared__ f32x4 Bbuf[2][16][128];
  __shared__ f32x4 Cring[3][12][128];
  const int t = wg_base + xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int per_chunk = mt.slots * mt.nbands;
  const int xc = t / per_chunk, r = t % per_chunk;
  const int band = r / mt.slots;
  int s = r % mt.slots, zi = 0;
  for (; zi < mt.ntiles - 1; ++zi) {          // which z-tile the slot belongs to
    const int n = 4 >> mt.kind[zi];
    if (s < n) break;
    s -= n;
  }
  const int kind = mt.kind[zi];
  const int zb = a.z_m + mt.zoff[zi];
  const bool cut_l = zi > 0, cut_r = zi < mt.ntiles - 1;
  const int yt = band * (4 >> kind) + s;        // y-tile index in units of this kind's rows per workgroup
  if (MASK && a.y_m + yt * (M2_ROWS << kind) > a.y_M) return;   // a workgroup wholly beyond the last row (last band)
  if (kind == 0) march2_tile<NT, 2, MASK, 64, DAMP>(a, xchunk, zb, cut_l, cut_r, yt, xc, Bbuf, Cring);
  else if (kind == 1) march2_tile<NT, 1, MASK, 64, DAMP, false>(a, xchunk, zb, cut_l, cut_r, yt, xc, Bbuf, Cring);
  else march2_tile<NT, 1, MASK, 32, DAMP>(a, xchunk, zb, cut_l, cut_r, yt, xc, Bbuf, Cring);
}

// D on the four cells around every z-cut of the sweep above (cells cut-2 .. cut+1), from the C
// level in memory once the sweep has completed it: same expression, same operand order.  One
// thread per (row, cut, plane) computes the four cells as two pairs; the four cells start 8 bytes
// off a 16-byte boundary, which a 16-byte global access tolerates (4-byte alignment is enough).
// 256 rows per workgroup, grid (row groups, cuts, planes).  (A version that marched along x with
// the centre row in registers moved half the bytes and was no faster: the accesses are 8 KB apart.)
struct SeamArgs {
  const float *C, *B, *m, *damp;
  float *D;
  long sx;
  int sy, x_m, y_m, y_M, z_m;
  int cut[8];         // first cell behind each cut, counted from z_m
  Coef2 k;
};
struct __attribute__((packed, aligned(4))) Quad { float v[4]; };
template <bool DAMP>
__global__ __launch_bounds__(256) void seam_kernel(SeamArgs a) {
  const int y = a.y_m + (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (y > a.y_M) return;
  const int x = a.x_m + (int)blockIdx.z, z = a.z_m + a.cut[blockIdx.y] - 2;
  const long i = (long)(x + FDTD_HALO) * a.sx + (long)(y + FDTD_HALO) * a.sy + (z + FDTD_HALO);
  auto ld = [](const float *p, long at) {
    const Quad q = *(const Quad *)(p + at);
    return f32x4{q.v[0], q.v[1], q.v[2], q.v[3]};
  };
  const f32x4 c = ld(a.C, i);
  const f32x2 zm = *(const f32x2 *)(a.C + i - 2), zp = *(const f32x2 *)(a.C + i + 4);
  const f32x4 xm2 = ld(a.C, i - 2 * a.sx), xm1 = ld(a.C, i - a.sx), xp1 = ld(a.C, i + a.sx), xp2 = ld(a.C, i + 2 * a.sx);
  const f32x4 ym2 = ld(a.C, i - 2 * a.sy), ym1 = ld(a.C, i - a.sy), yp1 = ld(a.C, i + a.sy), yp2 = ld(a.C, i + 2 * a.sy);
  const f32x4 up = ld(a.B, i), mc = ld(a.m, i);
  const f32x4 dc = DAMP ? ld(a.damp, i) : f32x4{0, 0, 0, 0};
  const f32x2 cl = lo2(c), ch = hi2(c);
  // cells 0,1: z-2: zm  z+2: ch   z-1: (zm.y, c.x)  z+1: (c.y, c.z)
  // cells 2,3: z-2: cl  z+2: zp   z-1: (c.y, c.z)   z+1: (c.w, zp.x)
  const f32x2 o0 = update_pair<DAMP>(a.k, cl, lo2(xm2), lo2(xm1), lo2(xp1), lo2(xp2), lo2(ym2), lo2(ym1), lo2(yp1),
                                     lo2(yp2), zm + ch, f32x2{zm.y + c.y, c.x + c.z}, lo2(up), lo2(mc), lo2(dc));
  const f32x2 o1 = update_pair<DAMP>(a.k, ch, hi2(xm2), hi2(xm1), hi2(xp1), hi2(xp2), hi2(ym2), hi2(ym1), hi2(yp1),
                                     hi2(yp2), cl + zp, f32x2{c.y + c.w, c.z + zp.x}, hi2(up), hi2(mc), hi2(dc));
  Quad o;
  o.v[0] = o0.x; o.v[1] = o0.y; o.v[2] = o1.x; o.v[3] = o1.y;
  *(Quad *)(a.D + i) = o;
}

// The same pass with the u(t+1) cells around the cut staged through LDS: a workgroup takes 32 rows x 8 planes of one
// cut, loads the eight cells cut-4 .. cut+3 of every site of that patch grown by two rows and two planes (two aligned
// 16-byte loads per site) and serves all x- and y-neighbours from there; only u(t), m (and the damping field) of the
// four cells come from memory per thread.  4 instead of 13 scattered line requests per site; same operands in the same
// order, hence the same bits.
constexpr int SEAM_ROWS = 32, SEAM_PLANES = 8;
template <bool DAMP>
__global__ __launch_bounds__(256) void seam_tile_kernel(SeamArgs a, int x_M) {
  __shared__ f32x4 cs[SEAM_PLANES + 4][SEAM_ROWS + 4][2];
  const int ty = threadIdx.x % SEAM_ROWS, tp = threadIdx.x / SEAM_ROWS;
  const int y0 = a.y_m + (int)blockIdx.x * SEAM_ROWS, x0 = a.x_m + (int)blockIdx.z * SEAM_PLANES;
  const int zc = a.z_m + a.cut[blockIdx.y];   // first cell behind the cut
  // rows and planes beyond the box repeat the last ones that exist around it; their values are never used
  for (int i = threadIdx.x; i < (SEAM_PLANES + 4) * (SEAM_ROWS + 4); i += 256) {
    const int pr = i / (SEAM_ROWS + 4), rr = i % (SEAM_ROWS + 4);
    const int x = min(x0 - 2 + pr, x_M + 2), y = min(y0 - 2 + rr, a.y_M + 2);
    const float *at = a.C + (long)(x + FDTD_HALO) * a.sx + (long)(y + FDTD_HALO) * a.sy + (zc - 4 + FDTD_HALO);
    cs[pr][rr][0] = *(const f32x4 *)at;
    cs[pr][rr][1] = *(const f32x4 *)(at + 4);
  }
  __syncthreads();
  const int x = x0 + tp, y = y0 + ty;
  if (x > x_M || y > a.y_M) return;
  auto quad = [&](int pr, int rr) {   // cells cut-2 .. cut+1 of a site
    const f32x4 lo = cs[pr][rr][0], hi = cs[pr][rr][1];
    return f32x4{lo.z, lo.w, hi.x, hi.y};
  };
  const int pr = tp + 2, rr = ty + 2;
  const f32x4 lo = cs[pr][rr][0], hi = cs[pr][rr][1];
  const f32x4 c = f32x4{lo.z, lo.w, hi.x, hi.y};
  const f32x2 zm = f32x2{lo.x, lo.y}, zp = f32x2{hi.z, hi.w};
  const f32x4 xm2 = quad(pr - 2, rr), xm1 = quad(pr - 1, rr), xp1 = quad(pr + 1, rr), xp2 = quad(pr + 2, rr);
  const f32x4 ym2 = quad(pr, rr - 2), ym1 = quad(pr, rr - 1), yp1 = quad(pr, rr + 1), yp2 = quad(pr, rr + 2);
  const long i = (long)(x + FDTD_HALO) * a.sx + (long)(y + FDTD_HALO) * a.sy + (zc - 2 + FDTD_HALO);
  auto ld = [](const float *p, long at) {
    const Quad q = *(const Quad *)(p + at);
    return f32x4{q.v[0], q.v[1], q.v[2], q.v[3]};
  };
  const f32x4 up = ld(a.B, i), mc = ld(a.m, i);
  const f32x4 dc = DAMP ? ld(a.damp, i) : f32x4{0, 0, 0, 0};
  const f32x2 cl = lo2(c), ch = hi2(c);
  const f32x2 o0 = update_pair<DAMP>(a.k, cl, lo2(xm2), lo2(xm1), lo2(xp1), lo2(xp2), lo2(ym2), lo2(ym1), lo2(yp1),
                                     lo2(yp2), zm + ch, f32x2{zm.y + c.y, c.x + c.z}, lo2(up), lo2(mc), lo2(dc));
  const f32x2 o1 = update_pair<DAMP>(a.k, ch, hi2(xm2), hi2(xm1), hi2(xp1), hi2(xp2), hi2(ym2), hi2(ym1), hi2(yp1),
                                     hi2(yp2), cl + zp, f32x2{c.y + c.w, c.z + zp.x}, hi2(up), hi2(mc), hi2(dc));
  Quad o;
  o.v[0] = o0.x; o.v[1] = o0.y; o.v[2] = o1.x; o.v[3] = o1.y;
  *(Quad *)(a.D + i) = o;
}

// More than 8 sources do not fit the sweep's one-lane-per-corner scatter.  The sweep then runs
// without sources, the first scatter is applied to the C level in memory afterwards, and this
// kernel recomputes D on the cells whose stencil contains a scattered cell: the 13 cells of the
// star around every corner that landed inside the box (same expression as the sweep and the seam
// pass; recomputing a cell twice gives the same bits).  The second scatter follows as a launch of
// its own.  One thread per (corner, star cell).
struct RepairArgs {
  const float *C, *B, *m, *coords, *damp;
  float *D;
  long sx;
  int sy, x_m, x_M, y_m, y_M, z_m, z_M;
  int cx_m, cx_M;     // planes whose scattered cells count (the range, grown by 2 inside the domain)
  int x_shift;        // source positions are in the coordinates of the whole domain
  int p_src_m, nsrc, coord_stride;
  float h_x, h_y, h_z, o_x, o_y, o_z;
  Coef2 k;
};
template <bool DAMP>
__global__ __launch_bounds__(256) void repair_kernel(RepairArgs a) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= a.nsrc * 8 * 13) return;
  const int star = tid % 13, corner = (tid / 13) & 7, p = a.p_src_m + tid / (13 * 8);
  const int rx = (corner >> 2) & 1, ry = (corner >> 1) & 1, rz = corner & 1;
  // positions as openacc.cpp:185-187 computes them
  const float gx = (-a.o_x + a.coords[(long)p * a.coord_stride + 0]) / a.h_x;
  const float gy = (-a.o_y + a.coords[(long)p * a.coord_stride + 1]) / a.h_y;
  const float gz = (-a.o_z + a.coords[(long)p * a.coord_stride + 2]) / a.h_z;
  const int tx = rx + (int)floorf(gx) - a.x_shift, ty = ry + (int)floorf(gy), tz = rz + (int)floorf(gz);
  // corners outside the box were in memory before the sweep started: it has seen them
  if (tx < a.cx_m || tx > a.cx_M || ty < a.y_m || ty > a.y_M || tz < a.z_m || tz > a.z_M) return;
  // star cell: 0 = the corner's cell, 1..4 = x-2, x-1, x+1, x+2, then y, then z
  const int axis = star == 0 ? -1 : (star - 1) / 4, d = star == 0 ? 0 : ((star - 1) % 4 < 2 ? (star - 1) % 4 - 2 : (star - 1) % 4 - 1);
  const int x = tx + (axis == 0 ? d : 0), y = ty + (axis == 1 ? d : 0), z = tz + (axis == 2 ? d : 0);
  if (x < a.x_m || x > a.x_M || y < a.y_m || y > a.y_M || z < a.z_m || z > a.z_M) return;
  const long i = (long)(x + FDTD_HALO) * a.sx + (long)(y + FDTD_HALO) * a.sy + (z + FDTD_HALO);
  auto two = [](float v) { return f32x2{v, v}; };
  const float *C = a.C;
  const f32x2 o = update_pair<DAMP>(a.k, two(C[i]), two(C[i - 2 * a.sx]), two(C[i - a.sx]), two(C[i + a.sx]),
                                    two(C[i + 2 * a.sx]), two(C[i - 2 * a.sy]), two(C[i - a.sy]), two(C[i + a.sy]),
                                    two(C[i + 2 * a.sy]), two(C[i - 2] + C[i + 2]), two(C[i - 1] + C[i + 1]), two(a.B[i]),
                                    two(a.m[i]), two(DAMP ? a.damp[i] : 0.0F));
  a.D[i] = o.x;
}

// The shell: the two layers of cells around the updated box, i.e. every halo cell a sweep can
// read (openacc.cpp:161-163) or a scatter can touch (openacc.cpp:192).  One wave per (x, y) row
// of the box grown by 2; rows outside the box in x or y are copied whole, rows inside it only
// contribute their 2 + 2 cells beyond the z range.
// mode 0: copy shell and box;  1: copy the shell only;  2: compare the shells of a and b bit for
// bit and set *flag when they differ.
struct ShellArgs {
  float *dst;
  const float *a, *b;
  int *flag;
  long sx, sy;
  int x_m, x_M, y_m, y_M, z_m, z_M;
  int mode;
};
__global__ __launch_bounds__(64) void shell_kernel(ShellArgs s) {
  const int ny = s.y_M - s.y_m + 5;
  const int x = s.x_m - 2 + (int)(blockIdx.x / ny), y = s.y_m - 2 + (int)(blockIdx.x % ny);
  const bool row_in_shell = x < s.x_m || x > s.x_M || y < s.y_m || y > s.y_M;
  const long row = (long)(x + FDTD_HALO) * s.sx + (long)(y + FDTD_HALO) * s.sy + FDTD_HALO;
  const int ez = s.z_M - s.z_m + 1;
  bool differ = false;
  auto visit = [&](int z) {
    const long c = row + z;
    if (s.mode == 2) differ |= __float_as_uint(s.a[c]) != __float_as_uint(s.b[c]);
    else s.dst[c] = s.a[c];
  };
  if (row_in_shell || s.mode == 0) {
    for (int i = threadIdx.x; i < ez + 4; i += 64) visit(s.z_m - 2 + i);
  } else if (threadIdx.x < 4) {
    visit(threadIdx.x < 2 ? s.z_m - 2 + (int)threadIdx.x : s.z_M - 1 + (int)threadIdx.x);
  }
  if (s.mode == 2 && __any(differ) && threadIdx.x == 0) atomicOr(s.flag, 1);
}

int launch_shell(const struct fdtd_geom *g, float *dst, const float *a, const float *b, int *flag, int mode,
                 hipStream_t st) {
  ShellArgs s;
  s.dst = dst; s.a = a; s.b = b; s.flag = flag;
  s.sx = (long)g->nyp * g->nzp; s.sy = g->nzp;
  s.x_m = g->x_m; s.x_M = g->x_M; s.y_m = g->y_m; s.y_M = g->y_M; s.z_m = g->z_m; s.z_M = g->z_M;
  s.mode = mode;
  const long rows = (long)(g->x_M - g->x_m + 5) * (g->y_M - g->y_m + 5);
  if (rows <= 0 || rows > 0x7fffffffL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(shell_kernel, dim3((unsigned)rows), dim3(64), 0, st, s);
  return (int)hipGetLastError();
}

long g_fused_launches = 0;   // statistics: launches of the two-step sweep in this process

int env_int2(const char *name, int fallback) {
  const char *v = getenv(name);
  return v ? atoi(v) : fallback;
}

}  // namespace
#ifdef FDTD_WG_CLOCKS
extern "C" int fdtd_debug_wg_clocks(unsigned long long *out) {   // 14 x 1024 values of the last launch
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_clock), sizeof(unsigned long long) * 14 * 1024);
}
#endif

// How the (y, z) cross-section of a box is cut into tiles: uniform z-tiles of 512, 256 or 128 cells
// (8, 16 or 32 rows per workgroup), or the mixed tiling; a cross-section that does not fill its tiles
// exactly runs the MASK variants.
struct M2Tiling {
  int zstep;     // cells per z-tile: 512 (one 8-row tile per workgroup), 256 (two), 128 (four); 0: mixed
  bool mask;
  int tile_rows, nyt, nzt;
  double fill;   // cells of the cross-section / cells the tiles span
  MixedTiling mixed;   // zstep == 0
  long wgs;      // workgroups per plane
  double cost;   // modelled ns per plane (see m2_tiling)
};
static M2Tiling m2_tiling(int ey, int ez) {
  // Cost model, from measurements on MI355X: a workgroup with full tiles of 512 or 256 cells takes 4.4 us
  // per plane-iteration, 256 workgroups at a time, i.e. 17.2 ns per workgroup and plane; one whose rows end
  // inside the tile less (4.0 us at three quarters: 0.7 + 0.3 x fill); one with four tiles of 128 cells 5.1
  // to 5.6 us (its row pieces are 512 bytes long); every z-cut adds a seam pass of 0.14 ns per row and plane.
  // (Later kernels: 4.2 / 4.3 / 4.8 us per plane at 512 / 256 / 128 cells and a cheaper seam pass; the choices the model
  // makes were measured again with them and are still the faster ones at 384, 640, 896 and 1152 cells per row.)
  constexpr double WG_NS = 17.2, NARROW = 1.3, SEAM_NS = 0.14;
  auto partial = [](int cells, int tile) { return 0.7 + 0.3 * (double)cells / tile; };
  auto make = [&](int zstep) {
    M2Tiling t;
    t.zstep = zstep;
    t.tile_rows = M2_ROWS * (M2_ZCELLS / zstep);
    t.nyt = (ey + t.tile_rows - 1) / t.tile_rows;
    t.nzt = (ez + t.zstep - 1) / t.zstep;
    t.mask = (ez % t.zstep) != 0 || (ey % t.tile_rows) != 0;
    t.fill = (double)ey * ez / ((double)t.nyt * t.tile_rows * t.nzt * t.zstep);
    t.wgs = (long)t.nyt * t.nzt;
    const int last = ez - (t.nzt - 1) * zstep;   // cells of a row in its last z-tile
    t.cost = t.nyt * WG_NS * (zstep == 128 ? NARROW : 1.0) * ((t.nzt - 1) + partial(last, zstep)) +
             (t.nzt - 1) * SEAM_NS * ey;
    t.mixed.ntiles = 0;
    return t;
  };
  // rows covered by tiles of 512 cells, then one of 256, then one of 128 (masked if the row ends inside it)
  auto make_mixed = [&]() {
    M2Tiling t = make(M2_ZCELLS);
    MixedTiling &mt = t.mixed;
    mt.ntiles = 0; mt.slots = 0;
    int z = 0, narrow_slots = 0;
    double wg_cost = 0.0;
    auto add = [&](int kind) {
      const int width = M2_ZCELLS >> kind, cells = ez - z < width ? ez - z : width;
      mt.kind[mt.ntiles] = kind; mt.zoff[mt.ntiles] = z; ++mt.ntiles;
      mt.slots += 4 >> kind;
      narrow_slots += kind == 2;
      wg_cost += (4 >> kind) * (kind == 2 ? NARROW : 1.0) * partial(cells, width);
      z += width;
    };
    while (ez - z >= M2_ZCELLS && mt.ntiles < 8) add(0);
    if (ez - z >= 256 + 128 + 4 && mt.ntiles < 8) add(0);        // more than 256 + 128 left: one masked tile of 512
    else {
      if (ez - z > 128 && mt.ntiles < 8) add(1);                 // 132 .. 384 left: a tile of 256 (masked up to 252)
      if (ez - z > 0 && mt.ntiles < 8) add(2);                   // up to 128 left
    }
    if (z < ez) { t.mixed.ntiles = 0; return t; }                // more than 8 z-tiles: not a case for this tiling
    mt.nbands = (ey + 31) / 32;
    t.zstep = 0;
    t.nzt = mt.ntiles;
    t.nyt = mt.nbands;
    t.tile_rows = 32;
    t.mask = z != ez || ey % 32 != 0;
    t.fill = (double)ey * ez / ((double)mt.nbands * 32 * z);
    t.wgs = (long)mt.nbands * mt.slots;
    t.cost = mt.nbands * wg_cost * WG_NS + (mt.ntiles - 1) * SEAM_NS * ey;
    // The workgroups with tiles of 128 cells are the slowest of a launch; where they are more than a quarter
    // of it the others wait for them (384 = 256 + 128: 313 against 329 Gcells/s with masked tiles of 512).
    if (4 * narrow_slots > mt.slots) t.cost *= 1.25;
    return t;
  };
  const int forced = env_int2("FDTD_FUSE2_TILE", 0);
  if (forced == 512 || forced == 256 || forced == 128) return make(forced);
  // Tiles of 512 or 256 cells: whichever fills better, the wider one on ties (fewer z-cuts, each of which
  // costs a seam pass over four cells per row and plane).  Tiles of 128 cells throughout are taken on request
  // only (FDTD_FUSE2_TILE=128, or FDTD_FUSE2_NARROW_GAIN=g: when they fill better by g percent): measured on
  // MI355X their sweep is about 10 % slower per cell than the 256-cell one and every additional cut costs
  // 0.14 us per 1000 rows-times-planes, which eats the better fill — 384^3: 294 against 330 Gcells/s with
  // masked 512-cell tiles, 640^3: 332 against 322, 520^3: 273 against 283, 768^3: 338 against 408.
  M2Tiling best = make(M2_ZCELLS);
  {
    const M2Tiling t = make(M2_ZCELLS / 2);
    if (t.fill > best.fill) best = t;
  }
  const int narrow_gain = env_int2("FDTD_FUSE2_NARROW_GAIN", 0);
  if (narrow_gain > 0) {
    const M2Tiling t = make(M2_ZCELLS / 4);
    if (t.fill > best.fill * (1.0 + narrow_gain / 100.0)) best = t;
  }
  // The mixed tiling where the model says it pays by at least 3 % (FDTD_FUSE2_MIXED=0 never, =2 whenever it
  // applies): one cut where the uniform tilings need two or waste a quarter of their lanes.  Measured
  // (Gcells/s, mixed against the better uniform tiling): 520^3 322 / 282, 640^3 372 / 322, 768^3 422 / 407,
  // 896^3 375 / 378, 384^3 313 / 329.
  const int mixed_mode = env_int2("FDTD_FUSE2_MIXED", 1);
  if (mixed_mode != 0 && forced == 0) {
    const M2Tiling t = make_mixed();
    if (t.mixed.ntiles > 1 && t.zstep == 0) {
      bool uniform = true;   // every tile of one kind and none masked in z: the uniform kernels do that
      for (int i = 1; i < t.mixed.ntiles; ++i) uniform = uniform && t.mixed.kind[i] == t.mixed.kind[0];
      if (!(uniform) && (mixed_mode == 2 || t.cost < 0.97 * best.cost)) best = t;
    }
  }
  return best;
}

// Can the fused kernel take this problem?  (Everything else goes through two single sweeps.)
bool fdtd_march2_eligible(const struct fdtd_geom *g, const float *m, const float *b, const float *a_prev,
                          const float *c, const float *d, int nsrc) {
  if (!g) return false;
  const int ex = g->x_M - g->x_m + 1, ey = g->y_M - g->y_m + 1, ez = g->z_M - g->z_m + 1;
  // rows of whole float4s
  if (ex < 1 || ez < 4 || ez % 4 || ey < 1) return false;
  if (g->nzp % 4 || g->z_m % 4 || g->z_m < 0) return false;
  if ((((uintptr_t)m | (uintptr_t)b | (uintptr_t)a_prev | (uintptr_t)c | (uintptr_t)d) % 16) != 0) return false;
  if (((long)g->nyp * g->nzp) % 4) return false;
  // the kernel touches planes x_m-4 .. x_M+4 and rows y_m-4 .. y_M+4 of the padded arrays, and in a
  // row the cells up to z_M+4 (lanes beyond the end of a row work on its four halo cells)
  if (g->x_m < 0 || g->x_M + FDTD_HALO + 4 > g->nxp - 1) return false;
  if (g->y_m < 0 || g->y_M + FDTD_HALO + 4 > g->nyp - 1) return false;
  if (g->z_M + FDTD_HALO + 4 > g->nzp - 1) return false;
  (void)nsrc;   // more than 8 sources: the sweep runs without them and the cells around them are redone
  return true;
}

// Does the automatic choice take the fused kernel for an eligible problem?  At least
// FDTD_FUSE2_MIN_PLANES (32) planes: thinner ranges spend too much on the four planes of u(time+1)
// each chunk recomputes.  At least FDTD_FUSE2_MIN_CELLS (16 M) cells: smaller grids sit in the
// last-level cache and have too few tiles to fill the device, and the marching kernel is the faster
// one there (192^3: 190 against 150 Gcells/s, 224^3: 233 against 223, 48 x 512 x 512: 340 against
// 276, 160 x 160 x 512: 317 against 286; the two cross between 13 M and 17 M cells: 240^3 242 against
// 262, 64 x 512 x 512 351 against 344, 256^3 301 against 325, 192 x 192 x 512 333 against 349).  Tiles filled to
// at least FDTD_FUSE2_MIN_FILL percent (50): the lowest fill measured is 50 % (260^3: 175 against 216
// Gcells/s; the marching kernel pads such rows too), below that nothing is known.
bool fdtd_march2_preferred(const struct fdtd_geom *g) {
  const int ex = g->x_M - g->x_m + 1, ey = g->y_M - g->y_m + 1, ez = g->z_M - g->z_m + 1;
  const M2Tiling t = m2_tiling(ey, ez);
  const long cells = (long)ex * ey * ez;
  return cells >= env_int2("FDTD_FUSE2_MIN_CELLS", 16000000) &&
         100.0 * t.fill >= env_int2("FDTD_FUSE2_MIN_FILL", 50) && ex >= env_int2("FDTD_FUSE2_MIN_PLANES", 32);
}

// Two steps (time, time + 1) on four level buffers: cur = u(time), prev = u(time-1) are read,
// next1 = u(time+1) and next2 = u(time+2) are written (next1 and next2 keep their halos).  Both
// source scatters are included.  Fused into one sweep when the shape allows, otherwise run as
// two sweeps; the result is the same bit for bit.
// ev_begin / ev_end, when not null, are recorded right before the first and right after the last
// sweep kernel.  flags bit 0: the caller guarantees that no source corner lands in the halo layer
// (the two small launches that serve such corners are skipped); bits 2 and 3: u(time+1) is stored
// on the two planes before x_m / behind x_M too, where those are inside the domain.
//
// Slab form (global_nx > 0): the arrays hold planes x_shift-4 .. of a domain of global_nx planes,
// g->x_m..x_M is the local range to update, source positions are global.  u(time+1) is needed two
// planes beyond the range on either side; inside the domain those planes are computed here as
// well (from cur 4 planes and prev 2 planes beyond the range), bit for bit as their owner
// computes them, and with the same source corners added.
// flags bit 16 (internal, set by fdtd_hip_two_steps_slab_noalloc only): `scratch` is the per-device static
// array.  A call that needs more of it than it holds is refused, and calls that use it are ordered one after
// the other whatever their streams (static_scratch_begin / _end below).
constexpr int M2_FLAG_STATIC_SCRATCH = 1 << 16;
static int static_scratch_begin(hipStream_t s);
static int static_scratch_end(hipStream_t s);

int fdtd_two_steps_impl(const struct fdtd_geom *g, const float *m, const float *src,
                        const float *src_coords, const float *cur, const float *prev, float *next1,
                        float *next2, int time, int p_src_m, int p_src_M, int src_stride, int coord_stride,
                        int variant, int flags, int x_shift, int global_nx, void *stream, hipEvent_t ev_begin,
                        hipEvent_t ev_end, float *scratch, long scratch_floats, const float *damp) {
  if (!g || !m || !cur || !prev || !next1 || !next2) return (int)hipErrorInvalidValue;
  {  // the two written levels must not overlap each other or anything that is read
    const size_t vol = (size_t)g->nxp * g->nyp * g->nzp;
    auto overlap = [vol](const float *a, const float *b) { return a < b + vol && b < a + vol; };
    if (overlap(next1, next2) || overlap(next1, cur) || overlap(next1, prev) || overlap(next1, m) ||
        overlap(next2, cur) || overlap(next2, prev) || overlap(next2, m))
      return (int)hipErrorInvalidValue;
    if (damp && (overlap(next1, damp) || overlap(next2, damp))) return (int)hipErrorInvalidValue;
  }
  // one sweep of the single-step kernels, with the damping term when there is a damping field
  auto sweep = [&](const fdtd_geom *gg, const float *u0, const float *u1, float *u2, int v) {
    return damp ? fdtd_stencil_levels_damped(gg, m, damp, u0, u1, u2, FDTD_STENCIL_ORDER, stream)
                : fdtd_stencil_levels(gg, m, u0, u1, u2, v, stream);
  };
  const bool has_src = p_src_M >= p_src_m;
  if (has_src && (!src || !src_coords || coord_stride < 3 || src_stride < 1)) return (int)hipErrorInvalidValue;
  const int nsrc = has_src ? p_src_M - p_src_m + 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  const int ex = g->x_M - g->x_m + 1, ey = g->y_M - g->y_m + 1;
  if (ex <= 0) return 0;
  // the domain's planes in local coordinates, and the planes whose halo-layer corners this call owns
  const int dx_m = global_nx > 0 ? -x_shift : g->x_m, dx_M = global_nx > 0 ? global_nx - 1 - x_shift : g->x_M;
  if (g->x_m < dx_m || g->x_M > dx_M) return (int)hipErrorInvalidValue;
  const int own_lo = (g->x_m == dx_m ? g->x_m - 1 : g->x_m) + x_shift;
  const int own_hi = (g->x_M == dx_M ? g->x_M + 1 : g->x_M) + x_shift;
  const int gx_m = dx_m + x_shift, gx_M = dx_M + x_shift;

  // Launch shape: one workgroup per CU at a time (the LDS of a CU holds one tile).  The x range is
  // cut into chunks of at most ~128 planes (workgroups marching side by side drift apart beyond
  // that and stop finding each other's rows in L2); every tile x chunk is a workgroup, and the
  // device works through ceil(workgroups / 256) rounds of chunk length + lead-in.  The number of
  // chunks is the one with the shortest total under that model.  Every round is a launch of its
  // own: workgroups of one launch start together and stay in step with their y-neighbours, which
  // is what makes one of two requests for a shared row hit in L2 (all rounds in one launch:
  // 1024^3 at 372 instead of 437 Gcells/s).
  const int ez = g->z_M - g->z_m + 1;
  const M2Tiling tl = m2_tiling(ey, ez);
  const int zstep = tl.zstep, nyt = tl.nyt, nzt = tl.nzt;   // zstep 0: mixed tiling (tl.mixed)
  const long ntile = tl.wgs;                                // workgroups per x-chunk
  // a workgroup addresses its xchunk + 9 planes with 32-bit offsets from its first plane
  const long plane_bytes = (long)g->nyp * g->nzp * 4;
  const long addressable = ((1L << 31) - 1) / plane_bytes - 9;
  int xchunk = env_int2("FDTD_FUSE2_XCHUNK", 0);
  if (xchunk <= 0) {
    const int longest = addressable < 128 ? (addressable > 1 ? (int)addressable : 1) : 128;
    const int shortest = ex < 16 ? ex : 16;   // each chunk recomputes 4 planes of C
    long best_cost = -1;
    // (cost in tenths of a plane-iteration.  Of a chunk's planes, the four lead-in planes and the last three go
    // through the checked form of the iteration and the rest in groups of five through the steady one; what a
    // group leaves over runs checked as well, at 5.3-5.7 us a plane instead of 3.9-4.2: a chunk of 78 planes
    // (15 groups) is faster than one of 77 (14 groups and 4 planes left over), so a few planes more per chunk
    // are tried as long as the number of chunks stays.)
    for (int c = (ex + longest - 1) / longest; c <= ex; ++c) {
      const int len0 = (ex + c - 1) / c;
      if (len0 < shortest) break;
      for (int len = len0; len <= len0 + 4 && len <= longest; ++len) {
        if ((ex + len - 1) / len != c) break;
        const long rounds = ((long)ntile * c + 255) / 256;
        const int left_over = len >= 8 ? (len - 3) % 5 : 0;
        const long cost = rounds * (10L * (len + 6) + 4L * left_over);
        if (best_cost < 0 || cost < best_cost) {
          best_cost = cost;
          xchunk = len;
        }
      }
    }
    if (xchunk <= 0) xchunk = ex;
    // flags bits 8..15: make the last chunk that many planes shorter than the others (one round of
    // several chunks only).  Its workgroups then finish early, and a kernel that cannot be placed
    // beside this one (RCCL's send/receive kernel, which needs more than half of a SIMD's
    // registers per wave) gets their CUs while the other chunks are still running.
    const int tail = (flags >> 8) & 0xff;
    const int chunks = (ex + xchunk - 1) / xchunk;
    if (tail > 0 && chunks >= 2 && (long)ntile * chunks <= 256) {
      const int longer = (ex + tail + chunks - 1) / chunks;
      if (longer <= longest && ex - (chunks - 1) * longer >= shortest) xchunk = longer;
    }
  }
  const bool offsets_ok = xchunk <= addressable;
  // More than 8 sources: scatter and repair passes after the sweep work on the C level in memory,
  // which must then hold the two planes beside the range as well: the range ends where the domain
  // does, or the caller asked for those planes to be kept (flags bits 2 and 3).
  const bool many = nsrc > M2_MAX_SRC;
  const bool sides_in_memory = (g->x_m == dx_m || (flags & 4)) && (g->x_M == dx_M || (flags & 8));
  const bool fuse = (variant == FDTD_VARIANT_AUTO || variant == FDTD_VARIANT_MARCH2) &&
                    env_int2("FDTD_FUSE2", 1) != 0 && offsets_ok && !(many && !sides_in_memory) &&
                    fdtd_march2_eligible(g, m, cur, prev, next1, next2, nsrc) && ((uintptr_t)damp % 16) == 0 &&
                    (variant == FDTD_VARIANT_MARCH2 || fdtd_march2_preferred(g));

  // corners of either scatter that land in the halo layer: applied to memory by the call that owns
  // their plane (the sweeps never write there and read next1's halo like any other halo value)
  if (has_src && !(flags & 1)) {
    int rc = fdtd_inject_level(g, m, src, src_coords, next1, time, p_src_m, p_src_M, src_stride, coord_stride,
                               x_shift, own_lo, own_hi, 1, gx_m, gx_M, stream);
    if (rc) return rc;
    rc = fdtd_inject_level(g, m, src, src_coords, next2, time + 1, p_src_m, p_src_M, src_stride, coord_stride,
                           x_shift, own_lo, own_hi, 1, gx_m, gx_M, stream);
    if (rc) return rc;
  }
  // One sweep launch and nothing else: the two events take the start and stop times of that
  // dispatch itself (what a profiler reports as the kernel's duration); otherwise they are
  // recorded on the stream around everything the call launches.
  int per_launch = env_int2("FDTD_FUSE2_WGS", 256);   // workgroups per launch (stride of the launch loop below)
  if (per_launch < 1) per_launch = 256;
  const bool dispatch_times = fuse && !damp && ev_begin && ev_end && nzt == 1 && !many &&
                              (long)ntile * ((ex + xchunk - 1) / xchunk) <= per_launch;
  if (ev_begin && !dispatch_times) FDTD_HIP_CHECK(hipEventRecord(ev_begin, s));

  const long plane = (long)g->nyp * g->nzp;
  const bool static_scratch_call = (flags & M2_FLAG_STATIC_SCRATCH) != 0;
  if (!fuse && static_scratch_call) {
    if (scratch_floats < (long)(ex + 4) * plane) {
      fprintf(stderr,
              "[fdtd-hip] fdtd_hip_two_steps_slab_noalloc: %d planes of %d x %d cells need %ld floats of work space, "
              "the static buffer holds %ld; use fdtd_hip_two_steps_slab with a caller buffer\n",
              ex, g->nyp, g->nzp, (long)(ex + 4) * plane, scratch_floats);
      return (int)hipErrorInvalidValue;
    }
    const int rc = static_scratch_begin(s);
    if (rc) return rc;
  }
  if (!fuse && scratch && scratch_floats >= (long)(ex + 4) * plane) {
    // Thin range next to other ranges that are being updated at the same time (the boundary zone
    // of a slab): u(time+1) on the range grown by two planes is built in `scratch`, so nothing is
    // stored outside the range.  scratch plane j stands for plane x_m - 2 + j.
    const int v1 = variant == FDTD_VARIANT_MARCH2 ? FDTD_VARIANT_AUTO : variant;
    float *c_lvl = scratch - (long)(g->x_m - 2 + FDTD_HALO) * plane;   // addressed like a level
    FDTD_HIP_CHECK(hipMemcpyAsync(scratch, next1 + (long)(g->x_m - 2 + FDTD_HALO) * plane,
                                  (size_t)(ex + 4) * plane * sizeof(float), hipMemcpyDeviceToDevice, s));
    fdtd_geom g1 = *g;
    g1.x_m = g->x_m - 2 > dx_m ? g->x_m - 2 : dx_m;
    g1.x_M = g->x_M + 2 < dx_M ? g->x_M + 2 : dx_M;
    int rc = sweep(&g1, cur, prev, c_lvl, v1);
    if (rc) return rc;
    if (has_src) {
      rc = fdtd_inject_level(g, m, src, src_coords, c_lvl, time, p_src_m, p_src_M, src_stride, coord_stride,
                             x_shift, g1.x_m + x_shift, g1.x_M + x_shift, 2, gx_m, gx_M, stream);
      if (rc) return rc;
    }
    rc = sweep(g, c_lvl, cur, next2, v1);
    if (rc) return rc;
    // back into next1: the range, and on request the extra planes the caller has no other source for
    const int back_lo = (flags & 4) ? g1.x_m : g->x_m, back_hi = (flags & 8) ? g1.x_M : g->x_M;
    FDTD_HIP_CHECK(hipMemcpyAsync(next1 + (long)(back_lo + FDTD_HALO) * plane,
                                  scratch + (long)(back_lo - (g->x_m - 2)) * plane,
                                  (size_t)(back_hi - back_lo + 1) * plane * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (static_scratch_call) {   // the array's last use in this call is the copy above
      const int rc2 = static_scratch_end(s);
      if (rc2) return rc2;
    }
    if (ev_end) FDTD_HIP_CHECK(hipEventRecord(ev_end, s));
    if (has_src)
      rc = fdtd_inject_level(g, m, src, src_coords, next2, time + 1, p_src_m, p_src_M, src_stride, coord_stride,
                             x_shift, g->x_m + x_shift, g->x_M + x_shift, 2, gx_m, gx_M, stream);
    return rc;
  }
  if (!fuse) {
    // two sweeps: the first over the range grown by two planes (inside the domain), with the
    // corners inside the box added on all of those planes; the second over the range itself
    const int v1 = variant == FDTD_VARIANT_MARCH2 ? FDTD_VARIANT_AUTO : variant;
    fdtd_geom g1 = *g;
    g1.x_m = g->x_m - 2 > dx_m ? g->x_m - 2 : dx_m;
    g1.x_M = g->x_M + 2 < dx_M ? g->x_M + 2 : dx_M;
    int rc = sweep(&g1, cur, prev, next1, v1);
    if (rc) return rc;
    if (has_src) {
      rc = fdtd_inject_level(g, m, src, src_coords, next1, time, p_src_m, p_src_M, src_stride, coord_stride,
                             x_shift, g1.x_m + x_shift, g1.x_M + x_shift, 2, gx_m, gx_M, stream);
      if (rc) return rc;
    }
    rc = sweep(g, next1, cur, next2, v1);
    if (rc) return rc;
    if (ev_end) FDTD_HIP_CHECK(hipEventRecord(ev_end, s));
    if (has_src)
      rc = fdtd_inject_level(g, m, src, src_coords, next2, time + 1, p_src_m, p_src_M, src_stride, coord_stride,
                             x_shift, g->x_m + x_shift, g->x_M + x_shift, 2, gx_m, gx_M, stream);
    return rc;
  }

  March2Args a;
  a.B = (const char *)cur; a.A = (const char *)prev; a.m = (const char *)m;
  a.damp = (const char *)damp;
  a.C = (char *)next1; a.D = (char *)next2;
  a.sx = (long)g->nyp * g->nzp; a.sy = g->nzp;
  a.x_m = g->x_m; a.x_M = g->x_M; a.dx_m = dx_m; a.dx_M = dx_M;
  a.y_m = g->y_m; a.y_M = g->y_M; a.z_m = g->z_m; a.z_M = g->z_M;
  a.scatter_d = nzt == 1 && !many;
  a.keep_lo = (flags & 4) != 0;
  a.keep_hi = (flags & 8) != 0;
  a.k.cm2 = -8.33333333e-2F; a.k.cm1 = 1.333333330F; a.k.c0 = -2.50F;
  a.k.dt2 = g->dt * g->dt;
  a.k.r1 = 1.0F / (g->dt * g->dt);
  a.k.r2 = 1.0F / (g->h_x * g->h_x);
  a.k.r3 = 1.0F / (g->h_y * g->h_y);
  a.k.r4 = 1.0F / (g->h_z * g->h_z);
  a.k.neg2r1 = -2.0F * a.k.r1;
  a.k.dt = g->dt;
  a.k.r0 = 1.0F / g->dt;
  a.src_row = has_src ? src + (long)time * src_stride : nullptr;
  a.src_row2 = has_src ? src + (long)(time + 1) * src_stride : nullptr;
  a.coords = src_coords; a.m_f = m;
  a.nsrc = many ? 0 : nsrc; a.p_src_m = p_src_m; a.coord_stride = coord_stride; a.x_shift = x_shift;
  a.h_x = g->h_x; a.h_y = g->h_y; a.h_z = g->h_z; a.o_x = g->o_x; a.o_y = g->o_y; a.o_z = g->o_z;
  const bool nt = env_int2("FDTD_NT", 1) != 0;
  {
    const long chunks = (ex + xchunk - 1) / xchunk;
    if ((long)ntile * chunks > 0x7fffffffL) return (int)hipErrorInvalidValue;
    const long total = (long)ntile * chunks;
    // (per_launch: one round of 256 workgroups whatever the number of tiles of a cross-section.  Whole
    // cross-sections per launch leave a partly filled last round when their tile count is not a
    // multiple of 256: 1536^3, 576 tiles, ran at 342 instead of 480 Gcells/s, 1032^3 at 245 instead of 346.)
    for (long base = 0; base < total; base += per_launch) {
      const dim3 grid((unsigned)(total - base < per_launch ? total - base : per_launch)), block(512);
      const int b = (int)base;
#define M2_LAUNCH(NT_, ZH_, MASK_, LPR_)                                                                    \
  do {                                                                                                       \
    if (damp)                                                                                                \
      hipLaunchKernelGGL((stencil_march2_kernel<true, ZH_, MASK_, LPR_, true>), grid, block, 0, s, a, xchunk, nyt, nzt, b); \
    else if (dispatch_times)                                                                                 \
      hipExtLaunchKernelGGL((stencil_march2_kernel<NT_, ZH_, MASK_, LPR_>), grid, block, 0, s, ev_begin, ev_end, 0, \
                            a, xchunk, nyt, nzt, b);                                                         \
    else                                                                                                     \
      hipLaunchKernelGGL((stencil_march2_kernel<NT_, ZH_, MASK_, LPR_>), grid, block, 0, s, a, xchunk, nyt, nzt, b); \
  } while (0)
      if (zstep == 0) {
        if (damp) {
          if (tl.mask) hipLaunchKernelGGL((stencil_march2_mixed_kernel<true, true, true>), grid, block, 0, s, a, xchunk, tl.mixed, b);
          else hipLaunchKernelGGL((stencil_march2_mixed_kernel<true, false, true>), grid, block, 0, s, a, xchunk, tl.mixed, b);
        } else {
          if (tl.mask) hipLaunchKernelGGL((stencil_march2_mixed_kernel<true, true>), grid, block, 0, s, a, xchunk, tl.mixed, b);
          else hipLaunchKernelGGL((stencil_march2_mixed_kernel<true, false>), grid, block, 0, s, a, xchunk, tl.mixed, b);
        }
      } else if (zstep == M2_ZCELLS) {
        if (tl.mask) { if (nt) M2_LAUNCH(true, 2, true, 64); else M2_LAUNCH(false, 2, true, 64); }
        else { if (nt) M2_LAUNCH(true, 2, false, 64); else M2_LAUNCH(false, 2, false, 64); }
      } else if (zstep == M2_ZCELLS / 2) {
        if (tl.mask) { if (nt) M2_LAUNCH(true, 1, true, 64); else M2_LAUNCH(false, 1, true, 64); }
        else { if (nt) M2_LAUNCH(true, 1, false, 64); else M2_LAUNCH(false, 1, false, 64); }
      } else {
        if (tl.mask) { if (nt) M2_LAUNCH(true, 1, true, 32); else M2_LAUNCH(false, 1, true, 32); }
        else { if (nt) M2_LAUNCH(true, 1, false, 32); else M2_LAUNCH(false, 1, false, 32); }
      }
#undef M2_LAUNCH
      __atomic_add_fetch(&g_fused_launches, 1, __ATOMIC_RELAXED);
    }
  }
  const int grown_lo = g->x_m - 2 > dx_m ? g->x_m - 2 : dx_m, grown_hi = g->x_M + 2 < dx_M ? g->x_M + 2 : dx_M;
  if (many) {   // the first scatter, on the C level in memory (corners inside the box; the others are there already)
    const int rc = fdtd_inject_level(g, m, src, src_coords, next1, time, p_src_m, p_src_M, src_stride, coord_stride,
                                     x_shift, grown_lo + x_shift, grown_hi + x_shift, 2, gx_m, gx_M, stream);
    if (rc) return rc;
  }
  if (nzt > 1) {
    SeamArgs sa;
    sa.C = next1; sa.B = cur; sa.m = m; sa.D = next2; sa.damp = damp;
    sa.sx = a.sx; sa.sy = a.sy; sa.x_m = g->x_m; sa.y_m = g->y_m; sa.y_M = g->y_M; sa.z_m = g->z_m;
    sa.k = a.k;
    // eight cuts per launch; grid.z is limited to 65535 (planes, or groups of 8 planes)
    const bool tiled = env_int2("FDTD_SEAM_TILE", 1) != 0 && (((uintptr_t)next1) % 16 == 0) && a.sx % 4 == 0 && a.sy % 4 == 0 &&
                       g->z_m % 4 == 0;
    for (int c0 = 1; c0 < nzt; c0 += 8) {
      const int ncut = nzt - c0 < 8 ? nzt - c0 : 8;
      for (int i = 0; i < ncut; ++i) sa.cut[i] = zstep ? (c0 + i) * zstep : tl.mixed.zoff[c0 + i];
      const int span = tiled ? 65535 * SEAM_PLANES : 65535;
      for (int x0 = g->x_m; x0 <= g->x_M; x0 += span) {
        sa.x_m = x0;
        const int planes = g->x_M - x0 + 1 < span ? g->x_M - x0 + 1 : span;
        if (tiled) {
          const dim3 sgrid((unsigned)((ey + SEAM_ROWS - 1) / SEAM_ROWS), (unsigned)ncut,
                           (unsigned)((planes + SEAM_PLANES - 1) / SEAM_PLANES));
          if (damp) hipLaunchKernelGGL(seam_tile_kernel<true>, sgrid, dim3(256), 0, s, sa, x0 + planes - 1);
          else hipLaunchKernelGGL(seam_tile_kernel<false>, sgrid, dim3(256), 0, s, sa, x0 + planes - 1);
        } else {
          const dim3 sgrid((unsigned)((ey + 255) / 256), (unsigned)ncut, (unsigned)planes);
          if (damp) hipLaunchKernelGGL(seam_kernel<true>, sgrid, dim3(256), 0, s, sa);
          else hipLaunchKernelGGL(seam_kernel<false>, sgrid, dim3(256), 0, s, sa);
        }
      }
    }
  }
  if (many) {
    RepairArgs ra;
    ra.C = next1; ra.B = cur; ra.m = m; ra.coords = src_coords; ra.D = next2; ra.damp = damp;
    ra.sx = a.sx; ra.sy = a.sy;
    ra.x_m = g->x_m; ra.x_M = g->x_M; ra.y_m = g->y_m; ra.y_M = g->y_M; ra.z_m = g->z_m; ra.z_M = g->z_M;
    ra.cx_m = grown_lo; ra.cx_M = grown_hi; ra.x_shift = x_shift;
    ra.p_src_m = p_src_m; ra.nsrc = nsrc; ra.coord_stride = coord_stride;
    ra.h_x = g->h_x; ra.h_y = g->h_y; ra.h_z = g->h_z; ra.o_x = g->o_x; ra.o_y = g->o_y; ra.o_z = g->o_z;
    ra.k = a.k;
    const long threads = (long)nsrc * 8 * 13;
    if (damp) hipLaunchKernelGGL(repair_kernel<true>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, ra);
    else hipLaunchKernelGGL(repair_kernel<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, ra);
  }
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  if ((nzt > 1 || many) && has_src)
    rc = fdtd_inject_level(g, m, src, src_coords, next2, time + 1, p_src_m, p_src_M, src_stride, coord_stride,
                           x_shift, g->x_m + x_shift, g->x_M + x_shift, 2, gx_m, gx_M, stream);
  if (ev_end && !dispatch_times) FDTD_HIP_CHECK(hipEventRecord(ev_end, s));
  return rc;
}

extern "C" int fdtd_hip_two_steps(const struct fdtd_geom *g, const float *m, const float *src,
                                  const float *src_coords, const float *cur, const float *prev,
                                  float *next1, float *next2, int time, int p_src_m, int p_src_M,
                                  int src_stride, int coord_stride, int variant, void *stream) {
  return fdtd_two_steps_impl(g, m, src, src_coords, cur, prev, next1, next2, time, p_src_m, p_src_M, src_stride,
                             coord_stride, variant, 0, 0, 0, stream, nullptr, nullptr, nullptr, 0, nullptr);
}

// The same two steps with the damping term of the forward operator (fdtd_forward.hip): damp has the padded shape of
// m.  No reference arithmetic exists for it (parity unpinned); with damp = 0 everywhere the results are those of
// fdtd_hip_two_steps.
extern "C" int fdtd_hip_two_steps_damped(const struct fdtd_geom *g, const float *m, const float *damp, const float *src,
                                         const float *src_coords, const float *cur, const float *prev, float *next1,
                                         float *next2, int time, int p_src_m, int p_src_M, int src_stride,
                                         int coord_stride, int variant, int flags, void *stream) {
  if (!damp) return (int)hipErrorInvalidValue;
  return fdtd_two_steps_impl(g, m, src, src_coords, cur, prev, next1, next2, time, p_src_m, p_src_M, src_stride,
                             coord_stride, variant, flags & 0xffff, 0, 0, stream, nullptr, nullptr, nullptr, 0, damp);
}

extern "C" int fdtd_hip_two_steps_ex(const struct fdtd_geom *g, const float *m, const float *src,
                                     const float *src_coords, const float *cur, const float *prev,
                                     float *next1, float *next2, int time, int p_src_m, int p_src_M,
                                     int src_stride, int coord_stride, int variant, int flags,
                                     void *ev_begin, void *ev_end, void *stream) {
  return fdtd_two_steps_impl(g, m, src, src_coords, cur, prev, next1, next2, time, p_src_m, p_src_M, src_stride,
                             coord_stride, variant, flags & 0xffff, 0, 0, stream, (hipEvent_t)ev_begin, (hipEvent_t)ev_end,
                             nullptr, 0, nullptr);
}

// Slab form of fdtd_hip_two_steps (see fdtd_two_steps_impl) and of fdtd_hip_one_step: local arrays
// that hold planes x_shift-4 .. of a domain of global_nx planes, global source positions.
extern "C" int fdtd_hip_two_steps_slab(const struct fdtd_geom *g, const float *m, const float *src,
                                       const float *src_coords, const float *cur, const float *prev,
                                       float *next1, float *next2, int time, int p_src_m, int p_src_M,
                                       int src_stride, int coord_stride, int variant, int flags, int x_shift,
                                       int global_nx, float *scratch, long scratch_floats, void *stream) {
  if (global_nx <= 0) return (int)hipErrorInvalidValue;
  return fdtd_two_steps_impl(g, m, src, src_coords, cur, prev, next1, next2, time, p_src_m, p_src_M, src_stride,
                             coord_stride, variant, flags & 0xffff, x_shift, global_nx, stream, nullptr, nullptr, scratch,
                             scratch_floats, nullptr);
}

// Allocation-free form of fdtd_hip_two_steps_slab for callers that cannot allocate device memory
// (HIP graph capture, serving loops, harnesses that forbid hipMalloc).  The work space of the
// two-sweep path — u(time+1) on the range grown by two planes — lives in a file-scope device
// array instead of a caller buffer.  Every GPU holds its own copy of that array at its own
// address (16 MiB per GPU from the moment the library's code object is loaded there until the
// process ends); the address is looked up once per device, with that device current, and from
// then on only that pointer is used: in host code the array's name is not a device address.
// The array keeps the previous call's contents; the path overwrites all (ex + 4) planes it reads
// before it reads them, so nothing has to be cleared.  All calls on one GPU share the array.  Calls
// that use it are ordered: each records an event behind its last use of the array, and a call on
// another stream makes its stream wait for that event before it touches the array (calls on one
// stream are ordered by the stream).  Ranges that take the fused sweep do not use the array at all
// and are neither limited in size nor ordered.
static __device__ __align__(256) float static_scratch[FDTD_STATIC_SCRATCH_FLOATS];

namespace {
constexpr int kMaxDevices = 64;
struct StaticScratch {
  float *ptr = nullptr;            // device address of static_scratch on this device
  hipEvent_t last_use = nullptr;   // recorded behind the last use of the array
  hipStream_t last_stream = nullptr;
  bool used = false;
};
StaticScratch g_static_scratch[kMaxDevices];
std::mutex g_static_scratch_mutex;

// Set-up for the current device at the first call there: the array's device address and the event that
// orders its users (created here so that later calls create nothing).
int static_scratch_on_current_device(StaticScratch **out) {
  int dev = -1;
  FDTD_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return (int)hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
  StaticScratch &ss = g_static_scratch[dev];
  if (!ss.ptr) {
    float *p = nullptr;
    FDTD_HIP_CHECK(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(static_scratch)));
    if (!p) return (int)hipErrorInvalidSymbol;
    FDTD_HIP_CHECK(hipEventCreateWithFlags(&ss.last_use, hipEventDisableTiming));
    ss.ptr = p;
  }
  *out = &ss;
  return 0;
}
}  // namespace

static int static_scratch_begin(hipStream_t s) {
  StaticScratch *ss = nullptr;
  const int rc = static_scratch_on_current_device(&ss);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
  if (ss->used && ss->last_stream != s) {
    const hipError_t e = hipStreamWaitEvent(s, ss->last_use, 0);
    if (e != hipSuccess) {
      fprintf(stderr, "[fdtd-hip] fdtd_hip_two_steps_slab_noalloc: cannot order this call behind the previous user of "
                      "the static work space on this GPU (%s)\n", hipGetErrorString(e));
      return (int)e;
    }
  }
  return 0;
}
static int static_scratch_end(hipStream_t s) {
  StaticScratch *ss = nullptr;
  const int rc = static_scratch_on_current_device(&ss);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
  FDTD_HIP_CHECK(hipEventRecord(ss->last_use, s));
  ss->last_stream = s;
  ss->used = true;
  return 0;
}

extern "C" int fdtd_hip_two_steps_slab_noalloc(const struct fdtd_geom *g, const float *m, const float *src,
                                               const float *src_coords, const float *cur, const float *prev,
                                               float *next1, float *next2, int time, int p_src_m, int p_src_M,
                                               int src_stride, int coord_stride, int variant, int flags,
                                               int x_shift, int global_nx, void *stream) {
  if (!g || global_nx <= 0) return (int)hipErrorInvalidValue;
  StaticScratch *ss = nullptr;
  const int rc = static_scratch_on_current_device(&ss);
  if (rc) return rc;
  // the size of the range is checked where the path is chosen: only the two-sweep path needs the array
  return fdtd_two_steps_impl(g, m, src, src_coords, cur, prev, next1, next2, time, p_src_m, p_src_M, src_stride,
                             coord_stride, variant, (flags & 0xffff) | M2_FLAG_STATIC_SCRATCH, x_shift, global_nx, stream,
                             nullptr, nullptr, ss->ptr, (long)FDTD_STATIC_SCRATCH_FLOATS, nullptr);
}

extern "C" int fdtd_hip_one_step_slab(const struct fdtd_geom *g, const float *m, const float *src,
                                      const float *src_coords, const float *cur, const float *prev, float *next,
                                      int time, int p_src_m, int p_src_M, int src_stride, int coord_stride,
                                      int variant, int x_shift, int global_nx, void *stream) {
  if (!g || !m || !cur || !prev || !next || global_nx <= 0) return (int)hipErrorInvalidValue;
  const bool has_src = p_src_M >= p_src_m;
  if (has_src && (!src || !src_coords || coord_stride < 3 || src_stride < 1)) return (int)hipErrorInvalidValue;
  if (g->x_M < g->x_m) return 0;
  const int rc = fdtd_stencil_levels(g, m, cur, prev, next, variant, stream);
  if (rc || !has_src) return rc;
  const int own_lo = (g->x_m + x_shift == 0 ? -1 : g->x_m + x_shift);
  const int own_hi = (g->x_M + x_shift == global_nx - 1 ? global_nx : g->x_M + x_shift);
  return fdtd_inject_level(g, m, src, src_coords, next, time, p_src_m, p_src_M, src_stride, coord_stride, x_shift,
                           own_lo, own_hi, 0, 0, 0, stream);
}

// Would fdtd_hip_two_steps run this box (with nsrc sources and 16-byte aligned levels) as the fused
// sweep?  variant as there; lets a caller choose its schedule or its accounting beforehand.
extern "C" int fdtd_hip_two_steps_fuses(const struct fdtd_geom *g, int nsrc, int variant) {
  if (!g || (variant != FDTD_VARIANT_AUTO && variant != FDTD_VARIANT_MARCH2)) return 0;
  if (env_int2("FDTD_FUSE2", 1) == 0) return 0;
  if (!fdtd_march2_eligible(g, nullptr, nullptr, nullptr, nullptr, nullptr, nsrc)) return 0;
  return (variant == FDTD_VARIANT_MARCH2 || fdtd_march2_preferred(g)) ? 1 : 0;
}

// How the two-step sweep tiles a cross-section of ey rows of ez cells: *zstep = 512, 256 or 128 (uniform
// z-tiles) or 0 (mixed: tiles of 512, then 256, then 128 cells), *ncuts = z-cuts per row (each costs a seam
// pass), *fill = cells / cells spanned by the tiles.  Returns the workgroups per plane.
extern "C" long fdtd_hip_two_steps_tiling(int ey, int ez, int *zstep, int *ncuts, double *fill) {
  if (ey < 1 || ez < 4) return 0;
  const M2Tiling t = m2_tiling(ey, ez);
  if (zstep) *zstep = t.zstep;
  if (ncuts) *ncuts = t.nzt - 1;
  if (fill) *fill = t.fill;
  return t.wgs;
}

// Number of two-step sweep launches issued by this process so far (lets a caller or a test see
// whether a run took the fused path or the two-sweep one).
extern "C" long fdtd_hip_fused_launch_count(void) { return __atomic_load_n(&g_fused_launches, __ATOMIC_RELAXED); }

// One step (sweep + scatter) on explicit level buffers: the single-step companion of
// fdtd_hip_two_steps for callers that rotate four buffers.
extern "C" int fdtd_hip_one_step(const struct fdtd_geom *g, const float *m, const float *src,
                                 const float *src_coords, const float *cur, const float *prev, float *next,
                                 int time, int p_src_m, int p_src_M, int src_stride, int coord_stride,
                                 int variant, void *stream) {
  if (!g || !m || !cur || !prev || !next) return (int)hipErrorInvalidValue;
  const bool has_src = p_src_M >= p_src_m;
  if (has_src && (!src || !src_coords || coord_stride < 3 || src_stride < 1)) return (int)hipErrorInvalidValue;
  int rc = fdtd_stencil_levels(g, m, cur, prev, next, variant, stream);
  if (rc || !has_src) return rc;
  return fdtd_inject_level(g, m, src, src_coords, next, time, p_src_m, p_src_M, src_stride, coord_stride, 0,
                           g->x_m - 1, g->x_M + 1, 0, 0, 0, stream);
}

// Copies the shell (mode 1) or the shell and the box (mode 0) of one level into another.  A time
// loop that rotates more than three level buffers uses it to give the buffer that receives
// u(t+2) the shell of u(t-1), which is what writing u(t+2) over u(t-1) in a ring of three keeps.
extern "C" int fdtd_hip_copy_shell(const struct fdtd_geom *g, float *dst_level, const float *src_level,
                                   int with_box, void *stream) {
  if (!g || !dst_level || !src_level) return (int)hipErrorInvalidValue;
  if (g->x_M < g->x_m || g->y_M < g->y_m || g->z_M < g->z_m) return 0;
  return launch_shell(g, dst_level, src_level, nullptr, nullptr, with_box ? 0 : 1, (hipStream_t)stream);
}

// *flag (device int, zeroed by the caller) is set when the shells of the two levels differ.
int fdtd_shells_differ(const struct fdtd_geom *g, const float *a, const float *b, int *flag, void *stream) {
  return launch_shell(g, nullptr, a, b, flag, 2, (hipStream_t)stream);
}
