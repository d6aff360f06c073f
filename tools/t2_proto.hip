// t2_proto — prototype of a two-time-steps-per-sweep stencil kernel (developer experiment).
//
// One launch advances the field by two leapfrog steps: reads B = u(t), A = u(t-1), m; writes
// C = u(t+1) and D = u(t+2).  D goes to a fourth level buffer (it cannot overwrite A in place:
// neighbouring tiles still read A for their redundant rows).  Algorithmic traffic: 5 arrays per
// two updates = 10 B per cell update instead of 16.
//
// Tile: 8 output rows x 512 cells (whole z extent), marching along x.  The workgroup (8 waves)
// computes C on 12 rows (2 redundant rows per side) from B on 16 rows, and D on the 8 rows, two
// planes behind C.  Per wave: z-half h, two output rows and one redundant row.
//   registers: 5-plane x-column of B (3 row segments), 5-plane x-column of C (2 segments)
//   LDS      : B centre planes, double buffered (16 rows), C planes in a ring of 3 (12 rows)
// Restrictions of the prototype: fp32, no sources, z extent exactly 512 (z_m = 0), y extent a
// multiple of 8.
//   usage: t2_proto [nx=512] [ny=512] [reps=20]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>
#include "../include/fdtd_abi.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Coef { float cm2, cm1, c0, dt2, r1, r2, r3, r4, neg2r1; };

__device__ __forceinline__ float update_cell(const Coef &k, float uc, float xm2, float xm1, float xp1, float xp2,
                                             float ym2, float ym1, float yp1, float yp2, float zm2, float zm1,
                                             float zp1, float zp2, float up, float mc) {
  const float r5 = k.c0 * uc;
  const float dxx = r5 + k.cm2 * (xm2 + xp2) + k.cm1 * (xm1 + xp1);
  const float dyy = r5 + k.cm2 * (ym2 + yp2) + k.cm1 * (ym1 + yp1);
  const float dzz = r5 + k.cm2 * (zm2 + zp2) + k.cm1 * (zm1 + zp1);
  return k.dt2 * (k.r2 * dxx + k.r3 * dyy + k.r4 * dzz - (k.neg2r1 * uc + k.r1 * up) * mc) / mc;
}

__device__ __forceinline__ float dpp_prev(float edge, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v),
                                                               0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_next(float edge, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v),
                                                               0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ int xcd_contiguous_tile(int b, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, xcd = b & 7, j = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

struct T2Args {
  const char *B, *A, *m;
  char *C, *D;
  long sx;   // cells per plane
  int sy;    // cells per row
  unsigned bytes;   // size of one level
  int x_m, x_M, y_m, y_M;
  Coef k;
};

// Memory access: buffer instructions.  Address = descriptor base (scalar) + plane offset (scalar)
// + 32-bit lane offset (one vector register per row, loop invariant): no per-lane 64-bit pointers,
// no address arithmetic on the vector unit.  aux 2 = non-temporal.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
template <bool NT>
__device__ __forceinline__ f32x4 ld4(rsrc_t r, int soff, unsigned voff) {
#ifdef EXP_NO_LOADS
  if (soff > 64) return f32x4{(float)voff, 1.f, 2.f, 3.f};
#endif
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, NT ? 2 : 0));
}
__device__ __forceinline__ f32x2 ld2(rsrc_t r, int soff, unsigned voff) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, soff, 0));
}
template <bool NT>
__device__ __forceinline__ void st4(rsrc_t r, int soff, unsigned voff, f32x4 v) {
#ifdef EXP_NO_STORES
  if (v.x != 12345.678f) return;
#endif
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, soff, NT ? 2 : 0);
}

// One row segment: 4 cells per lane, evaluated on register pairs.  A float4 lives in four consecutive
// registers, so its (x,y) and (z,w) halves are aligned pairs and every x/y-direction term maps onto
// packed fp32 instructions without any shuffling; only the z-direction mixes lanes of the pair.
// Same operations, same order and rounding per cell as update_cell.
__device__ __forceinline__ f32x2 lo2(f32x4 v) { return f32x2{v.x, v.y}; }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return f32x2{v.z, v.w}; }
__device__ __forceinline__ f32x2 update_pair(const Coef &k, f32x2 uc, f32x2 xm2, f32x2 xm1, f32x2 xp1, f32x2 xp2,
                                             f32x2 ym2, f32x2 ym1, f32x2 yp1, f32x2 yp2, f32x2 zs2, f32x2 zs1, f32x2 up,
                                             f32x2 mc) {
  // zs2 = (z-2) + (z+2), zs1 = (z-1) + (z+1), already added
  const f32x2 r5 = k.c0 * uc;
  const f32x2 dxx = r5 + k.cm2 * (xm2 + xp2) + k.cm1 * (xm1 + xp1);
  const f32x2 dyy = r5 + k.cm2 * (ym2 + yp2) + k.cm1 * (ym1 + yp1);
  const f32x2 dzz = r5 + k.cm2 * zs2 + k.cm1 * zs1;
  const f32x2 num = k.dt2 * (k.r2 * dxx + k.r3 * dyy + k.r4 * dzz - (k.neg2r1 * uc + k.r1 * up) * mc);
  return f32x2{num.x / mc.x, num.y / mc.y};
}
__device__ __forceinline__ f32x4 update_seg(const Coef &k, f32x4 c, f32x4 xm2, f32x4 xm1, f32x4 xp1, f32x4 xp2,
                                            f32x4 ym2, f32x4 ym1, f32x4 yp1, f32x4 yp2, f32x2 eL, f32x2 eR, f32x4 up,
                                            f32x4 mc) {
  const f32x2 l = f32x2{dpp_prev(eL.x, c.z), dpp_prev(eL.y, c.w)};   // cells z-2, z-1 of cell 0
  const f32x2 r = f32x2{dpp_next(eR.x, c.x), dpp_next(eR.y, c.y)};   // cells z+1, z+2 of cell 3
  const f32x2 cl = lo2(c), chh = hi2(c);
  // cells 0,1: z-2: (l.x, l.y)  z+2: (c.z, c.w)   z-1: (l.y, c.x)  z+1: (c.y, c.z)
  // cells 2,3: z-2: (c.x, c.y)  z+2: (r.x, r.y)   z-1: (c.y, c.z)  z+1: (c.w, r.x)
  const f32x2 zs2_lo = l + chh, zs2_hi = cl + r;
  const f32x2 zs1_lo = f32x2{l.y + c.y, c.x + c.z}, zs1_hi = f32x2{c.y + c.w, c.z + r.x};
  const f32x2 olo = update_pair(k, cl, lo2(xm2), lo2(xm1), lo2(xp1), lo2(xp2), lo2(ym2), lo2(ym1), lo2(yp1), lo2(yp2),
                                zs2_lo, zs1_lo, lo2(up), lo2(mc));
  const f32x2 ohi = update_pair(k, chh, hi2(xm2), hi2(xm1), hi2(xp1), hi2(xp2), hi2(ym2), hi2(ym1), hi2(yp1), hi2(yp2),
                                zs2_hi, zs1_hi, hi2(up), hi2(mc));
  return f32x4{olo.x, olo.y, ohi.x, ohi.y};
}

template <bool NT, bool NTC>
__global__ __launch_bounds__(512) void t2_kernel(T2Args a, int xchunk, int nyt) {
  __shared__ f32x4 Bbuf[2][16][128];
  __shared__ f32x4 Cring[3][12][128];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = wave & 1, q = wave >> 1;
  const int t = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int yt = t % nyt, xc = t / nyt;
  const int yb = a.y_m + yt * 8;
  const int x_begin = a.x_m + xc * xchunk;
  const int x_end = min(x_begin + xchunk - 1, a.x_M);
  const Coef k = a.k;

  // Rows of the three segments of wave (q, h) and their slots in the 12-row C stack (slot = row - yb + 2):
  //   segment 0: an inner output row (2..5 of the tile): nobody else loads it, its A and m stream (NT)
  //   segment 1: an outer output row (0, 1, 6, 7): the neighbouring tile recomputes C there
  //   segment 2: a redundant row (-2, -1, 8, 9): C only, an outer output row of the neighbour
  const int yrow[3] = {yb + 2 + q, q < 2 ? yb + q : yb + 4 + q, q < 2 ? yb - 2 + q : yb + 6 + q};
  const int cslot[3] = {4 + q, q < 2 ? 2 + q : 6 + q, q < 2 ? q : 8 + q};
  const bool seg2_outside = yrow[2] < a.y_m || yrow[2] > a.y_M;
  const int yhalo = q < 2 ? yb - 4 + q : yb + 8 + q;   // B row outside the stack
  const int hslot = q < 2 ? q : 12 + q;
  const int lz = h * 64 + lane;                         // float4 index inside a 512-cell row
  const unsigned zc = (unsigned)(lz * 4 + FDTD_HALO);
  unsigned voff[3], veoff[3], vceoff[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    // byte offsets inside a plane, 32 bit: address = wave-uniform plane pointer + zero-extended lane offset
    voff[s] = ((unsigned)(yrow[s] + FDTD_HALO) * (unsigned)a.sy + zc) * 4u;
    veoff[s] = lane == 63 ? voff[s] + 16u : voff[s] - 8u;
    // C array edges: only the two cells before lane 0 / behind lane 63 are ever used; the lanes in
    // between repeat lane 0's address instead of touching the whole row
    vceoff[s] = lane == 63 ? voff[s] + 16u : voff[s] - (unsigned)lane * 16u - 8u;
  }
  const unsigned vhoff = ((unsigned)(yhalo + FDTD_HALO) * (unsigned)a.sy + zc) * 4u;
  const int sxb = (int)(a.sx * 4);   // bytes per plane

  const int p0 = x_begin - 2, p_last = x_end + 2;
  const rsrc_t rB = make_rsrc(a.B, a.bytes), rA = make_rsrc(a.A, a.bytes), rM = make_rsrc(a.m, a.bytes);
  const rsrc_t rC = make_rsrc(a.C, a.bytes), rD = make_rsrc(a.D, a.bytes);
  int sp = (p0 + FDTD_HALO) * sxb;   // byte offset of plane p

#define LD4(r, so, off) ld4<false>(r, so, off)
#define LD4S(r, so, off) ld4<NT>(r, so, off)
#define LD2(r, so, off) ld2(r, so, off)

  // Register state.  Every value that comes from memory is reloaded for the next plane right
  // after its last use in this plane, into the same registers: one set, a whole iteration of
  // distance between a load and its use, and waits that are exact because loads are consumed in
  // the order they were issued.  The two x-columns are rings of 5 planes; the steady-state loop
  // is unrolled 5 times so that the ring position is a compile-time constant (no register moves).
  f32x4 bc[5][3];     // B planes p-2 .. p+2 (logical plane i lives in bc[(J+i)%5] at phase J)
  f32x4 cc[5][2];     // C planes p-4 .. p-1 and the new plane p (output rows)
  f32x4 mh[2][2];     // m planes p-2, p-1 (output rows)
  f32x4 av[3], mv[3]; // A, m of plane p
  f32x2 be[3];        // z-edge cells of B, plane p
  f32x4 bh, bh2;      // B row outside the 12-row stack, planes p+1 and p+2 (fetched when the
                      // neighbouring tile fetches the same row as one of its own: plane +3)
  f32x2 ce[2];        // z-edge cells of the C array (domain halo), plane p-2
  f32x4 ch;           // C row of the redundant segment when that row is outside the domain, plane p
#pragma unroll
  for (int s = 0; s < 3; ++s) {
#pragma unroll
    for (int i = 0; i < 5; ++i) bc[i][s] = LD4(rB, sp + (i - 2) * sxb, voff[s]);
    av[s] = s == 0 ? LD4S(rA, sp, voff[s]) : LD4(rA, sp, voff[s]);
    mv[s] = s == 0 ? LD4S(rM, sp, voff[s]) : LD4(rM, sp, voff[s]);
    be[s] = LD2(rB, sp, veoff[s]);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 5; ++i) cc[i][s] = f32x4{0, 0, 0, 0};
    mh[0][s] = mh[1][s] = f32x4{1, 1, 1, 1};
    ce[s] = f32x2{0, 0};
  }
  bh = LD4(rB, sp + sxb, vhoff);
  bh2 = LD4(rB, sp + 2 * sxb, vhoff);
  // the redundant segment's C row: from memory when the row is outside the domain; otherwise the
  // register is loaded with something that is being loaded anyway and never used
  ch = seg2_outside ? LD4(rC, sp, voff[2]) : LD4(rB, sp, voff[2]);
  // B(p0): own rows and the outside row
  {
    const int buf = p0 & 1;
#pragma unroll
    for (int s = 0; s < 3; ++s) Bbuf[buf][cslot[s] + 2][lz] = bc[2][s];
    Bbuf[buf][hslot][lz] = LD4(rB, sp, vhoff);
  }
  __syncthreads();
  int ring_w = ((p0 % 3) + 3) % 3;        // ring slot written this iteration (plane p)
  int ring_r = (((p0 - 2) % 3) + 3) % 3;  // ring slot of plane p-2

  // One plane.  STEADY: plane inside the domain and the chunk, D active, no branches, ring phase J.
  // Otherwise (J = 0): every condition checked, columns rotated by register moves.
  auto iteration = [&](auto phase, auto steady, int p) __attribute__((always_inline)) {
    constexpr int J = decltype(phase)::value;
    constexpr bool STEADY = decltype(steady)::value;
#define PB(i) bc[(J + (i)) % 5]
#define PC(i) cc[(J + (i)) % 5]
    const int buf = p & 1;
    const bool more = STEADY ? true : p < p_last;
    const int sq = sp + (more ? sxb : 0);   // plane p+1 (repeats plane p at the very end)
    const bool plane_inside = STEADY ? true : (p >= a.x_m && p <= a.x_M);
    const bool store_c = STEADY ? true : (plane_inside && p >= x_begin && p <= x_end);
    const int d = p - 2;
#ifdef EXP_NO_D
    const bool do_d = false;
#else
    const bool do_d = STEADY ? true : (d >= x_begin && d <= x_end);
#endif
    auto compute = [&](auto seg) __attribute__((always_inline)) {
      constexpr int s = decltype(seg)::value;
      // ---- C(p), row segment s ----
      const int r = cslot[s];   // B slot of the row is r + 2
      f32x4 cnew;
      if (plane_inside) {
        const f32x4 ym2 = Bbuf[buf][r][lz], ym1 = Bbuf[buf][r + 1][lz];
        const f32x4 yp1 = Bbuf[buf][r + 3][lz], yp2 = Bbuf[buf][r + 4][lz];
        cnew = update_seg(k, PB(2)[s], PB(0)[s], PB(1)[s], PB(3)[s], PB(4)[s], ym2, ym1, yp1, yp2, be[s], be[s], av[s],
                          mv[s]);
        if (s == 2) cnew = seg2_outside ? ch : cnew;
      } else {   // halo plane of the domain: the intermediate level keeps what memory holds there
        cnew = LD4(rC, sp, voff[s]);
      }
      if (s < 2 && store_c) st4<NTC>(rC, sp, voff[s], cnew);
      Cring[ring_w][r][lz] = cnew;
      // ---- D(p-2), row segment s ----
      if (s < 2) {
        if (do_d) {
          const f32x4 ym2 = Cring[ring_r][r - 2][lz], ym1 = Cring[ring_r][r - 1][lz];
          const f32x4 yp1 = Cring[ring_r][r + 1][lz], yp2 = Cring[ring_r][r + 2][lz];
          const f32x4 nb = Cring[ring_r][r][h ? 63 : 64];   // the other z-half of the same row
          const f32x2 eL = h ? f32x2{nb.z, nb.w} : ce[s];
          const f32x2 eR = h ? ce[s] : f32x2{nb.x, nb.y};
          const f32x4 o = update_seg(k, PC(2)[s], PC(0)[s], PC(1)[s], PC(3)[s], cnew, ym2, ym1, yp1, yp2, eL, eR,
                                     PB(0)[s], mh[0][s]);
          st4<NT>(rD, sp - 2 * sxb, voff[s], o);
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
      }
    };
    auto reload = [&](auto seg) __attribute__((always_inline)) {
      constexpr int s = decltype(seg)::value;
      const int r = cslot[s];
      if (s < 2) ce[s] = LD2(rC, sq - 2 * sxb, vceoff[s]);   // C array, plane (p+1)-2
      // ---- advance the column of this segment and reload what plane p+1 needs ----
      if (STEADY) {
        Bbuf[buf ^ 1][r + 2][lz] = PB(3)[s];           // B(p+1), centre plane of the next iteration
#ifdef EXP_B_NT
        PB(0)[s] = s == 0 ? LD4S(rB, sq + 2 * sxb, voff[s]) : LD4(rB, sq + 2 * sxb, voff[s]);   // B(p+3) replaces B(p-2)
#else
        PB(0)[s] = LD4(rB, sq + 2 * sxb, voff[s]);         // B(p+3) replaces B(p-2)
#endif
      } else {
        bc[0][s] = bc[1][s];
        bc[1][s] = bc[2][s];
        bc[2][s] = bc[3][s];
        bc[3][s] = bc[4][s];
        Bbuf[buf ^ 1][r + 2][lz] = bc[2][s];
        bc[4][s] = LD4(rB, sq + 2 * sxb, voff[s]);
      }
      av[s] = s == 0 ? LD4S(rA, sq, voff[s]) : LD4(rA, sq, voff[s]);
      mv[s] = s == 0 ? LD4S(rM, sq, voff[s]) : LD4(rM, sq, voff[s]);
      be[s] = LD2(rB, sq, veoff[s]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    // Outer and redundant rows first, and their reloads back to back: the neighbouring tile asks for
    // the same rows (its redundant / outer rows) at the same point of its iteration.
    compute(S1{});
    compute(S2{});
    reload(S1{});
    reload(S2{});
    compute(S0{});
    reload(S0{});
    Bbuf[buf ^ 1][hslot][lz] = bh;
    bh = bh2;
    bh2 = LD4(rB, sq + 2 * sxb, vhoff);
#ifdef EXP_CH_DUP
    ch = seg2_outside ? LD4(rC, sq, voff[2]) : LD4(rB, sq + 2 * sxb, voff[2]);
#else
    if (seg2_outside) ch = LD4(rC, sq, voff[2]);
#endif
#ifndef EXP_NO_BARRIER
    __syncthreads();
#endif
    ring_w = ring_w == 2 ? 0 : ring_w + 1;
    ring_r = ring_r == 2 ? 0 : ring_r + 1;
    sp += sxb;
#undef PB
#undef PC
  };
  using T = std::true_type;
  using F = std::false_type;
  int p = p0;
  // lead-in: two planes before the chunk, two planes without D
  for (; p < x_begin + 2 && p <= p_last; ++p) iteration(std::integral_constant<int, 0>{}, F{}, p);
  // steady state in groups of five planes: all of p .. p+4 inside [x_begin + 2, x_end - 1]
  // (x_end itself goes through the checked path: it is the last plane that may load plane p+1 freely)
  for (; p + 4 <= x_end - 1; p += 5) {
    iteration(std::integral_constant<int, 0>{}, T{}, p);
    iteration(std::integral_constant<int, 1>{}, T{}, p + 1);
    iteration(std::integral_constant<int, 2>{}, T{}, p + 2);
    iteration(std::integral_constant<int, 3>{}, T{}, p + 3);
    iteration(std::integral_constant<int, 4>{}, T{}, p + 4);
  }
  for (; p <= p_last; ++p) iteration(std::integral_constant<int, 0>{}, F{}, p);
}

// 3 reads + 2 writes of float4, the traffic mix of the two-step kernel, as a plain stream
template <int U>
__global__ __launch_bounds__(256) void stream_3r2w(const f32x4 *a, const f32x4 *b, const f32x4 *c, f32x4 *d, f32x4 *e,
                                                   size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
    f32x4 x[U], y[U], z[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      x[k] = a[i + k * stride];
      y[k] = __builtin_nontemporal_load(&b[i + k * stride]);
      z[k] = __builtin_nontemporal_load(&c[i + k * stride]);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      __builtin_nontemporal_store(x[k] + y[k] * z[k], &d[i + k * stride]);
      __builtin_nontemporal_store(x[k] - y[k] * z[k], &e[i + k * stride]);
    }
  }
}

// ------------------------------------------------------------------------------------------
static void fill(std::vector<float> &v, unsigned seed, float lo, float hi) {
  unsigned s = seed;
  for (auto &x : v) {
    s = s * 1664525u + 1013904223u;
    x = lo + (hi - lo) * (float)(s >> 8) / 16777216.0f;
  }
}

int main(int argc, char **argv) {
  const int nx = argc > 1 ? atoi(argv[1]) : 512, ny = argc > 2 ? atoi(argv[2]) : 512, nz = 512;
  const int reps = argc > 3 ? atoi(argv[3]) : 20;
  const int xchunk_arg = argc > 4 ? atoi(argv[4]) : 0;
  if (ny % 8) { printf("ny must be a multiple of 8\n"); return 1; }
  const int nxp = nx + 8, nyp = ny + 8, nzp = nz + 8;
  const size_t vol = (size_t)nxp * nyp * nzp;
  std::vector<float> hu(3 * vol), hm(vol);
  fill(hu, 1u, -1.0f, 1.0f);
  fill(hm, 2u, 1.0f, 2.0f);
  float *du, *dm, *dD, *dref;
  CK(hipMalloc(&du, 3 * vol * 4));
  CK(hipMalloc(&dref, 3 * vol * 4));
  CK(hipMalloc(&dm, vol * 4));
  CK(hipMalloc(&dD, vol * 4));
  CK(hipMemcpy(du, hu.data(), 3 * vol * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dref, hu.data(), 3 * vol * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dm, hm.data(), vol * 4, hipMemcpyHostToDevice));



  const float dt = 1e-3f, hx = 0.1f, hy = 0.11f, hz = 0.09f;
  struct fdtd_geom g = {nxp, nyp, nzp, 0, nx - 1, 0, ny - 1, 0, nz - 1, dt, hx, hy, hz, 0, 0, 0};
  // reference: two single steps with the product's kernels; time = 0: t0 = 0, t1 = 2, t2 = 1; then t0 = 1, t1 = 0, t2 = 2
  if (fdtd_hip_stencil_step(&g, dm, dref, 0, 2, 1, FDTD_VARIANT_AUTO, 0)) return 2;
  if (fdtd_hip_stencil_step(&g, dm, dref, 1, 0, 2, FDTD_VARIANT_AUTO, 0)) return 2;
  CK(hipDeviceSynchronize());

  T2Args a;
  a.B = (const char *)(du);             // level 0 = u(t)
  a.A = (const char *)(du + 2 * vol);   // level 2 = u(t-1)
  a.C = (char *)(du + vol);             // level 1 = u(t+1)
  a.D = (char *)dD;                     // u(t+2); reference writes it into level 2
  // D's halo must equal level 2's halo for the comparison
  CK(hipMemcpy(dD, du + 2 * vol, vol * 4, hipMemcpyDeviceToDevice));
  a.m = (const char *)dm;
  a.bytes = (unsigned)(vol * 4);
  a.sx = (long)nyp * nzp;
  a.sy = nzp;
  a.x_m = 0; a.x_M = nx - 1; a.y_m = 0; a.y_M = ny - 1;
  a.k.cm2 = -8.33333333e-2F; a.k.cm1 = 1.333333330F; a.k.c0 = -2.50F;
  a.k.dt2 = dt * dt;
  a.k.r1 = 1.0F / (dt * dt); a.k.r2 = 1.0F / (hx * hx); a.k.r3 = 1.0F / (hy * hy); a.k.r4 = 1.0F / (hz * hz);
  a.k.neg2r1 = -2.0F * a.k.r1;
  const int nyt = ny / 8;
  int nxc = 256 / nyt;
  if (nxc < 1) nxc = 1;
  int xchunk = xchunk_arg ? xchunk_arg : (nx + nxc - 1) / nxc;
  nxc = (nx + xchunk - 1) / xchunk;
  printf("grid %dx%dx%d: %d y-tiles x %d x-chunks of %d planes = %d workgroups\n", nx, ny, nz, nyt, nxc, xchunk, nyt * nxc);
  hipLaunchKernelGGL((t2_kernel<true, true>), dim3(nyt * nxc), dim3(512), 0, 0, a, xchunk, nyt);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());

  // compare C (level 1) and D against the reference's levels 1 and 2
  std::vector<float> c1(vol), c2(vol), r1(vol), r2(vol);
  CK(hipMemcpy(c1.data(), du + vol, vol * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c2.data(), dD, vol * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r1.data(), dref + vol, vol * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r2.data(), dref + 2 * vol, vol * 4, hipMemcpyDeviceToHost));
  size_t bad1 = 0, bad2 = 0, first = (size_t)-1;
  for (size_t i = 0; i < vol; ++i) {
    if (memcmp(&c1[i], &r1[i], 4)) { ++bad1; if (first == (size_t)-1) first = i; }
    if (memcmp(&c2[i], &r2[i], 4)) { ++bad2; if (first == (size_t)-1) first = i; }
  }
  printf("bitwise: C differs in %zu words, D differs in %zu words of %zu\n", bad1, bad2, vol);
  if (first != (size_t)-1) {
    size_t i = first;
    printf("  first at x=%zu y=%zu z=%zu (padded): C %g vs %g, D %g vs %g\n", i / ((size_t)nyp * nzp), (i / nzp) % nyp,
           i % nzp, c1[i], r1[i], c2[i], r2[i]);
  }

  // timing
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((t2_kernel<true, true>), dim3(nyt * nxc), dim3(512), 0, 0, a, xchunk, nyt);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((t2_kernel<true, true>), dim3(nyt * nxc), dim3(512), 0, 0, a, xchunk, nyt);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double cells = (double)nx * ny * nz;
  printf("two-step kernel: %.4f ms per launch = %.4f ms per time step, %.1f Gcell-updates/s, %.0f GB/s at 16 B/update\n", ms,
         ms / 2, 2 * cells / ms / 1e6, 2 * cells * 16 / ms / 1e6);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) {
    fdtd_hip_stencil_step(&g, dm, dref, 0, 2, 1, FDTD_VARIANT_AUTO, 0);
    fdtd_hip_stencil_step(&g, dm, dref, 1, 0, 2, FDTD_VARIANT_AUTO, 0);
  }
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("two single steps: %.4f ms = %.4f ms per time step, %.1f Gcell-updates/s\n", ms, ms / 2, 2 * cells / ms / 1e6);
  for (int grid : {2048, 8192, 16384}) {
    const size_t n4 = vol / 4;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i)
      hipLaunchKernelGGL((stream_3r2w<4>), dim3(grid), dim3(256), 0, 0, (const f32x4 *)dref, (const f32x4 *)(dref + vol),
                         (const f32x4 *)dm, (f32x4 *)(dref + 2 * vol), (f32x4 *)dD, n4);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("3R2W stream grid %d: %.4f ms, %.0f GB/s\n", grid, ms, 5.0 * vol * 4 / ms / 1e6);
  }
  return (bad1 || bad2) ? 3 : 0;
}
