// two_step_order_budget.hip — register and LDS budget of a two-steps-per-sweep tile at radius R (space order 2R).
//
// Not a product kernel and never launched: a skeleton with the state a two-step tile of radius R has to keep —
// the x-direction of B = u(t) and of C = u(t+1) in two register rings of 2R+1 planes (advanced by renaming:
// the plane loop is unrolled 2R+1 times), the centre planes in LDS for the y-neighbours (B double-buffered on
// ROWS + 4R rows, C in a ring of R+1 planes on ROWS + 2R rows), the z-neighbours by lane shifts, A = u(t-1),
// m of plane p and R planes of m history for the second step — and the arithmetic of the star update on it
// (openacc.cpp:159-164 with R pairs of neighbours per axis).  No masks, no sources, no z-cuts, no edge
// cells: everything the real kernel (csrc/fdtd_march2.hip, R = 2) adds on top costs registers too.
//
//   make -C tools budget      compiles the instantiations below with -Rpass-analysis=kernel-resource-usage and
//                             prints registers, spills and LDS of each (hipcc refuses the ones whose LDS
//                             exceeds the 160 KB of a CU; those are compiled with -DBUDGET_LDS_ONLY, which keeps
//                             the rings and drops the LDS arrays, to show the registers alone)
#include <hip/hip_runtime.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R> struct W { static constexpr float w[7] = {-2.72222222F, 1.5F, -1.5e-1F, 1.11111111e-2F, 0.F, 0.F, 0.F}; };
template <> struct W<2> { static constexpr float w[7] = {-2.50F, 1.333333330F, -8.33333333e-2F, 0.F, 0.F, 0.F, 0.F}; };
template <> struct W<4> { static constexpr float w[7] = {-2.84722222F, 1.6F, -2.0e-1F, 2.53968254e-2F, -1.78571429e-3F, 0.F, 0.F}; };

__device__ __forceinline__ float lane_prev(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

struct K { float dt2, r1, r2, r3, r4, neg2r1; };

// one row segment (4 cells per lane) of the star update; xs[0 .. 2R] = planes p-R .. p+R, ys[0 .. 2R-1] = rows
// y-R .. y-1, y+1 .. y+R of the centre plane
template <int R>
__device__ __forceinline__ f32x4 star(const K &k, const f32x4 (&xs)[2 * R + 1], const f32x4 (&ys)[2 * R], f32x4 up, f32x4 mc) {
  const f32x4 c = xs[R];
  // the cells before and behind the lane's four: the neighbouring lanes' float4s (R <= 4)
  const f32x4 lo = f32x4{lane_prev(c.x), lane_prev(c.y), lane_prev(c.z), lane_prev(c.w)};
  const f32x4 hi = f32x4{lane_next(c.x), lane_next(c.y), lane_next(c.z), lane_next(c.w)};
  const float win[12] = {lo.x, lo.y, lo.z, lo.w, c.x, c.y, c.z, c.w, hi.x, hi.y, hi.z, hi.w};
  const f32x4 r5 = W<R>::w[0] * c;
  f32x4 dxx = r5, dyy = r5, dzz = r5;
#pragma unroll
  for (int j = R; j >= 1; --j) {
    const float w = W<R>::w[j];
    dxx = dxx + w * (xs[R - j] + xs[R + j]);
    dyy = dyy + w * (ys[R - j] + ys[R + j - 1]);
    const f32x4 zm = f32x4{win[4 - j], win[5 - j], win[6 - j], win[7 - j]};
    const f32x4 zp = f32x4{win[4 + j], win[5 + j], win[6 + j], win[7 + j]};
    dzz = dzz + w * (zm + zp);
  }
  const f32x4 num = k.dt2 * (k.r2 * dxx + k.r3 * dyy + k.r4 * dzz - (k.neg2r1 * c + k.r1 * up) * mc);
  return f32x4{num.x / mc.x, num.y / mc.y, num.z / mc.z, num.w / mc.w};
}

// ROWS output rows x 512 cells per tile, 8 waves: (q, h), q = 0 .. 3 stacked in y, h = z-half.
template <int R, int ROWS>
__global__ __launch_bounds__(512) void two_step_tile_skeleton(const f32x4 *__restrict__ B, const f32x4 *__restrict__ A,
                                                              const f32x4 *__restrict__ M, f32x4 *__restrict__ C,
                                                              f32x4 *__restrict__ D, int sx, int sy, int planes, K k) {
  constexpr int NP = 2 * R + 1;                 // planes of a ring
  constexpr int CROWS = ROWS + 2 * R;           // rows of C a tile computes (R redundant rows per side)
  constexpr int NSB = (CROWS + 3) / 4;          // row segments of B (and of the first update) per wave
  constexpr int NSC = (ROWS + 3) / 4;           // row segments of C kept for the second update per wave
#ifndef BUDGET_LDS_ONLY
  __shared__ f32x4 Bbuf[2][ROWS + 4 * R][128];  // centre plane of B, double-buffered
  __shared__ f32x4 Cring[R + 1][CROWS][128];    // C planes p-R .. p
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = wave & 1, q = wave >> 1, lc = h * 64 + lane;
  f32x4 bc[NP][NSB], cc[NP][NSC], mh[R][NSC], av[NSB], mv[NSB];
  int row[NSB];
#pragma unroll
  for (int s = 0; s < NSB; ++s) row[s] = (blockIdx.x * ROWS + q + 4 * s) * sy + lc;
#pragma unroll
  for (int s = 0; s < NSB; ++s) {
#pragma unroll
    for (int i = 0; i < NP; ++i) bc[i][s] = B[(long)i * sx + row[s]];
    av[s] = A[(long)R * sx + row[s]];
    mv[s] = M[(long)R * sx + row[s]];
  }
#pragma unroll
  for (int s = 0; s < NSC; ++s) {
#pragma unroll
    for (int i = 0; i < NP; ++i) cc[i][s] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < R; ++i) mh[i][s] = f32x4{1, 1, 1, 1};
  }
  long sp = (long)R * sx;
  auto iteration = [&](auto phase, int p) __attribute__((always_inline)) {
    constexpr int J = decltype(phase)::value;
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
      // ---- first update: C(p) on this segment ----
      f32x4 xs[NP], ys[2 * R];
#pragma unroll
      for (int i = 0; i < NP; ++i) xs[i] = bc[(J + i) % NP][s];
#ifndef BUDGET_LDS_ONLY
#pragma unroll
      for (int j = 0; j < R; ++j) {
        ys[j] = Bbuf[p & 1][q + 4 * s + j][lc];
        ys[R + j] = Bbuf[p & 1][q + 4 * s + R + 1 + j][lc];
      }
#else
#pragma unroll
      for (int j = 0; j < 2 * R; ++j) ys[j] = B[sp + row[s] + (j - R) * sy];
#endif
      const f32x4 cnew = star<R>(k, xs, ys, av[s], mv[s]);
      if (s < NSC) C[sp + row[s]] = cnew;
#ifndef BUDGET_LDS_ONLY
      Cring[p % (R + 1)][q + 4 * s][lc] = cnew;
#endif
      // ---- second update: D(p - R) on the output segments ----
      if (s < NSC) {
        f32x4 xc[NP], yc[2 * R];
#pragma unroll
        for (int i = 0; i + 1 < NP; ++i) xc[i] = cc[(J + i) % NP][s];
        xc[NP - 1] = cnew;
#ifndef BUDGET_LDS_ONLY
#pragma unroll
        for (int j = 0; j < R; ++j) {
          yc[j] = Cring[(p + 1) % (R + 1)][q + 4 * s + j][lc];
          yc[R + j] = Cring[(p + 1) % (R + 1)][q + 4 * s + R + 1 + j][lc];
        }
#else
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) yc[j] = C[sp - (long)R * sx + row[s] + (j - R) * sy];
#endif
        D[sp - (long)R * sx + row[s]] = star<R>(k, xc, yc, bc[J % NP][s], mh[0][s]);
        cc[(J + NP - 1) % NP][s] = cnew;   // replaces the oldest plane: renaming, no moves
#pragma unroll
        for (int i = 0; i + 1 < R; ++i) mh[i][s] = mh[i + 1][s];
        mh[R - 1][s] = mv[s];
      }
      // ---- reload for plane p + 1 into the same registers ----
#ifndef BUDGET_LDS_ONLY
      Bbuf[(p & 1) ^ 1][q + 4 * s + R][lc] = bc[(J + R + 1) % NP][s];
#endif
      bc[J % NP][s] = B[sp + (long)(R + 1) * sx + row[s]];
      av[s] = A[sp + sx + row[s]];
      mv[s] = M[sp + sx + row[s]];
    }
    __syncthreads();
    sp += sx;
  };
  int p = 0;
  for (; p + NP <= planes; p += NP) {
    iteration(std::integral_constant<int, 0>{}, p);
    iteration(std::integral_constant<int, 1>{}, p + 1);
    iteration(std::integral_constant<int, 2>{}, p + 2);
    iteration(std::integral_constant<int, 3>{}, p + 3);
    iteration(std::integral_constant<int, 4>{}, p + 4);
    if constexpr (NP > 5) {
      iteration(std::integral_constant<int, 5>{}, p + 5);
      iteration(std::integral_constant<int, 6>{}, p + 6);
    }
    if constexpr (NP > 7) {
      iteration(std::integral_constant<int, 7>{}, p + 7);
      iteration(std::integral_constant<int, 8>{}, p + 8);
    }
  }
}

#ifndef BUDGET_R
#define BUDGET_R 2
#endif
#ifndef BUDGET_ROWS
#define BUDGET_ROWS 8
#endif
template __global__ void two_step_tile_skeleton<BUDGET_R, BUDGET_ROWS>(const f32x4 *, const f32x4 *, const f32x4 *, f32x4 *,
                                                                       f32x4 *, int, int, int, K);
