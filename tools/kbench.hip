// kbench — kernel-level timing harness for the stencil variants (developer tool).
// Allocates an n^3 problem on the GPU, fills it with a smooth random field, and times
// fdtd_hip_stencil_step for each requested variant with HIP events.  Also times a pure
// streaming kernel with the same traffic shape (3 reads + 1 write of float4) to show what
// the memory system delivers for this access mix.
//   usage: kbench [n=512] [reps=20]     variants/tunables via FDTD_* environment variables
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/fdtd_abi.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void stream_3r1w(const float4 *a, const float4 *b, const float4 *c, float4 *d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 x = a[i], y = b[i], z = c[i];
    d[i] = make_float4(x.x + y.x * z.x, x.y + y.y * z.y, x.z + y.z * z.z, x.w + y.w * z.w);
  }
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// 3 reads + 1 write, U float4 per thread per array in flight, optional non-temporal hints
template <int U, bool NT>
__global__ __launch_bounds__(256) void stream_u(const f32x4 *a, const f32x4 *b, const f32x4 *c, f32x4 *d, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
    f32x4 x[U], y[U], z[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      x[k] = a[i + k * stride];
      y[k] = NT ? __builtin_nontemporal_load(&b[i + k * stride]) : b[i + k * stride];
      z[k] = NT ? __builtin_nontemporal_load(&c[i + k * stride]) : c[i + k * stride];
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      f32x4 r = x[k] + y[k] * z[k];
      if (NT) __builtin_nontemporal_store(r, &d[i + k * stride]); else d[i + k * stride] = r;
    }
  }
}
template <int U>
__global__ __launch_bounds__(256) void copy_u(const f32x4 *a, f32x4 *d, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
    f32x4 x[U];
#pragma unroll
    for (int k = 0; k < U; ++k) x[k] = a[i + k * stride];
#pragma unroll
    for (int k = 0; k < U; ++k) d[i + k * stride] = x[k];
  }
}
template <int U>
__global__ __launch_bounds__(256) void read_u(const f32x4 *a, f32x4 *d, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  f32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
    for (int k = 0; k < U; ++k) acc += a[i + k * stride];
  }
  if (acc.x == 123.456f) d[0] = acc;
}
// block-contiguous 3R1W: each workgroup streams its own contiguous chunk
template <int U, int BS>
__global__ __launch_bounds__(BS) void stream_chunk(const f32x4 *a, const f32x4 *b, const f32x4 *c, f32x4 *d, size_t n) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = (lo + per < n) ? lo + per : n;
  for (size_t i = lo + threadIdx.x; i + (U - 1) * BS < hi; i += U * BS) {
    f32x4 x[U], y[U], z[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { x[k] = a[i + k * BS]; y[k] = b[i + k * BS]; z[k] = c[i + k * BS]; }
#pragma unroll
    for (int k = 0; k < U; ++k) d[i + k * BS] = x[k] + y[k] * z[k];
  }
}
// one-shot: every thread handles exactly U float4 per array, no loop (grid covers the array)
template <int U>
__global__ __launch_bounds__(256) void stream_oneshot(const f32x4 *a, const f32x4 *b, const f32x4 *c, f32x4 *d, size_t n) {
  const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  f32x4 x[U], y[U], z[U];
#pragma unroll
  for (int k = 0; k < U; ++k) { size_t i = base + k * 256; if (i < n) { x[k] = a[i]; y[k] = b[i]; z[k] = c[i]; } }
#pragma unroll
  for (int k = 0; k < U; ++k) { size_t i = base + k * 256; if (i < n) d[i] = x[k] + y[k] * z[k]; }
}
template <int U>
__global__ __launch_bounds__(256) void read3_u(const f32x4 *a, const f32x4 *b, const f32x4 *c, f32x4 *d, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  f32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
    for (int k = 0; k < U; ++k) acc += a[i + k * stride] + b[i + k * stride] * c[i + k * stride];
  }
  if (acc.x == 123.456f) d[0] = acc;
}
__global__ void count_diff(const unsigned *a, const unsigned *b, size_t n, unsigned long long *out) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    c += (a[i] != b[i]);
  if (c) atomicAdd(out, c);
}
__global__ void fill(float *p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xffff) / 65536.0f + 1.0f;
  }
}

int main(int argc, char **argv) {
  int n = argc > 1 ? atoi(argv[1]) : 512;
  int reps = argc > 2 ? atoi(argv[2]) : 20;
  int nx = n, ny = n, nz = n;
  if (argc > 5) { nx = atoi(argv[3]); ny = atoi(argv[4]); nz = atoi(argv[5]); }
  size_t vol = (size_t)(nx + 8) * (ny + 8) * (nz + 8);
  float *u, *m;
  CK(hipMalloc(&u, 3 * vol * sizeof(float)));
  CK(hipMalloc(&m, vol * sizeof(float)));
  fill<<<4096, 256>>>(u, 3 * vol, 1);
  fill<<<4096, 256>>>(m, vol, 7);
  CK(hipDeviceSynchronize());
  fdtd_geom g = {nx + 8, ny + 8, nz + 8, 0, nx - 1, 0, ny - 1, 0, nz - 1, 1e-3f, 0.1f, 0.1f, 0.1f, 0, 0, 0};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double cells = (double)nx * ny * nz;
  const char *names[] = {"flat", "naive", "march", "tile3d", "vec"};
  const char *sel = getenv("KBENCH_VARIANTS");
  for (int v = 0; v < 5; ++v) {
    if (sel && !strstr(sel, names[v])) continue;
    if (!sel && v != 2) continue;
    for (int w = 0; w < 3; ++w) fdtd_hip_stencil_step(&g, m, u, w % 3, (w + 2) % 3, (w + 1) % 3, v, 0);
    CK(hipDeviceSynchronize());
    std::vector<float> ms(reps);
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, 0));
      int rc = fdtd_hip_stencil_step(&g, m, u, r % 3, (r + 2) % 3, (r + 1) % 3, v, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      if (rc) { printf("rc=%d\n", rc); return 1; }
      CK(hipEventElapsedTime(&ms[r], e0, e1));
    }
    float best = 1e9, sum = 0;
    for (float x : ms) { if (x < best) best = x; sum += x; }
    float avg = sum / reps;
    printf("%-7s %dx%dx%d avg %.4f ms (%.1f Gcell/s, %.0f GB/s = %.1f%% of 8 TB/s)  best %.4f ms (%.0f GB/s)\n", names[v], nx, ny, nz, avg,
           cells / avg / 1e6, cells * 16 / avg / 1e6, cells * 16 / avg / 1e6 / 80.0, best, cells * 16 / best / 1e6);
  }
  if (getenv("KBENCH_FUSE")) {
    // Prototype of cache-resident time skewing: two time steps are interleaved slab by slab along x
    // (step t on slab k, then step t+1 on slab k-1) so that the second step finds its operands in the
    // Infinity Cache.  Reports time per step and checks the result against two plain sweeps.
    const int S = atoi(getenv("KBENCH_FUSE"));
    const int T = getenv("KBENCH_FUSE_T") ? atoi(getenv("KBENCH_FUSE_T")) : 2;
    auto sweep = [&](int time, int lo, int hi) {
      fdtd_geom gs = g; gs.x_m = lo; gs.x_M = hi;
      return fdtd_hip_stencil_step(&gs, m, u, time % 3, (time + 2) % 3, (time + 1) % 3, FDTD_VARIANT_MARCH, 0);
    };
    auto fused = [&](int time0) {
      // step s lags step s-1 by one slab
      const int K = (nx + S - 1) / S;
      for (int k = 0; k < K + T - 1; ++k)
        for (int s = 0; s < T; ++s) {
          const int kk = k - s;
          if (kk < 0 || kk >= K) continue;
          const int lo = kk * S, hi = (lo + S - 1 < nx - 1) ? lo + S - 1 : nx - 1;
          if (sweep(time0 + s, lo, hi)) { printf("sweep failed\n"); exit(1); }
        }
    };
    // correctness: T plain sweeps vs the fused schedule from the same start state
    float *save, *ref;
    CK(hipMalloc(&save, 3 * vol * sizeof(float)));
    CK(hipMalloc(&ref, 3 * vol * sizeof(float)));
    CK(hipMemcpy(save, u, 3 * vol * sizeof(float), hipMemcpyDeviceToDevice));
    for (int s = 0; s < T; ++s) sweep(s, 0, nx - 1);
    CK(hipMemcpy(ref, u, 3 * vol * sizeof(float), hipMemcpyDeviceToDevice));
    CK(hipMemcpy(u, save, 3 * vol * sizeof(float), hipMemcpyDeviceToDevice));
    fused(0);
    unsigned long long *cnt, h = 0;
    CK(hipMalloc(&cnt, 8)); CK(hipMemset(cnt, 0, 8));
    count_diff<<<4096, 256>>>((const unsigned *)ref, (const unsigned *)u, 3 * vol, cnt);
    CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
    printf("fuse T=%d slab=%d: differing words vs plain sweeps = %llu\n", T, S, h);
    CK(hipFree(save)); CK(hipFree(ref)); CK(hipFree(cnt));
    for (int w = 0; w < 2; ++w) fused(w * T);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) fused(r * T);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    float per_step = t / (reps * T);
    printf("fuse T=%d slab=%d: %.4f ms per step (%.1f Gcell/s, %.0f GB/s algorithmic = %.1f%% of 8 TB/s)\n", T, S, per_step,
           cells / per_step / 1e6, cells * 16 / per_step / 1e6, cells * 16 / per_step / 1e6 / 80.0);
  }
  if (getenv("KBENCH_VERIFY")) {
    // large-grid consistency: the marching kernel against the naive kernel, bit for bit, on level 2
    float *ref; unsigned long long *cnt, h = 0;
    CK(hipMalloc(&ref, vol * sizeof(float)));
    CK(hipMalloc(&cnt, 8));
    CK(hipMemset(cnt, 0, 8));
    fdtd_hip_stencil_step(&g, m, u, 0, 1, 2, FDTD_VARIANT_NAIVE, 0);
    CK(hipMemcpy(ref, u + 2 * vol, vol * sizeof(float), hipMemcpyDeviceToDevice));
    CK(hipMemset(u + 2 * vol, 0xff, vol * sizeof(float)));
    fdtd_hip_stencil_step(&g, m, u, 0, 1, 2, FDTD_VARIANT_MARCH, 0);
    // halo cells of level 2 were overwritten by the memset in this check; compare interior rows only
    // by re-running the naive kernel into the same poisoned buffer layout
    CK(hipMemcpy(ref, u + 2 * vol, vol * sizeof(float), hipMemcpyDeviceToDevice));
    CK(hipMemset(u + 2 * vol, 0xff, vol * sizeof(float)));
    fdtd_hip_stencil_step(&g, m, u, 0, 1, 2, FDTD_VARIANT_NAIVE, 0);
    count_diff<<<4096, 256>>>((const unsigned *)ref, (const unsigned *)(u + 2 * vol), vol, cnt);
    CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
    printf("verify %dx%dx%d (%zu cells per level): march vs naive differing words = %llu\n", nx, ny, nz, vol, h);
    CK(hipFree(ref)); CK(hipFree(cnt));
    if (h) return 2;
  }
  if (getenv("KBENCH_H16")) {
    // binary16 storage: reuse the first half of the u allocation as __half[3][vol]
    void *uh = (void *)u;
    // any finite data: m (vol floats) converted into each of the three half levels
    for (int l = 0; l < 3; ++l) fdtd_hip_convert_f32_to_h16(m, (char *)uh + (size_t)l * vol * 2, vol, 0);
    for (int w = 0; w < 3; ++w) fdtd_hip_stencil_step_h16(&g, m, uh, w % 3, (w + 2) % 3, (w + 1) % 3, 0);
    CK(hipDeviceSynchronize());
    float sum = 0, best = 1e9;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, 0));
      int rc = fdtd_hip_stencil_step_h16(&g, m, uh, r % 3, (r + 2) % 3, (r + 1) % 3, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      if (rc) { printf("rc=%d\n", rc); return 1; }
      float t; CK(hipEventElapsedTime(&t, e0, e1)); sum += t; if (t < best) best = t;
    }
    float avg = sum / reps;
    printf("h16     %dx%dx%d avg %.4f ms (%.1f Gcell/s, %.0f GB/s at 10 B/cell = %.1f%% of 8 TB/s)  best %.4f ms\n", nx, ny, nz,
           avg, cells / avg / 1e6, cells * 10 / avg / 1e6, cells * 10 / avg / 1e6 / 80.0, best);
  }
  if (getenv("KBENCH_STREAM_SWEEP")) {
    size_t n4 = vol / 4;
    const f32x4 *a = (const f32x4 *)u, *b = (const f32x4 *)(u + vol), *c = (const f32x4 *)m;
    f32x4 *d = (f32x4 *)(u + 2 * vol);
    auto timeit = [&](const char *name, double bytes, auto launch) {
      for (int w = 0; w < 2; ++w) launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); t /= reps;
      printf("%-28s %.4f ms  %.0f GB/s\n", name, t, bytes / t / 1e6);
    };
    double B4 = (double)vol * 16, B2 = (double)vol * 8, B1 = (double)vol * 4;
    if (getenv("KBENCH_STREAM_SWEEP")[0] == '2') {
      char nm[64];
      for (int g : {512, 1024, 2048, 4096}) {
        snprintf(nm, 64, "chunk U2 bs256 grid%d", g); timeit(nm, B4, [&] { stream_chunk<2, 256><<<g, 256>>>(a, b, c, d, n4); });
        snprintf(nm, 64, "chunk U4 bs256 grid%d", g); timeit(nm, B4, [&] { stream_chunk<4, 256><<<g, 256>>>(a, b, c, d, n4); });
        snprintf(nm, 64, "chunk U2 bs1024 grid%d", g); timeit(nm, B4, [&] { stream_chunk<2, 1024><<<g, 1024>>>(a, b, c, d, n4); });
      }
      for (int g : {32768, 65536, 131072}) {
        snprintf(nm, 64, "3R1W U2 grid%d", g); timeit(nm, B4, [&] { stream_u<2, false><<<g, 256>>>(a, b, c, d, n4); });
        snprintf(nm, 64, "3R1W U1 grid%d", g); timeit(nm, B4, [&] { stream_u<1, false><<<g, 256>>>(a, b, c, d, n4); });
        snprintf(nm, 64, "read3 U2 grid%d", g); timeit(nm, B1 * 3, [&] { read3_u<2><<<g, 256>>>(a, b, c, d, n4); });
        snprintf(nm, 64, "read U8 grid%d", g); timeit(nm, B1, [&] { read_u<8><<<g, 256>>>(a, d, n4); });
      }
      { unsigned g = (unsigned)((n4 + 255) / 256); timeit("oneshot U1", B4, [&] { stream_oneshot<1><<<g, 256>>>(a, b, c, d, n4); }); }
      { unsigned g = (unsigned)((n4 + 511) / 512); timeit("oneshot U2", B4, [&] { stream_oneshot<2><<<g, 256>>>(a, b, c, d, n4); }); }
      { unsigned g = (unsigned)((n4 + 1023) / 1024); timeit("oneshot U4", B4, [&] { stream_oneshot<4><<<g, 256>>>(a, b, c, d, n4); }); }
      for (int g : {2048, 16384}) { snprintf(nm, 64, "read3 U4 grid%d", g); timeit(nm, B1 * 3, [&] { read3_u<4><<<g, 256>>>(a, b, c, d, n4); }); }
    } else
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
      char nm[64];
      snprintf(nm, 64, "3R1W U1 grid%d", g); timeit(nm, B4, [&] { stream_u<1, false><<<g, 256>>>(a, b, c, d, n4); });
      snprintf(nm, 64, "3R1W U2 grid%d", g); timeit(nm, B4, [&] { stream_u<2, false><<<g, 256>>>(a, b, c, d, n4); });
      snprintf(nm, 64, "3R1W U4 grid%d", g); timeit(nm, B4, [&] { stream_u<4, false><<<g, 256>>>(a, b, c, d, n4); });
      snprintf(nm, 64, "3R1W U4 nt grid%d", g); timeit(nm, B4, [&] { stream_u<4, true><<<g, 256>>>(a, b, c, d, n4); });
      snprintf(nm, 64, "copy U4 grid%d", g); timeit(nm, B2, [&] { copy_u<4><<<g, 256>>>(a, d, n4); });
      snprintf(nm, 64, "copy U8 grid%d", g); timeit(nm, B2, [&] { copy_u<8><<<g, 256>>>(a, d, n4); });
      snprintf(nm, 64, "read U8 grid%d", g); timeit(nm, B1, [&] { read_u<8><<<g, 256>>>(a, d, n4); });
    }
  }
  if (!getenv("KBENCH_NO_STREAM")) {
    size_t n4 = vol / 4;
    const float4 *a = (const float4 *)u, *b = (const float4 *)(u + vol), *c = (const float4 *)m;
    float4 *d = (float4 *)(u + 2 * vol);
    for (int w = 0; w < 3; ++w) stream_3r1w<<<256 * 8, 256>>>(a, b, c, d, n4);
    CK(hipDeviceSynchronize());
    float tot = 0;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, 0));
      stream_3r1w<<<256 * 8, 256>>>(a, b, c, d, n4);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); tot += t;
    }
    float avg = tot / reps;
    printf("stream 3R+1W float4 over %zu padded cells: avg %.4f ms  %.0f GB/s\n", vol, avg, (double)vol * 16 / avg / 1e6);
  }
  return 0;
}
