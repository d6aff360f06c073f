/*
 * asan_driver.c — host-side memory check (AddressSanitizer, CPU only).  TEST INFRASTRUCTURE.
 *
 * Runs the CPU oracle (fdtd_oracle.c, built with -fsanitize=address) on small problems chosen for their
 * index arithmetic: sources in the outermost cell (scatter into the first halo layer, openacc.cpp:192),
 * sub-ranges, one-cell boxes, space orders 6 to 12; then, when built with FDTD_ASAN_WITH_ENGINE, the host
 * code of the engine library (libfdtd_hip.so built with -Xarch_host -fsanitize=address): input generators,
 * parity metrics, the shape logic of the two-step sweep and the operators' argument checks, which need no
 * GPU (an operator called without a device must return an error code, not crash).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/fdtd_abi.h"

int Oracle_Kernel_OpenACC(FDTD_OPERATOR_ARGS);
int Oracle_Kernel_SpaceOrder(FDTD_OPERATOR_ARGS, int order);

static int run_case(int order, int nx, int ny, int nz, int nt, int nsrc, const int *bounds) {
  const int H = order, nxp = nx + 2 * H, nyp = ny + 2 * H, nzp = nz + 2 * H;
  const size_t vol = (size_t)nxp * nyp * nzp;
  /* exact-size heap blocks: any access beyond an array is a report */
  float *u = malloc(3 * vol * sizeof(float)), *m = malloc(vol * sizeof(float));
  float *src = malloc((size_t)nt * (nsrc ? nsrc : 1) * sizeof(float));
  float *coords = malloc((size_t)(nsrc ? nsrc : 1) * 3 * sizeof(float));
  for (size_t i = 0; i < 3 * vol; ++i) u[i] = sinf((float)i * 0.01f);
  for (size_t i = 0; i < vol; ++i) m[i] = 1.5f + 0.001f * (float)(i % 97);
  for (int i = 0; i < nt * (nsrc ? nsrc : 1); ++i) src[i] = (float)(i % 7) - 3.0f;
  for (int p = 0; p < nsrc; ++p) {
    /* corners of the box: (0,0,0), the outermost cell, and points in between */
    const float f = nsrc > 1 ? (float)p / (float)(nsrc - 1) : 1.0f;
    coords[3 * p + 0] = f * ((float)(nx - 1) + 0.3f);
    coords[3 * p + 1] = (1.0f - f) * ((float)(ny - 1) + 0.6f);
    coords[3 * p + 2] = f * ((float)(nz - 1) + 0.9f);
  }
  int usz[4] = {3, nxp, nyp, nzp}, msz[3] = {nxp, nyp, nzp}, ssz[2] = {nt, nsrc ? nsrc : 1}, csz[2] = {nsrc ? nsrc : 1, 3};
  struct dataobj uv = {u, usz, 3 * vol * sizeof(float)}, mv = {m, msz, vol * sizeof(float)};
  struct dataobj sv = {src, ssz, 0}, cv = {coords, csz, 0};
  struct profiler t = {0, 0};
  const int b_def[6] = {0, nx - 1, 0, ny - 1, 0, nz - 1};
  const int *b = bounds ? bounds : b_def;
  int rc;
  if (order == 4)
    rc = Oracle_Kernel_OpenACC(&mv, &sv, &cv, &uv, b[1], b[0], b[3], b[2], b[5], b[4], 1e-3f, 1.0f, 0.9f, 1.1f, 0, 0, 0,
                               nsrc - 1, 0, nt - 1, 0, 0, 1, &t);
  else
    rc = Oracle_Kernel_SpaceOrder(&mv, &sv, &cv, &uv, b[1], b[0], b[3], b[2], b[5], b[4], 1e-3f, 1.0f, 0.9f, 1.1f, 0, 0, 0,
                                  nsrc - 1, 0, nt - 1, 0, 0, 1, &t, order);
  free(u); free(m); free(src); free(coords);
  return rc;
}

int main(void) {
  int fails = 0;
  const int sub[6] = {1, 5, 2, 6, 0, 9};
  const int one[6] = {3, 3, 3, 3, 3, 3};
  for (int order = 4; order <= 12; order += 2) {
    fails += run_case(order, 9, 8, 11, 7, 3, NULL) != 0;
    fails += run_case(order, 9, 8, 11, 7, 2, sub) != 0;
    fails += run_case(order, 7, 7, 7, 3, 0, one) != 0;
    fails += run_case(order, 1, 1, 1, 6, 1, NULL) != 0;
  }
  printf("oracle under AddressSanitizer: %s\n", fails ? "FAILED" : "ok");
#ifdef FDTD_ASAN_WITH_ENGINE
  {
    float src[40], coords[30 * 3];
    fdtd_fill_ricker(src, 20, 2, 1e-3f);
    fdtd_fill_source_coords(coords, 30, 16, 16, 16);
    const size_t volp = 12 * 12 * 12;
    float *u = malloc(3 * volp * sizeof(float)), *m = malloc(volp * sizeof(float)), *v = malloc(3 * volp * sizeof(float));
    fdtd_fill_correctness_inputs(u, m, volp);
    memcpy(v, u, 3 * volp * sizeof(float));
    v[5] += 1.0f;
    struct fdtd_parity p;
    fdtd_compare_fields(v, u, 3 * volp, &p);
    fails += !(p.max_abs > 0.9 && p.nan_count == 0);
    struct fdtd_geom g = {520, 520, 520, 0, 511, 0, 511, 0, 511, 1e-3f, 0.1f, 0.1f, 0.1f, 0, 0, 0};
    fails += fdtd_hip_two_steps_fuses(&g, 1, FDTD_VARIANT_AUTO) != 1;
    g.z_M = 509;
    fails += fdtd_hip_two_steps_fuses(&g, 1, FDTD_VARIANT_MARCH2) != 0;
    /* argument checks come before any device call; with no usable device the call returns an error code */
    int usz[4] = {3, 12, 12, 12}, bad[3] = {12, 12, 11}, ssz[2] = {0, 1}, csz[2] = {2, 0};
    struct dataobj uv = {u, usz, 0}, mv = {m, bad, 0}, sv = {NULL, ssz, 0}, cv = {NULL, csz, 0};
    struct profiler t = {0, 0};
    fails += Kernel_CUDA_Optimized(&mv, &sv, &cv, &uv, 3, 0, 3, 0, 3, 0, 1e-3f, 1, 1, 1, 0, 0, 0, -1, 0, 3, 0, 0, 1, &t) == 0;
    fails += Kernel_HIP_MultiGPU(&mv, &sv, &cv, &uv, 3, 0, 3, 0, 3, 0, 1e-3f, 1, 1, 1, 0, 0, 0, -1, 0, 3, 0, -1, 1, &t) == 0;
    fails += fdtd_hip_set_space_order(7) == 0;
    free(u); free(m); free(v);
    printf("engine host code under AddressSanitizer: %s\n", fails ? "FAILED" : "ok");
  }
#endif
  return fails ? 1 : 0;
}
