/*
 * Device-query names the reference's benchmark driver uses (main.cpp:24,155-199), mapped onto
 * the HIP runtime so that the driver compiles UNCHANGED on a ROCm box and links against
 * libfdtd_hip.so.  This header is on the include path only when oracle/Makefile builds the
 * reference driver under oracle/_ref/; no engine source includes it.
 * Every field the driver reads (name, major, minor, memoryClockRate, memoryBusWidth, clockRate,
 * multiProcessorCount) exists in hipDeviceProp_t under the same name.
 */
#pragma once
#include <hip/hip_runtime_api.h>

typedef hipError_t cudaError_t;
typedef hipDeviceProp_t cudaDeviceProp;
#define cudaSuccess hipSuccess

static inline cudaError_t cudaGetDeviceCount(int *count) { return hipGetDeviceCount(count); }
static inline cudaError_t cudaGetDeviceProperties(cudaDeviceProp *prop, int device) {
  return hipGetDeviceProperties(prop, device);
}
