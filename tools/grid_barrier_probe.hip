// grid_barrier_probe.hip — what a device-wide barrier costs on this GPU, as the price of a persistent
// multi-step kernel for small grids (one launch per call, a barrier per time step) against one launch
// per step.  `nwg` workgroups of 256 threads, all resident (nwg <= number of compute units); every
// iteration: workgroup barrier, one agent-scope release increment of a counter by thread 0, spin on an
// agent-scope acquire load until all workgroups of the iteration have arrived, workgroup barrier.  The
// release / acquire pair is what makes the cells one XCD wrote visible to the others (L2 write-back and
// invalidate), so it belongs to the price.  Spins are bounded: a workgroup that waits too long sets a flag
// and leaves, and so does everybody after it — the grid always drains.
//
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe tools/grid_barrier_probe.hip && ./grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(e)                                                                  \
  do {                                                                            \
    hipError_t _e = (e);                                                          \
    if (_e != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

template <bool FENCED>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned *counter, int *failed, float *data, int iters, int work) {
  const unsigned nwg = gridDim.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // a little streaming work per step, so that there is something for the fences to write back
    for (int k = 0; k < work; ++k) {
      const size_t i = ((size_t)k * nwg + blockIdx.x) * 256 + threadIdx.x;
      acc += data[i];
      data[i] = acc * 0.5f + (float)it;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned target = (unsigned)(it + 1) * nwg;
      if (FENCED) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (true) {
        const unsigned seen = FENCED ? __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                     : __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen >= target) break;
        if (++spins > 2000000 || __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
    if (__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  }
  if (acc == 123.456f) data[0] = acc;
}

__global__ void empty_kernel(float *data) {
  if (data[threadIdx.x] == 123.456f) data[0] = 1.f;
}

int main() {
  int dev = 0, cus = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  unsigned *counter;
  int *failed;
  float *data;
  const int max_work = 16;
  CHECK(hipMalloc(&counter, 4));
  CHECK(hipMalloc(&failed, 4));
  CHECK(hipMalloc(&data, (size_t)max_work * cus * 256 * sizeof(float)));
  CHECK(hipMemset(data, 0, (size_t)max_work * cus * 256 * sizeof(float)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 2000;
  printf("%d compute units; %d barriers per launch\n", cus, iters);
  for (int nwg : {32, 64, 128, cus}) {
    for (int work : {0, 4, 16}) {
      for (int fenced = 0; fenced < 2; ++fenced) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          CHECK(hipMemset(counter, 0, 4));
          CHECK(hipMemset(failed, 0, 4));
          CHECK(hipEventRecord(e0, 0));
          if (fenced) hipLaunchKernelGGL(barrier_kernel<true>, dim3(nwg), dim3(256), 0, 0, counter, failed, data, iters, work);
          else hipLaunchKernelGGL(barrier_kernel<false>, dim3(nwg), dim3(256), 0, 0, counter, failed, data, iters, work);
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipEventSynchronize(e1));
          int f = 0;
          CHECK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
          if (f) {
            printf("nwg %d: a barrier timed out\n", nwg);
            return 2;
          }
          float ms = 0;
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        printf("workgroups %3d  loads+stores per thread and step %2d  %s  %.2f us per step\n", nwg, work,
               fenced ? "release/acquire" : "relaxed        ", best * 1e3f / iters);
      }
    }
  }
  // for comparison: back-to-back dependent launches of an empty kernel
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(empty_kernel, dim3(cus), dim3(256), 0, 0, data);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("empty kernel, %d workgroups, launched back to back: %.2f us per launch\n", cus, ms * 1e3f / iters);
  return 0;
}
