// LDS canary (tools only): workgroups fill their LDS with a pattern, idle for a while and check it.  Built as a shared
// library with a C entry so that a Python test can run it on one stream while an engine runs on another: does any of
// the engine's kernels write outside its own LDS allocation?
#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" {
__global__ __launch_bounds__(256) void lds_canary_kernel(int words, long long spin, int* bad, int* first) {
  extern __shared__ unsigned int cz[];
  for (int i = threadIdx.x; i < words; i += 256) cz[i] = 0xC0FFEE00u ^ (unsigned)i;
  __syncthreads();
  const long long t0 = clock64();
  while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  int nb = 0;
  for (int i = threadIdx.x; i < words; i += 256)
    if (cz[i] != (0xC0FFEE00u ^ (unsigned)i)) { ++nb; atomicMin(first, i); }
  if (nb) atomicAdd(bad, nb);
}
int lds_canary_launch(int grid, int lds_bytes, long long spin, int* bad, int* first, void* stream) {
  static int set = 0;
  if (set < lds_bytes) { hipFuncSetAttribute((const void*)lds_canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); set = lds_bytes; }
  hipLaunchKernelGGL(lds_canary_kernel, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, lds_bytes / 4, spin, bad, first);
  return (int)hipGetLastError();
}
}
