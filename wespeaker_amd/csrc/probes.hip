// Measurement probes (bench.py only; no reference counterpart -- the reference has the wall-clock Timer of
// runtime/core/utils/timer.h:22-36 and nothing that looks at the device's clock).
//
// clock_probe_kernel: ONE wavefront that samples the shader-clock counter (s_memtime, what __builtin_readcyclecounter
// returns on gfx9) against the constant-rate counter (s_memrealtime) every `period` ticks of the latter, sleeping in
// between.  Launched on a side stream next to the bench's sustained window it co-resides with the persistent GEMM
// workgroups (no LDS, a handful of registers) and records the clock the chip actually HELD under that load:
//   shader MHz over sample i  =  (memtime[i] - memtime[i-1]) / (realtime[i] - realtime[i-1]) * realtime_MHz,
// with realtime_MHz calibrated by the caller from the first and last sample against HIP-event time.
#include "kernels.h"

namespace wsamd {

__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ out, int samples,
                                                         unsigned long long period_ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < samples; ++i) {
    const unsigned long long target = r0 + (unsigned long long)i * period_ticks;
    while (__builtin_amdgcn_s_memrealtime() < target) __builtin_amdgcn_s_sleep(64);
    const unsigned long long c = __builtin_readcyclecounter();
    const unsigned long long r = __builtin_amdgcn_s_memrealtime();
    out[2 * i] = c;
    out[2 * i + 1] = r;
  }
}

hipError_t launch_clock_probe(unsigned long long* out, int samples, unsigned long long period_ticks, hipStream_t stream) {
  if (samples <= 0) return hipSuccess;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, out, samples, period_ticks);
  return hipGetLastError();
}

}  // namespace wsamd
