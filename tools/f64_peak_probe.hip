// Register-only burner of v_mfma_f64_16x16x4_f64: what the chip sustains on the instruction the PLDA GEMM is built on
// (the 78.6 TFLOP/s FP64-matrix figure bench.py prices plda_gemm_f64 against is a datasheet
// number, the local guide has no FP64 row).  Every wavefront runs 8 independent accumulator chains, no memory traffic;
// the shader clock is sampled by the same wavefronts (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/f64_peak_probe.hip -o tools/bin/f64_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void burn(double* out, int iters, unsigned long long* clk) {
  f64x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f64x4){0., 0., 0., 0.};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double* out; unsigned long long* clk;
  for (int wpc = 1; wpc <= 8; wpc *= 2) {                     // workgroups of 4 wavefronts per CU: 1, 2, 4, 8 wavefronts per SIMD
    const int blocks = cus * wpc, iters = 200000;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 8)); CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, 0, out, iters / 10, clk);
    CK(hipDeviceSynchronize());
    double best = 0, mhz = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
      const double flops = (double)blocks * 4 * iters * 8 * 2048.0;      // 16 x 16 x 4 x 2 per instruction
      const double tf = flops / (ms * 1e-3) / 1e12;
      if (tf > best) { best = tf; mhz = (double)h[0] / (double)h[1] * 100.0; }
    }
    const double cyc_per_mfma = (cus * 4.0 * 2048.0 * mhz * 1e6) / (best * 1e12);
    printf("{\"instruction\": \"v_mfma_f64_16x16x4_f64\", \"cus\": %d, \"wavefronts_per_simd\": %d, \"tflops\": %.2f, "
           "\"shader_clock_mhz\": %.0f, \"cycles_per_mfma_per_simd\": %.2f, \"peak_at_this_clock_if_64_cycles\": %.2f}\n",
           cus, wpc, best, mhz, cyc_per_mfma, cus * 4.0 * 2048.0 / 64.0 * mhz * 1e6 / 1e12);
    CK(hipFree(out)); CK(hipFree(clk));
  }
  return 0;
}
