// Issue rate of v_mfma_f32_16x16x4_f32 (tools only): NACC independent accumulators round-robin, 1 or 2 wavefronts per
// SIMD, no memory traffic.  Prints cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int NACC, bool BIG>
__global__ void k(float* out, unsigned long long* cyc, int iters, float a, float b) {
  unsigned long long t0 = __builtin_readcyclecounter();
  if constexpr (!BIG) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC, bool BIG>
void run(int waves, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 1.0f, 0.5f);
    CK(hipDeviceSynchronize());
  }
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double per_simd = (double)iters * 8 * NACC * (waves / 4);
  printf("%s NACC %2d waves/SIMD %d: %.1f cycles per MFMA per SIMD\n", BIG ? "32x32x2" : "16x16x4", NACC, waves / 4, c / per_simd);
}
int main() {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 8));
  run<1, false>(4, out, cyc); run<2, false>(4, out, cyc); run<4, false>(4, out, cyc); run<13, false>(4, out, cyc);
  run<1, false>(8, out, cyc); run<2, false>(8, out, cyc); run<4, false>(8, out, cyc); run<13, false>(8, out, cyc);
  run<1, true>(4, out, cyc); run<2, true>(4, out, cyc); run<4, true>(4, out, cyc);
  run<1, true>(8, out, cyc); run<2, true>(8, out, cyc); run<4, true>(8, out, cyc);
  return 0;
}
