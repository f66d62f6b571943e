// v_mfma_f32_16x16x4_f32 with LDS fragment reads in the stream (tools only): RPM = ds_read_b128 per 8 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int READS, int WIDTH>      // READS b128 (WIDTH 16) / b64 (WIDTH 8) reads per 8 MFMAs
__global__ void k(float* out, unsigned long long* cyc, int iters, float bval) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1.0f / (1 + i);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const char* base = reinterpret_cast<const char*>(lds) + (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) * 16);
  f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x4 f[2][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) f[0][r] = f[1][r] = *reinterpret_cast<const f32x4*>(base + r * 2048);
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int cur = 0; cur < 2; ++cur) {
      const char* b2 = base + cur * 8192;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[cur][m >> 1][m & 3], bval, acc[m & 1], 0, 0, 0);
        if (WIDTH == 16) {
          if (m < READS) f[cur ^ 1][m & 3] = *reinterpret_cast<const f32x4*>(b2 + m * 2048);
        } else {
          if (m < READS) {
            const float2 v = *reinterpret_cast<const float2*>(b2 + m * 2048);
            f[cur ^ 1][m & 3][0] = v.x; f[cur ^ 1][m & 3][1] = v.y;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int READS, int WIDTH>
void run(int waves, float* out, unsigned long long* cyc) {
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<READS, WIDTH>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 0.5f);
    CK(hipDeviceSynchronize());
  }
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("reads of %2d B per 8 MFMAs: %d, waves/SIMD %d: %.1f cycles per MFMA per SIMD\n", WIDTH, READS, waves / 4,
         c / ((double)iters * 8 * (waves / 4)));
}
int main() {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 8));
  run<0, 16>(4, out, cyc); run<1, 16>(4, out, cyc); run<2, 16>(4, out, cyc); run<4, 16>(4, out, cyc);
  run<0, 16>(8, out, cyc); run<1, 16>(8, out, cyc); run<2, 16>(8, out, cyc); run<4, 16>(8, out, cyc);
  run<2, 8>(4, out, cyc); run<4, 8>(4, out, cyc); run<2, 8>(8, out, cyc); run<4, 8>(8, out, cyc);
  return 0;
}
