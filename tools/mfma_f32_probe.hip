// Micro-probe: what keeps v_mfma_f32_32x32x2_f32 below its 64-cycle issue rate inside a GEMM loop?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_probe.hip -o tools/bin/mfma_f32_probe
// variants: 0 registers only | 1 + LDS fragment reads (16 ds_read_b128 per 64 MFMAs) | 2 + a workgroup
// barrier per 64 MFMAs | 3 + 8 ds_write_b128 per 64 MFMAs | 4 + 8 global_load_dwordx4 per 64 MFMAs (1 KB
// contiguous per wave instruction) | 5 the same loads with the GEMM's row-strided pattern (8 rows x 128 B per
// wave instruction, 6 KB row stride, a new 128-row panel every 48 iterations)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ g, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 18432; i += 256) { unsigned h = (unsigned)i * 2654435761u + (unsigned)blockIdx.x * 40503u; h ^= h >> 13; lds[i] = RANDOM_DATA ? ((float)(h & 0xffff) / 32768.f - 1.f) : (float)(i & 15) * 0.01f; }
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f32x4 fa[2], fb[2];
  fa[0] = fa[1] = fb[0] = fb[1] = (f32x4){0.5f, 0.25f, 0.125f, 1.f};
  f32x4 ld[8];
  for (int j = 0; j < 8; ++j) ld[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* gp = g + (size_t)(blockIdx.x * 256 + tid) * 4;
  const int li = lane & 31, lh = lane >> 5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
      if (V >= 1) {
        const float* As = lds + (li) * 36 + lh * 4 + grp * 8;
        fa[0] = *reinterpret_cast<const f32x4*>(As);
        fa[1] = *reinterpret_cast<const f32x4*>(As + 32 * 36);
        fb[0] = *reinterpret_cast<const f32x4*>(As + 128 * 36);
        fb[1] = *reinterpret_cast<const f32x4*>(As + 160 * 36);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int im = 0; im < 2; ++im)
#pragma unroll
          for (int in = 0; in < 2; ++in) {
            acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[in][s], fa[im][s], acc[im][in], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int i = s * 4 + im * 2 + in;
            if (V == 4 && grp == 0 && i < 8) ld[i] = *reinterpret_cast<const f32x4*>(gp + (size_t)it * 1024 * 256 + i * 64);
            if (V == 5 && grp == 0 && i < 8) {
              const size_t panel = ((size_t)blockIdx.x * 37 + (size_t)(it / 48)) % 2000;
              ld[i] = *reinterpret_cast<const f32x4*>(g + (panel * 128 + (size_t)(tid >> 3) + 32 * (i & 3)) * 1536 +
                                                     (size_t)(it % 48) * 32 + (tid & 7) * 4 + (i >> 2) * 0);
            }
            if (V >= 3 && grp == 2 && i < 8)
              *reinterpret_cast<f32x4*>(&lds[9216 + ((tid >> 3) + 32 * i) * 36 + (tid & 7) * 4]) = ld[i];
            __builtin_amdgcn_sched_barrier(0);
          }
      if (V >= 2 && grp == 2) __syncthreads();
    }
  }
  float sum = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
  for (int j = 0; j < 8; ++j) sum += ld[j][0];
  out[blockIdx.x * 256 + tid] = sum;
}

template <int V>
void run(const float* g, float* out, int blocks_per_cu) {
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  const size_t lds = blocks_per_cu == 1 ? 100 * 1024 : 73728;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<V><<<blocks, 256, lds>>>(g, out, 10);
  hipEventRecord(e0);
  probe<V><<<blocks, 256, lds>>>(g, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
  printf("variant %d, %d workgroup(s)/CU: %.1f TF (%.3f of 157.3)\n", V, blocks_per_cu, flops / (ms * 1e-3) / 1e12,
         flops / (ms * 1e-3) / 157.3e12);
}

int main() {
  float *g, *out;
  hipMalloc(&g, (size_t)2100 * 1024 * 256 * 4 + (1 << 20)); hipMalloc(&out, 512 * 256 * 4);
  hipMemset(g, 0, (size_t)2100 * 1024 * 256 * 4 + (1 << 20));
  for (int bpc = 1; bpc <= 2; ++bpc) {
    run<0>(g, out, bpc); run<1>(g, out, bpc); run<2>(g, out, bpc); run<3>(g, out, bpc); run<4>(g, out, bpc);
    run<5>(g, out, bpc);
  }
  return 0;
}
