// Where does the dispatcher put the workgroups of a big launch?  (not part of the product)
//   hipcc --offload-arch=gfx950 -O2 tools/placement_probe.hip -o tools/bin/placement_probe
// Each workgroup (256 threads, 73 KB of LDS: the footprint of the fp32 128x128 GEMM tile, two per CU) records
// its XCC / SE / CU ids (HW_ID, XCC_ID registers), its start and end clock, then spins ~20 us.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
  extern __shared__ float lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long t0 = wall_clock64();
  float acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) { acc = acc * 1.0001f + 0.5f; lds[threadIdx.x] = acc; }
  unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = hw;
    out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = (unsigned)t0;
    out[blockIdx.x * 4 + 3] = (unsigned)(t1 - t0) + (acc == 1.2345f);
  }
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1584, spin = argc > 2 ? atoi(argv[2]) : 4000;
  unsigned* d; CK(hipMalloc(&d, blocks * 16));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 73728, 0, d, spin); CK(hipDeviceSynchronize()); }
  std::vector<unsigned> h(blocks * 4); CK(hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost));
  unsigned tmin = ~0u; for (int b = 0; b < blocks; ++b) if (h[b * 4 + 2] < tmin) tmin = h[b * 4 + 2];
  // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx94x: se [15:13])
  printf("# bid xcc se sh cu start_ticks dur_ticks (100 MHz wall clock)\n");
  for (int b = 0; b < blocks; ++b) {
    unsigned hw = h[b * 4];
    printf("%d %u %u %u %u %u %u\n", b, h[b * 4 + 1] & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, h[b * 4 + 2] - tmin, h[b * 4 + 3]);
  }
  return 0;
}
