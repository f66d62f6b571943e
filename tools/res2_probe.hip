// Cycle stamps of the fused fp32 Res2 chain (one workgroup, wavefronts 0 and 4 = the two wavefronts of one SIMD):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWS_TRACE tools/res2_probe.hip wespeaker_amd/csrc/res2_fused.hip -o tools/bin/res2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../wespeaker_amd/csrc/kernels.h"
using namespace wsamd;
namespace wsamd { unsigned long long* res2_trace_buffer_address(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main() {
  const int B = 256, T = 198, C = 512, W = 64;
  float *y1, *y2, *w, *v;
  CK(hipMalloc(&y1, (size_t)B * T * C * 4)); CK(hipMalloc(&y2, (size_t)B * T * C * 4));
  CK(hipMalloc(&w, 7 * 64 * 192 * 4)); CK(hipMalloc(&v, 7 * 3 * 64 * 4));
  std::vector<float> h((size_t)B * T * C);
  for (auto& x : h) x = (float)((rand() % 2001) - 1000) / 1000.f;
  CK(hipMemcpy(y1, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), 7 * 64 * 192 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(v, h.data() + 5000, 7 * 3 * 64 * 4, hipMemcpyHostToDevice));
  Res2ChainParams p = {};
  p.y1 = y1; p.ldy1 = C; p.y2 = y2; p.ldy2 = C; p.ldw = 192;
  for (int i = 0; i < 7; ++i) { p.w[i] = w + i * 64 * 192; p.bias[i] = v + i * 192; p.scale[i] = v + i * 192 + 64; p.shift[i] = v + i * 192 + 128; }
  p.B = B; p.T = T; p.W = W; p.dil = 2; p.prec = 0;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) CK(launch_res2_chain(p, 0));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 20; ++i) CK(launch_res2_chain(p, 0));
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("res2 chain 256 x 198 x 64: %.1f us per launch (ideal MFMA time at 224 rows 62.7 us)\n", ms * 50.0);
  unsigned long long tr[128];
  CK(hipMemcpy(tr, res2_trace_buffer_address(), sizeof(tr), hipMemcpyDeviceToHost));
  for (int wv = 0; wv < 2; ++wv) {
    const unsigned long long* t = tr + 64 * wv;
    printf("wave %d: zero+stage+weights %lld | ", 4 * wv, (long long)(t[1] - t[0]));
    for (int s = 0; s < 7; ++s)
      printf("step %d: mfma %lld bar %lld epi %lld bar %lld | ", s, (long long)(t[3 + 5 * s] - t[2 + 5 * s]), (long long)(t[4 + 5 * s] - t[3 + 5 * s]),
             (long long)(t[5 + 5 * s] - t[4 + 5 * s]), (long long)(t[6 + 5 * s] - t[5 + 5 * s]));
    printf("total %lld cycles\n", (long long)(t[40] - t[0]));
  }
  return 0;
}
