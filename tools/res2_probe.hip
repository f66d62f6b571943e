// Cycle stamps of the fused fp32 Res2 chain (one workgroup, wavefronts 0 and 4 = the two wavefronts of one SIMD):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWS_TRACE tools/res2_probe.hip wespeaker_amd/csrc/res2_fused.hip wespeaker_amd/csrc/res2_chain4.hip -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/bin/res2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "../wespeaker_amd/csrc/kernels.h"
using namespace wsamd;
namespace wsamd { unsigned long long* res2c4_trace_buffer_address(); }
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
  const int T0 = getenv("PROBE_T") ? atoi(getenv("PROBE_T")) : T;
  p.T = T0;
  // A/B: the eight-wavefront kernel (force_wave8) against the dispatcher's choice (T <= 208: res2_chain4_kernel):
  // whole-output bit compare for the three dilations, then steady-clock timing of both
  float* y2b; CK(hipMalloc(&y2b, (size_t)B * T * C * 4));
  std::vector<float> ha((size_t)B * T * C), hb((size_t)B * T * C);
  for (int dil = 2; dil <= 4; ++dil) {
    p.dil = dil;
    CK(hipMemset(y2, 0xff, (size_t)B * T * C * 4)); CK(hipMemset(y2b, 0xff, (size_t)B * T * C * 4));
    p.y2 = y2; p.force_wave8 = 1; CK(launch_res2_chain(p, 0));
    p.y2 = y2b; p.force_wave8 = 0; CK(launch_res2_chain(p, 0));
    CK(hipMemcpy(ha.data(), y2, ha.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), y2b, hb.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0, nanb = 0;
    for (size_t i = 0; i < ha.size(); ++i) { diff += memcmp(&ha[i], &hb[i], 4) != 0; nanb += hb[i] != hb[i]; }
    printf("dil %d T %d: %zu of %zu words differ between the two kernels (%zu NaN words in B = untouched 0xff fill)\n", dil, T0, diff, ha.size(), nanb);
  }
  p.y2 = y2; p.dil = 2;
  for (int rep = 0; rep < 2; ++rep)
    for (int f8 = 1; f8 >= 0; --f8) {
      p.force_wave8 = f8;
      for (int i = 0; i < 40; ++i) CK(launch_res2_chain(p, 0));
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; ++i) CK(launch_res2_chain(p, 0));
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("res2 chain 256 x %d x 64, %s: %.1f us per launch (ideal MFMA time at 208 rows 58.2 us)\n", T0,
             f8 ? "eight wavefronts" : "four wavefronts ", ms * 50.0);
    }
  unsigned long long tr[64];
  CK(hipMemcpy(tr, res2c4_trace_buffer_address(), sizeof(tr), hipMemcpyDeviceToHost));
  for (int wv = 0; wv < 1; ++wv) {
    const unsigned long long* t = tr + 64 * wv;
    printf("wave %d: zero+stage+weights %lld | ", 4 * wv, (long long)(t[1] - t[0]));
    for (int s = 0; s < 7; ++s)
      printf("step %d: mfma %lld bar %lld epi %lld bar %lld | ", s, (long long)(t[3 + 5 * s] - t[2 + 5 * s]), (long long)(t[4 + 5 * s] - t[3 + 5 * s]),
             (long long)(t[5 + 5 * s] - t[4 + 5 * s]), (long long)(t[6 + 5 * s] - t[5 + 5 * s]));
    printf("total %lld cycles\n", (long long)(t[40] - t[0]));
    printf("step 3, pair starts relative to the step's start:");
    for (int q = 0; q < 7; ++q) printf(" %lld", (long long)(t[41 + q] - t[2 + 5 * 3]));
    printf(" | end of MFMAs %lld\n", (long long)(t[3 + 5 * 3] - t[2 + 5 * 3]));
  }
  return 0;
}
