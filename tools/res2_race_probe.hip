// Two launches of the split-binary16 Res2 chain (small windows) on two streams, different output buffers, same
// inputs: do the concurrent results equal the serial ones?  (tools only)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/res2_race_probe.hip wespeaker_amd/csrc/res2_fused.hip -o tools/bin/res2_race_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../wespeaker_amd/csrc/kernels.h"
using namespace wsamd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, T = 198, C = 512, W = 64, prec = argc > 2 ? atoi(argv[2]) : 1;
  const size_t n = (size_t)B * T * C;
  std::vector<float> h(n), wv(7 * 64 * 192), vv(7 * 3 * 64);
  for (auto& x : h) x = (float)((rand() % 2001) - 1000) / 1000.f;
  for (auto& x : wv) x = (float)((rand() % 2001) - 1000) / 8000.f;
  for (auto& x : vv) x = (float)((rand() % 2001) - 1000) / 1000.f;
  std::vector<uint16_t> wh(wv.size()), wl(wv.size());
  for (size_t i = 0; i < wv.size(); ++i) { wh[i] = f2h(wv[i]); _Float16 hh; memcpy(&hh, &wh[i], 2); wl[i] = f2h(wv[i] - (float)hh); }
  float *y1, *w, *v, *y2[3]; uint16_t *dwh, *dwl;
  CK(hipMalloc(&y1, n * 4)); CK(hipMalloc(&w, wv.size() * 4)); CK(hipMalloc(&v, vv.size() * 4));
  CK(hipMalloc(&dwh, wh.size() * 2)); CK(hipMalloc(&dwl, wl.size() * 2));
  for (int i = 0; i < 3; ++i) { CK(hipMalloc(&y2[i], n * 4)); CK(hipMemset(y2[i], 0, n * 4)); }
  CK(hipMemcpy(y1, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, wv.data(), wv.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(v, vv.data(), vv.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dwh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dwl, wl.data(), wl.size() * 2, hipMemcpyHostToDevice));
  auto params = [&](float* out, int dil) {
    Res2ChainParams p = {};
    p.y1 = y1; p.ldy1 = C; p.y2 = out; p.ldy2 = C; p.ldw = 192;
    for (int i = 0; i < 7; ++i) { p.w[i] = w + i * 64 * 192; p.wh[i] = dwh + i * 64 * 192; p.wl[i] = dwl + i * 64 * 192;
      p.bias[i] = v + i * 192; p.scale[i] = v + i * 192 + 64; p.shift[i] = v + i * 192 + 128; }
    p.B = B; p.T = T; p.W = W; p.dil = dil; p.prec = prec;
    return p;
  };
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  std::vector<float> r0(n), r1(n), r2(n);
  for (int dil = 2; dil <= 4; ++dil) {
    CK(launch_res2_chain(params(y2[0], dil), s1)); CK(hipDeviceSynchronize());
    CK(hipMemcpy(r0.data(), y2[0], n * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int rep = 0; rep < 5; ++rep) {
      for (int i = 0; i < 3; ++i) { CK(launch_res2_chain(params(y2[1], dil), s1)); CK(launch_res2_chain(params(y2[2], dil), s2)); }
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(r1.data(), y2[1], n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), y2[2], n * 4, hipMemcpyDeviceToHost));
      size_t d1 = 0, d2 = 0, first = n;
      for (size_t i = 0; i < n; ++i) { if (r0[i] != r1[i]) { ++d1; if (first == n) first = i; } if (r0[i] != r2[i]) ++d2; }
      if (d1 || d2) { ++bad; printf("dil %d rep %d: %zu / %zu elements differ (first at row %zu col %zu)\n", dil, rep, d1, d2, first / C, first % C); }
    }
    printf("B %d prec %d dil %d: %d of 5 concurrent repetitions differ from the serial run\n", B, prec, dil, bad);
  }
  return 0;
}
