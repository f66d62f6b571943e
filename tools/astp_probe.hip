// Timing / stamp probe of astp_fused_kernel (tools only): random h, W1, W2; steady-clock timing; WS_TRACE builds print
// the cycle stamps of workgroup 9 (wavefronts 0 and 4).
#include "../wespeaker_amd/csrc/astp_fused.hip"
#include <hip/hip_runtime.h>
#include <vector>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 198;
  const size_t nh = (size_t)B * T * 1536, nw1 = 128 * 4608, nw2 = 1536 * 128;
  std::vector<float> h(nh), w1(nw1), w2(nw2), bi((size_t)B * 128);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : h) v = rnd() * 2.f;
  for (auto& v : w1) v = rnd() * 0.05f;
  for (auto& v : w2) v = rnd() * 0.2f;
  for (auto& v : bi) v = rnd();
  float *dh, *dw1, *dw2, *dbi, *dp;
  CK(hipMalloc(&dh, nh * 4)); CK(hipMalloc(&dw1, nw1 * 4)); CK(hipMalloc(&dw2, nw2 * 4)); CK(hipMalloc(&dbi, bi.size() * 4));
  CK(hipMalloc(&dp, (size_t)B * 3072 * 4));
  CK(hipMemcpy(dh, h.data(), nh * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw1, w1.data(), nw1 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw2, w2.data(), nw2 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbi, bi.data(), bi.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  auto run = [&]() { CK(wsamd::launch_astp_fused(dh, 1536, B, T, dw1, 4608, nullptr, dbi, dw2, 128, dp, nullptr, st)); };
  for (int i = 0; i < 30; ++i) run();
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i) run();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("B %d T %d: %.1f us per launch  (%.1f TF)\n", B, T, ms * 50.f, 2.0 * B * T * 128 * 1536 * 2 / (ms / 20 * 1e-3) * 1e-12);
  }
  // reference for utterance 3, column slice check on the host (double)
  std::vector<float> pooled((size_t)B * 3072);
  CK(hipMemcpy(pooled.data(), dp, pooled.size() * 4, hipMemcpyDeviceToHost));
  {
    const int u = B > 3 ? 3 : 0;
    std::vector<double> H((size_t)T * 128);
    for (int t = 0; t < T; ++t) for (int n = 0; n < 128; ++n) {
      double a = bi[(size_t)u * 128 + n];
      const float* x = &h[((size_t)u * T + t) * 1536];
      for (int k = 0; k < 1536; ++k) a += (double)x[k] * w1[(size_t)n * 4608 + k];
      H[(size_t)t * 128 + n] = tanh(a);
    }
    double worst = 0;
    for (int c = 0; c < 1536; c += 7) {
      std::vector<double> lg(T); double mx = -1e300;
      for (int t = 0; t < T; ++t) { double a = 0; for (int n = 0; n < 128; ++n) a += H[(size_t)t * 128 + n] * w2[(size_t)c * 128 + n]; lg[t] = a; mx = a > mx ? a : mx; }
      double s0 = 0, s1 = 0, s2 = 0;
      for (int t = 0; t < T; ++t) { double e = exp(lg[t] - mx), x = h[((size_t)u * T + t) * 1536 + c]; s0 += e; s1 += e * x; s2 += e * x * x; }
      double mean = s1 / s0, var = s2 / s0 - mean * mean, sd = sqrt(var > 1e-7 ? var : 1e-7);
      worst = fmax(worst, fabs(mean - pooled[(size_t)u * 3072 + c]));
      worst = fmax(worst, fabs(sd - pooled[(size_t)u * 3072 + 1536 + c]));
    }
    printf("max abs err vs double reference (utt %d, every 7th channel): %.3g\n", u, worst);
  }
#ifdef WS_TRACE
  unsigned long long tr[128];
  CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(wsamd::g_astp_trace), sizeof(tr)));
  for (int w = 0; w < 2; ++w) {
    printf("wave %d stamps (cycles from slot 0):", w * 4);
    for (int i = 0; i < 60; ++i) if (tr[w * 64 + i]) printf(" [%d]%lld", i, (long long)(tr[w * 64 + i] - tr[w * 64]));
    printf("\n");
  }
#endif
  return 0;
}
