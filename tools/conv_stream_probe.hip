// Stand-alone probe of the CONV form of the persistent fp32 GEMM (3x3 / stride 1 / pad 1 layers of the ResNets) against
// the implicit-GEMM tile kernels (not part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_stream_probe.hip wespeaker_amd/csrc/conv_gemm.hip \
//         wespeaker_amd/csrc/gemm_f32_stream.hip wespeaker_amd/csrc/conv3x3_direct.hip wespeaker_amd/csrc/ecapa_ops.hip \
//         -o tools/bin/conv_stream_probe
// Whole-output bit compare (WS_STREAM_CONV 0 / 1), which 256-row tiles differ, then timing with a warm clock.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../wespeaker_amd/csrc/kernels.h"
using namespace wsamd;
namespace wsamd {
extern int g_ws_stream_conv;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Case { int B, H, W, Cin, N, res; const char* name; };

int main(int argc, char** argv) {
  std::vector<Case> cases = {
      {256, 20, 50, 128, 128, 0, "ResNet34 stage 3 (B=256)"},
      {512, 20, 50, 128, 128, 1, "ResNet34 stage 3 +res (B=512)"},
      {512, 10, 25, 256, 256, 0, "ResNet34 stage 4 (B=512)"},
      {256, 20, 50, 128, 128, 1, "ResNet34 stage 3 +res (B=256)"},
      {512, 20, 50, 128, 128, 0, "ResNet34 stage 3 (B=512)"},
      {128, 40, 99, 64, 64, 0, "ResNet34 stage 2 (B=128): 256x64 tiles"},
      {128, 40, 99, 64, 64, 1, "ResNet34 stage 2 +res (B=128): 256x64 tiles"},
      {96, 20, 50, 128, 128, 0, "1.46 rounds (not taken)"},
  };
  const size_t maxRows = 512ull * 1000;
  float *A, *W, *D, *R, *Z, *bias;
  CK(hipMalloc(&A, maxRows * 128 * 4 + 4096)); CK(hipMemset(A + maxRows * 128, 0, 4096)); CK(hipMalloc(&W, 256ull * 2304 * 4)); CK(hipMalloc(&D, maxRows * 128 * 4));
  CK(hipMalloc(&R, maxRows * 128 * 4)); CK(hipMalloc(&Z, 256)); CK(hipMemset(Z, 0, 256)); CK(hipMalloc(&bias, 256 * 4));
  std::vector<float> h(maxRows * 128);
  srand(7);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((rand() % 2001) - 1000) / 1000.f;
  CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(R, h.data() + 999, (h.size() - 999) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data() + 12345, 256ull * 2304 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, h.data() + 777, 256 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& c : cases) {
    ConvGemmParams p; memset(&p, 0, sizeof(p));
    const int M = c.B * c.H * c.W, K = 9 * c.Cin;
    p.prec = 0; p.A = A; p.lda = c.Cin; p.W = W; p.ldw = K; p.D = D; p.ldd = c.N;
    p.M = M; p.N = c.N; p.K = K; p.Cin = c.Cin;
    p.Hin = p.Hout = c.H; p.Win = p.Wout = c.W; p.stride_h = p.stride_w = 1; p.kh = p.kw = 3; p.dil_h = p.dil_w = 1;
    p.pad_h = p.pad_w = 1; p.bias = bias; p.act = ACT_RELU; p.splitk = 1; p.zeros = Z;
    if (c.res) { p.residual = R; p.ldr = c.N; }
    p.a_zero_off = (long long)maxRows * 128 * 4;         // zeros behind the input buffer
    const size_t nD = (size_t)M * c.N;
    std::vector<float> ref(nD), got(nD);
    printf("== %s: M=%d N=%d K=%d\n", c.name, M, c.N, K);
    for (int mode = 0; mode < 2; ++mode) {
      g_ws_stream_conv = mode;
      for (int rep = 0; rep < (mode ? 2 : 1); ++rep) {
        CK(hipMemset(D, 0xff, nD * 4));
        CK(launch_conv_gemm(p, 0)); CK(hipDeviceSynchronize());
        CK(hipMemcpy((mode ? got : ref).data(), D, nD * 4, hipMemcpyDeviceToHost));
        if (!mode) continue;
        size_t nd = 0, first = (size_t)-1;
        const int tiles = (M + 255) / 256;
        std::vector<int> bad(tiles, 0);
        for (size_t i = 0; i < nD; ++i)
          if (!(got[i] == ref[i])) { ++nd; if (first == (size_t)-1) first = i; bad[i / c.N / 256] = 1; }
        int nbad = 0, firstbad = -1, lastgood = -1;
        for (int t = 0; t < tiles; ++t) { if (bad[t]) { ++nbad; if (firstbad < 0) firstbad = t; } else lastgood = t; }
        printf("  conv form rep %d: D differs at %zu of %zu; first at m=%zu n=%zu; bad row tiles %d of %d (first %d, last good %d)\n",
               rep, nd, nD, first == (size_t)-1 ? 0 : first / c.N, first == (size_t)-1 ? 0 : first % c.N, nbad, tiles,
               firstbad, lastgood);
        if (nd && argc > 1) {
          // rows of the first bad tile that differ, and by how much
          const size_t m0 = (size_t)firstbad * 256;
          for (int r = 0; r < 256; r += 8) {
            int cnt = 0; double worst = 0;
            for (int rr = r; rr < r + 8; ++rr)
              for (int n = 0; n < c.N; ++n) {
                const size_t i = (m0 + rr) * c.N + n;
                if (!(got[i] == ref[i])) { ++cnt; worst = fmax(worst, fabs((double)got[i] - ref[i])); }
              }
            printf("    rows %3d..%3d: %4d differ, worst %.3g\n", r, r + 7, cnt, worst);
          }
        }
      }
    }
    for (int pass = 0; pass < 2; ++pass)
      for (int mi = 0; mi < 2; ++mi) {
        const int mode = pass == 0 ? mi : 1 - mi;
        g_ws_stream_conv = mode;
        for (int i = 0; i < 10; ++i) CK(launch_conv_gemm(p, 0));
        CK(hipDeviceSynchronize());
        const int iters = 10;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) CK(launch_conv_gemm(p, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters, tf = 2.0 * M * c.N * K / (us * 1e-6) / 1e12;
        printf("  WS_STREAM_CONV=%d: %8.1f us  %6.1f TF  (%.3f of 157.3)\n", mode, us, tf, tf / 157.3);
      }
  }
  return 0;
}
