// Stand-alone timing probe for the conv-GEMM kernel (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_probe.hip wespeaker_amd/csrc/conv_gemm.hip wespeaker_amd/csrc/conv3x3_direct.hip -o /tmp/gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../wespeaker_amd/csrc/kernels.h"
using namespace wsamd;
#ifdef WS_TRACE
namespace wsamd { unsigned long long* trace_buffer_address(); }
#endif
namespace wsamd { extern int g_ws_big_tiles; extern int g_ws_big_conv; extern int g_ws_epi16; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char** argv) {
  struct Shape { int M, N, K, taps; const char* name; };
  std::vector<Shape> shapes = {{50688, 512, 512, 1, "1x1 512"}, {50688, 1536, 1536, 1, "cat 1536"},
                               {50688, 128, 1536, 1, "astp1"}, {50688, 1536, 128, 1, "astp2"},
                               {50688, 512, 400, 5, "layer1"}, {50688, 64, 192, 3, "res2"}, {50688, 32, 96, 3, "n32 k3"},
                               {49152, 512, 512, 1, "512 M=384t"}, {32768, 512, 512, 1, "512 M=256t"},
                               {16384, 512, 512, 1, "512 M=128t"}, {49152, 1536, 1536, 1, "cat M=384t"}};
  if (getenv("PROBE_ONLY_CONV2D")) shapes.clear();
  const int prec = argc > 1 ? atoi(argv[1]) : 0;
  float *A, *W, *D, *Z, *bias;
  uint16_t *Wh, *Wl;
  size_t maxA = 50688ull * 1536, maxW = 1536ull * 1536, maxD = 50688ull * 1536;
  CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&W, maxW * 4)); CK(hipMalloc(&D, maxD * 4));
  CK(hipMalloc(&Wh, maxW * 2)); CK(hipMalloc(&Wl, maxW * 2));
  CK(hipMemset(Wh, 0x3c, maxW * 2)); CK(hipMemset(Wl, 0x1c, maxW * 2));
  CK(hipMalloc(&Z, 256)); CK(hipMemset(Z, 0, 256)); CK(hipMalloc(&bias, 1536 * 4));
  std::vector<float> h(maxA);
  for (size_t i = 0; i < maxA; ++i) h[i] = (float)((rand() % 2001) - 1000) / 1000.f;
  CK(hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), maxW * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, h.data(), 1536 * 4, hipMemcpyHostToDevice));
  // data mode (argv[2]): 0 random A, constant weight halves; 1 all zeros; 2 random A and random
  // weight halves (what a real model looks like).  DVFS makes the run time data dependent.
  const int data_mode = argc > 2 ? atoi(argv[2]) : 0;
  if (data_mode == 1) {
    CK(hipMemset(A, 0, maxA * 4)); CK(hipMemset(W, 0, maxW * 4));
    CK(hipMemset(Wh, 0, maxW * 2)); CK(hipMemset(Wl, 0, maxW * 2));
  } else if (data_mode == 2) {
    std::vector<uint16_t> hh(maxW), hl(maxW);
    for (size_t i = 0; i < maxW; ++i) {
      hh[i] = (uint16_t)(0x3000 + (rand() % 0x0c00) + ((rand() & 1) << 15));   // |x| in [0.125, 1)
      hl[i] = (uint16_t)(0x1000 + (rand() % 0x0c00) + ((rand() & 1) << 15));
    }
    CK(hipMemcpy(Wh, hh.data(), maxW * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(Wl, hl.data(), maxW * 2, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& s : shapes) {
    ConvGemmParams p; memset(&p, 0, sizeof(p));
    int Cin = s.K / s.taps;
    p.prec = prec; p.Wh = Wh; p.Wl = Wl;
    p.A = A; p.lda = Cin; p.W = W; p.ldw = (s.K + 31) / 32 * 32; p.D = D; p.ldd = s.N;
    p.M = s.M; p.N = s.N; p.K = s.K; p.Cin = Cin;
    p.Hin = p.Hout = 1; p.Win = p.Wout = 198; p.stride_h = p.stride_w = 1; p.kh = 1; p.kw = s.taps;
    p.dil_h = 1; p.dil_w = s.taps == 3 ? 2 : 1; p.pad_w = p.dil_w * (s.taps / 2);
    p.bias = bias; p.act = ACT_RELU; p.post_scale = bias; p.post_shift = bias; p.splitk = 1; p.zeros = Z;
    static uint16_t* A16 = nullptr;
    if (getenv("PROBE_A16") && prec == 2) {
      if (!A16) {
        CK(hipMalloc(&A16, maxA * 2));
        std::vector<uint16_t> h16(maxA);
        for (size_t i = 0; i < maxA; ++i) { _Float16 v = (_Float16)h[i]; memcpy(&h16[i], &v, 2); }
        CK(hipMemcpy(A16, h16.data(), maxA * 2, hipMemcpyHostToDevice));
      }
      p.A16 = A16; p.lda16 = Cin;
    }
    double conv_diff = -1.0;
    if (p.A16 && s.taps > 1) {
      // convolution on binary16 activations (LDS-DMA kernel) against the fp32-activation kernel
      ConvGemmParams q = p; q.A16 = nullptr;
      std::vector<float> ref((size_t)s.M * s.N), got((size_t)s.M * s.N);
      CK(launch_conv_gemm(q, 0)); CK(hipDeviceSynchronize());
      CK(hipMemcpy(ref.data(), D, ref.size() * 4, hipMemcpyDeviceToHost));
      CK(launch_conv_gemm(p, 0)); CK(hipDeviceSynchronize());
      CK(hipMemcpy(got.data(), D, got.size() * 4, hipMemcpyDeviceToHost));
      conv_diff = 0.0;
      for (size_t i = 0; i < ref.size(); ++i) {
        const double e = fabs((double)ref[i] - got[i]) / (fabs((double)ref[i]) + 1.0);
        if (e > conv_diff) conv_diff = e;
      }
    }
    for (int i = 0; i < 3; ++i) CK(launch_conv_gemm(p, 0));
    CK(hipDeviceSynchronize());
    const int iters = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) CK(launch_conv_gemm(p, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters, tf = 2.0 * s.M * s.N * s.K / (us * 1e-6) / 1e12;
    // spot check of 64 outputs against a host evaluation (1x1 shapes only; bias/BN epilogue included)
    double max_err = -1.0;
    if (s.taps == 1 && argc > 3) {
      std::vector<float> hd((size_t)s.M * s.N);
      CK(hipMemcpy(hd.data(), D, hd.size() * 4, hipMemcpyDeviceToHost));
      std::vector<uint16_t> wh((size_t)s.N * p.ldw), wl((size_t)s.N * p.ldw);
      CK(hipMemcpy(wh.data(), Wh, wh.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(wl.data(), Wl, wl.size() * 2, hipMemcpyDeviceToHost));
      auto h2f = [](uint16_t b) { _Float16 v; memcpy(&v, &b, 2); return (double)(float)v; };
      max_err = 0.0;
      for (int t = 0; t < 64; ++t) {
        const int m = (int)(((long long)t * 7919 + 13) % s.M), n = (t * 131 + 7) % s.N;
        double acc = 0.0;
        for (int k = 0; k < s.K; ++k) {
          const float a = h[(size_t)m * Cin + k];
          double av = a, wv;
          if (prec == 2) { av = (double)(float)(_Float16)a; wv = h2f(wh[(size_t)n * p.ldw + k]); }
          else if (prec == 1) wv = h2f(wh[(size_t)n * p.ldw + k]) + h2f(wl[(size_t)n * p.ldw + k]);
          else wv = h[(size_t)n * p.ldw + k];
          acc += av * wv;
        }
        double v = acc + h[n];
        v = v > 0 ? v : 0;
        v = v * h[n] + h[n];
        const double e = fabs(v - hd[(size_t)m * s.N + n]) / (fabs(v) + 1.0);
        if (e > max_err) max_err = e;
      }
    }
#ifdef WS_TRACE
    if (s.K == 1536 && s.N == 1536) {
      unsigned long long tr[64 * 8];
      CK(hipMemcpy(tr, trace_buffer_address(), sizeof(tr), hipMemcpyDeviceToHost));
      printf("  marks: prologue %lld  loop %lld  epilogue %lld\n", (long long)(tr[63 * 8 + 1] - tr[63 * 8]),
             (long long)(tr[63 * 8 + 2] - tr[63 * 8 + 1]), (long long)(tr[63 * 8 + 3] - tr[63 * 8 + 2]));
      printf("  clock: %lld shader ticks in %lld realtime ticks (100 MHz) -> %.0f MHz\n",
             (long long)(tr[63 * 8 + 3] - tr[63 * 8]), (long long)(tr[63 * 8 + 5] - tr[63 * 8 + 4]),
             100.0 * (double)(tr[63 * 8 + 3] - tr[63 * 8]) / (double)(tr[63 * 8 + 5] - tr[63 * 8 + 4]));
      for (int k = 0; k < (getenv("PROBE_ITERS") ? atoi(getenv("PROBE_ITERS")) : 12); ++k) {
        if (getenv("PROBE_P8")) {
          const unsigned long long* t = tr + (32 + k) * 8;
          printf("  kt %2d: P1 mem+bar %5lld mma+bar %5lld | P2 %5lld %5lld | tile %5lld\n", k,
                 (long long)(t[1] - t[0]), (long long)(t[2] - t[1]), (long long)(t[3] - t[2]), (long long)(t[4] - t[3]),
                 (long long)(t[4] - t[0]));
          continue;
        }
        printf("  it %2d:", k);
        for (int j = 1; j < (getenv("PROBE_STAMPS") ? atoi(getenv("PROBE_STAMPS")) : getenv("PROBE_A16") ? 4 : 6); ++j)
          printf(" %6lld", (long long)(tr[k * 8 + j] - tr[k * 8 + j - 1]));
        printf("  | loop %6lld\n", (long long)(tr[(k + 1) * 8] - tr[k * 8]));
      }

    }
#endif
    if (getenv("PROBE_XCMP") && s.taps == 1 && s.N >= 1024) {
      // the same launch through the 128x128 kernel and through tile mode PROBE_XCMP, whole outputs compared
      std::vector<float> r0((size_t)s.M * s.N), r1((size_t)s.M * s.N);
      const int keep = g_ws_big_tiles;
      for (int rep = 0; rep < 3; ++rep) {
        g_ws_big_tiles = 0;
        CK(hipMemset(D, 0xff, r0.size() * 4));
        CK(launch_conv_gemm(p, 0)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(r0.data(), D, r0.size() * 4, hipMemcpyDeviceToHost));
        g_ws_big_tiles = atoi(getenv("PROBE_XCMP"));
        CK(hipMemset(D, 0xff, r0.size() * 4));
        CK(launch_conv_gemm(p, 0)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(r1.data(), D, r1.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0.0; size_t ndiff = 0, nnan = 0;
        for (size_t i = 0; i < r0.size(); ++i) {
          if (r1[i] != r1[i]) { ++nnan; continue; }
          if (r0[i] != r1[i]) { ++ndiff; const double e = fabs((double)r0[i] - r1[i]) / (fabs((double)r0[i]) + 1.0); if (e > worst) worst = e; }
        }
        printf("  xcmp[%d] %s: %zu of %zu differ, worst rel %.3e, nan %zu\n", rep, s.name, ndiff, r0.size(), worst, nnan);
      }
      g_ws_big_tiles = keep;
    }
    if (getenv("PROBE_D16") && s.taps == 1 && s.N % 256 == 0 && s.N >= 512 && p.A16) {
      // D16-only layer with column sums: fp32 two-phase epilogue vs binary16 one-phase epilogue of the 256x256 kernel
      static uint16_t* D16b = nullptr; static float* CS = nullptr;
      if (!D16b) { CK(hipMalloc(&D16b, maxD * 2)); CK(hipMalloc(&CS, ((50688 + 63) / 64 + 2) * 2 * 1536 * 4)); }
      ConvGemmParams q = p; q.D = nullptr; q.D16 = D16b; q.ldd16 = s.N; q.colsum = CS;
      const size_t nout = (size_t)s.M * s.N, ncs = (size_t)((s.M + 63) / 64) * 2 * s.N;
      std::vector<uint16_t> o0(nout), o1(nout); std::vector<float> c0(ncs), c1(ncs);
      const int keepb = g_ws_big_tiles; g_ws_big_tiles = 3;
      for (int rep = 0; rep < 2; ++rep) {
        g_ws_epi16 = 0;
        CK(hipMemset(D16b, 0xff, nout * 2)); CK(hipMemset(CS, 0, ncs * 4));
        CK(launch_conv_gemm(q, 0)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(o0.data(), D16b, nout * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(c0.data(), CS, ncs * 4, hipMemcpyDeviceToHost));
        g_ws_epi16 = 1;
        CK(hipMemset(D16b, 0xff, nout * 2)); CK(hipMemset(CS, 0, ncs * 4));
        CK(launch_conv_gemm(q, 0)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(o1.data(), D16b, nout * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), CS, ncs * 4, hipMemcpyDeviceToHost));
        size_t nd = 0; for (size_t i = 0; i < nout; ++i) nd += o0[i] != o1[i];
        double wc = 0; for (size_t i = 0; i < ncs; ++i) { const double e = fabs((double)c0[i] - c1[i]) / (fabs((double)c0[i]) + 64.0); if (e > wc) wc = e; }
        printf("  d16[%d] %s: %zu of %zu halfs differ; colsum worst rel %.3e\n", rep, s.name, nd, nout, wc);
      }
      for (int e16 = 0; e16 < 2; ++e16) {
        g_ws_epi16 = e16;
        for (int i = 0; i < 3; ++i) CK(launch_conv_gemm(q, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) CK(launch_conv_gemm(q, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
        printf("  d16 %s epi16=%d: %8.1f us\n", s.name, e16, ms2 * 50.0);
#ifdef WS_TRACE
        {
          unsigned long long tr2[64 * 8];
          CK(hipMemcpy(tr2, trace_buffer_address(), sizeof(tr2), hipMemcpyDeviceToHost));
          printf("    marks (epi16=%d): prologue %lld  loop %lld  epilogue %lld\n", e16, (long long)(tr2[63 * 8 + 1] - tr2[63 * 8]),
                 (long long)(tr2[63 * 8 + 2] - tr2[63 * 8 + 1]), (long long)(tr2[63 * 8 + 3] - tr2[63 * 8 + 2]));
        }
#endif
      }
      g_ws_epi16 = 0; g_ws_big_tiles = keepb;
    }
    if (getenv("PROBE_HASH") && s.taps == 1) {
      // full-output hash over repeated launches: equal across tile shapes (same k order) and across runs
      std::vector<float> hd((size_t)s.M * s.N);
      for (int rep = 0; rep < atoi(getenv("PROBE_HASH")); ++rep) {
        CK(hipMemset(D, 0xff, hd.size() * 4));
        CK(launch_conv_gemm(p, 0)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(hd.data(), D, hd.size() * 4, hipMemcpyDeviceToHost));
        unsigned long long hsh = 1469598103934665603ull;
        const uint32_t* u = reinterpret_cast<const uint32_t*>(hd.data());
        for (size_t i = 0; i < hd.size(); ++i) { hsh ^= u[i]; hsh *= 1099511628211ull; }
        printf("  hash[%d] %s %016llx\n", rep, s.name, hsh);
      }
    }
    printf("%-10s M=%d N=%d K=%d taps=%d : %8.1f us  %6.1f TF  spot-err %.2e  conv-vs-f32A %.2e\n", s.name,
           s.M, s.N, s.K, s.taps, us, tf, max_err, conv_diff);
  }
  if (getenv("PROBE_CONV2D") && prec == 2) {
    // 2-D 3x3 convolutions on binary16 maps (ResNet stage 4): the phase-staggered 256x256 kernel against the
    // 128x128 convolution kernel, whole outputs, with and without a binary16 residual; then timing of both
    struct C2 { int images, Hin, Win, Cin, N, stride; const char* name; };
    std::vector<C2> cases = {{512, 10, 25, 256, 256, 1, "s4 3x3 256->256"}, {512, 20, 50, 128, 256, 2, "s4 3x3/2 128->256"},
                             {37, 10, 25, 256, 256, 1, "ragged 37 images"}};
    uint16_t* A16c; uint16_t* R16; float* D2;
    const size_t maxIn = 512ull * 20 * 50 * 128 > 512ull * 10 * 25 * 256 ? 512ull * 20 * 50 * 128 : 512ull * 10 * 25 * 256;
    const size_t maxOut = 512ull * 10 * 25 * 256;
    CK(hipMalloc(&A16c, maxIn * 2)); CK(hipMalloc(&R16, maxOut * 2)); CK(hipMalloc(&D2, maxOut * 4));
    {
      std::vector<uint16_t> hin(maxIn), hr(maxOut);
      for (size_t i = 0; i < maxIn; ++i) { _Float16 v = (_Float16)h[i % h.size()]; memcpy(&hin[i], &v, 2); }
      for (size_t i = 0; i < maxOut; ++i) { _Float16 v = (_Float16)h[(i * 7 + 3) % h.size()]; memcpy(&hr[i], &v, 2); }
      CK(hipMemcpy(A16c, hin.data(), maxIn * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(R16, hr.data(), maxOut * 2, hipMemcpyHostToDevice));
    }
    for (auto& c : cases) {
      ConvGemmParams p; memset(&p, 0, sizeof(p));
      const int Hout = (c.Hin + 2 - 3) / c.stride + 1, Wout = (c.Win + 2 - 3) / c.stride + 1;
      p.prec = 2; p.Wh = Wh; p.Wl = Wl; p.W = W; p.A = A;
      p.A16 = A16c; p.lda16 = c.Cin; p.lda = c.Cin;
      p.M = c.images * Hout * Wout; p.N = c.N; p.K = 9 * c.Cin; p.Cin = c.Cin; p.ldw = p.K;
      p.Hin = c.Hin; p.Win = c.Win; p.Hout = Hout; p.Wout = Wout; p.stride_h = p.stride_w = c.stride;
      p.kh = p.kw = 3; p.dil_h = p.dil_w = 1; p.pad_h = p.pad_w = 1;
      p.bias = bias; p.act = ACT_RELU; p.splitk = 1; p.zeros = Z;
      p.D = D2; p.ldd = c.N;
      const size_t nout = (size_t)p.M * p.N;
      std::vector<float> r0(nout), r1(nout);
      for (int res = 0; res < 2; ++res) {
        p.residual16 = res ? R16 : nullptr; p.ldr = c.N; p.r_off = 0;
        g_ws_big_conv = 0;
        CK(hipMemset(D2, 0xff, nout * 4)); CK(launch_conv_gemm(p, 0)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(r0.data(), D2, nout * 4, hipMemcpyDeviceToHost));
        g_ws_big_conv = 1;
        CK(hipMemset(D2, 0xff, nout * 4)); CK(launch_conv_gemm(p, 0)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(r1.data(), D2, nout * 4, hipMemcpyDeviceToHost));
        size_t ndiff = 0, nnan = 0; double worst = 0;
        for (size_t i = 0; i < nout; ++i) {
          if (r1[i] != r1[i]) { ++nnan; continue; }
          if (r0[i] != r1[i]) { ++ndiff; const double e = fabs((double)r0[i] - r1[i]) / (fabs((double)r0[i]) + 1.0); if (e > worst) worst = e; }
        }
        printf("  conv2d %-20s residual16=%d: %zu of %zu differ, worst rel %.3e, nan %zu\n", c.name, res, ndiff, nout, worst, nnan);
      }
      for (int big = 0; big < 2; ++big) {
        g_ws_big_conv = big;
        for (int i = 0; i < 3; ++i) CK(launch_conv_gemm(p, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) CK(launch_conv_gemm(p, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  conv2d %-20s big=%d: %8.1f us  %6.1f TF\n", c.name, big, ms * 50.0, 2.0 * p.M * p.N * p.K / (ms * 50e-6) / 1e12);
      }
    }
  }
  return 0;
}
