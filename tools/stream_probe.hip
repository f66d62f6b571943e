// Stand-alone probe of the persistent fp32 GEMM (gemm_f32_stream.hip) against the tile kernels of
// conv_gemm.hip (not part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_probe.hip wespeaker_amd/csrc/conv_gemm.hip \
//         wespeaker_amd/csrc/gemm_f32_stream.hip wespeaker_amd/csrc/conv3x3_direct.hip -o tools/bin/stream_probe
// For every layer shape of the fp32 back-end's dominant class: whole-output bit compare (D, D2), column sums within
// rounding, three repeats per mode (race screen), then timing.  Modes (WS_STREAM): 0 tile kernels, 1 four wavefronts,
// 2 eight wavefronts, 3 the 256x128 tile with eight wavefronts, 4 the dispatcher's choice between 2 and 3.  -DWS_TRACE adds per-K-tile cycle stamps.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../wespeaker_amd/csrc/kernels.h"
using namespace wsamd;
namespace wsamd {
extern int g_ws_stream;
#ifdef WS_TRACE
unsigned long long* stream_trace_buffer_address();
#endif
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Case { int M, N, K, lda, a_off, d2, colsum; const char* name; };

int main(int argc, char** argv) {
  std::vector<Case> cases = {
      {50688, 512, 512, 512, 0, 0, 0, "512 plain"},
      {50688, 512, 512, 1536, 512, 1, 0, "512 blk0 (cat in, D2)"},
      {50688, 512, 512, 512, 0, 0, 1, "512 conv3 (colsum)"},
      {50688, 1536, 1536, 1536, 0, 0, 1, "cat 1536 (colsum)"},
      {50688, 1024, 1024, 1024, 0, 1, 0, "1024 blk0 (D2)"},
      {25344, 512, 512, 512, 0, 0, 1, "512 colsum M=128 utts"},
      {49152, 512, 512, 512, 0, 0, 0, "512 plain M=384t"},
  };
  const bool quick = argc > 1 && atoi(argv[1]) == 1;
  const size_t maxA = 50688ull * 1536, maxW = 1536ull * 1536, maxD = 50688ull * 1536;
  const size_t maxCS = (size_t)((50688 + 63) / 64 + 2) * 2 * 1536;
  float *A, *W, *D, *D2, *Z, *bias, *scale, *shift, *CS;
  CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&W, maxW * 4)); CK(hipMalloc(&D, maxD * 4));
  CK(hipMalloc(&D2, 50688ull * 1024 * 4)); CK(hipMalloc(&CS, maxCS * 4));
  CK(hipMalloc(&Z, 256)); CK(hipMemset(Z, 0, 256));
  CK(hipMalloc(&bias, 1536 * 4)); CK(hipMalloc(&scale, 1536 * 4)); CK(hipMalloc(&shift, 1536 * 4));
  std::vector<float> h(maxA);
  srand(7);
  for (size_t i = 0; i < maxA; ++i) h[i] = (float)((rand() % 2001) - 1000) / 1000.f;
  CK(hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data() + 12345, maxW * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, h.data() + 777, 1536 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(scale, h.data() + 4777, 1536 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(shift, h.data() + 9777, 1536 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int modes[] = {0, 2, 3, 4};
  for (auto& c : cases) {
    ConvGemmParams p; memset(&p, 0, sizeof(p));
    p.prec = 0;
    p.A = A; p.lda = c.lda; p.a_off = c.a_off; p.W = W; p.ldw = c.K; p.D = D; p.ldd = c.N;
    p.M = c.M; p.N = c.N; p.K = c.K; p.Cin = c.K;
    p.Hin = p.Hout = 1; p.Win = p.Wout = 198; p.stride_h = p.stride_w = 1; p.kh = p.kw = 1; p.dil_h = p.dil_w = 1;
    p.bias = bias; p.act = ACT_RELU; p.post_scale = scale; p.post_shift = shift; p.splitk = 1; p.zeros = Z;
    const int w8 = c.N / 8;
    if (c.d2) { p.D2 = D2; p.ldd2 = c.N; p.d2_off = 7 * w8; p.d2_col0 = 7 * w8; }
    if (c.colsum) p.colsum = CS;
    const size_t nD = (size_t)c.M * c.N, nCS = (size_t)((c.M + 63) / 64) * 2 * c.N;
    std::vector<float> ref(nD), got(nD), ref2, got2, refc, gotc;
    if (c.d2) { ref2.resize(nD); got2.resize(nD); }
    if (c.colsum) { refc.resize(nCS); gotc.resize(nCS); }
    printf("== %s: M=%d N=%d K=%d lda=%d\n", c.name, c.M, c.N, c.K, c.lda);
    for (int mode : modes) {
      g_ws_stream = mode;
      const int reps = mode == 0 ? 1 : (quick ? 1 : 3);
      for (int rep = 0; rep < reps; ++rep) {
        CK(hipMemset(D, 0xff, nD * 4));
        if (c.d2) CK(hipMemset(D2, 0xff, nD * 4));
        if (c.colsum) CK(hipMemset(CS, 0xff, nCS * 4));
        CK(launch_conv_gemm(p, 0)); CK(hipDeviceSynchronize());
        std::vector<float>& o = mode == 0 ? ref : got;
        CK(hipMemcpy(o.data(), D, nD * 4, hipMemcpyDeviceToHost));
        if (c.d2) CK(hipMemcpy((mode == 0 ? ref2 : got2).data(), D2, nD * 4, hipMemcpyDeviceToHost));
        if (c.colsum) CK(hipMemcpy((mode == 0 ? refc : gotc).data(), CS, nCS * 4, hipMemcpyDeviceToHost));
        if (mode == 0) continue;
        size_t nd = 0, nnan = 0, first = (size_t)-1;
        for (size_t i = 0; i < nD; ++i) {
          if (got[i] != got[i]) { ++nnan; if (first == (size_t)-1) first = i; continue; }
          if (got[i] != ref[i]) { ++nd; if (first == (size_t)-1) first = i; }
        }
        size_t nd2 = 0;
        if (c.d2)
          for (size_t m = 0; m < (size_t)c.M; ++m)
            for (int n = 0; n < w8; ++n) {           // only the pass-through columns are written
              const size_t i = m * c.N + 7 * w8 + n;
              if (!(got2[i] == ref2[i])) ++nd2;
            }
        double wc = 0;
        if (c.colsum)
          for (size_t i = 0; i < nCS; ++i) {
            const double e = fabs((double)refc[i] - gotc[i]) / (fabs((double)refc[i]) + 64.0);
            if (!(e <= wc)) wc = e;                  // NaN counts as worst
          }
        printf("  mode %d rep %d: D differs at %zu of %zu (nan %zu, first at m=%zu n=%zu)  D2 differs %zu  colsum worst rel %.3e\n",
               mode, rep, nd, nD, nnan, first == (size_t)-1 ? 0 : first / c.N, first == (size_t)-1 ? 0 : first % c.N, nd2, wc);
      }
    }
    // timing: the chip needs milliseconds of load to reach its steady clock, so every measurement is preceded by
    // 40 untimed launches of the same mode, and the modes are walked forwards and then backwards
    for (int pass = 0; pass < 2; ++pass)
    for (int mi = 0; mi < 4; ++mi) {
      const int mode = modes[pass == 0 ? mi : 3 - mi];
      g_ws_stream = mode;
      for (int i = 0; i < 40; ++i) CK(launch_conv_gemm(p, 0));
      CK(hipDeviceSynchronize());
      const int iters = 30;
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) CK(launch_conv_gemm(p, 0));
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters, tf = 2.0 * c.M * c.N * c.K / (us * 1e-6) / 1e12;
      printf("  mode %d: %8.1f us  %6.1f TF  (%.3f of 157.3)\n", mode, us, tf, tf / 157.3);
#ifdef WS_TRACE
      if (mode >= 1) {
        unsigned long long trb[512];
        CK(hipMemcpy(trb, stream_trace_buffer_address(), sizeof(trb), hipMemcpyDeviceToHost));
        for (int wv = 0; wv < (mode == 1 ? 1 : 2); ++wv) {
        const unsigned long long* tr = trb + 256 * wv;
        const int nk = c.K / 32, show = nk * 3 + 2 < 127 ? nk * 3 + 2 : 127;
        printf("    [wave %d, t0 %+lld] K-tile stamps (cycles; ideal 4096 per 64 MFMAs per SIMD): ", 4 * wv, (long long)(tr[0] - trb[0]));
        for (int i = 1; i < show; ++i) printf("%s%lld", (i - 1) % nk == 0 ? " | " : " ", (long long)(tr[i] - tr[i - 1]));
        printf("\n    regular K-tile: g0 %lld adv %lld g1 %lld g2 %lld wait+barrier %lld g3 %lld\n",
               (long long)(tr[129] - tr[128]), (long long)(tr[130] - tr[129]), (long long)(tr[131] - tr[130]),
               (long long)(tr[132] - tr[131]), (long long)(tr[133] - tr[132]), (long long)(tr[134] - tr[133]));
        printf("    first K-tile:   g0 %lld adv %lld g1 %lld g2 %lld wait+barrier %lld g3 %lld   (starts %+lld after wave 0's)\n",
               (long long)(tr[145] - tr[144]), (long long)(tr[146] - tr[145]), (long long)(tr[147] - tr[146]),
               (long long)(tr[148] - tr[147]), (long long)(tr[149] - tr[148]), (long long)(tr[150] - tr[149]),
               (long long)(tr[144] - trb[144]));
        printf("    of the wait+barrier: regular K-tile vmcnt/lgkmcnt wait %lld, first K-tile %lld\n",
               (long long)(tr[177] - tr[132]), (long long)(tr[178] - tr[148]));
        const int nb = mode == 2 ? 2 : 4;
        printf("    last K-tile:    block0 %lld wait+barrier %lld", (long long)(tr[161] - tr[160]), (long long)(tr[162] - tr[161]));
        for (int b = 1; b < nb; ++b)
          printf(" block%d %lld", b, (long long)(tr[b + 1 < nb ? 160 + 2 * (b + 1) : 168] - tr[160 + 2 * b]));
        printf(" tail %lld   (starts %+lld after wave 0's)\n", (long long)(tr[169] - tr[168]), (long long)(tr[160] - trb[160]));
        }
      }
#endif
    }
  }
  return 0;
}
