// The "M = B" layers of the fp32 back-end as ONE launch: a plain linear layer whose row count is the batch (256
// utterances) and whose K is the pooled statistics (3072 ... 5120): ECAPA's final BN + Linear (ecapa_tdnn.py:214-218)
// and the global-context half of the attention's first layer (pooling_layers.py:128-133), the ResNets' seg_1
// (resnet.py:196-202), CAM++'s dense layer (campplus.py:322-330).
//
// They ran as a split-K 128x128 tile GEMM + a reduce kernel: 22 + 7 us each on a problem of 0.3 GFLOP -- launch ramp,
// a tile kernel's prologue / epilogue and a second launch, twice per ECAPA forward (58 us of a 4.1-ms step).  Here:
// a workgroup owns a 16 x 16 output tile, its sixteen (four) wavefronts take a sixteenth (quarter) of K each (exact fp32
// v_mfma_f32_16x16x4_f32, operands straight from L2 into registers: one 16-B load per lane = the operands of four
// MFMAs, eight loads ahead), the partial tiles meet in LDS in a FIXED order (run-to-run bits) and wavefront 0
// finishes bias -> activation -> BN affine -> store.  16 x 12 = 192 workgroups for 256 x 192.
#include "kernels.h"

namespace wsamd {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SM_CH = 4;             // the smallest chunk of 16-k groups a wavefront's share of K is a multiple of

// NW wavefronts per 16 x 16 tile = parts of K; a wavefront walks its part in chunks of CH groups of 16 k
template <int NW, int CH>
__global__ __launch_bounds__(64 * NW) void small_m_gemm_f32_kernel(const ConvGemmParams p) {
  __shared__ f32x4 part[NW - 1][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  // lane (i, q) holds row i of the A tile and row i of the W tile at k = 16 g + 4 q + s: MFMA s of group g multiplies
  // A[i][16 g + 4 q + s] with W[j][16 g + 4 q + s] -- over the four lane groups q and the four s every k exactly once
  const int arow = m0 + li < p.M ? m0 + li : p.M - 1;           // (rows / columns past the edge: read the last one,
  const int wrow = n0 + li < p.N ? n0 + li : p.N - 1;           //  never stored)
  const int ks = p.K / NW;                                      // this wavefront's slice of K (a multiple of 16)
  const float* ap = p.A + (long long)arow * p.lda + p.a_off + wave * ks + lq * 4;
  const float* wp = p.W + (long long)wrow * p.ldw + wave * ks + lq * 4;
  const int groups = ks / 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // A chunk = CH groups: ALL of its 2 CH operand loads are issued, then its 4 CH MFMAs run as the loads land (in-order
  // counter, exact counts: the body is straight-line code).  K = 3072 over 16 wavefronts is ONE chunk of 12 groups --
  // the kernel is one L2 round trip long.  No register is carried from one chunk to the next and no branch sits inside
  // one (the first version guarded every group with `if (g < groups)` and prefetched through a
  // register ring; hipcc put `s_waitcnt vmcnt(0)` behind every join and sank the ring's loads: one L2 round trip per
  // group, 21 us).
  for (int g0 = 0; g0 < groups; g0 += CH) {
    f32x4 fa[CH], fw[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      fa[j] = *reinterpret_cast<const f32x4*>(ap + 16 * (g0 + j));
      fw[j] = *reinterpret_cast<const f32x4*>(wp + 16 * (g0 + j));
    }
    __builtin_amdgcn_sched_barrier(0);     // (hipcc otherwise sinks the loads between the MFMAs: four in flight)
    // C = A W^T: the MFMA's A operand is the activation row, its B operand the weight row (B[k][n] = W[n][k])
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j][s], fw[j][s], acc, 0, 0, 0);
  }
  if (wave > 0) part[wave - 1][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < NW - 1; ++w) acc += part[w][lane];         // (a fixed order: run-to-run bits)
  // C layout 16 x 16: lane -> column li (output channel n0 + li), rows 4 q .. 4 q + 3 (utterances m0 + 4 q + r)
  const int n = n0 + li;
  if (n >= p.N) return;
  const float b = p.bias ? p.bias[n] : 0.f;
  const float sc = p.post_scale ? p.post_scale[n] : 1.f, sh = p.post_scale ? p.post_shift[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + 4 * lq + r;
    if (m >= p.M) continue;
    float v = acc[r] + b;
    if (p.act == ACT_RELU) v = relu_f(v);
    else if (p.act == ACT_TANH) v = tanhf(v);
    if (p.post_scale) v = v * sc + sh;
    p.D[(long long)m * p.ldd + p.d_off + n] = v;
  }
}
}  // namespace

// a plain linear layer on fp32 rows with few of them and a long K, nothing fused but bias / activation / BN affine.
// Every back-end takes it: the binary16 back-ends ran these two 0.3-GFLOP layers as split-K GEMMs on converted operands
// (29 us each); the exact-fp32 form is both faster (13.6 us) and closer to the reference, and the fp32 weights are resident
// anyway.
bool small_m_gemm_f32_applies(const ConvGemmParams& p) {
  return p.W && p.A && p.D && !p.A16 && !p.A2 && !p.pre_scale && !p.D16 && !p.D2 && !p.D2_16 && p.kh == 1 &&
         p.kw == 1 && p.stride_h == 1 && p.stride_w == 1 && p.pad_h == 0 && p.pad_w == 0 && p.K == p.Cin &&
         p.m_begin == 0 && p.M >= 1 && p.M <= 2048 && p.N >= 1 && p.K >= 512 && p.K % (4 * 16 * SM_CH) == 0 &&
         p.ldw >= p.K &&
         !p.bias_img && !p.residual && !p.residual16 && !p.row_len && !p.seg_scale && !p.colsum && !p.pool_partial &&
         !p.pool_h && !p.pool_h16 && (p.lda & 3) == 0 && (p.a_off & 3) == 0 && (p.ldw & 3) == 0 &&
         (reinterpret_cast<unsigned long long>(p.A) & 15) == 0 && (reinterpret_cast<unsigned long long>(p.W) & 15) == 0;
}

hipError_t launch_small_m_gemm_f32(const ConvGemmParams& p, hipStream_t stream) {
  if (!small_m_gemm_f32_applies(p)) return hipErrorInvalidValue;
  // sixteen wavefronts per tile when K allows it, and the whole share of a wavefront as one chunk when that is
  // twelve groups (K = 3072: ECAPA's two layers, twice per forward)
  const dim3 grid((p.N + 15) / 16, (p.M + 15) / 16);
  if (p.K == 16 * 16 * 12) {
    if (dispatch_log_enabled()) dispatch_log_note(p, "small_m_gemm_f32_kernel<16x16 tile, K over 16 waves, one chunk>");
    hipLaunchKernelGGL((small_m_gemm_f32_kernel<16, 12>), grid, dim3(1024), 0, stream, p);
  } else if (p.K % (16 * 16 * SM_CH) == 0) {
    if (dispatch_log_enabled()) dispatch_log_note(p, "small_m_gemm_f32_kernel<16x16 tile, K over 16 waves>");
    hipLaunchKernelGGL((small_m_gemm_f32_kernel<16, SM_CH>), grid, dim3(1024), 0, stream, p);
  } else {
    if (dispatch_log_enabled()) dispatch_log_note(p, "small_m_gemm_f32_kernel<16x16 tile, K over 4 waves>");
    hipLaunchKernelGGL((small_m_gemm_f32_kernel<4, SM_CH>), grid, dim3(256), 0, stream, p);
  }
  return hipGetLastError();
}

}  // namespace wsamd
