// Reduction / element-wise kernels of the ECAPA-TDNN forward on channels-last activations
// (rows = (utterance, frame), contiguous channels).  All HBM-bound: 16-B loads, lanes run along
// the contiguous channel axis, reductions over time are per-lane serial sums (unrolled for memory
// level parallelism) finished through LDS / wavefront shuffles.
//
// Reference semantics:
//   SE_Connect           wespeaker/models/ecapa_tdnn.py:120-126
//   SE_Res2Block add     wespeaker/models/ecapa_tdnn.py:156-157
//   ASTP                 wespeaker/models/pooling_layers.py:119-144
#include "kernels.h"

namespace wsamd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------- SE pooling + FCs
// grid = B, block = 256.  C <= 1024 (C/4 float4 columns <= 256), bottleneck <= 256.
__global__ __launch_bounds__(256) void se_pool_fc_kernel(const float* __restrict__ y, int ldy, int T,
                                                         int C, const float* __restrict__ w1,
                                                         const float* __restrict__ b1,
                                                         const float* __restrict__ w2,
                                                         const float* __restrict__ b2, int bott,
                                                         float* __restrict__ s,
                                                         const int* __restrict__ lens) {
  __shared__ __attribute__((aligned(16))) float sm[1024 * 4 + 256];
  float* part = sm;             // [groups][C]
  float* hidden = sm + 1024 * 4;  // [bott]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int cols4 = C >> 2;                 // float4 columns
  const int groups = 256 / cols4;           // time groups (>=1)
  const int col = tid % cols4, grp = tid / cols4;
  const float* base = y + (long long)b * T * ldy + col * 4;
  if (lens) T = lens[b];                    // ragged batch: mean over the utterance's own frames
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
  if (grp < groups) {
    int t = grp;
    const int step = groups;
    for (; t + 3 * step < T; t += 4 * step) {
      f32x4 v0 = *reinterpret_cast<const f32x4*>(base + (long long)t * ldy);
      f32x4 v1 = *reinterpret_cast<const f32x4*>(base + (long long)(t + step) * ldy);
      f32x4 v2 = *reinterpret_cast<const f32x4*>(base + (long long)(t + 2 * step) * ldy);
      f32x4 v3 = *reinterpret_cast<const f32x4*>(base + (long long)(t + 3 * step) * ldy);
      acc0 += v0; acc1 += v1; acc2 += v2; acc3 += v3;
    }
    for (; t < T; t += step) acc0 += *reinterpret_cast<const f32x4*>(base + (long long)t * ldy);
    f32x4 a = (acc0 + acc1) + (acc2 + acc3);
    *reinterpret_cast<f32x4*>(&part[grp * C + col * 4]) = a;
  }
  __syncthreads();
  // mean[c]
  for (int c = tid; c < C; c += 256) {
    float v = 0.f;
    for (int g = 0; g < groups; ++g) v += part[g * C + c];
    part[c] = v / (float)T;     // group 0 slot reused: safe, each thread touches only column c
  }
  __syncthreads();
  // hidden = relu(W1 mean + b1): one wavefront per output row
  const int lane = tid & 63, wave = tid >> 6;
  for (int j = wave; j < bott; j += 4) {
    const float* wr = w1 + (long long)j * C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) v += wr[c] * part[c];
    v = wave_sum(v);
    if (lane == 0) hidden[j] = relu_f(v + b1[j]);
  }
  __syncthreads();
  // s = sigmoid(W2 hidden + b2): thread per output channel (rows of W2 are short: bott floats)
  for (int c = tid; c < C; c += 256) {
    const float* wr = w2 + (long long)c * bott;
    float v = 0.f;
    for (int j = 0; j < bott; j += 4) {
      f32x4 w = *reinterpret_cast<const f32x4*>(wr + j);
      v += w[0] * hidden[j] + w[1] * hidden[j + 1] + w[2] * hidden[j + 2] + w[3] * hidden[j + 3];
    }
    v += b2[c];
    s[(long long)b * C + c] = 1.f / (1.f + expf(-v));
  }
}

hipError_t launch_se_pool_fc(const float* y, int ldy, int B, int T, int C, const float* w1,
                             const float* b1, const float* w2, const float* b2, int bottleneck,
                             float* s, hipStream_t stream, const int* lens) {
  if (C > 1024 || (C & 3) || bottleneck > 256 || (bottleneck & 3) || 256 % (C >> 2) != 0)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(se_pool_fc_kernel, dim3(B), dim3(256), 0, stream, y, ldy, T, C, w1, b1, w2,
                     b2, bottleneck, s, lens);
  return hipGetLastError();
}

// ------------------------------------------------------ SE FCs from GEMM-epilogue column sums
// C <= 1024, bottleneck <= 256.  Both FCs keep many independent 16-B weight loads in flight per lane (the dependent
// one-row-at-a-time form was L2-latency bound).  One body for the stand-alone kernel (512 threads) and the fused
// FC + scale + residual kernel (1024 threads): an output's sum does not depend on the thread count (FC1: one
// wavefront per row, the lanes split the C columns the same way; FC2: one thread per channel), so both give the same
// bits.  `w2` is the TRANSPOSED second matrix, [bott][C].  Leaves s[0..C) in s_lds too when S_TO_LDS.
template <int NT, bool S_TO_LDS>
__device__ __forceinline__ void se_fc_body(const float* __restrict__ colsum, int b, int T, int C,
                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                           const float* __restrict__ w2, const float* __restrict__ b2, int bott,
                                           float* __restrict__ s_out, const int* __restrict__ lens, float* mean,
                                           float* hidden, float* s_lds) {
  const int tid = threadIdx.x;
  // ragged batch: the rows beyond lens[b] were stored as zeros, so the tile sums are right as they are and
  // only the divisor is the utterance's own length
  const float inv_len = 1.f / (float)(lens ? lens[b] : T);
  const long long r0 = (long long)b * T, r1 = r0 + T - 1;
  const int t_first = (int)(r0 / 64), t_last = (int)(r1 / 64);
  // the tile partials of an utterance (<= 8 for T <= 448, <= 16 for T <= 960; more fall back to the serial loop):
  // all requested at once, in tile order -- a serial loop exposed one L2 round trip per tile
  for (int c = tid; c < C; c += NT) {
    float v = 0.f;
    if (t_last - t_first < 8) {
      float part[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int tm = t_first + i <= t_last ? t_first + i : t_last;
        const int first_img = (int)(((long long)tm * 64) / T);
        const int which = (first_img == b) ? 0 : 1;
        part[i] = colsum[((long long)tm * 2 + which) * C + c];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) v += t_first + i <= t_last ? part[i] : 0.f;
    } else if (t_last - t_first < 16) {
      float part[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int tm = t_first + i <= t_last ? t_first + i : t_last;
        const int first_img = (int)(((long long)tm * 64) / T);
        const int which = (first_img == b) ? 0 : 1;
        part[i] = colsum[((long long)tm * 2 + which) * C + c];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) v += t_first + i <= t_last ? part[i] : 0.f;
    } else {
      for (int tm = t_first; tm <= t_last; ++tm) {
        const int first_img = (int)(((long long)tm * 64) / T);
        const int which = (first_img == b) ? 0 : 1;
        v += colsum[((long long)tm * 2 + which) * C + c];
      }
    }
    mean[c] = v * inv_len;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  // FC1: hidden = relu(W1 mean + b1); 8 rows per wavefront pass (8 x C/256 independent 16-B loads in flight:
  // the kernel is a chain of L2 round trips, so fewer, wider passes -- 2 instead of 4 for bott = 128 with 8 wavefronts)
  for (int j0 = wave * 8; j0 < bott; j0 += NT / 8) {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // (bott % 32 == 0 on the host: a wave's 8 rows are all inside or all outside)
    const f32x4 bias1a = *reinterpret_cast<const f32x4*>(b1 + j0);   // (not behind the lane-0 branch below)
    const f32x4 bias1b = *reinterpret_cast<const f32x4*>(b1 + j0 + 4);
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 m = *reinterpret_cast<const f32x4*>(&mean[c]);
      f32x4 w[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) w[q] = *reinterpret_cast<const f32x4*>(w1 + (long long)(j0 + q) * C + c);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += w[q][0] * m[0] + w[q][1] * m[1] + w[q][2] * m[2] + w[q][3] * m[3];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float t = wave_sum(v[q]);
      if (lane == 0) hidden[j0 + q] = relu_f(t + (q < 4 ? bias1a[q & 3] : bias1b[q & 3]));
    }
  }
  __syncthreads();
  // FC2: thread per output channel.  The weights are read from the TRANSPOSED copy w2t[k][c] (round 5): a wavefront's
  // load is then 256 contiguous bytes; with the row-major matrix every lane read 16 B of its own 512-B row -- 64 cache
  // lines per instruction, eight times the bytes through the L1.  32 loads in flight, the sums term for term as before.
  for (int c = tid; c < C; c += NT) {
    const float* wc = w2 + c;
    const float bias2 = b2[c];
    float v = 0.f;
    for (int k = 0; k < bott; k += 32) {
      float w[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) w[q] = wc[(long long)(k + q) * C];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 h = *reinterpret_cast<const f32x4*>(&hidden[k + q * 4]);
        v += w[4 * q] * h[0] + w[4 * q + 1] * h[1] + w[4 * q + 2] * h[2] + w[4 * q + 3] * h[3];
      }
    }
    const float sg = 1.f / (1.f + expf(-(v + bias2)));
    s_out[(long long)b * C + c] = sg;
    if (S_TO_LDS) s_lds[c] = sg;
  }
}

// grid = B, block = 512
__global__ __launch_bounds__(512) void se_fc_from_colsum_kernel(
    const float* __restrict__ colsum, int T, int C, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    int bott, float* __restrict__ s, const int* __restrict__ lens) {
  __shared__ __attribute__((aligned(16))) float mean[1024];
  __shared__ __attribute__((aligned(16))) float hidden[256];
  se_fc_body<512, false>(colsum, blockIdx.x, T, C, w1, b1, w2, b2, bott, s, lens, mean, hidden, nullptr);
}

hipError_t launch_se_fc_from_colsum(const float* colsum, int B, int T, int C, const float* w1,
                                    const float* b1, const float* w2, const float* b2,
                                    int bottleneck, float* s, hipStream_t stream, const int* lens) {
  if (C > 1024 || (C & 255) || bottleneck > 256 || (bottleneck & 31) || T < 64)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(se_fc_from_colsum_kernel, dim3(B), dim3(512), 0, stream, colsum, T, C, w1, b1,
                     w2, b2, bottleneck, s, lens);
  return hipGetLastError();
}

// --------------------------------------------------------------------- SE scale + block residual
__global__ __launch_bounds__(256) void se_scale_residual_kernel(
    const float* __restrict__ x, int ldx, int x_off, const float* __restrict__ y, int ldy,
    const float* __restrict__ s, float* __restrict__ out, int ldo, int o_off, int T, int C,
    long long total4, uint16_t* __restrict__ out16) {
  const int cols4 = C >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4;
       i += (long long)gridDim.x * 256) {
    const long long m = i / cols4;
    const int c = (int)(i - m * cols4) * 4;
    const int b = (int)(m / T);
    f32x4 xv = *reinterpret_cast<const f32x4*>(x + m * ldx + x_off + c);
    f32x4 yv = *reinterpret_cast<const f32x4*>(y + m * ldy + c);
    f32x4 sv = *reinterpret_cast<const f32x4*>(s + (long long)b * C + c);
    const f32x4 v = xv + yv * sv;
    *reinterpret_cast<f32x4*>(out + m * ldo + o_off + c) = v;
    if (out16) {                                   // binary16 copy for the f16 GEMM back-end
      typedef _Float16 f16x4e __attribute__((ext_vector_type(4)));
      f16x4e hv;
#pragma unroll
      for (int q = 0; q < 4; ++q) hv[q] = (_Float16)v[q];
      *reinterpret_cast<f16x4e*>(out16 + m * ldo + o_off + c) = hv;
    }
  }
}

hipError_t launch_se_scale_residual(const float* x, int ldx, int x_off, const float* y, int ldy,
                                    const float* s, float* out, int ldo, int o_off, int B, int T,
                                    int C, hipStream_t stream, uint16_t* out16) {
  const long long total4 = (long long)B * T * (C >> 2);
  long long blocks = (total4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(se_scale_residual_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, ldx,
                     x_off, y, ldy, s, out, ldo, o_off, T, C, total4, out16);
  return hipGetLastError();
}

// ------------------------------------------------- SE FCs + scale + block residual in ONE launch (round 5)
// grid = B, block = 1024: the workgroup of utterance b first requests the first rows of x and y (their latency runs
// under the FCs), computes s[b][:] from the column sums exactly like se_fc_from_colsum_kernel (se_fc_body: same bits),
// keeps it in LDS and streams the utterance's T rows: out = x + y * s (the expression of se_scale_residual_kernel:
// same bits).  A thread owns 4 fixed channels -- its s values are registers -- and every (1024 / (C/4))-th row;
// U row slots of x / y are in flight per thread (16 16-B loads: 256 KB per CU, what one workgroup per CU needs to
// keep HBM busy).  Replaces a 19-us latency-bound launch + a launch boundary per SE block.
template <bool OUT16>
__global__ __launch_bounds__(1024) void se_fc_scale_residual_kernel(
    const float* __restrict__ colsum, int T, int C, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, int bott, float* __restrict__ s_out,
    const int* __restrict__ lens, const float* __restrict__ x, int ldx, int x_off, const float* __restrict__ y,
    int ldy, float* __restrict__ out, int ldo, int o_off, uint16_t* __restrict__ out16, int nsplit) {
  __shared__ __attribute__((aligned(16))) float mean[1024];
  __shared__ __attribute__((aligned(16))) float hidden[256];
  __shared__ __attribute__((aligned(16))) float s_lds[1024];
  constexpr int U = 8;
  // nsplit (round 6): a batch of few, long utterances leaves most CUs without a workgroup -- nsplit workgroups per
  // utterance then each compute s (the same bits) and stream their share [t0, t1) of its rows
  const int b = blockIdx.x / nsplit, part = blockIdx.x - b * nsplit, tid = threadIdx.x;
  const int chunk = (T + nsplit - 1) / nsplit, t0 = part * chunk, t1 = t0 + chunk < T ? t0 + chunk : T;
  const int cols4 = C >> 2;                  // 128 / 256 (C = 512 / 1024): divides the 1024 threads
  const int c = (tid % cols4) * 4, r0 = tid / cols4, rp = 1024 / cols4;
  const long long m0 = (long long)b * T;
  const float* xp = x + m0 * ldx + x_off + c;
  const float* yp = y + m0 * ldy + c;
  constexpr int PRE = 4;                     // slots requested in front of the FCs (more would spill there)
  f32x4 xv[U], yv[U];
  auto request = [&](int u) {
    const int r = t0 + r0 + u * rp;
    if (r < t1) {
      xv[u] = *reinterpret_cast<const f32x4*>(xp + (long long)r * ldx);
      yv[u] = *reinterpret_cast<const f32x4*>(yp + (long long)r * ldy);
    }
  };
#pragma unroll
  for (int u = 0; u < PRE; ++u) request(u);
  se_fc_body<1024, true>(colsum, b, T, C, w1, b1, w2, b2, bott, s_out, lens, mean, hidden, s_lds);
#pragma unroll
  for (int u = PRE; u < U; ++u) request(u);
  __syncthreads();
  const f32x4 sv = *reinterpret_cast<const f32x4*>(&s_lds[c]);
  float* op = out + m0 * ldo + o_off + c;
  uint16_t* op16 = OUT16 ? out16 + m0 * ldo + o_off + c : nullptr;
  for (int rb = t0 + r0; rb < t1; rb += U * rp) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = rb + u * rp;
      if (r < t1) {
        const f32x4 v = xv[u] + yv[u] * sv;
        *reinterpret_cast<f32x4*>(op + (long long)r * ldo) = v;
        if (OUT16) {                                 // binary16 copy for the f16 GEMM back-end
          typedef _Float16 f16x4e __attribute__((ext_vector_type(4)));
          f16x4e hv;
#pragma unroll
          for (int q = 0; q < 4; ++q) hv[q] = (_Float16)v[q];
          *reinterpret_cast<f16x4e*>(op16 + (long long)r * ldo) = hv;
        }
        const int rn = r + U * rp;                   // this slot's next row
        if (rn < t1) {
          xv[u] = *reinterpret_cast<const f32x4*>(xp + (long long)rn * ldx);
          yv[u] = *reinterpret_cast<const f32x4*>(yp + (long long)rn * ldy);
        }
      }
    }
  }
}

// workgroups per utterance of the fused SE kernels: one when the batch covers the chip, else up to eight parts of
// >= 64 rows each
static int se_row_split(int B, int T) {
  const int cus = current_device_cus();
  int n = B >= cus ? 1 : cus / B;
  if (n > T / 64) n = T / 64;
  return n < 1 ? 1 : (n > 8 ? 8 : n);
}

bool se_fc_scale_residual_supported(int T, int C, int bottleneck) {
  return C <= 1024 && (C & 255) == 0 && bottleneck <= 256 && (bottleneck & 31) == 0 && T >= 64 &&
         1024 % (C >> 2) == 0;
}

hipError_t launch_se_fc_scale_residual(const float* colsum, int B, int T, int C, const float* w1, const float* b1,
                                       const float* w2, const float* b2, int bottleneck, float* s,
                                       const int* lens, const float* x, int ldx, int x_off, const float* y, int ldy,
                                       float* out, int ldo, int o_off, hipStream_t stream, uint16_t* out16) {
  if (!se_fc_scale_residual_supported(T, C, bottleneck) || ((ldx | x_off | ldy | ldo | o_off) & 3))
    return hipErrorInvalidValue;
  const int nsplit = se_row_split(B, T);
  if (out16)
    hipLaunchKernelGGL(se_fc_scale_residual_kernel<true>, dim3(B * nsplit), dim3(1024), 0, stream, colsum, T, C, w1,
                       b1, w2, b2, bottleneck, s, lens, x, ldx, x_off, y, ldy, out, ldo, o_off, out16, nsplit);
  else
    hipLaunchKernelGGL(se_fc_scale_residual_kernel<false>, dim3(B * nsplit), dim3(1024), 0, stream, colsum, T, C, w1,
                       b1, w2, b2, bottleneck, s, lens, x, ldx, x_off, y, ldy, out, ldo, o_off, out16, nsplit);
  return hipGetLastError();
}

// all-binary16 variant: 16-B loads/stores of 8 halfs, fp32 arithmetic.  312 -> 156 MB per block output.
__global__ __launch_bounds__(256) void se_scale_residual_f16_kernel(
    const uint16_t* __restrict__ x, int ldx, int x_off, const uint16_t* __restrict__ y, int ldy,
    const float* __restrict__ s, uint16_t* __restrict__ out, int ldo, int o_off, int T, int C,
    long long total8) {
  typedef _Float16 f16x8e __attribute__((ext_vector_type(8)));
  const int cols8 = C >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8;
       i += (long long)gridDim.x * 256) {
    const long long m = i / cols8;
    const int c = (int)(i - m * cols8) * 8;
    const int b = (int)(m / T);
    const f16x8e xv = *reinterpret_cast<const f16x8e*>(x + m * ldx + x_off + c);
    const f16x8e yv = *reinterpret_cast<const f16x8e*>(y + m * ldy + c);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(s + (long long)b * C + c);
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(s + (long long)b * C + c + 4);
    f16x8e o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      o[q] = (_Float16)((float)xv[q] + (float)yv[q] * s0[q]);
      o[4 + q] = (_Float16)((float)xv[4 + q] + (float)yv[4 + q] * s1[q]);
    }
    *reinterpret_cast<f16x8e*>(out + m * ldo + o_off + c) = o;
  }
}

hipError_t launch_se_scale_residual_f16(const uint16_t* x16, int ldx, int x_off, const uint16_t* y16,
                                        int ldy, const float* s, uint16_t* out16, int ldo, int o_off,
                                        int B, int T, int C, hipStream_t stream) {
  if ((ldx | x_off | ldy | ldo | o_off | C) & 7) return hipErrorInvalidValue;
  const long long total8 = (long long)B * T * (C >> 3);
  long long blocks = (total8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(se_scale_residual_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x16,
                     ldx, x_off, y16, ldy, s, out16, ldo, o_off, T, C, total8);
  return hipGetLastError();
}

// The SE block of the all-binary16 back-end in ONE launch (round 6), the twin of se_fc_scale_residual_kernel: a
// workgroup per utterance requests its first rows of x / y (binary16, 16-B = 8 channels per thread), runs the two FCs
// on the column sums the conv3 epilogue left (se_fc_body: the stand-alone kernel's code), and streams
// out = half(x + y * s) with s in registers.  Replaces se_fc_from_colsum (15 us of L2 round trips on 256 workgroups) +
// se_scale_residual_f16 (25 us) + a launch boundary, three times per forward of a 1.3-ms step.
__global__ __launch_bounds__(1024) void se_fc_scale_residual_f16_kernel(
    const float* __restrict__ colsum, int T, int C, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, int bott, float* __restrict__ s_out,
    const int* __restrict__ lens, const uint16_t* __restrict__ x, int ldx, int x_off, const uint16_t* __restrict__ y,
    int ldy, uint16_t* __restrict__ out, int ldo, int o_off, int nsplit) {
  typedef _Float16 f16x8e __attribute__((ext_vector_type(8)));
  __shared__ __attribute__((aligned(16))) float mean[1024];
  __shared__ __attribute__((aligned(16))) float hidden[256];
  __shared__ __attribute__((aligned(16))) float s_lds[1024];
  constexpr int U = 8, PRE = 4;
  const int b = blockIdx.x / nsplit, part = blockIdx.x - b * nsplit, tid = threadIdx.x;
  const int chunk = (T + nsplit - 1) / nsplit, t0 = part * chunk, t1 = t0 + chunk < T ? t0 + chunk : T;
  const int cols8 = C >> 3;                  // 64 / 128 (C = 512 / 1024): divides the 1024 threads
  const int c = (tid % cols8) * 8, r0 = tid / cols8, rp = 1024 / cols8;
  const long long m0 = (long long)b * T;
  const uint16_t* xp = x + m0 * ldx + x_off + c;
  const uint16_t* yp = y + m0 * ldy + c;
  f16x8e xv[U], yv[U];
  auto request = [&](int u) {
    const int r = t0 + r0 + u * rp;
    if (r < t1) {
      xv[u] = *reinterpret_cast<const f16x8e*>(xp + (long long)r * ldx);
      yv[u] = *reinterpret_cast<const f16x8e*>(yp + (long long)r * ldy);
    }
  };
#pragma unroll
  for (int u = 0; u < PRE; ++u) request(u);
  se_fc_body<1024, true>(colsum, b, T, C, w1, b1, w2, b2, bott, s_out, lens, mean, hidden, s_lds);
#pragma unroll
  for (int u = PRE; u < U; ++u) request(u);
  __syncthreads();
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(&s_lds[c]);
  const f32x4 s1 = *reinterpret_cast<const f32x4*>(&s_lds[c + 4]);
  uint16_t* op = out + m0 * ldo + o_off + c;
  for (int rb = t0 + r0; rb < t1; rb += U * rp) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = rb + u * rp;
      if (r < t1) {
        f16x8e o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                  // (the expression of se_scale_residual_f16_kernel: same bits)
          o[q] = (_Float16)((float)xv[u][q] + (float)yv[u][q] * s0[q]);
          o[4 + q] = (_Float16)((float)xv[u][4 + q] + (float)yv[u][4 + q] * s1[q]);
        }
        *reinterpret_cast<f16x8e*>(op + (long long)r * ldo) = o;
        const int rn = r + U * rp;
        if (rn < t1) {
          xv[u] = *reinterpret_cast<const f16x8e*>(xp + (long long)rn * ldx);
          yv[u] = *reinterpret_cast<const f16x8e*>(yp + (long long)rn * ldy);
        }
      }
    }
  }
}

bool se_fc_scale_residual_f16_supported(int T, int C, int bottleneck) {
  return C <= 1024 && (C & 255) == 0 && bottleneck <= 256 && (bottleneck & 31) == 0 && T >= 64 &&
         1024 % (C >> 3) == 0;
}

hipError_t launch_se_fc_scale_residual_f16(const float* colsum, int B, int T, int C, const float* w1, const float* b1,
                                           const float* w2t, const float* b2, int bottleneck, float* s,
                                           const int* lens, const uint16_t* x16, int ldx, int x_off,
                                           const uint16_t* y16, int ldy, uint16_t* out16, int ldo, int o_off,
                                           hipStream_t stream) {
  if (!se_fc_scale_residual_f16_supported(T, C, bottleneck) || ((ldx | x_off | ldy | ldo | o_off) & 7))
    return hipErrorInvalidValue;
  const int nsplit = se_row_split(B, T);
  hipLaunchKernelGGL(se_fc_scale_residual_f16_kernel, dim3(B * nsplit), dim3(1024), 0, stream, colsum, T, C, w1, b1, w2t,
                     b2, bottleneck, s, lens, x16, ldx, x_off, y16, ldy, out16, ldo, o_off, nsplit);
  return hipGetLastError();
}

// ------------------------------------------------------------------ ASTP global-context statistics
// grid = (B, C/256), block = 256: thread = 1 channel... channel-parallel, two passes over T
// (mean, then centred sum of squares: same two-pass form torch.var uses, no E[x^2]-m^2 cancellation).
__global__ __launch_bounds__(256) void astp_stats_kernel(const float* __restrict__ h, int ldh, int T,
                                                         int C, float* __restrict__ stats,
                                                         const int* __restrict__ lens) {
  const int b = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const float* base = h + (long long)b * T * ldh + c;
  if (lens) T = lens[b];
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int t = 0;
  for (; t + 3 < T; t += 4) {
    s0 += base[(long long)t * ldh];
    s1 += base[(long long)(t + 1) * ldh];
    s2 += base[(long long)(t + 2) * ldh];
    s3 += base[(long long)(t + 3) * ldh];
  }
  for (; t < T; ++t) s0 += base[(long long)t * ldh];
  const float mean = ((s0 + s1) + (s2 + s3)) / (float)T;
  float q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  t = 0;
  for (; t + 3 < T; t += 4) {
    float d0 = base[(long long)t * ldh] - mean, d1 = base[(long long)(t + 1) * ldh] - mean;
    float d2 = base[(long long)(t + 2) * ldh] - mean, d3 = base[(long long)(t + 3) * ldh] - mean;
    q0 += d0 * d0; q1 += d1 * d1; q2 += d2 * d2; q3 += d3 * d3;
  }
  for (; t < T; ++t) { float d = base[(long long)t * ldh] - mean; q0 += d * d; }
  const float var = ((q0 + q1) + (q2 + q3)) / (float)(T - 1);      // unbiased (torch.var default)
  stats[(long long)b * 2 * C + c] = mean;
  stats[(long long)b * 2 * C + C + c] = sqrtf(var + 1e-7f);
}

// bias_img[b][j] = b1[j] + sum_c W1[j][C + c] * stats[b][c]   (c over the 2C mean|std entries)
// grid = B, block = 256: wavefront per output row.
__global__ __launch_bounds__(256) void astp_context_bias_kernel(const float* __restrict__ stats,
                                                                int C2, const float* __restrict__ w1,
                                                                int ldw1, int w_off,
                                                                const float* __restrict__ b1,
                                                                int bott, float* __restrict__ out) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* st = stats + (long long)b * C2;
  for (int j = wave; j < bott; j += 4) {
    const float* wr = w1 + (long long)j * ldw1 + w_off;
    float v = 0.f;
    for (int c = lane * 4; c < C2; c += 256) {
      f32x4 w = *reinterpret_cast<const f32x4*>(wr + c);
      f32x4 x = *reinterpret_cast<const f32x4*>(st + c);
      v += w[0] * x[0] + w[1] * x[1] + w[2] * x[2] + w[3] * x[3];
    }
    v = wave_sum(v);
    if (lane == 0) out[(long long)b * bott + j] = v + b1[j];
  }
}

// Same statistics when the producing GEMM already left per-64-row-tile column sums (catconv with
// `colsum`): the mean comes from those, so h is read ONCE (centred squares, still two-pass exact).
// grid = (B, C/256), block = 256: lane = 4 channels (16-B loads), the 4 wavefronts split T,
// 8 independent row loads in flight per lane.
template <typename TH>
__device__ __forceinline__ f32x4 load4_as_f32(const TH* p);
template <>
__device__ __forceinline__ f32x4 load4_as_f32<float>(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}
template <>
__device__ __forceinline__ f32x4 load4_as_f32<uint16_t>(const uint16_t* p) {
  typedef _Float16 f16x4e __attribute__((ext_vector_type(4)));
  const f16x4e v = *reinterpret_cast<const f16x4e*>(p);
  return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

template <typename TH>
__global__ __launch_bounds__(256) void astp_std_from_colsum_kernel(
    const TH* __restrict__ h, int ldh, int T, int C, const float* __restrict__ colsum,
    float* __restrict__ stats, const int* __restrict__ lens) {
  __shared__ f32x4 red[4][64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y * 256 + lane * 4;
  const long long r0 = (long long)b * T, r1 = r0 + T - 1;
  const int t_first = (int)(r0 / 64), t_last = (int)(r1 / 64);
  f32x4 mean = {0.f, 0.f, 0.f, 0.f};
  for (int tm = t_first; tm <= t_last; ++tm) {
    const int first_img = (int)(((long long)tm * 64) / T);
    const int which = (first_img == b) ? 0 : 1;
    mean += *reinterpret_cast<const f32x4*>(colsum + ((long long)tm * 2 + which) * C + c);
  }
  const TH* base = h + r0 * ldh + c;
  if (lens) T = lens[b];              // ragged batch: zero rows beyond; statistics over the own frames
  mean *= 1.f / (float)T;
  f32x4 q[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) q[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int t = wave;
  for (; t + 28 < T; t += 32) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = load4_as_f32<TH>(base + (long long)(t + 4 * u) * ldh);
#pragma unroll
    for (int u = 0; u < 8; ++u) { const f32x4 d = v[u] - mean; q[u] += d * d; }
  }
  for (; t < T; t += 4) {
    const f32x4 d = load4_as_f32<TH>(base + (long long)t * ldh) - mean;
    q[0] += d * d;
  }
  red[wave][lane] = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
  __syncthreads();
  if (wave == 0) {
    const f32x4 var = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) * (1.f / (float)(T - 1));
    f32x4 sd;
#pragma unroll
    for (int k = 0; k < 4; ++k) sd[k] = sqrtf(var[k] + 1e-7f);
    *reinterpret_cast<f32x4*>(stats + (long long)b * 2 * C + c) = mean;
    *reinterpret_cast<f32x4*>(stats + (long long)b * 2 * C + C + c) = sd;
  }
}

// Column sums of squares of rows [m_begin, M) of D, in the layout of ConvGemmParams::colsum ([64-row tile][image
// part][N]): the rows of a layer that the persistent GEMM did not take (its epilogue sums the squares of its own
// rows).  One workgroup per (64-row tile, 256 columns); thread = 4 columns x 16 row slots.
__global__ __launch_bounds__(256) void colsumsq_rows_kernel(const float* __restrict__ D, int ldd, int d_off,
                                                            int m_begin, int M, int HW, int N,
                                                            float* __restrict__ colsumsq) {
  __shared__ f32x4 red[2][4][64];
  const int t64 = m_begin / 64 + blockIdx.x;
  const int m0 = t64 * 64;
  const int lane = threadIdx.x & 63, rs = threadIdx.x >> 6;             // 4 row slots of 16 rows
  const int n = blockIdx.y * 256 + lane * 4;
  const int rb = (m0 / HW + 1) * HW - m0;                                // rows of the tile in its first image
  f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
  if (n < N) {
#pragma unroll 4
    for (int r = rs * 16; r < rs * 16 + 16; ++r) {
      const int m = m0 + r;
      if (m < M) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(D + (long long)m * ldd + d_off + n);
        if (r < rb) q0 += v * v; else q1 += v * v;
      }
    }
  }
  red[0][rs][lane] = q0;
  red[1][rs][lane] = q1;
  __syncthreads();
  if (rs < 2 && n < N) {
    const f32x4 s = (red[rs][0][lane] + red[rs][1][lane]) + (red[rs][2][lane] + red[rs][3][lane]);
    *reinterpret_cast<f32x4*>(colsumsq + ((long long)t64 * 2 + rs) * N + n) = s;
  }
}

hipError_t launch_colsumsq_rows(const float* D, int ldd, int d_off, int m_begin, int M, int HW, int N,
                                float* colsumsq, hipStream_t stream) {
  if (m_begin >= M) return hipSuccess;
  if ((m_begin & 63) || (N & 3) || (ldd & 3) || (d_off & 3) || HW < 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(colsumsq_rows_kernel, dim3((M - m_begin + 63) / 64, (N + 255) / 256), dim3(256), 0, stream, D, ldd,
                     d_off, m_begin, M, HW, N, colsumsq);
  return hipGetLastError();
}

// ASTP global context from the GEMM epilogue's column sums AND sums of squares: mean = S1 / T,
// std = sqrt((S2 - T mean^2) / (T - 1) + 1e-7) (pooling_layers.py:128-133, unbiased).  No pass over h -- unless the
// single-pass form has cancelled: the fp32 sums carry ~1e-6 of S2 as error, so when S2 - T mean^2 < 1e-3 S2 (a channel
// whose std is below ~3 % of its mean: rare behind a ReLU, but torch.var is two-pass and would get it right) the
// variance of that one (utterance, channel) is recomputed from its T values of h around the mean.
__global__ __launch_bounds__(256) void astp_std_from_sums_kernel(const float* __restrict__ colsum,
                                                                 const float* __restrict__ colsumsq, int T, int C,
                                                                 float* __restrict__ stats,
                                                                 const float* __restrict__ h, int ldh) {
  const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const long long r0 = (long long)b * T, r1 = r0 + T - 1;
  const int t_first = (int)(r0 / 64), t_last = (int)(r1 / 64);
  float s1 = 0.f, s2 = 0.f;
  for (int tm = t_first; tm <= t_last; ++tm) {
    const int which = ((int)(((long long)tm * 64) / T) == b) ? 0 : 1;
    s1 += colsum[((long long)tm * 2 + which) * C + c];
    s2 += colsumsq[((long long)tm * 2 + which) * C + c];
  }
  const float mean = s1 / (float)T;
  float ss = s2 - (float)T * mean * mean;
  if (h && ss < 1e-3f * s2) {                      // cancelled: two-pass on this column
    const float* col = h + r0 * ldh + c;
    float a0 = 0.f, a1 = 0.f;
    int t = 0;
    for (; t + 1 < T; t += 2) {
      const float d0 = col[(long long)t * ldh] - mean, d1 = col[(long long)(t + 1) * ldh] - mean;
      a0 += d0 * d0; a1 += d1 * d1;
    }
    if (t < T) { const float d0 = col[(long long)t * ldh] - mean; a0 += d0 * d0; }
    ss = a0 + a1;
  }
  const float var = fmaxf(ss, 0.f) / (float)(T - 1);
  stats[(long long)b * 2 * C + c] = mean;
  stats[(long long)b * 2 * C + C + c] = sqrtf(var + 1e-7f);
}

hipError_t launch_astp_std_from_sums(const float* colsum, const float* colsumsq, int B, int T, int C, float* stats,
                                     hipStream_t stream, const float* h, int ldh) {
  if (T < 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(astp_std_from_sums_kernel, dim3(B, (C + 255) / 256), dim3(256), 0, stream, colsum, colsumsq, T, C,
                     stats, h, ldh);
  return hipGetLastError();
}

hipError_t launch_astp_std_from_colsum(const float* h, int ldh, int B, int T, int C,
                                       const float* colsum, float* stats, hipStream_t stream, const int* lens) {
  if ((C & 255) || T < 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(astp_std_from_colsum_kernel<float>, dim3(B, C / 256), dim3(256), 0, stream, h, ldh,
                     T, C, colsum, stats, lens);
  return hipGetLastError();
}

hipError_t launch_astp_std_from_colsum_f16(const uint16_t* h16, int ldh, int B, int T, int C,
                                           const float* colsum, float* stats, hipStream_t stream,
                                           const int* lens) {
  if ((C & 255) || T < 64 || (ldh & 3)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(astp_std_from_colsum_kernel<uint16_t>, dim3(B, C / 256), dim3(256), 0, stream, h16,
                     ldh, T, C, colsum, stats, lens);
  return hipGetLastError();
}

hipError_t launch_astp_stats(const float* h, int ldh, int B, int T, int C, float* stats,
                             hipStream_t stream, const int* lens) {
  hipLaunchKernelGGL(astp_stats_kernel, dim3(B, (C + 255) / 256), dim3(256), 0, stream, h, ldh, T,
                     C, stats, lens);
  return hipGetLastError();
}

hipError_t launch_astp_context_bias(const float* h, int ldh, int B, int T, int C, const float* w1,
                                    int ldw1, const float* b1, int bottleneck, float* stats,
                                    float* bias_img, hipStream_t stream) {
  const int* lens = nullptr;
  hipLaunchKernelGGL(astp_stats_kernel, dim3(B, (C + 255) / 256), dim3(256), 0, stream, h, ldh, T,
                     C, stats, lens);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(astp_context_bias_kernel, dim3(B), dim3(256), 0, stream, stats, 2 * C, w1,
                     ldw1, C, b1, bottleneck, bias_img);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------- ASTP pooling
// Per (b, c): alpha = softmax_t(e), mean = sum alpha h, var = sum alpha h^2 - mean^2,
// std = sqrt(max(var, 1e-7)).  Online softmax, one pass over e and h.
// grid = (B, C/256), block = 256 (thread = channel; lanes contiguous in c -> coalesced rows).
__global__ __launch_bounds__(256) void astp_pool_kernel(const float* __restrict__ e, int lde,
                                                        const float* __restrict__ h, int ldh,
                                                        int T, int C, float* __restrict__ pooled,
                                                        const int* __restrict__ lens) {
  const int b = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const float* ep = e + (long long)b * T * lde + c;
  const float* hp = h + (long long)b * T * ldh + c;
  if (lens) T = lens[b];                    // softmax over the utterance's own frames only
  // pass 1: max (keeps exp arguments <= 0 exactly like torch.softmax's max-subtraction)
  float mx = -INFINITY;
  int t = 0;
  float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
  for (; t + 3 < T; t += 4) {
    m0 = fmaxf(m0, ep[(long long)t * lde]);
    m1 = fmaxf(m1, ep[(long long)(t + 1) * lde]);
    m2 = fmaxf(m2, ep[(long long)(t + 2) * lde]);
    m3 = fmaxf(m3, ep[(long long)(t + 3) * lde]);
  }
  for (; t < T; ++t) m0 = fmaxf(m0, ep[(long long)t * lde]);
  mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
  // pass 2: sums (e is re-read from L2)
  float l0 = 0, l1 = 0, a0 = 0, a1 = 0, q0 = 0, q1 = 0;
  t = 0;
  for (; t + 1 < T; t += 2) {
    float w0 = expf(ep[(long long)t * lde] - mx), w1 = expf(ep[(long long)(t + 1) * lde] - mx);
    float x0 = hp[(long long)t * ldh], x1 = hp[(long long)(t + 1) * ldh];
    l0 += w0; l1 += w1;
    a0 += w0 * x0; a1 += w1 * x1;
    q0 += w0 * x0 * x0; q1 += w1 * x1 * x1;
  }
  for (; t < T; ++t) {
    float w0 = expf(ep[(long long)t * lde] - mx);
    float x0 = hp[(long long)t * ldh];
    l0 += w0; a0 += w0 * x0; q0 += w0 * x0 * x0;
  }
  const float inv = 1.f / (l0 + l1);
  const float mean = (a0 + a1) * inv;
  const float var = (q0 + q1) * inv - mean * mean;
  pooled[(long long)b * 2 * C + c] = mean;
  pooled[(long long)b * 2 * C + C + c] = sqrtf(fmaxf(var, 1e-7f));
}

// grid = (B, C/256): merge the per-tile online-softmax tuples written by the logit GEMM
__global__ __launch_bounds__(256) void astp_pool_from_partials_kernel(
    const float* __restrict__ partials, int T, int C, float* __restrict__ pooled) {
  const int b = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const long long r0 = (long long)b * T, r1 = r0 + T - 1;
  const int t_first = (int)(r0 / 64), t_last = (int)(r1 / 64);
  float mx = -1e30f;
  for (int tm = t_first; tm <= t_last; ++tm) {
    const int which = ((int)(((long long)tm * 64) / T) == b) ? 0 : 1;
    mx = fmaxf(mx, partials[(((long long)tm * 2 + which) * C + c) * 4]);
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int tm = t_first; tm <= t_last; ++tm) {
    const int which = ((int)(((long long)tm * 64) / T) == b) ? 0 : 1;
    const f32x4 e = *reinterpret_cast<const f32x4*>(partials + (((long long)tm * 2 + which) * C + c) * 4);
    const float sc = expf(e[0] - mx);
    s0 += e[1] * sc; s1 += e[2] * sc; s2 += e[3] * sc;
  }
  const float inv = 1.f / s0;
  const float mean = s1 * inv;
  const float var = s2 * inv - mean * mean;
  pooled[(long long)b * 2 * C + c] = mean;
  pooled[(long long)b * 2 * C + C + c] = sqrtf(fmaxf(var, 1e-7f));
}

hipError_t launch_astp_pool_from_partials(const float* partials, int B, int T, int C, float* pooled,
                                          hipStream_t stream) {
  if (T < 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(astp_pool_from_partials_kernel, dim3(B, (C + 255) / 256), dim3(256), 0, stream,
                     partials, T, C, pooled);
  return hipGetLastError();
}

hipError_t launch_astp_pool(const float* e, int lde, const float* h, int ldh, int B, int T, int C,
                            float* pooled, hipStream_t stream, const int* lens) {
  hipLaunchKernelGGL(astp_pool_kernel, dim3(B, (C + 255) / 256), dim3(256), 0, stream, e, lde, h,
                     ldh, T, C, pooled, lens);
  return hipGetLastError();
}

}  // namespace wsamd
