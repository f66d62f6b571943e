// Non-GEMM kernels of the ResNet and CAM++ forwards on channels-last activations.
//
// Reference semantics:
//   stem conv       wespeaker/models/resnet.py:128-134,175 / campplus.py:286-292,325
//   TSTP            wespeaker/models/pooling_layers.py:78-85 (unbiased var + 1e-7)
//   CAM context     wespeaker/models/campplus.py:108-135 (global mean + 100-frame segment means,
//                   ceil_mode, Linear -> ReLU -> Linear -> sigmoid)
#include "kernels.h"

#include <cstdlib>

namespace wsamd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// -------------------------------------------------------------------------------------- stem
// out[((b*F + f)*T + t)*C + c] = relu(b[c] + sum_{dy,dx} w[c][dy*3+dx] * img(f+dy-1, t+dx-1)),
// img(f, t) = feats[(b*T + t)*F + f];  C = 32.
// One workgroup = an 8 (f) x 32 (t) tile of output pixels, one thread per pixel and all 32 channels:
// the 10 x 34 input patch and the 32 x 9 weights are staged in LDS once (weights are read back as
// wave-uniform broadcasts), each thread keeps its 9 inputs in registers and writes its pixel's 32
// channels as one contiguous run (64 B binary16 / 128 B fp32; 32 neighbouring threads -> 2-4 KB).
// The tensor it writes (B*F*T*32 values) is 40x the size of what it reads: store-bound.
constexpr int ST_FH = 8, ST_TW = 32;

// Round 5: the same tile with the roles turned for the STORES.  Above, a store instruction of a wavefront writes 16 B
// of 64 different pixels (64 cache lines, a quarter of each: the L2 has to merge eight instructions per line) and the
// kernel runs at 3.1 TB/s.  Here a thread owns 4 channels (8 threads per pixel) of the 8 pixels of one tile COLUMN:
// lane -> (pixel column tx = lane / 8 of 8 neighbours, channel quad lane % 8), so one store instruction is 1 KB of
// contiguous memory (8 whole 128-B pixels); its 10 x 3 inputs and 4 x 10 weights live in registers.  The sum of an
// output is the expression of the kernel above, term for term.
template <bool OUT16>
__global__ __launch_bounds__(256) void stem_conv3x3_rows_kernel(const float* __restrict__ feats, int T,
                                                                int F, const float* __restrict__ w,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ out,
                                                                uint16_t* __restrict__ out16,
                                                                const int* __restrict__ lens) {
  __shared__ float in_s[ST_FH + 2][ST_TW + 2 + 1];
  const int t0 = blockIdx.x * ST_TW, f0 = blockIdx.y * ST_FH, b = blockIdx.z;
  const int tid = threadIdx.x;
  const float* img = feats + (long long)b * T * F;
  for (int i = tid; i < (ST_FH + 2) * (ST_TW + 2); i += 256) {
    const int tx = i / (ST_FH + 2), fy = i - tx * (ST_FH + 2);   // f fastest: contiguous in feats
    const int tt = t0 + tx - 1, ff = f0 + fy - 1;
    in_s[fy][tx] = (tt >= 0 && tt < T && ff >= 0 && ff < F) ? img[(long long)tt * F + ff] : 0.f;
  }
  const int c4 = tid & 7, tx = tid >> 3;                          // channels 4 c4 .. +3 of tile column tx
  float wv[4][9], bv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[q][k] = w[(c4 * 4 + q) * 9 + k];
    bv[q] = bias[c4 * 4 + q];
  }
  __syncthreads();
  float col[ST_FH + 2][3];
#pragma unroll
  for (int r = 0; r < ST_FH + 2; ++r)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) col[r][dx] = in_s[r][tx + dx];
  const int t = t0 + tx;
  if (t >= T) return;
  const bool padded = lens && t >= lens[b];     // ragged batch: columns beyond the utterance stay zero
  typedef _Float16 f16x4c __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int fy = 0; fy < ST_FH; ++fy) {
    const int f = f0 + fy;
    if (f >= F) break;
    const float* in0 = col[fy];
    const float* in1 = col[fy + 1];
    const float* in2 = col[fy + 2];
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sacc = bv[q];                          // bias
      sacc += wv[q][0] * in0[0] + wv[q][1] * in0[1] + wv[q][2] * in0[2] + wv[q][3] * in1[0];
      sacc += wv[q][4] * in1[1] + wv[q][5] * in1[2] + wv[q][6] * in2[0] + wv[q][7] * in2[1];
      sacc += wv[q][8] * in2[2];
      r[q] = padded ? 0.f : fmaxf(sacc, 0.f);
    }
    const long long pix = ((long long)b * F + f) * T + t;
    if (OUT16) {
      f16x4c hv;
#pragma unroll
      for (int q = 0; q < 4; ++q) hv[q] = (_Float16)r[q];
      *reinterpret_cast<f16x4c*>(out16 + pix * 32 + c4 * 4) = hv;
    } else {
      *reinterpret_cast<f32x4*>(out + pix * 32 + c4 * 4) = r;
    }
  }
}

hipError_t launch_stem_conv3x3(const float* feats, int B, int T, int F, const float* w,
                               const float* b, int C, float* out, hipStream_t stream,
                               uint16_t* out16, const int* lens) {
  if (C != 32 || B <= 0) return C != 32 ? hipErrorInvalidValue : hipSuccess;
  dim3 grid((T + ST_TW - 1) / ST_TW, (F + ST_FH - 1) / ST_FH, B);
  if (out16)
    hipLaunchKernelGGL(stem_conv3x3_rows_kernel<true>, grid, dim3(256), 0, stream, feats, T, F, w, b, out, out16, lens);
  else
    hipLaunchKernelGGL(stem_conv3x3_rows_kernel<false>, grid, dim3(256), 0, stream, feats, T, F, w, b, out, out16, lens);
  return hipGetLastError();
}

// -------------------------------------------------------------------------------------- TSTP
// grid = (B*F, ceil(C/64)); block = 256 = 4 time-groups x 64 channels.  Two passes over T
// (mean, then centred squares) like torch.var; partial sums combined through LDS.
__device__ __forceinline__ float load_act(const float* p) { return *p; }
__device__ __forceinline__ float load_act(const uint16_t* p) {
  return (float)*reinterpret_cast<const _Float16*>(p);
}

template <typename TX>
__global__ __launch_bounds__(256) void tstp_kernel(const TX* __restrict__ x, int ldx, int F, int T,
                                                   int C, const float* __restrict__ pre_scale,
                                                   const float* __restrict__ pre_shift,
                                                   float* __restrict__ pooled,
                                                   const int* __restrict__ lens) {
  __shared__ float red[4][64];
  const int bf = blockIdx.x, b = bf / F, f = bf - b * F;
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  const bool cok = c < C;
  const TX* base = x + (long long)bf * T * ldx + (cok ? c : 0);
  if (lens) T = lens[b];                      // ragged batch: statistics over the valid columns only
  float ps = 1.f, pb = 0.f;
  const bool pre = pre_scale != nullptr;
  if (pre && cok) { ps = pre_scale[c]; pb = pre_shift[c]; }
  float s = 0.f;
  for (int t = grp; t < T; t += 4) {
    float v = load_act(base + (long long)t * ldx);
    if (pre) v = relu_f(v * ps + pb);
    s += v;
  }
  red[grp][cl] = s;
  __syncthreads();
  const float mean = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)T;
  __syncthreads();
  float q = 0.f;
  for (int t = grp; t < T; t += 4) {
    float v = load_act(base + (long long)t * ldx);
    if (pre) v = relu_f(v * ps + pb);
    const float d = v - mean;
    q += d * d;
  }
  red[grp][cl] = q;
  __syncthreads();
  if (grp == 0 && cok) {
    const float var = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)(T - 1);
    const long long CF = (long long)C * F;
    pooled[(long long)b * 2 * CF + (long long)c * F + f] = mean;
    pooled[(long long)b * 2 * CF + CF + (long long)c * F + f] = sqrtf(var + 1e-7f);
  }
}

hipError_t launch_tstp(const float* x, int ldx, int B, int F, int T, int C, const float* pre_scale,
                       const float* pre_shift, float* pooled, hipStream_t stream, const int* lens) {
  hipLaunchKernelGGL(tstp_kernel<float>, dim3(B * F, (C + 63) / 64), dim3(256), 0, stream, x, ldx, F, T,
                     C, pre_scale, pre_shift, pooled, lens);
  return hipGetLastError();
}

hipError_t launch_tstp_f16(const uint16_t* x16, int ldx, int B, int F, int T, int C,
                           const float* pre_scale, const float* pre_shift, float* pooled,
                           hipStream_t stream, const int* lens) {
  hipLaunchKernelGGL(tstp_kernel<uint16_t>, dim3(B * F, (C + 63) / 64), dim3(256), 0, stream, x16, ldx, F,
                     T, C, pre_scale, pre_shift, pooled, lens);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------- CAM context mask
// grid = B, block = 256.  C <= 256 channels, hidden <= 128, Cout <= 64, segs <= 32.
// Segment sums over the T axis (lanes = channels, 256/C time groups), then the two small FCs
// per segment.  mask[b][seg][Cout].
__global__ __launch_bounds__(1024) void cam_context_kernel(const float* __restrict__ h, int ldh, int T,
                                                          int C, int seg_len, int segs,
                                                          const float* __restrict__ w1,
                                                          const float* __restrict__ b1, int hidden,
                                                          const float* __restrict__ w2,
                                                          const float* __restrict__ b2, int Cout,
                                                          float* __restrict__ mask,
                                                          const int* __restrict__ lens) {
  extern __shared__ float sm[];
  float* segsum = sm;                       // [groups][segs][C]
  const int NT = blockDim.x;                // 256, or 1024 for long utterances (more rows in flight per segment)
  const int groups = NT / C;
  float* ctx = sm + groups * segs * C;      // [C]
  float* hid = ctx + C;                     // [hidden]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int c = tid % C, grp = tid / C;
  const float* base = h + (long long)b * T * ldh + c;
  // ragged batch: this utterance has lens[b] <= T frames, i.e. ceil(lens[b] / seg_len) segments of its
  // own (avg_pool1d with ceil_mode on ITS length); the mask rows of the segments beyond are zeroed
  if (lens) T = lens[b];
  if (grp < groups) {
    for (int s = 0; s < segs; ++s) {
      const int t0 = s * seg_len, t1 = min(T, t0 + seg_len);
      // 8 independent loads in flight per lane: the serial form was latency-bound (~0.5 us per
      // dependent strided load, 50 of them per lane at T' = 99)
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int t = t0 + grp;
      for (; t + 7 * groups < t1; t += 8 * groups) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += base[(long long)(t + u * groups) * ldh];
      }
      for (; t < t1; t += groups) a[0] += base[(long long)t * ldh];
      segsum[(grp * segs + s) * C + c] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
  }
  __syncthreads();
  // fold the time groups: segsum[0][s][c] <- sum_g
  for (int i = tid; i < segs * C; i += NT) {
    float v = 0.f;
    for (int g = 0; g < groups; ++g) v += segsum[g * segs * C + i];
    segsum[i] = v;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  for (int s = 0; s < segs; ++s) {
    const int t0 = s * seg_len, t1 = min(T, t0 + seg_len);
    if (t1 <= t0) {                                    // (uniform) segment beyond this utterance
      if (tid < Cout) mask[((long long)b * segs + s) * Cout + tid] = 0.f;
      continue;
    }
    if (tid < C) {
      float tot = 0.f;
      for (int k = 0; k < segs; ++k) tot += segsum[k * C + tid];
      ctx[tid] = tot / (float)T + segsum[s * C + tid] / (float)(t1 - t0);
    }
    __syncthreads();
    for (int j = wave; j < hidden; j += NT / 64) {    // hid = relu(W1 ctx + b1)
      const float* wr = w1 + (long long)j * C;
      float v = 0.f;
      for (int k = lane; k < C; k += 64) v += wr[k] * ctx[k];
      v = wave_sum32(v);
      if (lane == 0) hid[j] = relu_f(v + b1[j]);
    }
    __syncthreads();
    if (tid < Cout) {                                  // m = sigmoid(W2 hid + b2)
      const float* wr = w2 + (long long)tid * hidden;
      float v = 0.f;
      for (int k = 0; k < hidden; ++k) v += wr[k] * hid[k];
      v += b2[tid];
      mask[((long long)b * segs + s) * Cout + tid] = 1.f / (1.f + expf(-v));
    }
    __syncthreads();
  }
}

// Single-segment case (T <= seg_len, i.e. utterances up to 4 s at the TDNN's stride 2): the segment
// mean equals the global mean, and that comes from the producing GEMM's per-64-row column sums
// (ConvGemmParams::colsum) -- h is not read again.  grid = B, block = 128 (C <= 128, hidden <= 128).
__global__ __launch_bounds__(128) void cam_context_from_colsum_kernel(
    const float* __restrict__ colsum, int T, int C, const float* __restrict__ w1,
    const float* __restrict__ b1, int hidden, const float* __restrict__ w2,
    const float* __restrict__ b2, int Cout, float* __restrict__ mask, const int* __restrict__ lens) {
  __shared__ float ctx[128], hid[128];
  const int b = blockIdx.x, tid = threadIdx.x;
  const long long r0 = (long long)b * T, r1 = r0 + T - 1;
  const int t_first = (int)(r0 / 64), t_last = (int)(r1 / 64);
  if (tid < C) {
    float v = 0.f;
    for (int tm = t_first; tm <= t_last; ++tm) {
      const int first_img = (int)(((long long)tm * 64) / T);
      v += colsum[((long long)tm * 2 + (first_img == b ? 0 : 1)) * C + tid];
    }
    // (ragged batch: rows beyond lens[b] are zeros; the divisor is the utterance's own length)
    ctx[tid] = 2.f * v / (float)(lens ? lens[b] : T);   // global mean + (identical) segment mean
  }
  __syncthreads();
  if (tid < hidden) {                            // hid = relu(W1 ctx + b1): thread per row, 16-B loads
    const float* wr = w1 + (long long)tid * C;
    float v = b1[tid];
    for (int k = 0; k < C; k += 4) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + k);
      v += w4[0] * ctx[k] + w4[1] * ctx[k + 1] + w4[2] * ctx[k + 2] + w4[3] * ctx[k + 3];
    }
    hid[tid] = relu_f(v);
  }
  __syncthreads();
  if (tid < Cout) {                              // m = sigmoid(W2 hid + b2)
    const float* wr = w2 + (long long)tid * hidden;
    float v = b2[tid];
    for (int k = 0; k < hidden; k += 4) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + k);
      v += w4[0] * hid[k] + w4[1] * hid[k + 1] + w4[2] * hid[k + 2] + w4[3] * hid[k + 3];
    }
    mask[(long long)b * Cout + tid] = 1.f / (1.f + expf(-v));
  }
}

hipError_t launch_cam_context_from_colsum(const float* colsum, int B, int T, int C, const float* w1,
                                          const float* b1, int hidden, const float* w2, const float* b2,
                                          int Cout, float* mask, hipStream_t stream, const int* lens) {
  if (C > 128 || (C & 3) || hidden > 128 || (hidden & 3) || Cout > 128 || T < 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cam_context_from_colsum_kernel, dim3(B), dim3(128), 0, stream, colsum, T, C, w1, b1,
                     hidden, w2, b2, Cout, mask, lens);
  return hipGetLastError();
}

hipError_t launch_cam_context(const float* h, int ldh, int B, int T, int C, int seg_len,
                              const float* w1, const float* b1, int hidden, const float* w2,
                              const float* b2, int Cout, float* mask, hipStream_t stream, const int* lens) {
  if (C > 256 || 256 % C != 0 || hidden > 128 || Cout > 64) return hipErrorInvalidValue;
  const int segs = (T + seg_len - 1) / seg_len;
  // round 6: 1024 threads per utterance from 128 frames on (the kernel is a chain of strided row loads: 35 us per
  // launch with 256 threads at T' = 199, 52 launches per forward of a CAM++ batch of 4-s utterances)
  int nt = T >= 128 && 1024 % C == 0 ? 1024 : 256;
  if (((size_t)(nt / C) * segs * C + C + hidden) * sizeof(float) > 60 * 1024) nt = 256;
  const int groups = nt / C;
  const size_t lds = ((size_t)groups * segs * C + C + hidden) * sizeof(float);
  if (lds > 60 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cam_context_kernel, dim3(B), dim3(nt), lds, stream, h, ldh, T, C, seg_len, segs,
                     w1, b1, hidden, w2, b2, Cout, mask, lens);
  return hipGetLastError();
}

}  // namespace wsamd
