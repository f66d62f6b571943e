// Cosine scoring and adaptive score normalisation (AS-norm / S-norm) on gfx950.
//
// Replaces the numpy / sklearn back-end of the reference's default scoring recipe:
//   wespeaker/bin/score.py:38-72        mean-subtract, cosine of each trial pair
//   wespeaker/bin/score_norm.py:26-36   get_mean_std: L2-normalise, emb x cohort^T, sort rows
//                                       descending, mean / std of the top-N cohort scores
//   wespeaker/bin/score_norm.py:93-115  0.5 * ((s - mu_e) / sd_e + (s - mu_t) / sd_t)
//
// Data layout: a "unit table" is (ceil4(n), ld) float32 with ld = ceil32(dim); rows are the
// mean-subtracted, L2-normalised embeddings, zero padded on both axes, so that the dense score
// matrix is one exact-fp32 MFMA "NT" GEMM of the conv kernel (both operands K-contiguous) without
// edge predicates.  Row statistics never sort: the N-th largest score of a row is found by a
// 4-pass most-significant-byte radix select on the order-preserving integer image of the floats
// (exact, ties included), then one pass accumulates sum and sum of squares above the threshold in
// float64.  All of it is HBM/L2-bound streaming over the score rows; the GEMM is MFMA-bound.
#include "kernels.h"

namespace wsamd {

typedef float f32x4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum_fs(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_ds(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------- unit tables
// One wavefront per output row (4 rows per workgroup).  rows >= n and columns >= dim are zeroed.
// mag[r] = || emb[r] - mean_vec ||  (the "enroll_mag / test_mag" columns of score_norm.py:107-108).
__global__ __launch_bounds__(256) void cos_prepare_kernel(
    const float* __restrict__ emb, const float* __restrict__ mean_vec, int n, int n_pad, int dim,
    int ld, float* __restrict__ unit, float* __restrict__ mag) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_pad) return;
  float* out = unit + (long long)row * ld;
  if (row >= n) {
    for (int d = lane; d < ld; d += 64) out[d] = 0.f;
    return;
  }
  const float* src = emb + (long long)row * dim;
  float ss = 0.f;
  for (int d = lane; d < dim; d += 64) {
    const float v = src[d] - (mean_vec ? mean_vec[d] : 0.f);
    ss += v * v;
  }
  const float nrm = sqrtf(wave_sum_fs(ss));
  for (int d = lane; d < ld; d += 64) {
    float v = 0.f;
    if (d < dim) v = (src[d] - (mean_vec ? mean_vec[d] : 0.f)) / nrm;
    out[d] = v;
  }
  if (mag && lane == 0) mag[row] = nrm;
}

hipError_t launch_cos_prepare(const float* emb, const float* mean_vec, int n, int dim, float* unit,
                              float* mag, hipStream_t stream) {
  const int n_pad = (n + 3) & ~3, ld = (dim + 31) & ~31;
  if (n_pad == 0) return hipSuccess;
  hipLaunchKernelGGL(cos_prepare_kernel, dim3((n_pad + 3) / 4), dim3(256), 0, stream, emb, mean_vec,
                     n, n_pad, dim, ld, unit, mag);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------- trial pairs
// 16 lanes per trial (4 trials per wavefront): gathers two unit rows (L2 resident for realistic
// tables: 4.9k x 256 floats = 5 MB) and reduces their dot product across the lane group.
__global__ __launch_bounds__(256) void cos_pairs_kernel(
    const float* __restrict__ ua, const float* __restrict__ ub, int ld,
    const int32_t* __restrict__ idx_a, const int32_t* __restrict__ idx_b, long long num,
    float* __restrict__ out) {
  const int sub = threadIdx.x & 15;
  const long long t = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const bool live = t < num;
  const long long tt = live ? t : 0;
  const f32x4s* a = reinterpret_cast<const f32x4s*>(ua + (long long)idx_a[tt] * ld);
  const f32x4s* b = reinterpret_cast<const f32x4s*>(ub + (long long)idx_b[tt] * ld);
  float s = 0.f;
  for (int c = sub; c < ld / 4; c += 16) {
    const f32x4s x = a[c], y = b[c];
    s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (live && sub == 0) out[t] = s;
}

hipError_t launch_cos_pairs(const float* ua, const float* ub, int ld, const int32_t* idx_a,
                            const int32_t* idx_b, long long num, float* out, hipStream_t stream) {
  if (num <= 0) return hipSuccess;
  const long long blocks = (num * 16 + 255) / 256;
  hipLaunchKernelGGL(cos_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ua, ub, ld,
                     idx_a, idx_b, num, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------- top-N statistics
// Order-preserving map float -> uint32 (larger float <=> larger key; -0 < +0 is harmless here).
__device__ __forceinline__ uint32_t f2key(float v) {
  const uint32_t b = __float_as_uint(v);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  const uint32_t b = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
  return __uint_as_float(b);
}

// One workgroup per score row.  top_n_eff = min(top_n, n_cols) >= 1.
// mean = (1/N) sum of the N largest scores; sd = sqrt((1/N) sum (x - mean)^2)  (np.std, ddof = 0).
__global__ __launch_bounds__(256) void topn_stats_kernel(
    const float* __restrict__ S, int ld, int n_cols, int top_n_eff, float* __restrict__ mean,
    float* __restrict__ sd) {
  __shared__ unsigned hist[4][256];          // one histogram per wavefront: 4x fewer collisions
  __shared__ unsigned s_prefix, s_remaining;
  __shared__ double red[2][4];
  const int tid = threadIdx.x, wave = tid >> 6;
  const float* row = S + (long long)blockIdx.x * ld;
  const int nvec = n_cols >> 2;
  const f32x4s* row4 = reinterpret_cast<const f32x4s*>(row);

  uint32_t thr_key = 0;                      // keys > thr_key are strictly inside the top-N
  unsigned ties = 0;                         // how many copies of thr_key complete the top-N
  const bool select = top_n_eff < n_cols;
  if (select) {
    if (tid == 0) { s_prefix = 0; s_remaining = (unsigned)top_n_eff; }
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int i = tid; i < 4 * 256; i += 256) (&hist[0][0])[i] = 0;
      __syncthreads();
      const uint32_t prefix = s_prefix;
      auto vote = [&](float v) {
        const uint32_t k = f2key(v);
        if (pass == 0 || (k >> (shift + 8)) == prefix) atomicAdd(&hist[wave][(k >> shift) & 255], 1u);
      };
      for (int i = tid; i < nvec; i += 256) {
        const f32x4s v = row4[i];
        vote(v[0]); vote(v[1]); vote(v[2]); vote(v[3]);
      }
      for (int i = nvec * 4 + tid; i < n_cols; i += 256) vote(row[i]);
      __syncthreads();
      // bins from the top: find the one where the running count reaches `remaining`
      if (wave == 0) {
        const int lane = tid;
        // lane l owns bins 255-4l .. 252-4l (descending order)
        unsigned c[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int b = 255 - (4 * lane + q);
          c[q] = hist[0][b] + hist[1][b] + hist[2][b] + hist[3][b];
          tot += c[q];
        }
        unsigned incl = tot;                 // inclusive prefix sum over lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned up = __shfl_up(incl, o, 64);
          if (lane >= o) incl += up;
        }
        const unsigned rem = s_remaining;
        unsigned before = incl - tot;        // scores in strictly higher bins
        if (before < rem && incl >= rem) {   // exactly one lane
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (before < rem && before + c[q] >= rem) {
              s_prefix = (prefix << 8) | (unsigned)(255 - (4 * lane + q));
              s_remaining = rem - before;
              before = rem;                  // stop
            } else {
              before += c[q];
            }
          }
        }
      }
      __syncthreads();
    }
    thr_key = s_prefix;
    ties = s_remaining;
  }

  double s1 = 0.0, s2 = 0.0;
  auto take = [&](float v) {
    if (!select || f2key(v) > thr_key) { s1 += (double)v; s2 += (double)v * (double)v; }
  };
  for (int i = tid; i < nvec; i += 256) {
    const f32x4s v = row4[i];
    take(v[0]); take(v[1]); take(v[2]); take(v[3]);
  }
  for (int i = nvec * 4 + tid; i < n_cols; i += 256) take(row[i]);
  s1 = wave_sum_ds(s1);
  s2 = wave_sum_ds(s2);
  if ((tid & 63) == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  __syncthreads();
  if (tid == 0) {
    double a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    double b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    if (select) {
      const double tv = (double)key2f(thr_key);
      a += (double)ties * tv;
      b += (double)ties * tv * tv;
    }
    const double N = (double)top_n_eff;
    const double m = a / N;
    const double var = b / N - m * m;
    mean[blockIdx.x] = (float)m;
    sd[blockIdx.x] = (float)sqrt(var > 0.0 ? var : 0.0);
  }
}

hipError_t launch_topn_stats(const float* S, int ld, int n_rows, int n_cols, int top_n, float* mean,
                             float* sd, hipStream_t stream) {
  if (n_rows <= 0) return hipSuccess;
  if (n_cols <= 0 || top_n <= 0 || (ld & 3)) return hipErrorInvalidValue;
  const int eff = top_n < n_cols ? top_n : n_cols;
  hipLaunchKernelGGL(topn_stats_kernel, dim3(n_rows), dim3(256), 0, stream, S, ld, n_cols, eff, mean,
                     sd);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------- AS-norm
__global__ __launch_bounds__(256) void asnorm_pairs_kernel(
    const float* __restrict__ score, const int32_t* __restrict__ idx_e,
    const int32_t* __restrict__ idx_t, const float* __restrict__ e_mean,
    const float* __restrict__ e_sd, const float* __restrict__ t_mean,
    const float* __restrict__ t_sd, long long num, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= num) return;
  const float s = score[i];
  const int e = idx_e[i], t = idx_t[i];
  out[i] = 0.5f * ((s - e_mean[e]) / e_sd[e] + (s - t_mean[t]) / t_sd[t]);
}

hipError_t launch_asnorm_pairs(const float* score, const int32_t* idx_e, const int32_t* idx_t,
                               const float* e_mean, const float* e_sd, const float* t_mean,
                               const float* t_sd, long long num, float* out, hipStream_t stream) {
  if (num <= 0) return hipSuccess;
  hipLaunchKernelGGL(asnorm_pairs_kernel, dim3((unsigned)((num + 255) / 256)), dim3(256), 0, stream,
                     score, idx_e, idx_t, e_mean, e_sd, t_mean, t_sd, num, out);
  return hipGetLastError();
}

}  // namespace wsamd
