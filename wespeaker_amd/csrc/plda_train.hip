// Sufficient statistics for two-covariance PLDA training / adaptation on gfx950 (float64).
//
// Replaces the per-speaker numpy loop of the reference:
//   wespeaker/utils/plda/two_cov_plda.py:48-66   PldaStats.add_samples  (class mean, offset scatter)
//   wespeaker/utils/plda/two_cov_plda.py:95-107  TwoCovPLDA.__init__    (train-set mean subtraction,
//                                                 length normalisation, one add_samples per speaker)
//   wespeaker/utils/plda/two_cov_plda.py:261-275 adapt: data mean / np.cov of the adaptation set
// The N x D^2 work (N = #utterances, up to ~10^6) lives here; the D x D algebra of the EM
// iterations (inverses, Cholesky, eigh) stays on the host in float64.
//
// Rows must be grouped by class: class c owns rows [group_offsets[c], group_offsets[c+1]).
//   y_i   = (x_i - mean_vec) * s_i,  s_i = sqrt(D)/|x_i - mean_vec| if normalize_length else 1
//   mu_c  = mean_{i in c} y_i
//   S     = sum_c sum_{i in c} (y_i - mu_c)(y_i - mu_c)^T          (offset_scatter)
// S is a "TN" contraction over the rows: v_mfma_f64_16x16x4_f64 on 64x64 output tiles, the row
// range split over grid.z into deterministic partial sums, rows staged (and centred, in float64)
// through LDS straight from the float32 embeddings -- y is never materialised in HBM.
#include "kernels.h"

namespace wsamd {

typedef double f64x4t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum_dt(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wavefront per row: s_i; one workgroup (4 waves) handles 4 rows
template <typename T>
__global__ __launch_bounds__(256) void stats_rowscale_kernel(const T* __restrict__ emb, int n,
                                                             int dim,
                                                             const double* __restrict__ mean_vec,
                                                             int normalize, double* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  double s = 1.0;
  if (normalize) {
    const T* x = emb + (long long)row * dim;
    double ss = 0.0;
    for (int d = lane; d < dim; d += 64) {
      const double v = (double)x[d] - (mean_vec ? mean_vec[d] : 0.0);
      ss += v * v;
    }
    s = sqrt((double)dim) / sqrt(wave_sum_dt(ss));
  }
  if (lane == 0) scale[row] = s;
}

// one workgroup per class: mu_c and the class index of its rows
template <typename T>
__global__ __launch_bounds__(256) void stats_class_mean_kernel(
    const T* __restrict__ emb, int dim, const int32_t* __restrict__ group_offsets,
    const double* __restrict__ mean_vec, const double* __restrict__ scale,
    double* __restrict__ class_mean, int32_t* __restrict__ row_class) {
  const int c = blockIdx.x, tid = threadIdx.x;
  const int r0 = group_offsets[c], r1 = group_offsets[c + 1];
  for (int r = r0 + tid; r < r1; r += 256) row_class[r] = c;
  for (int d = tid; d < dim; d += 256) {
    const double mv = mean_vec ? mean_vec[d] : 0.0;
    double acc = 0.0;
    for (int r = r0; r < r1; ++r) acc += ((double)emb[(long long)r * dim + d] - mv) * scale[r];
    class_mean[(long long)c * dim + d] = r1 > r0 ? acc / (double)(r1 - r0) : 0.0;
  }
}

constexpr int TK = 16;           // rows per staging step
constexpr int TS = 64 + 16;      // LDS row stride in doubles: rows k, k+1 land 32 banks apart, so the
                                 // (16 columns x 2 rows) of a 32-lane ds_read_b64 phase cover 64 banks

// grid = (tiles_d * tiles_d, Z); block = 256.  partial[z][D][D].
template <typename T>
__global__ __launch_bounds__(256) void stats_scatter_kernel(
    const T* __restrict__ emb, int n, int dim, const double* __restrict__ mean_vec,
    const double* __restrict__ scale, const double* __restrict__ class_mean,
    const int32_t* __restrict__ row_class, int rows_per_z, double* __restrict__ partial) {
  __shared__ double Ys[2][TK * TS];          // [operand][row k][64 columns]
  const int tiles = (dim + 63) / 64;
  const int tm = blockIdx.x / tiles, tn = blockIdx.x - tm * tiles;
  const int d1 = tm * 64, d2 = tn * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lk = lane >> 4;
  const int r_begin = blockIdx.y * rows_per_z;
  const int r_end = min(n, r_begin + rows_per_z);
  f64x4t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4t){0.0, 0.0, 0.0, 0.0};
  // staging role: thread -> (row k = tid >> 4, 4 columns starting at (tid & 15) * 4) of each operand
  const int sk = tid >> 4, sc = (tid & 15) * 4;
  for (int r0 = r_begin; r0 < r_end; r0 += TK) {
    const int r = r0 + sk;
    const bool live = r < r_end;
    const int cls = live ? row_class[r] : 0;
    const double s = live ? scale[r] : 0.0;
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const int dbase = (op == 0 ? d1 : d2) + sc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int d = dbase + q;
        double v = 0.0;
        if (live && d < dim) {
          const double mv = mean_vec ? mean_vec[d] : 0.0;
          v = ((double)emb[(long long)r * dim + d] - mv) * s - class_mean[(long long)cls * dim + d];
        }
        Ys[op][sk * TS + sc + q] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < TK; ks += 4) {
      double a[2], b[2];
#pragma unroll
      for (int im = 0; im < 2; ++im) a[im] = Ys[0][(ks + lk) * TS + wm * 32 + im * 16 + li];
#pragma unroll
      for (int in = 0; in < 2; ++in) b[in] = Ys[1][(ks + lk) * TS + wn * 32 + in * 16 + li];
#pragma unroll
      for (int im = 0; im < 2; ++im)
#pragma unroll
        for (int in = 0; in < 2; ++in)
          acc[im][in] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[im], b[in], acc[im][in], 0, 0, 0);
    }
    __syncthreads();
  }
  // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  double* out = partial + (long long)blockIdx.y * dim * dim;
#pragma unroll
  for (int im = 0; im < 2; ++im)
#pragma unroll
    for (int in = 0; in < 2; ++in)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int m = d1 + wm * 32 + im * 16 + lk + 4 * rg;
        const int nn = d2 + wn * 32 + in * 16 + li;
        if (m < dim && nn < dim) out[(long long)m * dim + nn] = acc[im][in][rg];
      }
}

__global__ __launch_bounds__(256) void stats_reduce_kernel(const double* __restrict__ partial, int Z,
                                                           long long count, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  double s = 0.0;
  for (int z = 0; z < Z; ++z) s += partial[(long long)z * count + i];
  out[i] = s;
}

// scratch layout (doubles): scale[n] | partial[Z][dim][dim] | row_class (n int32, rounded up)
static int scatter_splits(int n, int dim) {
  const int tiles = ((dim + 63) / 64) * ((dim + 63) / 64);
  int z = (1024 + tiles - 1) / tiles;                   // ~4 workgroups per CU
  const int max_z = (n + 255) / 256;                    // at least 256 rows per split
  if (z > max_z) z = max_z;
  return z < 1 ? 1 : z;
}

int64_t plda_stats_scratch_doubles(int n, int dim) {
  const int z = scatter_splits(n, dim);
  return (int64_t)n + (int64_t)z * dim * dim + ((int64_t)n + 1) / 2 + 8;
}

template <typename T>
static hipError_t launch_plda_stats_t(const T* emb, int n, int dim, const int32_t* group_offsets,
                                      int n_groups, const double* mean_vec, int normalize_length,
                                      double* class_mean, double* scatter, double* scratch,
                                      hipStream_t stream) {
  if (n <= 0 || n_groups <= 0) return hipSuccess;
  const int Z = scatter_splits(n, dim);
  double* scale = scratch;
  double* partial = scale + n;
  int32_t* row_class = reinterpret_cast<int32_t*>(partial + (long long)Z * dim * dim);
  hipLaunchKernelGGL(stats_rowscale_kernel<T>, dim3((n + 3) / 4), dim3(256), 0, stream, emb, n, dim,
                     mean_vec, normalize_length, scale);
  hipLaunchKernelGGL(stats_class_mean_kernel<T>, dim3(n_groups), dim3(256), 0, stream, emb, dim,
                     group_offsets, mean_vec, scale, class_mean, row_class);
  const int tiles = (dim + 63) / 64;
  int rows_per_z = (n + Z - 1) / Z;
  rows_per_z = (rows_per_z + TK - 1) / TK * TK;
  hipLaunchKernelGGL(stats_scatter_kernel<T>, dim3(tiles * tiles, Z), dim3(256), 0, stream, emb, n, dim,
                     mean_vec, scale, class_mean, row_class, rows_per_z, partial);
  const long long count = (long long)dim * dim;
  hipLaunchKernelGGL(stats_reduce_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                     partial, Z, count, scatter);
  return hipGetLastError();
}

// emb: float32 rows (extractor output) or float64 rows (output of an earlier embedding-processing link,
// which the reference keeps in float64: utils/embedding_processing.py:132-178)
hipError_t launch_plda_stats(const void* emb, int emb_is_f64, int n, int dim, const int32_t* group_offsets,
                             int n_groups, const double* mean_vec, int normalize_length,
                             double* class_mean, double* scatter, double* scratch,
                             hipStream_t stream) {
  if (emb_is_f64)
    return launch_plda_stats_t(reinterpret_cast<const double*>(emb), n, dim, group_offsets, n_groups,
                               mean_vec, normalize_length, class_mean, scatter, scratch, stream);
  return launch_plda_stats_t(reinterpret_cast<const float*>(emb), n, dim, group_offsets, n_groups,
                             mean_vec, normalize_length, class_mean, scatter, scratch, stream);
}

// ------------------------------------------------------------------------------------------------
// One link of the embedding-processing chain (utils/embedding_processing.py:177-217) applied to rows:
//   y = (x - sub) [M]      then optionally y /= |y|      (mean-subtract / lda / length-norm)
// x (n, d_in) float32 or float64, sub float64[d_in] or null, M float64 (d_in, d_out) row-major or null
// (then d_out == d_in), out float64 (n, d_out).  One workgroup per row; the row sits in LDS as f64.
template <typename TX>
__global__ __launch_bounds__(256) void rows_affine_kernel(const TX* __restrict__ x, int d_in,
                                                          const double* __restrict__ sub,
                                                          const double* __restrict__ M, int d_out,
                                                          int normalize, double* __restrict__ out) {
  extern __shared__ double rowv[];             // v[d_in] | y[d_out] | red[4]
  double* v = rowv;
  double* y = rowv + d_in;
  double* red = y + d_out;
  const long long r = blockIdx.x;
  const int tid = threadIdx.x;
  for (int d = tid; d < d_in; d += 256) v[d] = (double)x[r * d_in + d] - (sub ? sub[d] : 0.0);
  __syncthreads();
  double ss = 0.0;
  for (int j = tid; j < d_out; j += 256) {
    double acc;
    if (M) {
      acc = 0.0;
      for (int d = 0; d < d_in; ++d) acc += v[d] * M[(long long)d * d_out + j];
    } else {
      acc = v[j];
    }
    y[j] = acc;
    ss += acc * acc;
  }
  double scale = 1.0;
  if (normalize) {
    ss = wave_sum_dt(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    scale = 1.0 / sqrt((red[0] + red[1]) + (red[2] + red[3]));
  }
  __syncthreads();
  for (int j = tid; j < d_out; j += 256) out[r * d_out + j] = y[j] * scale;
}

hipError_t launch_rows_affine(const void* x, int x_is_f64, int n, int d_in, const double* sub,
                              const double* M, int d_out, int normalize, double* out,
                              hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (!M && d_out != d_in) return hipErrorInvalidValue;
  const size_t lds = (size_t)(d_in + d_out + 4) * sizeof(double);
  if (lds > 60 * 1024) return hipErrorInvalidValue;
  if (x_is_f64)
    hipLaunchKernelGGL(rows_affine_kernel<double>, dim3(n), dim3(256), lds, stream,
                       reinterpret_cast<const double*>(x), d_in, sub, M, d_out, normalize, out);
  else
    hipLaunchKernelGGL(rows_affine_kernel<float>, dim3(n), dim3(256), lds, stream,
                       reinterpret_cast<const float*>(x), d_in, sub, M, d_out, normalize, out);
  return hipGetLastError();
}

}  // namespace wsamd
