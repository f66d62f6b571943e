// Two-covariance PLDA scoring on gfx950 in float64 (the reference scores in numpy float64:
// wespeaker/utils/plda/two_cov_plda.py:156-184, eval_sv :204-256).
//
// Closed form used for the matrix / pair kernels.  With, per enrollment model i
// and dimension d:   c = n psi / (n psi + 1),  v = 1 + psi / (n psi + 1),
//   LLR[i][j] = K_i - 1/2 sum_d a_id t_jd^2 - 1/2 sum_d b_id e_id^2 + sum_d g_id e_id t_jd
//   K_i = -1/2 (sum_d log v - sum_d log(psi + 1)),  a = 1/v - 1/(psi + 1),  b = c^2 / v,  g = c / v
// i.e. one "NT" GEMM  [g*e | -a/2] (Ne x 2D)  x  [t | t^2]^T (2D x Nt)  plus a per-row constant, which
// stays valid when n differs per enrollment model.  The GEMM runs on v_mfma_f64_16x16x4_f64.
#include "kernels.h"

#include <cstdlib>

namespace wsamd {

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum over 256 threads (4 waves); red: >= 4 doubles of LDS
__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------- eval_sv pre-processing + transform
// One workgroup per output vector.  group_offsets == nullptr: output g <- input row g.
template <typename TIn>
__global__ __launch_bounds__(256) void plda_prepare_kernel(
    const TIn* __restrict__ emb, const int32_t* __restrict__ group_offsets, int dim,
    const double* __restrict__ mean_vec, const double* __restrict__ transform,
    const double* __restrict__ offset, int pre_norm, int post_norm, double* __restrict__ out) {
  extern __shared__ double smd[];      // v[dim] | y[dim] | red[4]
  double* v = smd;
  double* y = smd + dim;
  double* red = smd + 2 * dim;
  const int g = blockIdx.x, tid = threadIdx.x;
  int r0 = g, r1 = g + 1;
  if (group_offsets) { r0 = group_offsets[g]; r1 = group_offsets[g + 1]; }
  const double sq = sqrt((double)dim);
  double ss = 0.0;
  for (int d = tid; d < dim; d += 256) {
    double s = 0.0;
    const double mv = mean_vec ? mean_vec[d] : 0.0;
    for (int r = r0; r < r1; ++r) s += (double)emb[(long long)r * dim + d] - mv;
    s /= (double)(r1 - r0);
    v[d] = s;
    ss += s * s;
  }
  if (pre_norm) {                               // norm_embeddings(mean)  (plda_utils.py:46-58)
    const double nrm = sqrt(block_sum_d(ss, red));
    for (int d = tid; d < dim; d += 256) v[d] = sq * v[d] / nrm;
  }
  __syncthreads();
  // y = transform v + offset : wavefront per output row
  const int lane = tid & 63, wave = tid >> 6;
  for (int j = wave; j < dim; j += 4) {
    const double* tr = transform + (long long)j * dim;
    double s = 0.0;
    for (int d = lane; d < dim; d += 64) s += tr[d] * v[d];
    s = wave_sum_d(s);
    if (lane == 0) y[j] = s + offset[j];
  }
  __syncthreads();
  double factor = 1.0;
  if (post_norm) {                              // transform_embedding (two_cov_plda.py:159-162)
    double s2 = 0.0;
    for (int d = tid; d < dim; d += 256) s2 += y[d] * y[d];
    factor = sq / sqrt(block_sum_d(s2, red));
  }
  for (int d = tid; d < dim; d += 256) out[(long long)g * dim + d] = factor * y[d];
}

hipError_t launch_plda_prepare(const void* emb, int emb_is_f64, const int32_t* group_offsets,
                               int n_out, int dim, const double* mean_vec, const double* transform,
                               const double* offset, int pre_norm, int post_norm, double* out,
                               hipStream_t stream) {
  if (n_out <= 0) return hipSuccess;
  const size_t lds = (2 * (size_t)dim + 4) * sizeof(double);
  if (emb_is_f64)
    hipLaunchKernelGGL(plda_prepare_kernel<double>, dim3(n_out), dim3(256), lds, stream,
                       reinterpret_cast<const double*>(emb), group_offsets, dim, mean_vec, transform,
                       offset, pre_norm, post_norm, out);
  else
    hipLaunchKernelGGL(plda_prepare_kernel<float>, dim3(n_out), dim3(256), lds, stream,
                       reinterpret_cast<const float*>(emb), group_offsets, dim, mean_vec, transform,
                       offset, pre_norm, post_norm, out);
  return hipGetLastError();
}

// Large-N form of the same pre-processing: (1) rows -> V (mean-sub, group mean, pre-norm), (2) Y = V
// transform^T + offset on the f64 MFMA GEMM, (3) optional row re-normalisation.  One workgroup per
// vector re-reads the D x D transform from L2 N times; the GEMM reads it once per 64-row tile.
template <typename TIn>
__global__ __launch_bounds__(256) void plda_rows_kernel(const TIn* __restrict__ emb,
                                                        const int32_t* __restrict__ group_offsets,
                                                        int n_out, int dim,
                                                        const double* __restrict__ mean_vec,
                                                        int pre_norm, double* __restrict__ V) {
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= n_out) return;
  int r0 = g, r1 = g + 1;
  if (group_offsets) { r0 = group_offsets[g]; r1 = group_offsets[g + 1]; }
  double ss = 0.0;
  for (int d = lane; d < dim; d += 64) {
    double s = 0.0;
    const double mv = mean_vec ? mean_vec[d] : 0.0;
    for (int r = r0; r < r1; ++r) s += (double)emb[(long long)r * dim + d] - mv;
    s /= (double)(r1 - r0);
    V[(long long)g * dim + d] = s;
    ss += s * s;
  }
  if (pre_norm) {
    const double f = sqrt((double)dim) / sqrt(wave_sum_d(ss));
    for (int d = lane; d < dim; d += 64) V[(long long)g * dim + d] *= f;
  }
}

__global__ __launch_bounds__(256) void plda_rownorm_kernel(double* __restrict__ Y, int n, int dim) {
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= n) return;
  double ss = 0.0;
  for (int d = lane; d < dim; d += 64) { const double y = Y[(long long)g * dim + d]; ss += y * y; }
  const double f = sqrt((double)dim) / sqrt(wave_sum_d(ss));
  for (int d = lane; d < dim; d += 64) Y[(long long)g * dim + d] *= f;
}

hipError_t launch_plda_rows(const void* emb, int emb_is_f64, const int32_t* group_offsets, int n_out,
                            int dim, const double* mean_vec, int pre_norm, double* V,
                            hipStream_t stream) {
  if (n_out <= 0) return hipSuccess;
  if (emb_is_f64)
    hipLaunchKernelGGL(plda_rows_kernel<double>, dim3((n_out + 3) / 4), dim3(256), 0, stream,
                       reinterpret_cast<const double*>(emb), group_offsets, n_out, dim, mean_vec,
                       pre_norm, V);
  else
    hipLaunchKernelGGL(plda_rows_kernel<float>, dim3((n_out + 3) / 4), dim3(256), 0, stream,
                       reinterpret_cast<const float*>(emb), group_offsets, n_out, dim, mean_vec,
                       pre_norm, V);
  return hipGetLastError();
}

hipError_t launch_plda_rownorm(double* Y, int n, int dim, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(plda_rownorm_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, Y, n, dim);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------- GEMM operands
// one wavefront per enrollment row
// n_sessions == nullptr: every model has n_uniform sessions; then the t^2 coefficients do not depend
// on the row and only the g*e half is written (K = dim, the a-term becomes a per-test constant).
__global__ __launch_bounds__(256) void plda_enroll_terms_kernel(
    const double* __restrict__ enroll, const int32_t* __restrict__ n_sessions, int n_uniform,
    int n_enroll, int dim, const double* __restrict__ psi, double* __restrict__ EA,
    double* __restrict__ rowc) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n_enroll) return;
  const double n = n_sessions ? (double)n_sessions[i] : (double)n_uniform;
  const int ldo = n_sessions ? 2 * dim : dim;
  double kc = 0.0;
  for (int d = lane; d < dim; d += 64) {
    const double p = psi[d];
    const double c = n * p / (n * p + 1.0);
    const double v = 1.0 + p / (n * p + 1.0);
    const double e = enroll[(long long)i * dim + d];
    const double a = 1.0 / v - 1.0 / (p + 1.0);
    EA[(long long)i * ldo + d] = (c / v) * e;
    if (n_sessions) EA[(long long)i * ldo + dim + d] = -0.5 * a;
    kc += -0.5 * (log(v) - log(p + 1.0)) - 0.5 * (c * c / v) * e * e;
  }
  kc = wave_sum_d(kc);
  if (lane == 0) rowc[i] = kc;
}

hipError_t launch_plda_enroll_terms(const double* enroll, const int32_t* n_sessions, int n_uniform,
                                    int n_enroll, int dim, const double* psi, double* EA,
                                    double* rowc, hipStream_t stream) {
  if (n_enroll <= 0) return hipSuccess;
  hipLaunchKernelGGL(plda_enroll_terms_kernel, dim3((n_enroll + 3) / 4), dim3(256), 0, stream,
                     enroll, n_sessions, n_uniform, n_enroll, dim, psi, EA, rowc);
  return hipGetLastError();
}

// uniform n: colc[j] = -1/2 sum_d a_d t_jd^2   (one wavefront per test vector)
__global__ __launch_bounds__(256) void plda_test_colc_kernel(const double* __restrict__ test,
                                                             int n_test, int dim, int n_uniform,
                                                             const double* __restrict__ psi,
                                                             double* __restrict__ colc) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= n_test) return;
  const double n = (double)n_uniform;
  double acc = 0.0;
  for (int d = lane; d < dim; d += 64) {
    const double p = psi[d];
    const double v = 1.0 + p / (n * p + 1.0);
    const double a = 1.0 / v - 1.0 / (p + 1.0);
    const double t = test[(long long)j * dim + d];
    acc += -0.5 * a * t * t;
  }
  acc = wave_sum_d(acc);
  if (lane == 0) colc[j] = acc;
}

hipError_t launch_plda_test_colc(const double* test, int n_test, int dim, int n_uniform,
                                 const double* psi, double* colc, hipStream_t stream) {
  if (n_test <= 0) return hipSuccess;
  hipLaunchKernelGGL(plda_test_colc_kernel, dim3((n_test + 3) / 4), dim3(256), 0, stream, test,
                     n_test, dim, n_uniform, psi, colc);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void plda_test_terms_kernel(const double* __restrict__ test,
                                                              long long total, int dim,
                                                              double* __restrict__ TT) {
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * 256) {
    const long long j = idx / dim;
    const int d = (int)(idx - j * dim);
    const double t = test[idx];
    TT[j * 2 * dim + d] = t;
    TT[j * 2 * dim + dim + d] = t * t;
  }
}

hipError_t launch_plda_test_terms(const double* test, int n_test, int dim, double* TT,
                                  hipStream_t stream) {
  const long long total = (long long)n_test * dim;
  if (total <= 0) return hipSuccess;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(plda_test_terms_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, test,
                     total, dim, TT);
  return hipGetLastError();
}

// ------------------------------------------------------------------------- f64 MFMA "NT" GEMM
// out[i][j] = rowc[i] + sum_k A[i][k] B[j][k].   64x64 tile per workgroup, 4 waves as 2x2,
// each wave 32x32 = 2x2 MFMA tiles of 16x16x4 (f64).  LDS rows padded to 18 doubles so the
// ds_read_b64 of a 32-lane group (16 rows x 2 k) touches 32 distinct 8-byte slots.
constexpr int DBK = 16;
constexpr int DS = DBK + 2;

__global__ __launch_bounds__(256) void plda_gemm_f64_kernel(const double* __restrict__ A,
                                                            const double* __restrict__ rowc,
                                                            const double* __restrict__ colc, int M,
                                                            const double* __restrict__ Bm, int N,
                                                            int K, double* __restrict__ out) {
  __shared__ double As[64 * DS];
  __shared__ double Bs[64 * DS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int li = lane & 15, lk = lane >> 4;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};

  // staging: 64 rows x 16 doubles = 1024 doubles per operand; thread -> row tid>>2, 4 doubles.
  // Rows beyond M / N are clamped (their products reach no stored output), so with K % 16 == 0 the loads
  // are unconditional 16-byte pairs and the next K-tile is requested before the current one is
  // multiplied (the predicated scalar form below made hipcc wait after every 8-byte load).
  const int sr = tid >> 2, sk = (tid & 3) * 4;
  if ((K & 15) == 0) {
    const int ma = m0 + sr < M ? m0 + sr : M - 1, nb = n0 + sr < N ? n0 + sr : N - 1;
    const double* ap = A + (long long)ma * K + sk;
    const double* bp = Bm + (long long)nb * K + sk;
    double2 a0 = *reinterpret_cast<const double2*>(ap), a1 = *reinterpret_cast<const double2*>(ap + 2);
    double2 b0 = *reinterpret_cast<const double2*>(bp), b1 = *reinterpret_cast<const double2*>(bp + 2);
    for (int k0 = 0; k0 < K; k0 += DBK) {
      As[sr * DS + sk] = a0.x; As[sr * DS + sk + 1] = a0.y; As[sr * DS + sk + 2] = a1.x; As[sr * DS + sk + 3] = a1.y;
      Bs[sr * DS + sk] = b0.x; Bs[sr * DS + sk + 1] = b0.y; Bs[sr * DS + sk + 2] = b1.x; Bs[sr * DS + sk + 3] = b1.y;
      __syncthreads();
      if (k0 + DBK < K) {
        a0 = *reinterpret_cast<const double2*>(ap + k0 + DBK);
        a1 = *reinterpret_cast<const double2*>(ap + k0 + DBK + 2);
        b0 = *reinterpret_cast<const double2*>(bp + k0 + DBK);
        b1 = *reinterpret_cast<const double2*>(bp + k0 + DBK + 2);
      }
#pragma unroll
      for (int ks = 0; ks < DBK; ks += 4) {
        double a[2], b[2];
#pragma unroll
        for (int im = 0; im < 2; ++im) a[im] = As[(wm * 32 + im * 16 + li) * DS + ks + lk];
#pragma unroll
        for (int in = 0; in < 2; ++in) b[in] = Bs[(wn * 32 + in * 16 + li) * DS + ks + lk];
#pragma unroll
        for (int im = 0; im < 2; ++im)
#pragma unroll
          for (int in = 0; in < 2; ++in)
            acc[im][in] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[im], b[in], acc[im][in], 0, 0, 0);
      }
      __syncthreads();
    }
  } else
  for (int k0 = 0; k0 < K; k0 += DBK) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + sk + q;
      const int ma = m0 + sr, nb = n0 + sr;
      As[sr * DS + sk + q] = (ma < M && k < K) ? A[(long long)ma * K + k] : 0.0;
      Bs[sr * DS + sk + q] = (nb < N && k < K) ? Bm[(long long)nb * K + k] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < DBK; ks += 4) {
      double a[2], b[2];
#pragma unroll
      for (int im = 0; im < 2; ++im) a[im] = As[(wm * 32 + im * 16 + li) * DS + ks + lk];
#pragma unroll
      for (int in = 0; in < 2; ++in) b[in] = Bs[(wn * 32 + in * 16 + li) * DS + ks + lk];
#pragma unroll
      for (int im = 0; im < 2; ++im)
#pragma unroll
        for (int in = 0; in < 2; ++in)
          acc[im][in] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[im], b[in], acc[im][in], 0, 0, 0);
    }
    __syncthreads();
  }
  // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int im = 0; im < 2; ++im)
#pragma unroll
    for (int in = 0; in < 2; ++in)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + im * 16 + lk + 4 * r;
        const int n = n0 + wn * 32 + in * 16 + li;
        if (m < M && n < N)
          out[(long long)m * N + n] = acc[im][in][r] + (rowc ? rowc[m] : 0.0) + (colc ? colc[n] : 0.0);
      }
}

// The same product on 128x128 tiles for matrices with many tiles (the >= 1e8-trial matrices that
// llr_matrix_sharded hands every rank, two_cov_plda.py:165-184 evaluated for a whole trial matrix): four wavefronts as
// 2 x 2, each 64x64 = 4 x 4 MFMA tiles, so a k-step of 4 issues 16 MFMAs (1024 cycles of the f64 pipe) for 8 LDS reads
// and a 16-wide K-tile 64 MFMAs per wavefront between two barriers -- the 64x64 kernel has 16 there and runs the pipe at
// well under half its rate.  Every output accumulates its k-steps in the same order as in the 64x64 kernel: same bits.
// K % 16 == 0 only (the general case stays on the small kernel).
__global__ __launch_bounds__(256, 2) void plda_gemm_f64_big_kernel(const double* __restrict__ A,
                                                                   const double* __restrict__ rowc,
                                                                   const double* __restrict__ colc, int M,
                                                                   const double* __restrict__ Bm, int N, int K,
                                                                   int n_tiles, double* __restrict__ out) {
  __shared__ double As[128 * DS];
  __shared__ double Bs[128 * DS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_m = (M + 127) / 128;
  const int li = lane & 15, lk = lane >> 4;
  // staging: 128 rows x 16 doubles per operand; thread -> row tid >> 1, 8 doubles
  const int sr = tid >> 1, sk = (tid & 1) * 8;
  double2 av[4], bv[4];
  const double* ap = nullptr;
  const double* bp = nullptr;
  // consecutive tiles walk down a column of tiles: they share the B panel (and each A panel is reused by the tiles of
  // the next columns while it is still in L2)
  auto set_tile = [&](int tile, int& m0, int& n0) {
    const int tn = tile / tiles_m, tm = tile - tn * tiles_m;
    m0 = tm * 128; n0 = tn * 128;
    const int ma = m0 + sr < M ? m0 + sr : M - 1, nb = n0 + sr < N ? n0 + sr : N - 1;   // (clamped rows reach no store)
    ap = A + (long long)ma * K + sk;
    bp = Bm + (long long)nb * K + sk;
  };
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      av[q] = *reinterpret_cast<const double2*>(ap + k0 + 2 * q);
      bv[q] = *reinterpret_cast<const double2*>(bp + k0 + 2 * q);
    }
  };
  // Persistent workgroups (two per CU), tiles tile, tile + grid, ...: a tile's 128 KB of scores are STORED WHILE THE
  // NEXT TILE IS MULTIPLIED -- the next tile's first K-tile is requested before the stores are issued (vmcnt retires in
  // order: waiting for those loads then does not wait for the stores behind them).  Measured (10 000 x 10 000, round 5):
  // the stores cost 0.18 ms of 0.96 (D = 192) either way -- one workgroup per tile, two LDS stages with one barrier per
  // K-tile and this persistent form all land at 0.49 - 0.51 (D = 192) / 0.62 - 0.65 (D = 512) of the f64 peak at a held
  // clock of 2.3 GHz; with the stores compiled out 0.62 / 0.70: what is left is the K loop itself (8 ds_read_b64 per 16
  // MFMAs, two barriers per 16-wide K-tile) at ~0.75 of the pipe's rate.
  int m0, n0;
  int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  set_tile(tile, m0, n0);
  fetch(0);
  while (true) {
    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += DBK) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        As[sr * DS + sk + 2 * q] = av[q].x; As[sr * DS + sk + 2 * q + 1] = av[q].y;
        Bs[sr * DS + sk + 2 * q] = bv[q].x; Bs[sr * DS + sk + 2 * q + 1] = bv[q].y;
      }
      __syncthreads();
      if (k0 + DBK < K) fetch(k0 + DBK);
#pragma unroll
      for (int ks = 0; ks < DBK; ks += 4) {
        double a[4], b[4];
#pragma unroll
        for (int im = 0; im < 4; ++im) a[im] = As[(wm * 64 + im * 16 + li) * DS + ks + lk];
#pragma unroll
        for (int in = 0; in < 4; ++in) b[in] = Bs[(wn * 64 + in * 16 + li) * DS + ks + lk];
#pragma unroll
        for (int im = 0; im < 4; ++im)
#pragma unroll
          for (int in = 0; in < 4; ++in)
            acc[im][in] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[im], b[in], acc[im][in], 0, 0, 0);
      }
      __syncthreads();
    }
    const int cm0 = m0, cn0 = n0;
    const int next = tile + (int)gridDim.x;
    if (next < n_tiles) {
      set_tile(next, m0, n0);
      fetch(0);                                  // in front of this tile's stores
    }
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = cm0 + wm * 64 + im * 16 + lk + 4 * r;
        if (m >= M) continue;
        const double rc = rowc ? rowc[m] : 0.0;
#pragma unroll
        for (int in = 0; in < 4; ++in) {
          const int n = cn0 + wn * 64 + in * 16 + li;
          if (n < N) out[(long long)m * N + n] = acc[im][in][r] + rc + (colc ? colc[n] : 0.0);
        }
      }
    if (next >= n_tiles) break;
    tile = next;
  }
}

hipError_t launch_plda_llr_gemm(const double* EA, const double* rowc, const double* colc,
                                int n_enroll, const double* TT, int n_test, int K, double* out,
                                hipStream_t stream) {
  if (n_enroll <= 0 || n_test <= 0) return hipSuccess;
  // env WS_PLDA_BIG_TILES: 0 = always the 64x64 kernel, 1 = the 128x128 kernel whenever K % 16 == 0; unset = the big
  // tiles once they fill the chip twice over (>= 1024 of them)
  static const int big_mode = [] { const char* ev = getenv("WS_PLDA_BIG_TILES"); return ev ? atoi(ev) : -1; }();
  const long long big_tiles = (long long)((n_enroll + 127) / 128) * ((n_test + 127) / 128);
  if ((K & 15) == 0 && big_mode != 0 && (big_mode == 1 || big_tiles >= 1024) && big_tiles < (1LL << 31)) {
    const long long resident = 2LL * current_device_cus();
    hipLaunchKernelGGL(plda_gemm_f64_big_kernel, dim3((unsigned)(big_tiles < resident ? big_tiles : resident)), dim3(256),
                       0, stream, EA, rowc, colc, n_enroll, TT, n_test, K, (int)big_tiles, out);
    return hipGetLastError();
  }
  dim3 grid((n_test + 63) / 64, (n_enroll + 63) / 64);
  hipLaunchKernelGGL(plda_gemm_f64_kernel, grid, dim3(256), 0, stream, EA, rowc, colc, n_enroll, TT,
                     n_test, K, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------- explicit trial list
// 16 lanes per trial (4 trials per wavefront): gathers two 2D-double rows (L2 / Infinity-Cache
// resident tables), 16-lane shuffle reduction.
__global__ __launch_bounds__(256) void plda_llr_pairs_kernel(
    const double* __restrict__ EA, const double* __restrict__ rowc, const double* __restrict__ colc,
    const double* __restrict__ TT, int K, const int32_t* __restrict__ idx_e,
    const int32_t* __restrict__ idx_t, long long num_trials, double* __restrict__ out) {
  const int sub = threadIdx.x & 15;
  for (long long p = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4; p < num_trials;
       p += ((long long)gridDim.x * 256) >> 4) {
    const int i = idx_e[p], j = idx_t[p];
    const double* ea = EA + (long long)i * K;
    const double* tt = TT + (long long)j * K;
    double s = 0.0;
    if (K & 1) {        // odd row length: rows are only 8-byte aligned, no double2 gathers
      for (int k = sub; k < K; k += 16) s += ea[k] * tt[k];
    } else {
      for (int k = sub * 2; k < K; k += 32) {
        const double2 a = *reinterpret_cast<const double2*>(ea + k);
        const double2 t = *reinterpret_cast<const double2*>(tt + k);
        s += a.x * t.x + a.y * t.y;
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0) out[p] = s + rowc[i] + (colc ? colc[j] : 0.0);
  }
}

hipError_t launch_plda_llr_pairs(const double* EA, const double* rowc, const double* colc,
                                 const double* TT, int K, const int32_t* idx_e, const int32_t* idx_t,
                                 int64_t num_trials, double* out, hipStream_t stream) {
  if (num_trials <= 0) return hipSuccess;
  long long blocks = (num_trials * 16 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(plda_llr_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, EA, rowc,
                     colc, TT, K, idx_e, idx_t, (long long)num_trials, out);
  return hipGetLastError();
}

// (Round 4 measured two ways of halving the gather bytes of a list that is grouped by enrollment model -- as trial
// files are: a kernel that keeps the [g * e] row in LDS per run of equal idx_e, and bucketing an ungrouped list on the
// device first (histogram / scan / scatter).  Neither survived: on a grouped list the plain kernel above already
// reads the enrollment row from L1 (16 consecutive trials per workgroup share it): 1 M grouped trials in 203 us
// against 388 us in random order, and the LDS form took 232 us + its switch; bucketing by global atomics costs 135 us
// per million trials, most of what it saves.  So: one kernel; callers that build the list themselves -- eval_sv,
// wespeaker_amd/plda.py -- order it by enrollment row on the host.)

// Gather yardstick (bench.py's PLDA roofline): 16 lanes read one K-double row per index and reduce it to its sum --
// the row-gather path of the scoring kernels (L2 / Infinity Cache -> CU) with no second operand and 8 B out per row.
__global__ __launch_bounds__(256) void row_gather_probe_kernel(const double* __restrict__ T, int K,
                                                               const int32_t* __restrict__ idx, long long n,
                                                               double* __restrict__ out) {
  const int sub = threadIdx.x & 15;
  for (long long p = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4; p < n; p += ((long long)gridDim.x * 256) >> 4) {
    const double* r = T + (long long)idx[p] * K;
    double s = 0.0;
    for (int k = sub * 2; k < K; k += 32) {
      const double2 a = *reinterpret_cast<const double2*>(r + k);
      s += a.x + a.y;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0) out[p] = s;
  }
}

hipError_t launch_row_gather_probe(const double* T, int K, const int32_t* idx, int64_t n, double* out,
                                   hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (K & 1) return hipErrorInvalidValue;
  long long blocks = (n * 16 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(row_gather_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, T, K, idx, (long long)n, out);
  return hipGetLastError();
}

}  // namespace wsamd
