// Kaldi-compatible log-mel filterbank + CMN on gfx950.
//
// Replaces torchaudio.compliance.kaldi.fbank as the reference calls it
// (wespeaker/cli/speaker.py:92-97, dataset/processor.py:518-525; native twin
// runtime/core/frontend/fbank.h:138-198) and the CMN of cli/speaker.py:98-99.
//
// One 64-lane wavefront per frame, four frames per workgroup:
//   coalesced load of the 400 samples -> DC removal (wave shuffle reduce) -> pre-emphasis ->
//   window -> 512-point real FFT as a 256-point complex Stockham radix-4 FFT in LDS (one radix-4
//   butterfly per lane per stage, 4 stages) + split/unpack -> |X|^2 -> sparse triangular mel
//   filters (<= 2 lanes-worth of bins) -> log(max(., eps)) -> one contiguous 80-float row store.
// HBM-bound by construction (64 KB in, 63 KB out per 2 s utterance); the FFT never leaves LDS.
#include "kernels.h"
#include <atomic>

namespace wsamd {

constexpr int FFT_N = 512;            // padded window (round_to_power_of_two)
constexpr int CN = FFT_N / 2;         // complex FFT size
constexpr int FRAMES_PER_BLOCK = 16;    // wavefronts (= frames in flight) per workgroup
constexpr int MEL_W_MAX = 1024;       // packed triangular weights (2 per FFT bin at most: 512)
constexpr int MEL_WPAD_MAX = 2048;    // the same weights zero padded per 16-bin pass (80 bins at 16 kHz: 1536)
constexpr int FBANK_WGS_PER_CU = 2;   // resident workgroups per CU: 32 wavefronts = the 8 per SIMD that <= 64 VGPRs allow (79 KB of LDS each)

// Two builds of one kernel body (fbank_kernel.inc).  DESIGN.md 6.0: the round-3 build let hipcc's SLP vectoriser
// pair the power-spectrum arithmetic into packed-fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 with op_sel and an
// inline constant); next to the binary16 GEMMs of another stream those returned wrong values in lanes 48..63 (the
// fourth 16-lane pass of the instruction) -- on every MI355X leased, never alone.  The shipped build is compiled
// without the packed-fp32 instruction forms (same arithmetic, one fp32 operation per instruction); the packed build
// stays as the reproducer behind ws_debug_fbank_mode(1).
#define WS_FBANK_FN(name) name
#if defined(__HIP_DEVICE_COMPILE__)
#define WS_FBANK_ATTR __attribute__((target("no-packed-fp32-ops")))
#else
#define WS_FBANK_ATTR                     // the host pass only needs the kernel's stub
#endif
#include "fbank_kernel.inc"
#undef WS_FBANK_FN
#undef WS_FBANK_ATTR
#define WS_FBANK_FN(name) name##_packed
#define WS_FBANK_ATTR
#include "fbank_kernel.inc"
#undef WS_FBANK_FN
#undef WS_FBANK_ATTR

// ------------------------------------------------------------------ any power-of-two FFT length
// The kernel above is specialised for the 512-point transform of 16 kHz-class rates (25 ms = 400 samples).  The
// reference takes any rate: the CLI passes `sample_frequency=sample_rate` through (wespeaker/cli/speaker.py:90-97;
// the SRE recipe extracts at 8 kHz, examples/sre/v2/conf/resnet.yaml:31) and its native twin sizes the transform from
// the frame length (runtime/core/frontend/fbank.h:33-52, fft_points_ = UpperPowerOfTwo(frame_length_)).  This kernel
// is the same pipeline for fft_n = 16 .. 4096 (200 samples -> 256 points at 8 kHz, 800 -> 1024 at 32 kHz, 1102 / 1200
// -> 2048 at 44.1 / 48 kHz): one wavefront per frame, the real transform as a complex Stockham FFT of fft_n / 2 points
// in LDS -- radix-4 stages, one closing radix-2 stage when log2(fft_n / 2) is odd, every lane walking the butterflies
// lane, lane + 64, ... of a stage -- the same split / power step, and the mel filters four lanes per bin straight from
// the packed weights (taps beyond a filter's end are skipped, not zero padded).  At fft_n = 512 every value goes
// through the same operations in the same order as in the specialised kernel (ws_debug_fbank_mode(2) routes 16 kHz
// input here: the tests compare the two bit for bit); it is also where a 16 kHz-class frontend with more mel bins than
// the specialised kernel's padded weight table holds ends up.
__attribute__((target("no-packed-fp32-ops"))) __device__ __forceinline__ void fbank_any_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__attribute__((target("no-packed-fp32-ops"))) __device__ __forceinline__ float2 fbank_any_cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__attribute__((target("no-packed-fp32-ops"))) __global__ __launch_bounds__(1024) void fbank_any_kernel(
    const FbankTables tb, const void* __restrict__ wav, int wav_dtype, long long wav_stride, float scale,
    const float* __restrict__ window, int T, long long total_frames, float* __restrict__ feats,
    const int* __restrict__ frames) {
  extern __shared__ __attribute__((aligned(16))) unsigned char any_lds[];
  const int NF = tb.fft_n, CNn = NF >> 1;
  const int waves = blockDim.x >> 6, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // LDS: twiddles [NF] float2 | mel_start, mel_len, mel_off [128] ints each | per wavefront: A [CN] float2, B [CN + 4] float2
  float2* const tw = reinterpret_cast<float2*>(any_lds);
  int* const mel_start_s = reinterpret_cast<int*>(tw + NF);
  int* const mel_len_s = mel_start_s + 128;
  int* const mel_off_s = mel_len_s + 128;
  float2* const bufs = reinterpret_cast<float2*>(mel_off_s + 128);
  float2* const bufA = bufs + (size_t)wave * (2 * CNn + 4);
  float2* const bufB = bufA + CNn;
  for (int i = threadIdx.x; i < NF; i += blockDim.x) tw[i] = reinterpret_cast<const float2*>(tb.twiddle)[i];
  for (int i = threadIdx.x; i < 128; i += blockDim.x) {
    const bool has = i < tb.num_bins;
    mel_start_s[i] = has ? tb.mel_start[i] : 0; mel_len_s[i] = has ? tb.mel_len[i] : 0; mel_off_s[i] = has ? tb.mel_off[i] : 0;
  }
  __syncthreads();
  const int L = tb.frame_len;
  float* const xs = reinterpret_cast<float*>(bufB);      // raw samples (<= NF floats)
  float* const zr = reinterpret_cast<float*>(bufA);      // windowed, zero padded = complex input
  const long long fstride = (long long)gridDim.x * waves;
  for (long long frame = (long long)blockIdx.x * waves + wave; frame < total_frames; frame += fstride) {
    const int b = (int)(frame / T), f = (int)(frame - (long long)b * T);
    float* const frow = feats + frame * tb.num_bins;
    if (frames && f >= frames[b]) {
      for (int i = lane; i < tb.num_bins; i += 64) frow[i] = 0.f;
      continue;
    }
    const long long s0 = (long long)b * wav_stride + (long long)f * tb.frame_shift;
    // 1. load + DC offset
    float part = 0.f;
    for (int j = lane; j < L; j += 64) {
      float v;
      if (wav_dtype == 0) v = (float)reinterpret_cast<const short*>(wav)[s0 + j];
      else v = reinterpret_cast<const float*>(wav)[s0 + j];
      v *= scale;
      xs[j] = v;
      part += v;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
    const float mean = part / (float)L;
    fbank_any_wave_sync();
    // 2. pre-emphasis (replicate-pad first sample) + window, zero pad to NF
    for (int j = lane; j < NF; j += 64) {
      float y = 0.f;
      if (j < L) {
        const float cur = xs[j] - mean;
        const float prev = xs[j > 0 ? j - 1 : 0] - mean;
        y = (cur - 0.97f * prev) * window[j];
      }
      zr[j] = y;
    }
    fbank_any_wave_sync();
    // 3. CN-point complex FFT, Stockham autosort: radix-4 stages, then one radix-2 stage if a factor 2 is left
    float2* src = bufA;
    float2* dst = bufB;
    int Ns = 1;
    for (; Ns * 4 <= CNn; Ns *= 4) {
      const int nb = CNn >> 2;
      for (int j = lane; j < nb; j += 64) {
        const int k = j & (Ns - 1);
        float2 v0 = src[j], v1 = src[j + nb], v2 = src[j + 2 * nb], v3 = src[j + 3 * nb];
        if (Ns > 1) {
          const int step = (NF / (Ns * 4)) * k;
          v1 = fbank_any_cmul(v1, tw[step]);
          v2 = fbank_any_cmul(v2, tw[2 * step]);
          v3 = fbank_any_cmul(v3, tw[3 * step]);
        }
        const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y);
        const float2 a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
        const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y);
        const float2 a3 = make_float2(v1.y - v3.y, v3.x - v1.x);   // -i * (v1 - v3)
        const int d0 = ((j - k) << 2) + k;
        dst[d0] = make_float2(a0.x + a2.x, a0.y + a2.y);
        dst[d0 + Ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
        dst[d0 + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
        dst[d0 + 3 * Ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
      }
      fbank_any_wave_sync();
      float2* t = src; src = dst; dst = t;
    }
    if (Ns < CNn) {                                   // Ns * 2 == CN
      const int nb = CNn >> 1;
      for (int j = lane; j < nb; j += 64) {
        const int k = j & (Ns - 1);
        const float2 v0 = src[j];
        const float2 v1 = fbank_any_cmul(src[j + nb], tw[(NF / (Ns * 2)) * k]);
        const int d0 = ((j - k) << 1) + k;
        dst[d0] = make_float2(v0.x + v1.x, v0.y + v1.y);
        dst[d0 + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
      }
      fbank_any_wave_sync();
      float2* t = src; src = dst; dst = t;
    }
    // 4. unpack to the real-input spectrum, power: P[k], k = 0..CN
    float* P = reinterpret_cast<float*>(dst);
    for (int k = lane; k <= CNn; k += 64) {
      const float2 zk = src[k & (CNn - 1)];
      const float2 zc = src[(CNn - k) & (CNn - 1)];
      const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);
      const float orr = 0.5f * (zk.y + zc.y), oi = -0.5f * (zk.x - zc.x);
      const float2 w = tw[k & (NF - 1)];
      const float xr = er + (orr * w.x - oi * w.y);
      const float xi = ei + (orr * w.y + oi * w.x);
      P[k] = xr * xr + xi * xi;
    }
    fbank_any_wave_sync();
    // 5. mel filterbank + log: four lanes per bin, lane sub takes taps sub, sub + 4, ...
    for (int b0 = 0; b0 < tb.num_bins; b0 += 16) {
      const int bin = b0 + (lane >> 2), sub = lane & 3;
      const int len = mel_len_s[bin];                 // (0 beyond num_bins)
      const float* pp = P + mel_start_s[bin];
      const float* wp = tb.mel_w + mel_off_s[bin];
      float acc = 0.f;
      for (int i = sub; i < len; i += 4) acc += wp[i] * pp[i];
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      const float v = logf(fmaxf(acc, 1.1920928955078125e-07f));
      if (bin < tb.num_bins && sub == 0) frow[bin] = v;
    }
    fbank_any_wave_sync();
  }
}

// ws_debug_fbank_mode: 0 shipped kernel, 1 the packed-fp32 build, 2 the any-length kernel for every frontend (tests and tools/fbank_race_probe.py only)
static std::atomic<int> g_fbank_mode{0};
void set_fbank_debug_mode(int mode) { g_fbank_mode.store(mode, std::memory_order_relaxed); }

// LDS of the any-length kernel: twiddles, the three bin tables, per wavefront the two transform buffers
static size_t fbank_any_lds_bytes(int fft_n, int waves) {
  return (size_t)fft_n * 8 + 3 * 128 * 4 + (size_t)waves * ((size_t)fft_n + 4) * 8;
}
static int fbank_any_waves(int fft_n) {
  int w = 16;
  while (w > 1 && fbank_any_lds_bytes(fft_n, w) > 150 * 1024) --w;
  return w;
}
bool fbank_fast_kernel_fits(const FbankTables& t) {
  return t.fft_n == FFT_N && t.frame_len <= FFT_N && t.mel_w_total <= MEL_W_MAX && t.num_bins <= 128 &&
         t.mel_wpad_total <= MEL_WPAD_MAX && t.mel_pad_reach <= FFT_N;
}

hipError_t launch_fbank(const FbankTables& t, const void* wav, int wav_dtype, int B, int N,
                        int64_t wav_stride, float scale, int window_type, int T, float* feats,
                        hipStream_t stream, const int* frames) {
  if (T <= 0 || B <= 0) return hipSuccess;
  if (t.fft_n < 16 || t.fft_n > 4096 || (t.fft_n & (t.fft_n - 1)) || t.frame_len > t.fft_n || t.num_bins > 128)
    return hipErrorInvalidValue;
  const long long total = (long long)B * T;
  const int cus = current_device_cus();
  const float* window = window_type == 1 ? t.window_povey : t.window_hamming;
  const int mode = g_fbank_mode.load(std::memory_order_relaxed);
  if (mode == 2 || !fbank_fast_kernel_fits(t)) {
    const int waves = fbank_any_waves(t.fft_n);
    const size_t lds_bytes = fbank_any_lds_bytes(t.fft_n, waves);
    static size_t lds_granted[WS_MAX_DEVICES] = {};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(fbank_any_kernel), lds_bytes, lds_granted);
    if (e != hipSuccess) return e;
    long long blocks = (total + waves - 1) / waves;
    const long long per_cu = lds_bytes > 80 * 1024 ? 1 : (lds_bytes > 40 * 1024 ? 2 : 4);
    if (blocks > cus * per_cu) blocks = cus * per_cu;
    hipLaunchKernelGGL(fbank_any_kernel, dim3((unsigned)blocks), dim3(64 * waves), lds_bytes, stream, t, wav, wav_dtype,
                       (long long)wav_stride, scale, window, T, total, feats, frames);
    return hipGetLastError();
  }
  long long blocks = (total + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
  const long long resident = (long long)cus * FBANK_WGS_PER_CU;
  if (blocks > resident) blocks = resident;
  if (mode == 1)
    hipLaunchKernelGGL(fbank_kernel_packed, dim3((unsigned)blocks), dim3(64 * FRAMES_PER_BLOCK), 0, stream, t, wav,
                       wav_dtype, N, (long long)wav_stride, scale, window, T, total, feats, frames);
  else
    hipLaunchKernelGGL(fbank_kernel, dim3((unsigned)blocks), dim3(64 * FRAMES_PER_BLOCK), 0, stream, t, wav,
                       wav_dtype, N, (long long)wav_stride, scale, window, T, total, feats, frames);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ CMVN
// apply_cmvn (dataset/dataset_utils.py:19-26; the mean-only form is cli/speaker.py:98-99):
//   mode & 1 (norm_mean): feats[b, t, :] -= mean_t feats[b, :, :]
//   mode & 2 (norm_var):  feats[b, t, :] /= sqrt(var_t feats[b, :, :] + 1e-7), var = torch.var's unbiased estimate
//                         (sum of squared deviations / (T - 1): T = 1 gives 0 / 0 = NaN there and here)
// grid = B, block = 1024: one workgroup per utterance (the statistics need all of its frames, the update is in
// place), so its speed is the number of loads it keeps in flight -- 1024 / F row groups, four independent partial sums
// each (a 256-thread block with two took 74 us for 64 x 8 s utterances: 133 dependent round trips per thread).
constexpr int CMN_THREADS = 1024;
__global__ __launch_bounds__(CMN_THREADS) void cmn_kernel(float* __restrict__ feats, int T, int F,
                                                          const int* __restrict__ lens, int mode) {
  extern __shared__ float sm[];      // [groups][F], then [F] means (+ [F] divisors behind them with norm_var)
  const int b = blockIdx.x, tid = threadIdx.x;
  const int groups = CMN_THREADS / F > 0 ? CMN_THREADS / F : 1;
  float* base = feats + (long long)b * T * F;
  if (lens) T = lens[b];             // ragged batch: statistics over (and applied to) the valid frames only
  const int col = tid % F, grp = tid / F;
  if (grp < groups) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int t = grp;
    for (; t + 3 * groups < T; t += 4 * groups) {
      s0 += base[(long long)t * F + col];
      s1 += base[(long long)(t + groups) * F + col];
      s2 += base[(long long)(t + 2 * groups) * F + col];
      s3 += base[(long long)(t + 3 * groups) * F + col];
    }
    for (; t < T; t += groups) s0 += base[(long long)t * F + col];
    sm[grp * F + col] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  float mean = 0.f;
  if (tid < F) {
    float s = 0.f;
    for (int g = 0; g < groups; ++g) s += sm[g * F + tid];
    mean = s / (float)T;
  }
  __syncthreads();
  if (tid < F) sm[tid] = mean;
  __syncthreads();
  float* sdiv = sm + (size_t)groups * F;          // [F] divisors (norm_var only; the launch sizes the LDS for it)
  if (mode & 2) {
    // second pass over the (cache-resident) utterance: squared deviations from the mean just formed -- the reference
    // takes torch.var of the already mean-subtracted features, i.e. the two-pass form, not E[x^2] - E[x]^2
    float q0 = 0.f, q1 = 0.f;
    if (grp < groups) {
      const float m = sm[col];
      int t = grp;
      for (; t + groups < T; t += 2 * groups) {
        const float d0 = base[(long long)t * F + col] - m, d1 = base[(long long)(t + groups) * F + col] - m;
        q0 += d0 * d0; q1 += d1 * d1;
      }
      for (; t < T; t += groups) { const float d0 = base[(long long)t * F + col] - m; q0 += d0 * d0; }
    }
    __syncthreads();                               // (the means are in registers `m`; sm[F ..) is reused below)
    const float keep = tid < F ? sm[tid] : 0.f;
    __syncthreads();
    if (grp < groups) sm[grp * F + col] = q0 + q1;
    __syncthreads();
    float dv = 0.f;
    if (tid < F) {
      float s = 0.f;
      for (int g = 0; g < groups; ++g) s += sm[g * F + tid];
      dv = sqrtf(s / (float)(T - 1) + 1e-7f);
    }
    __syncthreads();
    if (tid < F) { sm[tid] = keep; sdiv[tid] = dv; }
    __syncthreads();
  }
  if (!(mode & 1)) {                               // norm_mean = False: nothing is subtracted
    __syncthreads();
    if (tid < F) sm[tid] = 0.f;
    __syncthreads();
  }
  const long long total = (long long)T * F;
  if (mode & 2) {
    for (long long i = tid; i < total; i += CMN_THREADS) {
      const int c = (int)(i % F);
      base[i] = (base[i] - sm[c]) / sdiv[c];
    }
  } else if ((F & 3) == 0 && (reinterpret_cast<unsigned long long>(base) & 15) == 0) {   // 16-B read-modify-write;
                                     // F % 4 == 0 keeps a lane inside one row
    typedef float f32x4e __attribute__((ext_vector_type(4)));
    f32x4e* b4 = reinterpret_cast<f32x4e*>(base);
    for (long long i = tid; i < total / 4; i += CMN_THREADS) {
      const int c = (int)((i * 4) % F);
      f32x4e v = b4[i];
      v[0] -= sm[c]; v[1] -= sm[c + 1]; v[2] -= sm[c + 2]; v[3] -= sm[c + 3];
      b4[i] = v;
    }
  } else {
    for (long long i = tid; i < total; i += CMN_THREADS) base[i] -= sm[(int)(i % F)];
  }
}

hipError_t launch_cmn(float* feats, int B, int T, int F, hipStream_t stream, const int* lens, int mode) {
  if (F > CMN_THREADS || F <= 0 || mode < 0 || mode > 3) return hipErrorInvalidValue;
  if (mode == 0) return hipSuccess;                // cmvn: False
  const int groups = CMN_THREADS / F;
  hipLaunchKernelGGL(cmn_kernel, dim3(B), dim3(CMN_THREADS), ((size_t)groups * F + F) * sizeof(float), stream,
                     feats, T, F, lens, mode);
  return hipGetLastError();
}

// dst[b][t][:] = t < lens[b] ? src[b][t][:] : 0   (ragged batch: canonical zero padding of the features)
__global__ __launch_bounds__(256) void copy_rows_masked_kernel(const float* __restrict__ src,
                                                               float* __restrict__ dst, int T, int F4,
                                                               const int* __restrict__ lens,
                                                               long long total4) {
  typedef float f32x4e __attribute__((ext_vector_type(4)));
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const long long row = i / F4;
    const int b = (int)(row / T), t = (int)(row - (long long)b * T);
    f32x4e v = {0.f, 0.f, 0.f, 0.f};
    if (t < lens[b]) v = reinterpret_cast<const f32x4e*>(src)[i];
    reinterpret_cast<f32x4e*>(dst)[i] = v;
  }
}

hipError_t launch_copy_rows_masked(const float* src, float* dst, int B, int T, int F, const int* lens,
                                   hipStream_t stream) {
  if (F & 3) return hipErrorInvalidValue;
  const long long total4 = (long long)B * T * (F / 4);
  if (total4 <= 0) return hipSuccess;
  long long blocks = (total4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(copy_rows_masked_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, T,
                     F / 4, lens, total4);
  return hipGetLastError();
}

// ------------------------------------------------------------------ polyphase sinc resampling
// torchaudio.transforms.Resample as the reference calls it (cli/speaker.py:157-160): output sample
// n = q * new + ph is  sum_k kernel[ph][k] * x[q * orig + k - width]  (zero outside the signal).
// kernel: float32 [new][taps], taps = 2 * width + orig.  One thread per output sample.
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, long long n_in,
                                                       const float* __restrict__ kern, int orig,
                                                       int nw, int width, int taps,
                                                       float* __restrict__ y, long long n_out) {
  const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= n_out) return;
  const long long q = n / nw;
  const int ph = (int)(n - q * nw);
  const float* kr = kern + (long long)ph * taps;
  const long long base = q * orig - width;
  float acc = 0.f;
  for (int k = 0; k < taps; ++k) {
    const long long i = base + k;
    if (i >= 0 && i < n_in) acc += kr[k] * x[i];
  }
  y[n] = acc;
}

hipError_t launch_resample(const float* x, long long n_in, const float* kern, int orig, int nw, int width,
                           float* y, long long n_out, hipStream_t stream) {
  if (n_out <= 0) return hipSuccess;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, stream, x, n_in,
                     kern, orig, nw, width, 2 * width + orig, y, n_out);
  return hipGetLastError();
}

// ------------------------------------------------------------------ binary16 im2col (layer 1)
// one thread per 8 output halfs (16-B store); F % 4 == 0 and ld % 8 == 0
__global__ __launch_bounds__(256) void im2col_f16_kernel(const float* __restrict__ feats, int T, int F,
                                                         int taps, int pad, uint16_t* __restrict__ out,
                                                         int ld, long long total8) {
  typedef _Float16 f16x8e __attribute__((ext_vector_type(8)));
  const int c8n = ld >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8;
       i += (long long)gridDim.x * 256) {
    const long long m = i / c8n;
    const int k0 = (int)(i - m * c8n) * 8;
    const int b = (int)(m / T), t = (int)(m - (long long)b * T);
    f16x8e o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                   // two float4 halves; a half never straddles a tap
      const int k = k0 + 4 * h;
      const int tap = k / F, f = k - tap * F;
      const int ts = t + tap - pad;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tap < taps && ts >= 0 && ts < T)
        v = *reinterpret_cast<const float4*>(feats + ((long long)b * T + ts) * F + f);
      o[4 * h + 0] = (_Float16)v.x; o[4 * h + 1] = (_Float16)v.y;
      o[4 * h + 2] = (_Float16)v.z; o[4 * h + 3] = (_Float16)v.w;
    }
    *reinterpret_cast<f16x8e*>(out + m * ld + k0) = o;
  }
}

hipError_t launch_im2col_f16(const float* feats, int B, int T, int F, int taps, int pad, uint16_t* out,
                             int ld, hipStream_t stream) {
  if ((F & 3) || (ld & 7) || ld < taps * F) return hipErrorInvalidValue;
  const long long total8 = (long long)B * T * (ld >> 3);
  if (total8 <= 0) return hipSuccess;
  long long blocks = (total8 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(im2col_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, feats, T, F, taps,
                     pad, out, ld, total8);
  return hipGetLastError();
}

// fp32 im2col of the same conv (parity-grade back-end): one thread per float4; a float4 never straddles a tap
// (F % 4 == 0); columns [taps * F, ld) are zeros
__global__ __launch_bounds__(256) void im2col_f32_kernel(const float* __restrict__ feats, int T, int F, int taps,
                                                         int pad, float* __restrict__ out, int ld, long long total4) {
  const int c4n = ld >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const long long m = i / c4n;
    const int k = (int)(i - m * c4n) * 4;
    const int b = (int)(m / T), t = (int)(m - (long long)b * T);
    const int tap = k / F, f = k - tap * F;
    const int ts = t + tap - pad;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tap < taps && ts >= 0 && ts < T) v = *reinterpret_cast<const float4*>(feats + ((long long)b * T + ts) * F + f);
    *reinterpret_cast<float4*>(out + m * ld + k) = v;
  }
}

hipError_t launch_im2col_f32(const float* feats, int B, int T, int F, int taps, int pad, float* out, int ld,
                             hipStream_t stream) {
  if ((F & 3) || (ld & 3) || ld < taps * F) return hipErrorInvalidValue;
  const long long total4 = (long long)B * T * (ld >> 2);
  if (total4 <= 0) return hipSuccess;
  long long blocks = (total4 + 255) / 256;
  if (blocks > 32768) blocks = 32768;
  hipLaunchKernelGGL(im2col_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, feats, T, F, taps, pad, out,
                     ld, total4);
  return hipGetLastError();
}

// ------------------------------------------------------------------ chunk-and-average mode
// SpeakerEngine::ExtractFeature, chunk-by-chunk branch (runtime/core/speaker/speaker_engine.cc:96-131):
// the utterance's frames [total][F] are cut into consecutive chunks of `cf` frames; a trailing
// partial chunk is completed with the HEAD frames of the first chunk, and an utterance shorter than
// one chunk is tiled cyclically.  dst [n_chunks][cf][F].  One float4 per thread.
__global__ __launch_bounds__(256) void chunk_gather_kernel(
    const float* __restrict__ feats, int total, int F, int cf, int n_full, int n_chunks,
    float* __restrict__ dst) {
  const int f4 = F >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n_chunks * cf * f4) return;
  const int c4 = (int)(i % f4);
  const long long rowi = i / f4;
  const int fr = (int)(rowi % cf), c = (int)(rowi / cf);
  int src;
  if (c < n_full) {
    src = c * cf + fr;
  } else {
    const int last = total - n_full * cf;
    if (n_full == 0) src = fr % last;
    else src = fr < last ? n_full * cf + fr : fr - last;
  }
  reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(feats)[(long long)src * f4 + c4];
}

hipError_t launch_chunk_gather(const float* feats, int total, int F, int cf, int n_full, int n_chunks,
                               float* dst, hipStream_t stream) {
  if (F & 3) return hipErrorInvalidValue;
  const long long n = (long long)n_chunks * cf * (F >> 2);
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(chunk_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     feats, total, F, cf, n_full, n_chunks, dst);
  return hipGetLastError();
}

// ------------------------------------------------------------------ diarization sub-segment windows
// subsegment() of wespeaker/diar/extract_emb.py:55-83 on the device: window w covers frames [w * period,
// min(w * period + window, seg_length)) of the segment's fbank -- clipped to the num_frames rows that exist, like the
// numpy slice -- and is completed to `window` rows by tiling ITS OWN rows cyclically (np.resize of the slice); a
// segment of seg_length <= window is ONE window tiled from the whole fbank.  A window without any row is zeros
// (np.resize of an empty array).  dst [n_windows][window][F], one float4 per thread.
__global__ __launch_bounds__(256) void window_gather_kernel(const float* __restrict__ feats, int num_frames, int F,
                                                            int window, int period, int seg_length, int n_windows,
                                                            float* __restrict__ dst) {
  const int f4 = F >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n_windows * window * f4) return;
  const int c4 = (int)(i % f4);
  const long long rowi = i / f4;
  const int r = (int)(rowi % window), w = (int)(rowi / window);
  int begin = 0, len = num_frames;
  if (seg_length > window) {
    begin = w * period;
    int end = begin + window < seg_length ? begin + window : seg_length;
    if (end > num_frames) end = num_frames;
    len = end - begin;
  }
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (len > 0) v = reinterpret_cast<const float4*>(feats)[(long long)(begin + r % len) * f4 + c4];
  reinterpret_cast<float4*>(dst)[i] = v;
}

hipError_t launch_window_gather(const float* feats, int num_frames, int F, int window, int period, int seg_length,
                                int n_windows, float* dst, hipStream_t stream) {
  if ((F & 3) || window <= 0 || period <= 0) return hipErrorInvalidValue;
  const long long n = (long long)n_windows * window * (F >> 2);
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(window_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, feats, num_frames,
                     F, window, period, seg_length, n_windows, dst);
  return hipGetLastError();
}

// avg[e] = (1/n) sum_c emb[c][e]   (speaker_engine.cc:147-158; summed in chunk order like the loop)
__global__ __launch_bounds__(256) void chunk_average_kernel(const float* __restrict__ emb, int n,
                                                            int E, float* __restrict__ avg) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  float s = 0.f;
  for (int c = 0; c < n; ++c) s += emb[(long long)c * E + e];
  avg[e] = s / (float)n;
}

// Range guard of the binary16 back-ends: counts non-finite outputs into a (host-mapped) counter.  The
// counter is only touched when something IS non-finite, so the normal path costs one tiny launch.
__global__ __launch_bounds__(256) void count_nonfinite_kernel(const float* __restrict__ x, long long n,
                                                              int* __restrict__ counter) {
  int bad = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = x[i];
    bad += !(fabsf(v) <= 3.402823466e38f);          // inf or NaN
  }
  if (bad) atomicAdd_system(counter, bad);
}

hipError_t launch_count_nonfinite(const float* x, long long n, int* counter, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  long long blocks = (n + 255) / 256;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, counter);
  return hipGetLastError();
}

hipError_t launch_chunk_average(const float* emb, int n, int E, float* avg, hipStream_t stream) {
  hipLaunchKernelGGL(chunk_average_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, emb, n, E, avg);
  return hipGetLastError();
}

}  // namespace wsamd
