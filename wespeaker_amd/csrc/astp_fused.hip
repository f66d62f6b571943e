// Attentive statistics pooling as ONE kernel (parity-grade fp32 back-end, T <= 208 frames per utterance).
//
// Replaces, for one utterance per workgroup, the chain of wespeaker/models/pooling_layers.py:119-144 (ASTP.forward):
//     alpha = tanh(linear1(x_in))                 (:136)   1536 (+ context columns, folded to a bias) -> 128
//     alpha = softmax(linear2(alpha), dim = 2)    (:137)   128 -> 1536, softmax over time
//     mean  = sum(alpha * x), var = sum(alpha * x^2) - mean^2, std = sqrt(var.clamp(min = 1e-7))   (:138-141)
//     return cat([mean, std])                     (:142)
// which the tile kernels ran as three launches (linear1 on 128x128 tiles: N = 128 leaves 198 row tiles for 256 CUs;
// linear2 with a softmax-partials epilogue: K = 128 is four K-tiles per tile, everything else epilogue; the merge
// of the partials).  DESIGN.md 4.2.6.
//
// Shape of the work.  A 2-s utterance is 198 frames: 12.4 row blocks of v_mfma_f32_16x16x4_f32.  Each of the
// eight wavefronts owns a 16-column slice of the output for ALL 13 row blocks (13 accumulators of 4 VGPRs), so
//   * every wavefront has the same work whatever T is (198 / 208 of the MFMAs are useful; the 32x32x2 shapes
//     of the GEMM kernels would need 7 x 32 = 224 rows and cannot be split evenly over four SIMDs),
//   * 256 utterances are exactly one round over 256 CUs,
//   * the softmax over time of a column lives in ONE wavefront: 13 x 4 registers x 4 lane groups.
// Phase A: H[208][128] = tanh(X[208][1536] W1^T + bias_u): X and W1 stream through a three-stage ring of 32-deep
//   K-tiles (LDS-DMA, 128-B rows, 16-B chunk index XOR (row & 7) on the source side and on the fragment reads),
//   two K-tiles in flight.  One ds_read_b128 = the operands of four MFMAs (k = 16 j + 4 q + s for lane group q:
//   any assignment of k to (step, lane group) is a valid GEMM as long as both operands use the same one).
// Phase B: H (106 KB) lies in LDS where the ring was.  For each of its twelve 16-column slices a wavefront computes
//   the logits L[208][16] = H W2[slice]^T (K = 128; the W2 rows come straight from L2 into registers, one slice
//   ahead), takes the column maxima over the live frames, and folds exp(L - max), x and x^2 into the three sums;
//   x is re-read from L2 / Infinity Cache (the same rows that phase A just streamed), requested at the start of
//   the slice's MFMAs.  The logits never exist in memory.  linear2's bias is constant over time and cancels in the
//   softmax, as in the tile kernel's pooling epilogue (conv_gemm.hip).
#include "kernels.h"

#include <cstdio>
#include <cstdlib>

namespace wsamd {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int F_RB = 13;                       // row blocks of 16 frames
constexpr int F_ROWS = 16 * F_RB;              // 208
constexpr int F_NH = 128;                      // bottleneck width
constexpr int F_BK = 32;
constexpr int F_APIECES = F_ROWS / 8;          // 26 pieces of 8 rows x 128 B
constexpr int F_WPIECES = F_NH / 8;            // 16
constexpr int F_PIECES = F_APIECES + F_WPIECES;   // 42
constexpr int F_NP = 6;                        // per wavefront (8 x 6 = 48 slots, the last 6 are copies)
constexpr int F_STAGE_BYTES = 48 * 1024;
constexpr int F_W_BYTE0 = F_ROWS * 128;
constexpr int F_NSTAGE = 3;
constexpr int F_LDS_BYTES = F_NSTAGE * F_STAGE_BYTES;      // 144 KB; H (208 x 512 B) aliases stages 0..2
constexpr int F_NC = 1536;                     // pooled channels
constexpr int F_SLICES = F_NC / 16 / 8;        // 12 per wavefront

__device__ __forceinline__ void f_dma_16B(const void* g, void* lds_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
#endif
}
template <int N>
__device__ __forceinline__ void f_wait_vm_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ f32x4 f_mfma(float a, float b, f32x4 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#else
  return c;
#endif
}

// tanh without branches: e = exp(-2|x|) in (0, 1], tanh|x| = (1 - e) / (1 + e); absolute error ~1e-7
__device__ __forceinline__ float f_tanh(float x) {
  float e = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
  e = __builtin_amdgcn_exp2f(-2.f * 1.4426950408889634f * __builtin_fabsf(x));
#endif
  const float t = __fdividef(1.f - e, 1.f + e);
  return __builtin_copysignf(t, x);
}

struct AstpFusedParams {
  const float* h; int ldh;          // [B*T][ldh] (1536 channels used)
  const float* w1; int ldw1;        // [128][ldw1], columns 0..1535 multiply h
  const float* bias;                // [128] or null
  const float* bias_img;            // [B][128] or null (context columns folded per utterance)
  const float* w2; int ldw2;        // [1536][ldw2] (K = 128)
  float* pooled;                    // [B][3072]
  const int* row_len;               // optional [B]
  int B, T;
};

__global__ __launch_bounds__(512, 2) void astp_fused_kernel(const AstpFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;
  const int r8 = lane >> 3, c8 = lane & 7;
  const int nk = F_NC / F_BK;                  // 48
  const int T = p.T;

  // fragment byte offsets inside a stage (phase A) and inside H (phase B)
  int offA[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) offA[jj] = r16 * 128 + (((4 * jj + q4) ^ (r16 & 7)) * 16);
  const int offW = F_W_BYTE0 + wave * 16 * 128;
  int offH[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) offH[j] = r16 * 512 + (((4 * j + q4) ^ (r16 & 7)) * 16);

  {
    // one utterance per workgroup (nothing carries over between utterances, so there is nothing to gain from a
    // persistent loop -- and the compiler hoists ~180 loop-invariant lane addresses out of one and spills them)
    const int u = blockIdx.x;
    const int len = p.row_len ? min(p.row_len[u], T) : T;
    const float* hu = p.h + (long long)u * T * p.ldh;

    // ---- DMA lane offsets of this wavefront's pieces (bytes from hu / w1, K offset added per K-tile)
    unsigned voff[F_NP];
#pragma unroll
    for (int i = 0; i < F_NP; ++i) {
      int q = wave * F_NP + i;
      q = q < F_PIECES ? q : F_PIECES - 1;          // slots 42..47: copies of the last piece (uniform vmcnt)
      const bool isw = q >= F_APIECES;
      const int row = (isw ? q - F_APIECES : q) * 8 + r8;
      const int c = c8 ^ (row & 7);
      const int grow = isw ? row : (row < T ? row : T - 1);
      voff[i] = (unsigned)((grow * (isw ? p.ldw1 : p.ldh)) * 4 + c * 16);
    }
    auto dma_ktile = [&](int kt, int stage) {
#pragma unroll
      for (int i = 0; i < F_NP; ++i) {
        const int q = wave * F_NP + i;
        const bool isw = (q < F_PIECES ? q : F_PIECES - 1) >= F_APIECES;       // wave-uniform
        const char* gb = isw ? reinterpret_cast<const char*>(p.w1) : reinterpret_cast<const char*>(hu);
        f_dma_16B(gb + (size_t)(unsigned)(kt * (F_BK * 4)) + voff[i], ldsb + stage * F_STAGE_BYTES + q * 1024);
      }
    };

    dma_ktile(0, 0);
    dma_ktile(1, 1);

    f32x4 acc[F_RB];
#pragma unroll
    for (int b = 0; b < F_RB; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---------------------------------------------------------------- phase A: H = tanh(X W1^T + bias)
    int stage = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) f_wait_vm_barrier<F_NP>(); else f_wait_vm_barrier<0>();
      if (kt + 2 < nk) dma_ktile(kt + 2, stage >= 1 ? stage - 1 : 2);
      const char* sb = ldsb + stage * F_STAGE_BYTES;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const f32x4 fb = *reinterpret_cast<const f32x4*>(sb + offW + offA[jj]);
#pragma unroll
        for (int b = 0; b < F_RB; ++b) {
          const f32x4 fa = *reinterpret_cast<const f32x4*>(sb + b * 2048 + offA[jj]);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[b] = f_mfma(fa[s], fb[s], acc[b]);
        }
      }
      stage = stage == F_NSTAGE - 1 ? 0 : stage + 1;
    }
    // every wavefront has read its last fragments, no DMA is in flight: H may overwrite the ring
    f_wait_vm_barrier<0>();
    {
      const int n = wave * 16 + r16;
      const float bv = p.bias_img ? p.bias_img[(long long)u * F_NH + n] : (p.bias ? p.bias[n] : 0.f);
      const int chunk = 4 * wave + (r16 >> 2);
#pragma unroll
      for (int b = 0; b < F_RB; ++b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * b + 4 * q4 + r;
          const float v = f_tanh(acc[b][r] + bv);
          *reinterpret_cast<float*>(ldsb + row * 512 + ((chunk ^ (row & 7)) * 16) + (r16 & 3) * 4) = v;
        }
      }
    }
    __syncthreads();

    // ---------------------------------------------------------------- phase B: logits -> softmax -> weighted sums
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hu), 0, 0xffffffff, 0x00020000);
    // lane offsets of the x rows of block 0 (row 4 q + r, column r16); rows past the utterance read its last frame
    int xoff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) xoff[r] = ((4 * q4 + r) * p.ldh + r16) * 4;
    const int last_off = ((T - 1) * p.ldh + r16) * 4;
    const int blk_bytes = 16 * p.ldh * 4;

    auto load_w2 = [&](int sl, f32x4 (&fw)[8]) {
      const float* wrow = p.w2 + (long long)(16 * sl + r16) * p.ldw2 + 4 * q4;
#pragma unroll
      for (int j = 0; j < 8; ++j) fw[j] = *reinterpret_cast<const f32x4*>(wrow + 16 * j);
    };
    auto read_h = [&](int step, f32x4 (&f)[4]) {          // step = 2 b + (half of K)
      const char* base = ldsb + (step >> 1) * 8192;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) f[jj] = *reinterpret_cast<const f32x4*>(base + offH[4 * (step & 1) + jj]);
    };
    f32x4 fw[2][8];
    load_w2(wave, fw[0]);
#pragma unroll 1
    for (int it = 0; it < F_SLICES; it += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int sl = (it + half) * 8 + wave;
        const int c0 = 16 * sl;
        if (it + half + 1 < F_SLICES) load_w2(sl + 8, fw[half ^ 1]);
        // (the 52 lane offsets and 52 row masks of a slice are two instructions each; opaque copies of their
        // inputs keep the compiler from hoisting all of them out of the slice loop and spilling them)
        int xo[4], lo = last_off, lim = len - 4 * q4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          xo[r] = xoff[r];
          asm volatile("" : "+v"(xo[r]));
        }
        asm volatile("" : "+v"(lo));
        asm volatile("" : "+v"(lim));
        // x of this slice: requested now, used after the MFMAs
        float xv[F_RB][4];
#pragma unroll
        for (int b = 0; b < F_RB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int vo = min(xo[r] + b * blk_bytes, lo);
            xv[b][r] = __int_as_float((int)__builtin_amdgcn_raw_buffer_load_b32(x_rsrc, vo, c0 * 4, 0));
          }
        f32x4 lg[F_RB];
        f32x4 fa[2][4];
        read_h(0, fa[0]);
#pragma unroll
        for (int step = 0; step < 2 * F_RB; ++step) {
          const int b = step >> 1, jh = step & 1;
          if (step + 1 < 2 * F_RB) read_h(step + 1, fa[(step + 1) & 1]);
          if (jh == 0) lg[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int s = 0; s < 4; ++s) lg[b] = f_mfma(fa[step & 1][jj][s], fw[half][4 * jh + jj][s], lg[b]);
#if defined(__HIP_DEVICE_COMPILE__)
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
        // softmax over the live frames of column c0 + r16 (exp2 domain, the max folded into one fma)
        constexpr float LOG2E = 1.4426950408889634f;
        float mx = -1e30f;
#pragma unroll
        for (int b = 0; b < F_RB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            mx = 16 * b + r < lim ? fmaxf(mx, lg[b][r]) : mx;
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float ml = mx * LOG2E;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int b = 0; b < F_RB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float e = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
            e = __builtin_amdgcn_exp2f(__builtin_fmaf(lg[b][r], LOG2E, -ml));
#endif
            e = 16 * b + r < lim ? e : 0.f;
            const float x = xv[b][r];
            const float ex = e * x;
            s0 += e;
            s1 += ex;
            s2 = __builtin_fmaf(ex, x, s2);
          }
        s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (q4 == 0) {
          const float inv = 1.f / s0;
          const float mean = s1 * inv;
          const float var = s2 * inv - mean * mean;
          float* out = p.pooled + (long long)u * 2 * F_NC + c0 + r16;
          out[0] = mean;
          out[F_NC] = sqrtf(fmaxf(var, 1e-7f));
        }
      }
    }
  }
}

}  // namespace

bool astp_fused_supported(int T, int C, int bottleneck) {
  static const int off = [] { const char* e = getenv("WS_ASTP_FUSED"); return e && atoi(e) == 0 ? 1 : 0; }();
  // all 13 row blocks are computed whatever T is: below ~10 blocks the tile kernels are the faster path
  return !off && C == F_NC && bottleneck == F_NH && T > 160 && T <= F_ROWS;
}

hipError_t launch_astp_fused(const float* h, int ldh, int B, int T, const float* w1, int ldw1, const float* bias,
                             const float* bias_img, const float* w2, int ldw2, float* pooled, const int* lens,
                             hipStream_t stream) {
  if (!astp_fused_supported(T, F_NC, F_NH) || (ldh & 3) || (ldw1 & 3) || (ldw2 & 3) || B <= 0)
    return hipErrorInvalidValue;
  // 32-bit byte offsets inside one utterance / the weight matrices
  if ((long long)T * ldh * 4 >= (1ll << 31) || (long long)F_NH * ldw1 * 4 >= (1ll << 31)) return hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(astp_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  AstpFusedParams p;
  p.h = h; p.ldh = ldh; p.w1 = w1; p.ldw1 = ldw1; p.bias = bias; p.bias_img = bias_img;
  p.w2 = w2; p.ldw2 = ldw2; p.pooled = pooled; p.row_len = lens; p.B = B; p.T = T;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  }
  (void)cus;
  const int grid = B;
  hipLaunchKernelGGL(astp_fused_kernel, dim3(grid), dim3(512), F_LDS_BYTES, stream, p);
  return hipGetLastError();
}

}  // namespace wsamd
