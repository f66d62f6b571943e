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
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int F_NH = 128;                      // bottleneck width
constexpr int F_BK = 32;
constexpr int F_WPIECES = F_NH / 8;            // 16 pieces of 8 rows x 128 B
constexpr int F_NSTAGE = 3;
// RB row blocks of 16 frames (13: T <= 208, 10: T <= 160, 7: T <= 112): A pieces, pieces per wavefront, stage bytes
constexpr int f_np(int rb) { return (2 * rb + F_WPIECES + 7) / 8; }
constexpr int f_stage_bytes(int rb) { return f_np(rb) * 8 * 1024; }
constexpr int f_lds_bytes(int rb) { return F_NSTAGE * f_stage_bytes(rb); }   // H (16 RB x 512 B) aliases the ring
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

#ifdef WS_TRACE
__device__ unsigned long long g_astp_trace[2 * 64];       // [wavefront 0 | wavefront 4] x 64 slots, workgroup 9
#define WS_ASTAMP(i)                                                                                    \
  { const int _i = (i); if (blockIdx.x == 9 && (threadIdx.x == 0 || threadIdx.x == 256) && _i < 64)     \
      g_astp_trace[_i + (threadIdx.x >> 2)] = __builtin_readcyclecounter(); }
#else
#define WS_ASTAMP(i)
#endif

struct AstpFusedParams {
  const float* h; int ldh;          // [B*T][ldh] (1536 channels used)
  const float* w1; int ldw1;        // [128][ldw1], columns 0..1535 multiply h
  const float* bias;                // [128] or null
  const float* bias_img;            // [B][128] or null (context columns folded per utterance)
  const float* w2; int ldw2;        // [1536][ldw2] (K = 128)
  float* pooled;                    // [B][3072]
  const int* row_len;               // optional [B]
  int B, T;
  // Segments (round 6, utterances longer than 208 frames): workgroup (u, seg) takes frames [seg * seg_rows, + seg_rows)
  // of utterance u and leaves its online-softmax tuple (max, sum e, sum e x, sum e x^2) per channel in
  // partial[(u * nseg + seg) * 1536 + c]; astp_combine_kernel merges an utterance's tuples.  nseg = 1: as before.
  int nseg, seg_rows;
  f32x4* partial;
};

template <bool RAGGED, int F_RB>
__global__ __launch_bounds__(512, 2) void astp_fused_kernel(const AstpFusedParams p) {
  constexpr int F_ROWS = 16 * F_RB;
  constexpr int F_WHOLE = F_RB - 3;              // 16 F_WHOLE <= T: these row blocks lie inside every utterance
  constexpr int F_APIECES = F_ROWS / 8, F_PIECES = F_APIECES + F_WPIECES;
  constexpr int F_NP = f_np(F_RB);               // per wavefront (8 F_NP slots >= F_PIECES, the surplus are copies)
  constexpr int F_STAGE_BYTES = f_stage_bytes(F_RB);
  constexpr int F_W_BYTE0 = F_ROWS * 128;
  constexpr int F_PAIRS = (F_RB + 1) / 2;        // row blocks are processed two at a time (+ a single one if odd)
  static_assert(F_NP <= F_PAIRS, "one DMA piece behind each pair of the first half of a K-tile");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;
  const int r8 = lane >> 3, c8 = lane & 7;
  const int nk = F_NC / F_BK;                  // 48

  // fragment byte offsets inside a stage (phase A) and inside H (phase B)
  int offA[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) offA[jj] = r16 * 128 + (((4 * jj + q4) ^ (r16 & 7)) * 16);
  const int offW = F_W_BYTE0 + wave * 16 * 128;
  // H fragment of (block b, k chunk j = 2 m + jl): the XOR key only reaches the low three chunk bits, so
  // byte = offH[jl] + 128 m + 8192 b: two lane registers, everything else is an immediate of ds_read_b128
  int offH[2];
#pragma unroll
  for (int jl = 0; jl < 2; ++jl) offH[jl] = r16 * 512 + (((4 * jl + q4) ^ (r16 & 7)) * 16);

  {
    // one utterance per workgroup (nothing carries over between utterances, so there is nothing to gain from a
    // persistent loop -- and the compiler hoists ~180 loop-invariant lane addresses out of one and spills them)
    const int nseg = p.nseg;
    const int u = nseg > 1 ? blockIdx.x / nseg : blockIdx.x;
    const int seg = blockIdx.x - u * nseg;
    const int row0 = seg * p.seg_rows;           // (0 without segments)
    const int T = min(p.seg_rows, p.T - row0);   // rows of this workgroup's part of the utterance's slot
    const int len = max(0, min((p.row_len ? min(p.row_len[u], p.T) : p.T) - row0, T));
    const float* hu = p.h + ((long long)u * p.T + row0) * p.ldh;

    // ---- DMA lane offsets of this wavefront's pieces (bytes from hu / w1, K offset added per K-tile)
    unsigned voff[F_NP];
#pragma unroll
    for (int i = 0; i < F_NP; ++i) {
      int q = wave * F_NP + i;
      q = q < F_PIECES ? q : F_PIECES - 1;          // surplus slots: copies of the last piece (uniform vmcnt)
      const bool isw = q >= F_APIECES;
      const int row = (isw ? q - F_APIECES : q) * 8 + r8;
      const int c = c8 ^ (row & 7);
      const int grow = isw ? row : (row < T ? row : T - 1);
      voff[i] = (unsigned)((grow * (isw ? p.ldw1 : p.ldh)) * 4 + c * 16);
    }
    auto dma_piece = [&](int kt, int stage, int i) {
      const int q = wave * F_NP + i;
      const bool isw = (q < F_PIECES ? q : F_PIECES - 1) >= F_APIECES;       // wave-uniform
      const char* gb = isw ? reinterpret_cast<const char*>(p.w1) : reinterpret_cast<const char*>(hu);
      f_dma_16B(gb + (size_t)(unsigned)(kt * (F_BK * 4)) + voff[i], ldsb + stage * F_STAGE_BYTES + q * 1024);
    };
    auto dma_ktile = [&](int kt, int stage) {
#pragma unroll
      for (int i = 0; i < F_NP; ++i) dma_piece(kt, stage, i);
    };

    WS_ASTAMP(0);
    dma_ktile(0, 0);
    dma_ktile(1, 1);

    f32x4 acc[F_RB];
#pragma unroll
    for (int b = 0; b < F_RB; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---------------------------------------------------------------- phase A: H = tanh(X W1^T + bias)
    // While the partner wavefront of the SIMD issues MFMAs back to back, NOTHING of this wavefront is issued (measured:
    // 14 ds_reads or 150 VALU instructions stretch to the end of the partner's MFMA burst, whatever s_setprio says), so
    // fragment reads / DMA issue as a separate phase cost their full length.  Hence one instruction stream per
    // wavefront in which every non-MFMA instruction sits behind an MFMA: the MFMAs of K-tile kt run on fragments read
    // during K-tile kt - 1, and the register of a fragment is refilled from stage kt + 1 right behind the MFMAs that
    // consumed it.  The rendezvous at the top of K-tile kt says: K-tile kt + 1 has landed (kt + 2 may be in flight)
    // and nobody reads stage kt any more -- it is refilled with K-tile kt + 3.
    f32x4 fA[2][F_RB], fB[2];
    dma_ktile(2, 2);
    f_wait_vm_barrier<2 * F_NP>();                 // K-tile 0 has landed (1 and 2 in flight)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      fB[jj] = *reinterpret_cast<const f32x4*>(ldsb + offW + offA[jj]);
#pragma unroll
      for (int b = 0; b < F_RB; ++b) fA[jj][b] = *reinterpret_cast<const f32x4*>(ldsb + b * 2048 + offA[jj]);
    }
    int stage = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      if (kt == 10) WS_ASTAMP(5);
      if (kt == 11) WS_ASTAMP(6);
      const int nxt = stage == F_NSTAGE - 1 ? 0 : stage + 1;
      const bool more = kt + 1 < nk;               // (the last K-tile re-reads its own stage: harmless, unused)
      if (kt == 10 || kt == 11) WS_ASTAMP(56 + 2 * (kt - 10));
      if (kt + 2 < nk) f_wait_vm_barrier<F_NP>(); else f_wait_vm_barrier<0>();
      if (kt == 10 || kt == 11) WS_ASTAMP(57 + 2 * (kt - 10));
      const char* nb = ldsb + (more ? nxt : stage) * F_STAGE_BYTES;
      const bool fill = kt + 3 < nk;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
        for (int b = 0; b < F_RB; b += 2) {
          const f32x4 a0 = fA[jj][b], a1 = fA[jj][b + 1 < F_RB ? b + 1 : b], w = fB[jj];
          if (b + 1 < F_RB) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              acc[b] = f_mfma(a0[s], w[s], acc[b]);
              acc[b + 1] = f_mfma(a1[s], w[s], acc[b + 1]);
              if (s == 0) {            // behind the first MFMAs of the pair: one DMA piece / the refills
                const int piece = jj * F_PAIRS + b / 2;
                if (fill && piece < F_NP) dma_piece(kt + 3, stage, piece);
              }
              if (s == 1) fA[jj][b] = *reinterpret_cast<const f32x4*>(nb + b * 2048 + offA[jj]);
              if (s == 2) fA[jj][b + 1] = *reinterpret_cast<const f32x4*>(nb + (b + 1) * 2048 + offA[jj]);
              // the W1 fragment of this half is free behind the last pair's last MFMAs (even block counts)
              if (s == 3 && b + 2 >= F_RB) fB[jj] = *reinterpret_cast<const f32x4*>(nb + offW + offA[jj]);
            }
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[b] = f_mfma(a0[s], w[s], acc[b]);
            {
              const int piece = jj * F_PAIRS + b / 2;     // (7 row blocks: the fourth piece belongs to the single block)
              if (fill && piece < F_NP) dma_piece(kt + 3, stage, piece);
            }
            fA[jj][b] = *reinterpret_cast<const f32x4*>(nb + b * 2048 + offA[jj]);
            fB[jj] = *reinterpret_cast<const f32x4*>(nb + offW + offA[jj]);
          }
#if defined(__HIP_DEVICE_COMPILE__)
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      }
      stage = nxt;
    }
    WS_ASTAMP(1);
    // every wavefront has read its last fragments, no DMA is in flight: H may overwrite the ring
    f_wait_vm_barrier<0>();
    WS_ASTAMP(2);
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
    WS_ASTAMP(3);
    __syncthreads();
    WS_ASTAMP(4);

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
    f32x4 fw[2][8];
    load_w2(wave, fw[0]);
#pragma unroll 1
    for (int it = 0; it < F_SLICES; it += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int sl = (it + half) * 8 + wave;
        const int c0 = 16 * sl;
        if (it + half + 1 < F_SLICES) load_w2(sl + 8, fw[half ^ 1]);
        // x of this slice: requested now, used after the MFMAs.  Row block b is a scalar offset (T > 160: the first
        // ten blocks are whole, only the last three clamp to the utterance's last frame).  Opaque copies of the lane
        // offsets keep the compiler from hoisting the per-slice address arithmetic out of the slice loop (it spills).
        int xo[4], lo = last_off, lim = len - 4 * q4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          xo[r] = xoff[r];
          asm volatile("" : "+v"(xo[r]));
        }
        asm volatile("" : "+v"(lo));
        asm volatile("" : "+v"(lim));
        f32x2 xv[F_RB][2];
#pragma unroll
        for (int b = 0; b < F_RB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v;
            if (b < F_WHOLE)
              v = __int_as_float((int)__builtin_amdgcn_raw_buffer_load_b32(x_rsrc, xo[r], c0 * 4 + b * blk_bytes, 0));
            else
              v = __int_as_float((int)__builtin_amdgcn_raw_buffer_load_b32(
                  x_rsrc, min(xo[r] + b * blk_bytes, lo), c0 * 4, 0));
            xv[b][r >> 1][r & 1] = v;
          }
        WS_ASTAMP(8 + 4 * (it + half));
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_setprio(0);
#endif
        f32x4 lg[F_RB];
#pragma unroll
        for (int b = 0; b < F_RB; ++b) lg[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // k chunk outermost, row blocks in pairs with alternating MFMAs (see phase A); the fragments of the next
        // pair are read before the MFMAs of this one
        f32x4 fa[2][2];
        auto read_pair = [&](int idx, f32x4 (&f)[2]) {       // idx = F_PAIRS j + pair; an odd last block is alone
          const int j = idx / F_PAIRS, pr = idx - F_PAIRS * j;
          f[0] = *reinterpret_cast<const f32x4*>(ldsb + offH[j & 1] + ((2 * pr) * 8192 + (j >> 1) * 128));
          if (2 * pr + 1 < F_RB) f[1] = *reinterpret_cast<const f32x4*>(ldsb + offH[j & 1] + ((2 * pr + 1) * 8192 + (j >> 1) * 128));
        };
        read_pair(0, fa[0]);
#pragma unroll
        for (int idx = 0; idx < 8 * F_PAIRS; ++idx) {
          const int j = idx / F_PAIRS, pr = idx - F_PAIRS * j, b = 2 * pr;
          if (idx + 1 < 8 * F_PAIRS) read_pair(idx + 1, fa[(idx + 1) & 1]);
          if (b + 1 < F_RB) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              lg[b] = f_mfma(fa[idx & 1][0][s], fw[half][j][s], lg[b]);
              lg[b + 1] = f_mfma(fa[idx & 1][1][s], fw[half][j][s], lg[b + 1]);
            }
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              lg[b] = f_mfma(fa[idx & 1][0][s], fw[half][j][s], lg[b]);
            }
          }
#if defined(__HIP_DEVICE_COMPILE__)
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
        WS_ASTAMP(9 + 4 * (it + half));
#if defined(__HIP_DEVICE_COMPILE__)
        // the softmax, the stores and the next slice's load requests win the issue arbitration against the other
        // wavefront's MFMAs (which need one issue slot in 32 cycles): without this they are starved for as long as
        // the other wavefront has MFMAs to issue, and the two wavefronts of a SIMD take turns instead of overlapping
        __builtin_amdgcn_s_setprio(3);
#endif
        // softmax over the live frames of column c0 + r16 (exp2 domain, the max folded into one fma; packed fp32
        // arithmetic on register pairs).  Rows past the utterance only exist in the last three blocks unless the
        // batch is ragged.
        constexpr float LOG2E = 1.4426950408889634f;
        float mx = -1e30f;
#pragma unroll
        for (int b = 0; b < F_RB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (RAGGED || b >= F_WHOLE) mx = 16 * b + r < lim ? fmaxf(mx, lg[b][r]) : mx;
            else mx = fmaxf(mx, lg[b][r]);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float ml = mx * LOG2E;
        WS_ASTAMP(11 + 4 * (it + half));
        f32x2 s0v = {0.f, 0.f}, s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
        const f32x2 mlv = {ml, ml};
#pragma unroll
        for (int b = 0; b < F_RB; ++b)
#pragma unroll
          for (int hp = 0; hp < 2; ++hp) {
            const f32x2 l2 = {lg[b][2 * hp], lg[b][2 * hp + 1]};
            const f32x2 arg = l2 * LOG2E - mlv;
            f32x2 e = {0.f, 0.f};
            (void)arg;
#if defined(__HIP_DEVICE_COMPILE__)
            e[0] = __builtin_amdgcn_exp2f(arg[0]);
            e[1] = __builtin_amdgcn_exp2f(arg[1]);
#endif
            if (RAGGED || b >= F_WHOLE) {
              e[0] = 16 * b + 2 * hp < lim ? e[0] : 0.f;
              e[1] = 16 * b + 2 * hp + 1 < lim ? e[1] : 0.f;
            }
            const f32x2 x = xv[b][hp];
            const f32x2 ex = e * x;
            s0v += e;
            s1v += ex;
            s2v += ex * x;
          }
        float s0 = s0v[0] + s0v[1], s1 = s1v[0] + s1v[1], s2 = s2v[0] + s2v[1];
        s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        WS_ASTAMP(10 + 4 * (it + half));
        if (q4 == 0) {
          if (nseg > 1) {
            p.partial[(long long)blockIdx.x * F_NC + c0 + r16] = (f32x4){mx, s0, s1, s2};
          } else {
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
}

// pooled[u] = [mean | std] from the nseg tuples of utterance u (the softmax's max differs per segment: rescale)
__global__ void astp_combine_kernel(const f32x4* __restrict__ partial, int nseg, float* __restrict__ pooled) {
  const int u = blockIdx.x;
  for (int c = threadIdx.x; c < F_NC; c += blockDim.x) {
    float M = -1e30f;
    for (int s = 0; s < nseg; ++s) M = fmaxf(M, partial[((long long)u * nseg + s) * F_NC + c][0]);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int s = 0; s < nseg; ++s) {
      const f32x4 t = partial[((long long)u * nseg + s) * F_NC + c];
      const float w = __expf(t[0] - M);
      s0 += t[1] * w;
      s1 += t[2] * w;
      s2 += t[3] * w;
    }
    const float inv = 1.f / s0, mean = s1 * inv, var = s2 * inv - mean * mean;
    pooled[(long long)u * 2 * F_NC + c] = mean;
    pooled[(long long)u * 2 * F_NC + F_NC + c] = sqrtf(fmaxf(var, 1e-7f));
  }
}

// segments of an utterance of T frames: as few as fit 208 rows each, equal lengths (within nseg - 1 rows), and the
// shortest one still covers the row blocks that its kernel form treats as whole (16 (RB - 3) rows)
struct AstpSegs { int nseg, rows, rb; };
AstpSegs astp_segments(int T) {
  for (int nseg = (T + 207) / 208; nseg <= 16; ++nseg) {
    const int rows = (T + nseg - 1) / nseg, last = T - (nseg - 1) * rows;
    const int rb = rows <= 112 ? 7 : (rows <= 160 ? 10 : 13);
    if (rows >= 64 && last >= 16 * (rb - 3) && last >= 1) return {nseg, rows, rb};
  }
  return {0, 0, 0};
}
}  // namespace

bool astp_fused_supported(int T, int C, int bottleneck) {
  static const int off = [] { const char* e = getenv("WS_ASTP_FUSED"); return e && atoi(e) == 0 ? 1 : 0; }();
  // 7, 10 or 13 row blocks of 16 frames; shorter utterances stay on the tile kernels, longer ones too (one workgroup
  // per utterance stops paying when an utterance needs several passes: with a constant rows budget the batch
  // then holds fewer utterances than there are CUs)
  // (round 6: longer utterances as segments of <= 208 frames + a merge of their softmax tuples, up to 16 segments)
  return !off && C == F_NC && bottleneck == F_NH && T >= 64 && astp_segments(T).nseg > 0;
}

// One workgroup per utterance costs a fixed ~25 us per row block and round of workgroups whatever the batch is; the
// three tile-kernel launches cost ~497 us per 256 x 198 rows in proportion to the rows.  A lone utterance (latency!)
// or a batch that fills a round badly stays on the tile kernels.
bool astp_fused_pays(int B, int T) {
  static const int force = [] { const char* e = getenv("WS_ASTP_FUSED"); return e && atoi(e) == 2 ? 1 : 0; }();
  if (force) return true;                        // (tests: the kernel on small batches)
  const int cus = current_device_cus();
  const AstpSegs sg = astp_segments(T);
  if (!sg.nseg) return false;
  const int rb = sg.rb;
  const double fused = (double)(((long long)B * sg.nseg + cus - 1) / cus) * (25.5 * rb) + (sg.nseg > 1 ? 8.0 : 0.0);
  const double tiles = 497.0 * ((double)B * T) / (256.0 * 198.0) + 20.0;
  return fused < tiles;
}

namespace {
template <int RB>
hipError_t launch_astp_rb(const AstpFusedParams& p, hipStream_t stream) {
  // (per device: several GPUs in one process each need the attribute)
  static size_t granted0[WS_MAX_DEVICES] = {}, granted1[WS_MAX_DEVICES] = {};
  hipError_t e = p.row_len ? ensure_dynamic_lds(reinterpret_cast<const void*>(astp_fused_kernel<true, RB>),
                                                f_lds_bytes(RB), granted1)
                           : ensure_dynamic_lds(reinterpret_cast<const void*>(astp_fused_kernel<false, RB>),
                                                f_lds_bytes(RB), granted0);
  if (e != hipSuccess) return e;
  const dim3 grid((unsigned)(p.B * p.nseg));
  if (p.row_len) hipLaunchKernelGGL((astp_fused_kernel<true, RB>), grid, dim3(512), f_lds_bytes(RB), stream, p);
  else hipLaunchKernelGGL((astp_fused_kernel<false, RB>), grid, dim3(512), f_lds_bytes(RB), stream, p);
  if (p.nseg > 1) hipLaunchKernelGGL(astp_combine_kernel, dim3(p.B), dim3(512), 0, stream, p.partial, p.nseg, p.pooled);
  return hipGetLastError();
}
}  // namespace

hipError_t launch_astp_fused(const float* h, int ldh, int B, int T, const float* w1, int ldw1, const float* bias,
                             const float* bias_img, const float* w2, int ldw2, float* pooled, const int* lens,
                             hipStream_t stream, float* seg_scratch) {
  if (!astp_fused_supported(T, F_NC, F_NH) || (ldh & 3) || (ldw1 & 3) || (ldw2 & 3) || B <= 0)
    return hipErrorInvalidValue;
  const AstpSegs sg = astp_segments(T);
  // (segments: B * nseg * 1536 tuples of 16 B in seg_scratch, 16-byte aligned)
  if (sg.nseg > 1 && (!seg_scratch || (reinterpret_cast<unsigned long long>(seg_scratch) & 15))) return hipErrorInvalidValue;
  // 32-bit byte offsets inside one utterance / the weight matrices
  if ((long long)T * ldh * 4 >= (1ll << 31) || (long long)F_NH * ldw1 * 4 >= (1ll << 31)) return hipErrorInvalidValue;
  AstpFusedParams p;
  p.h = h; p.ldh = ldh; p.w1 = w1; p.ldw1 = ldw1; p.bias = bias; p.bias_img = bias_img;
  p.w2 = w2; p.ldw2 = ldw2; p.pooled = pooled; p.row_len = lens; p.B = B; p.T = T;
  p.nseg = sg.nseg; p.seg_rows = sg.rows; p.partial = reinterpret_cast<f32x4*>(seg_scratch);
  if (sg.rb == 7) return launch_astp_rb<7>(p, stream);
  if (sg.rb == 10) return launch_astp_rb<10>(p, stream);
  return launch_astp_rb<13>(p, stream);
}

}  // namespace wsamd
