// Direct 3x3 convolution, C -> C channels (C = 32 or 64), on binary16 channels-last maps (f16 back-end).
//
// ResNet18/34's first stage (wespeaker/models/resnet.py:35-69,163-169: BasicBlock convs on the
// 32-channel 80 x T map) and CAM++'s FCM head (campplus.py:245-330) are 3x3 convolutions with 32
// input and output channels over millions of pixels.  As an implicit GEMM (K = 288, N = 32) every
// input pixel is fetched once per filter tap -- nine times -- and the loop is bound by the global->LDS
// rate.  Here a workgroup owns an 8 x 16 patch of output pixels, copies the (7 s_h + 3) x (15 s_w + 3)
// input patch with its halo into LDS ONCE by LDS-DMA (global_load_lds_dwordx4; 1.4x the unique
// bytes instead of 9x), keeps all 32 x 288 weights in registers (18 A-fragments per lane) and forms
// the nine taps by shifting the LDS read address: 18 x v_mfma_f32_32x32x16_f16 per wavefront and
// patch.  The weights are the MFMA's A operand, so a lane ends up with 16 channels of ONE pixel and
// the epilogue (bias, residual, ReLU, binary16) stores 32 contiguous bytes per lane straight to HBM.
// Workgroups are persistent over patches with two LDS stages: the next patch's halo is in flight
// while the current one is multiplied.  LDS pixels are 64 B; the four 16-B chunks of a pixel are
// XOR-swizzled by (pixel >> 2) & 3 on the DMA source side and on the reads (ds_read_b128 then hits
// 16 distinct 16-B slots per lane group up to a 2-way overlap at patch-row boundaries).
// C = 64 (ResNet stage 2): pixels are 128 B (8 chunks, key (pixel >> 1) & 7) and the workgroup has 8
// wavefronts = 4 pixel groups x 2 output-channel halves, so that each wavefront's 36 weight fragments
// (144 VGPRs) still live in registers; the two halves read the same activation fragments.
#include "kernels.h"
#include <cstdio>
#include <cstdlib>

namespace wsamd {

typedef float f32x16d __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8d __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void dma16_direct(const void* g, void* lds_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
#endif
}
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

constexpr int PH = 8, PW = 16;                   // output pixels per patch (rows x cols)

template <int C, int SH, int SW>
__global__ __launch_bounds__(C == 32 ? 256 : 512, 2) void conv3x3_direct_f16_kernel(const ConvGemmParams p,
                                                                                    int pyb, int pxb,
                                                                                    int total) {
  constexpr int NWV = C == 32 ? 4 : 8;           // wavefronts: 4 pixel groups x (C / 32) channel halves
  constexpr int CHK = C / 8;                     // 16-B chunks per pixel
  constexpr int PXP = 64 / CHK;                  // pixels per 1-KiB DMA piece
  constexpr int KSUB = C / 16;                   // 16-wide k-steps per tap
  constexpr int KS = 9 * KSUB;                   // k-steps in all
  constexpr int IH = (PH - 1) * SH + 3, IW = (PW - 1) * SW + 3;
  constexpr int RP = IH * IW;                    // region pixels
  constexpr int NP = (RP + PXP - 1) / PXP;       // 1-KiB pieces
  constexpr int PPW = (NP + NWV - 1) / NWV;      // pieces per wavefront
  constexpr int STAGE = NP * 1024;
  extern __shared__ __attribute__((aligned(16))) char lds_d[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = C == 32 ? wave : wave >> 1;     // pixel group (2 patch rows)
  const int ct = C == 32 ? 0 : wave & 1;         // 32-channel output tile
  const int li = lane & 31, lh = lane >> 5;
  auto key = [](int q) { return C == 32 ? (q >> 2) & 3 : (q >> 1) & 7; };

  // ---- weights: A fragments of the k-steps (k = 16 s + 8 lh .. +7 of row cout = 32 ct + li)
  f16x8d wf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s)
    wf[s] = *reinterpret_cast<const f16x8d*>(p.Wh + (long long)(32 * ct + li) * p.ldw + 16 * s + 8 * lh);

  // ---- DMA roles: piece pi covers region pixels PXP pi .., lane -> (pixel, physical chunk)
  int d_ry[PPW], d_rx[PPW], d_c[PPW], d_slot[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    int pi = wave * PPW + k;
    if (pi > NP - 1) pi = NP - 1;                // surplus wavefronts repeat the last piece
    const int q = pi * PXP + lane / CHK, pc = lane % CHK;
    d_ry[k] = q < RP ? q / IW : -(1 << 20);      // out of region -> predicate false
    d_rx[k] = q - (q / IW) * IW;
    d_c[k] = (pc ^ key(q)) * 8;                  // logical chunk (halfs) stored at physical slot pc
    d_slot[k] = pi * 1024;
  }
  auto issue = [&](int patch, int stage) {
    const int pxi = patch % pxb, t = patch / pxb;
    const int pyi = t % pyb, img = t / pyb;
    const int iy0 = pyi * PH * SH - 1, ix0 = pxi * PW * SW - 1;
    char* base = lds_d + stage * STAGE;
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int iy = iy0 + d_ry[k], ix = ix0 + d_rx[k];
      const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
      const uint16_t* src = ok ? p.A16 + (((long long)img * p.Hin + iy) * p.Win + ix) * C + d_c[k]
                               : reinterpret_cast<const uint16_t*>(p.zeros);
      dma16_direct(src, base + d_slot[k]);
    }
  };

  // ---- compute roles: lane li -> patch pixel (row 2 pg + li / 16, col li % 16)
  const int ppy = 2 * pg + (li >> 4), ppx = li & 15;
  const int q0 = ppy * SH * IW + ppx * SW;       // region pixel of tap (0, 0)
  // per-channel epilogue constants: after the lane exchange a lane owns channels 32 ct + 16 lh .. + 15
  const int ch0 = 32 * ct + 16 * lh;
  // (C = 64 has no registers to spare for the bias: it is re-read from L1/L2 per patch there)
  float bias[C == 32 ? 16 : 1];
  if constexpr (C == 32) {
#pragma unroll
    for (int c = 0; c < 16; ++c) bias[c] = p.bias ? p.bias[ch0 + c] : 0.f;
  }

  int stage = 0;
  int patch = blockIdx.x;
  if (patch < total) issue(patch, 0);
  for (; patch < total; patch += gridDim.x) {
    const int next = patch + gridDim.x;
    if (next < total) {
      issue(next, stage ^ 1);
      wait_vm_barrier<PPW>();                    // this patch's pieces landed (the next one's stay in flight)
    } else {
      wait_vm_barrier<0>();
    }
    const char* base = lds_d + stage * STAGE;
    f32x16d acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int q = q0 + (tap / 3) * IW + (tap % 3);
      const int kq = key(q);
#pragma unroll
      for (int sub = 0; sub < KSUB; ++sub) {
        const int c = sub * 2 + lh;
        const f16x8d a = *reinterpret_cast<const f16x8d*>(base + q * (C * 2) + ((c ^ kq) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[tap * KSUB + sub], a, acc, 0, 0, 0);
      }
    }
    // C^T layout: column = pixel li, rows (channels of this tile) = (r & 3) + 8 (r >> 2) + 4 lh.  Exchange
    // with lane ^ 32 so that lh = 0 owns channels 0..15 and lh = 1 owns 16..31 (contiguous 32 B each).
    float v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      // target channel 16 lh + c lives in the lane half s = (c >> 2) & 1, register ra (+8 for the upper
      // 16 channels): the owner half keeps its own register, the other half receives the partner's
      const int s = (c >> 2) & 1, ra = (c & 3) + 4 * (c >> 3);
      const float x = __shfl_xor(acc[ra + 8 * (1 - s)], 32, 64);
      v[c] = lh == s ? acc[ra + 8 * s] : x;
    }
    const int pxi = patch % pxb, t = patch / pxb;
    const int pyi = t % pyb, img = t / pyb;
    const int oy = pyi * PH + ppy, ox = pxi * PW + ppx;
    if (oy < p.Hout && ox < p.Wout) {
      const long long m = ((long long)img * p.Hout + oy) * p.Wout + ox;
      const bool padded = p.row_len && ox >= p.row_len[img];
      if (p.residual16) {
        const f16x8d r0 = *reinterpret_cast<const f16x8d*>(p.residual16 + m * p.ldr + ch0);
        const f16x8d r1 = *reinterpret_cast<const f16x8d*>(p.residual16 + m * p.ldr + ch0 + 8);
#pragma unroll
        for (int c = 0; c < 8; ++c) { v[c] += (float)r0[c]; v[8 + c] += (float)r1[c]; }
      }
      if constexpr (C != 32) {
        if (p.bias) {
#pragma unroll
          for (int c = 0; c < 16; c += 4) {
            const float4 b4 = *reinterpret_cast<const float4*>(p.bias + ch0 + c);
            v[c] += b4.x; v[c + 1] += b4.y; v[c + 2] += b4.z; v[c + 3] += b4.w;
          }
        }
      }
      f16x8d o0, o1;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float a = v[c], b = v[8 + c];
        if constexpr (C == 32) { a += bias[c]; b += bias[8 + c]; }
        if (p.act == ACT_RELU) { a = relu_f(a); b = relu_f(b); }
        if (padded) { a = 0.f; b = 0.f; }        // ragged batch: columns beyond the utterance stay zero
        o0[c] = (_Float16)a; o1[c] = (_Float16)b;
      }
      *reinterpret_cast<f16x8d*>(p.D16 + m * p.ldd16 + ch0) = o0;
      *reinterpret_cast<f16x8d*>(p.D16 + m * p.ldd16 + ch0 + 8) = o1;
    }
    __syncthreads();                             // everyone is done with this stage before it is refilled
    stage ^= 1;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same layers on the parity-grade fp32 back-end (round 3): 32 -> 32 channels, fp32 maps, exact fp32 products
// (v_mfma_f32_32x32x2_f32).  As an implicit GEMM these layers ran at 0.43 of the fp32 MFMA peak (N = 32: a 128 x 32
// tile re-stages 16 KB of activations for 16 MFMAs per wavefront and K-tile, nine times per input pixel).  Here, as in
// the binary16 kernel: persistent workgroups over 8 x 16 output patches, the input patch with its halo copied into LDS
// once by LDS-DMA (two stages), the nine taps formed by shifting the swizzled LDS read address, and ALL 32 x 288 weights
// in registers -- 144 VGPRs per lane: the weights are the MFMA's A operand, lane (li, lh) holds W[li][32 tap + 8 g +
// 4 lh + s].  Pixels are 128 B (8 chunks of 16 B, chunk index XOR (pixel >> 1) & 7); one ds_read_b128 of a pixel row
// feeds four MFMAs (k = 8 g + 4 lh + s: the k order of the GEMM kernels).  144 MFMAs per wavefront and patch.
// C^T accumulators: a lane owns channels 8 j + 4 lh .. + 3 (j = 0..3) of ONE pixel: four 16-B stores per lane.
template <int SH, int SW>
__global__ __launch_bounds__(256, 2) void conv3x3_direct_f32_kernel(const ConvGemmParams p, int pyb, int pxb,
                                                                    int total) {
  constexpr int C = 32, NWV = 4, PXP = 8;
  constexpr int IH = (PH - 1) * SH + 3, IW = (PW - 1) * SW + 3;
  constexpr int RP = IH * IW;
  constexpr int NP = (RP + PXP - 1) / PXP;       // 1-KiB pieces (8 pixels of 128 B)
  constexpr int PPW = (NP + NWV - 1) / NWV;
  constexpr int STAGE = NP * 1024;
  constexpr int ISSUE_UNROLL = PPW <= 6 ? PPW : 2;   // (the strided region has 10 pieces per wavefront: unrolled, their
                                                       // address arithmetic spills beside the 144 weight registers)
  typedef float f32x4d __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char lds_d[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  // chunk swizzle: eight consecutive pixels (= eight consecutive lanes of a fragment read) use eight different 16-B
  // slots of their 128-B rows, i.e. all 32 banks once (round 3's (q >> 1) & 7 gave every read a 2-way conflict)
  auto key = [](int q) { return q & 7; };
  // stride 1: the residual tile of a patch (128 pixels x 128 B) also comes by LDS-DMA, one patch ahead, into two 16-KB
  // stages behind the bias (chunk index XOR (pixel & 7)); the epilogue then has no global load left
  constexpr bool RES_DMA = SH == 1 && SW == 1;
  char* const res_s = lds_d + 2 * STAGE + 128;

  // ---- weights: lane (li, lh) keeps W[li][32 tap + 8 g + 4 lh .. + 3] for all 9 taps x 4 k-groups
  f32x4d wf[36];
#pragma unroll
  for (int s = 0; s < 36; ++s)
    wf[s] = *reinterpret_cast<const f32x4d*>(p.W + (long long)li * p.ldw + 8 * s + 4 * lh);

  // DMA roles: piece pi covers region pixels 8 pi .. 8 pi + 7, lane -> (pixel, physical chunk).  They are recomputed
  // per patch (a dozen integer operations per piece) instead of being kept: beside 144 weight registers and the
  // accumulators there is room for EITHER the role registers OR a second set of activation fragments, and the
  // fragments of tap t + 1 in flight behind the MFMAs of tap t are worth more
  auto issue = [&](int patch, int stage) {
    const int pxi = patch % pxb, t = patch / pxb;
    const int pyi = t % pyb, img = t / pyb;
    const int iy0 = pyi * PH * SH - 1, ix0 = pxi * PW * SW - 1;
    char* base = lds_d + stage * STAGE;
#pragma unroll ISSUE_UNROLL
    for (int k = 0; k < PPW; ++k) {
      int pi = wave * PPW + k;
      if (pi > NP - 1) pi = NP - 1;              // surplus slots repeat the last piece
      const int q = pi * PXP + (lane >> 3), pc = lane & 7;
      const int ry = q / IW, rx = q - ry * IW;
      const int iy = iy0 + ry, ix = ix0 + rx;
      const bool ok = q < RP && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
      const float* src = ok ? p.A + (((long long)img * p.Hin + iy) * p.Win + ix) * C + ((pc ^ key(q)) * 4) : p.zeros;
      dma16_direct(src, base + pi * 1024);
    }
  };

  // one piece of the next patch's halo: its address arithmetic + DMA ride in the shadow of one step's four MFMAs (a
  // wavefront's serial phases are not covered by its SIMD partner here: measured, every serial phase costs its own time)
  auto issue_piece = [&](int img, int iy0, int ix0, int stage, int k) {
    int pi = wave * PPW + k;
    if (pi > NP - 1) pi = NP - 1;
    const int q = pi * PXP + (lane >> 3), pc = lane & 7;
    const int ry = q / IW, rx = q - ry * IW;
    const int iy = iy0 + ry, ix = ix0 + rx;
    const bool ok = q < RP && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
    const float* src = ok ? p.A + (((long long)img * p.Hin + iy) * p.Win + ix) * C + ((pc ^ key(q)) * 4) : p.zeros;
    dma16_direct(src, lds_d + stage * STAGE + pi * 1024);
  };

  auto issue_res_piece = [&](int img, int oy0, int ox0, int stage, int r) {
    const int pp = (wave * 4 + r) * 8 + (lane >> 3), pc = lane & 7;
    const int oy = oy0 + (pp >> 4), ox = ox0 + (pp & 15);
    const bool ok = oy < p.Hout && ox < p.Wout;
    const float* src = ok ? p.residual + (((long long)img * p.Hout + oy) * p.Wout + ox) * p.ldr + p.r_off + ((pc ^ (pp & 7)) * 4)
                          : p.zeros;
    dma16_direct(src, res_s + stage * 16384 + (wave * 4 + r) * 1024);
  };

  // ---- compute roles: lane li -> patch pixel (row 2 wave + li / 16, col li % 16)
  const int ppy = 2 * wave + (li >> 4), ppx = li & 15;
  const int q0 = ppy * SH * IW + ppx * SW;       // region pixel of tap (0, 0)
  // the bias lives in LDS behind the two stages (no registers to spare beside the 144 weight registers; a global
  // load per patch, as in round 3, is a ~500-cycle round trip in the exposed epilogue)
  float* const bias_s = reinterpret_cast<float*>(lds_d + 2 * STAGE);
  if (tid < 32) bias_s[tid] = p.bias ? p.bias[tid] : 0.f;
  __syncthreads();

  int stage = 0;
  int patch = blockIdx.x;
  if (patch < total) {
    issue(patch, 0);
    if (RES_DMA && p.residual) {
      const int t0 = patch / pxb;
#pragma unroll
      for (int r = 0; r < 4; ++r) issue_res_piece(t0 / pyb, (t0 % pyb) * PH, (patch % pxb) * PW, 0, r);
    }
  }
  bool stores_pending = false;                   // (wave-uniform) the previous patch's four row stores were issued
  for (; patch < total; patch += gridDim.x) {
    // ONE rendezvous per patch: this patch's pieces have landed, and everybody is done with the previous patch -- its
    // stage is free for the next patch's pieces, which then have a whole patch time to arrive.  The previous patch's
    // output stores were issued BEHIND those pieces (VMEM operations retire in order), so they may stay in flight:
    // round 3 waited for them too (vmcnt(0)) and paid a store round trip per patch.
    if (stores_pending) wait_vm_barrier<4>();
    else wait_vm_barrier<0>();
    const int next = patch + gridDim.x;
    const bool more = next < total;              // (uniform) the next patch's pieces go out behind steps 2 .. 2 + PPW
    int n_img = 0, n_iy0 = 0, n_ix0 = 0;
    if (more) {
      const int nx = next % pxb, nt = next / pxb;
      n_img = nt / pyb;
      n_iy0 = (nt % pyb) * PH * SH - 1;
      n_ix0 = nx * PW * SW - 1;
    }
    const char* base = lds_d + stage * STAGE;
    const int pxi = patch % pxb, t = patch / pxb;
    const int pyi = t % pyb, img = t / pyb;
    const int oy = pyi * PH + ppy, ox = pxi * PW + ppx;
    const bool inside = oy < p.Hout && ox < p.Wout;
    const long long m = ((long long)img * p.Hout + (inside ? oy : 0)) * p.Wout + (inside ? ox : 0);
    f32x16d acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // 36 steps of (tap, k-group): one ds_read_b128 = the B operands of four MFMAs; the fragment of step i + 1 is
    // requested before the MFMAs of step i (four 64-cycle MFMAs cover the LDS round trip)
    f32x4d fa[3];                                // fragments two steps ahead of their MFMAs (512 cycles for the LDS
    auto read_step = [&](int i) {                // round trip under eight wavefronts' reads + the DMA writes)
      const int tap = i >> 2, g = i & 3;
      const int q = q0 + (tap / 3) * IW + (tap % 3);
      return *reinterpret_cast<const f32x4d*>(base + q * 128 + (((g * 2 + lh) ^ key(q)) << 4));
    };
    fa[0] = read_step(0);
    fa[1] = read_step(1);
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      if (i + 2 < 36) fa[(i + 2) % 3] = read_step(i + 2);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[i][s], fa[i % 3][s], acc, 0, 0, 0);
      if (PPW <= 26 && i >= 2 && i < 2 + PPW && more) issue_piece(n_img, n_iy0, n_ix0, stage ^ 1, i - 2);
      if (RES_DMA && PPW <= 26 && i >= 2 + PPW && i < 6 + PPW && more && p.residual)
        issue_res_piece(n_img, (n_iy0 + 1) / SH, (n_ix0 + 1) / SW, stage ^ 1, i - 2 - PPW);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (PPW > 26 && more) issue(next, stage ^ 1);
    // C^T layout: column = pixel li, rows (channels) = (r & 3) + 8 (r >> 2) + 4 lh
    const bool padded = p.row_len && ox >= p.row_len[img];
    float* const drow = p.D + m * p.ldd + p.d_off + 4 * lh;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4d v = {acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
      v += *reinterpret_cast<const f32x4d*>(bias_s + 8 * j + 4 * lh);
      if (p.residual) {
        if (RES_DMA) v += *reinterpret_cast<const f32x4d*>(res_s + stage * 16384 + (wave * 32 + li) * 128 +
                                                             (((2 * j + lh) ^ (li & 7)) << 4));
        else v += *reinterpret_cast<const f32x4d*>(p.residual + m * p.ldr + p.r_off + 8 * j + 4 * lh);
      }
      if (p.act == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = relu_f(v[e]);
      }
      if (padded) v = (f32x4d){0.f, 0.f, 0.f, 0.f};     // ragged batch: columns beyond the utterance stay zero
      if (inside) *reinterpret_cast<f32x4d*>(drow + 8 * j) = v;
    }
    // (vmcnt counts wave instructions: the four stores were issued unless no lane of the wavefront was inside)
    stores_pending = __builtin_amdgcn_ballot_w64(inside) != 0;
    stage ^= 1;
  }
}

// (Round 4 looked for what keeps this kernel at ~0.7 of the fp32 MFMA rate on ResNet34's first stage -- PMC: the matrix
// pipe busy 0.715 of the kernel's cycles at 2.34 GHz, full occupancy.  Ruled out, one rocprofv3 run each: the workgroup
// barrier per patch (a barrier-free form with one 2 x 16 patch and private LDS stages per wavefront: 1432 vs 1443 us),
// the dependent accumulation chain (two chains per wavefront: same), lock step of the two wavefronts of a SIMD
// (unequal s_setprio: same), LDS bank conflicts and the fragment prefetch distance (conflict-free swizzle, two steps
// ahead: same).  What the ablations showed instead: every SERIAL phase of a wavefront costs its own time -- the SIMD
// partner does not cover it -- without LDS-DMA the loop takes 1153 - 1296 us, without stores 1369, without residual
// loads 1373.  So the serial phases were moved into the shadow of the wavefront's OWN MFMAs, as the persistent GEMM
// does: the next patch's DMA pieces one per step (1437 -> 1372 us), the residual tile by LDS-DMA one patch ahead
// (-> 1332 us), the bias from LDS and counted vmcnt for the stores (earlier: 1505 -> 1443).  Left exposed: the
// rendezvous at the top of a patch and the epilogue's maths + stores; hiding those needs the 16 accumulator registers
// twice, i.e. half of the 144 weight registers moved to LDS fragments.)

bool conv3x3_direct_f32_supported(const ConvGemmParams& p) {
  return p.prec == 0 && p.A && p.D && !p.A16 && !p.D16 && !p.A2 && !p.pre_scale && p.Cin == 32 && p.N == 32 &&
         p.K == 288 && p.kh == 3 && p.kw == 3 && p.dil_h == 1 && p.dil_w == 1 && p.pad_h == 1 && p.pad_w == 1 &&
         p.lda == 32 && p.a_off == 0 && (p.ldd & 3) == 0 && (p.d_off & 3) == 0 && p.ldw >= p.K &&
         // stride (1,1), and CAM++'s frequency-only stride (2,1) (campplus.py:245-330: three FCM convolutions; the
         // DMA roles are recomputed per patch, so the larger region costs LDS -- 2 x 39 KB -- not registers)
         p.stride_w == 1 && (p.stride_h == 1 || p.stride_h == 2) &&
         !p.residual16 && !p.colsum && !p.colsumsq && !p.pool_partial &&
         !p.seg_scale && !p.post_scale && !p.bias_img && !p.D2 && p.splitk <= 1 && p.m_begin == 0 &&
         p.act != ACT_TANH && (!p.residual || ((p.ldr & 3) == 0 && (p.r_off & 3) == 0)) &&
         (long long)p.M * 32 >= (1 << 22);       // small maps stay on the tile kernels (persistent patches need work)
}

template <int SH, int SW>
static hipError_t launch_direct_f32(const ConvGemmParams& p, hipStream_t stream) {
  constexpr int IH = (PH - 1) * SH + 3, IW = (PW - 1) * SW + 3;
  constexpr int NP = (IH * IW + 7) / 8;
  constexpr size_t lds = 2 * (size_t)NP * 1024 + 128 + (SH == 1 && SW == 1 ? 2 * 16384 : 0);   // two stages + the bias (+ two residual stages)
  static_assert(lds <= 160 * 1024, "two stages fit the LDS");
  auto kern = conv3x3_direct_f32_kernel<SH, SW>;
  static size_t lds_granted[WS_MAX_DEVICES] = {};
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_granted);
    if (e != hipSuccess) return e;
  }
  const int images = p.M / (p.Hout * p.Wout);
  const int pyb = (p.Hout + PH - 1) / PH, pxb = (p.Wout + PW - 1) / PW;
  const long long total = (long long)images * pyb * pxb;
  if (total <= 0) return hipSuccess;
  const int cus = current_device_cus();
  const size_t by_lds = 160 * 1024 / lds;
  long long blocks = (long long)cus * (by_lds < 2 ? by_lds : 2);   // 144 weight VGPRs: two workgroups per CU
  if (blocks > total) blocks = total;
  if (dispatch_log_enabled()) {
    char k[64];
    snprintf(k, sizeof(k), "conv3x3_direct_f32_kernel<%d,%d>", SH, SW);
    dispatch_log_note(p, k);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, p, pyb, pxb, (int)total);
  return hipGetLastError();
}

hipError_t launch_conv3x3_direct_f32(const ConvGemmParams& p, hipStream_t stream) {
  if (!conv3x3_direct_f32_supported(p)) return hipErrorInvalidValue;
  if (p.stride_h == 2) return launch_direct_f32<2, 1>(p, stream);
  return launch_direct_f32<1, 1>(p, stream);
}

bool conv3x3_direct_supported(const ConvGemmParams& p) {
  const bool c32 = p.Cin == 32 && p.N == 32 && p.K == 288;
  const bool c64 = p.Cin == 64 && p.N == 64 && p.K == 576 && p.stride_h == 1 && p.stride_w == 1;   // strided regions would not fit LDS
  return p.prec == 2 && p.A16 && p.D16 && !p.D && !p.A2 && !p.pre_scale && p.kh == 3 && p.kw == 3 &&
         p.dil_h == 1 && p.dil_w == 1 && p.pad_h == 1 && p.pad_w == 1 && (c32 || c64) &&
         p.lda16 == p.Cin && p.ldd16 == p.N && p.a_off == 0 && p.d_off == 0 && p.ldw >= p.K &&
         (p.stride_h == 1 || p.stride_h == 2) && (p.stride_w == 1 || p.stride_w == 2) &&
         !(p.stride_h == 1 && p.stride_w == 2) && !p.residual && !p.colsum && !p.pool_partial &&
         !p.seg_scale && !p.post_scale && !p.bias_img && !p.D2 && p.splitk <= 1 && p.m_begin == 0 &&
         p.act != ACT_TANH && (!p.residual16 || (p.ldr == p.N && p.r_off == 0));
}

template <int C, int SH, int SW>
static hipError_t launch_direct(const ConvGemmParams& p, hipStream_t stream) {
  constexpr int IH = (PH - 1) * SH + 3, IW = (PW - 1) * SW + 3;
  constexpr int PXP = 64 / (C / 8);
  constexpr int NP = (IH * IW + PXP - 1) / PXP;
  constexpr size_t lds = 2 * (size_t)NP * 1024;
  auto kern = conv3x3_direct_f16_kernel<C, SH, SW>;
  static size_t lds_granted[WS_MAX_DEVICES] = {};
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_granted);
    if (e != hipSuccess) return e;
  }
  const int images = p.M / (p.Hout * p.Wout);
  const int pyb = (p.Hout + PH - 1) / PH, pxb = (p.Wout + PW - 1) / PW;
  const long long total = (long long)images * pyb * pxb;
  if (total <= 0) return hipSuccess;
  const int cus = current_device_cus();
  // resident workgroups per CU: the VGPR budget allows 8 wavefronts (2 x 4 waves for C = 32, 1 x 8 for 64)
  const size_t by_lds = 160 * 1024 / lds;
  const size_t by_regs = C == 32 ? 2 : 1;
  long long blocks = (long long)cus * (by_lds < by_regs ? by_lds : by_regs);
  if (blocks > total) blocks = total;
  if (dispatch_log_enabled()) {
    char k[64];
    snprintf(k, sizeof(k), "conv3x3_direct_f16_kernel<%d,%d,%d>", C, SH, SW);
    dispatch_log_note(p, k);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C == 32 ? 256 : 512), lds, stream, p, pyb, pxb,
                     (int)total);
  return hipGetLastError();
}

hipError_t launch_conv3x3_direct(const ConvGemmParams& p, hipStream_t stream) {
  if (p.Cin == 64) return launch_direct<64, 1, 1>(p, stream);
  if (p.stride_h == 1) return launch_direct<32, 1, 1>(p, stream);
  if (p.stride_w == 1) return launch_direct<32, 2, 1>(p, stream);
  return launch_direct<32, 2, 2>(p, stream);
}

}  // namespace wsamd
