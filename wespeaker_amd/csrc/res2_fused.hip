// Fused Res2 chain of an ECAPA SE-Res2Block: the 7 serial (w -> w, k = 3, dilation d) convolutions
//   sp_i = BN(ReLU(conv_i(sp_{i-1} + split_i)))          (wespeaker/models/ecapa_tdnn.py:58-78)
// in ONE launch, one workgroup per utterance.  The running activation lives in LDS for the whole
// chain (it never round-trips HBM between the 7 steps), each step is a [T x 3w] x [3w x w] GEMM on
// exact-fp32 v_mfma_f32_16x16x4_f32 with the step's weights held in registers.
//
//  * 8 wavefronts: wave -> (16-column output tile wn, row group wm); MTW 16-row tiles per wave.
//  * X (LDS): rows t = -d .. Tpad+d (zero halo = the conv's zero padding), row stride w+8 floats so
//    that a ds_read_b128 lane group (16 rows x 4 k-quads) touches 16 distinct 16-B slots.
//  * lane (i = l & 15, q = l >> 4) reads 4 consecutive k per ds_read_b128 and feeds 4 MFMAs; the
//    weight fragment uses the same k permutation, so the sum is over all k exactly once.
//  * epilogue per step: + bias, ReLU, BN affine -> store sp_i to y2[:, i*w ...] and write
//    sp_i + split_{i+1} back into X for the next step.
//  * utterances longer than a workgroup's rows (224 at w = 64, 208 at w = 128) are cut into TIME TILES, one
//    workgroup each: a tile owns `tile_rows` output rows and also carries a halo of 7 * d rows on each side,
//    because step s needs rows t +- d of step s - 1: whatever is computed within s * d rows of the window edge
//    is wrong, never stored, and after the seventh step still 0 rows short of the owned range.  (Before round 2
//    such utterances -- anything beyond 4.2 s -- took 21 separate GEMM launches per block.)
#include "kernels.h"
#include <cstdlib>

namespace wsamd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef WS_TRACE
__device__ unsigned long long g_res2_trace[128];          // [wavefront 0 | wavefront 4] x 64 stamps
unsigned long long* res2_trace_buffer_address() {
  unsigned long long* q = nullptr;
  (void)hipGetSymbolAddress(reinterpret_cast<void**>(&q), HIP_SYMBOL(g_res2_trace));
  return q;
}
#define WS_RSTAMP(i)                                                                              \
  if (blockIdx.x == 9 && (threadIdx.x == 0 || threadIdx.x == 256))                                \
    g_res2_trace[(i) + (threadIdx.x >> 2)] = __builtin_readcyclecounter();
#else
#define WS_RSTAMP(i)
#endif

template <int W, int MTW>
__global__ __launch_bounds__(512) void res2_chain_kernel(const Res2ChainParams p) {
  constexpr int NT = W / 16;            // output column tiles
  constexpr int MW = 8 / NT;            // row groups of waves
  constexpr int XS = W + 8;             // LDS row stride (floats)
  constexpr int KG = 3 * W / 16;        // k groups of 16 per step
  extern __shared__ __attribute__((aligned(16))) float X[];

  const int b = blockIdx.x / p.tiles, tile = blockIdx.x - b * p.tiles;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % NT, wm = wave / NT;
  const int li = lane & 15, lq = lane >> 4;
  const int d = p.dil;
  // ragged batch: rows [lens[b], p.T) of utterance b are padding -- never read, never written, and seen
  // by the dilated taps as the conv's zero padding (X stays zero there)
  const int T = p.lens ? p.lens[b] : p.T;
  constexpr int CAP = MW * MTW * 16;                    // rows a workgroup computes
  const int rows_total = CAP + 2 * d;                   // LDS rows incl. the conv's zero padding
  const long long m_base = (long long)b * p.T;
  // time tile: LDS row u <-> frame tbase + u; frames [own_lo, own_hi) are stored, the rest is halo
  const int own_lo = tile * p.tile_rows;
  const int own_hi = own_lo + p.tile_rows < T ? own_lo + p.tile_rows : T;
  if (own_lo >= T) return;                              // (ragged batch: a tile beyond this utterance's end)
  const int tbase = p.tiles > 1 ? own_lo - 7 * d : 0;

  WS_RSTAMP(0)
  // zero the whole X once: padding rows and rows outside [0, T) stay zero for all steps
  for (int i = tid * 4; i < rows_total * XS; i += 512 * 4)
    *reinterpret_cast<f32x4*>(&X[i]) = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  // stage split 0: X[u + d][c] = y1[frame tbase + u][c]
  {
    constexpr int C4 = W / 4;
    for (int i = tid; i < CAP * C4; i += 512) {
      const int u = i / C4, c = (i - u * C4) * 4, t = tbase + u;
      if (t >= 0 && t < T)
        *reinterpret_cast<f32x4*>(&X[(u + d) * XS + c]) =
            *reinterpret_cast<const f32x4*>(p.y1 + (m_base + t) * p.ldy1 + c);
    }
  }
  __syncthreads();

  // The weights are the MFMA's A operand and the activations its B operand (the same fragments: swapping the
  // operands transposes the product), i.e. a wavefront computes a 16 (channels) x 16 (time steps) block of Y^T
  // and a lane owns FOUR CONSECUTIVE CHANNELS (rows 4*lq .. 4*lq+3 of the C tile) of ONE time step (column li):
  // next-split loads, output stores and the LDS write-back of the running activation are 16-byte vectors
  // instead of four scalar accesses per accumulator (28 scalar 4-B global stores per lane and step before).
  const int co = wn * 16 + li;                          // weight row of this lane's A fragment
  const int c0 = wn * 16 + lq * 4;                      // first of the lane's 4 output channels
  constexpr bool PF = MTW <= 7;     // enough registers to prefetch the next split during the MFMAs
  const int tb = wm * MTW * 16 + li;                    // lane's time step inside row tile 0
  // weights of a step for row co: k = 16 g + 4 q + s   (packed [co][tap*W + ci], ld 3W)
  f32x4 bw[KG];
  auto load_weights = [&](int step) {
    const float* wrow = p.w[step] + (long long)co * p.ldw + lq * 4;
#pragma unroll
    for (int g = 0; g < KG; ++g) bw[g] = *reinterpret_cast<const f32x4*>(wrow + g * 16);
  };
  load_weights(0);
  WS_RSTAMP(1)
  for (int step = 0; step < 7; ++step) {
    WS_RSTAMP(2 + step * 5)
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias[step] + c0);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale[step] + c0);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift[step] + c0);
    // The row strides are laundered through an empty asm so the per-tile addresses are recomputed inside
    // the step instead of being hoisted out of the step loop (where they would live across the MFMA
    // section and spill).
    int ld1 = p.ldy1, ld2 = p.ldy2, xw = (tb + d) * XS + c0;
    asm volatile("" : "+v"(ld1), "+v"(ld2), "+v"(xw));
    // next split of y1 (added to this step's output to form the next input): issued now, consumed
    // in the epilogue, so its latency hides under the MFMAs
    f32x4 y1n[PF ? MTW : 1];
    if (PF && step < 6) {
      const float* y1u = p.y1 + (m_base + tbase + tb) * ld1 + (step + 1) * W + c0;
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        const int t = tbase + tb + mt * 16;
        y1n[PF ? mt : 0] = (t >= 0 && t < T) ? *reinterpret_cast<const f32x4*>(y1u + (long long)(mt * 16) * ld1)
                                             : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }

    f32x4 acc[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // rows of this wave's tiles: u = (wm*MTW + mt)*16 + li ; X row = u + d + (tap-1)*d = u + tap*d.
    // Two row tiles at a time (independent accumulators hide the 40-cycle dependent-MFMA latency),
    // activation fragments software-pipelined one k-group ahead; sched_barrier keeps hipcc from hoisting
    // every ds_read of the unrolled loop to the top (which spills).
    const float* xbase = &X[(wm * MTW * 16 + li) * XS + lq * 4];
    auto xaddr = [&](int mt, int g) {
      const int tap = g / (W / 16), cg = g % (W / 16);
      return xbase + (mt * 16 + tap * d) * XS + cg * 16;
    };
    // (the fragments of k-group g + 1 are requested BEFORE the MFMAs of k-group g, into the other of two register
    // sets: with one set and `a = next` behind the MFMAs hipcc reused the registers and issued each read behind the
    // group's last MFMA -- an exposed LDS round trip every 8 MFMAs, ~0.65 of the MFMA rate for the whole chain)
#pragma unroll
    for (int mp = 0; mp < MTW; mp += 2) {
      const bool two = mp + 1 < MTW;
      // a row tile pair that lies entirely behind the utterance's last frame has nothing to compute (T = 198 in
      // a 224-row window: the 14th tile -- 1/14 of the MFMAs of every step); wave-uniform
      if (tbase + (wm * MTW + mp) * 16 >= T) continue;
      f32x4 fa[2][2];
      fa[0][0] = *reinterpret_cast<const f32x4*>(xaddr(mp, 0));
      fa[0][1] = two ? *reinterpret_cast<const f32x4*>(xaddr(mp + 1, 0)) : fa[0][0];
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        if (g + 1 < KG) {
          fa[(g + 1) & 1][0] = *reinterpret_cast<const f32x4*>(xaddr(mp, g + 1));
          if (two) fa[(g + 1) & 1][1] = *reinterpret_cast<const f32x4*>(xaddr(mp + 1, g + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[mp] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[g][s], fa[g & 1][0][s], acc[mp], 0, 0, 0);
          if (two)
            acc[mp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[g][s], fa[g & 1][1][s], acc[mp + 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    WS_RSTAMP(3 + step * 5)
    // the weight registers are free now: fetch the next step's weights under the epilogue
    if (step < 6) load_weights(step + 1);
    __syncthreads();                                    // everyone finished reading X
    WS_RSTAMP(4 + step * 5)
    // C/D layout 16x16: col = lane & 15 (time step), row = (lane >> 4) * 4 + reg (channel)
    const float* y1e = p.y1 + (m_base + tbase + tb) * ld1 + (step + 1) * W + c0;
    float* y2u = p.y2 + (m_base + tbase + tb) * ld2 + step * W + c0;
    float* xo = X + xw;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int t = tbase + tb + mt * 16;
      if (t >= 0 && t < T) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = relu_f(acc[mt][r] + bias[r]) * sc[r] + sh[r];
        if (t >= own_lo && t < own_hi) *reinterpret_cast<f32x4*>(y2u + (long long)(mt * 16) * ld2) = v;
        if (step < 6)
          *reinterpret_cast<f32x4*>(xo + mt * 16 * XS) =
              v + (PF ? y1n[PF ? mt : 0] : *reinterpret_cast<const f32x4*>(y1e + (long long)(mt * 16) * ld1));
      }
    }
    WS_RSTAMP(5 + step * 5)
    __syncthreads();
    WS_RSTAMP(6 + step * 5)
  }
  WS_RSTAMP(40)
}

// ------------------------------------------------------------------------------------------------
// Same chain on the split-binary16 back-end: X is kept in LDS as hi / lo half planes, every product
// is hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 (fp32 accumulate).  Row stride W+16 halfs
// keeps the ds_read_b128 lane groups on 16 distinct 16-B slots.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int W, int MTW>
__global__ __launch_bounds__(512) void res2_chain_f16x3_kernel(const Res2ChainParams p) {
  constexpr int NT = W / 16;
  constexpr int MW = 8 / NT;
  constexpr int XS = W + 16;            // halfs per LDS row
  constexpr int KS = 3 * W / 32;        // 32-wide k-steps per conv step
  extern __shared__ __attribute__((aligned(16))) _Float16 Xs[];

  const int b = blockIdx.x / p.tiles, tile = blockIdx.x - b * p.tiles;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % NT, wm = wave / NT;
  const int li = lane & 15, lq = lane >> 4;
  const int d = p.dil;
  // ragged batch: rows [lens[b], p.T) of utterance b are padding -- never read, never written, and seen
  // by the dilated taps as the conv's zero padding (X stays zero there)
  const int T = p.lens ? p.lens[b] : p.T;
  constexpr int CAP = MW * MTW * 16;
  const int rows_total = CAP + 2 * d;
  _Float16* Xh = Xs;
  _Float16* Xl = Xs + rows_total * XS;
  // binary16 output staging [owned rows][W] (only when p.y2h is set): the step results leave the chip as
  // coalesced 16-B stores of halfs instead of 4*MTW scalar 4-B stores per lane
  _Float16* Y16 = Xs + 2 * rows_total * XS;
  const bool half_out = p.y2h != nullptr;
  const long long m_base = (long long)b * p.T;
  // time tile (see the fp32 kernel): LDS row u <-> frame tbase + u, frames [own_lo, own_hi) are stored
  const int own_lo = tile * p.tile_rows;
  const int own_hi = own_lo + p.tile_rows < T ? own_lo + p.tile_rows : T;
  if (own_lo >= T) return;
  const int tbase = p.tiles > 1 ? own_lo - 7 * d : 0;

  for (int i = tid * 8; i < 2 * rows_total * XS; i += 512 * 8)
    *reinterpret_cast<f16x8*>(&Xs[i]) = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();
  {
    constexpr int C4 = W / 4;
    for (int i = tid; i < CAP * C4; i += 512) {
      const int u = i / C4, c = (i - u * C4) * 4, t = tbase + u;
      if (t < 0 || t >= T) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.y1 + (m_base + t) * p.ldy1 + c);
      f16x4 hi, lo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        hi[q] = (_Float16)v[q];
        lo[q] = (_Float16)(v[q] - (float)hi[q]);
      }
      *reinterpret_cast<f16x4*>(&Xh[(u + d) * XS + c]) = hi;
      *reinterpret_cast<f16x4*>(&Xl[(u + d) * XS + c]) = lo;
    }
  }
  __syncthreads();

  // The weights are the MFMA's A operand and the activations its B operand, i.e. each wavefront
  // computes a 16 (channels) x 16 (time steps) block of Y^T: a lane then owns FOUR CONSECUTIVE
  // CHANNELS (rows 4*lq .. 4*lq+3 of the C tile) of ONE time step (column li), so the whole epilogue
  // -- next-split loads, LDS hi/lo writes of the running activation, output -- moves 8/16-byte
  // vectors instead of four scalar accesses per accumulator register.
  const int co = wn * 16 + li;                  // weight row of this lane's A fragment
  const int c0 = wn * 16 + lq * 4;              // first of the lane's 4 output channels
  constexpr bool PF = MTW <= 7;
  const int tb = wm * MTW * 16 + li;            // lane's time step inside m-tile 0
  // lane (i = co, q) holds k = 32 ks + 8 q .. +7 of its weight row
  f16x8 bh[KS], bl[KS];
  auto load_weights = [&](int step) {
    const long long off = (long long)co * p.ldw + lq * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bh[ks] = *reinterpret_cast<const f16x8*>(p.wh[step] + off + ks * 32);
      bl[ks] = *reinterpret_cast<const f16x8*>(p.wl[step] + off + ks * 32);
    }
  };
  load_weights(0);
  for (int step = 0; step < 7; ++step) {
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias[step] + c0);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale[step] + c0);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift[step] + c0);
    // Row strides / the LDS write base are laundered through an empty asm once per step so the
    // per-element addresses are recomputed inside the step instead of being hoisted out of the step
    // loop (where they would live across the MFMA section and spill).
    int ld1 = p.ldy1, ld2 = p.ldy2, xw = (tb + d) * XS + c0;
    asm volatile("" : "+v"(ld1), "+v"(ld2), "+v"(xw));
    f32x4 y1n[PF ? MTW : 1];
    if (PF && step < 6) {
      const float* y1u = p.y1 + (m_base + tbase + tb) * ld1 + (step + 1) * W + c0;
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        const int t = tbase + tb + mt * 16;
        y1n[PF ? mt : 0] = (t >= 0 && t < T) ? *reinterpret_cast<const f32x4*>(y1u + (long long)(mt * 16) * ld1)
                                             : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    f32x4 acc[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // k-step ks covers k in [32 ks, 32 ks + 32) = tap (32 ks) / W, channels (32 ks) % W ...
    const int xrow = wm * MTW * 16 + li;
    auto xoff = [&](int mt, int ks) {
      const int tap = (32 * ks) / W, cg = (32 * ks) % W;
      return (xrow + mt * 16 + tap * d) * XS + cg + lq * 8;
    };
#pragma unroll
    for (int mp = 0; mp < MTW; mp += 2) {
      const bool two = mp + 1 < MTW;
      int o0 = xoff(mp, 0), o1 = xoff(two ? mp + 1 : mp, 0);
      f16x8 a0h = *reinterpret_cast<const f16x8*>(&Xh[o0]);
      f16x8 a0l = *reinterpret_cast<const f16x8*>(&Xl[o0]);
      f16x8 a1h = *reinterpret_cast<const f16x8*>(&Xh[o1]);
      f16x8 a1l = *reinterpret_cast<const f16x8*>(&Xl[o1]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        f16x8 n0h = a0h, n0l = a0l, n1h = a1h, n1l = a1l;
        if (ks + 1 < KS) {                       // fragments of the next k-step, one step ahead
          o0 = xoff(mp, ks + 1); o1 = xoff(two ? mp + 1 : mp, ks + 1);
          n0h = *reinterpret_cast<const f16x8*>(&Xh[o0]);
          n0l = *reinterpret_cast<const f16x8*>(&Xl[o0]);
          if (two) {
            n1h = *reinterpret_cast<const f16x8*>(&Xh[o1]);
            n1l = *reinterpret_cast<const f16x8*>(&Xl[o1]);
          }
        }
        // (weights, activations): small cross terms first, hi*hi last
        acc[mp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ks], a0l, acc[mp], 0, 0, 0);
        if (two) acc[mp + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ks], a1l, acc[mp + 1], 0, 0, 0);
        acc[mp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[ks], a0h, acc[mp], 0, 0, 0);
        if (two) acc[mp + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[ks], a1h, acc[mp + 1], 0, 0, 0);
        acc[mp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ks], a0h, acc[mp], 0, 0, 0);
        if (two) acc[mp + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ks], a1h, acc[mp + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a0h = n0h; a0l = n0l; a1h = n1h; a1l = n1l;
      }
    }
    if (step < 6) load_weights(step + 1);
    __syncthreads();
    const float* y1e = p.y1 + (m_base + tbase + tb) * ld1 + (step + 1) * W + c0;
    float* y2u = p.y2 + (m_base + tbase + tb) * ld2 + step * W + c0;
    _Float16* xh = Xh + xw;
    _Float16* xl = Xl + xw;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int t = tbase + tb + mt * 16;
      if (t >= 0 && t < T) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = relu_f(acc[mt][r] + bias[r]) * sc[r] + sh[r];
        if (t >= own_lo && t < own_hi) {
          if (half_out) {
            f16x4 hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (_Float16)v[r];
            if (p.y2h_direct)
              *reinterpret_cast<f16x4*>(p.y2h + (m_base + t) * p.ldy2h + step * W + c0) = hv;
            else
              *reinterpret_cast<f16x4*>(&Y16[(t - own_lo) * W + c0]) = hv;
          } else {
            *reinterpret_cast<f32x4*>(y2u + (long long)(mt * 16) * ld2) = v;
          }
        }
        if (step < 6) {
          const f32x4 x = v + (PF ? y1n[PF ? mt : 0]
                                  : *reinterpret_cast<const f32x4*>(y1e + (long long)(mt * 16) * ld1));
          f16x4 hi, lo;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            hi[r] = (_Float16)x[r];
            lo[r] = (_Float16)(x[r] - (float)hi[r]);
          }
          *reinterpret_cast<f16x4*>(&xh[mt * 16 * XS]) = hi;
          *reinterpret_cast<f16x4*>(&xl[mt * 16 * XS]) = lo;
        }
      }
    }
    __syncthreads();
    if (half_out && !p.y2h_direct) {
      // (the next write to Y16 happens after the next step's mid barrier, i.e. after this copy)
      constexpr int C8 = W / 8;
      uint16_t* dst = p.y2h + (m_base + own_lo) * p.ldy2h + step * W;
      for (int i = tid; i < (own_hi - own_lo) * C8; i += 512) {
        const int t = i / C8, c = (i - t * C8) * 8;
        *reinterpret_cast<f16x8*>(dst + (long long)t * p.ldy2h + c) =
            *reinterpret_cast<const f16x8*>(&Y16[t * W + c]);
      }
    }
  }
}

// rows a workgroup of the <W, MTW> variant computes
static constexpr int chain_cap(int W, int MTW) { return (8 / (W / 16)) * MTW * 16; }
// the two window sizes per width: the big one for full batches, the small one (128 / 112 rows) when the batch
// would leave most CUs idle -- a chain is 7 serial steps of ~10 us whatever the batch, so a lone utterance is
// spread over three or four workgroups instead of one
static int chain_mtw(int W, bool small) { return W == 64 ? (small ? 4 : 7) : (small ? 7 : 13); }
static int device_cus() { return current_device_cus(); }
// tiles per utterance / owned rows per tile for frames T (whole utterance in one workgroup when it fits)
static void chain_tiling_for(int cap, int T, int dil, int* tiles, int* tile_rows) {
  if (T <= cap) { *tiles = 1; *tile_rows = cap; return; }
  const int own = cap - 2 * 7 * dil;
  *tiles = (T + own - 1) / own;
  *tile_rows = own;
}
// small windows when the big ones would occupy at most a quarter of the chip (env WS_CHAIN_SMALL=0/1 forces)
static bool chain_use_small(int W, int B, int T, int dil) {
  static int forced = -2;
  if (forced == -2) { const char* ev = getenv("WS_CHAIN_SMALL"); forced = ev ? atoi(ev) : -1; }
  if (forced >= 0) return forced != 0;
  if (T <= chain_cap(W, chain_mtw(W, true))) return true;   // a short utterance fits the small window whole:
                                                            // 128 (112) computed rows instead of 224 (208)
  int tiles, own;
  chain_tiling_for(chain_cap(W, chain_mtw(W, false)), T, dil, &tiles, &own);
  return (long long)B * tiles * 4 <= device_cus();
}

template <int W, int MTW>
static hipError_t launch_res2_f16_variant(Res2ChainParams p, hipStream_t stream) {
  chain_tiling_for(chain_cap(W, MTW), p.T, p.dil, &p.tiles, &p.tile_rows);
  const int own = p.tiles > 1 ? p.tile_rows : p.T;
  const size_t planes = (size_t)(chain_cap(W, MTW) + 2 * p.dil) * (W + 16) * 2 * sizeof(_Float16);
  const size_t stage = p.y2h ? (size_t)own * W * sizeof(_Float16) : 0;
  // binary16 rows: staged in LDS and copied out as 16-byte vectors when that fits beside the activation planes
  // (w = 64), else stored straight from the accumulators, 8 bytes per lane (w = 128: 124 KB of planes)
  p.y2h_direct = p.y2h && planes + stage > 160 * 1024;
  const size_t lds = planes + (p.y2h_direct ? 0 : stage);
  auto kern = res2_chain_f16x3_kernel<W, MTW>;
  static size_t lds_granted[WS_MAX_DEVICES] = {};
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_granted);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(p.B * p.tiles), dim3(512), lds, stream, p);
  return hipGetLastError();
}

// the four-wavefront kernel (res2_chain4.hip): fp32, w = 64, every utterance whole in one workgroup of 9 .. 13 row tiles
static bool chain4_takes(const Res2ChainParams& p) {
  return p.prec == 0 && p.W == 64 && p.T <= 13 * 16 && p.T > chain_cap(64, chain_mtw(64, true)) && !p.force_wave8;
}

template <int W, int MTW>
static hipError_t launch_res2_variant(Res2ChainParams p, hipStream_t stream) {
  chain_tiling_for(chain_cap(W, MTW), p.T, p.dil, &p.tiles, &p.tile_rows);
  const size_t lds = (size_t)(chain_cap(W, MTW) + 2 * p.dil) * (W + 8) * sizeof(float);
  // (Round 3, measured and dropped: two copies of the running activation -- step s reads X[s & 1] and writes the
  // other, one barrier per step, every pair of row tiles finished right behind its own MFMAs, next weights
  // requested a step ahead.  Bit-identical, and not a microsecond faster: PMC says the matrix pipe is busy 0.59 of
  // this kernel's cycles either way.)
  auto kern = res2_chain_kernel<W, MTW>;
  static size_t lds_granted[WS_MAX_DEVICES] = {};
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_granted);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(p.B * p.tiles), dim3(512), lds, stream, p);
  return hipGetLastError();
}

// binary16 output (Res2ChainParams::y2h): always available on the split-f16 kernel -- staged through LDS when it
// fits, stored directly otherwise (launch_res2_f16_variant)
bool res2_half_out_supported(int W, int T, int dil) { return res2_chain_supported(W, T, dil); }

// any length: utterances beyond one workgroup's rows run as time tiles with a 7 * dil halo
bool res2_chain_supported(int W, int T, int dil) {
  if (W != 64 && W != 128) return false;
  return T >= 1 && dil >= 1 && chain_cap(W, chain_mtw(W, true)) - 2 * 7 * dil >= 32;
}

// w = 64, utterances longer than one window: the window size (MTW row tiles per wave row, 32 MTW rows) that needs the
// fewest rounds of workgroups x the time of one (measured: 56.7 us at MTW = 4, 96 us at 7 -- linear in between).  A
// 224-row window owns 196 / 182 / 168 frames at dilation 2 / 3 / 4: 64 utterances of 8 s are 320 workgroups, 1.25
// rounds on 256 CUs, i.e. two; 256-row windows make it 256.  Every tiling gives the same bits
// (test_res2_time_tiles_are_bit_identical_to_whole_utterance_windows).
static int chain_pick_mtw64(int B, int T, int dil) {
  const int cus = device_cus();
  int best = 7;
  double best_cost = 1e30;
  const int cand[4] = {7, 8, 6, 5};
  for (int i = 0; i < 4; ++i) {
    const int mtw = cand[i], cap = chain_cap(64, mtw);
    if (cap - 2 * 7 * dil < 32) continue;
    int tiles, own;
    chain_tiling_for(cap, T, dil, &tiles, &own);
    const double cost = (double)(((long long)B * tiles + cus - 1) / cus) * (4.3 + 13.1 * mtw);
    if (cost < best_cost * 0.97) { best_cost = cost; best = mtw; }      // (3 % better or stay: fewer code paths in use)
  }
  return best;
}

// ... and for w = 128 (ECAPA-1024; one row of wavefronts, 16 MTW rows): 208- or 160-row windows (347 / 268 us)
static int chain_pick_mtw128(int B, int T, int dil) {
  const int cus = device_cus();
  int best = 13;
  double best_cost = 1e30;
  const int cand[2] = {13, 10};
  for (int i = 0; i < 2; ++i) {
    const int mtw = cand[i], cap = chain_cap(128, mtw);
    if (cap - 2 * 7 * dil < 32) continue;
    int tiles, own;
    chain_tiling_for(cap, T, dil, &tiles, &own);
    const double cost = (double)(((long long)B * tiles + cus - 1) / cus) * (4.7 + 26.3 * mtw);
    if (cost < best_cost * 0.97) { best_cost = cost; best = mtw; }
  }
  return best;
}

hipError_t launch_res2_chain(const Res2ChainParams& p, hipStream_t stream) {
  if (p.B <= 0) return hipSuccess;
  if ((p.ldy1 | p.ldy2 | p.ldw) & 3) return hipErrorInvalidValue;
  if (p.y2h && (p.prec < 1 || (p.ldy2h & 7))) return hipErrorInvalidValue;
  if (!res2_chain_supported(p.W, p.T, p.dil)) return hipErrorInvalidValue;
  const bool small = chain_use_small(p.W, p.B, p.T, p.dil);
  if (p.prec >= 1) {   // (the f16 mode reuses the split kernel: more precise, same launch count)
    if (p.W == 64) {
      if (small) return launch_res2_f16_variant<64, 4>(p, stream);
      switch (chain_pick_mtw64(p.B, p.T, p.dil)) {
        case 5: return launch_res2_f16_variant<64, 5>(p, stream);
        case 6: return launch_res2_f16_variant<64, 6>(p, stream);
        case 8: return launch_res2_f16_variant<64, 8>(p, stream);
        default: return launch_res2_f16_variant<64, 7>(p, stream);
      }
    }
    if (small) return launch_res2_f16_variant<128, 7>(p, stream);
    return chain_pick_mtw128(p.B, p.T, p.dil) == 10 ? launch_res2_f16_variant<128, 10>(p, stream)
                                                    : launch_res2_f16_variant<128, 13>(p, stream);
  }
  if (!small && chain4_takes(p)) return launch_res2_chain4(p, stream);
  if (p.W == 64 && !small) {
    switch (chain_pick_mtw64(p.B, p.T, p.dil)) {
      case 5: return launch_res2_variant<64, 5>(p, stream);
      case 6: return launch_res2_variant<64, 6>(p, stream);
      case 8: return launch_res2_variant<64, 8>(p, stream);
      default: return launch_res2_variant<64, 7>(p, stream);
    }
  }
  if (p.W == 64) return small ? launch_res2_variant<64, 4>(p, stream) : launch_res2_variant<64, 7>(p, stream);
  if (small) return launch_res2_variant<128, 7>(p, stream);
  return chain_pick_mtw128(p.B, p.T, p.dil) == 10 ? launch_res2_variant<128, 10>(p, stream)
                                                  : launch_res2_variant<128, 13>(p, stream);
}

}  // namespace wsamd
