// One CAM++ dense layer as ONE kernel (fp32 back-end, utterances of <= 128 trunk frames).
//
// Replaces the four launches per layer of CAMDenseTDNNLayer.forward (wespeaker/models/campplus.py:138-170) /
// CAMLayer.forward (:86-135):
//     x -> BN-ReLU -> Conv1d(C_i -> 128, k1) -> BN-ReLU = h
//     m = sigmoid(W2 relu(W1 (mean_T(h) + segmean_100(h)) + b1) + b2)            (per utterance and 100-frame segment)
//     y = Conv1d(128 -> 32, k3, dilation d)(h) * m ;  x = cat(x, y)
// Round 3 ran them as GEMM (C_i -> 128, 128 x 128 tiles over the batch) -> context kernel -> GEMM (k3, N = 32 tiles)
// with h making a round trip through HBM: 52 layers x 3 launches of 7 - 100 us, the k3 GEMM at 0.14 of the MFMA peak
// (N = 32: 396 tiles of a few us) and 728 context launches of 7 us -- a dispatcher-bound 0.5 of the fp32 MFMA peak
// for the whole model.
//
// Here a workgroup owns one UTTERANCE (T' <= 128 rows = one 128-row tile; two workgroups share a CU):
//   1. h = the 128 x 128 tile of the 1x1 convolution: K loop over C_i in 32-wide K-tiles, A (with the pre-activation
//      BN-ReLU applied while staging) and W1 through a double-buffered LDS stage, v_mfma_f32_32x32x2_f32, one
//      32-row block x four 32-column blocks per wavefront;
//   2. bias + ReLU, rows beyond the utterance's frames zeroed, h parked in LDS (over the stages) with two zero halo
//      rows on either side; column means per segment -> the two context FCs -> the mask, all in LDS;
//   3. the dilated k3 convolution straight from the LDS copy of h (the taps are row shifts), weights prefetched from
//      L2 through a register ring, times the mask, 32 channels appended to x at the layer's channel offset.
// h never leaves the CU; 52 launches instead of 208.
#include "kernels.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace wsamd {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef WS_TRACE
// (trace builds only, tools/trace_cam_dense.py) cycle stamps of wavefronts 0 and 3 of workgroup 9, one row of 16 per
// input width: [cin / 32][wavefront 0 | wavefront 3][32]
__device__ unsigned long long g_cam_trace[32 * 64];
#define WS_CSTAMP(i)                                                                              \
  if (blockIdx.x == 9 && (threadIdx.x == 0 || threadIdx.x == 192))                                \
    g_cam_trace[((p.cin / 32) & 31) * 64 + (threadIdx.x ? 32 : 0) + (i)] = __builtin_readcyclecounter();
#define WS_CPIN(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); WS_CSTAMP(i) asm volatile("" ::: "memory"); }
#else
#define WS_CSTAMP(i)
#define WS_CPIN(i)
#endif

constexpr int CD_BK = 32;                  // K-tile
constexpr int CD_S = CD_BK + 4;            // stage row stride (floats): conflict-free 16-B fragment reads
constexpr int CD_HS = 132;                 // row stride of the LDS copy of h
constexpr int CD_HALO = 2;                 // zero rows before / behind h (dilation <= 2)
constexpr int CD_STAGE_FLOATS = 2 * 128 * CD_S;                    // A rows + W rows of one stage
constexpr int CD_MAIN_FLOATS = 2 * CD_STAGE_FLOATS;                // two stages; h (132 x 132) lives over them
constexpr int CD_SMALL_FLOATS = 2 * 2 * 128 + 2 * 128 + 2 * 64 + 2 * 32;   // partial sums, ctx, hidden, mask
static_assert((128 + 2 * CD_HALO) * CD_HS <= CD_MAIN_FLOATS, "h fits over the stages");
constexpr size_t CD_LDS_BYTES = (size_t)(CD_MAIN_FLOATS + CD_SMALL_FLOATS) * 4;

// NB: 32-row blocks of the utterance that run on v_mfma_f32_32x32x2_f32; TAIL: the last 1..4 rows (T' = 32 NB + 1..4,
// e.g. the 99 trunk frames of a 2-s utterance = 3 x 32 + 3) run on v_mfma_f32_4x4x1_16B_f32 instead of a fourth, 29/32
// empty block: the instruction multiplies 16 independent 4 x 4 blocks, here 16 groups of four channels x the four
// tail rows, one k per instruction -- the same 64 FLOP per cycle, 3 % of the rows instead of 25 % of the work.
// Wavefront w owns the 32 channels 32 w .. 32 w + 31 of h for all NB row blocks (one W fragment, NB A fragments and
// 4 NB MFMAs per 8 k); wavefronts 0 / 1 also the tail rows of channels 0..63 / 64..127.
template <int NB, bool TAIL>
__device__ __forceinline__ void cam_dense_layer_body(const CamDenseParams& p, float* const lds) {
  constexpr int ROWS = 32 * NB + (TAIL ? 4 : 0);   // rows of h that are computed (>= T')
  constexpr int GM = 4 * NB;                       // 32x32x2 MFMAs per k-group and wavefront
  constexpr int NFR = 1 + NB;                      // fragment reads per k-group
  constexpr int NFRT = NFR + (TAIL ? 4 : 0);       // ... incl. the tail's
  float* const small = lds + CD_MAIN_FLOATS;
  float* const part = small;                       // [2 row parities][2 segments][128]
  float* const ctx = part + 2 * 2 * 128;           // [2 segments][128]
  float* const hid = ctx + 2 * 128;                // [2][64]
  float* const maskv = hid + 2 * 64;               // [2][32]
  // (an opaque copy of the thread index: inside cam_dense_block_kernel's layer loop everything derived from it would
  // otherwise be hoisted out of the loop and held in registers across the layers -- 416 B of scratch)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x;
  const int Tp = p.Tp;
  const int len = p.lens ? p.lens[b] : Tp;         // this utterance's frames (<= Tp <= ROWS)
  const long long row0 = (long long)b * Tp;
  const int nk = p.cin / CD_BK;
  WS_CSTAMP(0)
  // Every layer's weights are cold (28 MB of parameters and 18 GB of traffic per forward lie between two uses of a
  // layer), and under the load of 512 K loops a dependent global access costs 2 - 4 k cycles even when it hits L2.
  // The phases behind the K loop used to start with one each -- b1, the tail's b1, Wl, cw1 / cb1, cw2 / cb2 -- and the
  // k3 convolution walks Wl with eight steps of lookahead: 50 k cycles per layer behind the K loop for 12 k of MFMAs
  // (tools/trace_cam_dense.py).  Now nothing behind the K loop waits for a first touch:
  //   * b1 / cb1 / cb2 (224 floats) are requested first of all and parked in LDS in front of the first barrier;
  //   * one load per 128-B line of Wl, cw1, cw2 is sent behind the first two K-tiles and retires in the shadow of the
  //     K loop (L2 warm-up; the values are not used);
  //   * cw1 / cw2 go to registers at the end of the K loop (the staging registers are free by then) and the barriers
  //     of phase 2 wait for LDS only, not for them.
  float* const b1s = part;                         // [128]   (the `part` region is otherwise unused)
  float* const cb1s = part + 128;                  // [64]
  float* const cb2s = part + 192;                  // [32]
  float small_v = 0.f;
  if (tid < 224) small_v = tid < 128 ? p.b1[tid] : tid < 192 ? p.cb1[tid - 128] : p.cb2[tid - 192];

  // ------------------------------------------------------------------ 1. h = relu(W1 . relu(bn(x)) + b1)
  const int kc = tid & 7, r0 = tid >> 3;           // this thread's 16-B chunk column and first row of a stage
  const float* a_ptr[4];
  const float* w_ptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = r0 + 32 * i;
    row = row < Tp ? row : Tp - 1;                 // (rows beyond the utterance are zeroed in h: any finite value)
    a_ptr[i] = p.X + (row0 + row) * p.ldx + kc * 4;
    w_ptr[i] = p.W1 + (long long)(r0 + 32 * i) * p.ldw1 + kc * 4;
  }
  const float* ps_ptr = p.pre_s + kc * 4;
  const float* pb_ptr = p.pre_b + kc * 4;
  // two sets of staging registers: K-tile kt + 2 is requested at the top of K-tile kt and staged into LDS during
  // K-tile kt + 1
  f32x4 ra[2][4], rw[2][4], s4[2], b4[2];
  auto load_tile = [&](int set) {
    s4[set] = *reinterpret_cast<const f32x4*>(ps_ptr);
    b4[set] = *reinterpret_cast<const f32x4*>(pb_ptr);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[set][i] = *reinterpret_cast<const f32x4*>(a_ptr[i]);
      rw[set][i] = *reinterpret_cast<const f32x4*>(w_ptr[i]);
      a_ptr[i] += CD_BK; w_ptr[i] += CD_BK;
    }
    ps_ptr += CD_BK; pb_ptr += CD_BK;
  };
  // staging piece j of the K-tile in register set `set`: 0..3 = pre-activation of A chunk j, 4..7 = its LDS store,
  // 8..11 = the W chunks (each small enough for the shadow of one MFMA)
  auto store_piece = [&](int buf, int set, int j) {
    float* As = lds + buf * CD_STAGE_FLOATS;
    float* Ws = As + 128 * CD_S;
    if (j < 4) {
      f32x4 v = ra[set][j] * s4[set] + b4[set];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = relu_f(v[e]);
      ra[set][j] = v;
    } else if (j < 8) {
      *reinterpret_cast<f32x4*>(&As[(r0 + 32 * (j - 4)) * CD_S + kc * 4]) = ra[set][j - 4];
    } else if (j < 12) {
      *reinterpret_cast<f32x4*>(&Ws[(r0 + 32 * (j - 8)) * CD_S + kc * 4]) = rw[set][j - 8];
    }
  };
  f32x16 acc[NB];
#pragma unroll
  for (int mb = 0; mb < NB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
  f32x4 tacc = {0.f, 0.f, 0.f, 0.f};               // TAIL (wavefronts 0, 1): channels 64 wave + 4 (lane >> 2) .. + 3 of
                                                   // tail row (lane & 3)
  const bool tail_wave = TAIL && wave < 2;         // (uniform)
  // fragments of one k-group (8 k): this wavefront's 32 rows of W, the NB row blocks of A; two register slots
  f32x4 fw[2], fa[2][NB];
  f32x4 tw[2][2], tx[2][2];                        // TAIL: W row 64 wave + lane and A row 32 NB + (lane & 3), 8 k each
  auto frag_read = [&](int buf, int g, int slot, int j) {        // j = 0: W, 1..NB: A block j - 1, then the tail's four
    const float* st = lds + buf * CD_STAGE_FLOATS;
    if (j == 0) fw[slot] = *reinterpret_cast<const f32x4*>(st + 128 * CD_S + (32 * wave + li) * CD_S + lh * 4 + g * 8);
    else if (j <= NB) fa[slot][j - 1] = *reinterpret_cast<const f32x4*>(st + (32 * (j - 1) + li) * CD_S + lh * 4 + g * 8);
    else if (TAIL && tail_wave) {
      const int q = j - NB - 1;                    // 0, 1: W halves; 2, 3: A halves
      if (q < 2) tw[slot][q] = *reinterpret_cast<const f32x4*>(st + 128 * CD_S + (64 * wave + lane) * CD_S + g * 8 + 4 * q);
      else tx[slot][q - 2] = *reinterpret_cast<const f32x4*>(st + (32 * NB + (lane & 3)) * CD_S + g * 8 + 4 * (q - 2));
    }
  };
  // the MFMAs of the k-group in `slot`; filler(i) is issued behind MFMA i (an fp32 32x32x2 MFMA holds the matrix pipe
  // for 64 cycles: LDS reads / stores and the pre-activation maths of the next K-tile ride in its shadow -- left to
  // itself hipcc puts them between the K-tiles, where the two workgroups of a CU, running in lock step, both idle
  // the pipe)
  auto mma = [&](int slot, auto&& filler) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[slot][s], fa[slot][mb][s], acc[mb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        filler(s * NB + mb);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
    for (int i = GM; i < NFRT; ++i) filler(i);     // (one row block + tail: more fragment reads than MFMA gaps)
    if (TAIL && tail_wave) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          tacc = __builtin_amdgcn_mfma_f32_4x4x1f32(tw[slot][q][s], tx[slot][q][s], tacc, 0, 0, 0);
    }
  };
  // the 12 staging pieces go into the gaps of k-groups 1 and 2 that carry no fragment read (NB >= 3: all of them;
  // smaller utterances finish the rest in front of the barrier)
  constexpr int GAP = GM > NFRT ? GM - NFRT : 0;   // free gaps per k-group
  load_tile(0);
  if (nk > 1) load_tile(1);
  float warm[3];
  {
    const int i1 = tid + 256;                      // Wl: 32 rows x 384 floats = 32 x 12 lines
    warm[0] = p.Wl[(long long)(tid / 12) * p.ldwl + (tid % 12) * 32];
    warm[1] = i1 < 32 * 12 ? p.Wl[(long long)(i1 / 12) * p.ldwl + (i1 % 12) * 32] : p.cw2[((i1 - 32 * 12) & 63) * 32];   // (64 lines: stay inside cw2)
    warm[2] = p.cw1[tid * 32];                     // cw1 [64][128] = 256 lines; cw2 [32][64] = 64 lines (above)
  }
  __builtin_amdgcn_sched_barrier(0);
  if (tid < 224) part[tid] = small_v;
#pragma unroll
  for (int j = 0; j < 12; ++j) store_piece(0, 0, j);
  __syncthreads();
  WS_CSTAMP(1)
#pragma unroll
  for (int j = 0; j < NFRT; ++j) frag_read(0, 0, 0, j);
  int buf = 0;
  // K-tile kt: its MFMAs; K-tile kt + 1 (register set (kt + 1) & 1) goes to LDS stage buf ^ 1 behind them; K-tile
  // kt + 2 is requested into the set K-tile kt came from
  auto ktile = [&](int kt, auto par_tag) {
    constexpr int PAR = decltype(par_tag)::value;  // kt & 1
    const bool more = kt + 1 < nk;                 // (uniform)
    if (kt + 2 < nk) load_tile(PAR);
    __builtin_amdgcn_sched_barrier(0);
    mma(0, [&](int i) { if (i < NFRT) frag_read(buf, 1, 1, i); });
    mma(1, [&](int i) {                            // g1: fragments of g2, then staging pieces 0 .. GAP - 1
      if (i < NFRT) frag_read(buf, 2, 0, i);
      else if (more) store_piece(buf ^ 1, PAR ^ 1, i - NFRT);
    });
    mma(0, [&](int i) {                            // g2: fragments of g3, then staging pieces GAP .. 2 GAP - 1
      if (i < NFRT) frag_read(buf, 3, 1, i);
      else if (more) store_piece(buf ^ 1, PAR ^ 1, GAP + i - NFRT);
    });
    if (more) {
#pragma unroll
      for (int j = 2 * GAP; j < 12; ++j) store_piece(buf ^ 1, PAR ^ 1, j);
    }
    // stage buf^1 complete; nobody reads stage buf any more (g3 is in registers).  NOT __syncthreads(): its fence
    // waits for vmcnt(0), i.e. for the global loads of K-tile kt + 2 that were just sent ahead
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    buf ^= 1;
    mma(1, [&](int i) { if (more && i < NFRT) frag_read(buf, 0, 0, i); });
  };
  for (int kt = 0; kt < nk; kt += 2) {
    ktile(kt, std::integral_constant<int, 0>{});
    if (kt + 1 < nk) ktile(kt + 1, std::integral_constant<int, 1>{});
  }
  WS_CSTAMP(2)
  asm volatile("" ::"v"(warm[0]), "v"(warm[1]), "v"(warm[2]));   // (the warm-up loads end here)
  // context FC weights of this thread: FC1 row j = tid >> 2, quarter q = tid & 3 (32 floats); FC2 row o, quarter q
  // (16 floats) -- in flight during phase 2's first steps
  f32x4 cw1r[8], cw2r[4];
  {
    const float* w1p = p.cw1 + (tid >> 2) * 128 + (tid & 3) * 32;
    const float* w2p = p.cw2 + ((tid >> 2) & 31) * 64 + (tid & 3) * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) cw1r[k] = *reinterpret_cast<const f32x4*>(w1p + 4 * k);
#pragma unroll
    for (int k = 0; k < 4; ++k) cw2r[k] = *reinterpret_cast<const f32x4*>(w2p + 4 * k);
  }
  __builtin_amdgcn_sched_barrier(0);
  // every wavefront has left the stages: h goes over them.  (LDS-only barriers from here to the k3 convolution:
  // __syncthreads() would wait for the loads just sent)
#define WS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  WS_LDS_BARRIER();
  WS_CSTAMP(3)

  // ------------------------------------------------------------------ 2. h -> LDS, context mask
  float* const Hs = lds;                           // [ROWS + 2 halo][CD_HS], row t at index t + CD_HALO
  {
    // (the bias fragments are read up front: a ds_read between two ds_writes is not reordered by hipcc -- the 12
    // read -> wait -> write round trips of the first version took 7 - 10 k cycles)
    f32x4 bias4[4], biast = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4*>(b1s + 32 * wave + 8 * g + 4 * lh);
    if (TAIL && tail_wave) biast = *reinterpret_cast<const f32x4*>(b1s + 64 * wave + 4 * (lane >> 2));
    WS_CPIN(16)
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
      const int t = 32 * mb + li;
      const bool valid = t < len;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = 32 * wave + 8 * g + 4 * lh;
        const f32x4 bias = bias4[g];
        // (two selects, kept apart by the opaque copy: merged, hipcc builds `valid && !(x < 0)` with one
        // v_cmp -> s_and_b64 -> v_cndmask round trip through the scalar unit per ELEMENT, all 48 of them serialised on
        // one SGPR pair -- 5 - 7 k cycles for this loop, measured; apart, the row's mask is one SGPR pair per block)
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float z = relu_f(acc[mb][4 * g + r] + bias[r]);
          asm volatile("" : "+v"(z));
          v[r] = valid ? z : 0.f;
        }
        *reinterpret_cast<f32x4*>(&Hs[(t + CD_HALO) * CD_HS + n0]) = v;
      }
    }
    WS_CPIN(17)
    if (TAIL && tail_wave) {
      const int t = 32 * NB + (lane & 3), n0 = 64 * wave + 4 * (lane >> 2);
      const bool valid = t < len;
      const f32x4 bias = biast;
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float z = relu_f(tacc[r] + bias[r]);
        asm volatile("" : "+v"(z));
        v[r] = valid ? z : 0.f;
      }
      *reinterpret_cast<f32x4*>(&Hs[(t + CD_HALO) * CD_HS + n0]) = v;
    }
    // halo rows: 0, 1 and ROWS + 2, ROWS + 3 (4 rows x 128 floats)
    if (tid < 128) {
      const int hr = tid >> 5, c = (tid & 31) * 4;
      *reinterpret_cast<f32x4*>(&Hs[(hr < 2 ? hr : ROWS + hr) * CD_HS + c]) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  WS_CPIN(18)
  WS_CSTAMP(7)
  // the first weight fragments of the k3 convolution: in flight while the mask is computed
  constexpr int RING = 8;
  f32x4 wl[RING];
  const bool tail3 = TAIL && wave == NB;           // (uniform) the wavefront that convolves the tail rows
  // main: row o = li, K offset 8 j + 4 lh; tail: row o = lane & 31, K offset 4 j, lanes 32..63 the second half of K
  const float* wl_ptr = tail3 ? p.Wl + (long long)(lane & 31) * p.ldwl + 192 * lh : p.Wl + (long long)li * p.ldwl + lh * 4;
  const int wl_step = tail3 ? 4 : 8;
#pragma unroll
  for (int j = 0; j < RING; ++j) wl[j] = *reinterpret_cast<const f32x4*>(wl_ptr + wl_step * j);
  WS_CPIN(19)
  WS_LDS_BARRIER();
  WS_CSTAMP(4)
  const int nseg = len > 100 ? 2 : 1;              // (T' <= 128: at most two 100-frame segments)
  {
    // column sums of h over segment 0 (frames < 100) and segment 1, then ctx[s][c] = mean over the utterance + mean
    // over segment s.  Wavefront w takes columns 32 w .. 32 w + 31: lane (column quad cq, row class rsub of 16) reads
    // rows 16 i + 4 k + rsub with ds_read_b128 -- the 16 lanes of one LDS pass touch two rows 8 apart, i.e. 32 banks
    // apart -- and three butterfly steps add the eight row classes.  (Rows in [len, ROWS) are zero in Hs, row ROWS is
    // the zero halo: no predicate.)  One pass of reads in flight instead of 50 dependent scalar ones per thread.
    const int cq = lane & 7, rsub = 8 * ((lane >> 3) & 1) + (lane >> 4);
    const float* hcol = Hs + CD_HALO * CD_HS + 32 * wave + 4 * cq;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (ROWS + 15) / 16; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = 16 * i + 4 * k + rsub;
        const int tt = 16 * i + 4 * k + 11 < ROWS ? t : (t < ROWS ? t : ROWS);
        const f32x4 v = *reinterpret_cast<const f32x4*>(hcol + tt * CD_HS);
        if (16 * i + 4 * k + 11 < 100) {
          s0 += v;
        } else {
          const bool lo = t < 100;
#pragma unroll
          for (int r = 0; r < 4; ++r) { s0[r] += lo ? v[r] : 0.f; s1[r] += lo ? 0.f : v[r]; }
        }
      }
#pragma unroll
    for (int m = 8; m <= 32; m <<= 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] += __shfl_xor(s0[r], m, 64);
        s1[r] += __shfl_xor(s1[r], m, 64);
      }
    if (lane < 8) {
      const int n0 = len < 100 ? len : 100, n1 = len - 100;
      f32x4 c0, c1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float tot = s0[r] + s1[r];
        c0[r] = n0 > 0 ? tot / (float)len + s0[r] / (float)n0 : 0.f;
        c1[r] = n1 > 0 ? tot / (float)len + s1[r] / (float)n1 : 0.f;
      }
      *reinterpret_cast<f32x4*>(ctx + 32 * wave + 4 * cq) = c0;
      *reinterpret_cast<f32x4*>(ctx + 128 + 32 * wave + 4 * cq) = c1;
    }
  }
  WS_CSTAMP(8)
  WS_LDS_BARRIER();
  WS_CSTAMP(9)
  for (int s = 0; s < nseg; ++s) {                 // hid[s][j] = relu(cw1[j] . ctx[s] + cb1[j]): 4 lanes per j
    const int j = tid >> 2, q = tid & 3;
    const float* cx = ctx + s * 128 + q * 32;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(cx + 4 * k);
      v += cw1r[k][0] * c4[0] + cw1r[k][1] * c4[1] + cw1r[k][2] * c4[2] + cw1r[k][3] * c4[3];
    }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    if (q == 0) hid[s * 64 + j] = relu_f(v + cb1s[j]);
  }
  WS_CSTAMP(10)
  WS_LDS_BARRIER();
  WS_CSTAMP(11)
  {                                                // mask[s][o] = sigmoid(cw2[o] . hid[s] + cb2[o]): 4 lanes per (s, o)
    const int sg = tid >> 7, o = (tid >> 2) & 31, q = tid & 3;
    const float* hv = hid + sg * 64 + q * 16;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 h4 = *reinterpret_cast<const f32x4*>(hv + 4 * k);
      v += cw2r[k][0] * h4[0] + cw2r[k][1] * h4[1] + cw2r[k][2] * h4[2] + cw2r[k][3] * h4[3];
    }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    // (segment 1 of a one-segment utterance: hid[1] was never written -- whatever v is, the mask is 0)
    if (q == 0) maskv[sg * 32 + o] = sg < nseg ? 1.f / (1.f + expf(-(v + cb2s[o]))) : 0.f;
  }
  WS_CSTAMP(12)
  WS_LDS_BARRIER();
  WS_CSTAMP(5)

  // ------------------------------------------------------------------ 3. y = conv_k3(h) * mask, appended to x
  if (wave < NB) {                                 // row block `wave`: 32 rows x 32 output channels, K = 3 x 128
    f32x16 y;
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = 0.f;
    const int t = 32 * wave + li;
    const float* hrow = Hs + (t + CD_HALO) * CD_HS + lh * 4;
#pragma unroll
    for (int j = 0; j < 48; ++j) {                 // K in steps of 8
      const int tap = j >> 4, g = j & 15;
      const f32x4 fa3 = *reinterpret_cast<const f32x4*>(hrow + (tap - 1) * p.dil * CD_HS + g * 8);
      const f32x4 fw3 = wl[j % RING];
      if (j + RING < 48) wl[j % RING] = *reinterpret_cast<const f32x4*>(wl_ptr + 8 * (j + RING));
#pragma unroll
      for (int s = 0; s < 4; ++s) y = __builtin_amdgcn_mfma_f32_32x32x2f32(fw3[s], fa3[s], y, 0, 0, 0);
      if (j == 15) { WS_CSTAMP(13) }
      if (j == 31) { WS_CSTAMP(14) }
    }
    WS_CSTAMP(15)
    if (t < Tp) {
      const float* mrow = maskv + (t >= 100 ? 32 : 0);
      float* orow = p.Xout + (row0 + t) * p.ldx + p.c_off;
      const bool valid = t < len;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int o0 = 8 * g + 4 * lh;
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = valid ? y[4 * g + r] * mrow[o0 + r] : 0.f;
        *reinterpret_cast<f32x4*>(orow + o0) = v;
      }
    }
  } else if (TAIL && tail3) {
    // the tail rows: 4x4x1 blocks = (4 output channels) x (4 rows), one k per instruction.  The 8 channel groups fill
    // lanes 0..31; lanes 32..63 take the second half of K (steps 48..95 of 96), and each half runs two accumulation
    // chains with the next activation fragment requested ahead: the loop was 96 steps of (LDS round trip -> four
    // dependent MFMAs), 22 - 31 k cycles -- the last wavefront of every workgroup to finish.
    f32x4 ya = {0.f, 0.f, 0.f, 0.f}, yb = {0.f, 0.f, 0.f, 0.f};
    const int t = 32 * NB + (lane & 3);
    const float* hrow = Hs + (t + CD_HALO) * CD_HS;
    const int dstep = p.dil * CD_HS;
    auto hoff = [&](int jj) { const int j = jj + 48 * lh; return ((j >> 5) - 1) * dstep + (j & 31) * 4; };
    f32x4 fa_n = *reinterpret_cast<const f32x4*>(hrow + hoff(0));
#pragma unroll
    for (int jj = 0; jj < 48; ++jj) {              // K in steps of 4
      const f32x4 fa3 = fa_n;
      if (jj + 1 < 48) fa_n = *reinterpret_cast<const f32x4*>(hrow + hoff(jj + 1));
      const f32x4 fw3 = wl[jj % RING];
      if (jj + RING < 48) wl[jj % RING] = *reinterpret_cast<const f32x4*>(wl_ptr + 4 * (jj + RING));
      ya = __builtin_amdgcn_mfma_f32_4x4x1f32(fw3[0], fa3[0], ya, 0, 0, 0);
      yb = __builtin_amdgcn_mfma_f32_4x4x1f32(fw3[1], fa3[1], yb, 0, 0, 0);
      ya = __builtin_amdgcn_mfma_f32_4x4x1f32(fw3[2], fa3[2], ya, 0, 0, 0);
      yb = __builtin_amdgcn_mfma_f32_4x4x1f32(fw3[3], fa3[3], yb, 0, 0, 0);
    }
    f32x4 y4 = ya + yb;
#pragma unroll
    for (int r = 0; r < 4; ++r) y4[r] += __shfl_xor(y4[r], 32, 64);
    if (t < Tp && lane < 32) {
      const float* mrow = maskv + (t >= 100 ? 32 : 0);
      const int o0 = 4 * (lane >> 2);
      const bool valid = t < len;
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = valid ? y4[r] * mrow[o0 + r] : 0.f;
      *reinterpret_cast<f32x4*>(p.Xout + (row0 + t) * p.ldx + p.c_off + o0) = v;
    }
  }
  WS_CSTAMP(6)
#undef WS_LDS_BARRIER
}

template <int NB, bool TAIL>
__global__ __launch_bounds__(256, 2) void cam_dense_layer_kernel(const CamDenseParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  cam_dense_layer_body<NB, TAIL>(p, lds);
}

// A whole dense block in one launch (CamDenseBlockParams, kernels.h): the workgroup of an utterance runs the layers
// back to back.  Between two layers one __syncthreads(): its workgroup-scope fence completes the stores of the 32 new
// channels before any wavefront of the workgroup loads them (same CU, same L1: nothing to invalidate), and every
// wavefront has left the LDS copy of h before the next layer stages over it.  3 launches per forward instead of 52.
// Measured (tools/trace_cam_dense.py, 512 utterances): the layers take the same cycles as in 52 launches -- the
// phases behind the K loop are NOT instruction-cache misses of code a launch runs once, and the two workgroups of a CU
// stay in lock step; starting the second one half a layer late (tried: WS_CAM_STAGGER, 30 / 50 / 80 %) changes
// nothing either, because a workgroup alone runs its K loop at 4.7 k cycles per K-tile against 6.9 k for the pair
// (0.70 vs 0.96 of the MFMA rate): what the partner's idle phases give back, the solo K loop loses.  What the launch
// count does buy is the latency of small batches (49 launch gaps less per utterance).
template <int NB, bool TAIL>
__global__ __launch_bounds__(256, 2) void cam_dense_block_kernel(const CamDenseBlockParams bp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int l = 0; l < bp.n_layers; ++l) {
    CamDenseParams p = bp.base;
    const CamDenseLayerW& w = bp.layers[l];
    p.cin = bp.base.cin + 32 * l;
    p.c_off = p.cin;
    p.pre_s = w.pre_s; p.pre_b = w.pre_b;
    p.W1 = w.W1; p.b1 = w.b1; p.ldw1 = w.ldw1;
    p.Wl = w.Wl; p.ldwl = w.ldwl;
    p.cw1 = w.cw1; p.cb1 = w.cb1; p.cw2 = w.cw2; p.cb2 = w.cb2;
    cam_dense_layer_body<NB, TAIL>(p, lds);
    __syncthreads();
  }
}

template <int NB, bool TAIL>
hipError_t launch_cam(const CamDenseParams& p, int B, hipStream_t stream) {
  static size_t granted[WS_MAX_DEVICES] = {};
  auto kern = cam_dense_layer_kernel<NB, TAIL>;
  hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), CD_LDS_BYTES, granted);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), CD_LDS_BYTES, stream, p);
  return hipGetLastError();
}

hipError_t launch_cam_variant(int nb, bool tail, const CamDenseParams& p, int B, hipStream_t stream) {
  if (tail) {
    if (nb == 1) return launch_cam<1, true>(p, B, stream);
    if (nb == 2) return launch_cam<2, true>(p, B, stream);
    if (nb == 3) return launch_cam<3, true>(p, B, stream);
    return hipErrorInvalidValue;
  }
  if (nb == 1) return launch_cam<1, false>(p, B, stream);
  if (nb == 2) return launch_cam<2, false>(p, B, stream);
  if (nb == 3) return launch_cam<3, false>(p, B, stream);
  if (nb == 4) return launch_cam<4, false>(p, B, stream);
  return hipErrorInvalidValue;
}

template <int NB, bool TAIL>
hipError_t launch_cam_block(const CamDenseBlockParams& bp, int B, hipStream_t stream) {
  static size_t granted[WS_MAX_DEVICES] = {};
  auto kern = cam_dense_block_kernel<NB, TAIL>;
  hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), CD_LDS_BYTES, granted);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), CD_LDS_BYTES, stream, bp);
  return hipGetLastError();
}

hipError_t launch_cam_block_variant(int nb, bool tail, const CamDenseBlockParams& bp, int B, hipStream_t stream) {
  if (tail) {
    if (nb == 1) return launch_cam_block<1, true>(bp, B, stream);
    if (nb == 2) return launch_cam_block<2, true>(bp, B, stream);
    if (nb == 3) return launch_cam_block<3, true>(bp, B, stream);
    return hipErrorInvalidValue;
  }
  if (nb == 1) return launch_cam_block<1, false>(bp, B, stream);
  if (nb == 2) return launch_cam_block<2, false>(bp, B, stream);
  if (nb == 3) return launch_cam_block<3, false>(bp, B, stream);
  if (nb == 4) return launch_cam_block<4, false>(bp, B, stream);
  return hipErrorInvalidValue;
}

}  // namespace

#ifdef WS_TRACE
unsigned long long* cam_trace_buffer_address() {
  unsigned long long* q = nullptr;
  (void)hipGetSymbolAddress(reinterpret_cast<void**>(&q), HIP_SYMBOL(g_cam_trace));
  return q;
}
#endif

bool cam_dense_fused_applies(int Tp, int cin, int dil) {
  return Tp >= 1 && Tp <= 128 && cin >= CD_BK && cin % CD_BK == 0 && dil >= 1 && dil <= CD_HALO;
}

hipError_t launch_cam_dense_block(const CamDenseBlockParams& bp, int B, hipStream_t stream) {
  if (B <= 0 || bp.n_layers <= 0) return hipSuccess;
  const CamDenseParams& p = bp.base;
  if (bp.n_layers > WS_CAM_MAX_LAYERS || p.X != p.Xout || p.c_off != p.cin || (p.ldx & 31) || (p.cin & 31) ||
      p.cin + 32 * bp.n_layers > p.ldx || (reinterpret_cast<uintptr_t>(p.X) & 127))
    return hipErrorInvalidValue;                   // (in place, each layer's 32 channels = one whole 128-B line per row)
  for (int l = 0; l < bp.n_layers; ++l) {
    const CamDenseLayerW& w = bp.layers[l];
    if (!cam_dense_fused_applies(p.Tp, p.cin + 32 * l, p.dil) || (w.ldw1 & 3) || (w.ldwl & 3) || w.ldwl < 3 * 128)
      return hipErrorInvalidValue;
  }
  if (dispatch_log_enabled()) {
    char key[200];
    snprintf(key, sizeof(key), "CAM dense block: utterances=%d T'=%d Cin=%d..%d dil=%d prec=0%s -> cam_dense_block_kernel (%d layers)",
             B, p.Tp, p.cin, p.cin + 32 * (bp.n_layers - 1), p.dil, p.lens ? " +mask" : "", bp.n_layers);
    dispatch_log_note_text(key);
  }
  const int q = p.Tp / 32, r = p.Tp % 32;
  const bool tail = r >= 1 && r <= 4 && q >= 1;
  const int nb = tail || r == 0 ? q : q + 1;
  return launch_cam_block_variant(nb, tail, bp, B, stream);
}

hipError_t launch_cam_dense_layer(const CamDenseParams& p, int B, hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  if (!cam_dense_fused_applies(p.Tp, p.cin, p.dil) || (p.ldx & 3) || (p.c_off & 3) || (p.ldw1 & 3) || (p.ldwl & 3) ||
      p.ldwl < 3 * 128)
    return hipErrorInvalidValue;
  if (dispatch_log_enabled()) {
    char key[200];
    snprintf(key, sizeof(key), "CAM dense layer: utterances=%d T'=%d Cin=%d dil=%d prec=0%s -> cam_dense_layer_kernel", B, p.Tp,
             p.cin, p.dil, p.lens ? " +mask" : "");
    dispatch_log_note_text(key);
  }
  // row blocks: T' = 32 q + r; r in 1..4 -> q blocks + the 4x4x1 tail, else ceil(T' / 32) blocks
  const int q = p.Tp / 32, r = p.Tp % 32;
  const bool tail = r >= 1 && r <= 4 && q >= 1;
  const int nb = tail || r == 0 ? q : q + 1;
  return launch_cam_variant(nb, tail, p, B, stream);
}

}  // namespace wsamd
