// Persistent fp32 GEMM for the plain 1x1 layers (the dominant class of the parity-grade back-end).
//
// Replaces the conv1d(k=1) / linear ATen calls of the reference forward whose epilogue is only
// bias -> ReLU -> BN affine (wespeaker/models/ecapa_tdnn.py:85-106 Conv1dReluBn as used at :133-157 for the
// two 1x1 convolutions of every SE-Res2Block, and :196 for the cat convolution): 6 x 512^2 + 1536^2 of
// ECAPA-512 = 87 % of its FLOPs.  Everything else (taps, residuals, masks, pooling, split-K) stays on
// conv_gemm.hip.
//
// Why a separate kernel.  The 128x128 tile kernel of conv_gemm.hip runs its K loop at
// ~0.89 of the fp32 MFMA rate, but a K = 512 tile is only 16 K-tiles long: prologue (exposed first loads),
// epilogue (all 512 resident workgroups store their 64 KB at the same moment) and the 3.09-round tile count
// take the layer to 0.67-0.72.  Here ONE workgroup per CU stays resident and walks its tiles:
//   * the operands are a continuous stream of 32-KB K-tiles (128 rows of A + 128 rows of W, 128 B each)
//     copied global -> LDS by LDS-DMA (global_load_lds_dwordx4) into a ring of three stages, two K-tiles
//     ahead of the MFMAs and straight across tile boundaries: no prologue per tile.  Unpadded 128-B rows,
//     the 16-B chunk index XOR-swizzled with (row >> 1) & 7 on the SOURCE address and on the fragment reads
//     (every ds_read_b128 lane group then hits 16 distinct 16-B slots);
//   * every non-MFMA instruction is issued behind one MFMA (an fp32 MFMA holds the pipe for 64 cycles):
//     DMA pieces, fragment reads, the barrier's waits, and the whole epilogue;
//   * the LAST K-tile of a tile runs block-major (all 16 k-steps of one 32x32 accumulator block, then the
//     next block: a dependent fp32 MFMA chain issues at the full rate), so a finished block's epilogue --
//     wave-private 32x36 LDS transpose, bias / ReLU / BN, 128-B row segments to HBM, column sums -- runs in the
//     shadow of the next block's MFMAs (only the last block's is exposed).  Same k order per accumulator as
//     conv_gemm.hip: the outputs are bit-identical;
//   * only whole rounds of tiles are taken (the dispatcher hands the remaining rows to the 64x64 kernel),
//     XCD-aware order inside a round (an XCD's 32 workgroups share A row panels in its private L2).
// Round 5: the same kernel also takes the 3x3 / stride-1 convolutions whose K-tiles lie inside one filter tap (CONV:
// wespeaker/models/resnet.py:35-107, stages 2-4) and, on a 256x64 tile (WNP = 2), the 64-channel layers.
#include "kernels.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace wsamd {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int S_BK = 32;
constexpr int S_SCR_STRIDE = 36;                              // floats per row of the transpose scratch
constexpr int S_SCR_BYTES = 32 * S_SCR_STRIDE * 4;            // per wavefront

__device__ __forceinline__ void s_dma_16B(const void* g, void* lds_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
#endif
}
#ifdef WS_TRACE
__device__ unsigned long long g_stream_trace[512];      // [wavefront 0 | wavefront 4] x 256 slots
__device__ int g_trace_split = 1;
#endif
template <int N>
__device__ __forceinline__ void s_wait_lds_vm_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  // this wavefront's fragment reads of the stage are complete, its DMA pieces of the next stage have landed
  // (all but the N newest VMEM operations), then the workgroup barrier
#if defined(WS_TRACE)               // the wait and the rendezvous stamped separately (slot 176: after the wait)
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
  if (g_trace_split && blockIdx.x == 9 && (threadIdx.x == 0 || threadIdx.x == 256))
    g_stream_trace[176 + threadIdx.x] = __builtin_readcyclecounter();
  asm volatile("s_barrier" ::: "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
#endif
}

#ifdef WS_TRACE
#define WS_SSTAMP(i)                                                                   \
  { const int _i = (i); if (blockIdx.x == 9 && (threadIdx.x == 0 || threadIdx.x == 256) && _i < 128) g_stream_trace[_i + threadIdx.x] = __builtin_readcyclecounter(); }
// fine stamps inside ONE K-tile of each kind (slots 128..: regular, 144..: first, 160..: last)
#define WS_FSTAMP(on, slot)                                                            \
  if ((on) && blockIdx.x == 9 && (threadIdx.x == 0 || threadIdx.x == 256)) g_stream_trace[(slot) + threadIdx.x] = __builtin_readcyclecounter();
#else
#define WS_SSTAMP(i)
#define WS_FSTAMP(on, slot)
#endif

// Wavefront grid WM x WN, every wavefront 64 rows x 32 TN columns (TM = 2 row blocks); BM = 64 WM rows, BN = 128:
//   <2, 2>     four wavefronts of 64x64, 128x128 tile (one wavefront per SIMD)
//   <2, 1>     eight wavefronts of 64x32, 128x128 tile
//   <4, 2>     eight wavefronts of 64x64, 256x128 tile, ring of 48-KB K-tiles: two wavefronts per SIMD (a lone
//              wavefront issues an fp32 MFMA every ~68 cycles, two reach 64) AND 16 fragment reads per 64 MFMAs
//              instead of 24 -- the K loop of the first two forms runs at ~4400 cycles per 64 MFMAs per SIMD with
//              or without its barrier (measured with the barrier compiled out): it pays for the fragment reads
//              and the issue cadence, not for the rendezvous.
// Three stages always: the DMA pieces of K-tile j + 2 are issued in k-group 0 of K-tile j (stage (j+2)%3 was last
// read in K-tile j - 1), i.e. two K-tile times in flight.  (A two-stage ring for the 256x128 form -- pieces issued
// behind the barrier, one K-tile time in flight -- was measured: every K-tile then waits ~350 cycles for its data.)
// LDS: 3 x 48 KB of stages leave no room for anything else, so (a) the epilogue's transpose scratch of a wavefront
// lives in the stage that the tile's last K-tile has just left, inside the slice that the SAME wavefront's next
// DMA pieces will overwrite (nobody else touches it in between), and (b) the bias / scale / shift values are not
// kept for all N columns: every wavefront DMAs the 3 x 64 values of ITS columns at the start of each tile.
// RES (round 4): the residual of a ResNet Bottleneck's third convolution (wespeaker/models/resnet.py:72-107:
// out = relu(bn3(conv3(.)) + shortcut(x))) is added in the epilogue, (acc + bias) + residual -> ReLU like the tile
// kernels (same bits).  Its rows -- the same 128-B row segments the block's stores write -- are requested into
// registers one K-tile ahead of the tile's last K-tile (16 or 8 raw buffer loads per wavefront in the free MFMA gaps
// of k-groups 1 and 2) and consumed in epilogue steps 5..8.
// CONV (round 5): the A operand of a 3x3 / stride 1 / pad 1 convolution over channels-last images (the BasicBlock
// and Bottleneck 3x3 layers of wespeaker/models/resnet.py:35-107 with 128 / 256 planes).  K-tile kt is 32 channels of
// ONE filter tap (k = tap * Cin + ci, the implicit-GEMM kernels' order: same bits), so a piece's source is the row's
// centre pixel + a wave-uniform tap offset; a tap that falls outside the image reads zeros that the caller keeps
// behind the tensor (ConvGemmParams::a_zero_off: Cin zero floats; nine validity bits per piece row, computed once per
// tile; the lane offsets of the current tap are selected once per TAP) -- every piece is still exactly one DMA
// operation, which the vmcnt arithmetic of the barriers relies on.
// WNP (round 5): wavefront columns, 4 / TN (a 128-column tile) unless given: <4, 1, .., 2> is eight wavefronts of
// 64 x 32 over a 256 x 64 tile for the 64-channel layers (ResNet stage 2 / the 64-plane bottleneck layers).
// MASKT (round 6): the RES / CONV forms of a ragged batch (ConvGemmParams::row_len) are instantiations of their own
// (those forms have no registers to spare: the uniform ones stay what they were); the plain forms mask at run time.
template <int WM, int TN, int COLSUM, bool RES = false, bool CONV = false, int WNP = 4 / TN, bool MASKT = false>   // COLSUM: 0 none, 1 column sums, 2 + sums of squares
__global__ __launch_bounds__(64 * WM * WNP, WM * WNP / 4)
void gemm_f32_stream_kernel(const ConvGemmParams p) {
  constexpr int WN = WNP;                    // wavefront columns
  constexpr int NW = WM * WN;
  constexpr int S_BM = 64 * WM;
  constexpr int S_BN = 32 * TN * WN;         // 128 (64: WNP = 2 with TN = 1)
  constexpr int WPIECES = S_BN / 8;          // 1-KiB pieces of W rows per stage
  constexpr int S_STAGE_BYTES = (S_BM + S_BN) * S_BK * 4;       // 32 / 48 / 40 KiB
  constexpr int S_W_BYTE0 = S_BM * S_BK * 4;                    // W rows behind the A rows of a stage
  constexpr int S_NSTAGE = 3;
  constexpr int APIECES = S_BM / 8;          // 1-KiB pieces of A rows per stage (8 rows of 128 B each)
  constexpr int GM = 8 * TN;                 // MFMAs per k-group (8 k) of a K-tile
  constexpr int NP = (APIECES + WPIECES) / NW;   // 1-KiB DMA pieces per wavefront and K-tile
  constexpr int NF = 2 + TN;                 // fragment reads per k-group
  static_assert((APIECES + WPIECES) % NW == 0 && NP <= 8, "pieces divide over the wavefronts");
  // the scratch inside the wavefront's own DMA slice of the free stage when it fits, else behind the stages
  constexpr bool SCR_ALIAS = NP * 1024 >= S_SCR_BYTES;
  constexpr int SCR_OFF = S_NSTAGE * S_STAGE_BYTES;
  constexpr int VEC_OFF = SCR_OFF + (SCR_ALIAS ? 0 : NW * S_SCR_BYTES);
  constexpr int VEC_WAVE_BYTES = 2 * 3 * 64 * 4;               // [tile parity][bias | scale | shift][64 columns]
  // VMEM operations that may stay in flight at a K-tile's barrier: the pieces of K-tile j + 2; in the FIRST K-tile of
  // a tile also the three channel-vector pieces and the stores of the previous tile's epilogue, which were all
  // issued behind the pieces this barrier needs (vmcnt retires in order): 4 row stores per 32x32 block (more with
  // D2: then the wait is a little early, never late) and the column sums
  constexpr int WAITN = NP;
  constexpr int WAITN_FIRST = NP + 3 + 2 * TN * 4 + COLSUM * 2 * TN;
  // DEFER (16-MFMA k-groups only: enough free gaps): the last block of a tile has no MFMAs of its own tile left to
  // hide behind, and the two wavefronts of a SIMD reach that point together -- ~1200 (plain) to ~3500 (column sums)
  // idle cycles per tile.  Its accumulators go through the scratch into registers at once (the scratch is about to
  // be overwritten by this wavefront's DMA pieces); maths, stores and the column-sum fold run behind the MFMAs of
  // the NEXT tile's first K-tile.  Those stores are issued behind that K-tile's DMA pieces, so the K-tile after it
  // lets them stay in flight too (WAITN_SECOND).
  constexpr bool DEFER = GM == 16;
  constexpr int WAITN_SECOND = DEFER ? NP + 4 + COLSUM * 2 * TN : NP;
  static_assert(WAITN_FIRST < 64, "vmcnt is six bits");
  static_assert(!RES || (2 * TN * 4 <= GM && NP + 2 * TN * 4 < 64), "the residual loads fit the free gaps of g1 / g2");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int li = lane & 31, lh = lane >> 5;
  const int r8 = lane >> 3, c8 = lane & 7;
  const int nk = p.K / S_BK;
  const int tiles_n = p.N / S_BN;
  const int n_tiles = p.n_big;
  const int HW = p.Hout * p.Wout;

  // ---- this workgroup's tile sequence: seq -> round * grid + (XCD-contiguous index inside the round)
  const int nwg = gridDim.x;
  int remap;
  {
    const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    remap = xcd * q + (xcd < r ? xcd : r) + local;
  }
  if (remap >= n_tiles) return;
  const int my_nt = (n_tiles - remap + nwg - 1) / nwg;
  // MASK (round 6): a ragged batch (ConvGemmParams::row_len) -- output pixels at or beyond their utterance's own width
  // are stored as zeros, like the tile kernels do.  The batch's widths (<= 1024 of them) are copied into the last
  // 4 KB of LDS once, before the operand stream starts; at the start of a tile a lane works out which of its eight
  // rows (r8 + 8 i + 32 im of the wavefront's 64-row half) are real -- two integer divisions for the first row, the
  // others by stepping (image size >= 8 pixels, width >= 8) -- and keeps one bit per row; the epilogue selects.
  const bool masked = MASKT || (!RES && !CONV && p.row_len != nullptr);
  int* const lens_s = reinterpret_cast<int*>(ldsb + VEC_OFF + NW * VEC_WAVE_BYTES);
  const int nimg = masked ? p.M / HW : 0;
  if (masked) {
    for (int i = tid; i < nimg; i += 64 * NW) lens_s[i] = p.row_len[i];
    __syncthreads();                         // (waits for the loads too: nothing of the stream is in flight yet)
  }
  auto row_mask = [&](int mh) {              // bit (4 im + i): row mh + 32 im + 8 i + r8 is inside its utterance
    const int m = mh + r8;
    int img = m / HW, rem = m - img * HW;
    int ox = rem - (rem / p.Wout) * p.Wout;
    unsigned mk = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mk |= (ox < lens_s[img < nimg ? img : nimg - 1] ? 1u : 0u) << j;
      rem += 8;
      ox += 8;
      if (rem >= HW) { rem -= HW; ++img; ox = rem; }      // (rem < 8 <= Wout behind the wrap)
      else if (ox >= p.Wout) ox -= p.Wout;
    }
    return mk;
  };
  auto tile_of = [&](int seq, int& m0, int& n0) {
    const int work = seq * nwg + remap;
    const int tm = work / tiles_n;
    m0 = p.m_begin + tm * S_BM;
    n0 = (work - tm * tiles_n) * S_BN;
  };

  // ---- operand stream (prefetch side).  Piece q of a stage = rows [8 q, +8) of A (q < APIECES) or rows
  // [8 (q - APIECES), +8) of W; wavefront w copies pieces [w NP, (w + 1) NP).
  // lane l of a piece: row rr = l >> 3, physical 16-B chunk pc = l & 7 <- logical chunk pc ^ key(row).
  unsigned voff[NP];
  unsigned vmask[CONV ? NP : 1];             // CONV: bit (3 ty + tx) = tap (ty, tx) of the piece row lies inside the image
  auto piece_is_w = [&](int i) { return wave * NP + i >= APIECES; };          // wave-uniform
  auto set_tile_offsets = [&](int seq) {
    int m0, n0;
    tile_of(seq, m0, n0);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool isw = piece_is_w(i);
      const int q = wave * NP + i - (isw ? APIECES : 0);
      const int row = q * 8 + r8;
      const int c = c8 ^ ((row >> 1) & 7);
      const int ld = isw ? p.ldw : p.lda, off = isw ? 0 : p.a_off, row0 = isw ? n0 : m0;
      unsigned long long lin = (unsigned long long)(row0 + row);      // plain forms: the operand's own row
      if (CONV) {
        // the row's centre pixel (stride 1, pad 1: the input pixel under tap (1, 1)) and which taps lie inside the
        // image; (computed for the W pieces too and not used there: a select instead of a branch)
        const int m = m0 + row;
        const int img = m / HW, rem = m - img * HW;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        const unsigned pix = (unsigned)((img * p.Hin + oy) * p.Win + ox);
        lin = isw ? lin : (unsigned long long)pix;
        const unsigned cm = (ox > 0 ? 1u : 0u) | 2u | (ox + 1 < p.Win ? 4u : 0u);
        const unsigned rm = (oy > 0 ? 1u : 0u) | 8u | (oy + 1 < p.Hin ? 64u : 0u);
        vmask[i] = cm * rm;
      }
      voff[i] = (unsigned)((lin * ld + off) * 4ull + c * 16);   // (< 2^32: the guard)
    }
  };
  int pf_seq = 0, pf_kt = 0, pf_stage = 0;
  // CONV: filter tap / 32-channel group of K-tile pf_kt, and the byte offset of that tap's pixel from the centre pixel
  const int cpt = CONV ? p.Cin / S_BK : 1;
  int pf_tap = 0, pf_kc = 0;
  long long pf_delta = CONV ? -(long long)(p.Win + 1) * p.lda * 4 : 0;
  // ... and where the 16 zero bytes behind the tensor lie from that tap's base: the offset of a lane whose tap falls
  // outside the image (one lane select per piece; the base stays scalar: the saddr form like the plain pieces)
  const unsigned zero_lo = CONV ? (unsigned)p.a_zero_off : 0u;        // (a_zero_off - delta < 2^32: the guard)
  // The lane offsets of the CURRENT tap: the centre-pixel offset, or -- the tap outside the image -- the zeros minus
  // the tap's pixel offset, so that base(tap) + 128 kc + loff lands inside the Cin zero floats for every channel
  // group kc.  Recomputed when the tap changes (every Cin / 32 K-tiles), not per piece and K-tile: the DMA issue of a
  // CONV piece is then the plain form's (scalar base + lane offset).
  unsigned loff[CONV ? NP : 1];
  auto set_tap = [&](int tap) {
    const int ty = tap / 3, tx = tap - 3 * ty;
    const unsigned tapdelta = (unsigned)(((ty - 1) * p.Win + (tx - 1)) * p.lda * 4);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool ok = piece_is_w(i) || ((vmask[CONV ? i : 0] >> tap) & 1u);
      loff[CONV ? i : 0] = ok ? voff[i] : zero_lo - tapdelta;
    }
  };
  set_tile_offsets(0);
  if (CONV) set_tap(0);
  auto dma_piece = [&](int i) {
    char* dst = ldsb + pf_stage * S_STAGE_BYTES + (wave * NP + i) * 1024;
    if (CONV) {
      // one instruction stream for the A and the W pieces (the split is a wave-uniform RUN-TIME property: a branch
      // here would sit between the MFMAs): scalar select of the base
      const char* base = piece_is_w(i) ? reinterpret_cast<const char*>(p.W) + (size_t)(unsigned)(pf_kt * (S_BK * 4))
                                       : reinterpret_cast<const char*>(p.A) + pf_delta;
      s_dma_16B(base + loff[CONV ? i : 0], dst);
      return;
    }
    // wave-uniform 64-bit base (operand + K offset) + 32-bit lane offset: the saddr form of the instruction
    const char* gbase = piece_is_w(i) ? reinterpret_cast<const char*>(p.W) : reinterpret_cast<const char*>(p.A);
    const char* kb = gbase + (size_t)(unsigned)(pf_kt * (S_BK * 4));
    s_dma_16B(kb + voff[i], dst);
  };
  // the channel vectors of tile `seq`: three 256-B pieces (64 lanes x 4 B) of this wavefront's columns into its
  // private slots; always three, so that every wavefront counts the same VMEM operations (a missing vector is read
  // from the 16 zero bytes / replaced by constants in the epilogue)
  char* const vec_w = ldsb + VEC_OFF + wave * VEC_WAVE_BYTES;
  auto vec_pieces = [&](int seq) {
    int m0, n0;
    tile_of(seq, m0, n0);
    int col = n0 + wn * 32 * TN + lane;
    col = col < p.N ? col : p.N - 1;
#if defined(__HIP_DEVICE_COMPILE__)
    char* dst = vec_w + (seq & 1) * (3 * 64 * 4);
    const float* zero = p.zeros;
    __builtin_amdgcn_global_load_lds(p.bias ? p.bias + col : zero, (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
    __builtin_amdgcn_global_load_lds(p.post_scale ? p.post_scale + col : zero,
                                     (__attribute__((address_space(3))) void*)(dst + 256), 4, 0, 0);
    __builtin_amdgcn_global_load_lds(p.post_scale ? p.post_shift + col : zero,
                                     (__attribute__((address_space(3))) void*)(dst + 512), 4, 0, 0);
#endif
  };
  auto pf_advance = [&]() {
    pf_stage = pf_stage == S_NSTAGE - 1 ? 0 : pf_stage + 1;
    if (++pf_kt == nk) {
      if (pf_seq + 1 < my_nt) {
        pf_kt = 0;
        ++pf_seq;
        set_tile_offsets(pf_seq);
      } else {
        pf_kt = nk - 1;       // past the end: repeat the last K-tile (keeps the vmcnt arithmetic uniform)
        return;
      }
    }
    if (CONV) {
      if (pf_kt == 0) { pf_tap = 0; pf_kc = 0; }
      else if (++pf_kc == cpt) { pf_kc = 0; ++pf_tap; }
      const int ty = pf_tap / 3, tx = pf_tap - 3 * ty;
      pf_delta = ((long long)(ty - 1) * p.Win + (tx - 1)) * p.lda * 4 + pf_kc * (S_BK * 4);
      if (pf_kc == 0) set_tap(pf_tap);
    }
  };

  // ---- fragment addresses (bytes inside the current stage; advanced at every barrier)
  int aaddr[4], waddr[4];
  {
    const int key = (li >> 1) & 7;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int pc = ((2 * g) | lh) ^ key;
      aaddr[g] = (wm * 64 + li) * 128 + pc * 16;
      waddr[g] = S_W_BYTE0 + (wn * 32 * TN + li) * 128 + pc * 16;
    }
  }
  int cur_stage = 0;
  auto advance_frag_addrs = [&]() {
    const int d = cur_stage == S_NSTAGE - 1 ? -(S_NSTAGE - 1) * S_STAGE_BYTES : S_STAGE_BYTES;
    cur_stage = cur_stage == S_NSTAGE - 1 ? 0 : cur_stage + 1;
#pragma unroll
    for (int g = 0; g < 4; ++g) { aaddr[g] += d; waddr[g] += d; }
  };

  f32x4 fa[2][2], fb[2][TN];                 // [slot][block]: one 16-B read feeds 4 MFMAs (k = 8g + 4lh + s)
  auto frag_read = [&](int slot, int g, int idx) {        // idx 0,1: A blocks; 2..: W blocks
    if (idx < 2) fa[slot][idx] = *reinterpret_cast<const f32x4*>(ldsb + aaddr[g] + idx * 4096);
    else fb[slot][idx - 2] = *reinterpret_cast<const f32x4*>(ldsb + waddr[g] + (idx - 2) * 4096);
  };

  f32x16 acc[2][TN];
  int cur_seq = 0;
  bool trace_now = false;                    // (WS_TRACE builds: fine stamps inside the K-tile that sets it)
  (void)trace_now;

  // ---- epilogue state
  // (SCR_ALIAS: re-pointed by the last K-tile of every tile into the stage it has just left)
  float* scr = reinterpret_cast<float*>(ldsb + (SCR_ALIAS ? wave * NP * 1024 : SCR_OFF + wave * S_SCR_BYTES));
  const bool has_post = p.post_scale != nullptr;
  // raw buffer descriptors of D / D2: one store instruction per row = lane offset (VGPR) + row offset (SGPR),
  // no 64-bit address arithmetic on the vector ALU
  // (the descriptors END at row M: the row units' last, partial strip stores rows that do not exist -- dropped)
  const __amdgpu_buffer_rsrc_t d_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.D, 0, (unsigned)((unsigned long long)p.M * p.ldd * 4ull), 0x00020000);
  const __amdgpu_buffer_rsrc_t d2_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.D2 ? p.D2 : p.D, 0,
                                        (unsigned)((unsigned long long)p.M * (p.D2 ? p.ldd2 : p.ldd) * 4ull), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      RES ? const_cast<float*>(p.residual) : p.D, 0, 0xffffffff, 0x00020000);
  // ReLU as ONE integer max per value: the int image of a float is >= 0 exactly for +0, positive values, +inf and
  // positive NaNs, so max(bits, 0) maps every negative value (and -0) to +0 and keeps NaN a NaN like relu_f
  // does; without an activation the bound is INT_MIN (identity)
  const int relu_bits = p.act == ACT_RELU ? 0 : (int)0x80000000;
  // the tile whose blocks are being finished
  struct TileOut {
    unsigned dvoff;        // byte offset of D[m0 + wm*64 + r8][d_off + n0 + wn*32*TN + 4 c8]
    unsigned d2voff;       // the same for D2 (columns n - d2_col0)
    unsigned rvoff;        // the same for the residual
    const float* vec;      // this wavefront's [bias | scale | shift][64] slots of the tile
    int ncol;              // n0 + wn*32*TN + 4 c8: first of this lane's 4 columns in block 0
    int nblk;              // n0 + wn*32*TN (wave-uniform)
    int rb;                // rows of the wavefront's 64-row half that belong to its first image
    int t64;               // (m0 + wm*64) / 64
    unsigned okmask;       // MASK: bit (4 im + i) = row 32 im + 8 i + r8 of the half lies inside its utterance
  };
  TileOut cur = {}, prev = {};
  bool pending = false;                      // DEFER: `prev`'s last block is waiting in ev[]
  auto set_tile_out = [&](int seq, TileOut& t) {
    int m0, n0;
    tile_of(seq, m0, n0);
    const int mh = m0 + wm * 64;
    t.vec = reinterpret_cast<const float*>(vec_w + (seq & 1) * (3 * 64 * 4));
    t.nblk = n0 + wn * 32 * TN;
    t.ncol = t.nblk + c8 * 4;
    t.dvoff = (unsigned)(((unsigned long long)(mh + r8) * p.ldd + p.d_off + t.ncol) * 4ull);
    t.d2voff = p.D2 ? (unsigned)(((unsigned long long)(mh + r8) * p.ldd2 + p.d2_off + t.ncol - p.d2_col0) * 4ull) : 0u;
    t.rvoff = RES ? (unsigned)(((unsigned long long)(mh + r8) * p.ldr + p.r_off + t.ncol) * 4ull) : 0u;
    t.rb = COLSUM ? (mh / HW + 1) * HW - mh : 64;
    t.t64 = mh >> 6;
    t.okmask = masked ? row_mask(mh) : 0xffu;
  };
  f32x4 cs[TN][2];                           // column sums of the stored values: [block column][image part]
  f32x4 cq[COLSUM == 2 ? TN : 1][2];         // ... and of their squares (the context std of the pooling layer)
#pragma unroll
  for (int in = 0; in < TN; ++in) cs[in][0] = cs[in][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int in = 0; in < (COLSUM == 2 ? TN : 1); ++in) cq[in][0] = cq[in][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 ev[4], vb, vs, vt;                   // rows r8 + 8 i of the block being finished; bias / scale / shift
  bool unit_partial = false;                 // (wave-uniform) a row unit whose strip runs past row M: its okmask drops those rows
  f32x4 rres[RES ? 2 * TN : 1][4];           // RES: the residual rows of all blocks of the tile ([im * TN + in][i])
  constexpr int NRES = RES ? 2 * TN * 4 : 0;
  auto res_load = [&](int j, const TileOut& t) {           // j = (im * TN + in) * 4 + i
    if (!RES || j >= NRES) return;
    const int blk = j >> 2, i = j & 3, im = blk / TN, in = blk - im * TN;
    const int srow = im * 32 + 8 * i;
    typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
    const u32x4r raw = __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, t.rvoff, (srow * p.ldr + in * 32) * 4, 0);
    rres[RES ? blk : 0][i] = __builtin_bit_cast(f32x4, raw);
  };

  // One block's epilogue in 15 steps (each small enough for the shadow of one MFMA).  im / in: the block,
  // t: its tile; steps 0-1 take the accumulator quads `q` (C^T block: lane (li, lh) holds channels
  // 8 g + 4 lh .. +3 of pixel li in registers 4 g .. 4 g + 3).
  // TAIL: the step runs with no MFMA in front of it (the two MFMA-free tails below): its stores then carry the row
  // offset in the VGPR offset instead of the scalar one -- see the note at the tail behind the tile loop
  auto epi_step = [&](int step, int im, int in, const TileOut& t, const f32x16& q, bool tail = false) {
    if (step == 0 || step == 1) {
#pragma unroll
      for (int g = 2 * step; g < 2 * step + 2; ++g)
        *reinterpret_cast<f32x4*>(&scr[li * S_SCR_STRIDE + 8 * g + 4 * lh]) =
            (f32x4){q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3]};
    } else if (step == 2 || step == 3) {
#pragma unroll
      for (int i = 2 * (step - 2); i < 2 * (step - 2) + 2; ++i)
        ev[i] = *reinterpret_cast<const f32x4*>(&scr[(r8 + 8 * i) * S_SCR_STRIDE + c8 * 4]);
    } else if (step == 4) {
      const int n = in * 32 + c8 * 4;
      vb = *reinterpret_cast<const f32x4*>(&t.vec[n]);
      vs = has_post ? *reinterpret_cast<const f32x4*>(&t.vec[64 + n]) : (f32x4){1.f, 1.f, 1.f, 1.f};
      vt = has_post ? *reinterpret_cast<const f32x4*>(&t.vec[128 + n]) : (f32x4){0.f, 0.f, 0.f, 0.f};
    } else if (step >= 5 && step <= 8) {
      const int i = step - 5;
      f32x4 v = ev[i] + vb;
      if (RES) v += rres[RES ? im * TN + in : 0][i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // (a scalar copy first: __builtin_bit_cast on a vector ELEMENT reads element 0 whatever the index)
        const float f = v[e];
        v[e] = __int_as_float(__builtin_elementwise_max(__float_as_int(f), relu_bits));
      }
      ev[i] = v * vs + vt;
      if ((masked || unit_partial) && !((t.okmask >> (4 * im + i)) & 1u)) ev[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    } else if (step >= 9 && step <= 12) {
      const int i = step - 9;
      const int srow = im * 32 + 8 * i;
      const int so = (srow * p.ldd + in * 32) * 4, so2 = (srow * p.ldd2 + in * 32) * 4;
      // (the four-wavefront form too: hipcc sinks its rows' maths behind the previous row's stores inside one MFMA gap)
      // (... and the masked RES / CONV twins: their select leaves a VALU write of ev[] next to a store; build.check_isa found it)
      const bool voff_form = tail || NW == 4 || MASKT;
      if (voff_form) __builtin_amdgcn_raw_buffer_store_b128(ev[i], d_rsrc, t.dvoff + so, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(ev[i], d_rsrc, t.dvoff, so, 0);
      // (d2_col0 is a multiple of 32: a 32-column block goes to D2 as a whole -- a wave-uniform branch)
      if (p.D2 && t.nblk + in * 32 >= p.d2_col0) {
        if (voff_form) __builtin_amdgcn_raw_buffer_store_b128(ev[i], d2_rsrc, t.d2voff + so2, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(ev[i], d2_rsrc, t.d2voff, so2, 0);
      }
    } else if (COLSUM && (step == 13 || step == 14)) {
#pragma unroll
      for (int i = 2 * (step - 13); i < 2 * (step - 13) + 2; ++i) {
        const float f0 = im * 32 + r8 + 8 * i < t.rb ? 1.f : 0.f, f1 = 1.f - f0;
        cs[in][0] += ev[i] * f0;
        cs[in][1] += ev[i] * f1;
        if (COLSUM == 2) {
          const f32x4 sq = ev[i] * ev[i];
          cq[in][0] += sq * f0;
          cq[in][1] += sq * f1;
        }
      }
    }
  };
  // column sums of a finished tile: fold the 8 row groups (lane bits 3..5), lanes 0..7 store, reset.
  // step 0..2: one butterfly round each; step 3: store + reset
  auto colsum_step = [&](int step, const TileOut& t) {
    if (!COLSUM) return;
    if (step < 3) {
      const int m = 8 << step;
#pragma unroll
      for (int in = 0; in < TN; ++in)
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float f = cs[in][w][e];
            cs[in][w][e] = f + __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ m) << 2, __float_as_int(f)));
            if (COLSUM == 2) {
              const float g = cq[in][w][e];
              cq[in][w][e] = g + __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ m) << 2, __float_as_int(g)));
            }
          }
    } else {
      if (r8 == 0) {
#pragma unroll
        for (int in = 0; in < TN; ++in)
#pragma unroll
          for (int w = 0; w < 2; ++w)
            *reinterpret_cast<f32x4*>(p.colsum + ((size_t)t.t64 * 2 + w) * p.N + t.ncol + in * 32) = cs[in][w];
        if (COLSUM == 2) {
#pragma unroll
          for (int in = 0; in < TN; ++in)
#pragma unroll
            for (int w = 0; w < 2; ++w)
              *reinterpret_cast<f32x4*>(p.colsumsq + ((size_t)t.t64 * 2 + w) * p.N + t.ncol + in * 32) = cq[in][w];
        }
      }
#pragma unroll
      for (int in = 0; in < TN; ++in) cs[in][0] = cs[in][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int in = 0; in < (COLSUM == 2 ? TN : 1); ++in) cq[in][0] = cq[in][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  // the epilogue of a tile's LAST block behind its scratch round trip (ev[] holds its rows): d = 0 channel vectors,
  // 1-4 row maths, 5-8 row stores, 9-10 column-sum accumulation, 11-13 column-sum butterfly, 14 column-sum store
  constexpr int DEF_STEPS = 15;
  auto deferred_step = [&](int d, const TileOut& t, bool tail = false) {
    const f32x16 none = {};
    if (d < 11) epi_step(d + 4, 1, TN - 1, t, none, tail);
    else colsum_step(d - 11, t);
  };
  // ---- prologue: two K-tiles in flight, the first one landed, its first fragments in registers
#pragma unroll
  for (int i = 0; i < NP; ++i) dma_piece(i);
  pf_advance();
#pragma unroll
  for (int i = 0; i < NP; ++i) dma_piece(i);
  pf_advance();
  s_wait_lds_vm_barrier<NP>();                // (the second K-tile stays in flight)
#pragma unroll
  for (int i = 0; i < NF; ++i) frag_read(0, 0, i);

  // 4 * TM * TN MFMAs of one k-group from fragment slot `slot`; filler(i) is issued behind MFMA i
  auto mma_group = [&](int slot, bool zero_c, auto&& filler) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int im = 0; im < 2; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in) {
          if (zero_c && s == 0) {
            const f32x16 z = {};
            acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[slot][in][s], fa[slot][im][s], z, 0, 0, 0);
          } else {
            acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[slot][in][s], fa[slot][im][s], acc[im][in], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          filler((s * 2 + im) * TN + in);
          __builtin_amdgcn_sched_barrier(0);
        }
  };

  // One K-tile, MFMAs round-robin over the accumulator blocks.  FIRST: first K-tile of a tile (fresh accumulators).
  auto ktile = [&](auto kind_tag, bool load_res) {
    constexpr int KIND = decltype(kind_tag)::value;        // 0 regular, 1 first K-tile of a tile, 2 the one after it
    constexpr bool FIRST = KIND == 1;
    // DEFER: deferred step d of the previous tile sits in the d-th gap of k-groups 0..2 that carries neither a DMA
    // piece nor a fragment read -- all of them in front of this K-tile's barrier
    auto deferred_at = [&](int g, int i) {
      if (!(FIRST && DEFER)) return;
      int d = 0;
      bool hit = false;
#pragma unroll
      for (int gg = 0; gg < 3; ++gg)
#pragma unroll
        for (int ii = 0; ii < GM; ++ii) {
          const bool busy = (gg == 0 && ii < NP) || (ii >= GM / 2 && ii < GM / 2 + NF);
          if (!busy) {
            if (gg == g && ii == i && d < DEF_STEPS) hit = true;
            if (!hit) ++d;
          }
        }
      if (hit && pending) deferred_step(d, prev);
    };
    const int fs = FIRST ? 144 : 128;
    (void)fs;
    WS_FSTAMP(trace_now, fs + 0)
    mma_group(0, FIRST, [&](int i) {                       // g0: DMA pieces of K-tile +2, fragments of g1
      if (FIRST && i == 0) vec_pieces(cur_seq);            // (first K-tile of a tile: its channel vectors)
      if (i < NP) dma_piece(i);
      if (i >= GM / 2 && i < GM / 2 + NF) frag_read(1, 1, i - GM / 2);
      deferred_at(0, i);
    });
    WS_FSTAMP(trace_now, fs + 1)
    pf_advance();
    WS_FSTAMP(trace_now, fs + 2)
    mma_group(1, false, [&](int i) {                       // g1: fragments of g2
      if (i >= GM / 2 && i < GM / 2 + NF) frag_read(0, 2, i - GM / 2);
      if (RES && KIND == 0 && load_res && i < GM / 2) res_load(i, cur);
      deferred_at(1, i);
    });
    WS_FSTAMP(trace_now, fs + 3)
    mma_group(0, false, [&](int i) {                       // g2: fragments of g3
      if (i >= GM / 2 && i < GM / 2 + NF) frag_read(1, 3, i - GM / 2);
      if (RES && KIND == 0 && load_res && i < GM / 2) res_load(GM / 2 + i, cur);
      deferred_at(2, i);
    });
    WS_FSTAMP(trace_now, fs + 4)
    if (FIRST && cur_seq > 0) s_wait_lds_vm_barrier<WAITN_FIRST>();
    else if (FIRST) s_wait_lds_vm_barrier<NP + 3>();       // (the very first tile: no epilogue stores in flight)
    else if (KIND == 2 && cur_seq > 0) s_wait_lds_vm_barrier<WAITN_SECOND>();
    else if (RES && KIND == 0 && load_res) s_wait_lds_vm_barrier<WAITN + NRES>();   // (the residual rows stay in flight)
    else s_wait_lds_vm_barrier<WAITN>();
    WS_FSTAMP(trace_now, fs + 5)
#ifdef WS_TRACE
    if (trace_now && blockIdx.x == 9 && (threadIdx.x == 0 || threadIdx.x == 256))
      g_stream_trace[(FIRST ? 178 : 177) + threadIdx.x] = g_stream_trace[176 + threadIdx.x];
#endif
    advance_frag_addrs();
    mma_group(1, false, [&](int i) {                       // g3 (fragments in registers): g0 of the next K-tile
      if (i < NF) frag_read(0, 0, i);
    });
    WS_FSTAMP(trace_now, fs + 6)
  };

  // The last K-tile of a tile, block-major.
  auto ktile_last = [&]() {
    f32x4 la[2][4], lb[TN][4];                           // [block][k-group]
#pragma unroll
    for (int im = 0; im < 2; ++im) la[im][0] = fa[0][im];
#pragma unroll
    for (int in = 0; in < TN; ++in) lb[in][0] = fb[0][in];
    auto lread = [&](int idx, int g) {                   // idx 0,1: A blocks; 2..: W blocks
      if (idx < 2) la[idx][g] = *reinterpret_cast<const f32x4*>(ldsb + aaddr[g] + idx * 4096);
      else lb[idx - 2][g] = *reinterpret_cast<const f32x4*>(ldsb + waddr[g] + (idx - 2) * 4096);
    };
    // the reads still missing, in the order block 0 needs them: (A0, W0) of g1, g2, g3, then the rest
    constexpr int NREST = 3 * NF - 6;                    // A1 (and W1) of g1..g3
#pragma unroll
    for (int b = 0; b < 2 * TN; ++b) {
      WS_FSTAMP(trace_now, 160 + 2 * b)
      const int im = b / TN, in = b - im * TN;
      const int pim = (b - 1) / TN, pin = (b - 1) - pim * TN;      // the block finished behind this one
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(lb[in][g][s], la[im][g][s], acc[im][in], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          const int i = g * 4 + s;
          if (b == 0) {
            // (hipcc waits with lgkmcnt(0): a read is placed >= 4 MFMAs in front of the first MFMA that waits)
            auto rest = [&](int j) {                     // j -> A1 g1..g3, then W1 g1..g3
              if (j >= NREST) return;
              if (j < 3) lread(1, j + 1);
              else lread(3, j - 3 + 1);
            };
            if (i == 0) { lread(0, 1); lread(2, 1); lread(0, 2); lread(2, 2); }
            else if (i == 4) { lread(0, 3); lread(2, 3); rest(0); rest(1); }
            else if (i == 5) { rest(2); rest(3); }
            else if (i == 6) { rest(4); rest(5); }
            else if (i >= 7 && i < 7 + NP) dma_piece(i - 7);
          } else {
            if (i < 15) epi_step(i, pim, pin, cur, acc[pim][pin]);
            if (b == 2 * TN - 1 && i >= 16 - NF) frag_read(0, 0, i - (16 - NF));     // next K-tile's first fragments
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      if (b == 0) {
        WS_FSTAMP(trace_now, 160 + 1)
        pf_advance();
        s_wait_lds_vm_barrier<WAITN>();
        // the stage this K-tile has just left is free until this wavefront's own pieces of K-tile +3 land in it
        if (SCR_ALIAS) scr = reinterpret_cast<float*>(ldsb + cur_stage * S_STAGE_BYTES + wave * NP * 1024);
        advance_frag_addrs();
      }
    }
    WS_FSTAMP(trace_now, 160 + 8)
    // the last block: through the scratch into ev[] now; the rest behind the next tile's MFMAs (DEFER) or here
    epi_step(0, 1, TN - 1, cur, acc[1][TN - 1]);
    epi_step(1, 1, TN - 1, cur, acc[1][TN - 1]);
    epi_step(2, 1, TN - 1, cur, acc[1][TN - 1]);
    epi_step(3, 1, TN - 1, cur, acc[1][TN - 1]);
    if (DEFER) {
      prev = cur;
      pending = true;
    } else {
#pragma unroll
      for (int d = 0; d < DEF_STEPS; ++d) deferred_step(d, cur, true);   // (TAIL: see behind the tile loop)
    }
    WS_FSTAMP(trace_now, 160 + 9)
  };

  int stamp = 0;
  (void)stamp;
  for (int seq = 0; seq < my_nt; ++seq) {
    cur_seq = seq;
    set_tile_out(seq, cur);
    WS_SSTAMP(stamp++)
    trace_now = seq == 1;
    ktile(std::integral_constant<int, 1>{}, false);
    pending = false;
    WS_SSTAMP(stamp++)
    trace_now = false;
    ktile(std::integral_constant<int, 2>{}, false);
    for (int kt = 2; kt + 1 < nk; ++kt) {
      WS_SSTAMP(stamp++)
      trace_now = seq == 1 && kt == 5;
      ktile(std::integral_constant<int, 0>{}, kt + 2 == nk);       // (RES: the last regular K-tile requests the residual)
    }
    WS_SSTAMP(stamp++)
    trace_now = seq == 1;
    ktile_last();
    trace_now = false;
  }
  WS_SSTAMP(stamp++)
  // No MFMAs between these steps, so hipcc packs the rows' maths and stores tightly, and the register file of the
  // CONV form left `buffer_store_dwordx4 v[12:15], v185, s[28:31], s0 offen` DIRECTLY in front of
  // `v_pk_add_f32 v[12:13], ..`: MI355X then stores the NEW values in some lanes (a few wrong elements in rows 8..15
  // of the last block of every workgroup's last tile, different from run to run; found with tools/conv_stream_probe).
  // LLVM's hazard recogniser pads a >64-bit store in front of a VALU write of its data registers only when the store
  // has NO scalar-offset register; with one it assumes there is no hazard.  So the MFMA-free tails store with the row
  // offset in the VGPR offset (TAIL) -- the form the recogniser does pad -- and build.check_isa refuses the unpadded
  // pair anywhere in the library.
  if (DEFER && pending) {
#pragma unroll
    for (int d = 0; d < DEF_STEPS; ++d) deferred_step(d, prev, true);
  }
  // ---- Row units (round 6): the rows behind the whole rounds of tiles, INSIDE this launch.  A separate launch for
  // the last 1 536 of ECAPA's 50 688 rows cost 16 - 25 us per N = K = 512 layer and 105 us on the 1536-wide one (a
  // launch ramp, a cold first K-tile and an epilogue for 7 / 46 us of matrix work: DESIGN.md 4.2.4 / 4.2.6); here the
  // workgroups that have finished their tiles take 64 x 64 units of those rows -- unit u = (64-row strip, 64-column
  // block), 192 of them for N = 512 -- through the same LDS ring: wavefronts 0-3 own one 32 x 32 block each (the
  // same k order per accumulator as the tiles: the same bits), all eight copy the 16-KB K-tiles (64 rows of A, 64
  // of W) two ahead, and the block's epilogue is the tiles' (epi_step).  Column sums: the two wavefronts of a
  // 64-row strip add their halves through LDS in a fixed order.
  // (both eight-wavefront plain forms: the 256x128 tile and the 128x128 tile of <2, 1, .., 4>)
  constexpr bool UNITS = NW == 8 && ((WM == 4 && TN == 2 && WNP == 2) || (WM == 2 && TN == 1 && WNP == 4)) && !RES && !CONV;
  if constexpr (UNITS) {
    if (p.n_units > 0) {
      // the operand stream has ended: its pieces landed, everybody is done with the stages and the scratch
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      constexpr int U_STAGE = 16 * 1024;                      // 64 rows of A + 64 rows of W, 128 B each
      constexpr int U_SCR = 3 * U_STAGE;                      // the transpose scratch of the four block owners
      constexpr int U_XCH = U_SCR + 4 * S_SCR_BYTES;          // column-sum exchange: [wn][cs | cq][part][8 lanes] x 16 B
      static_assert(U_XCH + 2 * 2 * 2 * 8 * 16 <= VEC_OFF, "the unit layout ends in front of the channel vectors");
      const int utn = p.N >> 6;
      const bool active = wave < 4;                           // (wave-uniform) block owners
      const int wm2 = (wave >> 1) & 1, wn2 = wave & 1;
      const bool isw = wave >= 4;                             // wavefronts 0-3 copy the A rows, 4-7 the W rows
      const char* const ubase = isw ? reinterpret_cast<const char*>(p.W) : reinterpret_cast<const char*>(p.A);
      int ua[4], uw[4];
      {
        const int key = (li >> 1) & 7;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int pc = ((2 * g) | lh) ^ key;
          ua[g] = (wm2 * 32 + li) * 128 + pc * 16;
          uw[g] = 8192 + (wn2 * 32 + li) * 128 + pc * 16;
        }
      }
      for (int u = remap; u < p.n_units; u += nwg) {
        const int us = u / utn;
        const int um0 = p.tail_begin + 64 * us, un0 = (u - us * utn) * 64;
        unsigned uvoff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = ((wave & 3) * 2 + i) * 8 + r8;
          const int c = c8 ^ ((row >> 1) & 7);
          // (a partial last strip: the rows past M read row M - 1 again, their results are masked and their stores dropped)
          const int arow = um0 + row < p.M ? um0 + row : p.M - 1;
          const unsigned long long lin = (unsigned long long)(isw ? un0 + row : arow);
          uvoff[i] = (unsigned)((lin * (isw ? p.ldw : p.lda) + (isw ? 0 : p.a_off)) * 4ull + c * 16);
        }
        auto udma = [&](int kt, int stage) {
          const char* kb = ubase + (size_t)(unsigned)(kt * (S_BK * 4));
#pragma unroll
          for (int i = 0; i < 2; ++i) s_dma_16B(kb + uvoff[i], ldsb + stage * U_STAGE + (wave * 2 + i) * 1024);
        };
        {                                                     // the channel vectors of this wavefront's 32 columns
          int col = un0 + wn2 * 32 + (lane & 31);
          col = col < p.N ? col : p.N - 1;
#if defined(__HIP_DEVICE_COMPILE__)
          const float* zero = p.zeros;
          __builtin_amdgcn_global_load_lds(p.bias ? p.bias + col : zero, (__attribute__((address_space(3))) void*)vec_w, 4, 0, 0);
          __builtin_amdgcn_global_load_lds(p.post_scale ? p.post_scale + col : zero,
                                           (__attribute__((address_space(3))) void*)(vec_w + 256), 4, 0, 0);
          __builtin_amdgcn_global_load_lds(p.post_scale ? p.post_shift + col : zero,
                                           (__attribute__((address_space(3))) void*)(vec_w + 512), 4, 0, 0);
#endif
        }
        udma(0, 0);
        udma(1, 1);                                           // (nk >= 4)
        s_wait_lds_vm_barrier<2>();                           // K-tile 0 (and everything older) landed
        f32x16 uacc = {};
        int st = 0;
        for (int kt = 0; kt < nk; ++kt) {
          const int pst = st == 0 ? 2 : st - 1;
          udma(kt + 2 < nk ? kt + 2 : nk - 1, pst);           // (past the end: the last K-tile again, uniform counts)
          if (active) {
            const char* sb = ldsb + st * U_STAGE;
            f32x4 xa[4], xw[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {                     // all eight reads first (LDS answers in order: the
              xa[g] = *reinterpret_cast<const f32x4*>(sb + ua[g]);   // MFMAs of group g wait for 2 g + 2 of them)
              xw[g] = *reinterpret_cast<const f32x4*>(sb + uw[g]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                uacc = __builtin_amdgcn_mfma_f32_32x32x2f32(xw[g][e], xa[g][e], uacc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          s_wait_lds_vm_barrier<2>();                         // K-tile kt + 1 landed, everybody left stage st
          st = st == 2 ? 0 : st + 1;
        }
        TileOut ut = {};
        ut.vec = reinterpret_cast<const float*>(vec_w);
        ut.nblk = un0 + wn2 * 32;
        ut.ncol = ut.nblk + c8 * 4;
        ut.dvoff = (unsigned)(((unsigned long long)(um0 + r8) * p.ldd + p.d_off + ut.ncol) * 4ull);
        ut.d2voff = p.D2 ? (unsigned)(((unsigned long long)(um0 + r8) * p.ldd2 + p.d2_off + ut.ncol - p.d2_col0) * 4ull) : 0u;
        ut.rb = COLSUM ? (um0 / HW + 1) * HW - um0 : 64;
        ut.t64 = um0 >> 6;
        ut.okmask = masked ? row_mask(um0) : 0xffu;
        unit_partial = um0 + 64 > p.M;
        if (unit_partial) {
          unsigned inr = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) inr |= (um0 + 8 * j + r8 < p.M ? 1u : 0u) << j;
          ut.okmask &= inr;
        }
        if (active) {
          scr = reinterpret_cast<float*>(ldsb + U_SCR + wave * S_SCR_BYTES);
#pragma unroll
          for (int step = 0; step < 15; ++step) epi_step(step, wm2, 0, ut, uacc, true);
          colsum_step(0, ut);
          colsum_step(1, ut);
          colsum_step(2, ut);
        }
        if (COLSUM) {
          f32x4* const xch = reinterpret_cast<f32x4*>(ldsb + U_XCH) + wn2 * 32;      // [cs | cq][part][c8]
          if (active && wm2 == 1 && r8 == 0) {
            xch[c8] = cs[0][0];
            xch[8 + c8] = cs[0][1];
            if (COLSUM == 2) {
              xch[16 + c8] = cq[0][0];
              xch[24 + c8] = cq[0][1];
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (active && wm2 == 0 && r8 == 0) {
#pragma unroll
            for (int w = 0; w < 2; ++w) {
              *reinterpret_cast<f32x4*>(p.colsum + ((size_t)ut.t64 * 2 + w) * p.N + ut.ncol) = cs[0][w] + xch[8 * w + c8];
              if (COLSUM == 2)
                *reinterpret_cast<f32x4*>(p.colsumsq + ((size_t)ut.t64 * 2 + w) * p.N + ut.ncol) =
                    cq[0][w] + xch[16 + 8 * w + c8];
            }
          }
          cs[0][0] = cs[0][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
          cq[0][0] = cq[0][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
          // (the next unit's first barrier orders these exchange reads in front of its writers)
        }
      }
    }
  }
  // drain the DMA pieces that ran past the end of the stream before the LDS goes away
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int WM, int TN, int COLSUM, bool RES = false, bool CONV = false, int WNP = 4 / TN, bool MASKT = false>
hipError_t launch_stream(const ConvGemmParams& p, int grid, hipStream_t stream) {
  constexpr int NW = WM * WNP, BN = 32 * TN * WNP, NP = (8 * WM + BN / 8) / NW;
  constexpr bool alias = NP * 1024 >= S_SCR_BYTES;
  constexpr size_t lds_bytes = (size_t)3 * (64 * WM + BN) * S_BK * 4 + (alias ? 0 : (size_t)NW * S_SCR_BYTES) +
                               (size_t)NW * 2 * 3 * 64 * 4 + (MASKT || (!RES && !CONV) ? 4096 : 0);   // (+ a ragged batch's widths)
  static_assert(lds_bytes <= 160 * 1024, "LDS budget");
  static size_t lds_granted[WS_MAX_DEVICES] = {};
  auto kern = gemm_f32_stream_kernel<WM, TN, COLSUM, RES, CONV, WNP, MASKT>;
  hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_granted);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, stream, p);
  return hipGetLastError();
}

}  // namespace

// env WS_STREAM: 0 off; 1 = 128x128 tile, four wavefronts; 2 = 128x128 tile, eight wavefronts; 3 = 256x128 tile,
// eight wavefronts; 4 (default) = 2 or 3, whichever the cost model below prefers for the problem
int g_ws_stream = -1;
// 0: the rows behind the whole rounds go to the tile kernels as before round 6 (A/B probe)
int g_ws_stream_units = 1;

#ifdef WS_TRACE
unsigned long long* stream_trace_buffer_address() {
  unsigned long long* q = nullptr;
  (void)hipGetSymbolAddress(reinterpret_cast<void**>(&q), HIP_SYMBOL(g_stream_trace));
  return q;
}
#endif

namespace {
// Cycles per K-tile of one tile (measured, steady state) and the per-tile overhead of the first / last K-tile.
struct StreamPlan { int mode, bm, rows; long long cycles; int main_rows; };
// mode 5 (round 5): the 256 x 64 tile of the 64-channel layers, eight wavefronts of 64 x 32 (per wavefront the work of
// mode 2: 32 MFMAs per K-tile)
constexpr int bn_of(int mode) { return mode == 5 ? 64 : 128; }
StreamPlan plan_mode(const ConvGemmParams& p, int cus, int mode) {
  const int bm = mode == 3 || mode == 5 ? 256 : 128, S_BN = bn_of(mode);
  const long long per_kt = mode == 3 ? 8500 : 4400, per_tile = mode == 3 ? 5500 : 2500;
  const long long tiles_n = p.N / S_BN, tiles_m = (p.M - p.m_begin) / bm, nk = p.K / S_BK;
  const long long total = tiles_m * tiles_n, rounds = total / cus;
  StreamPlan pl = {mode, bm, 0, 0, 0};
  if (rounds < 2) return pl;
  const long long tile_time = nk * per_kt + per_tile;
  long long main_tiles_m = rounds * cus / tiles_n;
  // The tiles beyond the whole rounds: one more (partial) round here costs a whole tile time; as 64x64 tiles on the
  // tile kernel (512 block slots, ~2600 cycles per K-tile and round, measured) plus the kernel boundary they cost
  // ceil(rem64 / 512) rounds -- take the cheaper.  Rows beyond the last whole tile row always go there.
  // Round 6: the plain eight-wavefront forms (256x128 and 128x128 tiles) take whole 64-row strips of those rows themselves, as 64 x 64 units behind its
  // tiles (~1400 cycles per K-tile and round of `cus` units); the last strip may be partial (rows past M are masked, their stores fall outside the output descriptor).
  const bool units = g_ws_stream_units != 0 && (mode == 3 || mode == 2) && p.kh == 1 && !p.residual;
  const long long rem = total - main_tiles_m * tiles_n;
  const long long rest_rows = (p.M - p.m_begin) - tiles_m * bm;
  auto tile_kernel = [&](long long rows) {
    if (rows <= 0) return 0LL;
    const long long t64 = (rows + 63) / 64 * ((p.N + 63) / 64);
    return (t64 + 2 * cus - 1) / (2 * cus) * nk * 2600 + 12000;
  };
  auto beyond = [&](long long rows) {
    if (!units) return tile_kernel(rows);
    const long long strips = (rows + 63) / 64, n_units = strips * (p.N / 64);      // (a last partial strip too)
    return (n_units + cus - 1) / cus * (nk * 1400 + 4000);
  };
  long long cyc = (main_tiles_m * tiles_n / cus) * tile_time;
  const long long here = tile_time + beyond(rest_rows);
  const long long there = beyond((tiles_m - main_tiles_m) * bm + rest_rows);
  if (rem > 0 && here <= there) {
    main_tiles_m = tiles_m;
    cyc += here;
  } else {
    cyc += there;
  }
  pl.main_rows = (int)(main_tiles_m * bm);
  pl.rows = units ? p.M - p.m_begin : pl.main_rows;      // (units: every remaining row, the last strip may be partial)
  pl.cycles = cyc;
  return pl;
}
StreamPlan plan(const ConvGemmParams& p, int cus) {
  const int m = g_ws_stream;
  if (p.N % 128 != 0) return plan_mode(p, cus, 5);       // (N % 64 == 0: the 256x64 tile)
  if (p.kh == 3) return plan_mode(p, cus, 3);            // (the convolution form exists for the 256-row tiles only)
  if (p.row_len && p.residual) return plan_mode(p, cus, 3);      // (ragged + residual: instantiated for that tile only)
  if (m >= 1 && m <= 3) return plan_mode(p, cus, m);
  const StreamPlan a = plan_mode(p, cus, 2), b = plan_mode(p, cus, 3);
  if (!b.rows) return a;
  if (!a.rows) return b;
  return b.cycles <= a.cycles ? b : a;
}
}  // namespace

// (probe hooks, tools/conv_stream_probe: 0 keeps the 3x3 layers / the 64-channel layers on the implicit-GEMM tile kernels)
int g_ws_stream_conv = 1;
int g_ws_stream64 = 1;

// a 3x3 / stride 1 / pad 1 / dilation 1 convolution over channels-last images whose K-tiles lie inside one tap
bool gemm_f32_stream_is_conv3(const ConvGemmParams& p) {
  return g_ws_stream_conv != 0 && p.prec == 0 && !p.A16 && !p.A2 && !p.pre_scale && p.kh == 3 && p.kw == 3 &&
         p.stride_h == 1 && p.stride_w == 1 && p.pad_h == 1 && p.pad_w == 1 && p.dil_h == 1 && p.dil_w == 1 &&
         p.Hin == p.Hout && p.Win == p.Wout && p.Cin % S_BK == 0 && p.K == 9 * p.Cin && p.D && !p.D16 && !p.D2_16 &&
         !p.bias_img && !p.residual16 && !p.seg_scale && !p.pool_partial && p.splitk <= 1 &&
         (!p.row_len || (p.Wout >= 8 && p.M / (p.Hout * p.Wout) <= 1024)) &&      // (ragged: the kernel's MASK)
         (p.act == ACT_NONE || p.act == ACT_RELU) && p.a_zero_off > 0 &&
         p.Cin <= 512 &&     // (the zero pad behind an activation buffer is 512 floats: resnet_model.hip reserve())
         (long long)p.M * p.lda * 4 + ((long long)p.Win + 2) * p.lda * 4 < (1LL << 32) &&
         p.a_zero_off + ((long long)p.Win + 2) * p.lda * 4 + 4LL * p.Cin + 16 < (1LL << 32);
}

// Rows of [p.m_begin, p.M) that the persistent kernel should take (whole tile rows: whole rounds of tiles over
// `cus` workgroups, or all of them); 0 = not this kernel's problem.
int gemm_f32_stream_rows(const ConvGemmParams& p, int cus) {
  if (g_ws_stream < 0) {
    const char* ev = getenv("WS_STREAM");
    g_ws_stream = ev ? atoi(ev) : 4;
  }
  if (g_ws_stream <= 0) return 0;
  const bool conv3 = gemm_f32_stream_is_conv3(p);
  // a ragged batch (the kernel's MASK): at most 1024 utterances, maps of >= 8 pixels and >= 8 columns
  const bool mask_ok = !p.row_len || (p.Hout * p.Wout >= 8 && p.Wout >= 8 && p.M / (p.Hout * p.Wout) <= 1024);
  const bool plain = p.prec == 0 && !p.A16 && !p.A2 && !p.pre_scale && p.kh == 1 && p.kw == 1 && p.stride_h == 1 &&
                     p.stride_w == 1 && p.pad_h == 0 && p.pad_w == 0 && p.K == p.Cin && p.D && !p.D16 && !p.D2_16 &&
                     !p.bias_img && !p.residual16 && mask_ok && !p.seg_scale && !p.pool_partial &&
                     p.splitk <= 1 && (p.act == ACT_NONE || p.act == ACT_RELU);
  if (!plain && !conv3) return 0;
  // the convolution form: the 256-row tiles, no column sums (rows behind the last whole tile go to the tile kernels
  // like the plain form's; their taps may read any input pixel, so only a launch from row 0 is taken)
  if (conv3 && (p.colsum || p.D2 || (g_ws_stream != 3 && g_ws_stream != 4) || p.m_begin != 0)) return 0;
  if (p.N % 64 != 0 || p.K % S_BK != 0 || p.K < 4 * S_BK) return 0;
  // the 256x64 tile: plain / convolution layers without column sums or a second output, default dispatch only
  if (p.N % 128 != 0 && (g_ws_stream64 <= 0 || g_ws_stream != 4 || p.colsum || p.D2 || (p.residual && !conv3))) return 0;
  if (p.colsum && p.Hout * p.Wout < 64) return 0;
  if ((p.m_begin & 63) || ((p.lda | p.a_off | p.ldd | p.d_off) & 3)) return 0;
  if (p.D2 && (((p.ldd2 | p.d2_off) & 3) || (p.d2_col0 & 31))) return 0;
  // a residual: the eight-wavefront forms only (no registers for it beside the 4-wave form's 64 accumulators), no
  // column sums / second output with it (no caller has both)
  if (p.residual && (g_ws_stream == 1 || p.colsum || p.D2 || ((p.ldr | p.r_off) & 3) ||
                     (long long)p.M * p.ldr * 4 >= (1LL << 32)))
    return 0;
  // 32-bit byte offsets
  // (+ 64 rows: a row unit's partial last strip forms offsets of rows that do not exist; they must not wrap)
  if ((long long)(p.M + 64) * p.lda * 4 >= (1LL << 32) || (long long)(p.M + 64) * p.ldd * 4 >= (1LL << 32) ||
      (long long)p.N * p.ldw * 4 >= (1LL << 32) || (p.D2 && (long long)(p.M + 64) * p.ldd2 * 4 >= (1LL << 32)))
    return 0;
  return plan(p, cus).rows;
}

hipError_t launch_gemm_f32_stream(const ConvGemmParams& p0, int rows, int cus, hipStream_t stream) {
  ConvGemmParams p = p0;
  const StreamPlan pl = plan(p0, cus);
  if (pl.rows != rows) return hipErrorInvalidValue;      // (rows must come from gemm_f32_stream_rows)
  p.tail_begin = p.m_begin + pl.main_rows;               // the first row of the 64 x 64 units
  p.n_units = (rows - pl.main_rows + 63) / 64 * (p.N / 64);
  const int mode = pl.mode, bm = pl.bm;
  p.n_big = pl.main_rows / bm * (p.N / bn_of(mode));
  const int grid = p.n_big < cus ? p.n_big : cus;
  if (dispatch_log_enabled()) {
    char k[96];
    if (p.n_units)
      snprintf(k, sizeof(k), "gemm_f32_stream_kernel<%dx%d tile, %d waves> tiles=%d + %d 64x64 units", bm, bn_of(mode),
               mode == 1 ? 4 : 8, p.n_big, p.n_units);
    else
      snprintf(k, sizeof(k), "gemm_f32_stream_kernel<%dx%d tile, %d waves%s> tiles=%d", bm, bn_of(mode), mode == 1 ? 4 : 8,
               p.kh == 3 ? ", conv" : "", p.n_big);
    dispatch_log_note(p, k);
  }
  const int cs = !p.colsum ? 0 : (p.colsumsq ? 2 : 1);
  const bool mk = p.row_len != nullptr;          // a ragged batch: the RES / CONV forms have masked twins (MASKT)
  if (mode == 5) {
    if (p.kh == 3) {
      if (mk) return p.residual ? launch_stream<4, 1, 0, true, true, 2, true>(p, grid, stream)
                                : launch_stream<4, 1, 0, false, true, 2, true>(p, grid, stream);
      return p.residual ? launch_stream<4, 1, 0, true, true, 2>(p, grid, stream)
                        : launch_stream<4, 1, 0, false, true, 2>(p, grid, stream);
    }
    return p.residual ? hipErrorInvalidValue : launch_stream<4, 1, 0, false, false, 2>(p, grid, stream);
  }
  if (p.kh == 3) {
    if (mode != 3) return hipErrorInvalidValue;
    if (mk) return p.residual ? launch_stream<4, 2, 0, true, true, 2, true>(p, grid, stream)
                              : launch_stream<4, 2, 0, false, true, 2, true>(p, grid, stream);
    return p.residual ? launch_stream<4, 2, 0, true, true>(p, grid, stream)
                      : launch_stream<4, 2, 0, false, true>(p, grid, stream);
  }
  if (p.residual) {
    if (mk) return mode == 3 ? launch_stream<4, 2, 0, true, false, 2, true>(p, grid, stream) : hipErrorInvalidValue;
    if (mode == 2) return launch_stream<2, 1, 0, true>(p, grid, stream);
    if (mode == 3) return launch_stream<4, 2, 0, true>(p, grid, stream);
    return hipErrorInvalidValue;
  }
  if (mode == 1) return cs == 2 ? launch_stream<2, 2, 2>(p, grid, stream)
                                : cs ? launch_stream<2, 2, 1>(p, grid, stream) : launch_stream<2, 2, 0>(p, grid, stream);
  if (mode == 2) return cs == 2 ? launch_stream<2, 1, 2>(p, grid, stream)
                                : cs ? launch_stream<2, 1, 1>(p, grid, stream) : launch_stream<2, 1, 0>(p, grid, stream);
  return cs == 2 ? launch_stream<4, 2, 2>(p, grid, stream)
                 : cs ? launch_stream<4, 2, 1>(p, grid, stream) : launch_stream<4, 2, 0>(p, grid, stream);
}

}  // namespace wsamd
