// Implicit-GEMM convolution on channels-last activations for gfx950 (MI355X).
//
// Replaces the conv1d / conv2d / linear ATen calls of the reference forward
// (wespeaker/models/ecapa_tdnn.py:85-106 Conv1dReluBn, :58-78 Res2 convs, :196 cat conv;
//  pooling_layers.py:108-117 ASTP 1x1 convs; resnet.py Conv2d 3x3/1x1; campplus.py TDNN convs).
//
// Design (MI355X-first, not a port of any CUDA tiling):
//  * activations are [pixel][channel] (channels-last), weights [cout][tap*Cin + ci]: both GEMM
//    operands are K-contiguous, so global loads are 16 B/lane along K and both LDS tiles are
//    read with ds_read_b128 along K.
//  * two contraction back-ends behind one tiling (template PREC):
//      PREC 0  v_mfma_f32_32x32x2_f32: exact fp32 products (bit-identical to an fmaf chain),
//              157 TF peak.  One ds_read_b128 feeds 4 MFMAs per tile (k-permuted lanes).
//      PREC 2  "f16": operands rounded to binary16 (hi planes only), ONE MFMA pass, fp32 accumulation
//              -- the arithmetic of the reference's own TensorRT-fp16 GPU runtime.  Plain 1x1 layers
//              take dedicated K-tile-64 kernels further down: gemm_f16_dma_kernel when the producer
//              left a binary16 copy of the activations (both operands by LDS-DMA), gemm_f16_kernel
//              otherwise (fp32 activations converted while staged);
//      PREC 1  "f16x3": every fp32 operand is split x = hi + lo (hi = half(x), lo = half(x - hi),
//              22 significant bits) and the product is formed as hi*hi + hi*lo + lo*hi on
//              v_mfma_f32_32x32x16_f16 with fp32 accumulation: ~2^-21 relative product error
//              (fp32-grade) at 3/16 of the fp32-MFMA issue cost.  Weights are pre-split on the
//              host; activations are split while they are staged into LDS.
//  * 128x128x32 tile, 4 wavefronts x (2x2) 32x32 accumulators (128x64 and 64x64 variants);
//    LDS row strides (36 floats / 40 halfs) make every ds_read_b128 lane group hit 16 distinct
//    16-B slots (conflict-free).
//  * im2col-free: a K-chunk of 4 floats lies inside one filter tap; the tap's pixel offset and the
//    zero-padding predicate are evaluated per 16-B chunk while staging (masked chunks are read
//    from a 16-B zero page, so the loads stay unconditional and pipelined).  1x1 / stride-1 layers
//    (most of the FLOPs) use running row pointers with no per-tile address arithmetic (SIMPLE).
//  * register-prefetch double buffering: tile k+1's global loads are issued before tile k's
//    MFMAs and written to the other LDS buffer after them -> one barrier per K-tile.
//  * XCD-aware bijective tile order; tail peeling: rows beyond the last full round of
//    (2 tiles per CU) go to a 64x64-tile launch (a partial last round costs a whole tile time).
//  * one shared epilogue (gemm_epilogue) transposed through LDS -> 16-B stores; fused bias
//    (+ per-utterance bias), residual, ReLU/tanh, BN-after-activation affine, dual store (Res2
//    pass-through split), binary16 twins of the stores, segment mask, deterministic per-64-row
//    column sums (SE mean / context statistics), online-softmax pooling partials, or raw split-K
//    partials.
#include "kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace wsamd {

// ------------------------------------------------------------------------------------- dispatch log
namespace {
std::atomic<int> g_dlog_on{-1};
std::mutex g_dlog_mu;
std::map<std::string, long> g_dlog;
}  // namespace
bool dispatch_log_enabled() {
  int v = g_dlog_on.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* ev = getenv("WS_DISPATCH_LOG");
    v = ev && atoi(ev) != 0;
    g_dlog_on.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}
void dispatch_log_enable(bool on) { g_dlog_on.store(on ? 1 : 0, std::memory_order_relaxed); }
void dispatch_log_clear() {
  std::lock_guard<std::mutex> lk(g_dlog_mu);
  g_dlog.clear();
}
void dispatch_log_note(const ConvGemmParams& p, const char* kernel) {
  char key[320];
  // the problem as the dispatcher sees it: rows of this launch, N, K, taps / strides, back-end, operand forms,
  // fused extras
  snprintf(key, sizeof(key),
           "rows=%d N=%d K=%d k=%dx%d s=%dx%d d=%dx%d prec=%d A=%s out=%s%s%s%s%s%s%s%s%s -> %s",
           p.M - p.m_begin, p.N, p.K, p.kh, p.kw, p.stride_h, p.stride_w, p.dil_h, p.dil_w, p.prec,
           p.A16 ? "f16" : "f32", p.D ? "f32" : "", p.D16 ? "f16" : "", p.A2 ? " +A2" : "",
           p.pre_scale ? " +pre" : "", p.residual || p.residual16 ? " +res" : "", p.colsum ? " +colsum" : "",
           p.pool_partial ? " +pool" : "", p.splitk > 1 ? " +splitk" : "", p.row_len ? " +mask" : "", kernel);
  std::lock_guard<std::mutex> lk(g_dlog_mu);
  ++g_dlog[key];
}
void dispatch_log_note_text(const char* key) {      // kernels outside the conv-GEMM dispatcher (cam_dense.hip)
  std::lock_guard<std::mutex> lk(g_dlog_mu);
  ++g_dlog[key];
}
size_t dispatch_log_dump(char* buf, size_t cap) {
  std::lock_guard<std::mutex> lk(g_dlog_mu);
  size_t need = 0;
  for (const auto& kv : g_dlog) {
    char line[400];
    const int n = snprintf(line, sizeof(line), "%s  x%ld\n", kv.first.c_str(), kv.second);
    if (buf && need + (size_t)n < cap) memcpy(buf + need, line, (size_t)n);
    need += (size_t)n;
  }
  if (buf && cap) buf[need < cap ? need : cap - 1] = 0;
  return need + 1;
}
#define WS_DLOG(p, ...)                                         \
  do {                                                          \
    if (dispatch_log_enabled()) {                               \
      char _k[96];                                              \
      snprintf(_k, sizeof(_k), __VA_ARGS__);                    \
      dispatch_log_note(p, _k);                                 \
    }                                                           \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDS_STRIDE = BK + 4;   // fp32 path: floats per LDS row
constexpr int HS = BK + 8;           // f16x3 path: halfs per LDS row (80 B)

template <int BM, int BN, int PREC>
constexpr size_t tile_lds_bytes() {
  // per buffer: A and W tiles; PREC 1 keeps a hi and a lo half plane per tile (same bytes as fp32)
  const size_t stage = PREC == 0 ? (size_t)(BM + BN) * LDS_STRIDE * 4
                                  : (size_t)(BM + BN) * HS * 2 * (PREC == 1 ? 2 : 1);
  const size_t epi = (size_t)BM * (BN + 4) * 4;
  return 2 * stage > epi ? 2 * stage : epi;
}

#ifdef WS_TRACE
extern __device__ unsigned long long g_trace[64 * 8];
#define WS_EMARK(slot) \
  if (BM == 128 && p.pool_partial && blockIdx.x == 8 && threadIdx.x == 0) g_trace[62 * 8 + (slot)] = __builtin_readcyclecounter();
#else
#define WS_EMARK(slot)
#endif
// ---------------------------------------------------------------------------------------------
// Shared epilogue of the MFMA GEMM kernels: takes the 32x32 accumulator blocks of the workgroup's
// BM x BN tile, returns after all global stores (it ends on code every thread executes).
// `tid` is the thread's index inside the group of 64*WM*WN threads that owns this tile (= threadIdx.x
// for the one-tile kernels; the 256x256 kernel runs four such groups).  Every thread of the
// WORKGROUP must call it the same number of times (the barriers are workgroup-wide); a group whose
// `active` is false only keeps the barriers company -- unless it passes `stid` (its index among SNT threads
// that finish the tile's rows: the 256x256 kernels let both wave rows store each quadrant).  POOL = false
// compiles the fused-pooling path out.
template <int BM, int BN, int WM, int WN, bool POOL = true, int SNT = 64 * WM * WN, bool ROWOPS_ = POOL>
__device__ __forceinline__ void gemm_epilogue(const ConvGemmParams& p,
                                              f32x16 (&acc)[BM / WM / 32][BN / WN / 32],
                                              float* lds, int m0, int n0, int tid, bool active = true,
                                              int stid = -1) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int li = lane & 31, lh = lane >> 5;
  const int HW = p.Hout * p.Wout;

  // ---------------- epilogue.  The kernels issue mfma(W fragment, A fragment): the accumulator block is
  // C^T[n][m] with the 32x32 C/D layout col = lane & 31 (= output pixel m), row = (reg & 3) +
  // 8 * (reg >> 2) + 4 * (lane >> 5) (= output channel n).
  if (p.splitk > 1) {
    if (!active) return;
#pragma unroll
    for (int in = 0; in < TN; ++in) {
#pragma unroll
      for (int im = 0; im < TM; ++im) {
        const int m = m0 + (wm * TM + im) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + (wn * TN + in) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < p.M && n < p.N) p.partial[((long long)blockIdx.y * p.M + m) * p.N + n] = acc[im][in][r];
        }
      }
    }
    return;
  }
  // Transpose the accumulators through LDS (the staging buffers are free now) so that every lane
  // finishes 4 consecutive output channels of one row: 16-B global stores, 512 B contiguous per
  // output row, instead of 4-B stores (which are ~6x slower per byte on gfx950).
  constexpr int ES = BN + 4;
  WS_EMARK(0)
  if (active) {
    float* Es = lds;
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // weights are the MFMA's A operand: a lane holds 4 consecutive output channels (rows of the
          // C^T block) of ONE output pixel (column li) per register quad -> one 16-B LDS store each
          const int row = (wm * TM + im) * 32 + li;
          const int col = (wn * TN + in) * 32 + 8 * g + 4 * lh;
          *reinterpret_cast<f32x4*>(&Es[row * ES + col]) =
              (f32x4){acc[im][in][4 * g], acc[im][in][4 * g + 1], acc[im][in][4 * g + 2], acc[im][in][4 * g + 3]};
        }
  }
  __syncthreads();
  WS_EMARK(1)
  {
    constexpr int C4 = BN / 4;                 // float4 columns per tile row
    // the threads that finish rows: the tile's own group, or (256x256 kernels) SNT threads of both wave rows
    const int st = stid >= 0 ? stid : tid;
    const bool storer = stid >= 0 ? true : active;
    constexpr int RPP = SNT / C4;              // rows per pass
    const int c4 = st % C4, rr = st / C4;
    const int n = n0 + c4 * 4;
    constexpr int NH = BM / 64;                // 64-row halves (column-sum granularity)
    f32x4 cs[NH][2];
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) cs[hf][0] = cs[hf][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // softmax-pooling partials (ASTP fused into the logit GEMM): running max + 3 weighted sums
    f32x4 pm[NH][2], p0[NH][2], p1[NH][2], p2[NH][2];
#pragma unroll
    for (int hf = 0; hf < NH; ++hf)
#pragma unroll
      for (int wh = 0; wh < 2; ++wh) {
        pm[hf][wh] = (f32x4){-1e30f, -1e30f, -1e30f, -1e30f};
        p0[hf][wh] = p1[hf][wh] = p2[hf][wh] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    if (POOL && p.pool_partial) {
      // ASTP fused into the logit GEMM: the logits are never written to HBM.  All h rows of a
      // 64-row half are fetched up front (independent 16-B loads in flight), then folded into the
      // online-softmax tuples of the (half, image part) groups.
      if (storer && n < p.N) {
        // All h rows of BOTH 64-row halves are requested before anything is consumed, and the
        // binary16 / fp32 choice is made outside the load loops: with the branch inside, hipcc put an
        // s_waitcnt vmcnt(0) behind every single load (16 serialized HBM round trips per tile, 2/3 of
        // this kernel's time by s_memtime stamps).  (Requesting them before the K loop was measured too:
        // the wait only moves, the tile takes the same 30 k cycles.)
        constexpr int RI = 64 / RPP;
        f16x4 hraw[NH][RI];
        f32x4 hall[NH][RI];
        if (p.pool_h16) {
#pragma unroll
          for (int hf = 0; hf < NH; ++hf)
#pragma unroll
            for (int i = 0; i < RI; ++i) {
              const int m = m0 + hf * 64 + rr + RPP * i;
              const int mc = m < p.M ? m : p.M - 1;
              hraw[hf][i] = *reinterpret_cast<const f16x4*>(p.pool_h16 + (long long)mc * p.ldh + n);
            }
        } else {
#pragma unroll
          for (int hf = 0; hf < NH; ++hf)
#pragma unroll
            for (int i = 0; i < RI; ++i) {
              const int m = m0 + hf * 64 + rr + RPP * i;
              const int mc = m < p.M ? m : p.M - 1;
              hall[hf][i] = *reinterpret_cast<const f32x4*>(p.pool_h + (long long)mc * p.ldh + n);
            }
        }
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
          const int mh = m0 + hf * 64;
          const int imgA = mh / HW;
          const int rb = (imgA + 1) * HW - mh;
          // ragged batch: only the utterance's own frames enter its softmax (H = 1: HW = frames per slot)
          int lenA = 0x7fffffff, lenB = 0x7fffffff;
          if (p.row_len) {
            const int last = (p.M - 1) / HW;
            lenA = p.row_len[imgA < last ? imgA : last];
            lenB = p.row_len[imgA + 1 < last ? imgA + 1 : last];
          }
          const int tA0 = mh - imgA * HW;
          auto live = [&](int rl) { return mh + rl < p.M && (rl < rb ? tA0 + rl < lenA : rl - rb < lenB); };
          f32x4 hv[RI];
#pragma unroll
          for (int i = 0; i < RI; ++i) {
            if (p.pool_h16) {
#pragma unroll
              for (int q = 0; q < 4; ++q) hv[i][q] = (float)hraw[hf][i][q];
            } else {
              hv[i] = hall[hf][i];
            }
          }
          // two passes over the lane's RI rows of this half: the maxima of the two image parts first
          // (LDS reads only), then ONE exponential per logit and no running rescale.  The column bias is
          // NOT added: it is constant over the rows, so it cancels in softmax_t (the unfused path adds it
          // and gets the same weights); the exponent runs in the exp2 domain with the max folded into one
          // fma, h*h is formed off the critical path -- 8 VALU ops per logit instead of 12 (this loop was
          // VALU-bound: 12.5 k of the tile's 30 k cycles by s_memtime stamps).
          constexpr float LOG2E = 1.4426950408889634f;
          f32x4 mx0 = {-1e30f, -1e30f, -1e30f, -1e30f}, mx1 = mx0;
#pragma unroll
          for (int i = 0; i < RI; ++i) {
            const int rl = rr + RPP * i;
            if (live(rl)) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[(hf * 64 + rl) * ES + c4 * 4]);
              const bool second = rl >= rb;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                if (second) mx1[q] = fmaxf(mx1[q], v[q]);
                else mx0[q] = fmaxf(mx0[q], v[q]);
              }
            }
          }
          pm[hf][0] = mx0;
          pm[hf][1] = mx1;
          const f32x4 ml0 = mx0 * LOG2E, ml1 = mx1 * LOG2E;
#pragma unroll
          for (int i = 0; i < RI; ++i) {
            const int rl = rr + RPP * i;
            if (live(rl)) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[(hf * 64 + rl) * ES + c4 * 4]);
              const bool second = rl >= rb;
              const f32x4 arg = v * LOG2E - (second ? ml1 : ml0);
              const f32x4 h = hv[i], hh = h * h;
              f32x4 pe;
#pragma unroll
              for (int q = 0; q < 4; ++q) pe[q] = __builtin_amdgcn_exp2f(arg[q]);
              if (second) {
                p0[hf][1] += pe; p1[hf][1] += pe * h; p2[hf][1] += pe * hh;
              } else {
                p0[hf][0] += pe; p1[hf][0] += pe * h; p2[hf][0] += pe * hh;
              }
            }
          }
        }
      }
    } else
    if (storer && n < p.N) {                   // N % 4 == 0 (checked on the host)
      f32x4 bias = {0.f, 0.f, 0.f, 0.f}, ps = {1.f, 1.f, 1.f, 1.f}, pb = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.post_scale) {
        ps = *reinterpret_cast<const f32x4*>(p.post_scale + n);
        pb = *reinterpret_cast<const f32x4*>(p.post_shift + n);
      }
      const bool to_d2 = p.D2 && n >= p.d2_col0;
      auto store_row = [&](int m, const f32x4& v) {
        if (p.D) *reinterpret_cast<f32x4*>(p.D + (long long)m * p.ldd + p.d_off + n) = v;
        if (p.D16) {
          f16x4 hv;
#pragma unroll
          for (int q = 0; q < 4; ++q) hv[q] = (_Float16)v[q];
          *reinterpret_cast<f16x4*>(p.D16 + (long long)m * p.ldd16 + p.d_off + n) = hv;
        }
        if (to_d2) {
          *reinterpret_cast<f32x4*>(p.D2 + (long long)m * p.ldd2 + p.d2_off + (n - p.d2_col0)) = v;
          if (p.D2_16) {
            f16x4 hv;
#pragma unroll
            for (int q = 0; q < 4; ++q) hv[q] = (_Float16)v[q];
            *reinterpret_cast<f16x4*>(p.D2_16 + (long long)m * p.ldd2_16 + p.d2_off + (n - p.d2_col0)) = hv;
          }
        }
      };
#pragma unroll
      for (int hf = 0; hf < NH; ++hf) {
        // rows >= rb (relative to this 64-row half) belong to the next image; HW >= 64 on the host
        const int mh = m0 + hf * 64;
        const int rb = (mh / HW + 1) * HW - mh;
        const bool rowops = ROWOPS_ && (p.bias_img || p.residual || p.residual16 || p.seg_scale || p.row_len);
        if (!rowops) {
          // plain layer: row by row, nothing to wait for
#pragma unroll 4
          for (int rl = rr; rl < 64; rl += RPP) {
            const int row = hf * 64 + rl;
            const int m = m0 + row;
            if (m >= p.M) break;
            f32x4 v = *reinterpret_cast<const f32x4*>(&lds[row * ES + c4 * 4]) + bias;
            if (p.act == ACT_RELU) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = relu_f(v[q]);
            } else if (p.act == ACT_TANH) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = tanhf(v[q]);
            }
            if (p.post_scale) v = v * ps + pb;
            store_row(m, v);
            if (p.colsum) {
              if (rl < rb) cs[hf][0] += v; else cs[hf][1] += v;
            }
          }
          continue;
        }
        // Rows in groups of G: the optional per-row operands (image bias, residuals, segment scale) of a
        // whole group are requested first, each kind behind ONE wave-uniform branch, then the group is
        // finished.  With the branches inside the row loop hipcc put an s_waitcnt vmcnt(0) behind every
        // load -- one exposed HBM round trip per row, which also waited for the previous row's stores.
        // (POOL = false callers -- the 256x256 kernels, whose accumulators are still live here -- are never
        // given these operands by the dispatcher; the code is compiled out for them.)
        constexpr bool ROWOPS = ROWOPS_;
        constexpr int GMAX = SNT == 64 * WM * WN ? 4 : 2;     // (256x256 kernels: accumulators still live)
        constexpr int RPT = 64 / RPP, G = RPT < GMAX ? RPT : GMAX;
#pragma unroll
        for (int g0 = 0; g0 < RPT; g0 += G) {
          int mrow[G], rlen[G];
          f32x4 bi[G], r32[G], ss[G], vv[G];
          f16x4 r16[G];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const int m = m0 + hf * 64 + rr + RPP * (g0 + g);
            mrow[g] = m < p.M ? m : p.M - 1;       // clamped: loads stay unconditional, stores are masked
          }
          if (ROWOPS && p.bias_img) {
#pragma unroll
            for (int g = 0; g < G; ++g)
              bi[g] = *reinterpret_cast<const f32x4*>(p.bias_img + (long long)(mrow[g] / HW) * p.N + n);
          }
          if (ROWOPS && p.residual) {
#pragma unroll
            for (int g = 0; g < G; ++g)
              r32[g] = *reinterpret_cast<const f32x4*>(p.residual + (long long)mrow[g] * p.ldr + p.r_off + n);
          }
          if (ROWOPS && p.residual16) {
#pragma unroll
            for (int g = 0; g < G; ++g)
              r16[g] = *reinterpret_cast<const f16x4*>(p.residual16 + (long long)mrow[g] * p.ldr + p.r_off + n);
          }
          if (ROWOPS && p.row_len) {
#pragma unroll
            for (int g = 0; g < G; ++g) rlen[g] = p.row_len[mrow[g] / HW];
          }
          if (ROWOPS && p.seg_scale) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const int img = mrow[g] / HW, ox = (mrow[g] - img * HW) % p.Wout;
              ss[g] = *reinterpret_cast<const f32x4*>(
                  p.seg_scale + ((long long)img * p.segs_per_img + ox / p.seg_len) * p.N + n);
            }
          }
          // every row of the group is finished in registers ...
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const int rl = rr + RPP * (g0 + g);
            const int row = hf * 64 + rl;
            const int m = m0 + row;
            vv[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (m >= p.M) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(&lds[row * ES + c4 * 4]) + bias;
            if (ROWOPS && p.bias_img) v += bi[g];
            if (ROWOPS && p.residual) v += r32[g];
            if (ROWOPS && p.residual16) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] += (float)r16[g][q];
            }
            if (p.act == ACT_RELU) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = relu_f(v[q]);
            } else if (p.act == ACT_TANH) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = tanhf(v[q]);
            }
            if (p.post_scale) v = v * ps + pb;
            if (ROWOPS && p.seg_scale) v *= ss[g];
            if (ROWOPS && p.row_len) {
              // ragged batch: pixels beyond the utterance's own width are padding -> stored as zeros, so
              // that every later convolution tap and reduction sees the zero padding of a batch-1 run
              const int img = m / HW;
              if ((m - img * HW) % p.Wout >= rlen[g]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (p.colsum) {
              if (rl < rb) cs[hf][0] += v; else cs[hf][1] += v;
            }
            if constexpr (ROWOPS) vv[g] = v;
            else store_row(m, v);
          }
          // ... and only then stored: a load's first use waits for everything older in the VMEM queue,
          // which must not include this group's own stores
          if constexpr (ROWOPS) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const int m = m0 + hf * 64 + rr + RPP * (g0 + g);
              if (m < p.M) store_row(m, vv[g]);
            }
          }
        }
      }
    }
    if (p.colsum) {
      // Deterministic per-(64-row tile, image) column sums of the stored values (SE / statistics
      // pooling without re-reading the tensor): fold the RPP row phases through LDS, then one
      // plain store per column -> colsum[(tile64*2 + which)][N].
      __syncthreads();                                  // everyone is done reading the E tile
      float* red = lds;                                 // [NH][2][RPP][BN]
      if (storer) {
#pragma unroll
        for (int hf = 0; hf < NH; ++hf)
#pragma unroll
          for (int wh = 0; wh < 2; ++wh)
            *reinterpret_cast<f32x4*>(&red[((hf * 2 + wh) * RPP + rr) * BN + c4 * 4]) = cs[hf][wh];
      }
      __syncthreads();
      for (int o = storer ? st : NH * 2 * BN; o < NH * 2 * BN; o += SNT) {
        const int hw = o / BN, col = o - hw * BN;        // hw = hf*2 + which
        float sacc = 0.f;
#pragma unroll
        for (int q = 0; q < RPP; ++q) sacc += red[(hw * RPP + q) * BN + col];
        if (n0 + col < p.N)
          p.colsum[((long long)(m0 / 64) * 2 + hw) * p.N + n0 + col] = sacc;
      }
    }
    WS_EMARK(2)
    if (POOL && p.pool_partial) {
      // fold the RPP row phases: (max, s0, s1, s2) tuples combined with the usual rescaling;
      // one [4]-tuple per (64-row tile, image part, column) -> pool_partial[tile64*2 + which][N][4]
      __syncthreads();
      WS_EMARK(3)
      float* red = lds;                                 // [NH*2][RPP][4][BN]
#pragma unroll
      for (int hf = 0; hf < NH; ++hf)
#pragma unroll
        for (int wh = 0; wh < 2; ++wh) {
          float* r0p = &red[(((hf * 2 + wh) * RPP + rr) * 4) * BN + c4 * 4];
          *reinterpret_cast<f32x4*>(r0p) = pm[hf][wh];
          *reinterpret_cast<f32x4*>(r0p + BN) = p0[hf][wh];
          *reinterpret_cast<f32x4*>(r0p + 2 * BN) = p1[hf][wh];
          *reinterpret_cast<f32x4*>(r0p + 3 * BN) = p2[hf][wh];
        }
      __syncthreads();
      WS_EMARK(4)
      for (int o = st; o < NH * 2 * BN; o += SNT) {
        const int hw = o / BN, col = o - hw * BN;
        float mx = -1e30f;
#pragma unroll
        for (int q = 0; q < RPP; ++q) mx = fmaxf(mx, red[((hw * RPP + q) * 4) * BN + col]);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int q = 0; q < RPP; ++q) {
          const float* e = &red[((hw * RPP + q) * 4) * BN + col];
          const float sc = __expf(e[0] - mx);
          a0 += e[BN] * sc; a1 += e[2 * BN] * sc; a2 += e[3 * BN] * sc;
        }
        if (n0 + col < p.N) {
          f32x4 outv = {mx, a0, a1, a2};
          *reinterpret_cast<f32x4*>(
              p.pool_partial + (((long long)(m0 / 64) * 2 + hw) * p.N + n0 + col) * 4) = outv;
        }
      }
      WS_EMARK(5)
    }
  }
}
// One BM x BN output tile.  bid / nblk: this workgroup's index and the number of workgroups of ITS tile
// class (the plain kernel passes blockIdx.x / gridDim.x; the dual kernel below runs a 128x128 class and a
// 64x64 class in one grid); m_begin: first output row of the class.
// (A K-tile-16 / three-workgroups-per-CU fp32 form with a two-half epilogue was measured and dropped: 291 vs
// 280 us on the N = K = 512 layer, 19 spilled VGPRs at the 168-register budget.)
template <int BM, int BN, int WM, int WN, bool HAS_A2, bool HAS_PRE, bool SIMPLE, int PREC>
__device__ __forceinline__ void conv_gemm_body(const ConvGemmParams& p, float* lds, const int bid,
                                               const int nblk, const int m_begin) {
  constexpr int S = LDS_STRIDE;
  constexpr int NT = 64 * WM * WN;                    // threads per workgroup (4 or 8 wavefronts)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AROWS = NT / 8, WROWS = NT / 4;       // rows covered per staging pass
  constexpr int A_IT = BM / AROWS;                    // float4 chunks of A per thread per K-tile
  // W chunks per thread (per plane for PREC 1); a 32-column tile uses only the first BN weight rows
  constexpr int W_IT = PREC == 0 ? BN / AROWS : (BN / WROWS > 0 ? BN / WROWS : 1);
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
  static_assert(A_IT >= 1 && W_IT >= 1, "tile too small for the thread count");

  const int tid = threadIdx.x;
  const int tiles_n = (p.N + BN - 1) / BN;
  // XCD-aware tile order: the dispatcher places block b on XCD b % 8, each XCD has a private L2.
  // Give every XCD one CONTIGUOUS range of the (tile_m major, tile_n minor) work order, so the
  // blocks that run concurrently on an XCD share their A row-panel (all n-tiles of an m-tile) and
  // the weight panels; placement only affects speed, the map is a bijection for any grid size.
  int work;
  {
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    work = xcd * q + (xcd < r ? xcd : r) + local;
  }
  const int tile_m = work / tiles_n;
  const int tile_n = work - tile_m * tiles_n;
  const int m0 = m_begin + tile_m * BM, n0 = tile_n * BN;

  // ---------------- staging roles: thread -> (16-B chunk kc of the K-tile, rows r0 + 32 i)
  const int kc = tid & 7;
  const int r0 = tid >> 3;
  // f16x3 weight planes: thread -> (16-B chunk of 8 halfs wc, rows wr0 + 64 i)
  const int wc = tid & 3;
  const int wr0 = tid >> 2;
  const int HW = p.Hout * p.Wout;
  int a_pix[A_IT], a_iy[A_IT], a_ix[A_IT];
  if (!SIMPLE) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int m = m0 + r0 + AROWS * i;
      bool ok = m < p.M;
      int mm = ok ? m : 0;
      int img = mm / HW;
      int rem = mm - img * HW;
      int oy = rem / p.Wout;
      int ox = rem - oy * p.Wout;
      int iy0 = oy * p.stride_h - p.pad_h;
      int ix0 = ox * p.stride_w - p.pad_w;
      a_pix[i] = (img * p.Hin + iy0) * p.Win + ix0;
      a_iy[i] = ok ? iy0 : -(1 << 28);   // forces the bounds predicate false for rows >= M
      a_ix[i] = ix0;
    }
  }

  const int nk_total = (p.K + BK - 1) / BK;
  int kt_begin = 0, kt_end = nk_total;
  if (p.splitk > 1) {
    kt_begin = (int)(((long long)nk_total * blockIdx.y) / p.splitk);
    kt_end = (int)(((long long)nk_total * (blockIdx.y + 1)) / p.splitk);
  }

  // SIMPLE mode (1x1, stride 1, no padding, K % 32 == 0: most of the FLOPs): running per-row
  // pointers, no tap arithmetic and no predicates inside the K loop.
  const float* sa_ptr[A_IT];
  int sa_inc[A_IT];
  if (SIMPLE) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int m = m0 + r0 + AROWS * i;
      const bool ok = m < p.M;
      sa_ptr[i] = ok ? p.A + (long long)m * p.lda + p.a_off + (long long)kt_begin * BK + kc * 4 : p.zeros;
      sa_inc[i] = ok ? BK : 0;
    }
  }
  // SIMPLE + PRE (pre-activated 1x1 layers, CAM++'s dense / transit layers): the BN vectors of this thread's
  // four K columns run along with the rows
  const float* sps_ptr = nullptr;
  const float* spt_ptr = nullptr;
  if (SIMPLE && HAS_PRE) {
    sps_ptr = p.pre_scale + (long long)kt_begin * BK + kc * 4;
    spt_ptr = p.pre_shift + (long long)kt_begin * BK + kc * 4;
  }
  // running weight pointers (all modes): fp32 rows, or hi/lo half planes
  const float* sw_ptr[W_IT];
  const uint16_t* swh_ptr[W_IT];
  const uint16_t* swl_ptr[W_IT];
  int sw_inc[W_IT];
#pragma unroll
  for (int i = 0; i < W_IT; ++i) {
    if (PREC == 0) {
      const int n = n0 + r0 + AROWS * i;
      const bool ok = n < p.N;
      sw_ptr[i] = ok ? p.W + (long long)n * p.ldw + (long long)kt_begin * BK + kc * 4 : p.zeros;
      sw_inc[i] = ok ? BK : 0;
    } else {
      const int n = n0 + wr0 + WROWS * i;
      const bool ok = n < p.N && wr0 + WROWS * i < BN;
      const long long off = (long long)n * p.ldw + (long long)kt_begin * BK + wc * 8;
      swh_ptr[i] = ok ? p.Wh + off : reinterpret_cast<const uint16_t*>(p.zeros);
      swl_ptr[i] = ok ? p.Wl + off : reinterpret_cast<const uint16_t*>(p.zeros);
      sw_inc[i] = ok ? BK : 0;
    }
  }

  f32x4 ra[A_IT];
  f32x4 ra2[HAS_A2 ? A_IT : 1];                       // second addend of the staged pieces (finish_piece)
  f32x4 pre_s4 = {1.f, 1.f, 1.f, 1.f}, pre_t4 = {0.f, 0.f, 0.f, 0.f};
  unsigned pre_ok = 0;                                // bit i: piece i lies inside the image / K range
  if (SIMPLE && HAS_PRE) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) pre_ok |= (m0 + r0 + AROWS * i < p.M) ? (1u << i) : 0u;
  }
  f32x4 rw[PREC == 0 ? W_IT : 1];
  u32x4 rwh[PREC >= 1 ? W_IT : 1], rwl[PREC == 1 ? W_IT : 1];
  auto load_tile = [&](int kt) {
    // ---- weights
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      if (PREC == 0) {
        rw[i] = *reinterpret_cast<const f32x4*>(sw_ptr[i]);
        sw_ptr[i] += sw_inc[i];
      } else {
        rwh[i] = *reinterpret_cast<const u32x4*>(swh_ptr[i]);
        swh_ptr[i] += sw_inc[i];
        if (PREC == 1) {
          rwl[i] = *reinterpret_cast<const u32x4*>(swl_ptr[i]);
          swl_ptr[i] += sw_inc[i];
        }
      }
    }
    // ---- activations
    if (SIMPLE) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        ra[i] = *reinterpret_cast<const f32x4*>(sa_ptr[i]);
        sa_ptr[i] += sa_inc[i];
      }
      if (HAS_PRE) {
        pre_s4 = *reinterpret_cast<const f32x4*>(sps_ptr);
        pre_t4 = *reinterpret_cast<const f32x4*>(spt_ptr);
        sps_ptr += BK; spt_ptr += BK;
      }
      return;
    }
    const int kg = kt * BK + kc * 4;
    const bool kok = kg < p.K;
    const int tap = kg / p.Cin;
    const int ci = kg - tap * p.Cin;
    const int ty = tap / p.kw;
    const int tx = tap - ty * p.kw;
    const int dy = ty * p.dil_h, dx = tx * p.dil_w;
    const int tapoff = dy * p.Win + dx;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
      const bool ok = kok && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
      const long long pix = a_pix[i] + tapoff;
      const float* src = ok ? p.A + pix * p.lda + p.a_off + ci : p.zeros;
      ra[i] = *reinterpret_cast<const f32x4*>(src);
      if (HAS_A2) {
        const float* src2 = ok ? p.A2 + pix * p.lda2 + p.a2_off + ci : p.zeros;
        ra2[i] = *reinterpret_cast<const f32x4*>(src2);
      }
      if (HAS_PRE) pre_ok = ok ? (pre_ok | (1u << i)) : (pre_ok & ~(1u << i));
    }
    if (HAS_PRE) {
      // ci < Cin is guaranteed when kok; clamp keeps the (discarded) tail loads in range
      const int cc = kok ? ci : 0;
      pre_s4 = *reinterpret_cast<const f32x4*>(p.pre_scale + cc);
      pre_t4 = *reinterpret_cast<const f32x4*>(p.pre_shift + cc);
    }
  };
  // Second addend / BN-ReLU pre-activation of staged piece i: applied when the piece is WRITTEN to LDS, not
  // when it is requested -- done in load_tile, the first dependent VALU op waited for the global loads in front of
  // the running tile's MFMAs (one exposed L2 / HBM round trip per K-tile in every A2 / PRE layer).
  auto finish_piece = [&](int i) {
    if (HAS_A2) ra[i] += ra2[i];
    if (HAS_PRE) {
      f32x4 v = ra[i] * pre_s4 + pre_t4;
      const bool ok = (pre_ok >> i) & 1u;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = ok ? relu_f(v[q]) : 0.f;
      ra[i] = v;
    }
  };

  // LDS map.  PREC 0: [buf][A rows | W rows][36 floats].
  //           PREC 1: [buf][A_hi | A_lo | W_hi | W_lo] planes of [rows][40 halfs].
  constexpr int STAGE_FLOATS =
      PREC == 0 ? (BM + BN) * S : (PREC == 1 ? (BM + BN) * HS : (BM + BN) * HS / 2);   // floats per buffer
  auto store_tile = [&](int buf) {
    if (HAS_A2 || HAS_PRE) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) finish_piece(i);
    }
    if (PREC == 0) {
      float* As = lds + buf * STAGE_FLOATS;
      float* Ws = As + BM * S;
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        *reinterpret_cast<f32x4*>(&As[(r0 + AROWS * i) * S + kc * 4]) = ra[i];
#pragma unroll
      for (int i = 0; i < W_IT; ++i)
        *reinterpret_cast<f32x4*>(&Ws[(r0 + AROWS * i) * S + kc * 4]) = rw[i];
    } else {
      _Float16* base = reinterpret_cast<_Float16*>(lds + buf * STAGE_FLOATS);
      // PREC 1: [A_hi | A_lo | W_hi | W_lo]; PREC 2 (plain binary16 operands): [A_hi | W_hi]
      _Float16* Ah = base;
      _Float16* Al = Ah + BM * HS;
      _Float16* Wh = PREC == 1 ? Al + BM * HS : Ah + BM * HS;
      _Float16* Wl = Wh + BN * HS;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        f16x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const _Float16 h = (_Float16)ra[i][q];
          hi[q] = h;
          if (PREC == 1) lo[q] = (_Float16)(ra[i][q] - (float)h);
        }
        *reinterpret_cast<f16x4*>(&Ah[(r0 + AROWS * i) * HS + kc * 4]) = hi;
        if (PREC == 1) *reinterpret_cast<f16x4*>(&Al[(r0 + AROWS * i) * HS + kc * 4]) = lo;
      }
#pragma unroll
      for (int i = 0; i < W_IT; ++i) {
        if (BN >= WROWS || wr0 < BN) {
          *reinterpret_cast<u32x4*>(&Wh[(wr0 + WROWS * i) * HS + wc * 8]) = rwh[i];
          if (PREC == 1) *reinterpret_cast<u32x4*>(&Wl[(wr0 + WROWS * i) * HS + wc * 8]) = rwl[i];
        }
      }
    }
  };

  // ---------------- compute roles
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int im = 0; im < TM; ++im)
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[im][in][r] = 0.f;

  if (kt_begin < kt_end) {
    load_tile(kt_begin);
    store_tile(0);
  }
  __syncthreads();
  auto compute_tile = [&](int buf) {
    __builtin_amdgcn_s_setprio(1);
    if (PREC == 0) {
      const float* As = lds + buf * STAGE_FLOATS + (wm * TM * 32 + li) * S + lh * 4;
      const float* Ws = lds + buf * STAGE_FLOATS + BM * S + (wn * TN * 32 + li) * S + lh * 4;
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int im = 0; im < TM; ++im)
          a[im] = *reinterpret_cast<const f32x4*>(&As[im * 32 * S + g * 8]);
#pragma unroll
        for (int in = 0; in < TN; ++in)
          b[in] = *reinterpret_cast<const f32x4*>(&Ws[in * 32 * S + g * 8]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int im = 0; im < TM; ++im)
#pragma unroll
            for (int in = 0; in < TN; ++in)
              acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[in][s], a[im][s], acc[im][in], 0, 0, 0);
      }
    } else {
      // 32x32x16 f16 MFMA: lane (i = l & 31, h = l >> 5) holds k = 8h .. 8h+7 of row/col i
      const _Float16* base = reinterpret_cast<const _Float16*>(lds + buf * STAGE_FLOATS);
      const _Float16* Ah = base + (wm * TM * 32 + li) * HS + lh * 8;
      const _Float16* Al = Ah + BM * HS;
      const _Float16* Wh = base + (PREC == 1 ? 2 : 1) * BM * HS + (wn * TN * 32 + li) * HS + lh * 8;
      const _Float16* Wl = Wh + BN * HS;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int im = 0; im < TM; ++im) {
          ah[im] = *reinterpret_cast<const f16x8*>(&Ah[im * 32 * HS + ks * 16]);
          if (PREC == 1) al[im] = *reinterpret_cast<const f16x8*>(&Al[im * 32 * HS + ks * 16]);
        }
#pragma unroll
        for (int in = 0; in < TN; ++in) {
          bh[in] = *reinterpret_cast<const f16x8*>(&Wh[in * 32 * HS + ks * 16]);
          if (PREC == 1) bl[in] = *reinterpret_cast<const f16x8*>(&Wl[in * 32 * HS + ks * 16]);
        }
        if (PREC == 1) {
          // small cross terms first, the hi*hi term last
#pragma unroll
          for (int im = 0; im < TM; ++im)
#pragma unroll
            for (int in = 0; in < TN; ++in)
              acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[in], al[im], acc[im][in], 0, 0, 0);
#pragma unroll
          for (int im = 0; im < TM; ++im)
#pragma unroll
            for (int in = 0; in < TN; ++in)
              acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[in], ah[im], acc[im][in], 0, 0, 0);
        }
#pragma unroll
        for (int im = 0; im < TM; ++im)
#pragma unroll
          for (int in = 0; in < TN; ++in)
            acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[in], ah[im], acc[im][in], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  if constexpr (PREC == 0 && TM == 2 && TN == 2) {
    // fp32 schedule of the 128x128 tile.  A wave sits in MFMA issue for a whole k-group (16 x 64 cycles)
    // and the two workgroups of a CU run in lock step (equal tiles, started together), so anything placed
    // BETWEEN the K-tiles -- LDS stores of the next tile, the barrier, the first fragment reads -- idled the
    // matrix pipe (cycle stamps on the K = 1536 layer: ~1600 of every 9800 cycles per K-tile unused; each
    // wave's ~800 cycles of staging were NOT covered by its SIMD partner, whose barrier couples it to three
    // other SIMDs).  Here every such step is issued BEHIND one MFMA of the running tile (an fp32 MFMA
    // occupies the pipe for 64 cycles; sched_barrier keeps the filler in its shadow):
    //   g0: global loads of tile k+1, fragments of g1 | g1: fragments of g2 | g2: LDS stores of tile k+1,
    //   fragments of g3 | barrier | g3: fragments of group 0 of tile k+1
    // The barrier may sit before the last group because that group's fragments are already in registers:
    // behind it nobody reads the current buffer any more and the next one is complete.
    // Same-box A/B (tools/gemm_probe): K = N = 1536 layer 1952 -> 1857 us, N = K = 512 layer 275 -> 264 us.
    // What it does not move: one wavefront per SIMD alone already reaches the same 0.82 of the fp32 MFMA
    // peak inside the loop as two (forced by an LDS pad), two-tiles-ahead register prefetch changes nothing,
    // swapping the weight fragment's register banks changes nothing.
    f32x4 fa[2][TM], fb[2][TN];
    auto frag_piece = [&](int bufi, int g, int slot, int j) {      // j: 0,1 = A blocks, 2,3 = W blocks
      const float* As = lds + bufi * STAGE_FLOATS + (wm * TM * 32 + li) * S + lh * 4;
      const float* Ws = lds + bufi * STAGE_FLOATS + BM * S + (wn * TN * 32 + li) * S + lh * 4;
      if (j < 2) fa[slot][j] = *reinterpret_cast<const f32x4*>(&As[j * 32 * S + g * 8]);
      else fb[slot][j - 2] = *reinterpret_cast<const f32x4*>(&Ws[(j - 2) * 32 * S + g * 8]);
    };
    // 16 MFMAs of the k-group held in fragment slot `slot`; filler(i) is issued behind MFMA i
    auto mma = [&](int slot, auto&& filler) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int sq = 0; sq < 4; ++sq)
#pragma unroll
        for (int im = 0; im < TM; ++im)
#pragma unroll
          for (int in = 0; in < TN; ++in) {
            acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[slot][in][sq], fa[slot][im][sq], acc[im][in], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            filler(sq * 4 + im * 2 + in);
            __builtin_amdgcn_sched_barrier(0);
          }
      __builtin_amdgcn_s_setprio(0);
    };
    // staging pieces of the next K-tile: 0 .. W_IT-1 weights, W_IT .. W_IT+A_IT-1 activations
    constexpr int NP = W_IT + A_IT;
    static_assert(NP <= 8, "at most eight staging pieces per thread");
    auto load_piece = [&](int j) {
      if (j < W_IT) {
        rw[j] = *reinterpret_cast<const f32x4*>(sw_ptr[j]);
        sw_ptr[j] += sw_inc[j];
      } else if (j < NP) {
        ra[j - W_IT] = *reinterpret_cast<const f32x4*>(sa_ptr[j - W_IT]);
        sa_ptr[j - W_IT] += sa_inc[j - W_IT];
        if (HAS_PRE && j == W_IT) {
          pre_s4 = *reinterpret_cast<const f32x4*>(sps_ptr);
          pre_t4 = *reinterpret_cast<const f32x4*>(spt_ptr);
          sps_ptr += BK; spt_ptr += BK;
        }
      }
    };
    auto store_piece = [&](int bufi, int j) {
      float* As = lds + bufi * STAGE_FLOATS;
      float* Ws = As + BM * S;
      if (j < W_IT) *reinterpret_cast<f32x4*>(&Ws[(r0 + AROWS * j) * S + kc * 4]) = rw[j];
      else if (j < NP) {
        if (HAS_A2 || HAS_PRE) finish_piece(j - W_IT);
        *reinterpret_cast<f32x4*>(&As[(r0 + AROWS * (j - W_IT)) * S + kc * 4]) = ra[j - W_IT];
      }
    };
    int buf = 0;
    if (kt_begin < kt_end) {
#pragma unroll
      for (int j = 0; j < 4; ++j) frag_piece(0, 0, 0, j);
    }
    for (int kt = kt_begin; kt + 1 < kt_end; ++kt) {
      if (!SIMPLE) {                       // general convolution: tap decode once, then all rows
        load_tile(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      mma(0, [&](int i) {                  // g0: global loads of tile kt+1, then the fragments of g1
        if (SIMPLE && i < 8) load_piece(i);
        if (i >= 8 && i < 12) frag_piece(buf, 1, 1, i - 8);
      });
      mma(1, [&](int i) {                  // g1: fragments of g2
        if (i >= 8 && i < 12) frag_piece(buf, 2, 0, i - 8);
      });
      mma(0, [&](int i) {                  // g2: LDS stores of tile kt+1, fragments of g3
        if (i < 8) store_piece(buf ^ 1, i);
        if (i >= 8 && i < 12) frag_piece(buf, 3, 1, i - 8);
      });
      __syncthreads();
      mma(1, [&](int i) {                  // g3 (fragments already in registers): group 0 of tile kt+1
        if (i < 4) frag_piece(buf ^ 1, 0, 0, i);
      });
      buf ^= 1;
    }
    if (kt_begin < kt_end) {
      mma(0, [&](int i) { if (i >= 8 && i < 12) frag_piece(buf, 1, 1, i - 8); });
      mma(1, [&](int i) { if (i >= 8 && i < 12) frag_piece(buf, 2, 0, i - 8); });
      mma(0, [&](int i) { if (i >= 8 && i < 12) frag_piece(buf, 3, 1, i - 8); });
      mma(1, [](int) {});
    }
    __syncthreads();
  } else {
  // Branch-free steady state (the last K-tile is peeled) so that the scheduler may interleave the
  // conversion / LDS writes of tile k+1 with the MFMAs of tile k.
  int buf = 0;
  for (int kt = kt_begin; kt + 1 < kt_end; ++kt) {
    load_tile(kt + 1);
    compute_tile(buf);
    store_tile(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (kt_begin < kt_end) compute_tile(buf);
  __syncthreads();
  }

  gemm_epilogue<BM, BN, WM, WN>(p, acc, lds, m0, n0, threadIdx.x);
}

template <int BM, int BN, int WM, int WN, bool HAS_A2, bool HAS_PRE, bool SIMPLE, int PREC>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 8 ? 4 : 2))
void conv_gemm_kernel(const ConvGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  conv_gemm_body<BM, BN, WM, WN, HAS_A2, HAS_PRE, SIMPLE, PREC>(p, lds, blockIdx.x, gridDim.x, p.m_begin);
}

// Whole rounds of 128x128 tiles AND the 64x64 tiles of the remaining rows in ONE grid.  A layer whose tile
// count is not a multiple of the chip's 512 block slots used to run as two launches (128x128 rounds, then a
// 64x64 tail): here workgroups [0, n_big) take the big tiles and the rest the small ones; the dispatcher
// hands out workgroups in index order as slots free up.  (Gain: the kernel boundary only, ~4 % on the
// N = K = 512 layer -- the rounds of equal tiles end together, so the small tiles still start behind them.)
template <bool HAS_A2, bool HAS_PRE, bool SIMPLE, int PREC>
__global__ __launch_bounds__(256, 2)
void conv_gemm_dual_kernel(const ConvGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int bid = blockIdx.x;
  if (bid < p.n_big)
    conv_gemm_body<128, 128, 2, 2, HAS_A2, HAS_PRE, SIMPLE, PREC>(p, lds, bid, p.n_big, p.m_begin);
  else
    conv_gemm_body<64, 64, 2, 2, HAS_A2, HAS_PRE, SIMPLE, PREC>(p, lds, bid - p.n_big, (int)gridDim.x - p.n_big,
                                                                p.tail_begin);
}

// ---------------------------------------------------------------------------------------------
// f16 fast path (PREC 2, 1x1 layers with K % 64 == 0): K-tile of 64, single binary16 plane per
// operand, 16-B staging chunks of 8 halfs.  The 3-pass kernel above spends ~770 MFMA cycles per
// K-tile and wave; one pass leaves 256, far less than an L2 round trip, so this variant doubles the
// work per barrier and (AF32 = false) reads activations that the producing layer already stored as
// binary16 (ConvGemmParams::A16): no conversion, half the activation bytes.
constexpr int FBK = 64;
constexpr int FHS = FBK + 8;     // halfs per LDS row: 144 B = 36 dwords, the conflict-free stride of
                                 // the fp32 layout (ds_read_b128) with whole rows per store group

#ifdef WS_TRACE
// phase stamps (s_memtime) of wavefront 0 of workgroup 0: [iteration][6]
__device__ unsigned long long g_trace[64 * 8];
unsigned long long* trace_buffer_address() {
  unsigned long long* p = nullptr;
  (void)hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(g_trace));
  return p;
}
#define WS_STAMP(slot)                                                              \
  if (blockIdx.x == 8 && tid == 0 && kt < 63) g_trace[kt * 8 + (slot)] = __builtin_readcyclecounter();
#define WS_STAMP2(slot)                                                             \
  if (blockIdx.x == 8 && tid == 0 && kt < 31) g_trace[(32 + kt) * 8 + (slot)] = __builtin_readcyclecounter();
#define WS_MARK(slot)                                                                \
  if (blockIdx.x == 8 && threadIdx.x == 0) {                                          \
    g_trace[63 * 8 + (slot)] = __builtin_readcyclecounter();                          \
    if ((slot) == 0) g_trace[63 * 8 + 4] = __builtin_amdgcn_s_memrealtime();         \
    if ((slot) == 3) g_trace[63 * 8 + 5] = __builtin_amdgcn_s_memrealtime();         \
  }
#else
#define WS_STAMP(slot)
#define WS_STAMP2(slot)
#define WS_MARK(slot)
#endif

template <int BM, int BN>
constexpr size_t f16_lds_bytes() {
  const size_t stage = (size_t)(BM + BN) * FHS * 2;
  const size_t epi = (size_t)BM * (BN + 4) * 4;
  const size_t pool = (size_t)BM * 256 * 2;
  size_t m = 2 * stage > epi ? 2 * stage : epi;
  return m > pool ? m : pool;
}

template <int BM, int BN, int WM, int WN, bool AF32>
__global__ __launch_bounds__(64 * WM * WN, 2)
void gemm_f16_kernel(const ConvGemmParams p) {
  constexpr int NT = 64 * WM * WN;
  static_assert(NT == 256, "4 wavefronts");
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int SROWS = NT / 8;                  // rows per staging pass (8 chunks of 16 B per row)
  constexpr int A_IT = BM / SROWS, W_IT = BN / SROWS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int tiles_n = (p.N + BN - 1) / BN;
  int work;
  {
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    work = xcd * q + (xcd < r ? xcd : r) + local;
  }
  const int tile_m = work / tiles_n;
  const int tile_n = work - tile_m * tiles_n;
  const int m0 = p.m_begin + tile_m * BM, n0 = tile_n * BN;
  const int c8 = tid & 7, sr = tid >> 3;
  const int nk = p.K / FBK;

  const uint16_t* a16_ptr[A_IT];
  const float* a32_ptr[A_IT];
  int a_inc[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + sr + SROWS * i;
    const bool ok = m < p.M;
    if (AF32) a32_ptr[i] = ok ? p.A + (long long)m * p.lda + p.a_off + c8 * 8 : p.zeros;
    else a16_ptr[i] = ok ? p.A16 + (long long)m * p.lda16 + p.a_off + c8 * 8
                         : reinterpret_cast<const uint16_t*>(p.zeros);
    a_inc[i] = ok ? FBK : 0;
  }
  const uint16_t* w_ptr[W_IT];
  int w_inc[W_IT];
#pragma unroll
  for (int i = 0; i < W_IT; ++i) {
    const int n = n0 + sr + SROWS * i;
    const bool ok = n < p.N;
    w_ptr[i] = ok ? p.Wh + (long long)n * p.ldw + c8 * 8 : reinterpret_cast<const uint16_t*>(p.zeros);
    w_inc[i] = ok ? FBK : 0;
  }
  u32x4 ra[A_IT], rw[W_IT];
  f32x4 rf[AF32 ? A_IT : 1][2];
  auto load_tile = [&]() {
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      rw[i] = *reinterpret_cast<const u32x4*>(w_ptr[i]);
      w_ptr[i] += w_inc[i];
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      if (AF32) {
        rf[i][0] = *reinterpret_cast<const f32x4*>(a32_ptr[i]);
        rf[i][1] = *reinterpret_cast<const f32x4*>(a32_ptr[i] + 4);
        a32_ptr[i] += a_inc[i];
      } else {
        ra[i] = *reinterpret_cast<const u32x4*>(a16_ptr[i]);
        a16_ptr[i] += a_inc[i];
      }
    }
  };
  constexpr int STAGE_HALFS = (BM + BN) * FHS;
  auto store_tile = [&](int buf) {
    _Float16* As = reinterpret_cast<_Float16*>(lds) + buf * STAGE_HALFS;
    _Float16* Ws = As + BM * FHS;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      if (AF32) {
        f16x8 h;
#pragma unroll
        for (int q = 0; q < 4; ++q) { h[q] = (_Float16)rf[i][0][q]; h[4 + q] = (_Float16)rf[i][1][q]; }
        *reinterpret_cast<f16x8*>(&As[(sr + SROWS * i) * FHS + c8 * 8]) = h;
      } else {
        *reinterpret_cast<u32x4*>(&As[(sr + SROWS * i) * FHS + c8 * 8]) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i)
      *reinterpret_cast<u32x4*>(&Ws[(sr + SROWS * i) * FHS + c8 * 8]) = rw[i];
  };

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int im = 0; im < TM; ++im)
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[im][in][r] = 0.f;
  auto compute_tile = [&](int buf) {
    __builtin_amdgcn_s_setprio(1);
    const _Float16* As = reinterpret_cast<const _Float16*>(lds) + buf * STAGE_HALFS +
                         (wm * TM * 32 + li) * FHS + lh * 8;
    const _Float16* Ws = reinterpret_cast<const _Float16*>(lds) + buf * STAGE_HALFS + BM * FHS +
                         (wn * TN * 32 + li) * FHS + lh * 8;
#pragma unroll
    for (int ks = 0; ks < FBK / 16; ++ks) {
      f16x8 a[TM], b[TN];
#pragma unroll
      for (int im = 0; im < TM; ++im) a[im] = *reinterpret_cast<const f16x8*>(&As[im * 32 * FHS + ks * 16]);
#pragma unroll
      for (int in = 0; in < TN; ++in) b[in] = *reinterpret_cast<const f16x8*>(&Ws[in * 32 * FHS + ks * 16]);
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in)
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[in], a[im], acc[im][in], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  if (nk > 0) {
    load_tile();
    store_tile(0);
  }
  __syncthreads();
  int buf = 0;
  for (int kt = 0; kt + 1 < nk; ++kt) {
    WS_STAMP(0)
    load_tile();
    WS_STAMP(1)
    compute_tile(buf);
    WS_STAMP(2)
#ifdef WS_TRACE
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) only (lgkmcnt/expcnt untouched)
    WS_STAMP(3)
#endif
    store_tile(buf ^ 1);
    WS_STAMP(4)
    __syncthreads();
    WS_STAMP(5)
    buf ^= 1;
  }
  if (nk > 0) compute_tile(buf);
  __syncthreads();
  gemm_epilogue<BM, BN, WM, WN>(p, acc, lds, m0, n0, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// f16 path with LDS-DMA staging (`global_load_lds_dwordx4`): both operands are binary16 in HBM, so
// no register round trip, no conversion and -- the point -- no ds_write: s_memtime stamps on the
// register-staged kernel above show each K-tile spending ~900 of ~3300 cycles ISSUING its eight
// ds_write_b128 (the VGPR->LDS transfer path, MI355X_MICROARCH.md) and another ~900 waiting for
// loads behind them.  A DMA instruction fills 1 KiB of LDS lane-linearly, so the tile is stored
// unpadded ([rows][BKT] halfs) and the bank spread comes from an XOR swizzle of the 16-B chunk index
// applied on the SOURCE address and again on the fragment reads: chunk ^= (row >> s) & (CH - 1),
// s = 1 for 128-B rows (CH = 8), 2 for 64-B rows (CH = 4) -- conflict-free for every ds_read_b128
// lane group {0-3,12-15,20-27}, ...  NSTAGE LDS stages keep NSTAGE-1 K-tiles in flight across the
// per-tile barrier (counted vmcnt, raw s_barrier: __syncthreads would drain the queue).
template <int BM, int BN, int BKT, int NSTAGE, int NW>
constexpr size_t f16_dma_lds_bytes() {
  const size_t stages = (size_t)NSTAGE * (BM + BN) * BKT * 2;
  // 256x256 tiles finish as four 128x128 quadrants, two at a time (two transpose regions)
  const size_t epi = BM == 256 ? 2 * (size_t)128 * (128 + 4) * 4 : (size_t)BM * (BN + 4) * 4;
  const size_t pool = BM == 256 ? 0 : (size_t)BM * (64 * NW) * 2;
  size_t m = stages > epi ? stages : epi;
  return m > pool ? m : pool;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  // vmcnt(N) then barrier; "memory" keeps LDS accesses on their side of it
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}
// one 1-KiB LDS-DMA piece: lane l's 16 bytes at g land at lds_base + 16 l (lds_base wave-uniform)
__device__ __forceinline__ void dma_16B(const void* g, void* lds_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
#endif
}

// Wavefront grid: 2 x 2 for the 128x128 / 64x64 tiles (two workgroups per CU), 2 x 4 wavefronts of
// 128x64 outputs each for the 256x256 tile (512 threads, one workgroup per CU, 128 accumulator VGPRs
// per lane).  What limits this loop is the global -> LDS rate (~22 B/clk/CU measured, by DMA or
// through registers alike), so the 256x256 tile, which moves half the bytes per flop and reads 25 %
// fewer fragment bytes per MFMA, is the one that pays.
// CONV = true: general convolution (taps, stride, dilation, zero padding) on binary16 channels-last
// activations.  A 16-B chunk is 8 consecutive channels of one filter tap (Cin % 8 == 0), so each lane
// of a DMA piece computes its own source address per K-tile -- tap offset + bounds predicate, the
// zero page for padding / rows beyond M / k beyond K -- and the rest of the pipeline is unchanged.
template <int BM, int BN, int BKT, int NSTAGE, int NW, int WM_, bool CONV>
__global__ __launch_bounds__(64 * NW, (BN <= 32 ? 4 : 2))
void gemm_f16_dma_kernel(const ConvGemmParams p) {
  constexpr int WM = WM_, WN = NW / WM_;
  constexpr int NWAVES = NW;
  static_assert((NW == 4 || NW == 8) && (BM != 256 || NW == 8) && WM * WN == NW, "wavefront grid");
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int CH = BKT / 8;                    // 16-B chunks per row
  constexpr int RPD = 64 / CH;                   // rows covered by one 1-KiB DMA instruction
  constexpr int SWS = CH == 8 ? 1 : 2;           // swizzle key = (row >> SWS) & (CH - 1)
  constexpr int A_BYTES = BM * BKT * 2, W_BYTES = BN * BKT * 2, STAGE_BYTES = A_BYTES + W_BYTES;
  // 1-KiB pieces per wavefront and K-tile.  A narrow weight tile has fewer pieces than wavefronts:
  // the surplus wavefronts repeat a piece (same bytes to the same place) so that every wavefront
  // counts the same number of loads per tile (the vmcnt arithmetic below relies on it).
  constexpr int W_PIECES = W_BYTES / 1024;
  constexpr int A_DMA = A_BYTES / 1024 / NWAVES, W_DMA = W_PIECES >= NWAVES ? W_PIECES / NWAVES : 1;
  constexpr int LPT = A_DMA + W_DMA;
  constexpr int D = NSTAGE - 1;                  // K-tiles in flight
  static_assert(A_DMA >= 1 && W_PIECES >= 1 && (CH == 8 || CH == 4), "tile / K-tile combination");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  WS_EMARK(6)

  const int tid = threadIdx.x;
  const int tiles_n = (p.N + BN - 1) / BN;
  int work;
  {
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    work = xcd * q + (xcd < r ? xcd : r) + local;
  }
  const int tile_m = work / tiles_n;
  const int tile_n = work - tile_m * tiles_n;
  const int m0 = p.m_begin + tile_m * BM, n0 = tile_n * BN;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = CONV ? (p.K + BKT - 1) / BKT : p.K / BKT;

  // per-lane DMA sources: slot (row rr, physical chunk pc) of a 1-KiB piece <- logical chunk pc ^ key(row).
  // Rows beyond M / N are clamped to the last valid row (their products only reach outputs that are
  // never stored), so every piece advances by the same BKT halfs per K-tile: the persistent state is
  // one 32-bit element offset per piece plus a scalar K offset.
  int a_off32[A_DMA], w_off32[W_DMA];
  int a_pix[CONV ? A_DMA : 1], a_iy[CONV ? A_DMA : 1], a_ix[CONV ? A_DMA : 1], a_kc[CONV ? A_DMA : 1];
  {
    const int rr = lane / CH, pc = lane % CH;
#pragma unroll
    for (int j = 0; j < A_DMA; ++j) {
      const int row = (wave * A_DMA + j) * RPD + rr;
      const int c = pc ^ ((row >> SWS) & (CH - 1));
      if (CONV) {
        const int HWo = p.Hout * p.Wout;
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int img = mm / HWo, rem = mm - img * HWo;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        const int iy0 = oy * p.stride_h - p.pad_h, ix0 = ox * p.stride_w - p.pad_w;
        a_pix[j] = (img * p.Hin + iy0) * p.Win + ix0;
        a_iy[j] = ok ? iy0 : -(1 << 28);          // forces the bounds predicate false for rows >= M
        a_ix[j] = ix0;
        a_kc[j] = c * 8;
        a_off32[j] = 0;
      } else {
        const int m = m0 + row < p.M ? m0 + row : p.M - 1;
        a_off32[j] = m * p.lda16 + p.a_off + c * 8;
      }
    }
#pragma unroll
    for (int j = 0; j < W_DMA; ++j) {
      const int row = ((wave * W_DMA + j) % W_PIECES) * RPD + rr;
      const int c = pc ^ ((row >> SWS) & (CH - 1));
      const int n = n0 + row < p.N ? n0 + row : p.N - 1;
      w_off32[j] = n * p.ldw + c * 8;
    }
  }
  int k_off = 0;                                 // halfs; wave-uniform
  auto issue = [&](int stage) {
    char* base = ldsb + stage * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < A_DMA; ++j) {
      if (CONV) {
        const int kg = k_off + a_kc[j];
        const int tap = kg / p.Cin, ci = kg - tap * p.Cin;
        const int ty = tap / p.kw, tx = tap - ty * p.kw;
        const int dy = ty * p.dil_h, dx = tx * p.dil_w;
        const int iy = a_iy[j] + dy, ix = a_ix[j] + dx;
        const bool ok = kg < p.K && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
        const uint16_t* src = ok ? p.A16 + ((long long)(a_pix[j] + dy * p.Win + dx) * p.lda16 + p.a_off + ci)
                                 : reinterpret_cast<const uint16_t*>(p.zeros);
        dma_16B(src, base + (wave * A_DMA + j) * 1024);
      } else {
        dma_16B(p.A16 + (unsigned)(a_off32[j] + k_off), base + (wave * A_DMA + j) * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < W_DMA; ++j)
      dma_16B(p.Wh + (unsigned)(w_off32[j] + k_off), base + A_BYTES + ((wave * W_DMA + j) % W_PIECES) * 1024);
    k_off += BKT;
  };

  const int wm = wave / WN, wn = wave - wm * WN;
  const int li = lane & 31, lh = lane >> 5;
  const int sw = (li >> SWS) & (CH - 1);
  int koff[BKT / 16];
#pragma unroll
  for (int ks = 0; ks < BKT / 16; ++ks) koff[ks] = (((ks << 1) | lh) ^ sw) << 3;   // halfs
  f32x16 acc[TM][TN];
#pragma unroll
  for (int im = 0; im < TM; ++im)
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[im][in][r] = 0.f;
  auto compute_tile = [&](int stage) {
    const _Float16* As = reinterpret_cast<const _Float16*>(ldsb + stage * STAGE_BYTES) +
                         (wm * TM * 32 + li) * BKT;
    const _Float16* Ws = reinterpret_cast<const _Float16*>(ldsb + stage * STAGE_BYTES + A_BYTES) +
                         (wn * TN * 32 + li) * BKT;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ++ks) {
      f16x8 a[TM], b[TN];
#pragma unroll
      for (int im = 0; im < TM; ++im) a[im] = *reinterpret_cast<const f16x8*>(&As[im * 32 * BKT + koff[ks]]);
#pragma unroll
      for (int in = 0; in < TN; ++in) b[in] = *reinterpret_cast<const f16x8*>(&Ws[in * 32 * BKT + koff[ks]]);
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in)
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[in], a[im], acc[im][in], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // prologue: D tiles in flight, tile 0 landed
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nk) issue(s);
  if (nk >= D) wait_vmcnt_barrier<LPT*(D - 1)>();
  else wait_vmcnt_barrier<0>();
  int st_c = 0, st_i = D % NSTAGE;              // stage computed on / stage the next issue fills
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + D < nk;
    WS_STAMP(0)
    if (more) issue(st_i);
    WS_STAMP(1)
    compute_tile(st_c);
    WS_STAMP(2)
    // tile kt+1 must have landed in every wavefront before anyone reads it; tiles kt+2.. stay in flight
    if (more) wait_vmcnt_barrier<LPT*(D - 1)>();
    else wait_vmcnt_barrier<0>();
    WS_STAMP(3)
    st_c = st_c + 1 == NSTAGE ? 0 : st_c + 1;
    st_i = st_i + 1 == NSTAGE ? 0 : st_i + 1;
  }
  if constexpr (BM != 256) {
    gemm_epilogue<BM, BN, WM, WN>(p, acc, lds, m0, n0, threadIdx.x);
    WS_EMARK(7)
  } else {
    // four 128x128 quadrants, each owned by two wavefronts (128x64 each, a 1 x 2 grid); the two
    // quadrants of a row half run together on separate transpose regions
    const int qn = wn >> 1;
    const int tid_q = ((wn & 1) << 6) | lane;
    float* region = lds + qn * (128 * (128 + 4));
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
      gemm_epilogue<128, 128, 1, 2, false>(p, acc, region, m0 + ph * 128, n0 + qn * 128, tid_q, wm == ph);
      __syncthreads();
    }
  }
}

template <int BM, int BN, int BKT, int NSTAGE, int NW = (BM == 256 ? 8 : 4), int WM_ = 2, bool CONV = false>
static hipError_t launch_f16_dma(const ConvGemmParams& p, hipStream_t stream) {
  constexpr size_t lds_bytes = f16_dma_lds_bytes<BM, BN, BKT, NSTAGE, NW>();
  static size_t lds_granted[WS_MAX_DEVICES] = {};
  auto kern = gemm_f16_dma_kernel<BM, BN, BKT, NSTAGE, NW, WM_, CONV>;
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_granted);
    if (e != hipSuccess) return e;
  }
  const int tiles_m = (p.M - p.m_begin + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  if (tiles_m <= 0) return hipSuccess;
  WS_DLOG(p, "gemm_f16_dma_kernel<%d,%d,%d,%d%s>", BM, BN, BKT, NSTAGE, CONV ? ",conv" : "");
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(64 * NW), lds_bytes, stream, p);
  return hipGetLastError();
}

// ---- 256x256 tile, phase-staggered pipeline (plain GEMM on binary16 activations) -----------------------
// Eight wavefronts (2 x 4, 128x64 outputs each) in two groups that run one barrier interval apart:
// while one group issues its 8 (or 16) MFMAs of a phase the other one does the phase's LDS fragment
// reads, so a SIMD's two resident wavefronts alternate between the matrix pipe and
// the memory pipes instead of colliding in them.  A K-tile of 64 is two phases (one 64x64 half of the
// wavefront's outputs each: measured, a barrier interval costs ~110 cycles on top of its MFMAs, so 16
// MFMAs per phase beat 8); its operands are staged as four 16-KiB half-tiles (A0 = row sub-tile 0 of both wave
// rows, B0 / B1 = column sub-tiles of all four wave columns, A1), each read in exactly one phase
// (A0 + B0 + B1 in phase 1, A1 in phase 2 of its tile), so a slot is refilled one phase after that read
// and the DMA pieces are issued inside the MFMA runs (their ~100-cycle issue cost hides there):
//   phase 1 of tile kt: vmcnt(6) [retires A1(kt)]       | 16 MFMA + stage A1(kt+1)        -> other buffer
//   phase 2 of tile kt: vmcnt(2) [retires A0/B0/B1(kt+1)] | 16 MFMA + stage A0, B0, B1(kt+2) -> this buffer
// i.e. every piece is in flight for ~5 barrier intervals (~3000 cycles) before its first reader, which
// covers the ~1500-cycle LDS-DMA round trip that the 2-stage loop above waits for.  Reads happen one
// phase after the wait + barrier that retires their data (MI355X guide: nothing else orders LDS-DMA).
constexpr int P8_CONST_OFF = 2 * 128 * (128 + 4) * 4;      // behind the stage buffers / transpose regions
constexpr size_t f16_p8_lds_bytes() {
  const size_t stages = 2 * 4 * 16384, epi = 2 * (size_t)128 * (128 + 4) * 4;
  return (stages > epi ? stages : epi) + 3 * 256 * 4;      // + per-channel constants of the binary16 epilogue
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void raw_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// Binary16 one-phase epilogue of the 256x256 kernel for layers that only leave a binary16 tensor (D16, no
// D / D2 / per-row operands, act none or ReLU).  The two-phase fp32 transpose above costs a K = 512 tile more
// than its K loop; here bias / ReLU / BN are applied in registers on the C^T blocks (a lane owns 4
// consecutive channels of one pixel, so the per-channel constants are one f32x4 each), the values are
// converted and the WHOLE tile is transposed as halfs ([256][264]: 135 KB), then every thread stores
// 16-byte runs of finished rows.  Column sums (ConvGemmParams::colsum) are taken from the stored binary16
// values -- per (64-row group, image part, column), 16 row phases folded through LDS in a fixed order.
__device__ __forceinline__ void p8_epilogue_f16(const ConvGemmParams& p, f32x16 (&acc)[4][2], char* ldsb,
                                                int m0, int n0, int tid) {
  constexpr int YS = 264;                         // halfs per LDS row (528 B: 16-byte aligned rows)
  _Float16* Y = reinterpret_cast<_Float16*>(ldsb);
  const float* kc = reinterpret_cast<const float*>(ldsb + P8_CONST_OFF);   // [bias | scale | shift][256]
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int in = 0; in < 2; ++in)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = wc * 64 + in * 32 + 8 * g + 4 * lh;
      // per-channel constants of this tile's 256 columns: put into LDS by the kernel before its K loop
      const f32x4 bias = *reinterpret_cast<const f32x4*>(&kc[col]);
      const f32x4 ps = *reinterpret_cast<const f32x4*>(&kc[256 + col]);
      const f32x4 pb = *reinterpret_cast<const f32x4*>(&kc[512 + col]);
#pragma unroll
      for (int im = 0; im < 4; ++im) {
        f32x4 v = (f32x4){acc[im][in][4 * g], acc[im][in][4 * g + 1], acc[im][in][4 * g + 2],
                          acc[im][in][4 * g + 3]} + bias;
        if (p.act == ACT_RELU) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = relu_f(v[q]);
        }
        if (p.post_scale) v = v * ps + pb;
        f16x4 hv;
#pragma unroll
        for (int q = 0; q < 4; ++q) hv[q] = (_Float16)v[q];
        const int row = wr * 128 + im * 32 + li;
        *reinterpret_cast<f16x4*>(&Y[row * YS + col]) = hv;
      }
    }
  __syncthreads();
  const int c8 = tid & 31, rr = tid >> 5;         // 8-channel group, row phase (16 phases)
  const int HW = p.Hout * p.Wout;
  float cs[4][2][8];
  if (p.colsum) {
#pragma unroll
    for (int hf = 0; hf < 4; ++hf)
#pragma unroll
      for (int wh = 0; wh < 2; ++wh)
#pragma unroll
        for (int q = 0; q < 8; ++q) cs[hf][wh][q] = 0.f;
  }
  uint16_t* dst = p.D16 + p.d_off + n0 + c8 * 8;
#pragma unroll
  for (int hf = 0; hf < 4; ++hf) {
    const int mh = m0 + hf * 64;
    const int imgA = mh / HW;
    const int rb = (imgA + 1) * HW - mh;          // rows >= rb of this 64-row group belong to the next image
    // ragged batch (H = 1 layers only): rows beyond the utterance's own length are stored as zeros
    int lenA = 0x7fffffff, lenB = 0x7fffffff;
    if (p.row_len) {
      const int last = (p.M - 1) / HW;
      lenA = p.row_len[imgA < last ? imgA : last];
      lenB = p.row_len[imgA + 1 < last ? imgA + 1 : last];
    }
    const int tA0 = mh - imgA * HW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = rr + 16 * i;
      const int row = hf * 64 + rl;
      const int m = m0 + row;
      if (m < p.M) {
        f16x8 hv = *reinterpret_cast<const f16x8*>(&Y[row * YS + c8 * 8]);
        if (rl < rb ? tA0 + rl >= lenA : rl - rb >= lenB) {
#pragma unroll
          for (int q = 0; q < 8; ++q) hv[q] = (_Float16)0.f;
        }
        *reinterpret_cast<f16x8*>(dst + (long long)m * p.ldd16) = hv;
        if (p.colsum) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (rl < rb) cs[hf][0][q] += (float)hv[q]; else cs[hf][1][q] += (float)hv[q];
          }
        }
      }
    }
  }
  if (p.colsum) {
    __syncthreads();                              // everyone is done reading Y
    float* red = reinterpret_cast<float*>(ldsb);  // [hw = hf*2 + which (8)][rr (16)][256]
#pragma unroll
    for (int hf = 0; hf < 4; ++hf)
#pragma unroll
      for (int wh = 0; wh < 2; ++wh) {
        float* r = &red[((hf * 2 + wh) * 16 + rr) * 256 + c8 * 8];
        *reinterpret_cast<f32x4*>(r) = (f32x4){cs[hf][wh][0], cs[hf][wh][1], cs[hf][wh][2], cs[hf][wh][3]};
        *reinterpret_cast<f32x4*>(r + 4) = (f32x4){cs[hf][wh][4], cs[hf][wh][5], cs[hf][wh][6], cs[hf][wh][7]};
      }
    __syncthreads();
    for (int o = tid; o < 8 * 256; o += 512) {
      const int hw = o >> 8, col = o & 255;
      float sacc = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) sacc += red[(hw * 16 + q) * 256 + col];
      const int hf = hw >> 1;
      if (m0 + hf * 64 < p.M)
        p.colsum[((long long)((m0 + hf * 64) / 64) * 2 + (hw & 1)) * p.N + n0 + col] = sacc;
    }
  }
}

// CONV = true: convolution form (binary16 channels-last maps, Cin % 64 == 0 so that a K-tile lies inside one
// filter tap: the tap decode is wave-uniform, each lane only adds the tap offset to its pixel and checks
// the bounds; padding / rows beyond M read the zero page) -- ResNet's 256-channel 3x3 layers.
template <bool CONV>
__global__ __launch_bounds__(512, 1)
void gemm_f16_p8_kernel(const ConvGemmParams p) {
  constexpr int HT = 16384, BUF = 4 * HT;        // half-tile, buffer (A0 | B0 | B1 | A1)
  constexpr int SLOT_A[2] = {0, 3 * HT}, SLOT_B[2] = {HT, 2 * HT};
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x;
  const int tiles_n = (p.N + 255) / 256;
  int work;
  {
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    work = xcd * q + (xcd < r ? xcd : r) + local;
  }
  const int tile_m = work / tiles_n;
  const int tile_n = work - tile_m * tiles_n;
  const int m0 = p.m_begin + tile_m * 256, n0 = tile_n * 256;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nk = p.K / 64;
  if (!CONV && p.epi16 && tid < 192) {
    // binary16 epilogue: bias / BN scale / BN shift of this tile's columns -> LDS (read after many barriers)
    const int which = tid >> 6, c = (tid & 63) * 4;
    f32x4 v = which == 1 ? (f32x4){1.f, 1.f, 1.f, 1.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* src = which == 0 ? p.bias : (which == 1 ? p.post_scale : p.post_shift);
    if (src) v = *reinterpret_cast<const f32x4*>(src + n0 + c);
    *reinterpret_cast<f32x4*>(ldsb + P8_CONST_OFF + (which * 256 + c) * 4) = v;
  }

  // DMA sources of this lane: half-tile h, piece 2 wave + j -> local rows 8 (2 wave + j) + lane / 8
  int a_off32[2][2], w_off32[2][2];
  int a_pix[CONV ? 2 : 1][2], a_iy[CONV ? 2 : 1][2], a_ix[CONV ? 2 : 1][2];
  {
    const int rr = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lr = (wave * 2 + j) * 8 + rr;
        const int c = pc ^ ((lr >> 1) & 7);
        const int trow = (lr >> 6) * 128 + h * 64 + (lr & 63);
        if (CONV) {
          const int HWo = p.Hout * p.Wout;
          const int m = m0 + trow;
          const bool ok = m < p.M;
          const int mm = ok ? m : 0;
          const int img = mm / HWo, rem = mm - img * HWo;
          const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
          const int iy0 = oy * p.stride_h - p.pad_h, ix0 = ox * p.stride_w - p.pad_w;
          a_pix[CONV ? h : 0][j] = (img * p.Hin + iy0) * p.Win + ix0;
          a_iy[CONV ? h : 0][j] = ok ? iy0 : -(1 << 28);     // rows >= M: bounds predicate always false
          a_ix[CONV ? h : 0][j] = ix0;
          a_off32[h][j] = p.a_off + c * 8;                   // channel offset inside the pixel
        } else {
          const int m = m0 + trow < p.M ? m0 + trow : p.M - 1;
          a_off32[h][j] = m * p.lda16 + p.a_off + c * 8;
        }
        const int tcol = (lr >> 5) * 64 + h * 32 + (lr & 31);
        const int n = n0 + tcol < p.N ? n0 + tcol : p.N - 1;
        w_off32[h][j] = n * p.ldw + c * 8;
      }
  }
  // one A piece of K-tile k_off (halfs): j-th piece of half-tile h
  auto a_src = [&](int h, int j, int k_off) -> const uint16_t* {
    if (CONV) {
      const int tap = k_off / p.Cin, ci0 = k_off - tap * p.Cin;          // wave-uniform (Cin % 64 == 0)
      const int ty = tap / p.kw, tx = tap - ty * p.kw;
      const int dy = ty * p.dil_h, dx = tx * p.dil_w;
      const int iy = a_iy[CONV ? h : 0][j] + dy, ix = a_ix[CONV ? h : 0][j] + dx;
      const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
      return ok ? p.A16 + ((long long)(a_pix[CONV ? h : 0][j] + dy * p.Win + dx) * p.lda16 + ci0 + a_off32[h][j])
                : reinterpret_cast<const uint16_t*>(p.zeros);
    }
    return p.A16 + (unsigned)(a_off32[h][j] + k_off);
  };
  auto stage_a = [&](int h, int buf, int k_off) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      dma_16B(a_src(h, j, k_off), ldsb + buf * BUF + SLOT_A[h] + (wave * 2 + j) * 1024);
  };
  auto stage_b = [&](int h, int buf, int k_off) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      dma_16B(p.Wh + (unsigned)(w_off32[h][j] + k_off), ldsb + buf * BUF + SLOT_B[h] + (wave * 2 + j) * 1024);
  };

  const int li = lane & 31, lh = lane >> 5;
  const int sw = (li >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = (((ks << 1) | lh) ^ sw) << 4;   // bytes
  const int a_row = (wr * 64 + li) * 128, b_row = (wc * 32 + li) * 128;     // bytes inside a half-tile
  f32x16 acc[4][2];
#pragma unroll
  for (int im = 0; im < 4; ++im)
#pragma unroll
    for (int in = 0; in < 2; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[im][in][r] = 0.f;
  f16x8 a[2][4], b0[4], b1[4];
  auto read_a = [&](const char* base, int h) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        a[t][ks] = *reinterpret_cast<const f16x8*>(base + SLOT_A[h] + a_row + t * 32 * 128 + koff[ks]);
  };
  auto read_b = [&](const char* base, int h, f16x8* b) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      b[ks] = *reinterpret_cast<const f16x8*>(base + SLOT_B[h] + b_row + koff[ks]);
  };
  // 16 MFMAs of one 64x64 half of the wavefront's outputs (row sub-tile mh x both column sub-tiles) as 8
  // pairs.  `piece(i)` issues at most one DMA piece after pair i (spread out, each ~60-100-cycle issue
  // hides under matrix work).  The phase-closing barrier sits BEFORE the last pair: it orders LDS traffic
  // only, and releasing the other group while this one still has two MFMAs to issue keeps the matrix
  // pipe fed across the hand-over (measured: a hand-over bubble of ~110 cycles per interval otherwise).
  auto mma = [&](int mh, const f16x8* bf, const f16x8* bs, int inf, int ins, auto&& piece) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f16x8* b = i < 4 ? bf : bs;
      const int in = i < 4 ? inf : ins, ks = i & 3;
      if (i == 7) raw_barrier();
#pragma unroll
      for (int t = 0; t < 2; ++t)
        acc[2 * mh + t][in] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ks], a[t][ks], acc[2 * mh + t][in], 0, 0, 0);
      if (i < 7) {
        __builtin_amdgcn_sched_barrier(0);
        piece(i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto piece_a = [&](int h, int j, int buf, int k_off) {
    dma_16B(a_src(h, j, k_off), ldsb + buf * BUF + SLOT_A[h] + (wave * 2 + j) * 1024);
  };
  auto piece_b = [&](int h, int j, int buf, int k_off) {
    dma_16B(p.Wh + (unsigned)(w_off32[h][j] + k_off), ldsb + buf * BUF + SLOT_B[h] + (wave * 2 + j) * 1024);
  };

  // prologue: all of tile 0 and A0 / B0 / B1 of tile 1
  WS_MARK(0)
  stage_a(0, 0, 0); stage_b(0, 0, 0); stage_b(1, 0, 0); stage_a(1, 0, 0);
  if (nk > 1) { stage_a(0, 1, 64); stage_b(0, 1, 64); stage_b(1, 1, 64); wait_vmcnt<6>(); }
  else wait_vmcnt<0>();
  raw_barrier();
  WS_MARK(1)
  if (wr == 1) raw_barrier();                    // group 1 runs one barrier interval behind group 0
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const char* base = ldsb + buf * BUF;
    const bool n1 = kt + 1 < nk, n2 = kt + 2 < nk;
    const int k1 = (kt + 1) * 64, k2 = (kt + 2) * 64;
    // phase 1: row sub-tile 0 x (B0, B1).  Its wait leaves only the six pieces of A0 / B0 / B1(kt+1) in
    // flight, i.e. retires A1(kt), which phase 2 reads.
    WS_STAMP2(0)
    read_a(base, 0);
    read_b(base, 0, b0);
    read_b(base, 1, b1);
    if (n1) wait_vmcnt<6>();
    else wait_vmcnt<0>();
    raw_barrier();
    WS_STAMP2(1)
    mma(0, b0, b1, 0, 1, [&](int i) {
      if (n1 && (i == 1 || i == 4)) piece_a(1, i == 4, buf ^ 1, k1);
    });
    // phase 2: row sub-tile 1 x (B1, B0).  Its wait leaves only A1(kt+1)'s two pieces in flight:
    // A0 / B0 / B1 of tile kt+1 have landed for phase 1 of the next tile.
    WS_STAMP2(2)
    read_a(base, 1);
    if (n1) wait_vmcnt<2>();
    else wait_vmcnt<0>();
    raw_barrier();
    WS_STAMP2(3)
    mma(1, b1, b0, 1, 0, [&](int i) {
      if (n2) {
        if (i == 0) piece_a(0, 0, buf, k2);
        if (i == 1) piece_a(0, 1, buf, k2);
        if (i == 2) piece_b(0, 0, buf, k2);
        if (i == 3) piece_b(0, 1, buf, k2);
        if (i == 4) piece_b(1, 0, buf, k2);
        if (i == 5) piece_b(1, 1, buf, k2);
      }
    });
    WS_STAMP2(4)
  }
  if (wr == 0) raw_barrier();
  __syncthreads();
  WS_MARK(2)
  if (!CONV && p.epi16) {
    p8_epilogue_f16(p, acc, ldsb, m0, n0, tid);
    WS_MARK(3)
    return;
  }
  // four 128x128 quadrants, each owned by two wavefronts (a 1 x 2 grid of 128x64); the two quadrants
  // of a row half run together on separate transpose regions
  const int qn = wc >> 1;
  const int tid_q = ((wc & 1) << 6) | lane;
  float* region = lds + qn * (128 * (128 + 4));
#pragma unroll 1
  for (int ph = 0; ph < 2; ++ph) {
    gemm_epilogue<128, 128, 1, 2, false, 256, CONV>(p, acc, region, m0 + ph * 128, n0 + qn * 128, tid_q,
                                                    wr == ph, (wr << 7) | tid_q);
    __syncthreads();
  }
  WS_MARK(3)
}

template <bool CONV = false>
static hipError_t launch_f16_p8(const ConvGemmParams& p, hipStream_t stream) {
  constexpr size_t lds_bytes = f16_p8_lds_bytes();
  static size_t lds_granted[WS_MAX_DEVICES] = {};
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_f16_p8_kernel<CONV>), lds_bytes, lds_granted);
    if (e != hipSuccess) return e;
  }
  const int tiles_m = (p.M - p.m_begin + 255) / 256, tiles_n = (p.N + 255) / 256;
  if (tiles_m <= 0) return hipSuccess;
  WS_DLOG(p, "gemm_f16_p8_kernel<256,256%s>%s", CONV ? ",conv" : "", p.epi16 ? " epi16" : "");
  hipLaunchKernelGGL(gemm_f16_p8_kernel<CONV>, dim3(tiles_m * tiles_n), dim3(512), lds_bytes, stream, p);
  return hipGetLastError();
}

template <int BM, int BN, bool AF32>
static hipError_t launch_f16_fast(const ConvGemmParams& p, hipStream_t stream) {
  constexpr size_t lds_bytes = f16_lds_bytes<BM, BN>();
  static size_t lds_granted[WS_MAX_DEVICES] = {};
  auto kern = gemm_f16_kernel<BM, BN, 2, 2, AF32>;
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_granted);
    if (e != hipSuccess) return e;
  }
  const int tiles_m = (p.M - p.m_begin + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  if (tiles_m <= 0) return hipSuccess;
  WS_DLOG(p, "gemm_f16_kernel<%d,%d,%s>", BM, BN, AF32 ? "A=f32" : "A=f16");
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds_bytes, stream, p);
  return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, bool HAS_A2, bool HAS_PRE, bool SIMPLE, int PREC>
static hipError_t launch_one(const ConvGemmParams& p, hipStream_t stream) {
  // staging buffers / epilogue transpose tile, and -- only for the fused-pooling epilogue -- its
  // [NH*2][RPP][4][BN] reduction array (= BM * threads / 2 floats), which exceeds the former for the
  // single-plane f16 back-end on small tiles
  constexpr size_t base_bytes = tile_lds_bytes<BM, BN, PREC>();
  constexpr size_t pool_bytes = (size_t)BM * (64 * WM * WN) * 2;
  constexpr size_t max_bytes = base_bytes > pool_bytes ? base_bytes : pool_bytes;
  const size_t lds_bytes = p.pool_partial ? max_bytes : base_bytes;
  static size_t lds_granted[WS_MAX_DEVICES] = {};
  auto kern = conv_gemm_kernel<BM, BN, WM, WN, HAS_A2, HAS_PRE, SIMPLE, PREC>;
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), max_bytes, lds_granted);
    if (e != hipSuccess) return e;
  }
  const int tiles_m = (p.M - p.m_begin + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  if (tiles_m <= 0) return hipSuccess;
  dim3 grid(tiles_m * tiles_n, p.splitk > 1 ? p.splitk : 1, 1);
  WS_DLOG(p, "conv_gemm_kernel<%d,%d,%s>", BM, BN,
          SIMPLE ? (HAS_PRE ? "pre 1x1" : "1x1") : (HAS_A2 ? "A2" : (HAS_PRE ? "pre" : "conv")));
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds_bytes, stream, p);
  return hipGetLastError();
}

// one grid: 128x128 tiles over rows [m_begin, tail_begin), 64x64 tiles over [tail_begin, M)
template <bool HAS_A2, bool HAS_PRE, bool SIMPLE, int PREC>
static hipError_t launch_dual(ConvGemmParams p, int tail_begin, hipStream_t stream) {
  constexpr size_t big = tile_lds_bytes<128, 128, PREC>(), pool = (size_t)128 * 256 * 2;
  constexpr size_t max_bytes = big > pool ? big : pool;
  const size_t lds_bytes = p.pool_partial ? max_bytes : big;
  static size_t lds_granted[WS_MAX_DEVICES] = {};
  auto kern = conv_gemm_dual_kernel<HAS_A2, HAS_PRE, SIMPLE, PREC>;
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), max_bytes, lds_granted);
    if (e != hipSuccess) return e;
  }
  const int tiles_n_big = (p.N + 127) / 128, tiles_n_small = (p.N + 63) / 64;
  p.tail_begin = tail_begin;
  p.n_big = (tail_begin - p.m_begin) / 128 * tiles_n_big;
  const int n_small = (p.M - tail_begin + 63) / 64 * tiles_n_small;
  WS_DLOG(p, "conv_gemm_dual_kernel<128,128 + 64,64,%s> big=%d small=%d", SIMPLE ? "1x1" : "conv", p.n_big, n_small);
  hipLaunchKernelGGL(kern, dim3(p.n_big + n_small), dim3(256), lds_bytes, stream, p);
  return hipGetLastError();
}

// mode: 0 general, 1 A2 (second addend), 2 PRE (BN-ReLU on A), 3 SIMPLE (1x1 running pointers)
template <int BM, int BN, int WM, int WN, int PREC>
static hipError_t launch_mode(const ConvGemmParams& p, int mode, hipStream_t stream) {
  switch (mode) {
    case 1: return launch_one<BM, BN, WM, WN, true, false, false, PREC>(p, stream);
    case 2: return launch_one<BM, BN, WM, WN, false, true, false, PREC>(p, stream);
    case 3: return launch_one<BM, BN, WM, WN, false, false, true, PREC>(p, stream);
    case 4:   // pre-activated 1x1 layer: running pointers (instantiated for the two square tiles only)
      if constexpr (BM == BN) return launch_one<BM, BN, WM, WN, false, true, true, PREC>(p, stream);
      else return launch_one<BM, BN, WM, WN, false, true, false, PREC>(p, stream);
    default: return launch_one<BM, BN, WM, WN, false, false, false, PREC>(p, stream);
  }
}

// tile-shape switch for the big f16 GEMMs (env WS_BIG_TILES at first use; tools/gemm_probe flips it)
int g_ws_big_tiles = -1;
int g_ws_big_conv = -1;       // the same for the convolution form (tools/gemm_probe's A/B)
int g_ws_epi16 = -1;          // binary16 one-phase epilogue of the 256x256 kernel (tools/gemm_probe's A/B)

template <int PREC>
static hipError_t launch_prec(const ConvGemmParams& p, hipStream_t stream) {
  const bool simple = !p.pre_scale && !p.A2 && p.kh == 1 && p.kw == 1 && p.stride_h == 1 &&
                      p.stride_w == 1 && p.pad_h == 0 && p.pad_w == 0 && p.K % BK == 0 &&
                      p.K == p.Cin;
  const bool simple_geom = p.kh == 1 && p.kw == 1 && p.stride_h == 1 && p.stride_w == 1 && p.pad_h == 0 &&
                           p.pad_w == 0 && p.K % BK == 0 && p.K == p.Cin;
  const int mode = p.A2 ? 1 : (p.pre_scale ? (simple_geom ? 4 : 2) : (simple ? 3 : 0));
  const int slots = 2 * current_device_cus();
  // fp32 back-end, plain 1x1 layer with a bias / ReLU / BN epilogue: whole rounds of 128x128 tiles go to the
  // persistent kernel (gemm_f32_stream.hip), the remaining rows re-enter below with m_begin set
  if constexpr (PREC == 0) {
    // (... and the 3x3 / stride 1 layers with 32-channel K-tiles inside one tap: the same kernel's CONV form)
    if (mode == 3 || (mode == 0 && gemm_f32_stream_is_conv3(p))) {
      const int srows = gemm_f32_stream_rows(p, slots / 2);
      if (srows > 0) {
        hipError_t e = launch_gemm_f32_stream(p, srows, slots / 2, stream);
        if (e != hipSuccess || p.m_begin + srows >= p.M) return e;
        ConvGemmParams rest = p;
        rest.m_begin = p.m_begin + srows;
        rest.colsumsq = nullptr;                  // (the tile kernels leave column sums only)
        e = launch_prec<PREC>(rest, stream);
        if (e == hipSuccess && p.colsumsq)
          e = launch_colsumsq_rows(p.D, p.ldd, p.d_off, rest.m_begin, p.M, p.Hout * p.Wout, p.N, p.colsumsq, stream);
        return e;
      }
    }
    if (p.colsumsq) {
      // sums of squares of a layer that the persistent kernel does not take: the tile kernels as usual, then one pass
      // over the stored rows
      ConvGemmParams q = p;
      q.colsumsq = nullptr;
      hipError_t e = launch_prec<PREC>(q, stream);
      if (e == hipSuccess)
        e = launch_colsumsq_rows(p.D, p.ldd, p.d_off, p.m_begin, p.M, p.Hout * p.Wout, p.N, p.colsumsq, stream);
      return e;
    }
  }
  // f16 back-end, plain 1x1 layer: the K-tile-64 kernels (binary16 activations when the producer left
  // them -- then the fp32 tensor may not even exist, so every tile size must take this route)
  const bool fast16 = PREC == 2 && mode == 3 && p.splitk <= 1 && p.K % FBK == 0 && p.N > 64 &&
                      (!p.A16 || (p.lda16 & 7) == 0) && (p.a_off & 7) == 0 && (p.lda & 7) == 0;
  constexpr int dma = 1;
  // (the DMA kernel addresses its operands with 32-bit element offsets)
  const bool use_dma = fast16 && p.A16 && dma && (long long)p.M * p.lda16 < (1LL << 31) &&
                       (long long)p.N * p.ldw < (1LL << 31);
  const int rows = p.M - p.m_begin;             // rows this launch covers (m_begin > 0: a peeled tail)
  // f16 back-end, general convolution on binary16 activations: the DMA kernel in CONV form
  // (32-bit element offsets again: the whole input tensor and the weight matrix must lie below 2^31 elements; the
  // 2-D models refuse an engine chunk beyond that at reserve() -- ResNet221's 128-channel stage 1 reaches it at
  // ~1060 two-second utterances -- so this is a second line of defence for direct callers)
  const long long in_elems = (long long)(p.M / (p.Hout * p.Wout > 0 ? p.Hout * p.Wout : 1)) * p.Hin * p.Win * p.lda16;
  const bool conv16 = PREC == 2 && dma && p.A16 && !p.A2 && !p.pre_scale && !fast16 && p.splitk <= 1 &&
                      (p.Cin & 7) == 0 && (p.lda16 & 7) == 0 && (p.a_off & 7) == 0 && !p.pool_partial &&
                      in_elems < (1LL << 31) && (long long)p.N * p.ldw < (1LL << 31);
  if (conv16) {
    // N <= 32 (ResNet stage 1, CAM++ head: K = 9 * 32): K-tile 32 divides K exactly, 10-KB stages
    // -> 4 stages and 4 workgroups per CU hide the DMA round trip
    if (p.N <= 32) return launch_f16_dma<128, 32, 32, 4, 4, 4, true>(p, stream);
    if (p.N <= 64) return launch_f16_dma<128, 64, 64, 2, 4, 2, true>(p, stream);
    // 256-wide layers whose K-tiles lie inside one filter tap: whole rounds on the phase-staggered 256x256
    // kernel (ResNet stage 4 / the 256-plane bottlenecks), the rest re-enters with m_begin set
    int& bigc = g_ws_big_conv;
    if (bigc < 0) bigc = 1;                                   // (tools/gemm_probe sets it to 0 for its A/B)
    if (bigc && p.N % 256 == 0 && p.Cin % 64 == 0 && p.K % 64 == 0 && !p.bias_img && !p.residual &&
        !p.seg_scale && !p.colsum && !p.D2) {
      const long long cus = slots / 2, tiles_n = p.N / 256, tiles_m = (rows + 255) / 256;
      const long long rounds = tiles_m * tiles_n / cus;
      if (rounds >= 1) {
        long long main_tiles_m = rounds * cus / tiles_n;
        if ((tiles_m * tiles_n) % cus == 0 || (tiles_m * tiles_n) % cus * 10 > cus * 8) main_tiles_m = tiles_m;
        ConvGemmParams mainb = p;
        if (main_tiles_m < tiles_m) mainb.M = p.m_begin + (int)(main_tiles_m * 256);
        hipError_t e = launch_f16_p8<true>(mainb, stream);
        if (e != hipSuccess || main_tiles_m >= tiles_m) return e;
        ConvGemmParams rest = p;
        rest.m_begin = mainb.M;
        return launch_prec<PREC>(rest, stream);
      }
    }
    const long long blocks128 = (long long)((rows + 127) / 128) * ((p.N + 127) / 128);
    if (blocks128 * 2 < slots) return launch_f16_dma<64, 64, 64, 2, 4, 2, true>(p, stream);
    ConvGemmParams mainc = p, tailc = p;
    const long long tiles_m = (rows + 127) / 128, tiles_n = (p.N + 127) / 128;
    const long long total = tiles_m * tiles_n, rem = total % slots;
    bool peelc = false;
    if (total > slots && rem != 0 && rem * 10 <= slots * 7) {
      const long long main_tiles_m = (total - rem) / tiles_n;
      if (main_tiles_m > 0 && main_tiles_m < tiles_m) {
        mainc.M = p.m_begin + (int)(main_tiles_m * 128);
        tailc.m_begin = mainc.M;
        peelc = true;
      }
    }
    hipError_t e = launch_f16_dma<128, 128, 64, 2, 4, 2, true>(mainc, stream);
    if (e != hipSuccess || !peelc) return e;
    return launch_f16_dma<64, 64, 64, 2, 4, 2, true>(tailc, stream);
  }
  // 256x256 tiles (one 8-wave workgroup per CU) for whole rounds of the big f16 GEMMs; what is left
  // re-enters below with m_begin set
  int& big = g_ws_big_tiles;
  // OFF by default.  Measured in tools/gemm_probe (bare bias/ReLU epilogue): +6 % at N = K = 1536,
  // -17 % at N = K = 512 (one workgroup per CU quantises badly); inside the model, where the wide
  // layer also emits column sums and a binary16 copy through the quadrant epilogue, -3 % end to end.
  if (big < 0) { const char* ev = getenv("WS_BIG_TILES"); big = ev ? atoi(ev) : 3; }
  if (g_ws_epi16 < 0) g_ws_epi16 = 1;                         // (tools/gemm_probe sets it to 0 for its A/B)
  // D16-only layers finish through the binary16 one-phase epilogue (g_ws_epi16 = 0: the fp32 two-phase one)
  const bool epi16_ok = g_ws_epi16 && big >= 2 && p.D16 && !p.D && !p.D2 && p.act != ACT_TANH && (p.ldd16 & 7) == 0 &&
                        (p.d_off & 7) == 0 && (!p.colsum || p.Hout * p.Wout >= 64);
  // (the row mask of ragged batches lives in that binary16 epilogue only -- the fp32 quadrant epilogue has no row
  // operands -- and is written for H = 1; other masked layers stay on the 128x128 kernels)
  // (its lenA / lenB logic lets a 64-row group span at most two utterances: frames per slot >= 64)
  const bool mask_ok = !p.row_len || (epi16_ok && p.Hout == 1 && p.Hout * p.Wout >= 64);
  if (use_dma && big && p.N % 256 == 0 && p.N >= (big >= 3 ? 512 : 1024) && !p.pool_partial && !p.bias_img &&
      !p.residual && !p.residual16 && !p.seg_scale && mask_ok) {
    const long long cus = slots / 2, tiles_n = p.N / 256, tiles_m = (rows + 255) / 256;
    const long long rounds = tiles_m * tiles_n / cus;
    if (rounds >= 1) {
      long long main_tiles_m = rounds * cus / tiles_n;
      // the last round may also be nearly full: then no tail at all
      if ((tiles_m * tiles_n) % cus == 0 || (tiles_m * tiles_n) % cus * 10 > cus * (big >= 3 ? 4 : 8)) main_tiles_m = tiles_m;
      ConvGemmParams mainb = p;
      if (main_tiles_m < tiles_m) mainb.M = p.m_begin + (int)(main_tiles_m * 256);
      mainb.epi16 = epi16_ok;

      hipError_t e = big >= 2 ? launch_f16_p8<false>(mainb, stream) : launch_f16_dma<256, 256, 64, 2>(mainb, stream);
      if (e != hipSuccess || main_tiles_m >= tiles_m) return e;
      ConvGemmParams rest = p;
      rest.m_begin = mainb.M;
      return launch_prec<PREC>(rest, stream);
    }
  }
  // Small problems (fewer 128-row tiles than half the chip's block slots): 64x64 tiles put 2-4x
  // more workgroups in flight (CAM++'s dense layers are [B*T/2 x 32..128] GEMMs).
  {
    const long long blocks128 = (long long)((rows + 127) / 128) * ((p.N + 127) / 128);
    if (p.splitk <= 1 && blocks128 * 2 < slots) {
      if (use_dma) return launch_f16_dma<64, 64, 64, 2>(p, stream);
      if (fast16) return p.A16 ? launch_f16_fast<64, 64, false>(p, stream) : launch_f16_fast<64, 64, true>(p, stream);
      return launch_mode<64, 64, 2, 2, PREC>(p, mode, stream);
    }
  }
  if (p.N <= 32) return launch_mode<128, 32, 4, 1, PREC>(p, mode, stream);
  if (p.N <= 64) return launch_mode<128, 64, 4, 1, PREC>(p, mode, stream);
  // Tail peeling.  128x128 tiles run two per CU; a last partial round of tiles costs a whole tile
  // time on a mostly idle chip (measured: 94 -> 123 TF at K = 512 when the tile count is a
  // multiple of the slot count).  Rows beyond the last full round go to a 64x64-tile launch that
  // spreads them over all CUs.  (A forked side stream for the tail was measured slower than
  // same-stream order: the cross-queue event dependencies cost more than the overlap gains.)
  ConvGemmParams main = p, tail = p;
  bool peel = false;
  if (p.splitk <= 1) {
    const long long tiles_m = (rows + 127) / 128, tiles_n = (p.N + 127) / 128;
    const long long total = tiles_m * tiles_n, rem = total % slots;
    if (total > slots && rem != 0 && rem * 10 <= slots * 7) {
      const long long main_tiles_m = (total - rem) / tiles_n;
      if (main_tiles_m > 0 && main_tiles_m < tiles_m) {
        main.M = p.m_begin + (int)(main_tiles_m * 128);
        tail.m_begin = main.M;
        peel = true;
      }
    }
  }
  if (peel && !use_dma && !fast16 && (mode == 0 || mode == 3)) {
    // both tile classes in one grid (conv_gemm_dual_kernel)
    return mode == 3 ? launch_dual<false, false, true, PREC>(p, main.M, stream)
                     : launch_dual<false, false, false, PREC>(p, main.M, stream);
  }
  hipError_t e;
  if (use_dma)   // measured alternatives: 3 stages / 4 waves / one workgroup per CU -16 %; 3 stages / 8 waves
                 // (64x32 per wave) / one workgroup per CU -10 %; 4 stages of K-tile 32 -5 %
    e = launch_f16_dma<128, 128, 64, 2>(main, stream);
  else
    e = fast16 ? (main.A16 ? launch_f16_fast<128, 128, false>(main, stream)
                           : launch_f16_fast<128, 128, true>(main, stream))
               : launch_mode<128, 128, 2, 2, PREC>(main, mode, stream);
  if (e != hipSuccess || !peel) return e;
  if (use_dma) return launch_f16_dma<64, 64, 64, 2>(tail, stream);
  if (fast16)
    return tail.A16 ? launch_f16_fast<64, 64, false>(tail, stream) : launch_f16_fast<64, 64, true>(tail, stream);
  return launch_mode<64, 64, 2, 2, PREC>(tail, mode, stream);
}

hipError_t launch_conv_gemm(const ConvGemmParams& p, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0) return hipSuccess;
  // 16-byte paths: channel counts / offsets / row strides must be multiples of 4 floats
  if ((p.N | p.Cin | p.lda | p.a_off | p.ldw | p.ldd | p.d_off) & 3) return hipErrorInvalidValue;
  if (p.ldw & 31) return hipErrorInvalidValue;
  if (p.A2 && ((p.lda2 | p.a2_off) & 3)) return hipErrorInvalidValue;
  if (p.D2 && ((p.ldd2 | p.d2_off | p.d2_col0) & 3)) return hipErrorInvalidValue;
  if (p.residual && ((p.ldr | p.r_off) & 3)) return hipErrorInvalidValue;
  if (p.pre_scale && p.A2) return hipErrorInvalidValue;
  if (p.colsum && (p.splitk > 1 || p.Hout * p.Wout < 64)) return hipErrorInvalidValue;
  if (p.pool_partial && (p.splitk > 1 || p.Hout * p.Wout < 64 || (!p.pool_h && !p.pool_h16) || (p.ldh & 3)))
    return hipErrorInvalidValue;
  // ragged batches: the row mask (row operand epilogue, pooling epilogue, binary16 256x256 epilogue) is
  // not combined with split-K; the pooling / binary16 forms are written for H = 1 (time on the W axis)
  if (p.row_len && (p.splitk > 1 || (p.pool_partial && p.Hout != 1))) return hipErrorInvalidValue;
  if (p.m_begin & 63) return hipErrorInvalidValue;
  if (p.prec == 1) {
    if (!p.Wh || !p.Wl) return hipErrorInvalidValue;
    return launch_prec<1>(p, stream);
  }
  if (p.prec == 2) {
    if (!p.Wh) return hipErrorInvalidValue;
    if (conv3x3_direct_supported(p)) return launch_conv3x3_direct(p, stream);
    return launch_prec<2>(p, stream);
  }
  if (p.colsumsq && (!p.colsum || !p.D)) return hipErrorInvalidValue;
  if (conv3x3_direct_f32_supported(p)) return launch_conv3x3_direct_f32(p, stream);
  return launch_prec<0>(p, stream);
}

// --------------------------------------------------------------------------- split-K reduce
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvGemmParams p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)p.M * p.N) return;
  const int m = (int)(idx / p.N), n = (int)(idx - (long long)m * p.N);
  float v = 0.f;
  for (int z = 0; z < p.splitk; ++z) v += p.partial[((long long)z * p.M + m) * p.N + n];
  if (p.bias) v += p.bias[n];
  if (p.bias_img) v += p.bias_img[(long long)(m / (p.Hout * p.Wout)) * p.N + n];
  if (p.residual) v += p.residual[(long long)m * p.ldr + p.r_off + n];
  if (p.act == ACT_RELU) v = relu_f(v);
  else if (p.act == ACT_TANH) v = tanhf(v);
  if (p.post_scale) v = v * p.post_scale[n] + p.post_shift[n];
  p.D[(long long)m * p.ldd + p.d_off + n] = v;
  if (p.D2 && n >= p.d2_col0) p.D2[(long long)m * p.ldd2 + p.d2_off + (n - p.d2_col0)] = v;
}

hipError_t launch_splitk_reduce(const ConvGemmParams& p, hipStream_t stream) {
  const long long total = (long long)p.M * p.N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     stream, p);
  return hipGetLastError();
}

}  // namespace wsamd
