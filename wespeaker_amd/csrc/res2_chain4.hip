// WS_BUILD_FLAGS: -mllvm -amdgpu-mfma-vgpr-form=1
// (the MFMA accumulators of this file live in VGPRs: with the AGPR form hipcc reuses two accumulator quads for all
//  row tiles and copies every finished tile out behind `s_nop 8` -- an exposed MFMA latency per pair of row tiles)
//
// Fused Res2 chain of an ECAPA SE-Res2Block (wespeaker/models/ecapa_tdnn.py:58-78), four-wavefront fp32 form.
// The eight-wavefront kernels, the binary16 forms and the dispatcher are in res2_fused.hip.
#include "kernels.h"

namespace wsamd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef WS_TRACE
__device__ unsigned long long g_res2c4_trace[64];
unsigned long long* res2c4_trace_buffer_address() {
  unsigned long long* q = nullptr;
  (void)hipGetSymbolAddress(reinterpret_cast<void**>(&q), HIP_SYMBOL(g_res2c4_trace));
  return q;
}
#define WS_RSTAMP(i) \
  if (blockIdx.x == 9 && threadIdx.x == 0) g_res2c4_trace[(i)] = __builtin_readcyclecounter();
#else
#define WS_RSTAMP(i)
#endif

// ------------------------------------------------------------------------------------------------
// Four-wavefront form of the fp32 chain at w = 64 (round 6): ONE wavefront per SIMD.
// The eight-wavefront kernel above puts two wavefronts on every SIMD, and the matrix pipe serves their MFMA bursts
// one after the other (tools/res2_probe: per step the older wavefront's 336 MFMAs, then the younger one's, then both
// epilogues and two barriers -- 96 us per 256 x 2 s launch for 55 us of MFMA work).  Here wavefront w owns the 16
// output channels [16 w, 16 w + 16) of ALL row tiles, so that everything that is not an MFMA sits behind an MFMA of
// the SAME instruction stream (the discipline of gemm_f32_stream.hip / astp_fused.hip):
//  * two copies of the running activation: step s reads X[s & 1] and writes X[(s + 1) & 1] -- one barrier per step,
//    and a pair of row tiles is finished (bias, ReLU, BN, + next split, global store, LDS write-back) in small pieces
//    behind the MFMAs of the NEXT pair; only the last pair's epilogue is exposed;
//  * the activation fragments run one k-group ahead across pair boundaries;
//  * next step's weights and the next split of y1 are requested at the start of the step into their own registers
//    (one wavefront per SIMD: 512 VGPRs are there).
// Same k order per accumulator as the kernel above (k-groups ascending, 4 k per group) -> the same bits.
// Whole utterances only, NTILE = ceil(T / 16) row tiles (9 .. 13: 129 .. 208 frames, one workgroup each); everything
// else stays on the kernel above.
//
// What the instruction stream must not contain (measured with tools/res2_probe, round 6): a global load or store is
// ~4 k cycles away under this kernel's load, and gfx950 has ONE in-order counter for loads and stores, so
//  * a wait for a load is a wait for every store issued before it;
//  * hipcc keeps the data registers of a store untouched until the store has left the counter;
//  * at a control-flow join its counter model falls back to `s_waitcnt vmcnt(0)`.
// Hence: no runtime branch inside a step (NTILE is a template parameter; rows behind a shorter utterance's end are
// dropped / zero-filled by the bounds check of raw buffer accesses, the last step issues the same -- unused -- loads
// as the others); every global load is issued at least three pairs of row tiles (~10 k cycles) before its first use;
// the finished values wait for their store in four register sets of their own.
template <int NTILE>
__global__ __launch_bounds__(256) void res2_chain4_kernel(const Res2ChainParams p) {
  constexpr int W = 64, XS = W + 8, KG = 3 * W / 16, CAP = NTILE * 16, NP = (NTILE + 1) / 2;
  typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float X[];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int d = p.dil;
  // (ragged batch: rows beyond an utterance's own length are the conv's zero padding; readfirstlane: the buffer
  // descriptors below must be scalar, or every access becomes a waterfall loop)
  const int T = __builtin_amdgcn_readfirstlane(p.lens ? p.lens[b] : p.T);
  if (T <= 0) return;
  const int plane = (CAP + 2 * d) * XS;                    // floats per copy of the running activation
  // the utterance's rows as two raw buffers that END behind its last frame: a row >= T loads zeros and is not stored
  const unsigned ld1 = (unsigned)p.ldy1, ld2 = (unsigned)p.ldy2;
  const __amdgpu_buffer_rsrc_t y1r = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.y1) + (long long)b * p.T * p.ldy1, 0, (unsigned)T * ld1 * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t y2r = __builtin_amdgcn_make_buffer_rsrc(
      p.y2 + (long long)b * p.T * p.ldy2, 0, (unsigned)T * ld2 * 4u, 0x00020000);
  const int co = wave * 16 + li;                           // weight row of this lane's A fragment
  const int c0 = wave * 16 + lq * 4;                       // first of the lane's 4 output channels
  const unsigned v1 = ((unsigned)li * ld1 + c0) * 4u;      // lane parts of the y1 / y2 addresses (bytes); the row
  const unsigned v2 = ((unsigned)li * ld2 + c0) * 4u;      // tile and the split go into the scalar offset
  const unsigned woff = (unsigned)co * (unsigned)p.ldw + lq * 4;

  WS_RSTAMP(0)
  // split 0 through registers; the first step's weights, channel vectors and y1 rows are requested right behind it,
  // and both planes are zeroed while all of that is in flight
  f32x4 bw[KG], bwn[KG], bias, sc, sh, biasn, scn, shn;
  u32x4r y1n[NTILE];
  {
    constexpr int NSTG = CAP * (W / 4) / 256;
    u32x4r stg[NSTG];
    const unsigned vs = ((unsigned)(tid >> 4) * ld1 + (tid & 15) * 4) * 4u;
#pragma unroll
    for (int k = 0; k < NSTG; ++k) stg[k] = __builtin_amdgcn_raw_buffer_load_b128(y1r, vs, k * 16 * ld1 * 4u, 0);
#pragma unroll
    for (int g = 0; g < KG; ++g) bw[g] = *reinterpret_cast<const f32x4*>(p.w[0] + woff + g * 16);
    bias = *reinterpret_cast<const f32x4*>(p.bias[0] + c0);
    sc = *reinterpret_cast<const f32x4*>(p.scale[0] + c0);
    sh = *reinterpret_cast<const f32x4*>(p.shift[0] + c0);
#pragma unroll
    for (int mt = 0; mt < NTILE; ++mt)
      y1n[mt] = __builtin_amdgcn_raw_buffer_load_b128(y1r, v1, (mt * 16 * ld1 + W) * 4u, 0);
    for (int i = tid * 4; i < 2 * plane; i += 256 * 4)
      *reinterpret_cast<f32x4*>(&X[i]) = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int k = 0; k < NSTG; ++k)
      *reinterpret_cast<u32x4r*>(&X[((tid >> 4) + k * 16 + d) * XS + (tid & 15) * 4]) = stg[k];
  }
  __syncthreads();
  // (collected HERE: a load still in flight at the loop's entry makes hipcc's counter model wait for it in every
  // iteration -- one `s_waitcnt vmcnt` per k-group of pair 0, each of them a wait for the step's own fresh loads)
#pragma unroll
  for (int g = 0; g < KG; ++g) asm volatile("" : "+v"(bw[g]));
#pragma unroll
  for (int mt = 0; mt < NTILE; ++mt) asm volatile("" : "+v"(y1n[mt]));
  asm volatile("" : "+v"(bias), "+v"(sc), "+v"(sh));
  WS_RSTAMP(1)

  f32x4 ev[4] = {};
  for (int step = 0; step < 7; ++step) {
    WS_RSTAMP(2 + step * 5)
    asm volatile("" : "+v"(ev[0]), "+v"(ev[1]), "+v"(ev[2]), "+v"(ev[3]));
    const float* cur = X + (step & 1) * plane;
    float* nxt = X + ((step + 1) & 1) * plane;
    const int sn = step < 6 ? step + 1 : 6;                // the last step repeats its own (unused) loads
    // (measured: pinning these four pointers in SGPRs at the top of the step costs 200 cycles per PAIR, 3.75 k
    //  instead of 3.55 k; left to hipcc, their scalar loads cost pair 2 ~400 cycles once)
    const float *wn = p.w[sn], *bn = p.bias[sn], *scp = p.scale[sn], *shp = p.shift[sn];
    const unsigned split2 = (unsigned)(step < 6 ? step + 2 : 7) * W * 4u;   // y1 split the step AFTER next adds
    f32x4 acc[NTILE];
#pragma unroll
    for (int mt = 0; mt < NTILE; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* xbase = cur + li * XS + lq * 4;
    auto xaddr = [&](int mt, int g) {
      const int tap = g / (W / 16), cg = g % (W / 16);
      return xbase + (mt * 16 + tap * d) * XS + cg * 16;
    };
    float* xo = nxt + (li + d) * XS + c0;
    // The epilogue of pair (mp, mp + 1) as 36 micro-operations of one or two instructions, one behind every second
    // MFMA of the NEXT pair (a clump of ~30 VALU instructions in front of a k-group's MFMAs idles the matrix pipe:
    // 37.7 instead of 34.2 cycles per MFMA, tools/res2_probe).  Per tile: 4 channels x (bias add | ReLU | BN) , then
    // store | + next split | zero-fill select x 2 | LDS write-back of the next step's input | request for the y1 rows
    // of the step after next.
    float et = 0.f;
    f32x4 enx = {};
    auto epi_op = [&](int mp, int m) {
      if (m < 0 || m >= 36) return;
      const int mt = mp + m / 18, j = m % 18, set = (mp & 2) + m / 18;
      if (mt >= NTILE) return;
      if (j < 12) {
        const int r = j / 3;
        if (j % 3 == 0) et = acc[mt][r] + bias[r];
        else if (j % 3 == 1) et = relu_f(et);
        else ev[set][r] = et * sc[r] + sh[r];
        return;
      }
      const bool ok = mt * 16 + li < T;
      if (j == 12)     // (offset in the VGPR: hipcc pads a wide store against an overwrite of its data registers only
                       //  when the store has no scalar offset -- DESIGN.md 6.0, second ISA rule)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4r, ev[set]), y2r,
                                               v2 + (mt * 16 * ld2 + step * W) * 4u, 0, 0);
      else if (j == 13) enx = ev[set] + __builtin_bit_cast(f32x4, y1n[mt]);
      else if (j == 14) { enx[0] = ok ? enx[0] : 0.f; enx[1] = ok ? enx[1] : 0.f; }   // (rows behind the end stay
      else if (j == 15) { enx[2] = ok ? enx[2] : 0.f; enx[3] = ok ? enx[3] : 0.f; }   //  the conv's zero padding)
      else if (j == 16) *reinterpret_cast<f32x4*>(xo + mt * 16 * XS) = enx;
      else y1n[mt] = __builtin_amdgcn_raw_buffer_load_b128(y1r, v1, mt * 16 * ld1 * 4u + split2, 0);
    };

    f32x4 fa[2][2];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int mp = 2 * q;
      const bool two = mp + 1 < NTILE;
#ifdef WS_TRACE
      if (step == 3) { WS_RSTAMP(41 + q) }
#endif
      if (q == 0) {
        fa[0][0] = *reinterpret_cast<const f32x4*>(xaddr(0, 0));
        fa[0][1] = *reinterpret_cast<const f32x4*>(xaddr(1, 0));
      }
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        if (g + 1 < KG) {
          fa[(g + 1) & 1][0] = *reinterpret_cast<const f32x4*>(xaddr(mp, g + 1));
          if (two) fa[(g + 1) & 1][1] = *reinterpret_cast<const f32x4*>(xaddr(mp + 1, g + 1));
        } else if (q + 1 < NP) {                           // (KG is even: the next pair's group 0 lands in set 0)
          fa[0][0] = *reinterpret_cast<const f32x4*>(xaddr(mp + 2, 0));
          if (mp + 3 < NTILE) fa[0][1] = *reinterpret_cast<const f32x4*>(xaddr(mp + 3, 0));
        }
        if (q == 2 && g < 6) {                             // next step's weights / channel vectors: two per k-group
          bwn[2 * g] = *reinterpret_cast<const f32x4*>(wn + woff + (2 * g) * 16);
          bwn[2 * g + 1] = *reinterpret_cast<const f32x4*>(wn + woff + (2 * g + 1) * 16);
        }
        if (q == 2 && g == 6) {
          biasn = *reinterpret_cast<const f32x4*>(bn + c0);
          scn = *reinterpret_cast<const f32x4*>(scp + c0);
          shn = *reinterpret_cast<const f32x4*>(shp + c0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[mp] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[g][s], fa[g & 1][0][s], acc[mp], 0, 0, 0);
          if (two)
            acc[mp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[g][s], fa[g & 1][1][s], acc[mp + 1], 0, 0, 0);
          if (q > 0) epi_op(mp - 2, 4 * g + s - 4);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 36; ++m) epi_op(2 * (NP - 1), m);
    WS_RSTAMP(3 + step * 5)
#pragma unroll
    for (int g = 0; g < KG; ++g) bw[g] = bwn[g];
    bias = biasn; sc = scn; sh = shn;
    WS_RSTAMP(4 + step * 5)
    WS_RSTAMP(5 + step * 5)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the next step reads what everyone wrote
    WS_RSTAMP(6 + step * 5)
  }
  asm volatile("" ::"v"(ev[0]), "v"(ev[1]), "v"(ev[2]), "v"(ev[3]));
  WS_RSTAMP(40)
}

template <int NTILE>
static hipError_t launch_res2_chain4_n(const Res2ChainParams& p, hipStream_t stream) {
  const size_t lds = (size_t)2 * (NTILE * 16 + 2 * p.dil) * (64 + 8) * sizeof(float);
  auto kern = res2_chain4_kernel<NTILE>;
  static size_t lds_granted[WS_MAX_DEVICES] = {};
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_granted);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(p.B), dim3(256), lds, stream, p);
  return hipGetLastError();
}
hipError_t launch_res2_chain4(Res2ChainParams p, hipStream_t stream) {
  p.tiles = 1; p.tile_rows = 13 * 16;
  switch ((p.T + 15) / 16) {
    case 9: return launch_res2_chain4_n<9>(p, stream);
    case 10: return launch_res2_chain4_n<10>(p, stream);
    case 11: return launch_res2_chain4_n<11>(p, stream);
    case 12: return launch_res2_chain4_n<12>(p, stream);
    default: return launch_res2_chain4_n<13>(p, stream);
  }
}

}  // namespace wsamd
