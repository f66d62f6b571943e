// conv3 of Bottleneck block i (+ residual, ReLU) and conv1 of block i + 1 (+ ReLU) in ONE pass over the block output
// (wespeaker/models/resnet.py:72-107; fp32 back-end, stages 1 - 2 of the Bottleneck ResNets: planes P = 32 / 64).
//
// Those layers are HBM-bound: conv3 reads y2 (M x P) and the residual (M x 4P) and writes the block
// output (M x 4P); the next block's conv1 reads the block output AGAIN and writes y1 (M x P) -- 7.25 GB per block of
// stage 1 at 256 utterances, of which 2.07 GB are that second read.  Here the block output goes from the accumulators
// of the first MFMA straight into the second one and is stored once:
//   * a wavefront owns 32 pixels; per block of 32 output channels `ob` it runs the P / 2 k-steps of conv3
//     (v_mfma_f32_32x32x2_f32: A = weights, B = y2 -- C^T blocks, so that a lane owns four consecutive channels of ONE
//     pixel and every global access is 16 bytes), adds bias + residual, applies ReLU, stores, and feeds the 16 registers
//     of the block -- lane (pixel, half h) holds channels 32 ob + (r & 3) + 8 (r >> 2) + 4 h -- to conv1 as its B
//     operand: k-step r of block ob multiplies them with W1[:, that channel], so no value moves between lanes;
//   * the same permutation trick feeds conv3: lane half h loads the 16-B chunks c = h (mod 2) of its pixel's y2 row
//     and k-step j pairs channels 8 (j >> 2) + 4 h + (j & 3);
//   * both weight matrices live in LDS in exactly the order the lanes consume them (one ds_read_b128 = the A operands
//     of four k-steps), re-laid by the workgroup at start; biases beside them;
//   * raw buffer loads / stores whose descriptors end at row M: the last, partial block of pixels needs no branch
//     (no control flow inside the streaming loop).
// Summation order differs from the tile kernels' (k permuted): results agree to fp32 rounding, not bit for bit.
#include "kernels.h"
#include <cstdio>

namespace wsamd {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));

// P = 32: 256 threads, four workgroups per CU (33 KB of LDS each); P = 64: the two weight matrices fill 129 KB -- one
// workgroup of 512 threads per CU
// PN: planes of the NEXT block (= P inside a stage, 2 P at the stage 1 -> 2 transition)
// MASKT: a ragged batch (BneckFuseParams::row_len) -- pixels at or beyond their utterance's own width are stored as
// zeros in BOTH outputs (the next block's conv1 of a zero pixel would be relu(bias)); the batch's widths sit in LDS
// behind the biases, a lane works out once per block of pixels whether ITS pixel is real (two integer divisions).
template <int P, int PN, int NT, int MINB, bool MASKT = false>
__global__ __launch_bounds__(NT, MINB) void bneck_c3c1_kernel(const BneckFuseParams p) {
  constexpr int NWV = NT / 64;
  constexpr int CX = 4 * P, OB = CX / 32, OB2 = PN / 32, KS3 = P / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const W3l = lds;                                  // [OB][KS3 / 4][64][4]
  float* const W1l = W3l + OB * KS3 * 64;                  // [OB2][OB][4][64][4]
  float* const b3l = W1l + OB2 * OB * 16 * 64;             // [CX]
  float* const b1l = b3l + CX;                             // [PN]
  int* const lens_l = reinterpret_cast<int*>(b1l + PN);    // MASKT: [M / HW <= 1024]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  // ---- weights into LDS in consumption order
  for (int i = tid; i < OB * KS3 * 64; i += NT) {
    const int e = i & 3, l = (i >> 2) & 63, q = (i >> 8) % (KS3 / 4), ob = i / (KS3 * 64);
    const int j = 4 * q + e;                                // k-step; lane l = (row l & 31, half l >> 5)
    W3l[i] = p.W3[(long long)(32 * ob + (l & 31)) * p.ldw3 + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3)];
  }
  for (int i = tid; i < OB2 * OB * 16 * 64; i += NT) {
    const int e = i & 3, l = (i >> 2) & 63, g = (i >> 8) & 3, ob = (i >> 10) % OB, ob2 = i / (OB * 1024);
    const int r = 4 * g + e;                                // register of the conv3 block = k-step of conv1
    W1l[i] = p.W1[(long long)(32 * ob2 + (l & 31)) * p.ldw1 + 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)];
  }
  for (int i = tid; i < CX; i += NT) b3l[i] = p.b3 ? p.b3[i] : 0.f;
  for (int i = tid; i < PN; i += NT) b1l[i] = p.b1 ? p.b1[i] : 0.f;
  const int nimg = MASKT ? p.M / p.HW : 0;
  if (MASKT)
    for (int i = tid; i < nimg; i += NT) lens_l[i] = p.row_len[i];
  __syncthreads();

  const __amdgpu_buffer_rsrc_t y2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.y2), 0, (unsigned)((long long)p.M * P * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, (unsigned)((long long)p.M * p.ldr * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t outr = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (unsigned)((long long)p.M * CX * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t y1r = __builtin_amdgcn_make_buffer_rsrc(p.y1, 0, (unsigned)((long long)p.M * PN * 4), 0x00020000);
  const int nblk = (p.M + 31) / 32;
  const int nwaves = gridDim.x * NWV;
  int blk = blockIdx.x * NWV + wave;
  // lane parts of the addresses (bytes): pixel li of the block, the lane half's 16-B chunk
  const unsigned y2v = (unsigned)li * (P * 4) + lh * 16;           // + 32 * q: chunk 2 q + lh
  const unsigned resv = (unsigned)li * (p.ldr * 4) + lh * 16;      // + 128 * ob + 32 * g
  const unsigned outv_ = (unsigned)li * (CX * 4) + lh * 16;
  u32x4r y2f[P / 8], rs[4];
  auto load_y2 = [&](int b) {
    const unsigned s = (unsigned)b * (32u * P * 4u);
#pragma unroll
    for (int q = 0; q < P / 8; ++q) y2f[q] = __builtin_amdgcn_raw_buffer_load_b128(y2r, y2v + 32 * q, s, 0);
  };
  auto load_res = [&](int b, int ob) {
    const unsigned s = (unsigned)b * (32u * (unsigned)p.ldr * 4u) + 128u * ob;
#pragma unroll
    for (int g = 0; g < 4; ++g) rs[g] = __builtin_amdgcn_raw_buffer_load_b128(rsr, resv + 32 * g, s, 0);
  };
  if (blk < nblk) { load_y2(blk); load_res(blk, 0); }
  for (; blk < nblk; blk += nwaves) {
    f32x16 y1a[OB2];
#pragma unroll
    for (int o = 0; o < OB2; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) y1a[o][r] = 0.f;
    const int nxt = blk + nwaves < nblk ? blk + nwaves : blk;       // (past the end: this block again, unused)
    bool ok = true;
    if (MASKT) {
      const int m = blk * 32 + li;
      const int img = m / p.HW, rem = m - img * p.HW;
      ok = rem - (rem / p.W) * p.W < lens_l[img < nimg ? img : nimg - 1];
    }
    // (the weight fragments are re-read from LDS for every block of pixels: hoisted out of this loop -- which hipcc
    // does when it can see that the addresses do not change -- they are 128 registers, and the kernel lives on
    // having sixteen wavefronts per CU in flight, not on saving 32 ds_read_b128 per 128 MFMAs)
    int wo = lane * 4;                      // (the OFFSET is made opaque, not the pointer: a laundered pointer loses its
    asm volatile("" : "+v"(wo));            //  LDS address space and every read becomes a flat load on both counters)
    const float* w3p = W3l + wo;
    const float* w1p = W1l + wo;
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) {
      // ---- conv3, output channels [32 ob, 32 ob + 32) of the wavefront's 32 pixels
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int q = 0; q < KS3 / 4; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(w3p + (ob * (KS3 / 4) + q) * 256);
        const f32x4 a = __builtin_bit_cast(f32x4, y2f[q]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], a[e], acc, 0, 0, 0);
      }
      // ---- + bias + residual, ReLU, store; the residual of the next channel block (or of the next pixel block,
      // with its y2 rows) is requested behind the last use of the registers it lands in
      f32x4 o4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(b3l + 32 * ob + 8 * g + 4 * lh);
        const f32x4 r = __builtin_bit_cast(f32x4, rs[g]);
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[g][e] = relu_f((acc[4 * g + e] + b[e]) + r[e]);
        if (MASKT && !ok) o4[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if (ob + 1 < OB) load_res(blk, ob + 1);
      else { load_y2(nxt); load_res(nxt, 0); }
      {
        const unsigned s = (unsigned)blk * (32u * CX * 4u) + 128u * ob;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4r, o4[g]), outr, outv_ + 32 * g + s, 0, 0);
      }
      // ---- conv1 of the next block: these 32 channels are 16 of its k-steps
#pragma unroll
      for (int o = 0; o < OB2; ++o)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(w1p + ((o * OB + ob) * 4 + g) * 256);
#pragma unroll
          for (int e = 0; e < 4; ++e) y1a[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], o4[g][e], y1a[o], 0, 0, 0);
        }
    }
    // ---- y1 of the next block: bias, ReLU, store (M x PN rows)
#pragma unroll
    for (int o = 0; o < OB2; ++o) {
      const unsigned s = (unsigned)blk * (32u * PN * 4u) + 128u * o;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(b1l + 32 * o + 8 * g + 4 * lh);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = relu_f(y1a[o][4 * g + e] + b[e]);
        if (MASKT && !ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4r, v), y1r, (unsigned)li * (PN * 4) + lh * 16 + 32 * g + s, 0, 0);
      }
    }
  }
}
}  // namespace

bool bneck_fuse_supported(const BneckFuseParams& p) {
  return ((p.P == 32 && (p.PN == 32 || p.PN == 64)) || (p.P == 64 && p.PN == 64)) && p.M > 0 && p.ldw3 >= p.P && p.ldw1 >= 4 * p.P && (p.ldr & 3) == 0 &&
         p.ldr >= 4 * p.P && (long long)p.M * p.ldr * 4 < (1LL << 32) && (long long)p.M * 4 * p.P * 4 < (1LL << 32) &&
         ((reinterpret_cast<unsigned long long>(p.y2) | reinterpret_cast<unsigned long long>(p.res) |
           reinterpret_cast<unsigned long long>(p.out) | reinterpret_cast<unsigned long long>(p.y1)) & 15) == 0 &&
         // a ragged batch: whole images, at most 1024 of them (their widths go to LDS)
         (!p.row_len || (p.HW > 0 && p.W > 0 && p.HW % p.W == 0 && p.M % p.HW == 0 && p.M / p.HW <= 1024));
}

template <int P, int PN, int NT, int MINB, bool MASKT = false>
static hipError_t launch_bneck_fuse_p(const BneckFuseParams& p, hipStream_t stream) {
  constexpr int CX = 4 * P, OB = CX / 32, OB2 = PN / 32, KS3 = P / 2, NWV = NT / 64;
  const size_t lds = (size_t)(OB * KS3 * 64 + OB2 * OB * 16 * 64 + CX + PN) * sizeof(float) + (MASKT ? 4096 : 0);
  auto kern = bneck_c3c1_kernel<P, PN, NT, MINB, MASKT>;
  static size_t lds_granted[WS_MAX_DEVICES] = {};
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_granted);
    if (e != hipSuccess) return e;
  }
  const int nblk = (p.M + 31) / 32;
  int grid = current_device_cus() * MINB;
  if (grid > (nblk + NWV - 1) / NWV) grid = (nblk + NWV - 1) / NWV;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, stream, p);
  return hipGetLastError();
}

hipError_t launch_bneck_fuse(const BneckFuseParams& p, hipStream_t stream) {
  if (!bneck_fuse_supported(p)) return hipErrorInvalidValue;
  if (dispatch_log_enabled()) {
    char key[160];
    snprintf(key, sizeof(key), "rows=%d planes=%d next=%d%s -> bneck_c3c1_kernel (conv3 + residual + ReLU, next block's conv1 + ReLU)",
             p.M, p.P, p.PN, p.row_len ? " +mask" : "");
    dispatch_log_note_text(key);
  }
  if (p.row_len) {
    if (p.P == 64) return launch_bneck_fuse_p<64, 64, 512, 1, true>(p, stream);
    return p.PN == 32 ? launch_bneck_fuse_p<32, 32, 256, 4, true>(p, stream)
                      : launch_bneck_fuse_p<32, 64, 256, 3, true>(p, stream);
  }
  if (p.P == 64) return launch_bneck_fuse_p<64, 64, 512, 1>(p, stream);
  return p.PN == 32 ? launch_bneck_fuse_p<32, 32, 256, 4>(p, stream) : launch_bneck_fuse_p<32, 64, 256, 3>(p, stream);
}

}  // namespace wsamd
