// Direct 3x3 convolution, 32 -> 32 channels, on binary16 channels-last maps (f16 back-end).
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
#include "kernels.h"

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

template <int SH, int SW>
__global__ __launch_bounds__(256, 2) void conv3x3_c32_f16_kernel(const ConvGemmParams p, int pyb, int pxb,
                                                                 int total) {
  constexpr int IH = (PH - 1) * SH + 3, IW = (PW - 1) * SW + 3;
  constexpr int RP = IH * IW;                    // region pixels
  constexpr int NP = (RP + 15) / 16;             // 1-KiB pieces (16 pixels each)
  constexpr int PPW = (NP + 3) / 4;              // pieces per wavefront
  constexpr int STAGE = NP * 1024;
  extern __shared__ __attribute__((aligned(16))) char lds_d[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  // ---- weights: A fragments of the 18 k-steps (k = 16 s + 8 lh .. +7 of row cout = li)
  f16x8d wf[18];
#pragma unroll
  for (int s = 0; s < 18; ++s)
    wf[s] = *reinterpret_cast<const f16x8d*>(p.Wh + (long long)li * p.ldw + 16 * s + 8 * lh);

  // ---- DMA roles: piece pi covers region pixels 16 pi .. 16 pi + 15, lane -> (pixel, physical chunk)
  int d_ry[PPW], d_rx[PPW], d_c[PPW], d_slot[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    int pi = wave * PPW + k;
    if (pi > NP - 1) pi = NP - 1;                // surplus wavefronts repeat the last piece
    const int q = pi * 16 + (lane >> 2), pc = lane & 3;
    d_ry[k] = q < RP ? q / IW : -(1 << 20);      // out of region -> predicate false
    d_rx[k] = q - (q / IW) * IW;
    d_c[k] = (pc ^ ((q >> 2) & 3)) * 8;          // logical chunk (halfs) stored at physical slot pc
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
      const uint16_t* src = ok ? p.A16 + (((long long)img * p.Hin + iy) * p.Win + ix) * 32 + d_c[k]
                               : reinterpret_cast<const uint16_t*>(p.zeros);
      dma16_direct(src, base + d_slot[k]);
    }
  };

  // ---- compute roles: lane li -> patch pixel (row 2 wave + li / 16, col li % 16)
  const int ppy = 2 * wave + (li >> 4), ppx = li & 15;
  const int q0 = ppy * SH * IW + ppx * SW;       // region pixel of tap (0, 0)
  // per-channel epilogue constants: after the lane exchange a lane owns channels 16 lh .. 16 lh + 15
  float bias[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) bias[c] = p.bias ? p.bias[16 * lh + c] : 0.f;

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
      const int key = (q >> 2) & 3;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int c = half * 2 + lh;
        const f16x8d a = *reinterpret_cast<const f16x8d*>(base + q * 64 + ((c ^ key) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[2 * tap + half], a, acc, 0, 0, 0);
      }
    }
    // C^T layout: column = pixel li, rows (channels) = (r & 3) + 8 (r >> 2) + 4 lh.  Exchange quads with
    // lane ^ 32 so that lh = 0 owns channels 0..15 and lh = 1 owns 16..31 (contiguous 32 B each).
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
      if (p.residual16) {
        const f16x8d r0 = *reinterpret_cast<const f16x8d*>(p.residual16 + m * p.ldr + 16 * lh);
        const f16x8d r1 = *reinterpret_cast<const f16x8d*>(p.residual16 + m * p.ldr + 16 * lh + 8);
#pragma unroll
        for (int c = 0; c < 8; ++c) { v[c] += (float)r0[c]; v[8 + c] += (float)r1[c]; }
      }
      f16x8d o0, o1;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float a = v[c] + bias[c], b = v[8 + c] + bias[8 + c];
        if (p.act == ACT_RELU) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
        o0[c] = (_Float16)a; o1[c] = (_Float16)b;
      }
      *reinterpret_cast<f16x8d*>(p.D16 + m * p.ldd16 + 16 * lh) = o0;
      *reinterpret_cast<f16x8d*>(p.D16 + m * p.ldd16 + 16 * lh + 8) = o1;
    }
    __syncthreads();                             // everyone is done with this stage before it is refilled
    stage ^= 1;
  }
}

bool conv3x3_direct_supported(const ConvGemmParams& p) {
  return p.prec == 2 && p.A16 && p.D16 && !p.D && !p.A2 && !p.pre_scale && p.kh == 3 && p.kw == 3 &&
         p.dil_h == 1 && p.dil_w == 1 && p.pad_h == 1 && p.pad_w == 1 && p.Cin == 32 && p.N == 32 &&
         p.lda16 == 32 && p.ldd16 == 32 && p.a_off == 0 && p.d_off == 0 && p.K == 288 && p.ldw >= 288 &&
         (p.stride_h == 1 || p.stride_h == 2) && (p.stride_w == 1 || p.stride_w == 2) &&
         !(p.stride_h == 1 && p.stride_w == 2) && !p.residual && !p.colsum && !p.pool_partial &&
         !p.seg_scale && !p.post_scale && !p.bias_img && !p.D2 && p.splitk <= 1 && p.m_begin == 0 &&
         p.act != ACT_TANH && (!p.residual16 || (p.ldr == 32 && p.r_off == 0));
}

template <int SH, int SW>
static hipError_t launch_direct(const ConvGemmParams& p, hipStream_t stream) {
  constexpr int IH = (PH - 1) * SH + 3, IW = (PW - 1) * SW + 3;
  constexpr int NP = (IH * IW + 15) / 16;
  constexpr size_t lds = 2 * (size_t)NP * 1024;
  auto kern = conv3x3_c32_f16_kernel<SH, SW>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int images = p.M / (p.Hout * p.Wout);
  const int pyb = (p.Hout + PH - 1) / PH, pxb = (p.Wout + PW - 1) / PW;
  const long long total = (long long)images * pyb * pxb;
  if (total <= 0) return hipSuccess;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  }
  const size_t per_cu = 160 * 1024 / lds;             // LDS limit; the ~170 VGPRs allow 2 workgroups per CU
  long long blocks = (long long)cus * (per_cu < 2 ? per_cu : 2);
  if (blocks > total) blocks = total;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, p, pyb, pxb, (int)total);
  return hipGetLastError();
}

hipError_t launch_conv3x3_direct(const ConvGemmParams& p, hipStream_t stream) {
  if (p.stride_h == 1) return launch_direct<1, 1>(p, stream);
  if (p.stride_w == 1) return launch_direct<2, 1>(p, stream);
  return launch_direct<2, 2>(p, stream);
}

}  // namespace wsamd
