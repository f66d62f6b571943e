// MFMA burner (tools only): every wavefront issues back-to-back MFMAs on registers for `iters` iterations -- no memory
// traffic, no LDS.  kind 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_f16, 2: v_mfma_f32_16x16x32_f16.
// Built as a shared library with a C entry; tools/burner_vs_fbank.py runs it next to the fbank kernel.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
extern "C" {
__global__ __launch_bounds__(256) void mfma_burner_kernel(int kind, int iters, float* out) {
  f32x16 a0 = {}, a1 = {};
  f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  const float x = 1.0f + threadIdx.x * 1e-3f;
  f16x8 h;
  for (int i = 0; i < 8; ++i) h[i] = (_Float16)(0.5f + 0.01f * i);
  for (int it = 0; it < iters; ++it) {
    if (kind == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 0.5f, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(0.25f, x, a1, 0, 0, 0); }
    } else if (kind == 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a1, 0, 0, 0); }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) { c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, c1, 0, 0, 0); }
    }
  }
  // kinds 3..6: VALU conversion loops (3: fp32 -> binary16 DENORMALS and back, 4: the same on normal values,
  // 5: 16x16x32 f16 MFMAs fed with binary16 denormals, 6: the hi / lo split of the f16x3 back-end on values whose lo
  // parts are denormal)
  float acc = 0.f;
  if (kind >= 3 && kind <= 6) {
    float v = (kind == 4 ? 1.0f : 3.0e-6f) * (1.0f + threadIdx.x * 1e-3f);
    f16x8 d;
    for (int i = 0; i < 8; ++i) d[i] = (_Float16)(2.0e-7f * (i + 1));          // binary16 denormals
    for (int it = 0; it < iters; ++it) {
      if (kind == 5) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(d, h, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, d, c1, 0, 0, 0); }
      } else if (kind == 6) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float xx = 1.0e-3f * (1.0f + 0.37f * r) + acc * 1e-9f;
          const _Float16 hi = (_Float16)xx;
          const _Float16 lo = (_Float16)(xx - (float)hi);
          acc += (float)hi + (float)lo;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const _Float16 q = (_Float16)(v + acc * 1e-12f);
          acc += (float)q;
          asm volatile("" : "+v"(acc));
        }
      }
    }
  }
  // kinds 7..9: 7 = v_cvt_f32_f16_sdwa (sub-dword operand select, as the binary16 kernels unpack high halves),
  // 8 = LDS traffic of the binary16 kernels' shape (ds_write_b64 / ds_read_b128), 9 = v_cvt_pk_f16_f32 + v_pk ops
  if (kind == 7) {
    unsigned u = 0x3c003800u + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float f;
        asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "v"(u));
        acc += f;
        u += 0x00010000u;
      }
    }
  } else if (kind == 8) {
    __shared__ __attribute__((aligned(16))) unsigned long long buf[2048];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        buf[(threadIdx.x + 64 * r + it) & 2047] = (unsigned long long)it * 0x100010001ull + threadIdx.x;
        const f32x4 q = *reinterpret_cast<const f32x4*>(&buf[((threadIdx.x * 2 + 128 * r) & 2046)]);
        acc += q[0] + q[3];
      }
    }
  } else if (kind == 9) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 v2 = {1.0f + threadIdx.x * 1e-3f, 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const f16x2 hh = __builtin_convertvector(v2, f16x2);
        const f32x2 back = __builtin_convertvector(hh, f32x2);
        v2 = v2 * 0.999f + (v2 - back) * 1.5f;
      }
    }
    acc += v2[0] + v2[1];
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[3] + c0[1] + c1[2] + acc;
}
int mfma_burner_launch(int kind, int grid, int iters, float* out, void* stream) {
  hipLaunchKernelGGL(mfma_burner_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, kind, iters, out);
  return (int)hipGetLastError();
}
}
