// Host-side helpers shared by the C-ABI implementation files.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/wespeaker_amd.h"

namespace wsamd {

void set_error(const char* fmt, ...);
const char* get_error();

#define WS_HIP_CHECK(expr)                                                                  \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      ::wsamd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                         __LINE__);                                                         \
      return WS_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

// RAII device allocation
struct DevBuf {
  void* ptr = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
  hipError_t alloc(size_t n) {
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&ptr, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <typename T>
  T* as() const { return reinterpret_cast<T*>(ptr); }
};

// Many small weight tensors packed into one device allocation (16-byte aligned slices).
struct WeightArena {
  std::vector<float> host;
  DevBuf dev;
  // returns the float offset of the slice
  size_t add(const float* data, size_t n) {
    size_t off = (host.size() + 3) & ~size_t(3);
    host.resize(off + n);
    if (data) std::copy(data, data + n, host.begin() + off);
    return off;
  }
  size_t add(const std::vector<float>& v) { return add(v.data(), v.size()); }
  hipError_t upload() {
    hipError_t e = dev.alloc(host.size() * sizeof(float) + 64);
    if (e != hipSuccess) return e;
    return hipMemcpy(dev.ptr, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  const float* at(size_t off) const { return dev.as<float>() + off; }
};

// IEEE binary16 conversion, round-to-nearest-even, subnormals preserved (host side; used to
// pre-split weights into hi + lo half planes for the 3-pass f16 MFMA path).
inline uint16_t float_to_half_bits(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7FFFFFFFu;
  if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? 0x200u : 0u));
  if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);            // overflow -> inf
  if (x < 0x38800000u) {                                               // subnormal half (or zero)
    if (x < 0x33000000u) return (uint16_t)sign;
    const int shift = 126 - (int)(x >> 23);                            // 14..24
    uint32_t mant = (x & 0x7FFFFFu) | 0x800000u;
    uint32_t h = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1))) ++h;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((x - 0x38000000u) >> 13);
  const uint32_t rem = x & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
  return (uint16_t)(sign | h);
}
inline float half_bits_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, x;
  if (e == 0) {
    if (m == 0) x = sign;
    else {
      int s = 0;
      while (!(m & 0x400u)) { m <<= 1; ++s; }
      x = sign | ((uint32_t)(113 - s) << 23) | ((m & 0x3FFu) << 13);
    }
  } else if (e == 31) x = sign | 0x7F800000u | (m << 13);
  else x = sign | ((e + 112) << 23) | (m << 13);
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

// Per-kernel-class timing with HIP events recorded on the launch stream (bench.py's roofline leg:
// achieved FLOP/s of the dominant kernel = sum of algorithmic flops / sum of event-to-event ms).
struct KernelProfiler {
  enum { kClasses = 4 };      // 0: conv_gemm 128x128 tile, 1: conv_gemm 128x64 tile, 2: other, 3: split-K path
  bool enabled = false;
  unsigned mask = 0xF;          // which classes are recorded (events between kernels cost a few %)
  bool open = false;
  struct Rec { int cls; double flops, bytes; hipEvent_t a, b; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  ~KernelProfiler() { for (auto e : pool) (void)hipEventDestroy(e); }
  hipEvent_t take() {
    if (used == pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[used++];
  }
  // call before / after a launch
  void begin(int cls, double flops, double bytes, hipStream_t st) {
    open = false;
    if (!enabled || !(mask & (1u << cls))) return;
    Rec r{cls, flops, bytes, take(), take()};
    if (!r.a || !r.b) return;
    (void)hipEventRecord(r.a, st);
    recs.push_back(r);
    open = true;
  }
  void end(hipStream_t st) {
    if (!open) return;
    (void)hipEventRecord(recs.back().b, st);
    open = false;
  }
  // synchronises; sums per class; clears
  void read(double* ms, double* flops, double* bytes, int* launches) {
    for (int c = 0; c < kClasses; ++c) { ms[c] = 0; flops[c] = 0; bytes[c] = 0; launches[c] = 0; }
    for (auto& r : recs) {
      (void)hipEventSynchronize(r.b);
      float t = 0.f;
      if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
        ms[r.cls] += t; flops[r.cls] += r.flops; bytes[r.cls] += r.bytes; launches[r.cls] += 1;
      }
    }
    recs.clear();
    used = 0;
  }
};

// Model-family interface behind ws_engine (native twin of runtime/core/speaker/speaker_model.h:25-32)
struct Model {
  KernelProfiler prof;
  virtual ~Model() {}
  // returns true if `key` belongs to this architecture
  virtual bool wants(const std::string& key) const = 0;
  virtual int finalize(const std::map<std::string, HostTensor>& sd, int max_batch,
                       int max_frames) = 0;
  // re-size the workspace of a finalized model (synchronises the device)
  virtual int reserve(int max_batch, int max_frames) = 0;
  virtual int forward(const float* feats, int batch, int frames, float* emb,
                      hipStream_t stream) = 0;
  // Ragged batch: utterance b has lens_host[b] <= frames valid rows (HOST array); the rows beyond are
  // padding (their content is ignored).
  // cmvn_mode (bit 0 norm_mean, bit 1 norm_var; 0 = the features are already normalised): apply_cmvn over every
  // utterance's own frames, applied to the masked copy in the workspace (the caller's tensor is not modified)
  virtual int forward_ragged(const float* feats, int batch, int frames, const int32_t* lens_host,
                             float* emb, hipStream_t stream, int cmvn_mode = 0) = 0;
  // Building blocks of the fused ragged extract (c_api.hip: fbank -> CMN -> forward per chunk):
  // upload_lens validates and copies the table, returns its device address (level-major, `batch` entries
  // per time-stride level; level 0 first) or null; forward_chunk_ragged runs rows [b0, b0 + nb) whose
  // padding rows are already zero; finish_forward closes the call (range guard).
  virtual const int* upload_lens(const int32_t* lens_host, int batch, int frames, hipStream_t stream) = 0;
  virtual int forward_chunk_ragged(const float* feats, int nb, int frames, const int* lens_dev, int batch,
                                   int b0, float* emb, hipStream_t stream) = 0;
  virtual int finish_forward(const float* emb, int batch, hipStream_t stream) = 0;
  virtual double flops(int batch, int frames) const = 0;
  // non-finite embedding values produced by the binary16 back-ends since the last call (reads and
  // clears the host-mapped counter; the caller has synchronised the stream)
  virtual int take_nonfinite() = 0;
  // Engine-internal invariants that a kernel bug could break silently (ws_engine_check_range calls it behind its
  // stream synchronisation): 0, or a WS_ERR_* with the message set.  ResNet: the zero pads behind its activation buffers.
  virtual int check_invariants() { return 0; }
  virtual int set_precision(int mode) = 0;   // 0 exact fp32 MFMA, 1 split-f16 x3 MFMA
  virtual float* feats_workspace() = 0;      // (max_batch, max_frames, feat_dim) floats
  virtual int max_batch() const = 0;
  virtual int max_frames() const = 0;
};

Model* make_ecapa(const std::string& model_name, int feat_dim, int embed_dim);
Model* make_resnet(const std::string& model_name, int feat_dim, int embed_dim);
Model* make_campplus(const std::string& model_name, int feat_dim, int embed_dim);

}  // namespace wsamd

struct ws_engine {
  std::string model_name;
  int feat_dim = 0, embed_dim = 0, device = 0;
  bool finalized = false;
  std::map<std::string, wsamd::HostTensor> sd;
  wsamd::Model* model = nullptr;
  wsamd::DevBuf chunk_scratch;   // ws_extract_chunked: utterance feats | chunk tensor | chunk embeddings
};
