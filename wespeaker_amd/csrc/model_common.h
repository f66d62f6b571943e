// Shared host-side machinery of the model engines (ECAPA-TDNN, ResNet, CAM++): weight ingestion by
// the reference's state_dict names, BatchNorm folding, conv-GEMM parameter construction and the
// event-profiled launch wrappers.
#pragma once
#include <cmath>
#include <cstring>
#include <functional>

#include "common.h"
#include "kernels.h"

namespace wsamd {

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct ConvW {        // one conv / linear layer resident on the device
  size_t w = 0, b = 0, scale = 0, shift = 0;
  size_t wh = 0, wl = 0;    // hi / lo binary16 planes of w (same [N][ldw] layout), arena float offsets
  bool has_b = false, has_post = false;
  int N = 0, Cin = 0, kh = 1, kw = 1, ldw = 0;
  int taps() const { return kh * kw; }
};

#define WS_LAUNCH(expr)                                                                            \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__);    \
      return WS_ERR_HIP;                                                                           \
    }                                                                                              \
  } while (0)

struct ModelBase : Model {
  std::string name;
  int feat_dim = 0, embed_dim = 0;
  int maxB = 0, maxT = 0;
  WeightArena arena;
  size_t zeros_off = 0;
  DevBuf ws;
  float* feats_ws = nullptr;
  typedef std::map<std::string, HostTensor> SD;

  ModelBase(const std::string& n, int fd, int ed) : name(n), feat_dim(fd), embed_dim(ed) {
    zeros_off = arena.add(nullptr, 64);
  }
  int set_precision(int mode) override {
    if (mode < 0 || mode > 2) return WS_ERR_INVALID_ARG;
    gemm_precision = mode;
    return WS_OK;
  }
  // ---- range guard of the binary16 back-ends (activations beyond 65504 become inf and surface as
  // non-finite embeddings): a host-mapped counter bumped by a tiny kernel after every forward
  int* nonfinite_host = nullptr;
  int* nonfinite_dev = nullptr;
  ~ModelBase() override {
    if (nonfinite_host) (void)hipHostFree(nonfinite_host);
    if (lens_pinned) (void)hipHostFree(lens_pinned);
    if (lens_copied) (void)hipEventDestroy(lens_copied);
  }
  int range_guard(const float* emb, int batch, hipStream_t st) {
    if (gemm_precision == 0 || !nonfinite_dev) return 0;
    WS_LAUNCH(launch_count_nonfinite(emb, (long long)batch * embed_dim, nonfinite_dev, st));
    return 0;
  }
  int take_nonfinite() override {
    if (!nonfinite_host) return 0;
    // one atomic exchange: a forward still running on another stream may atomicAdd_system into the counter at
    // any moment; read-then-clear would drop what lands in between.  (ws_engine_check_range synchronises only the
    // stream it is given: the count covers what that stream has finished.)
    return __atomic_exchange_n(reinterpret_cast<int*>(nonfinite_host), 0, __ATOMIC_ACQ_REL);
  }
  // ---- ragged batches: per-utterance valid lengths at the four time-stride levels of the 2-D models
  // (level l+1 = (level l - 1) / 2 + 1, the output width of a k3/p1/s2 -- or k1/s2, k5/p2/s2 -- conv)
  enum { kLenLevels = 4 };
  const int* cur_lens[kLenLevels] = {nullptr, nullptr, nullptr, nullptr};   // of the chunk in flight
  DevBuf lens_dev;                 // [kLenLevels][lens_cap] int32
  int* lens_pinned = nullptr;      // host staging of the same
  size_t lens_cap = 0;
  hipEvent_t lens_copied = nullptr;
  bool ragged() const { return cur_lens[0] != nullptr; }

  virtual int forward_chunk(const float* feats, int B, int T, float* emb, hipStream_t st) = 0;
  virtual int min_frames() const = 0;

  const int* upload_lens(const int32_t* lens_host, int batch, int frames, hipStream_t st) override {
    if ((size_t)batch > lens_cap) {
      if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
      if (lens_pinned) (void)hipHostFree(lens_pinned);
      lens_pinned = nullptr;
      const size_t cap = (size_t)batch + batch / 2 + 64;
      if (lens_dev.alloc(cap * kLenLevels * sizeof(int)) != hipSuccess ||
          hipHostMalloc(reinterpret_cast<void**>(&lens_pinned), cap * kLenLevels * sizeof(int), 0) != hipSuccess) {
        set_error("ragged batch: length table allocation failed");
        return nullptr;
      }
      lens_cap = cap;
      if (!lens_copied && hipEventCreateWithFlags(&lens_copied, hipEventDisableTiming) != hipSuccess) return nullptr;
    } else if (lens_copied) {
      (void)hipEventSynchronize(lens_copied);          // the previous upload has left the staging buffer
    }
    for (int b = 0; b < batch; ++b) {
      int L = lens_host[b];
      if (L < min_frames() || L > frames) {
        set_error("ragged batch: utterance %d has %d frames, valid range is [%d, %d]", b, L, min_frames(), frames);
        return nullptr;
      }
      for (int l = 0; l < kLenLevels; ++l) {
        lens_pinned[(size_t)l * batch + b] = L;
        L = (L - 1) / 2 + 1;
      }
    }
    if (hipMemcpyAsync(lens_dev.ptr, lens_pinned, (size_t)batch * kLenLevels * sizeof(int),
                       hipMemcpyHostToDevice, st) != hipSuccess ||
        hipEventRecord(lens_copied, st) != hipSuccess) {
      set_error("ragged batch: length table upload failed");
      return nullptr;
    }
    return lens_dev.as<int>();
  }

  int check_frames(int frames) {
    if (frames > maxT || frames < min_frames()) {
      set_error("num_frames %d outside the finalized capacity [%d, %d]", frames, min_frames(), maxT);
      return WS_ERR_CAPACITY;
    }
    return 0;
  }

  int forward(const float* feats, int batch, int frames, float* emb, hipStream_t st) override {
    if (int r = check_frames(frames)) return r;
    for (int l = 0; l < kLenLevels; ++l) cur_lens[l] = nullptr;
    for (int b0 = 0; b0 < batch; b0 += maxB) {
      const int nb = batch - b0 < maxB ? batch - b0 : maxB;
      int r = forward_chunk(feats + (size_t)b0 * frames * feat_dim, nb, frames,
                            emb + (size_t)b0 * embed_dim, st);
      if (r) return r;
    }
    return range_guard(emb, batch, st);
  }

  int forward_chunk_ragged(const float* feats, int nb, int frames, const int* dev, int batch, int b0,
                           float* emb, hipStream_t st) override {
    for (int l = 0; l < kLenLevels; ++l) cur_lens[l] = dev + (size_t)l * batch + b0;
    const int rc = forward_chunk(feats, nb, frames, emb, st);
    for (int l = 0; l < kLenLevels; ++l) cur_lens[l] = nullptr;
    return rc;
  }
  int finish_forward(const float* emb, int batch, hipStream_t st) override { return range_guard(emb, batch, st); }

  int forward_ragged(const float* feats, int batch, int frames, const int32_t* lens_host, float* emb,
                     hipStream_t st, int cmvn_mode = 0) override {
    if (int r = check_frames(frames)) return r;
    const int* dev = upload_lens(lens_host, batch, frames, st);
    if (!dev) return WS_ERR_INVALID_ARG;
    for (int b0 = 0; b0 < batch; b0 += maxB) {
      const int nb = batch - b0 < maxB ? batch - b0 : maxB;
      // zero the padding rows: they are what the first convolution's taps must see
      hipError_t he = launch_copy_rows_masked(feats + (size_t)b0 * frames * feat_dim, feats_ws, nb, frames,
                                              feat_dim, dev + b0, st);
      if (he != hipSuccess) {
        set_error("masked feature copy failed: %s", hipGetErrorString(he));
        return WS_ERR_HIP;
      }
      if (cmvn_mode) {
        he = launch_cmn(feats_ws, nb, frames, feat_dim, st, dev + b0, cmvn_mode);
        if (he != hipSuccess) {
          set_error("feature CMVN failed: %s", hipGetErrorString(he));
          return WS_ERR_HIP;
        }
      }
      int r = forward_chunk_ragged(feats_ws, nb, frames, dev, batch, b0, emb + (size_t)b0 * embed_dim, st);
      if (r) return r;
    }
    return range_guard(emb, batch, st);
  }
  float* feats_workspace() override { return feats_ws; }
  int max_batch() const override { return maxB; }
  int max_frames() const override { return maxT; }

  // w = hi + lo with hi = half(w), lo = half(w - hi): the weight operand of the 3-pass f16 MFMA
  // path (hi*hi + hi*lo + lo*hi reproduces the fp32 product to ~2^-21).
  void add_split(ConvW* cw, const std::vector<float>& packed) {
    const size_t n = packed.size();            // multiple of 32
    std::vector<uint16_t> hi(n), lo(n);
    for (size_t i = 0; i < n; ++i) {
      hi[i] = float_to_half_bits(packed[i]);
      lo[i] = float_to_half_bits(packed[i] - half_bits_to_float(hi[i]));
    }
    cw->wh = arena.add(nullptr, n / 2);
    std::memcpy(arena.host.data() + cw->wh, hi.data(), n * 2);
    cw->wl = arena.add(nullptr, n / 2);
    std::memcpy(arena.host.data() + cw->wl, lo.data(), n * 2);
  }
  int gemm_precision = 0;     // 0: exact fp32 MFMA, 1: 3-pass split-f16 MFMA, 2: 1-pass f16 (fp32 accumulate)

  // ------------------------------------------------------------------------- tensor lookup
  const HostTensor* get(const SD& sd, const std::string& key, std::vector<int64_t> want, int* err) {
    auto it = sd.find(key);
    if (it == sd.end()) {
      set_error("missing tensor '%s' for model %s", key.c_str(), name.c_str());
      *err = WS_ERR_MISSING_TENSOR;
      return nullptr;
    }
    if (it->second.shape != want) {
      std::string got, exp;
      for (auto s : it->second.shape) got += std::to_string(s) + ",";
      for (auto s : want) exp += std::to_string(s) + ",";
      set_error("tensor '%s' has shape (%s) but model %s expects (%s)", key.c_str(), got.c_str(),
                name.c_str(), exp.c_str());
      *err = WS_ERR_SHAPE;
      return nullptr;
    }
    return &it->second;
  }

  // eval-mode BatchNorm (eps 1e-5) -> y = x*scale + shift, in float64
  int bn_affine(const SD& sd, const std::string& prefix, int n, std::vector<double>* scale,
                std::vector<double>* shift, bool affine = true) {
    int err = 0;
    const HostTensor* mean = get(sd, prefix + ".running_mean", {n}, &err);
    if (!mean) return err;
    const HostTensor* var = get(sd, prefix + ".running_var", {n}, &err);
    if (!var) return err;
    const HostTensor *g = nullptr, *bt = nullptr;
    if (affine) {
      if (!(g = get(sd, prefix + ".weight", {n}, &err))) return err;
      if (!(bt = get(sd, prefix + ".bias", {n}, &err))) return err;
    }
    scale->resize(n);
    shift->resize(n);
    for (int i = 0; i < n; ++i) {
      const double inv = 1.0 / std::sqrt((double)var->data[i] + 1e-5);
      const double s = (g ? (double)g->data[i] : 1.0) * inv;
      (*scale)[i] = s;
      (*shift)[i] = (bt ? (double)bt->data[i] : 0.0) - (double)mean->data[i] * s;
    }
    return 0;
  }

  int add_vec(const SD& sd, const std::string& key, int n, size_t* off) {
    int err = 0;
    const HostTensor* t = get(sd, key, {n}, &err);
    if (!t) return err;
    *off = arena.add(t->data);
    return 0;
  }

  // per-channel BN-ReLU that precedes a conv (CAM++ pre-activation): scale/shift vectors
  int add_bn_vectors(const SD& sd, const std::string& prefix, int n, size_t* scale, size_t* shift,
                     bool affine = true) {
    std::vector<double> sc, sh;
    int err = bn_affine(sd, prefix, n, &sc, &sh, affine);
    if (err) return err;
    std::vector<float> f(sc.begin(), sc.end()), g(sh.begin(), sh.end());
    *scale = arena.add(f);
    *shift = arena.add(g);
    return 0;
  }

  // Generic conv / linear packer.
  //   wshape : expected shape of `<prefix>.weight`
  //   src(n, tap, ci) -> flat index into the source weight
  //   bias   : `<prefix>.bias` is required iff has_bias
  //   fold_bn: BN applied directly to the conv output (conv -> BN): folded into W and bias
  //   post_bn: BN applied after the activation (ECAPA conv -> ReLU -> BN): epilogue affine
  int pack_conv(const SD& sd, const std::string& prefix, std::vector<int64_t> wshape, int N, int Cin,
                int kh, int kw, const std::function<size_t(int, int, int)>& src, bool has_bias,
                const std::string& fold_bn, const std::string& post_bn, ConvW* out,
                bool fold_affine = true) {
    int err = 0;
    const HostTensor* wt = get(sd, prefix + ".weight", wshape, &err);
    if (!wt) return err;
    const HostTensor* bs = nullptr;
    if (has_bias && !(bs = get(sd, prefix + ".bias", {N}, &err))) return err;
    std::vector<double> fsc(N, 1.0), fsh(N, 0.0);
    if (!fold_bn.empty() && (err = bn_affine(sd, fold_bn, N, &fsc, &fsh, fold_affine))) return err;
    out->N = N; out->Cin = Cin; out->kh = kh; out->kw = kw;
    const int taps = kh * kw;
    out->ldw = round_up(Cin * taps, 64);      // 64: the K-tile of the f16 kernels (zero padded)
    std::vector<float> packed((size_t)N * out->ldw, 0.f);
    for (int n = 0; n < N; ++n)
      for (int tp = 0; tp < taps; ++tp)
        for (int ci = 0; ci < Cin; ++ci)
          packed[(size_t)n * out->ldw + (size_t)tp * Cin + ci] =
              (float)((double)wt->data[src(n, tp, ci)] * fsc[n]);
    out->w = arena.add(packed);
    add_split(out, packed);
    if (has_bias || !fold_bn.empty()) {
      std::vector<float> b(N);
      for (int n = 0; n < N; ++n) b[n] = (float)((bs ? (double)bs->data[n] : 0.0) * fsc[n] + fsh[n]);
      out->b = arena.add(b);
      out->has_b = true;
    }
    if (!post_bn.empty()) {
      std::vector<double> sc, sh;
      if ((err = bn_affine(sd, post_bn, N, &sc, &sh))) return err;
      std::vector<float> f(sc.begin(), sc.end()), g(sh.begin(), sh.end());
      out->scale = arena.add(f);
      out->shift = arena.add(g);
      out->has_post = true;
    }
    return 0;
  }

  // Conv1d weight (N, Cin, k) -> [n][tap*Cin + ci]
  int pack_conv1d(const SD& sd, const std::string& prefix, int N, int Cin, int k, bool has_bias,
                  const std::string& fold_bn, const std::string& post_bn, ConvW* out) {
    return pack_conv(sd, prefix, {N, Cin, k}, N, Cin, 1, k,
                     [=](int n, int tp, int ci) { return ((size_t)n * Cin + ci) * k + tp; }, has_bias,
                     fold_bn, post_bn, out);
  }
  // Conv2d weight (N, Cin, kh, kw) -> [n][(ty*kw + tx)*Cin + ci]
  int pack_conv2d(const SD& sd, const std::string& prefix, int N, int Cin, int kh, int kw,
                  const std::string& fold_bn, ConvW* out) {
    return pack_conv(sd, prefix, {N, Cin, kh, kw}, N, Cin, kh, kw,
                     [=](int n, int tp, int ci) { return ((size_t)n * Cin + ci) * kh * kw + tp; },
                     false, fold_bn, "", out);
  }
  // Linear weight (N, Cin)
  int pack_linear(const SD& sd, const std::string& prefix, int N, int Cin, bool has_bias, ConvW* out) {
    return pack_conv(sd, prefix, {N, Cin}, N, Cin, 1, 1,
                     [=](int n, int, int ci) { return (size_t)n * Cin + ci; }, has_bias, "", "", out);
  }

  int upload_weights() {
    hipError_t he = hipHostMalloc(reinterpret_cast<void**>(&nonfinite_host), sizeof(int), hipHostMallocMapped);
    if (he == hipSuccess) {
      *nonfinite_host = 0;
      he = hipHostGetDevicePointer(reinterpret_cast<void**>(&nonfinite_dev), nonfinite_host, 0);
    }
    if (he != hipSuccess) {
      set_error("range-guard counter allocation failed: %s", hipGetErrorString(he));
      return WS_ERR_HIP;
    }
    he = arena.upload();
    if (he != hipSuccess) {
      set_error("weight upload failed: %s", hipGetErrorString(he));
      return WS_ERR_HIP;
    }
    return 0;
  }
  // (re)allocates the workspace; earlier launches may still be using the old one
  int alloc_workspace(size_t ws_floats) {
    hipError_t he = hipDeviceSynchronize();
    if (he == hipSuccess) he = ws.alloc(ws_floats * sizeof(float));
    if (he != hipSuccess) {
      set_error("workspace allocation of %zu MB failed: %s", (ws_floats * 4) >> 20,
                hipGetErrorString(he));
      return WS_ERR_HIP;
    }
    return 0;
  }

  // --------------------------------------------------------------------- conv-GEMM parameters
  // generic 2-D convolution over images [B][Hin][Win][lda]
  ConvGemmParams conv2d(const ConvW& cw, const float* A, int lda, int a_off, float* D, int ldd,
                        int d_off, int B, int Hin, int Win, int stride_h, int stride_w, int dil_h,
                        int dil_w, int pad_h, int pad_w, int act) const {
    ConvGemmParams p;
    std::memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.a_off = a_off;
    p.W = arena.at(cw.w); p.ldw = cw.ldw;
    p.Wh = reinterpret_cast<const uint16_t*>(arena.at(cw.wh));
    p.Wl = reinterpret_cast<const uint16_t*>(arena.at(cw.wl));
    p.prec = gemm_precision;
    p.D = D; p.ldd = ldd; p.d_off = d_off;
    p.Hin = Hin; p.Win = Win;
    p.Hout = (Hin + 2 * pad_h - dil_h * (cw.kh - 1) - 1) / stride_h + 1;
    p.Wout = (Win + 2 * pad_w - dil_w * (cw.kw - 1) - 1) / stride_w + 1;
    p.M = B * p.Hout * p.Wout; p.N = cw.N; p.K = cw.Cin * cw.taps(); p.Cin = cw.Cin;
    p.stride_h = stride_h; p.stride_w = stride_w; p.kh = cw.kh; p.kw = cw.kw;
    p.dil_h = dil_h; p.dil_w = dil_w; p.pad_h = pad_h; p.pad_w = pad_w;
    p.bias = cw.has_b ? arena.at(cw.b) : nullptr;
    p.act = act;
    if (cw.has_post) { p.post_scale = arena.at(cw.scale); p.post_shift = arena.at(cw.shift); }
    p.splitk = 1;
    p.zeros = arena.at(zeros_off);
    return p;
  }
  // Conv1d over time ("same" padding for odd kernels), rows = B*T
  ConvGemmParams conv1d(const ConvW& cw, const float* A, int lda, int a_off, float* D, int ldd,
                        int d_off, int B, int T, int dil, int act) const {
    return conv2d(cw, A, lda, a_off, D, ldd, d_off, B, 1, T, 1, 1, 1, dil, 0, dil * (cw.kw / 2), act);
  }

  // ------------------------------------------------------------------------- profiled launches
  // k_alg: the algorithmic contraction length when p.K carries zero padding (im2col images)
  hipError_t gemm(const ConvGemmParams& p, hipStream_t st, int k_alg = 0) {
    if (prof.enabled) {
      const double flops = 2.0 * p.M * (double)p.N * (k_alg ? k_alg : p.K);
      // algorithmic bytes: A once, W once, every stored copy of D once (binary16 tensors count 2 B)
      const double a_b = p.prec == 2 && p.A16 ? 2.0 : 4.0 * (p.A2 ? 2 : 1);
      const double w_b = p.prec == 0 ? 4.0 : (p.prec == 1 ? 4.0 : 2.0);
      const double d_b = (p.D ? 4.0 : 0.0) + (p.D16 ? 2.0 : 0.0);
      const double bytes = a_b * p.M * (double)p.Cin + w_b * p.N * (double)p.K + d_b * p.M * (double)p.N;
      prof.begin(p.splitk > 1 ? 3 : (p.N <= 64 ? 1 : 0), flops, bytes, st);
    }
    hipError_t e = launch_conv_gemm(p, st);
    prof.end(st);
    return e;
  }
  // split-K GEMM + reduce (M small, K large: the embedding layers)
  hipError_t gemm_splitk(ConvGemmParams p, float* partial, int splitk, hipStream_t st) {
    if (small_m_gemm_f32_applies(p)) {          // fp32, plain epilogue: one launch, no partial sums through HBM
      if (prof.enabled) prof.begin(3, 2.0 * p.M * (double)p.N * p.K, 4.0 * (p.M * (double)p.K + p.N * (double)p.K), st);
      hipError_t e = launch_small_m_gemm_f32(p, st);
      prof.end(st);
      return e;
    }
    p.splitk = splitk;
    p.partial = partial;
    hipError_t e = gemm(p, st);
    if (e != hipSuccess) return e;
    return launch_splitk_reduce(p, st);
  }
  template <typename F>
  hipError_t other(double bytes, hipStream_t st, F&& f) {
    prof.begin(2, 0.0, bytes, st);
    hipError_t e = f();
    prof.end(st);
    return e;
  }
};

}  // namespace wsamd
