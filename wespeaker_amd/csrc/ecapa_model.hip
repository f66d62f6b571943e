// ECAPA-TDNN forward scheduled onto the gfx950 kernels (host side).
//
// Architecture and parameter names follow the reference (file:line in wenet-e2e/wespeaker):
//   wespeaker/models/ecapa_tdnn.py:160-234  ECAPA_TDNN (layer1, layer2-4 SE-Res2Blocks d=2/3/4,
//                                           cat -> conv 1x1 -> ReLU -> ASTP -> BN -> Linear [-> bn2])
//   wespeaker/models/ecapa_tdnn.py:237-274  ctor names ECAPA_TDNN[_GLOB]_c{512,1024}
//   wespeaker/models/pooling_layers.py:92-148 ASTP
//
// Layout: every activation is channels-last [utterance*frame][channel] fp32 in HBM; the three
// block outputs are written straight into one [M][3C] buffer (cat-free), the Res2 pass-through
// split is dual-stored by the producing GEMM, eval-mode BN is an epilogue affine after the ReLU
// (the reference order is conv -> ReLU -> BN, so it cannot be folded into the conv weights), the
// last BN + Linear (+ bn2) are folded on the host in float64.
#include <cmath>
#include <cstring>

#include "common.h"
#include "kernels.h"

namespace wsamd {

namespace {

struct ConvW {        // conv/linear with optional bias and post-activation BN affine
  size_t w = 0, b = 0, scale = 0, shift = 0;
  bool has_b = false, has_bn = false;
  int N = 0, Cin = 0, taps = 1, ldw = 0;
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct EcapaModel : Model {
  std::string name;
  int feat_dim, embed_dim;
  int C = 512, w = 64;
  bool glob = false;
  int maxB = 0, maxT = 0;

  WeightArena arena;
  ConvW layer1, blk0[3], res2[3][7], blk2[3], catconv, pool1, pool2, final_lin;
  size_t se_w1[3], se_b1[3], se_w2[3], se_b2[3];
  size_t zeros_off = 0;

  DevBuf ws;                         // activation workspace
  float *out1 = nullptr, *y1 = nullptr, *y2 = nullptr, *y3 = nullptr, *cat = nullptr, *h = nullptr,
        *att = nullptr, *e = nullptr, *se_s = nullptr, *stats = nullptr, *bias_img = nullptr,
        *pooled = nullptr, *partial = nullptr, *feats_ws = nullptr;
  static constexpr int kSplitK = 16;

  EcapaModel(const std::string& n, int fd, int ed) : name(n), feat_dim(fd), embed_dim(ed) {
    glob = n.find("GLOB") != std::string::npos;
    C = n.find("c1024") != std::string::npos ? 1024 : 512;
    w = C / 8;
  }

  bool wants(const std::string& key) const override {
    static const char* prefixes[] = {"layer1.", "layer2.", "layer3.", "layer4.", "conv.",
                                     "pool.", "bn.", "linear.", "bn2."};
    if (key.size() > 20 && key.compare(key.size() - 19, 19, "num_batches_tracked") == 0)
      return false;
    for (auto p : prefixes)
      if (key.compare(0, std::strlen(p), p) == 0) return true;
    return false;
  }

  // ------------------------------------------------------------------ weight ingestion helpers
  const HostTensor* get(const std::map<std::string, HostTensor>& sd, const std::string& key,
                        std::initializer_list<int64_t> shape, int* err) {
    auto it = sd.find(key);
    if (it == sd.end()) {
      set_error("missing tensor '%s' for model %s", key.c_str(), name.c_str());
      *err = WS_ERR_MISSING_TENSOR;
      return nullptr;
    }
    std::vector<int64_t> want(shape);
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

  // eval-mode BatchNorm -> y = x*scale + shift (float64 on the host)
  int bn_affine(const std::map<std::string, HostTensor>& sd, const std::string& prefix, int n,
                std::vector<double>* scale, std::vector<double>* shift, bool affine = true) {
    int err = 0;
    const HostTensor* mean = get(sd, prefix + ".running_mean", {n}, &err);
    if (!mean) return err;
    const HostTensor* var = get(sd, prefix + ".running_var", {n}, &err);
    if (!var) return err;
    const HostTensor *g = nullptr, *bt = nullptr;
    if (affine) {
      g = get(sd, prefix + ".weight", {n}, &err);
      if (!g) return err;
      bt = get(sd, prefix + ".bias", {n}, &err);
      if (!bt) return err;
    }
    scale->resize(n);
    shift->resize(n);
    for (int i = 0; i < n; ++i) {
      double inv = 1.0 / std::sqrt((double)var->data[i] + 1e-5);
      double s = (g ? (double)g->data[i] : 1.0) * inv;
      (*scale)[i] = s;
      (*shift)[i] = (bt ? (double)bt->data[i] : 0.0) - (double)mean->data[i] * s;
    }
    return 0;
  }

  // conv weight (N, Cin, taps) [or (N, Cin)] -> [N][tap*Cin + ci], rows padded to a multiple of 32
  int add_conv(const std::map<std::string, HostTensor>& sd, const std::string& prefix, int N,
               int Cin, int taps, bool conv3d, const std::string& bn_prefix, ConvW* out) {
    int err = 0;
    const HostTensor* wt = conv3d ? get(sd, prefix + ".weight", {N, Cin, taps}, &err)
                                  : get(sd, prefix + ".weight", {N, Cin}, &err);
    if (!wt) return err;
    const HostTensor* bs = get(sd, prefix + ".bias", {N}, &err);
    if (!bs) return err;
    out->N = N; out->Cin = Cin; out->taps = taps;
    out->ldw = round_up(Cin * taps, 32);
    std::vector<float> packed((size_t)N * out->ldw, 0.f);
    for (int n = 0; n < N; ++n)
      for (int ci = 0; ci < Cin; ++ci)
        for (int j = 0; j < taps; ++j)
          packed[(size_t)n * out->ldw + (size_t)j * Cin + ci] =
              wt->data[((size_t)n * Cin + ci) * taps + j];
    out->w = arena.add(packed);
    out->b = arena.add(bs->data);
    out->has_b = true;
    if (!bn_prefix.empty()) {
      std::vector<double> sc, sh;
      if ((err = bn_affine(sd, bn_prefix, N, &sc, &sh))) return err;
      std::vector<float> f(sc.begin(), sc.end()), g(sh.begin(), sh.end());
      out->scale = arena.add(f);
      out->shift = arena.add(g);
      out->has_bn = true;
    }
    return 0;
  }

  int finalize(const std::map<std::string, HostTensor>& sd, int max_batch,
               int max_frames) override {
    int err = 0;
    zeros_off = arena.add(nullptr, 64);
    if ((err = add_conv(sd, "layer1.conv", C, feat_dim, 5, true, "layer1.bn", &layer1))) return err;
    for (int L = 0; L < 3; ++L) {
      std::string p = "layer" + std::to_string(L + 2) + ".se_res2block";
      if ((err = add_conv(sd, p + ".0.conv", C, C, 1, true, p + ".0.bn", &blk0[L]))) return err;
      for (int i = 0; i < 7; ++i)
        if ((err = add_conv(sd, p + ".1.convs." + std::to_string(i), w, w, 3, true,
                            p + ".1.bns." + std::to_string(i), &res2[L][i])))
          return err;
      if ((err = add_conv(sd, p + ".2.conv", C, C, 1, true, p + ".2.bn", &blk2[L]))) return err;
      const HostTensor* t;
      if (!(t = get(sd, p + ".3.linear1.weight", {128, C}, &err))) return err;
      se_w1[L] = arena.add(t->data);
      if (!(t = get(sd, p + ".3.linear1.bias", {128}, &err))) return err;
      se_b1[L] = arena.add(t->data);
      if (!(t = get(sd, p + ".3.linear2.weight", {C, 128}, &err))) return err;
      se_w2[L] = arena.add(t->data);
      if (!(t = get(sd, p + ".3.linear2.bias", {C}, &err))) return err;
      se_b2[L] = arena.add(t->data);
    }
    if ((err = add_conv(sd, "conv", 1536, 3 * C, 1, true, "", &catconv))) return err;
    if ((err = add_conv(sd, "pool.linear1", 128, glob ? 4608 : 1536, 1, true, "", &pool1)))
      return err;
    if ((err = add_conv(sd, "pool.linear2", 1536, 128, 1, true, "", &pool2))) return err;
    // bn (3072) -> linear (E x 3072) [-> bn2]: folded in float64
    {
      std::vector<double> sc, sh;
      if ((err = bn_affine(sd, "bn", 3072, &sc, &sh))) return err;
      const HostTensor* lw = get(sd, "linear.weight", {embed_dim, 3072}, &err);
      if (!lw) return err;
      const HostTensor* lb = get(sd, "linear.bias", {embed_dim}, &err);
      if (!lb) return err;
      std::vector<double> s2(embed_dim, 1.0), t2(embed_dim, 0.0);
      if (sd.count("bn2.running_mean"))
        if ((err = bn_affine(sd, "bn2", embed_dim, &s2, &t2))) return err;
      std::vector<float> W((size_t)embed_dim * 3072), B(embed_dim);
      for (int o = 0; o < embed_dim; ++o) {
        double acc = lb->data[o];
        for (int k = 0; k < 3072; ++k) {
          double wv = lw->data[(size_t)o * 3072 + k];
          acc += wv * sh[k];
          W[(size_t)o * 3072 + k] = (float)(wv * sc[k] * s2[o]);
        }
        B[o] = (float)(acc * s2[o] + t2[o]);
      }
      final_lin.N = embed_dim; final_lin.Cin = 3072; final_lin.taps = 1; final_lin.ldw = 3072;
      final_lin.w = arena.add(W);
      final_lin.b = arena.add(B);
      final_lin.has_b = true;
    }
    hipError_t he = arena.upload();
    if (he != hipSuccess) {
      set_error("weight upload failed: %s", hipGetErrorString(he));
      return WS_ERR_HIP;
    }

    maxB = max_batch; maxT = max_frames;
    const size_t M = (size_t)maxB * maxT;
    size_t total = 0;
    auto take = [&](size_t n) { size_t o = total; total += (n + 63) & ~size_t(63); return o; };
    size_t o_out1 = take(M * C), o_y1 = take(M * C), o_y2 = take(M * C), o_y3 = take(M * C),
           o_cat = take(M * 3 * C), o_h = take(M * 1536), o_att = take(M * 128),
           o_e = take(M * 1536), o_s = take((size_t)maxB * C), o_stats = take((size_t)maxB * 3072),
           o_bias = take((size_t)maxB * 128), o_pool = take((size_t)maxB * 3072),
           o_part = take((size_t)kSplitK * maxB * embed_dim), o_feats = take(M * feat_dim);
    he = ws.alloc(total * sizeof(float));
    if (he != hipSuccess) {
      set_error("workspace allocation of %zu MB failed: %s", total * 4 >> 20, hipGetErrorString(he));
      return WS_ERR_HIP;
    }
    float* base = ws.as<float>();
    out1 = base + o_out1; y1 = base + o_y1; y2 = base + o_y2; y3 = base + o_y3; cat = base + o_cat;
    h = base + o_h; att = base + o_att; e = base + o_e; se_s = base + o_s; stats = base + o_stats;
    bias_img = base + o_bias; pooled = base + o_pool; partial = base + o_part;
    feats_ws = base + o_feats;
    return 0;
  }

  float* feats_workspace() override { return feats_ws; }
  int max_batch() const override { return maxB; }
  int max_frames() const override { return maxT; }

  // ------------------------------------------------------------------------------- launch helper
  ConvGemmParams conv1d(const ConvW& cw, const float* A, int lda, int a_off, float* D, int ldd,
                        int d_off, int B, int T, int dil, int act) const {
    ConvGemmParams p;
    std::memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.a_off = a_off;
    p.W = arena.at(cw.w); p.ldw = cw.ldw;
    p.D = D; p.ldd = ldd; p.d_off = d_off;
    p.M = B * T; p.N = cw.N; p.K = cw.Cin * cw.taps; p.Cin = cw.Cin;
    p.Hin = 1; p.Hout = 1; p.Win = T; p.Wout = T;
    p.stride_h = 1; p.stride_w = 1; p.kh = 1; p.kw = cw.taps; p.dil_h = 1; p.dil_w = dil;
    p.pad_h = 0; p.pad_w = dil * (cw.taps / 2);
    p.bias = cw.has_b ? arena.at(cw.b) : nullptr;
    p.act = act;
    if (cw.has_bn) { p.post_scale = arena.at(cw.scale); p.post_shift = arena.at(cw.shift); }
    p.splitk = 1;
    p.zeros = arena.at(zeros_off);
    return p;
  }

#define WS_LAUNCH(expr)                                                               \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return WS_ERR_HIP;                                                              \
    }                                                                                 \
  } while (0)

  // conv-GEMM launch with optional event timing (class 0/1 = MFMA tile variant, 3 = split-K)
  hipError_t gemm(const ConvGemmParams& p, hipStream_t st) {
    if (prof.enabled) {
      const double flops = 2.0 * p.M * (double)p.N * p.K;
      const double bytes = 4.0 * ((double)p.M * p.Cin * (p.A2 ? 2 : 1) + (double)p.N * p.K +
                                  (double)p.M * p.N);
      prof.begin(p.splitk > 1 ? 3 : (p.N <= 64 ? 1 : 0), flops, bytes, st);
    }
    hipError_t e = launch_conv_gemm(p, st);
    prof.end(st);
    return e;
  }
  template <typename F>
  hipError_t other(double bytes, hipStream_t st, F&& f) {
    prof.begin(2, 0.0, bytes, st);
    hipError_t e = f();
    prof.end(st);
    return e;
  }

  int forward_chunk(const float* feats, int B, int T, float* emb, hipStream_t st) {
    // layer1: Conv1d(F -> C, k5, p2) -> ReLU -> BN
    WS_LAUNCH(gemm(conv1d(layer1, feats, feat_dim, 0, out1, C, 0, B, T, 1, ACT_RELU), st));
    for (int L = 0; L < 3; ++L) {
      const int d = L + 2;
      const float* x = L == 0 ? out1 : cat;
      const int ldx = L == 0 ? C : 3 * C;
      const int x_off = L == 0 ? 0 : (L - 1) * C;
      // 1x1 conv -> ReLU -> BN; the last Res2 split is passed through untouched: dual store
      ConvGemmParams p = conv1d(blk0[L], x, ldx, x_off, y1, C, 0, B, T, 1, ACT_RELU);
      p.D2 = y2; p.ldd2 = C; p.d2_off = 7 * w; p.d2_col0 = 7 * w;
      WS_LAUNCH(gemm(p, st));
      // Res2: sp_i = BN(ReLU(conv_k3_dil(sp_{i-1} + split_i)))
      for (int i = 0; i < 7; ++i) {
        ConvGemmParams q = conv1d(res2[L][i], y1, C, i * w, y2, C, i * w, B, T, d, ACT_RELU);
        if (i >= 1) { q.A2 = y2; q.lda2 = C; q.a2_off = (i - 1) * w; }
        WS_LAUNCH(gemm(q, st));
      }
      WS_LAUNCH(gemm(conv1d(blk2[L], y2, C, 0, y3, C, 0, B, T, 1, ACT_RELU), st));
      const double mc = 4.0 * B * (double)T * C;
      WS_LAUNCH(other(mc, st, [&] {
        return launch_se_pool_fc(y3, C, B, T, C, arena.at(se_w1[L]), arena.at(se_b1[L]),
                                 arena.at(se_w2[L]), arena.at(se_b2[L]), 128, se_s, st);
      }));
      WS_LAUNCH(other(3 * mc, st, [&] {
        return launch_se_scale_residual(x, ldx, x_off, y3, C, se_s, cat, 3 * C, L * C, B, T, C, st);
      }));
    }
    // cat -> Conv1d(3C -> 1536, k1) -> ReLU
    WS_LAUNCH(gemm(conv1d(catconv, cat, 3 * C, 0, h, 1536, 0, B, T, 1, ACT_RELU), st));
    // ASTP
    ConvGemmParams a1 = conv1d(pool1, h, 1536, 0, att, 128, 0, B, T, 1, ACT_TANH);
    a1.K = 1536; a1.Cin = 1536;                      // GLOB: only the first 1536 columns multiply h
    if (glob) {
      WS_LAUNCH(other(4.0 * B * (double)T * 1536, st, [&] {
        return launch_astp_context_bias(h, 1536, B, T, 1536, arena.at(pool1.w), pool1.ldw,
                                        arena.at(pool1.b), 128, stats, bias_img, st);
      }));
      a1.bias = nullptr;
      a1.bias_img = bias_img;
    }
    WS_LAUNCH(gemm(a1, st));
    WS_LAUNCH(gemm(conv1d(pool2, att, 128, 0, e, 1536, 0, B, T, 1, ACT_NONE), st));
    WS_LAUNCH(other(8.0 * B * (double)T * 1536, st, [&] {
      return launch_astp_pool(e, 1536, h, 1536, B, T, 1536, pooled, st);
    }));
    // BN + Linear (+bn2), folded: split-K GEMM over K = 3072
    ConvGemmParams f = conv1d(final_lin, pooled, 3072, 0, emb, embed_dim, 0, B, 1, 1, ACT_NONE);
    f.splitk = kSplitK;
    f.partial = partial;
    WS_LAUNCH(gemm(f, st));
    WS_LAUNCH(launch_splitk_reduce(f, st));
    return 0;
  }

  int forward(const float* feats, int batch, int frames, float* emb, hipStream_t st) override {
    if (frames > maxT || frames < 1) {
      set_error("num_frames %d outside the finalized capacity [1, %d]", frames, maxT);
      return WS_ERR_CAPACITY;
    }
    // chunk so that chunk_B * frames <= maxB * maxT
    const long long cap = (long long)maxB * maxT;
    int chunk = (int)(cap / frames);
    if (chunk > 4 * maxB) chunk = 4 * maxB;      // per-utterance buffers are sized 1x maxB ... keep safe
    if (chunk > maxB) chunk = maxB;
    for (int b0 = 0; b0 < batch; b0 += chunk) {
      const int nb = batch - b0 < chunk ? batch - b0 : chunk;
      int r = forward_chunk(feats + (size_t)b0 * frames * feat_dim, nb, frames,
                            emb + (size_t)b0 * embed_dim, st);
      if (r) return r;
    }
    return 0;
  }

  double flops(int batch, int T) const override {
    double macs = 0;
    macs += (double)feat_dim * 5 * C;                       // layer1
    macs += 3.0 * (2.0 * C * C + 7.0 * w * w * 3);          // blocks (per frame)
    macs += 3.0 * C * 1536;                                 // cat conv
    macs += 1536.0 * 128 * 2;                               // ASTP linear1 (h part) + linear2
    double per_utt = macs * T;
    per_utt += 3.0 * (2.0 * C * 128);                       // SE FCs
    if (glob) per_utt += 128.0 * 3072;                      // ASTP context columns (folded to a bias)
    per_utt += 3072.0 * embed_dim;                          // final linear
    return 2.0 * per_utt * batch;
  }
};

}  // namespace

Model* make_ecapa(const std::string& model_name, int feat_dim, int embed_dim) {
  static const char* names[] = {"ECAPA_TDNN_c512", "ECAPA_TDNN_GLOB_c512", "ECAPA_TDNN_c1024",
                                "ECAPA_TDNN_GLOB_c1024"};
  for (auto n : names)
    if (model_name == n) return new EcapaModel(model_name, feat_dim, embed_dim);
  return nullptr;
}

}  // namespace wsamd
