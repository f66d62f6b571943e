// ResNet speaker-embedding forward (ResNet18/34/50/101/152/221/293) scheduled onto the gfx950
// kernels.
//
// Reference (file:line in wenet-e2e/wespeaker):
//   wespeaker/models/resnet.py:110-204  ResNet (conv1+bn1+relu, 4 stages, TSTP, seg_1 [, seg_bn_1, seg_2])
//   wespeaker/models/resnet.py:35-69    BasicBlock;  :72-107 Bottleneck (stride on the 3x3)
//   wespeaker/models/resnet.py:207-260  ctor names
//   wespeaker/models/pooling_layers.py:78-85 TSTP
//
// Layout: the (B, T, F) feature tensor is the single-channel (F x T) image; every activation is
// channels-last [b][f][t][c] fp32, so each Conv2d is one implicit GEMM with K = kh*kw*Cin contiguous.
// conv -> BN has no non-linearity in between, so every BatchNorm2d is folded into its conv's
// weights + bias on the host (float64); the residual add and ReLU live in the GEMM epilogue.
#include "model_common.h"

namespace wsamd {

namespace {

struct Layout { const char* name; bool bottleneck; int blocks[4]; };
const Layout kLayouts[] = {
    {"ResNet18", false, {2, 2, 2, 2}},   {"ResNet34", false, {3, 4, 6, 3}},
    {"ResNet50", true, {3, 4, 6, 3}},    {"ResNet101", true, {3, 4, 23, 3}},
    {"ResNet152", true, {3, 8, 36, 3}},  {"ResNet221", true, {6, 16, 48, 3}},
    {"ResNet293", true, {10, 20, 64, 3}},
};

struct Block {
  ConvW c1, c2, c3, sc;
  bool has_sc = false;
  int stride = 1, in_planes = 0, planes = 0;
};

struct ResNetModel : ModelBase {
  Layout lay;
  int exp = 1;
  size_t stem_w = 0, stem_b = 0;
  std::vector<Block> blocks;
  ConvW seg1, seg2;
  bool two_emb = false;
  size_t seg_bn_scale = 0, seg_bn_shift = 0;
  float* buf[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t act_floats = 0;                     // floats per activation buffer (512 zero floats follow each)
  float *pooled = nullptr, *partial = nullptr, *emb_a = nullptr;
  int stats_dim = 0;
  static constexpr int kSplitK = 16;

  ResNetModel(const Layout& l, int fd, int ed) : ModelBase(l.name, fd, ed), lay(l) {
    exp = l.bottleneck ? 4 : 1;
  }

  bool wants(const std::string& key) const override {
    if (key.size() > 20 && key.compare(key.size() - 19, 19, "num_batches_tracked") == 0)
      return false;
    static const char* prefixes[] = {"conv1.", "bn1.", "layer1.", "layer2.", "layer3.", "layer4.",
                                     "seg_1.", "seg_bn_1.", "seg_2."};
    for (auto p : prefixes)
      if (key.compare(0, std::strlen(p), p) == 0) return true;
    return false;
  }

  int finalize(const SD& sd, int max_batch, int max_frames) override {
    int err = 0;
    const int m = 32;
    if (feat_dim % 8) {
      set_error("ResNet needs feat_dim %% 8 == 0, got %d", feat_dim);
      return WS_ERR_INVALID_ARG;
    }
    {   // stem: Conv2d(1, 32, 3x3) + bn1 folded -> [C][9] + [C]
      const HostTensor* wt = get(sd, "conv1.weight", {m, 1, 3, 3}, &err);
      if (!wt) return err;
      std::vector<double> sc, sh;
      if ((err = bn_affine(sd, "bn1", m, &sc, &sh))) return err;
      std::vector<float> wf(m * 9), bf(m);
      for (int c = 0; c < m; ++c) {
        for (int k = 0; k < 9; ++k) wf[c * 9 + k] = (float)(wt->data[c * 9 + k] * sc[c]);
        bf[c] = (float)sh[c];
      }
      stem_w = arena.add(wf);
      stem_b = arena.add(bf);
    }
    int in_planes = m;
    for (int s = 0; s < 4; ++s) {
      const int planes = m << s;
      for (int b = 0; b < lay.blocks[s]; ++b) {
        Block blk;
        blk.stride = (b == 0 && s > 0) ? 2 : 1;
        blk.in_planes = in_planes; blk.planes = planes;
        const std::string p = "layer" + std::to_string(s + 1) + "." + std::to_string(b);
        if (lay.bottleneck) {
          if ((err = pack_conv2d(sd, p + ".conv1", planes, in_planes, 1, 1, p + ".bn1", &blk.c1))) return err;
          if ((err = pack_conv2d(sd, p + ".conv2", planes, planes, 3, 3, p + ".bn2", &blk.c2))) return err;
          if ((err = pack_conv2d(sd, p + ".conv3", planes * 4, planes, 1, 1, p + ".bn3", &blk.c3))) return err;
        } else {
          if ((err = pack_conv2d(sd, p + ".conv1", planes, in_planes, 3, 3, p + ".bn1", &blk.c1))) return err;
          if ((err = pack_conv2d(sd, p + ".conv2", planes, planes, 3, 3, p + ".bn2", &blk.c2))) return err;
        }
        if (blk.stride != 1 || in_planes != planes * exp) {
          blk.has_sc = true;
          if ((err = pack_conv2d(sd, p + ".shortcut.0", planes * exp, in_planes, 1, 1,
                                 p + ".shortcut.1", &blk.sc)))
            return err;
        }
        in_planes = planes * exp;
        blocks.push_back(blk);
      }
    }
    stats_dim = (feat_dim / 8) * m * 8 * exp;          // C_last * F_last   (resnet.py:123)
    if ((err = pack_linear(sd, "seg_1", embed_dim, 2 * stats_dim, true, &seg1))) return err;
    two_emb = sd.count("seg_2.weight") > 0;
    if (two_emb) {
      if ((err = add_bn_vectors(sd, "seg_bn_1", embed_dim, &seg_bn_scale, &seg_bn_shift, false))) return err;
      seg1.scale = seg_bn_scale; seg1.shift = seg_bn_shift;     // relu -> BN(affine=False) epilogue
      if ((err = pack_linear(sd, "seg_2", embed_dim, embed_dim, true, &seg2))) return err;
    }
    if ((err = upload_weights())) return err;
    return reserve(max_batch, max_frames);
  }

  int reserve(int max_batch, int max_frames) override {
    int err = 0;
    maxB = max_batch; maxT = max_frames;
    // largest activation: stage-1 output (F x T x 32*exp); scratch planes are never larger
    const size_t act = (size_t)maxB * feat_dim * maxT * (size_t)(32 * exp);
    // (the binary16 convolution kernels address a tensor with 32-bit element offsets: one engine chunk keeps
    // every activation map below 2^31 elements -- ~4200 x 2 s utterances for BasicBlock nets, ~1060 for
    // Bottleneck nets; larger batches are processed in chunks anyway)
    if (act >= (size_t)1 << 31) {
      set_error("%s: max_batch %d x max_frames %d puts %zu elements in one activation map (limit 2^31); "
                "use a smaller engine chunk", name.c_str(), max_batch, max_frames, act);
      return WS_ERR_CAPACITY;
    }
    size_t total = 0;
    auto take = [&](size_t n) { size_t o = total; total += (n + 63) & ~size_t(63); return o; };
    size_t ob[4];
    // (+ 512 floats of zeros behind every activation buffer, never written: the border taps of the persistent
    // kernel's CONV form read Cin <= 512 of them -- ConvGemmParams::a_zero_off)
    for (int i = 0; i < 4; ++i) ob[i] = take(act + 512);
    act_floats = act;
    size_t o_pool = take((size_t)maxB * 2 * stats_dim),
           o_part = take((size_t)kSplitK * maxB * embed_dim), o_emba = take((size_t)maxB * embed_dim),
           o_feats = take((size_t)maxB * maxT * feat_dim);
    if ((err = alloc_workspace(total))) return err;
    float* base = ws.as<float>();
    for (int i = 0; i < 4; ++i) {
      buf[i] = base + ob[i];
      if (hipMemset(buf[i] + act, 0, 512 * sizeof(float)) != hipSuccess) {
        set_error("zero pad of activation buffer %d", i);
        return WS_ERR_HIP;
      }
    }
    pooled = base + o_pool; partial = base + o_part; emb_a = base + o_emba; feats_ws = base + o_feats;
    // (hipMemset runs on the null stream, the forwards on the caller's: the pads are zero before reserve() returns)
    if (hipDeviceSynchronize() != hipSuccess) return WS_ERR_HIP;
    return 0;
  }

  int min_frames() const override { return 2; }

  //  The CONV form of the persistent GEMM reads border taps from the 512 zero floats behind each activation
  // buffer.  Nothing may ever store there -- a whole-tile store without a row guard would, and every later image
  // border would be silently wrong: ws_engine_check_range looks (8 KB device -> host, behind its synchronisation).
  int check_invariants() override {
    if (!act_floats) return 0;
    std::vector<float> pad(512);
    for (int i = 0; i < 4; ++i) {
      if (hipMemcpy(pad.data(), buf[i] + act_floats, 512 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("%s: reading the zero pad of activation buffer %d failed", name.c_str(), i);
        return WS_ERR_HIP;
      }
      for (int k = 0; k < 512; ++k)
        if (pad[k] != 0.f) {
          set_error("%s: the zero pad behind activation buffer %d was overwritten (float %d = %g): a kernel stored "
                    "past its tensor; convolution borders of later forwards are wrong", name.c_str(), i, k, pad[k]);
          return WS_ERR_STATE;
        }
    }
    return 0;
  }

  int forward_chunk(const float* feats, int B, int T, float* emb, hipStream_t st) override {
    int H = feat_dim, W = T;
    // ragged chunk: the time axis of utterance b holds cur_lens[level][b] valid columns at stride level
    // `level`; every convolution stores zeros beyond them (row_len), TSTP runs over the valid columns
    int lvl = 0;
    // f16 back-end: every activation map lives in HBM as binary16 only (same buffers, half used);
    // all convolutions then run on the LDS-DMA kernel (conv form), residuals are read as halfs
    const bool f16io = gemm_precision == 2;
    float* x = buf[0];
    WS_LAUNCH(other(4.0 * B * H * (double)W * 33, st, [&] {
      return launch_stem_conv3x3(feats, B, T, feat_dim, arena.at(stem_w), arena.at(stem_b), 32, x, st,
                                 f16io ? reinterpret_cast<uint16_t*>(x) : nullptr, cur_lens[0]);
    }));
    // in -> out convolution with the optional residual, in the tensor format of the active back-end
    auto conv = [&](const ConvW& cw, const float* in, int Cin_, float* out, int Cout_, int Hin_, int Win_,
                    int s_, int pad, int act, const float* res, int ldr, int out_lvl) {
      ConvGemmParams p = conv2d(cw, in, Cin_, 0, out, Cout_, 0, B, Hin_, Win_, s_, s_, 1, 1, pad, pad, act);
      p.row_len = cur_lens[out_lvl];                       // stride level of this launch's OUTPUT width
      for (int i = 0; i < 4; ++i)                          // the zero pad behind the input's buffer
        if (in == buf[i] && Cin_ <= 512) p.a_zero_off = (long long)act_floats * (long long)sizeof(float);
      if (f16io) {
        p.A16 = reinterpret_cast<const uint16_t*>(in); p.lda16 = Cin_;
        p.D = nullptr; p.D16 = reinterpret_cast<uint16_t*>(out); p.ldd16 = Cout_;
        if (res) { p.residual16 = reinterpret_cast<const uint16_t*>(res); p.ldr = ldr; p.r_off = 0; }
      } else if (res) {
        p.residual = res; p.ldr = ldr; p.r_off = 0;
      }
      return gemm(p, st);
    };
    int cur = 0;          // index of the buffer holding x
    bool have_y1 = false; // the current block's conv1 output already lies in buf[y1_idx] (written by the previous block)
    int y1_idx = 0;
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
      const Block& blk = blocks[bi];
      // buffer roles of this block: x = buf[cur]; y1 (conv1's output, later the block's result) = buf[ia]; the spare
      // one (shortcut output, or the NEXT block's y1 when conv3 is fused with its conv1) = buf[ic]; y2 = buf[ib].
      // Without a y1 handed over by the previous block this is the plain rotation cur + 1 / + 2 / + 3.
      const int ia = (lay.bottleneck && have_y1) ? y1_idx : (cur + 1) & 3;
      int rest[2], nrest = 0;
      for (int k = 1; k <= 3; ++k)
        if (((cur + k) & 3) != ia) rest[nrest++] = (cur + k) & 3;
      const int ic = rest[0], ib = rest[1];
      float* t1 = buf[ia];
      float* t2 = buf[ic];
      float* out = buf[ib];
      const int s = blk.stride;
      const int Ho = (H - 1) / s + 1, Wo = (W - 1) / s + 1;
      const int Cx = blk.in_planes, P = blk.planes, Co = P * exp;
      const int lo = lvl + (s == 2 ? 1 : 0);               // stride level of the block's output
      const float* res = x;
      int ldr = Cx;
      if (blk.has_sc) {     // 1x1 stride-s conv + BN on the block input
        WS_LAUNCH(conv(blk.sc, x, Cx, t2, Co, H, W, s, 0, ACT_NONE, nullptr, 0, lo));
        res = t2;
        ldr = Co;
      }
      if (lay.bottleneck) {
        if (!have_y1) WS_LAUNCH(conv(blk.c1, x, Cx, t1, P, H, W, 1, 0, ACT_RELU, nullptr, 0, lvl));
        have_y1 = false;                // (else: the previous block's fused launch left this block's y1 in buf[ia])
        WS_LAUNCH(conv(blk.c2, t1, P, out, P, H, W, s, 1, ACT_RELU, nullptr, 0, lo));
        // conv3 (+ residual, ReLU) of this block and conv1 (+ ReLU) of the next one in ONE pass over the block output
        // (bneck_fuse.hip): fp32, 32 / 64 planes, the next block with the same or twice the planes
        // (its conv1 has no stride: a block's stride sits in conv2).  The next block's y1 goes into the one buffer that
        // is dead by now: the spare one, or -- when this block has a shortcut convolution, whose output IS the spare
        // one -- the block's own input x (read by conv1 and the shortcut only)
        const Block* nb = bi + 1 < blocks.size() ? &blocks[bi + 1] : nullptr;
        BneckFuseParams fp = {};
        if (gemm_precision == 0 && nb && (nb->planes == P || nb->planes == 2 * P) && nb->in_planes == Co &&
            blk.c3.has_b && nb->c1.has_b && !blk.c3.has_post && !nb->c1.has_post) {
          fp.y2 = out; fp.W3 = arena.at(blk.c3.w); fp.ldw3 = blk.c3.ldw; fp.b3 = arena.at(blk.c3.b);
          fp.res = res; fp.ldr = ldr; fp.out = t1;
          fp.W1 = arena.at(nb->c1.w); fp.ldw1 = nb->c1.ldw; fp.b1 = arena.at(nb->c1.b);
          fp.y1 = blk.has_sc ? buf[cur] : t2;
          fp.M = B * Ho * Wo; fp.P = P; fp.PN = nb->planes;
          fp.row_len = cur_lens[lo]; fp.HW = Ho * Wo; fp.W = Wo;      // (a ragged chunk: null otherwise)
        }
        if (fp.y2 && bneck_fuse_supported(fp)) {
          const double M_ = (double)fp.M;
          if (prof.enabled) prof.begin(0, 2.0 * M_ * (P * (double)Co + Co * (double)P), 4.0 * M_ * (P + 2.0 * Co + P), st);
          hipError_t fe = launch_bneck_fuse(fp, st);
          prof.end(st);
          WS_LAUNCH(fe);
          have_y1 = true;
          y1_idx = blk.has_sc ? cur : ic;
        } else {
          WS_LAUNCH(conv(blk.c3, out, P, t1, Co, Ho, Wo, 1, 0, ACT_RELU, res, ldr, lo));
        }
        cur = ia;                       // result in t1
      } else {
        WS_LAUNCH(conv(blk.c1, x, Cx, t1, P, H, W, s, 1, ACT_RELU, nullptr, 0, lo));
        WS_LAUNCH(conv(blk.c2, t1, P, out, Co, Ho, Wo, 1, 1, ACT_RELU, res, ldr, lo));
        cur = (cur + 3) & 3;            // result in out
      }
      x = buf[cur];
      H = Ho; W = Wo;
      lvl = lo;
    }
    const int Cl = 256 * exp;           // channels of the last stage
    WS_LAUNCH(other((f16io ? 4.0 : 8.0) * B * H * (double)W * Cl, st, [&] {
      if (f16io)
        return launch_tstp_f16(reinterpret_cast<const uint16_t*>(x), Cl, B, H, W, Cl, nullptr, nullptr,
                               pooled, st, cur_lens[lvl]);
      return launch_tstp(x, Cl, B, H, W, Cl, nullptr, nullptr, pooled, st, cur_lens[lvl]);
    }));
    if (!two_emb) {
      WS_LAUNCH(gemm_splitk(conv1d(seg1, pooled, 2 * stats_dim, 0, emb, embed_dim, 0, B, 1, 1, ACT_NONE),
                            partial, kSplitK, st));
    } else {            // embed_b = seg_2(seg_bn_1(relu(seg_1(stats))))   (resnet.py:198-202)
      ConvGemmParams a = conv1d(seg1, pooled, 2 * stats_dim, 0, emb_a, embed_dim, 0, B, 1, 1, ACT_RELU);
      a.post_scale = arena.at(seg_bn_scale); a.post_shift = arena.at(seg_bn_shift);
      WS_LAUNCH(gemm_splitk(a, partial, kSplitK, st));
      WS_LAUNCH(gemm(conv1d(seg2, emb_a, embed_dim, 0, emb, embed_dim, 0, B, 1, 1, ACT_NONE), st));
    }
    return 0;
  }

  double flops(int batch, int T) const override {
    double macs = 0;
    int H = feat_dim, W = T;
    macs += (double)H * W * 9 * 32;
    for (const Block& blk : blocks) {
      const int s = blk.stride, Ho = (H - 1) / s + 1, Wo = (W - 1) / s + 1;
      const double in = blk.in_planes, P = blk.planes, Co = P * exp;
      if (lay.bottleneck)
        macs += (double)H * W * in * P + (double)Ho * Wo * 9 * P * P + (double)Ho * Wo * P * Co;
      else
        macs += (double)Ho * Wo * 9 * in * P + (double)Ho * Wo * 9 * P * P;
      if (blk.has_sc) macs += (double)Ho * Wo * in * Co;
      H = Ho; W = Wo;
    }
    macs += 2.0 * stats_dim * embed_dim;
    if (two_emb) macs += (double)embed_dim * embed_dim;
    return 2.0 * macs * batch;
  }
};

}  // namespace

Model* make_resnet(const std::string& model_name, int feat_dim, int embed_dim) {
  for (const Layout& l : kLayouts)
    if (model_name == l.name) return new ResNetModel(l, feat_dim, embed_dim);
  return nullptr;
}

}  // namespace wsamd
