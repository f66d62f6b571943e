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
#include <cstdlib>

#include "model_common.h"

namespace wsamd {

namespace {

struct EcapaModel : ModelBase {
  int C = 512, w = 64;
  bool glob = false;
  ConvW layer1, blk0[3], res2[3][7], blk2[3], catconv, pool1, pool2, final_lin;
  size_t se_w1[3], se_b1[3], se_w2[3], se_b2[3], se_w2t[3];
  float *out1 = nullptr, *y1 = nullptr, *y2 = nullptr, *y3 = nullptr, *cat = nullptr, *h = nullptr,
        *att = nullptr, *e = nullptr, *se_s = nullptr, *stats = nullptr, *bias_img = nullptr,
        *pooled = nullptr, *partial = nullptr, *colsum = nullptr, *colsumsq = nullptr;
  uint16_t *h16 = nullptr, *cat16 = nullptr, *out1_16 = nullptr, *att16 = nullptr, *y2_16 = nullptr, *y3_16 = nullptr, *col16 = nullptr;
  static constexpr int kSplitK = 16;

  EcapaModel(const std::string& n, int fd, int ed) : ModelBase(n, fd, ed) {
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

  int finalize(const SD& sd, int max_batch, int max_frames) override {
    int err = 0;
    if ((err = pack_conv1d(sd, "layer1.conv", C, feat_dim, 5, true, "", "layer1.bn", &layer1))) return err;
    for (int L = 0; L < 3; ++L) {
      std::string p = "layer" + std::to_string(L + 2) + ".se_res2block";
      if ((err = pack_conv1d(sd, p + ".0.conv", C, C, 1, true, "", p + ".0.bn", &blk0[L]))) return err;
      for (int i = 0; i < 7; ++i)
        if ((err = pack_conv1d(sd, p + ".1.convs." + std::to_string(i), w, w, 3, true, "",
                               p + ".1.bns." + std::to_string(i), &res2[L][i])))
          return err;
      if ((err = pack_conv1d(sd, p + ".2.conv", C, C, 1, true, "", p + ".2.bn", &blk2[L]))) return err;
      const HostTensor* t;
      if (!(t = get(sd, p + ".3.linear1.weight", {128, C}, &err))) return err;
      se_w1[L] = arena.add(t->data);
      if ((err = add_vec(sd, p + ".3.linear1.bias", 128, &se_b1[L]))) return err;
      if (!(t = get(sd, p + ".3.linear2.weight", {C, 128}, &err))) return err;
      se_w2[L] = arena.add(t->data);
      {   // [128][C] copy for the FC kernels that run from the column sums (a wavefront reads 256 contiguous bytes)
        std::vector<float> w2t((size_t)C * 128);
        for (int c = 0; c < C; ++c)
          for (int k = 0; k < 128; ++k) w2t[(size_t)k * C + c] = t->data[(size_t)c * 128 + k];
        se_w2t[L] = arena.add(w2t);
      }
      if ((err = add_vec(sd, p + ".3.linear2.bias", C, &se_b2[L]))) return err;
    }
    if ((err = pack_conv1d(sd, "conv", 1536, 3 * C, 1, true, "", "", &catconv))) return err;
    if ((err = pack_conv1d(sd, "pool.linear1", 128, glob ? 4608 : 1536, 1, true, "", "", &pool1)))
      return err;
    if ((err = pack_conv1d(sd, "pool.linear2", 1536, 128, 1, true, "", "", &pool2))) return err;
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
          const double wv = lw->data[(size_t)o * 3072 + k];
          acc += wv * sh[k];
          W[(size_t)o * 3072 + k] = (float)(wv * sc[k] * s2[o]);
        }
        B[o] = (float)(acc * s2[o] + t2[o]);
      }
      final_lin.N = embed_dim; final_lin.Cin = 3072; final_lin.ldw = 3072;
      final_lin.w = arena.add(W);
      add_split(&final_lin, W);
      final_lin.b = arena.add(B);
      final_lin.has_b = true;
    }
    if ((err = upload_weights())) return err;
    return reserve(max_batch, max_frames);
  }

  // workspace layout for chunks of max_batch utterances x max_frames frames (re-callable:
  // ws_engine_reserve grows it when a longer utterance arrives)
  int reserve(int max_batch, int max_frames) override {
    int err = 0;
    maxB = max_batch; maxT = max_frames;
    const size_t M = (size_t)maxB * maxT;
    size_t total = 0;
    auto take = [&](size_t n) { size_t o = total; total += (n + 63) & ~size_t(63); return o; };
    size_t o_out1 = take(M * C), o_y1 = take(M * C), o_y2 = take(M * C), o_y3 = take(M * C),
           o_cat = take(M * 3 * C), o_h = take(M * 1536), o_att = take(M * 128),
           o_e = take(M * 1536), o_s = take((size_t)maxB * C), o_stats = take((size_t)maxB * 3072),
           o_bias = take((size_t)maxB * 128), o_pool = take((size_t)maxB * 3072),
           o_part = take((size_t)kSplitK * maxB * (embed_dim > 128 ? embed_dim : 128)),
           o_h16 = take((M * (size_t)(1536 + 3 * C + C + 128 + C + C + 512) + 1) / 2),   // binary16 copies (f16 back-end)
           o_colsum = take(((M + 63) / 64 + 2) * 2 * (C > 1536 ? C : 1536)),
           o_colsumsq = take(((M + 63) / 64 + 2) * 2 * 1536), o_feats = take(M * feat_dim);
    if ((err = alloc_workspace(total))) return err;
    float* base = ws.as<float>();
    out1 = base + o_out1; y1 = base + o_y1; y2 = base + o_y2; y3 = base + o_y3; cat = base + o_cat;
    h = base + o_h; att = base + o_att; e = base + o_e; se_s = base + o_s; stats = base + o_stats;
    bias_img = base + o_bias; pooled = base + o_pool; partial = base + o_part;
    colsum = base + o_colsum; colsumsq = base + o_colsumsq; feats_ws = base + o_feats;
    h16 = reinterpret_cast<uint16_t*>(base + o_h16);
    cat16 = h16 + M * 1536; out1_16 = cat16 + M * 3 * C; att16 = out1_16 + M * C; y2_16 = att16 + M * 128; y3_16 = y2_16 + M * C; col16 = y3_16 + M * C;
    return 0;
  }

  int min_frames() const override { return 1; }

  int forward_chunk(const float* feats, int B, int T, float* emb, hipStream_t st) override {
    // Ragged chunk (cur_lens set): utterance b owns rows [0, lens[b]) of its T-row slot.  Every conv/linear
    // launch stores zeros in the padding rows (ConvGemmParams::row_len), so the column sums that the GEMM
    // epilogues leave for the SE / context statistics are already right (only their divisors become lens[b]);
    // the fused Res2 chain, the centred second moments and the softmax-pooling epilogue run over lens[b] rows.
    const int* L0 = cur_lens[0];
    // layer1: Conv1d(F -> C, k5, p2) -> ReLU -> BN
    // f16 back-end: the layers that feed 1x1 GEMMs also leave a binary16 copy of their output, which
    // those GEMMs read instead of the fp32 tensor (half the bytes, no conversion while staging)
    const bool f16io = gemm_precision == 2;
    // ... and with T >= 64 (SE statistics from the epilogue) the block chain out1 -> y3 -> cat lives in
    // binary16 only: the residual stream is rounded once per block like in any fp16 inference engine
    const bool allf16 = f16io && T >= 64;
    {
      ConvGemmParams p0 = conv1d(layer1, feats, feat_dim, 0, out1, C, 0, B, T, 1, ACT_RELU);
      if (f16io) { p0.D16 = out1_16; p0.ldd16 = C; }
      if (allf16) p0.D = nullptr;
      if (f16io && layer1.ldw <= 512 && feat_dim % 4 == 0) {
        // k5 conv as a plain GEMM over a binary16 im2col image (K = ldw = taps*F rounded up to 64)
        WS_LAUNCH(other(2.0 * B * (double)T * layer1.ldw, st, [&] {
          return launch_im2col_f16(feats, B, T, feat_dim, 5, 2, col16, layer1.ldw, st);
        }));
        p0.A16 = col16; p0.lda16 = layer1.ldw; p0.lda = layer1.ldw;
        p0.K = layer1.ldw; p0.Cin = layer1.ldw; p0.kw = 1; p0.pad_w = 0; p0.dil_w = 1;
      }
      // fp32 back-end, full batches: the same trick in fp32 -- the im2col image (K = taps*F rounded up to 32
      // columns: 13 K-tiles for 5 x 80) lives in the not-yet-used h buffer and the layer becomes a plain 1x1 GEMM of
      // the persistent kernel (the implicit-GEMM form stays for ragged and small batches: same k order, same bits)
      const int kcol = (5 * feat_dim + 31) & ~31;
      if (gemm_precision == 0 && !L0 && feat_dim % 4 == 0 && kcol <= layer1.ldw && kcol <= 1536 &&
          (long long)B * T >= 16384) {
        WS_LAUNCH(other(4.0 * B * (double)T * (kcol + feat_dim), st, [&] {
          return launch_im2col_f32(feats, B, T, feat_dim, 5, 2, h, kcol, st);
        }));
        p0.A = h; p0.lda = kcol; p0.a_off = 0;
        p0.K = kcol; p0.Cin = kcol; p0.kw = 1; p0.pad_w = 0; p0.dil_w = 1;
      }
      p0.row_len = L0;
      WS_LAUNCH(gemm(p0, st, 5 * feat_dim));
    }
    for (int L = 0; L < 3; ++L) {
      const int d = L + 2;
      const float* x = L == 0 ? out1 : cat;
      const int ldx = L == 0 ? C : 3 * C;
      const int x_off = L == 0 ? 0 : (L - 1) * C;
      // 1x1 conv -> ReLU -> BN; the last Res2 split is passed through untouched: dual store
      ConvGemmParams p = conv1d(blk0[L], x, ldx, x_off, y1, C, 0, B, T, 1, ACT_RELU);
      p.D2 = y2; p.ldd2 = C; p.d2_off = 7 * w; p.d2_col0 = 7 * w;
      // f16 back-end + fused Res2 chain: y2 exists only as binary16 (conv3 reads it by LDS-DMA)
      const bool y2_half = f16io && res2_half_out_supported(w, T, d);
      if (f16io) { p.A16 = L == 0 ? out1_16 : cat16; p.lda16 = ldx; }
      if (y2_half) { p.D2_16 = y2_16; p.ldd2_16 = C; }
      p.row_len = L0;
      WS_LAUNCH(gemm(p, st));
      // Res2: sp_i = BN(ReLU(conv_k3_dil(sp_{i-1} + split_i)))
      if (res2_chain_supported(w, T, d)) {       // one launch, running activation kept in LDS
        Res2ChainParams r = {};
        r.y1 = y1; r.ldy1 = C; r.y2 = y2; r.ldy2 = C; r.ldw = res2[L][0].ldw;
        for (int i = 0; i < 7; ++i) {
          r.w[i] = arena.at(res2[L][i].w); r.bias[i] = arena.at(res2[L][i].b);
          r.wh[i] = reinterpret_cast<const uint16_t*>(arena.at(res2[L][i].wh));
          r.wl[i] = reinterpret_cast<const uint16_t*>(arena.at(res2[L][i].wl));
          r.scale[i] = arena.at(res2[L][i].scale); r.shift[i] = arena.at(res2[L][i].shift);
        }
        r.B = B; r.T = T; r.W = w; r.dil = d; r.prec = gemm_precision; r.lens = L0;
        r.y2h = y2_half ? y2_16 : nullptr; r.ldy2h = C;
        if (prof.enabled) prof.begin(1, 2.0 * B * (double)T * w * 3 * w * 7, 4.0 * B * (double)T * C * 2, st);
        hipError_t re = launch_res2_chain(r, st);
        prof.end(st);
        WS_LAUNCH(re);
      } else {
        for (int i = 0; i < 7; ++i) {
          ConvGemmParams q = conv1d(res2[L][i], y1, C, i * w, y2, C, i * w, B, T, d, ACT_RELU);
          if (i >= 1) { q.A2 = y2; q.lda2 = C; q.a2_off = (i - 1) * w; }
          q.row_len = L0;
          WS_LAUNCH(gemm(q, st));
        }
      }
      const double mc = 4.0 * B * (double)T * C;
      ConvGemmParams p3 = conv1d(blk2[L], y2, C, 0, y3, C, 0, B, T, 1, ACT_RELU);
      if (y2_half) { p3.A16 = y2_16; p3.lda16 = C; }
      if (allf16) { p3.D = nullptr; p3.D16 = y3_16; p3.ldd16 = C; }
      p3.row_len = L0;
      // fp32 activations, T >= 64: the SE FCs, the scale and the block residual are ONE launch (a workgroup per
      // utterance; the two-launch form below gives the same bits)
      if (T >= 64 && !allf16 && se_fc_scale_residual_supported(T, C, 128)) {
        p3.colsum = colsum;
        WS_LAUNCH(gemm(p3, st));
        WS_LAUNCH(other(3 * mc, st, [&] {
          return launch_se_fc_scale_residual(colsum, B, T, C, arena.at(se_w1[L]), arena.at(se_b1[L]),
                                             arena.at(se_w2t[L]), arena.at(se_b2[L]), 128, se_s, L0, x, ldx, x_off,
                                             y3, C, cat, 3 * C, L * C, st, f16io ? cat16 : nullptr);
        }));
        continue;
      }
      if (T >= 64 && allf16 && se_fc_scale_residual_f16_supported(T, C, 128)) {      // the same on binary16 rows
        p3.colsum = colsum;
        WS_LAUNCH(gemm(p3, st));
        WS_LAUNCH(other(1.5 * mc, st, [&] {
          return launch_se_fc_scale_residual_f16(colsum, B, T, C, arena.at(se_w1[L]), arena.at(se_b1[L]),
                                                 arena.at(se_w2t[L]), arena.at(se_b2[L]), 128, se_s, L0,
                                                 L == 0 ? out1_16 : cat16, ldx, x_off, y3_16, C, cat16, 3 * C, L * C, st);
        }));
        continue;
      }
      if (T >= 64) {
        // SE time-mean from the GEMM epilogue's per-tile column sums: y3 is not re-read
        p3.colsum = colsum;
        WS_LAUNCH(gemm(p3, st));
        WS_LAUNCH(other(0.0, st, [&] {
          return launch_se_fc_from_colsum(colsum, B, T, C, arena.at(se_w1[L]), arena.at(se_b1[L]),
                                          arena.at(se_w2t[L]), arena.at(se_b2[L]), 128, se_s, st, L0);
        }));
      } else {
        WS_LAUNCH(gemm(p3, st));
        WS_LAUNCH(other(mc, st, [&] {
          return launch_se_pool_fc(y3, C, B, T, C, arena.at(se_w1[L]), arena.at(se_b1[L]),
                                   arena.at(se_w2[L]), arena.at(se_b2[L]), 128, se_s, st, L0);
        }));
      }
      WS_LAUNCH(other(allf16 ? 1.5 * mc : 3 * mc, st, [&] {
        if (allf16)
          return launch_se_scale_residual_f16(L == 0 ? out1_16 : cat16, ldx, x_off, y3_16, C, se_s, cat16,
                                              3 * C, L * C, B, T, C, st);
        return launch_se_scale_residual(x, ldx, x_off, y3, C, se_s, cat, 3 * C, L * C, B, T, C, st,
                                        f16io ? cat16 : nullptr);
      }));
    }
    // cat -> Conv1d(3C -> 1536, k1) -> ReLU
    // (GLOB, T >= 64: the epilogue also leaves per-tile column sums of h for the context statistics)
    const bool stats_from_colsum = glob && T >= 64;
    const bool stats_from_sums = stats_from_colsum && gemm_precision == 0 && !L0;
    const bool h_half = allf16;
    {
      ConvGemmParams pc = conv1d(catconv, cat, 3 * C, 0, h, 1536, 0, B, T, 1, ACT_RELU);
      if (stats_from_colsum) pc.colsum = colsum;
      // fp32, full batches: the epilogue also leaves the column sums of SQUARES, and the context std needs no pass
      // over h (311 MB at 256 x 2 s) any more
      if (stats_from_sums) pc.colsumsq = colsumsq;
      if (f16io) { pc.A16 = cat16; pc.lda16 = 3 * C; pc.D16 = h16; pc.ldd16 = 1536; }
      if (h_half) pc.D = nullptr;              // h exists as binary16 only
      pc.row_len = L0;
      WS_LAUNCH(gemm(pc, st));
    }
    // ASTP
    ConvGemmParams a1 = conv1d(pool1, h, 1536, 0, att, 128, 0, B, T, 1, ACT_TANH);
    a1.K = 1536; a1.Cin = 1536;                      // GLOB: only the first 1536 columns multiply h
    if (f16io) { a1.A16 = h16; a1.lda16 = 1536; a1.D16 = att16; a1.ldd16 = 128; }
    if (glob) {
      // [mean; std] statistics, then bias_img = W1[:, C:3C] [mean; std] + b1 as a split-K GEMM
      WS_LAUNCH(other(4.0 * B * (double)T * 1536, st, [&] {
        if (stats_from_colsum && h_half)
          return launch_astp_std_from_colsum_f16(h16, 1536, B, T, 1536, colsum, stats, st, L0);
        if (stats_from_sums) return launch_astp_std_from_sums(colsum, colsumsq, B, T, 1536, stats, st, h, 1536);
        if (stats_from_colsum) return launch_astp_std_from_colsum(h, 1536, B, T, 1536, colsum, stats, st, L0);
        return launch_astp_stats(h, 1536, B, T, 1536, stats, st, L0);
      }));
      ConvGemmParams cb = conv1d(pool1, stats, 3072, 0, bias_img, 128, 0, B, 1, 1, ACT_NONE);
      cb.W = arena.at(pool1.w) + 1536; cb.Wh += 1536; cb.Wl += 1536; cb.K = 3072; cb.Cin = 3072;
      WS_LAUNCH(gemm_splitk(cb, partial, kSplitK, st));
      a1.bias = nullptr;
      a1.bias_img = bias_img;
    }
    a1.row_len = L0;
    if (gemm_precision == 0 && astp_fused_supported(T, 1536, 128) && astp_fused_pays(B, T)) {
      // linear1 -> tanh -> linear2 -> softmax over time -> weighted mean / std: one workgroup per utterance,
      // neither the bottleneck activations nor the logits leave the chip (astp_fused.hip)
      if (prof.enabled) {
        const double m = (double)B * T;
        prof.begin(0, 2.0 * m * 128 * 1536 * 2, 4.0 * (2.0 * m * 1536 + 2.0 * 128 * 1536 + 2.0 * B * 1536), st);
      }
      if (dispatch_log_enabled()) {
        ConvGemmParams note = a1;
        note.N = 1536 + 128;
        dispatch_log_note(note, "astp_fused_kernel (linear1 + tanh + linear2 + softmax pooling)");
      }
      hipError_t fe = launch_astp_fused(h, 1536, B, T, arena.at(pool1.w), pool1.ldw, glob ? nullptr : a1.bias,
                                        glob ? bias_img : nullptr, arena.at(pool2.w), pool2.ldw, pooled, L0, st,
                                        e);       // (e: the logits buffer of the unfused path, free here: segment tuples)
      prof.end(st);
      WS_LAUNCH(fe);
    } else {
    WS_LAUNCH(gemm(a1, st));
    if (T >= 64) {
      // logits never leave the chip: the GEMM epilogue reduces them to online-softmax partials
      ConvGemmParams l2 = conv1d(pool2, att, 128, 0, nullptr, 1536, 0, B, T, 1, ACT_NONE);
      l2.pool_h = h; l2.ldh = 1536; l2.pool_partial = e;      // e doubles as the partials buffer
      if (h_half) { l2.pool_h = nullptr; l2.pool_h16 = h16; }
      if (f16io) { l2.A16 = att16; l2.lda16 = 128; }
      l2.row_len = L0;
      WS_LAUNCH(gemm(l2, st));
      WS_LAUNCH(other(0.0, st, [&] {
        return launch_astp_pool_from_partials(e, B, T, 1536, pooled, st);
      }));
    } else {
      WS_LAUNCH(gemm(conv1d(pool2, att, 128, 0, e, 1536, 0, B, T, 1, ACT_NONE), st));
      WS_LAUNCH(other(8.0 * B * (double)T * 1536, st, [&] {
        return launch_astp_pool(e, 1536, h, 1536, B, T, 1536, pooled, st, L0);
      }));
    }
    }
    // BN + Linear (+bn2), folded: split-K GEMM over K = 3072
    WS_LAUNCH(gemm_splitk(conv1d(final_lin, pooled, 3072, 0, emb, embed_dim, 0, B, 1, 1, ACT_NONE),
                          partial, kSplitK, st));
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
