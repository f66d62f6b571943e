// Internal launcher interface between the host engine (engine.cpp / plda.cpp) and the gfx950
// kernels.  Everything here is HIP-only (no torch); all launchers enqueue on `stream` and
// return the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wsamd {

// Compute units of the CURRENT device, cached per device id (grids and cost models are sized by it; a process may
// drive several GPUs with different counts).
inline int current_device_cus() {
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cache[dev]) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cache[dev] = cus > 0 ? cus : 256;
  }
  return cache[dev];
}


// Opt-in to more than 64 KB of dynamic LDS: hipFuncSetAttribute acts on the CURRENT device's copy of the kernel, so
// "done" is remembered per (kernel, device) -- a per-kernel `static bool` let the second GPU of a process launch
// without the attribute (one process per GPU is how this library is deployed, but the C-ABI takes a device index).
// `granted[d]` = bytes already granted on device d; the launchers keep one such array per kernel instantiation.
constexpr int WS_MAX_DEVICES = 64;
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes, size_t (&granted)[WS_MAX_DEVICES]) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  size_t& g = granted[dev & (WS_MAX_DEVICES - 1)];
  if (bytes <= g) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) g = bytes;
  return e;
}


enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

// ReLU with torch.relu's treatment of non-finite values: NaN stays NaN, +inf stays +inf (bit-identical
// to fmaxf(v, 0) on finite input).  fmaxf returns the non-NaN operand, i.e. it would turn the NaN that a
// binary16 overflow upstream produces (inf - inf, inf * 0) back into an innocent 0 and let the damage
// reach the embedding as a finite wrong number; with this form it arrives as NaN and
// ws_engine_check_range reports it.
__device__ __forceinline__ float relu_f(float v) { return v < 0.f ? 0.f : v; }


// Implicit-GEMM convolution / linear layer on channels-last activations:
//   D[m][n] = epilogue( sum_{tap, ci} A[pix(m, tap)][a_off + ci] * W[n][tap*Cin + ci] )
// m enumerates output pixels (img, oy, ox) row-major; pix() applies stride/dilation/zero padding.
// Conv1d over time is the H=1 case (time on the W axis); a Linear layer is 1x1 with Hout=Wout=1.
struct ConvGemmParams {
  const float* A;  int lda;  int a_off;       // input rows: lda floats per pixel
  const uint16_t* A16; int lda16;             // optional binary16 copy of A (same a_off), read instead of A
                                              // by the f16 back-end on 1x1 layers (no conversion, half the bytes)
  const float* A2; int lda2; int a2_off;      // optional second input, added element-wise to A
  const float* pre_scale; const float* pre_shift;   // optional pre-activation on A (CAM++ BN-ReLU
                                              // before the conv): a = relu(A*pre_scale[ci] + pre_shift[ci]),
                                              // applied to in-bounds pixels only (padding stays 0)
  const float* W;  int ldw;                   // [N][ldw]; ldw >= taps*Cin, multiple of 32, zero padded
  const uint16_t* Wh; const uint16_t* Wl;     // hi / lo binary16 planes of W (same layout), prec == 1
  int prec;                                   // 0: v_mfma_f32_32x32x2_f32 (exact fp32)
                                              // 1: 3 x v_mfma_f32_32x32x16_f16 on hi/lo splits
  float* D;  int ldd;  int d_off;             // output rows (may be null when only D16 is wanted)
  uint16_t* D16; int ldd16;                   // optional binary16 copy of the stored values: D16[m][d_off + n]
  float* D2; int ldd2; int d2_off; int d2_col0;
  uint16_t* D2_16; int ldd2_16;               // optional binary16 twin of D2 (same d2_off / d2_col0)  // optional: columns n >= d2_col0 also go to D2[m][d2_off + n - d2_col0]
  int M, N, K;                                // M output pixels, N output channels, K = taps*Cin
  int m_begin;                                // first output pixel of this launch (multiple of 64;
                                              // used by the launcher's tail peeling)
  int Cin;
  int Hin, Win, Hout, Wout;
  int stride_h, stride_w, kh, kw, dil_h, dil_w, pad_h, pad_w;
  const float* bias;                          // [N] or null
  const float* bias_img;                      // [num_images][N] or null (per-utterance bias)
  const float* residual; int ldr; int r_off;  // optional, added before the activation
  const uint16_t* residual16;                 // binary16 form of the residual (same ldr / r_off)
  int act;
  const float* post_scale; const float* post_shift;   // y = act(.)*scale[n] + shift[n] (BN after ReLU)
  const int* row_len;                         // optional [num_images] (ragged batch): output pixels with
                                              // ox >= row_len[img] are padding -- stored as zeros
  const float* seg_scale; int seg_len; int segs_per_img;  // optional y *= seg_scale[(img*segs + ox/seg_len)][n]
                                              // (CAM++ context mask, campplus.py:110-115)
  float* colsum;                              // optional [ceil(M/64)][2][N]: column sums of the stored
                                              // values per 64-row tile, split at the image boundary
                                              // inside the tile (needs Hout*Wout >= 64 rows per image)
  float* colsumsq;                            // optional, only with colsum: column sums of the SQUARES of the
                                              // stored values, same layout (fp32 back-end; the rows the persistent
                                              // kernel does not take are summed by colsumsq_rows_kernel)
  const uint16_t* pool_h16;                   // binary16 twin of pool_h (same ldh), used instead when set
  const float* pool_h; int ldh;               // optional fused attentive-statistics pooling: the GEMM
  float* pool_partial;                        // output is the LOGIT tensor e; instead of storing it,
                                              // online-softmax partials over rows of (max, sum w,
                                              // sum w*h, sum w*h^2), w = exp(e - max), per 64-row tile
                                              // and image part -> pool_partial[ceil(M/64)*2][N][4]
  float* partial;  int splitk;                // splitk > 1: raw partial sums -> partial[z][M][N]
  const float* zeros;                         // >= 16 B of zeros in device memory (masked loads)
  long long a_zero_off;                       // optional: byte offset FROM A of Cin zero floats that lie behind the
                                              // tensor within 32-bit reach (ResNet: a pad behind every activation
                                              // buffer); the CONV form of the persistent fp32 GEMM reads a border
                                              // tap from there with the same scalar base as its valid lanes
  int epi16;                                  // set by the dispatcher: the 256x256 kernel finishes a D16-only
                                              // layer through its binary16 one-phase epilogue
  int n_big, tail_begin;                      // set by the dispatcher (conv_gemm_dual_kernel): workgroups
                                              // [0, n_big) run 128x128 tiles over rows [m_begin, tail_begin),
                                              // the others 64x64 tiles over rows [tail_begin, M)
  int n_units;                                // set by launch_gemm_f32_stream: 64x64 row units over the rows
                                              // [tail_begin, tail_begin + 64 * n_units / (N / 64)) behind its tiles
};
hipError_t launch_conv_gemm(const ConvGemmParams& p, hipStream_t stream);

// Dispatch log (diagnostic, off by default; env WS_DISPATCH_LOG=1 or dispatch_log_enable): every kernel launch
// of the conv-GEMM family notes (problem, kernel) once, so that "which kernel did this layer get" -- and a layer
// that silently fell off the fast kernels -- can be read back (ws_debug_dispatch_report, tests/golden/dispatch_*).
void dispatch_log_enable(bool on);
bool dispatch_log_enabled();
void dispatch_log_note(const ConvGemmParams& p, const char* kernel);
void dispatch_log_note_text(const char* key);
size_t dispatch_log_dump(char* buf, size_t cap);      // text lines; returns the bytes the full report needs
void dispatch_log_clear();

// Persistent fp32 GEMM of the plain 1x1 layers (gemm_f32_stream.hip).  gemm_f32_stream_rows: how many rows of
// [p.m_begin, p.M) that kernel takes (whole rounds of 128x128 tiles over `cus` workgroups; 0 = not its problem);
// the caller sends the remaining rows through the tile kernels with m_begin advanced.
int gemm_f32_stream_rows(const ConvGemmParams& p, int cus);
// ... and of the 3x3 / stride 1 / pad 1 convolutions whose K-tiles lie inside one filter tap (Cin % 32 == 0,
// N % 128 == 0): the same kernel with the A pieces addressed per tap (CONV form)
bool gemm_f32_stream_is_conv3(const ConvGemmParams& p);
hipError_t launch_gemm_f32_stream(const ConvGemmParams& p, int rows, int cus, hipStream_t stream);

// out[m][n] = epilogue(sum_z partial[z][m][n]) with the same epilogue fields as ConvGemmParams.
hipError_t launch_splitk_reduce(const ConvGemmParams& p, hipStream_t stream);
// conv3 (+ residual, ReLU) of a Bottleneck block and conv1 (+ ReLU) of the next block in one pass over the block output
// (bneck_fuse.hip; fp32, planes P = 32): the block output is stored once and not read again by conv1
struct BneckFuseParams {
  const float* y2;                            // (M, P): conv2's output rows
  const float* W3; int ldw3; const float* b3; // conv3 + bn3 folded: [4P][ldw3], [4P]
  const float* res; int ldr;                  // (M, ldr >= 4P): the block's residual (its input, or the shortcut conv's output)
  float* out;                                 // (M, 4P): the block output
  const float* W1; int ldw1; const float* b1; // the NEXT block's conv1 + bn1 folded: [PN][ldw1], [PN]
  float* y1;                                  // (M, PN): the next block's conv1 output
  int M, P, PN;                               // PN: the next block's planes (P, or 2 P at the stage 1 -> 2 transition)
  const int* row_len; int HW, W;              // optional (ragged batch): [M / HW] widths, pixels per image, image width;
                                              // pixels with ox >= row_len[img] are stored as zeros in out AND y1
};
bool bneck_fuse_supported(const BneckFuseParams& p);
hipError_t launch_bneck_fuse(const BneckFuseParams& p, hipStream_t stream);
// The "M = batch" linear layers of the fp32 back-end in ONE launch instead of split-K GEMM + reduce (small_m_gemm.hip)
bool small_m_gemm_f32_applies(const ConvGemmParams& p);
hipError_t launch_small_m_gemm_f32(const ConvGemmParams& p, hipStream_t stream);

// Fused Res2 chain (ecapa_tdnn.py:58-78): 7 serial k=3 dilated convs of width W, one workgroup per
// utterance, running activation kept in LDS.  y1: output of the block's first 1x1 conv [B*T][ldy1]
// (8 splits of W channels); y2: [B*T][ldy2], splits 0..6 written here (split 7 is the pass-through).
struct Res2ChainParams {
  const float* y1; int ldy1;
  float* y2; int ldy2;
  uint16_t* y2h; int ldy2h;                 // optional: write the splits as binary16 HERE INSTEAD of y2
                                            // (f16 back-end: the following 1x1 conv reads them by LDS-DMA)
  const float* w[7]; int ldw;               // packed [W][tap*W + ci], ldw = 3W
  const uint16_t* wh[7]; const uint16_t* wl[7];   // hi / lo binary16 planes of w (prec == 1)
  const float* bias[7]; const float* scale[7]; const float* shift[7];
  int B, T, W, dil;
  int prec;                                 // 0 exact fp32 MFMA, 1 split-f16 x3 MFMA
  const int* lens;                          // optional [B]: valid frames per utterance (<= T); rows beyond
                                            // are padding (ragged batch)
  int y2h_direct;                           // set by launch_res2_chain: the binary16 rows go to y2h as 8-byte
                                            // stores from the accumulators (no LDS staging: it does not fit
                                            // beside the w = 128 activation planes)
  int force_wave8;                          // probes / tests: keep the eight-wavefront kernel (res2_chain4_kernel's A/B)
  int tiles, tile_rows;                     // set by launch_res2_chain: time tiles per utterance (1 = the whole
                                            // utterance in one workgroup) and the rows each tile OWNS; a tile
                                            // also recomputes a halo of 7 * dil rows on each side
};
bool res2_chain_supported(int W, int T, int dil);
bool res2_half_out_supported(int W, int T, int dil);   // Res2ChainParams::y2h allowed
hipError_t launch_res2_chain(const Res2ChainParams& p, hipStream_t stream);
hipError_t launch_res2_chain4(Res2ChainParams p, hipStream_t stream);   // res2_chain4.hip (called by launch_res2_chain)

// SE FCs from the GEMM's per-tile column sums (ConvGemmParams::colsum, row tile = 64 rows):
// mean[b][c] = (sum of the tile partials covering rows [b*T, (b+1)*T)) / T, then the two FCs.
// w2t: the second matrix TRANSPOSED, [bottleneck][C] (here and in launch_se_fc_scale_residual)
hipError_t launch_se_fc_from_colsum(const float* colsum, int B, int T, int C, const float* w1,
                                    const float* b1, const float* w2t, const float* b2,
                                    int bottleneck, float* s, hipStream_t stream, const int* lens = nullptr);
// SE_Connect (ecapa_tdnn.py:120-126): s[b][c] = sigmoid(W2 relu(W1 mean_t(y[b,t,:]) + b1) + b2)
hipError_t launch_se_pool_fc(const float* y, int ldy, int B, int T, int C, const float* w1,
                             const float* b1, const float* w2, const float* b2, int bottleneck,
                             float* s, hipStream_t stream, const int* lens = nullptr);
// out[m][o_off + c] = x[m][x_off + c] + y[m][c] * s[b][c]      (SE scale + residual)
// all-binary16 form (f16 back-end): out16 = x16 + y16 * s, 8 halfs per thread
hipError_t launch_se_scale_residual_f16(const uint16_t* x16, int ldx, int x_off, const uint16_t* y16,
                                        int ldy, const float* s, uint16_t* out16, int ldo, int o_off,
                                        int B, int T, int C, hipStream_t stream);
hipError_t launch_se_scale_residual(const float* x, int ldx, int x_off, const float* y, int ldy,
                                    const float* s, float* out, int ldo, int o_off, int B, int T,
                                    int C, hipStream_t stream, uint16_t* out16 = nullptr);
// the SE FCs and the scale + residual pass in one launch (one workgroup per utterance; same bits as the two launches)
bool se_fc_scale_residual_supported(int T, int C, int bottleneck);
// ... and its all-binary16 twin (x / y / out as halfs; f16 back-end)
bool se_fc_scale_residual_f16_supported(int T, int C, int bottleneck);
hipError_t launch_se_fc_scale_residual_f16(const float* colsum, int B, int T, int C, const float* w1, const float* b1,
                                           const float* w2t, const float* b2, int bottleneck, float* s,
                                           const int* lens, const uint16_t* x16, int ldx, int x_off,
                                           const uint16_t* y16, int ldy, uint16_t* out16, int ldo, int o_off,
                                           hipStream_t stream);
hipError_t launch_se_fc_scale_residual(const float* colsum, int B, int T, int C, const float* w1, const float* b1,
                                       const float* w2t, const float* b2, int bottleneck, float* s,
                                       const int* lens, const float* x, int ldx, int x_off, const float* y, int ldy,
                                       float* out, int ldo, int o_off, hipStream_t stream, uint16_t* out16 = nullptr);
// ASTP global context (pooling_layers.py:128-133): per (b, c) mean and sqrt(unbiased var + 1e-7)
// over T of h, then bias_img[b][j] = b1[j] + W1[j][C:2C].mean + W1[j][2C:3C].std
hipError_t launch_astp_std_from_colsum(const float* h, int ldh, int B, int T, int C,
                                       const float* colsum, float* stats, hipStream_t stream,
                                       const int* lens = nullptr);
// the same statistics from column sums + sums of squares (ConvGemmParams::colsumsq): no pass over h
hipError_t launch_astp_std_from_sums(const float* colsum, const float* colsumsq, int B, int T, int C, float* stats,
                                     hipStream_t stream, const float* h = nullptr, int ldh = 0);
// sums of squares of rows [m_begin, M) of a stored layer output, colsum layout (the rows the persistent GEMM left)
hipError_t launch_colsumsq_rows(const float* D, int ldd, int d_off, int m_begin, int M, int HW, int N,
                                float* colsumsq, hipStream_t stream);
hipError_t launch_astp_std_from_colsum_f16(const uint16_t* h16, int ldh, int B, int T, int C,
                                           const float* colsum, float* stats, hipStream_t stream,
                                           const int* lens = nullptr);
hipError_t launch_astp_stats(const float* h, int ldh, int B, int T, int C, float* stats,
                             hipStream_t stream, const int* lens = nullptr);
hipError_t launch_astp_context_bias(const float* h, int ldh, int B, int T, int C, const float* w1,
                                    int ldw1, const float* b1, int bottleneck, float* stats,
                                    float* bias_img, hipStream_t stream);
// ASTP linear1 -> tanh -> linear2 -> softmax over time -> weighted mean / std as one kernel, one workgroup per
// utterance -- or per segment of <= 208 frames of a longer one, + a merge -- (astp_fused.hip; fp32, T >= 64, C = 1536, bottleneck 128).  pooled[b] = [mean(C) | std(C)].
bool astp_fused_supported(int T, int C, int bottleneck);
bool astp_fused_pays(int B, int T);          // cost model: one workgroup per utterance vs the three tile launches
hipError_t launch_astp_fused(const float* h, int ldh, int B, int T, const float* w1, int ldw1, const float* bias,
                             const float* bias_img, const float* w2, int ldw2, float* pooled, const int* lens,
                             hipStream_t stream, float* seg_scratch = nullptr);   // T > 208: B * ceil(T / 208 ..) * 1536 * 4 floats
// Final combine of ConvGemmParams::pool_partial: per (b, c) merge the tile tuples of utterance b,
// mean = S1/S0, std = sqrt(max(S2/S0 - mean^2, 1e-7)) -> pooled[b] = [mean(C) | std(C)]
hipError_t launch_astp_pool_from_partials(const float* partials, int B, int T, int C, float* pooled,
                                          hipStream_t stream);
// ASTP pooling (pooling_layers.py:138-144): softmax over T of logits e, weighted mean / std of h
// -> pooled[b] = [mean(C) | std(C)]
hipError_t launch_astp_pool(const float* e, int lde, const float* h, int ldh, int B, int T, int C,
                            float* pooled, hipStream_t stream, const int* lens = nullptr);

// ResNet / FCM stem: Conv2d(1 -> C, 3x3, pad 1, no bias) + folded BN + ReLU, reading the (B, T, F)
// feature tensor as the (F x T) image and writing channels-last [B][F][T][C].  w: [C][9], b: [C].
hipError_t launch_stem_conv3x3(const float* feats, int B, int T, int F, const float* w,
                               const float* b, int C, float* out, hipStream_t stream,
                               uint16_t* out16 = nullptr, const int* lens = nullptr);
// TSTP (pooling_layers.py:78-85) over channels-last x[(b*F + f)*T + t][c]: mean and
// sqrt(unbiased var + 1e-7) over t; optional pre-activation relu(x*pre_scale[c] + pre_shift[c])
// (CAM++ out_nonlinear).  pooled[b][c*F + f] = mean, pooled[b][C*F + c*F + f] = std.
hipError_t launch_tstp(const float* x, int ldx, int B, int F, int T, int C, const float* pre_scale,
                       const float* pre_shift, float* pooled, hipStream_t stream, const int* lens = nullptr);
hipError_t launch_tstp_f16(const uint16_t* x16, int ldx, int B, int F, int T, int C,
                           const float* pre_scale, const float* pre_shift, float* pooled,
                           hipStream_t stream, const int* lens = nullptr);
// CAM++ context (campplus.py:108-135): ctx = mean_T(h) + segmean_100(h); m = sigmoid(W2 relu(W1 ctx + b1) + b2)
// h: [B*T][C] (C = 128); mask out: [B][segs][Cout]
hipError_t launch_cam_context_from_colsum(const float* colsum, int B, int T, int C, const float* w1,
                                          const float* b1, int hidden, const float* w2, const float* b2,
                                          int Cout, float* mask, hipStream_t stream, const int* lens = nullptr);
hipError_t launch_cam_context(const float* h, int ldh, int B, int T, int C, int seg_len,
                              const float* w1, const float* b1, int hidden, const float* w2,
                              const float* b2, int Cout, float* mask, hipStream_t stream,
                              const int* lens = nullptr);

// One CAM++ dense layer as one kernel (cam_dense.hip): x[:, 0:cin) -> 32 channels appended at c_off of the same buffer.
struct CamDenseParams {
  const float* X; float* Xout; int ldx, cin, c_off;      // [B*Tp][ldx]
  int Tp; const int* lens;                                // trunk frames per utterance (<= 128); valid frames or null
  const float *pre_s, *pre_b;                             // nonlinear1 BN on the layer input [cin]
  const float *W1, *b1; int ldw1;                         // linear1 with nonlinear2's BN folded: [128][ldw1], [128]
  const float* Wl; int ldwl;                              // cam_layer.linear_local: [32][tap * 128 + c]
  const float *cw1, *cb1, *cw2, *cb2;                     // cam_layer.linear1 [64][128], linear2 [32][64]
  int dil;
};
bool cam_dense_fused_applies(int Tp, int cin, int dil);
hipError_t launch_cam_dense_layer(const CamDenseParams& p, int B, hipStream_t stream);
// A whole CAMDenseTDNNBlock (campplus.py:173-205) as ONE launch: layer l reads x[:, 0 : cin0 + 32 l) and appends its 32
// channels, and nothing but the utterance's own rows is read or written -- the workgroup that owns the utterance runs
// the layers back to back.  `base` carries what the layers share (X, ldx, Tp, lens, dil; cin / c_off of layer 0).
constexpr int WS_CAM_MAX_LAYERS = 24;
struct CamDenseLayerW {
  const float *pre_s, *pre_b, *W1, *b1, *Wl, *cw1, *cb1, *cw2, *cb2;
  int ldw1, ldwl;
};
struct CamDenseBlockParams {
  CamDenseParams base;
  int n_layers;
  CamDenseLayerW layers[WS_CAM_MAX_LAYERS];
};
hipError_t launch_cam_dense_block(const CamDenseBlockParams& bp, int B, hipStream_t stream);

// ---- frontend
struct FbankTables {
  const float* window_hamming;   // [frame_len]
  const float* window_povey;     // [frame_len]
  const float* twiddle;          // [fft_n/2] (cos, sin) pairs for the complex FFT of size fft_n/2 + unpack
  const int* mel_start;          // [num_bins] first fft bin
  const int* mel_len;            // [num_bins]
  const int* mel_off;            // [num_bins] offset into mel_w
  const float* mel_w;            // packed weights
  int frame_len, frame_shift, fft_n, num_bins;
  int mel_w_total;               // number of packed weights
  int mel_wpad_total;            // floats of the per-pass zero-padded weight table the kernel builds in LDS
  int mel_pad_reach;             // 1 + the largest power-spectrum index a zero-weight (padded) tap of the 512-point
                                 // kernel reads.  Those reads land in bufB behind P[0..256]: words 257..511 hold the
                                 // frame's OWN third-stage FFT output (stage 2 rewrites all of bufB[0..511] every
                                 // frame, so they are finite for finite input and get multiplied by 0); words 512..519
                                 // (the row padding) are never written.  Hence reach <= 512 = fft_n; a table that
                                 // reaches further -- or a change of the FFT ping-pong that stops rewriting bufB --
                                 // must go through the any-length kernel, which skips taps instead of padding them
};
hipError_t launch_fbank(const FbankTables& t, const void* wav, int wav_dtype, int B, int N,
                        int64_t wav_stride, float scale, int window_type, int T, float* feats,
                        hipStream_t stream, const int* frames = nullptr);
bool fbank_fast_kernel_fits(const FbankTables& t);   // the specialised 512-point kernel takes this frontend
void set_fbank_debug_mode(int mode);   // ws_debug_fbank_mode (0 shipped kernels, 1 packed-fp32 reproducer, 2 any-length kernel for all)
// lens (optional, [B]): valid frames per utterance of a ragged batch -- fbank writes zero rows beyond
// them, CMN averages over / subtracts from the valid rows only
hipError_t launch_cmn(float* feats, int B, int T, int F, hipStream_t stream, const int* lens = nullptr,
                      int mode = 1);   // mode: bit 0 norm_mean, bit 1 norm_var (apply_cmvn, dataset_utils.py:19-26)
hipError_t launch_copy_rows_masked(const float* src, float* dst, int B, int T, int F, const int* lens,
                                   hipStream_t stream);
// diarization sub-segment windows of one segment's fbank (diar/extract_emb.py:55-83): dst [n_windows][window][F]
hipError_t launch_window_gather(const float* feats, int num_frames, int F, int window, int period, int seg_length,
                                int n_windows, float* dst, hipStream_t stream);
hipError_t launch_resample(const float* x, long long n_in, const float* kern, int orig, int nw, int width,
                           float* y, long long n_out, hipStream_t stream);
// binary16 im2col of a k-tap "same" Conv1d over time: out[(b,t)][tap*F + f] = feats[b][t + tap - pad][f]
// (0 outside the utterance), row length ld (>= taps*F, tail zero filled) -- turns the first TDNN layer
// into a plain GEMM for the LDS-DMA kernel
hipError_t launch_im2col_f16(const float* feats, int B, int T, int F, int taps, int pad, uint16_t* out,
                             int ld, hipStream_t stream);
// the same image in fp32 (columns [taps*F, ld) zero): the k5 layer as a plain GEMM for the persistent fp32 kernel
hipError_t launch_im2col_f32(const float* feats, int B, int T, int F, int taps, int pad, float* out, int ld,
                             hipStream_t stream);
// chunk-and-average mode of the native runtime (speaker_engine.cc:83-159)
hipError_t launch_chunk_gather(const float* feats, int total, int F, int cf, int n_full, int n_chunks,
                               float* dst, hipStream_t stream);
hipError_t launch_chunk_average(const float* emb, int n, int E, float* avg, hipStream_t stream);
// *counter += number of non-finite values among x[0..n) (counter: device-visible, system-scope atomics)
hipError_t launch_count_nonfinite(const float* x, long long n, int* counter, hipStream_t stream);

// ---- PLDA (float64)
hipError_t launch_plda_prepare(const void* emb, int emb_is_f64, const int32_t* group_offsets,
                               int n_out, int dim, const double* mean_vec, const double* transform,
                               const double* offset, int pre_norm, int post_norm, double* out,
                               hipStream_t stream);
hipError_t launch_plda_rows(const void* emb, int emb_is_f64, const int32_t* group_offsets, int n_out,
                            int dim, const double* mean_vec, int pre_norm, double* V,
                            hipStream_t stream);
hipError_t launch_plda_rownorm(double* Y, int n, int dim, hipStream_t stream);
// builds the GEMM operands: EA[i] = [g(n_i) * e_i | -0.5 a(n_i)], rowc[i] = K(n_i) - 0.5 sum b e^2.
// n_sessions == nullptr: uniform n -> EA[i] = g * e_i only (K = dim).
hipError_t launch_plda_enroll_terms(const double* enroll, const int32_t* n_sessions, int n_uniform,
                                    int n_enroll, int dim, const double* psi, double* EA,
                                    double* rowc, hipStream_t stream);
// TT[j] = [t_j | t_j^2]
hipError_t launch_plda_test_terms(const double* test, int n_test, int dim, double* TT,
                                  hipStream_t stream);
// uniform n: colc[j] = -1/2 sum_d a_d t_jd^2
hipError_t launch_plda_test_colc(const double* test, int n_test, int dim, int n_uniform,
                                 const double* psi, double* colc, hipStream_t stream);
// out[i][j] = rowc[i] + colc[j] + sum_k EA[i][k] TT[j][k]   (f64 MFMA; rowc / colc may be null)
hipError_t launch_plda_llr_gemm(const double* EA, const double* rowc, const double* colc,
                                int n_enroll, const double* TT, int n_test, int K, double* out,
                                hipStream_t stream);
hipError_t launch_plda_llr_pairs(const double* EA, const double* rowc, const double* colc,
                                 const double* TT, int K, const int32_t* idx_e, const int32_t* idx_t,
                                 int64_t num_trials, double* out, hipStream_t stream);
// one wavefront sampling (shader-clock counter, constant-rate counter) pairs every period_ticks (probes.hip)
hipError_t launch_clock_probe(unsigned long long* out, int samples, unsigned long long period_ticks, hipStream_t stream);
hipError_t launch_row_gather_probe(const double* T, int K, const int32_t* idx, int64_t n, double* out,
                                   hipStream_t stream);   // ws_debug_row_gather (bench yardstick)

// -------- direct 3x3, 32 -> 32 channel convolution on binary16 maps (conv3x3_direct.hip)
bool conv3x3_direct_supported(const ConvGemmParams& p);
hipError_t launch_conv3x3_direct(const ConvGemmParams& p, hipStream_t stream);
// the 32 -> 32 channel layers on the fp32 back-end (exact fp32 MFMA, weights in registers)
bool conv3x3_direct_f32_supported(const ConvGemmParams& p);
hipError_t launch_conv3x3_direct_f32(const ConvGemmParams& p, hipStream_t stream);

// -------- PLDA training statistics (plda_train.hip; two_cov_plda.py:48-66,95-107,261-275)
int64_t plda_stats_scratch_doubles(int n, int dim);
hipError_t launch_plda_stats(const void* emb, int emb_is_f64, int n, int dim, const int32_t* group_offsets,
                             int n_groups, const double* mean_vec, int normalize_length,
                             double* class_mean, double* scatter, double* scratch,
                             hipStream_t stream);

hipError_t launch_rows_affine(const void* x, int x_is_f64, int n, int d_in, const double* sub,
                              const double* M, int d_out, int normalize, double* out,
                              hipStream_t stream);

// -------- cosine scoring + score normalisation (score.hip; bin/score.py, bin/score_norm.py)
hipError_t launch_cos_prepare(const float* emb, const float* mean_vec, int n, int dim, float* unit,
                              float* mag, hipStream_t stream);
hipError_t launch_cos_pairs(const float* ua, const float* ub, int ld, const int32_t* idx_a,
                            const int32_t* idx_b, long long num, float* out, hipStream_t stream);
hipError_t launch_topn_stats(const float* S, int ld, int n_rows, int n_cols, int top_n, float* mean,
                             float* sd, hipStream_t stream);
hipError_t launch_asnorm_pairs(const float* score, const int32_t* idx_e, const int32_t* idx_t,
                               const float* e_mean, const float* e_sd, const float* t_mean,
                               const float* t_sd, long long num, float* out, hipStream_t stream);

}  // namespace wsamd
