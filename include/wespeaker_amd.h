/*
 * wespeaker_amd.h -- C-ABI of the MI355X-native speaker-embedding + PLDA engine.
 *
 * This is the drop-in boundary for the hot path of wenet-e2e/wespeaker
 * (fbank -> ECAPA-TDNN / ResNet / CAM++ forward -> two-covariance PLDA LLR).
 * The reference has no FFI of its own for this path (it is torch.nn.Modules + numpy); each entry
 * point below names the reference interface it replaces (file:line relative to the reference
 * repo).  INTEGRATION.md shows the ctypes / C++ bindings a reference maintainer would add.
 *
 * Conventions:
 *   - extern "C", plain pointers and sizes only; no C++ or torch types cross the boundary.
 *   - every call returns int: 0 = ok, negative = error (never aborts);
 *     ws_last_error() returns a thread-local human-readable message for the last failure.
 *   - the caller owns every input/output buffer; the handle owns weights + workspace.
 *   - pointers marked DEVICE are HIP device pointers on the handle's device; HOST are host pointers.
 *   - all GPU work is enqueued on the caller's stream (a hipStream_t passed as void*; NULL = the
 *     default stream); no call synchronises the device unless documented.
 *   - one handle per (process, GPU); handles are not thread-safe.  Every call that takes a handle makes the
 *     handle's device the calling thread's current HIP device (hipSetDevice) and leaves it so; the handle-less
 *     helpers (ws_cos_*, ws_cohort_stats, ws_asnorm_pairs, ws_plda_stats, ws_rows_affine, ws_resample) launch
 *     on the calling thread's current device: make the device of their DEVICE pointers current first.
 */
#ifndef WESPEAKER_AMD_H_
#define WESPEAKER_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: only the entry points below are exported. */
#if defined(__GNUC__) || defined(__clang__)
#define WS_API __attribute__((visibility("default")))
#else
#define WS_API
#endif

#define WS_OK 0
#define WS_ERR_INVALID_ARG (-1)
#define WS_ERR_UNKNOWN_MODEL (-2)
#define WS_ERR_MISSING_TENSOR (-3)
#define WS_ERR_SHAPE (-4)
#define WS_ERR_HIP (-5)
#define WS_ERR_STATE (-6)
#define WS_ERR_CAPACITY (-7)
#define WS_ERR_RANGE (-8) /* binary16 back-end: an activation left the binary16 range */

#define WS_WINDOW_HAMMING 0 /* Speaker default, cli/speaker.py:50 */
#define WS_WINDOW_POVEY 1   /* Speaker.set_window_type('povey'), cli/speaker.py:66-67 */

#define WS_WAV_INT16 0 /* PCM16 samples (torchaudio.load(normalize=False), cli/speaker.py:126) */
#define WS_WAV_FLOAT32 1

typedef struct ws_engine ws_engine;     /* one speaker-embedding model on one GPU */
typedef struct ws_frontend ws_frontend; /* Kaldi fbank + CMN frontend on one GPU  */
typedef struct ws_plda ws_plda;         /* two-covariance PLDA scorer on one GPU  */
typedef void* ws_stream;                /* hipStream_t */

/* ------------------------------------------------------------------------------------ misc */
/* Bumped whenever an exported signature changes (101: ws_plda_stats takes `emb_is_f64` and a void* table since
 * round 2 -- a caller built against 100 would pass shifted arguments; 104: ws_frontend_set_cmvn / ws_cmvn added, the
 * `cmn` flags now mean "apply the frontend's CMVN"; 105: ws_forward_ragged_cmvn added).  ws_version() returns the library's value:
 * compare it with the header's before calling anything else. */
#define WS_VERSION 105
WS_API int ws_version(void);
WS_API const char* ws_last_error(void);
/* Number of fbank frames for num_samples at snip_edges=True (25 ms / 10 ms):
 * 1 + (N - 400) / 160 at 16 kHz; 0 if N < frame length. */
WS_API int ws_num_frames(int num_samples, int sample_rate);

/* ------------------------------------------------------------------------ host-side wave loader
 * Decode threads of the batch driver (no GPU work): what the reference does in DataLoader worker processes
 * (dataset/processor.py:119-136 parse_raw / read_audio; bin/extract.py:99-103) and in wenet::WavReader
 * (runtime/core/frontend/wav.h:71-117).  RIFF/WAVE 16-bit integer PCM only; channel 0 is kept.
 * ws_wav_probe: per file the number of samples per channel (-1: not readable as PCM16) and the sample rate;
 * returns the number of unreadable files.
 * ws_wav_load_rows: samples [start[i], start[i] + count[i]) of file i (start NULL = 0) into row i of dst
 * (HOST, e.g. pinned; n rows of row_stride int16; the tail of a row is left untouched).  WS_OK or a negative
 * error naming the first file that could not supply its samples.  `threads` std::threads share the files. */
WS_API int ws_wav_probe(const char* const* paths, int n, int threads, int32_t* num_samples, int32_t* sample_rate);
WS_API int ws_wav_load_rows(const char* const* paths, int n, int threads, int16_t* dst, int64_t row_stride,
                     const int32_t* start, const int32_t* count);

/* -------------------------------------------------------------------------------- frontend */
/* Replaces torchaudio.compliance.kaldi.fbank as called at cli/speaker.py:92-97 and
 * dataset/processor.py:518-525 (+ CMN cli/speaker.py:98-99, dataset_utils.py:19-26); native twin
 * runtime/core/frontend/fbank.h:33-97 (constructor: mel banks, window).  dither is always 0.
 * Any sample rate whose 25 ms frame pads to a transform of 16 .. 4096 points (fbank.h:33-52: fft_points =
 * UpperPowerOfTwo(frame_length)): 8 kHz (the SRE recipe, examples/sre/v2/conf/resnet.yaml:31) -> 256, 16 kHz -> 512,
 * 32 kHz -> 1024, 44.1 / 48 kHz -> 2048; 1..128 mel bins; low_freq 20 Hz, high_freq = Nyquist as in the reference's
 * calls.  WS_ERR_INVALID_ARG outside that range. */
WS_API int ws_frontend_create(int sample_rate, int num_mel_bins, int device_id, ws_frontend** out);
WS_API void ws_frontend_destroy(ws_frontend* fe);
/* Which apply_cmvn (dataset/dataset_utils.py:19-26) the frontend's CMVN step is: test_conf['cmvn_args'] of
 * bin/extract.py:124-127.  norm_mean subtracts the per-utterance mean over T from every mel bin; norm_var divides by
 * sqrt(var + 1e-7), var = the unbiased estimate over T (torch.var).  Default (1, 0) = cli/speaker.py:98-99.
 * (0, 0) = `cmvn: False`: no normalisation anywhere.  Applies wherever a call of this frontend normalises:
 * ws_fbank / ws_fbank_ragged with cmn != 0 and the fused ws_extract / ws_extract_ragged. */
WS_API int ws_frontend_set_cmvn(ws_frontend* fe, int norm_mean, int norm_var);
/* wav: DEVICE (B, wav_stride) samples, the first num_samples of each row are used.
 * scale multiplies samples on load (1.0 for int16-range input; 32768.0 reproduces
 * processor.py:516 `waveform * (1 << 15)` for [-1,1] floats).
 * feats: DEVICE (B, T, num_mel_bins) float32, T = ws_num_frames(num_samples).
 * cmn != 0 subtracts the per-utterance mean over T from every mel bin. */
WS_API int ws_fbank(ws_frontend* fe, const void* wav, int wav_dtype, int batch, int num_samples,
             int64_t wav_stride, float scale, int window_type, int cmn, float* feats,
             ws_stream stream);

/* ---------------------------------------------------------------------------------- engine */
/* Replaces get_speaker_model(name)(**model_args) + load_checkpoint (models/speaker_model.py:31-62,
 * utils/checkpoint.py:20-85, cli/speaker.py:306-335).  model_name is the reference constructor
 * name ("ECAPA_TDNN_GLOB_c512", "ECAPA_TDNN_c1024", ...).  Tensors are supplied under the
 * reference's state_dict key names (e.g. "layer2.se_res2block.1.convs.3.weight"); unknown keys
 * ("projection.*", "*.num_batches_tracked") are ignored like strict=False does.  Returns the
 * native twin of runtime/core/speaker/speaker_model.h:25-32 (SpeakerModel). */
WS_API int ws_engine_create(const char* model_name, int feat_dim, int embed_dim, int device_id,
                     ws_engine** out);
/* data: HOST float32, C-contiguous, ndim <= 4.  Returns 1 if the key was consumed, 0 if ignored. */
WS_API int ws_engine_set_tensor(ws_engine* eng, const char* key, const float* data, int ndim,
                         const int64_t* shape);
/* Checks that every required tensor arrived (WS_ERR_MISSING_TENSOR names the first missing key),
 * folds/re-lays-out the weights, uploads them and allocates workspace for max_batch utterances
 * of max_frames frames per forward chunk (larger batches are processed in chunks). */
WS_API int ws_engine_finalize(ws_engine* eng, int max_batch, int max_frames);
/* create + set_tensor x N + finalize from one flat weight file (wespeaker_amd.engine.save_native_model;
 * layout in csrc/c_api.hip): what OnnxSpeakerModel(model_path) is to the reference's native runtime
 * (runtime/core/speaker/onnx_speaker_model.cc:40-75, speaker_engine.cc:28-60) -- a caller without
 * Python hands over a path and gets a ready engine. */
WS_API int ws_engine_load(const char* path, int device_id, int max_batch, int max_frames, ws_engine** out);
/* Re-sizes the workspace of a finalized engine (weights stay): the reference accepts utterances of any
 * length (cli/speaker.py:125-167 has no cap), so callers grow the engine when a longer one arrives.
 * Synchronises the device (earlier launches may still use the old workspace). */
WS_API int ws_engine_reserve(ws_engine* eng, int max_batch, int max_frames);
WS_API int ws_engine_max_batch(const ws_engine* eng);
WS_API int ws_engine_max_frames(const ws_engine* eng);
WS_API void ws_engine_destroy(ws_engine* eng);
WS_API int ws_engine_embed_dim(const ws_engine* eng);
WS_API int ws_engine_feat_dim(const ws_engine* eng);
/* Replaces model(feats)[-1] (cli/speaker.py:163-166; bin/extract.py:133-135):
 * feats DEVICE (B, T, feat_dim) float32 (already CMN'd) -> emb DEVICE (B, embed_dim) float32. */
WS_API int ws_forward(ws_engine* eng, const float* feats, int batch, int num_frames, float* emb,
               ws_stream stream);
/* Fused Speaker.extract_embedding_from_pcm (cli/speaker.py:156-166): wav -> fbank -> CMN ->
 * forward, feature tensor kept inside the engine's workspace. */
WS_API int ws_extract(ws_engine* eng, ws_frontend* fe, const void* wav, int wav_dtype, int batch,
               int num_samples, int64_t wav_stride, float scale, int window_type, float* emb,
               ws_stream stream);
/* ---- ragged batches: utterances of DIFFERENT lengths in one device batch.
 * The reference extracts test sets one utterance at a time (batch_size 1, bin/extract.py:95,
 * cli/speaker.py:169-178) because its modules have no length argument; a batch-1 loop leaves an MI355X
 * launch-bound.  Here utterance b occupies the first num_*[b] entries of its row of the padded
 * (batch, max_*) tensor; the result of every row equals what the batch-1 call returns for it: padding never
 * enters a convolution tap (it is the conv's zero padding), CMN / SE means / attentive and statistics pooling
 * / CAM++ context segments run over the utterance's own frames (pooling_layers.py:78-85,119-144;
 * campplus.py:108-135; ecapa_tdnn.py:120-126).  The length arrays are HOST pointers (read before the call
 * returns).
 * Streams: a ws_frontend may be shared by calls on DIFFERENT streams as long as the calls themselves come from one
 * host thread at a time (no handle is thread-safe).  ws_fbank / ws_extract* only read the frontend's constant tables.
 * ws_fbank_ragged also uploads a per-call frame-count table; those tables live in a ring of 4 slots, a slot is reused
 * only after the kernels that read it have finished (the call then waits on the host for that earlier call), so up to
 * 4 ragged calls may be in flight on any mix of streams. */
WS_API int ws_fbank_ragged(ws_frontend* fe, const void* wav, int wav_dtype, int batch, const int32_t* num_samples,
                    int max_samples, int64_t wav_stride, float scale, int window_type, int cmn, float* feats,
                    ws_stream stream);   /* feats (batch, ws_num_frames(max_samples), bins); rows beyond an utterance's frames = 0 */
WS_API int ws_forward_ragged(ws_engine* eng, const float* feats, int batch, int max_frames, const int32_t* num_frames,
                      float* emb, ws_stream stream);   /* feats (batch, max_frames, feat_dim); padding rows are ignored */
/* ws_forward_ragged on features that are NOT normalised yet: apply_cmvn(norm_mean, norm_var) over every utterance's own
 * frames first (dataset/dataset_utils.py:19-26), on the engine's masked copy -- the caller's tensor is left as it is.
 * The path of `data_type: feat` lists (dataset/processor.py:171-196 parse_feat -> bin/extract.py:112-127): Kaldi
 * feature matrices come from disk, CMVN and the forward run here. */
WS_API int ws_forward_ragged_cmvn(ws_engine* eng, const float* feats, int batch, int max_frames,
                           const int32_t* num_frames, int norm_mean, int norm_var, float* emb, ws_stream stream);
WS_API int ws_extract_ragged(ws_engine* eng, ws_frontend* fe, const void* wav, int wav_dtype, int batch,
                      const int32_t* num_samples, int max_samples, int64_t wav_stride, float scale,
                      int window_type, float* emb, ws_stream stream);
/* Polyphase windowed-sinc resampling of one channel: the arithmetic of
 * torchaudio.transforms.Resample(orig_freq, new_freq) as Speaker.extract_embedding_from_pcm applies it
 * (cli/speaker.py:157-160).  orig / new are the rates divided by their gcd; kernel DEVICE float32
 * [new][2 * width + orig] is the filter bank (host-built: wespeaker_amd.audio.resample_kernel);
 * x DEVICE float32[n_in]; y DEVICE float32[n_out], n_out = ceil(new * n_in / orig). */
WS_API int ws_resample(const float* x, int64_t n_in, const float* kernel, int orig, int new_rate, int width,
                float* y, int64_t n_out, ws_stream stream);

/* Chunk-and-average extraction of ONE utterance, the native runtime's
 * SpeakerEngine::ExtractEmbedding(const int16_t*, int, std::vector<float>* avg_emb)
 * (runtime/core/speaker/speaker_engine.cc:83-159; SamplesPerChunk ctor argument speaker_engine.h:29-31).
 * fbank over the whole waveform; frames cut into chunks of 1 + (samples_per_chunk - 25 ms) / 10 ms
 * frames; a trailing partial chunk is completed with the head frames of the first chunk, an utterance
 * shorter than one chunk is tiled cyclically; per-chunk CMN (ApplyMean); all chunks go through the
 * model as one batch; emb = mean of the chunk embeddings.  samples_per_chunk <= 0 = "full mode" (one
 * chunk holding every frame).  wav DEVICE (num_samples) int16 or float32 as in ws_fbank;
 * emb DEVICE float32[embed_dim].  Returns the number of chunks (>= 1) or a negative WS_ERR_*.
 * Grows an engine-owned scratch on first use / larger input (synchronises then). */
WS_API int ws_extract_chunked(ws_engine* eng, ws_frontend* fe, const void* wav, int wav_dtype,
                       int num_samples, int samples_per_chunk, float scale, int window_type,
                       float* emb, ws_stream stream);

/* Cepstral mean normalisation of features that already live on the device, in place:
 * feats[b, t, :] -= mean_t feats[b, :, :]  (cli/speaker.py:98-99; the per-window form of
 * Speaker.extract_embedding_from_feats(..., subseg_cmn=True), cli/speaker.py:108-112, and
 * diar/extract_emb.py:88-90).  feats DEVICE float32 (batch, num_frames, feat_dim). */
WS_API int ws_cmn(float* feats, int batch, int num_frames, int feat_dim, ws_stream stream);
/* apply_cmvn(feats, norm_mean, norm_var) (dataset/dataset_utils.py:19-26) in place on device features. */
WS_API int ws_cmvn(float* feats, int batch, int num_frames, int feat_dim, int norm_mean, int norm_var,
            ws_stream stream);

/* Diarization sub-segment extraction of ONE speech segment in one call: the loop body of Speaker.diarize
 * (cli/speaker.py:232-251) = compute_features(cmn=False) -> subsegment() (diar/extract_emb.py:55-83) ->
 * extract_embedding_from_feats (cli/speaker.py:108-123).
 * fbank over the whole segment; windows of `window_frames` frames every `period_frames` (150 / 75 for the
 * reference's 1.5 s / 0.75 s at a 10 ms shift), laid out from `seg_length` -- the reference derives it from the
 * segment's time stamps, (end_ms - begin_ms) / frame_shift, which is ws_num_frames() + 2 for its own VAD segments
 * (the comment at extract_emb.py:62-65) -- with every slice clipped to the frames that exist and completed to a
 * full window by tiling its own rows cyclically (np.resize); seg_length <= window_frames: ONE window tiled from
 * the whole fbank.  Optional per-window CMN; all windows go through the model as one batch.
 * ws_num_windows: how many windows that is (0 for a non-positive argument).
 * wav DEVICE (num_samples) int16 or float32 as in ws_fbank; emb DEVICE float32 [max_windows][embed_dim], rows
 * [0, n) are written.  Returns n (>= 1) or a negative WS_ERR_* (WS_ERR_CAPACITY: more windows than max_windows,
 * or window_frames beyond the finalized frame capacity).  Shares the engine-owned scratch of ws_extract_chunked. */
WS_API int ws_num_windows(int seg_length, int window_frames, int period_frames);
WS_API int ws_extract_windows(ws_engine* eng, ws_frontend* fe, const void* wav, int wav_dtype, int num_samples,
                       int seg_length, int window_frames, int period_frames, float scale, int window_type,
                       int subseg_cmn, float* emb, int max_windows, ws_stream stream);

/* Contraction back-end of every conv/linear GEMM (the reference computes them in fp32):
 *   WS_PREC_FP32    exact fp32 products on v_mfma_f32_32x32x2_f32 (default; 157 TF peak)
 *   WS_PREC_F16X3   each fp32 operand split x = hi + lo into two binary16 values (22 significant
 *                   bits) and the product formed as hi*hi + hi*lo + lo*hi on
 *                   v_mfma_f32_32x32x16_f16 with fp32 accumulation: ~2^-21 relative product error,
 *                   i.e. fp32-grade results at 3/16 of the fp32-MFMA issue cost.
 *   WS_PREC_F16     operands rounded to binary16, ONE MFMA pass, fp32 accumulation -- the arithmetic
 *                   of the reference's own GPU deployment (TensorRT fp16, runtime/server/x86_gpu):
 *                   ~5e-4 relative on the embedding, 1 - cos ~ 4e-7 against the fp32 reference
 *                   (the 1e-4 bar holds with > 100x margin); inputs and activations must stay inside
 *                   the binary16 range (65504), which int16-scale fbank + CMN features and BN-normalised
 *                   activations do -- ws_engine_check_range reports a checkpoint that does not.
 * May be switched at any time between forwards. */
#define WS_PREC_FP32 0
#define WS_PREC_F16X3 1
#define WS_PREC_F16 2
WS_API int ws_engine_set_precision(ws_engine* eng, int mode);
/* Range guard of WS_PREC_F16 / WS_PREC_F16X3: a checkpoint whose activations pass 65504 turns them into
 * inf, which reaches the embedding as inf / NaN.  Every forward in those modes ends with a tiny kernel
 * that counts non-finite embedding values into a host-visible counter; this call synchronises `stream`,
 * returns WS_ERR_RANGE (message in ws_last_error) if any were produced since the last call, else WS_OK,
 * and clears the counter.  (The reference's fp32 path has no such limit: switch to WS_PREC_FP32.) */
WS_API int ws_engine_check_range(ws_engine* eng, ws_stream stream);
/* Diagnostic (no reference counterpart): which kernel every distinct conv / linear problem was given.  The
 * dispatcher chooses among a dozen tile shapes and staging forms by shape and operand type; a layer that falls
 * off the fast forms costs speed, never correctness, so nothing else would show it.  mode 0 off, 1 on, 2 on and
 * forget what was noted (also: environment WS_DISPATCH_LOG=1 at load time).  The report is text, one line per
 * distinct (problem, kernel) with its launch count; the call returns the bytes the whole report needs
 * (including the terminator) and writes at most cap of them -- call with (NULL, 0) for the size.
 * tests/golden/dispatch_*.txt pin the tables of the BASELINE models. */
WS_API int ws_debug_dispatch_log(int mode);
WS_API long long ws_debug_dispatch_report(char* buf, long long cap);
/* Gather yardstick of bench.py's PLDA roofline (no reference counterpart): out[p] = sum of the row_len doubles of row
 * idx[p] of `table` -- the row-gather path of ws_plda_llr_pairs (tables resident in L2 / Infinity Cache) with no second
 * operand.  Device pointers; row_len even. */
WS_API int ws_debug_row_gather(const double* table, int row_len, const int32_t* idx, int64_t n, double* out,
                               ws_stream stream);
/* Clock probe of bench.py's sustained leg (no reference counterpart): ONE wavefront that stores `samples` pairs
 * (shader-clock counter, constant-rate counter) into out[2 * samples] (DEVICE), one pair every `period_ticks` ticks of
 * the constant-rate counter, sleeping in between.  Enqueued on a side stream it runs next to whatever the other
 * streams execute and shows the shader clock the chip held under that load.  1..4096 samples, period 1..2^24 ticks. */
WS_API int ws_debug_clock_probe(uint64_t* out, int samples, int64_t period_ticks, ws_stream stream);
/* Reproducer switch of the fbank kernel (process-wide; tests and tools/fbank_race_probe.py only -- DESIGN.md 6.0).
 * The round-3 build of runtime/core/frontend/fbank.h:138-198's arithmetic used the packed-fp32 instruction forms in
 * its power-spectrum loop; next to binary16 GEMMs on another stream those returned wrong values in lanes 48..63.
 * mode 0 = the shipped kernels (no packed-fp32 forms), 1 = the round-3 packed build, 2 = the any-length kernel (the one
 * that serves every rate whose 25 ms frame does not pad to 512 points) also for 512-point frontends: at 16 kHz its
 * outputs are the specialised kernel's bits, which is how the tests pin it.  Returns 0, or WS_ERR_INVALID_ARG for an
 * unknown mode. */
WS_API int ws_debug_fbank_mode(int mode);
/* Algorithmic FLOPs (2 x MACs of every conv/linear) of one forward at (batch, num_frames). */
WS_API double ws_engine_flops(const ws_engine* eng, int batch, int num_frames);

/* Measurement hooks (bench.py roofline leg; no reference counterpart -- the reference only has the
 * wall-clock Timer of runtime/core/utils/timer.h:22-36).  While enabled, every kernel launch of
 * ws_forward/ws_extract is bracketed by HIP events on the launch stream.  ws_engine_profile_read
 * synchronises them and fills 4-element arrays indexed by kernel class
 * (0 = conv/linear GEMM launches with N > 64 -- whichever tile / back-end kernel the dispatcher picks:
 * the dominant class, 1 = narrow GEMMs (N <= 64) and the fused Res2 chain, 2 = reductions / element-wise
 * / frontend, 3 = split-K GEMMs): summed milliseconds, algorithmic FLOPs, algorithmic bytes, launch counts.
 * `on` is a bit mask of the classes to record (0 = off, 0xF = all, 1 = only the dominant GEMM):
 * timing events between kernels cost a few per cent of throughput, so the headline run records
 * only the dominant kernel.  Returns the number of classes. */
WS_API int ws_engine_profile_enable(ws_engine* eng, int on);
WS_API int ws_engine_profile_read(ws_engine* eng, double* ms, double* flops, double* bytes, int* launches);

/* ------------------------------------------------------------------------------------ PLDA */
/* Replaces TwoCovPLDA.load_model's in-memory state (utils/plda/two_cov_plda.py:341-363):
 * mu, psi, offset HOST float64[dim]; transform HOST float64[dim*dim] row-major. */
WS_API int ws_plda_create(int dim, const double* mu, const double* transform, const double* psi,
                   const double* offset, int normalize_length, int device_id, ws_plda** out);
WS_API void ws_plda_destroy(ws_plda* plda);
/* eval_sv pre-processing for enrollment models (two_cov_plda.py:216-235): rows of emb are grouped
 * contiguously, group g = rows [group_offsets[g], group_offsets[g+1]).  Per group: subtract
 * mean_vec (HOST float64[dim] or NULL), mean over rows, length-norm of the mean if
 * normalize_length, transform_embedding.  emb DEVICE float32 (n_rows, dim); group_offsets DEVICE
 * int32[n_groups+1]; out DEVICE float64 (n_groups, dim). */
WS_API int ws_plda_prepare_enroll(ws_plda* plda, const float* emb, const int32_t* group_offsets,
                           int n_groups, const double* mean_vec, double* out, ws_stream stream);
/* eval_sv pre-processing for test utterances (two_cov_plda.py:237-244) == transform_embedding
 * (:156-163) after optional mean subtraction / length-norm.  emb DEVICE float32 (n, dim);
 * out DEVICE float64 (n, dim). */
WS_API int ws_plda_prepare_test(ws_plda* plda, const float* emb, int n, const double* mean_vec,
                         double* out, ws_stream stream);
/* TwoCovPLDA.transform_embedding (two_cov_plda.py:156-163) verbatim: y = transform x + offset, then
 * y *= sqrt(D)/|y| iff normalize_length.  x DEVICE float64 (n, dim); out DEVICE float64 (n, dim). */
WS_API int ws_plda_transform(ws_plda* plda, const double* x, int n, double* out, ws_stream stream);
/* Dense LLR matrix: log_likelihood_ratio (two_cov_plda.py:165-184) for every (enroll i, test j).
 * enroll DEVICE float64 (n_enroll, dim) transformed; n_sessions DEVICE int32[n_enroll] (the `n`
 * argument: 1 if multisession_avg else #utts, :219-222), or NULL when every model has the same
 * count n_uniform > 0 (multisession_avg=True => n_uniform = 1): the score is then one dim-long
 * contraction plus a per-model and a per-test constant; test DEVICE float64 (n_test, dim);
 * out DEVICE float64 (n_enroll, n_test). */
WS_API int ws_plda_llr_matrix(ws_plda* plda, const double* enroll, const int32_t* n_sessions,
                       int n_uniform, int n_enroll, const double* test, int n_test, double* out,
                       ws_stream stream);
/* Explicit trial list (the eval_sv trial loop :246-256): out[p] = LLR(enroll[idx_e[p]],
 * test[idx_t[p]], n_sessions[idx_e[p]]).  idx_* DEVICE int32[num_trials]; out DEVICE float64. */
WS_API int ws_plda_llr_pairs(ws_plda* plda, const double* enroll, const int32_t* n_sessions,
                      int n_uniform, int n_enroll, const double* test, int n_test,
                      const int32_t* idx_e, const int32_t* idx_t, int64_t num_trials, double* out,
                      ws_stream stream);

/* ------------------------------------------------------------------ PLDA training statistics
 * The N x D^2 part of TwoCovPLDA training / adaptation: PldaStats.add_samples over every speaker
 * (two_cov_plda.py:48-66) after the constructor's pre-processing (:95-107: subtract the train-set
 * mean, optional length normalisation), and -- with one group -- the mean / np.cov of adapt() (:261-275).
 * emb DEVICE (n, dim) float32 (emb_is_f64 = 0) or float64 (1: rows that already went through a link of
 * the embedding-processing chain, which the reference keeps in float64), rows grouped by class: class c = rows [group_offsets[c],
 * group_offsets[c+1]); group_offsets DEVICE int32[n_groups+1]; mean_vec DEVICE float64[dim] or NULL;
 * class_mean DEVICE float64 (n_groups, dim) <- mu_c; scatter DEVICE float64 (dim, dim) <-
 * sum_c sum_{i in c} (y_i - mu_c)(y_i - mu_c)^T with y = normalised (x - mean_vec);
 * scratch DEVICE float64[>= ws_plda_stats_scratch(n, dim)].  The D x D algebra of the EM steps
 * (inv / cholesky / eigh, :116-154) stays with the caller in float64. */
WS_API int64_t ws_plda_stats_scratch(int n, int dim);
WS_API int ws_plda_stats(const void* emb, int emb_is_f64, int n, int dim, const int32_t* group_offsets, int n_groups,
                  const double* mean_vec, int normalize_length, double* class_mean, double* scatter,
                  double* scratch, int64_t scratch_doubles, ws_stream stream);

/* One link of the embedding-processing chain (wespeaker/utils/embedding_processing.py: MeanSubtraction
 * :204-216, Length_norm :181-195, Lda.__call__ :177-178) applied to n rows:
 *   y = (x - sub) M ;  if normalize: y /= |y|.
 * x DEVICE (n, d_in) float32 (x_is_f64 = 0) or float64 (1); sub DEVICE float64[d_in] or NULL;
 * M DEVICE float64 (d_in, d_out) row-major or NULL (identity, d_out must equal d_in);
 * out DEVICE float64 (n, d_out).  The LDA / mean statistics come from ws_plda_stats. */
WS_API int ws_rows_affine(const void* x, int x_is_f64, int n, int d_in, const double* sub, const double* M,
                   int d_out, int normalize, double* out, ws_stream stream);

/* ------------------------------------------------------------------ cosine scoring + AS-norm
 * Stateless device functions replacing the numpy/sklearn back-end of wespeaker/bin/score.py:38-72
 * (trials_cosine_score) and wespeaker/bin/score_norm.py:26-36,93-115 (get_mean_std + the
 * normalisation loop).  A "unit table" holds mean-subtracted, L2-normalised embeddings as
 * (ws_cos_table_rows(n), ws_cos_table_ld(dim)) float32, zero padded on both axes. */
WS_API int ws_cos_table_rows(int n);   /* n rounded up to a multiple of 4 */
WS_API int ws_cos_table_ld(int dim);   /* dim rounded up to a multiple of 32 (floats per row) */
/* emb DEVICE float32 (n, dim) dense; mean_vec DEVICE float32[dim] or NULL (score.py:44-53:
 * emb - mean_vec); unit DEVICE table as above (written in full, padding zeroed); mag DEVICE
 * float32[n] or NULL receives |emb - mean_vec| (the enroll_mag/test_mag columns, score_norm.py:107). */
WS_API int ws_cos_prepare(const float* emb, const float* mean_vec, int n, int dim, float* unit, float* mag,
                   ws_stream stream);
/* out[p] = <unit_a[idx_a[p]], unit_b[idx_b[p]]> = cosine_similarity (score.py:62-63).
 * idx_* DEVICE int32[num_trials]; out DEVICE float32[num_trials]. */
WS_API int ws_cos_pairs(const float* unit_a, const float* unit_b, int dim, const int32_t* idx_a,
                 const int32_t* idx_b, int64_t num_trials, float* out, ws_stream stream);
/* Dense cosine matrix out[i][j] = <unit_a[i], unit_b[j]> (score_norm.py:29 np.matmul(emb, cohort.T))
 * on the exact-fp32 MFMA GEMM.  out DEVICE float32 (n_a, ldo), ldo >= ws_cos_table_rows(n_b);
 * columns n_b..ws_cos_table_rows(n_b)-1 receive zeros. */
WS_API int ws_cos_matrix(const float* unit_a, int n_a, const float* unit_b, int n_b, int dim, float* out,
                  int ldo, ws_stream stream);
/* get_mean_std (score_norm.py:26-36): for every row of `unit`, mean and population standard
 * deviation (np.std) of its min(top_n, n_cohort) largest cosine scores against the cohort
 * (asnorm: top_n; snorm: top_n >= n_cohort).  scratch DEVICE float32[scratch_floats] holds score
 * rows in flight; it needs at least 128 * ws_cos_table_rows(n_cohort) floats and the rows are
 * processed in as few chunks as it allows.  mean, sd DEVICE float32[n]. */
WS_API int ws_cohort_stats(const float* unit, int n, const float* unit_cohort, int n_cohort, int dim,
                    int top_n, float* scratch, int64_t scratch_floats, float* mean, float* sd,
                    ws_stream stream);
/* score_norm.py:100-103: out[p] = 0.5 * ((s[p] - e_mean[ie]) / e_sd[ie] + (s[p] - t_mean[it]) / t_sd[it]).
 * All DEVICE; score/out float32[num_trials]. */
WS_API int ws_asnorm_pairs(const float* score, const int32_t* idx_e, const int32_t* idx_t,
                    const float* e_mean, const float* e_sd, const float* t_mean, const float* t_sd,
                    int64_t num_trials, float* out, ws_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* WESPEAKER_AMD_H_ */
