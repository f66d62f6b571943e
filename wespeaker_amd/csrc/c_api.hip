// C-ABI entry points (include/wespeaker_amd.h): error plumbing, the fbank frontend tables,
// engine lifecycle and the PLDA scorer.  Host code only; kernels live in the other .hip files.
#include <cmath>
#include <cstring>

#include "common.h"
#include "kernels.h"

namespace wsamd {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

}  // namespace wsamd

using namespace wsamd;

// ===================================================================================== frontend
struct ws_frontend {
  int sample_rate = 16000, num_bins = 80, device = 0;
  int frame_len = 400, frame_shift = 160, fft_n = 512;
  int cmvn_mode = 1;                 // bit 0 norm_mean, bit 1 norm_var (ws_frontend_set_cmvn); 0 = cmvn: False
  DevBuf window_h, window_p, twiddle, mel_start, mel_len, mel_off, mel_w;
  FbankTables tables;
  // ws_fbank_ragged: per-utterance frame counts (device) + their pinned staging, in a ring of slots.  A slot's `done`
  // event is recorded BEHIND the kernels that read its table, on the stream of that call, and waited for (host side)
  // before the slot is written again -- so consecutive calls may use different streams (a table that a kernel of
  // another stream still reads is never overwritten), and up to FRAME_SLOTS calls are in flight without a host wait.
  static constexpr int FRAME_SLOTS = 4;
  struct FrameSlot {
    DevBuf dev;
    int* pinned = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool used = false;
  } slots[FRAME_SLOTS];
  int next_slot = 0;
  ~ws_frontend() {
    for (auto& sl : slots) {
      if (sl.done) { (void)hipEventSynchronize(sl.done); (void)hipEventDestroy(sl.done); }
      if (sl.pinned) (void)hipHostFree(sl.pinned);
    }
  }
};

struct ws_plda {
  int dim = 0, device = 0, normalize_length = 0;
  DevBuf mu, transform, psi, offset, mean_vec;
  DevBuf EA, rowc, TT, colc;      // GEMM operand scratch (grown on demand)
  DevBuf V;                       // pre-processed rows before the transform GEMM
  size_t cap_e = 0, cap_t = 0, cap_v = 0;
};

extern "C" {

int ws_version(void) { return WS_VERSION; }
const char* ws_last_error(void) { return get_error(); }

int ws_num_frames(int num_samples, int sample_rate) {
  const int flen = (int)(sample_rate * 25.0 * 0.001), fshift = (int)(sample_rate * 10.0 * 0.001);
  if (fshift <= 0 || num_samples < flen) return 0;
  return 1 + (num_samples - flen) / fshift;
}

// ------------------------------------------------------------------------------------ frontend
static int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

int ws_frontend_create(int sample_rate, int num_mel_bins, int device_id, ws_frontend** out) {
  if (!out || sample_rate <= 0 || num_mel_bins <= 0 || num_mel_bins > 128) {
    set_error("ws_frontend_create: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  WS_HIP_CHECK(hipSetDevice(device_id));
  ws_frontend* fe = new ws_frontend;
  fe->sample_rate = sample_rate; fe->num_bins = num_mel_bins; fe->device = device_id;
  fe->frame_len = (int)(sample_rate * 25.0 * 0.001);
  fe->frame_shift = (int)(sample_rate * 10.0 * 0.001);
  fe->fft_n = next_pow2(fe->frame_len);
  // any rate whose 25 ms frame pads to 16 .. 4096 points (640 Hz .. 163 kHz): 8 kHz -> 256 points (the reference's SRE
  // recipe, examples/sre/v2/conf/resnet.yaml:31), 16 kHz -> 512 (the specialised kernel), 32 kHz -> 1024, 44.1 / 48 kHz
  // -> 2048; the transform length follows the frame like runtime/core/frontend/fbank.h:33-52
  if (fe->frame_shift <= 0 || fe->fft_n < 16 || fe->fft_n > 4096) {
    const int got = fe->fft_n;
    delete fe;
    set_error("ws_frontend_create: a 25 ms frame at %d Hz pads to %d points (supported: 16 .. 4096)", sample_rate, got);
    return WS_ERR_INVALID_ARG;
  }
  const int L = fe->frame_len, NF = fe->fft_n;
  // windows (torch.hamming_window / hann_window(periodic=False)^0.85)
  std::vector<float> wh(L), wp(L);
  const double a = 2.0 * M_PI / (L - 1);
  for (int j = 0; j < L; ++j) {
    wh[j] = (float)(0.54 - 0.46 * std::cos(a * j));
    wp[j] = (float)std::pow(0.5 - 0.5 * std::cos(a * j), 0.85);
  }
  // 512-th roots of unity exp(-2 pi i m / 512)
  std::vector<float> tw(2 * NF);
  for (int m = 0; m < NF; ++m) {
    tw[2 * m] = (float)std::cos(2.0 * M_PI * m / NF);
    tw[2 * m + 1] = (float)(-std::sin(2.0 * M_PI * m / NF));
  }
  // mel banks, float32 arithmetic as in torchaudio.compliance.kaldi.get_mel_banks
  const int num_fft_bins = NF / 2;
  const double nyquist = 0.5 * sample_rate, low = 20.0, high = nyquist;
  const double fft_bin_width = (double)sample_rate / NF;
  const double mel_low = 1127.0 * std::log(1.0 + low / 700.0);
  const double mel_high = 1127.0 * std::log(1.0 + high / 700.0);
  const double delta = (mel_high - mel_low) / (num_mel_bins + 1);
  std::vector<int> st(num_mel_bins), ln(num_mel_bins), off(num_mel_bins);
  std::vector<float> wts;
  for (int b = 0; b < num_mel_bins; ++b) {
    const float left = (float)mel_low + (float)b * (float)delta;
    const float center = (float)mel_low + ((float)b + 1.0f) * (float)delta;
    const float right = (float)mel_low + ((float)b + 2.0f) * (float)delta;
    int first = -1, last = -1;
    std::vector<float> row(num_fft_bins, 0.f);
    for (int i = 0; i < num_fft_bins; ++i) {
      const float freq = (float)fft_bin_width * (float)i;
      const float mel = 1127.0f * std::log(1.0f + freq / 700.0f);
      const float up = (mel - left) / (center - left);
      const float down = (right - mel) / (right - center);
      const float wv = std::fmax(0.f, std::fmin(up, down));
      row[i] = wv;
      if (wv > 0.f) { if (first < 0) first = i; last = i; }
    }
    if (first < 0) { first = 0; last = 0; }
    st[b] = first; ln[b] = last - first + 1; off[b] = (int)wts.size();
    for (int i = first; i <= last; ++i) wts.push_back(row[i]);
  }
  auto up = [&](DevBuf& d, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = d.alloc(bytes);
    if (e != hipSuccess) return e;
    return hipMemcpy(d.ptr, src, bytes, hipMemcpyHostToDevice);
  };
  hipError_t e = hipSuccess;
  if ((e = up(fe->window_h, wh.data(), wh.size() * 4)) != hipSuccess ||
      (e = up(fe->window_p, wp.data(), wp.size() * 4)) != hipSuccess ||
      (e = up(fe->twiddle, tw.data(), tw.size() * 4)) != hipSuccess ||
      (e = up(fe->mel_start, st.data(), st.size() * 4)) != hipSuccess ||
      (e = up(fe->mel_len, ln.data(), ln.size() * 4)) != hipSuccess ||
      (e = up(fe->mel_off, off.data(), off.size() * 4)) != hipSuccess ||
      (e = up(fe->mel_w, wts.data(), wts.size() * 4)) != hipSuccess) {
    delete fe;
    set_error("ws_frontend_create: device upload failed: %s", hipGetErrorString(e));
    return WS_ERR_HIP;
  }
  FbankTables& t = fe->tables;
  t.window_hamming = fe->window_h.as<float>(); t.window_povey = fe->window_p.as<float>();
  t.twiddle = fe->twiddle.as<float>();
  t.mel_start = fe->mel_start.as<int>(); t.mel_len = fe->mel_len.as<int>();
  t.mel_off = fe->mel_off.as<int>(); t.mel_w = fe->mel_w.as<float>();
  t.frame_len = L; t.frame_shift = fe->frame_shift; t.fft_n = NF; t.num_bins = num_mel_bins;
  t.mel_w_total = (int)wts.size();
  {   // the kernel's per-pass zero-padded weight table (fbank_kernel.inc): its size and how far its padded taps reach
    int total = 0, reach = 0;
    for (int ps = 0; ps * 16 < num_mel_bins; ++ps) {
      int mx = 0;
      for (int q = 0; q < 16 && ps * 16 + q < num_mel_bins; ++q) mx = std::max(mx, ln[ps * 16 + q]);
      const int its = (((mx + 3) >> 2) + 1) & ~1;
      total += its * 64;
      for (int q = 0; q < 16 && ps * 16 + q < num_mel_bins; ++q) reach = std::max(reach, st[ps * 16 + q] + 4 * its);
    }
    t.mel_wpad_total = total; t.mel_pad_reach = reach;
  }
  // (a table the specialised 512-point kernel cannot hold -- more mel bins than its padded weights allow -- is not an
  // error any more: launch_fbank sends such a frontend through the any-length kernel)
  *out = fe;
  return WS_OK;
}

void ws_frontend_destroy(ws_frontend* fe) { delete fe; }

int ws_frontend_set_cmvn(ws_frontend* fe, int norm_mean, int norm_var) {
  if (!fe) { set_error("ws_frontend_set_cmvn: invalid argument"); return WS_ERR_INVALID_ARG; }
  fe->cmvn_mode = (norm_mean ? 1 : 0) | (norm_var ? 2 : 0);
  return WS_OK;
}

int ws_fbank(ws_frontend* fe, const void* wav, int wav_dtype, int batch, int num_samples,
             int64_t wav_stride, float scale, int window_type, int cmn, float* feats,
             ws_stream stream) {
  if (!fe || !wav || !feats || batch < 0 || (wav_dtype != WS_WAV_INT16 && wav_dtype != WS_WAV_FLOAT32) ||
      (window_type != WS_WINDOW_HAMMING && window_type != WS_WINDOW_POVEY) ||
      wav_stride < num_samples) {
    set_error("ws_fbank: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  const int T = ws_num_frames(num_samples, fe->sample_rate);
  if (T == 0 || batch == 0) return WS_OK;          // shorter than one frame: empty output
  hipStream_t st = (hipStream_t)stream;
  WS_HIP_CHECK(hipSetDevice(fe->device));
  WS_HIP_CHECK(launch_fbank(fe->tables, wav, wav_dtype, batch, num_samples, wav_stride, scale,
                            window_type, T, feats, st));
  if (cmn) WS_HIP_CHECK(launch_cmn(feats, batch, T, fe->num_bins, st, nullptr, fe->cmvn_mode));
  return WS_OK;
}

int ws_fbank_ragged(ws_frontend* fe, const void* wav, int wav_dtype, int batch, const int32_t* num_samples,
                    int max_samples, int64_t wav_stride, float scale, int window_type, int cmn, float* feats,
                    ws_stream stream) {
  if (!fe || !wav || !feats || !num_samples || batch < 0 ||
      (wav_dtype != WS_WAV_INT16 && wav_dtype != WS_WAV_FLOAT32) ||
      (window_type != WS_WINDOW_HAMMING && window_type != WS_WINDOW_POVEY) || wav_stride < max_samples) {
    set_error("ws_fbank_ragged: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  const int T = ws_num_frames(max_samples, fe->sample_rate);
  if (T == 0 || batch == 0) return WS_OK;
  hipStream_t st = (hipStream_t)stream;
  WS_HIP_CHECK(hipSetDevice(fe->device));
  // (validated BEFORE a ring slot is taken: a refused call must not advance the ring or touch a slot's table)
  for (int b = 0; b < batch; ++b) {
    if (num_samples[b] < 0 || num_samples[b] > max_samples) {
      set_error("ws_fbank_ragged: utterance %d has %d samples, max_samples is %d", b, num_samples[b], max_samples);
      return WS_ERR_INVALID_ARG;
    }
  }
  ws_frontend::FrameSlot& sl = fe->slots[fe->next_slot];
  if (!sl.done) WS_HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  if (sl.used) WS_HIP_CHECK(hipEventSynchronize(sl.done));     // the kernels that read this slot's table have finished
  if ((size_t)batch > sl.cap) {
    if (sl.pinned) (void)hipHostFree(sl.pinned);
    sl.pinned = nullptr; sl.cap = 0;
    const size_t cap = (size_t)batch + batch / 2 + 64;
    WS_HIP_CHECK(sl.dev.alloc(cap * sizeof(int)));
    WS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&sl.pinned), cap * sizeof(int), 0));
    sl.cap = cap;
  }
  fe->next_slot = (fe->next_slot + 1) % ws_frontend::FRAME_SLOTS;
  for (int b = 0; b < batch; ++b) sl.pinned[b] = ws_num_frames(num_samples[b], fe->sample_rate);
  // From the copy on, the slot's pinned / device tables may be in use by the stream: whatever fails below, the event
  // is recorded behind everything that was enqueued and the slot is marked used, so that its next user waits for it
  // (a failed launch used to leave `done` unrecorded: the next reuse could rewrite the table under a copy in flight)
  hipError_t he = hipMemcpyAsync(sl.dev.ptr, sl.pinned, (size_t)batch * sizeof(int), hipMemcpyHostToDevice, st);
  if (he == hipSuccess)
    he = launch_fbank(fe->tables, wav, wav_dtype, batch, max_samples, wav_stride, scale, window_type, T, feats, st,
                      sl.dev.as<int>());
  if (he == hipSuccess && cmn) he = launch_cmn(feats, batch, T, fe->num_bins, st, sl.dev.as<int>(), fe->cmvn_mode);
  const hipError_t re = hipEventRecord(sl.done, st);
  sl.used = true;
  if (re != hipSuccess) (void)hipStreamSynchronize(st);        // no event to wait for: drain the stream instead
  WS_HIP_CHECK(he);
  WS_HIP_CHECK(re);
  return WS_OK;
}

// -------------------------------------------------------------------------------------- engine
int ws_engine_create(const char* model_name, int feat_dim, int embed_dim, int device_id,
                     ws_engine** out) {
  if (!model_name || !out || feat_dim <= 0 || embed_dim <= 0 || (feat_dim & 3)) {
    set_error("ws_engine_create: invalid argument (feat_dim must be a positive multiple of 4)");
    return WS_ERR_INVALID_ARG;
  }
  WS_HIP_CHECK(hipSetDevice(device_id));
  Model* m = make_ecapa(model_name, feat_dim, embed_dim);
  if (!m) m = make_resnet(model_name, feat_dim, embed_dim);
  if (!m) m = make_campplus(model_name, feat_dim, embed_dim);
  if (!m) {
    set_error("ws_engine_create: unknown model '%s'", model_name);
    return WS_ERR_UNKNOWN_MODEL;
  }
  ws_engine* e = new ws_engine;
  e->model_name = model_name; e->feat_dim = feat_dim; e->embed_dim = embed_dim;
  e->device = device_id; e->model = m;
  *out = e;
  return WS_OK;
}

int ws_engine_set_tensor(ws_engine* eng, const char* key, const float* data, int ndim,
                         const int64_t* shape) {
  if (!eng || !key || !data || ndim < 0 || ndim > 4 || (ndim && !shape)) {
    set_error("ws_engine_set_tensor: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (eng->finalized) {
    set_error("ws_engine_set_tensor: engine already finalized");
    return WS_ERR_STATE;
  }
  if (!eng->model->wants(key)) return 0;     // strict=False: unknown keys are ignored
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(data, data + t.numel());
  eng->sd[key] = std::move(t);
  return 1;
}

int ws_engine_finalize(ws_engine* eng, int max_batch, int max_frames) {
  if (!eng || max_batch <= 0 || max_frames <= 0) {
    set_error("ws_engine_finalize: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (eng->finalized) { set_error("ws_engine_finalize: already finalized"); return WS_ERR_STATE; }
  WS_HIP_CHECK(hipSetDevice(eng->device));
  int r = eng->model->finalize(eng->sd, max_batch, max_frames);
  if (r) return r;
  eng->sd.clear();
  eng->finalized = true;
  return WS_OK;
}

// Flat weight file written by wespeaker_amd.engine.save_native_model (little endian):
//   "WSAMDW01" | i32 name_len | name | i32 feat_dim | i32 embed_dim | i32 n_tensors |
//   n_tensors x ( i32 key_len | key | i32 ndim | i64 shape[ndim] | f32 data[prod(shape)] )
int ws_engine_load(const char* path, int device_id, int max_batch, int max_frames, ws_engine** out) {
  if (!path || !out || max_batch <= 0 || max_frames <= 0) {
    set_error("ws_engine_load: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  FILE* f = std::fopen(path, "rb");
  if (!f) { set_error("ws_engine_load: cannot open '%s'", path); return WS_ERR_INVALID_ARG; }
  ws_engine* eng = nullptr;
  int rc = WS_OK;
  auto fail = [&](int code, const char* what) {
    set_error("ws_engine_load: %s in '%s'", what, path);
    rc = code;
  };
  auto rd = [&](void* dst, size_t n) { return std::fread(dst, 1, n, f) == n; };
  auto rd_str = [&](std::string* s) {
    int32_t n = 0;
    if (!rd(&n, 4) || n < 0 || n > 4096) return false;
    s->resize(n);
    return n == 0 || rd(&(*s)[0], n);
  };
  char magic[8];
  std::string name;
  int32_t feat_dim = 0, embed_dim = 0, n_tensors = 0;
  if (!rd(magic, 8) || std::memcmp(magic, "WSAMDW01", 8) != 0) fail(WS_ERR_INVALID_ARG, "bad magic");
  else if (!rd_str(&name) || !rd(&feat_dim, 4) || !rd(&embed_dim, 4) || !rd(&n_tensors, 4) || n_tensors < 0)
    fail(WS_ERR_INVALID_ARG, "truncated header");
  if (rc == WS_OK) rc = ws_engine_create(name.c_str(), feat_dim, embed_dim, device_id, &eng);
  std::vector<float> data;
  for (int t = 0; rc == WS_OK && t < n_tensors; ++t) {
    std::string key;
    int32_t ndim = 0;
    int64_t shape[4] = {0, 0, 0, 0};
    if (!rd_str(&key) || !rd(&ndim, 4) || ndim < 0 || ndim > 4 || !rd(shape, 8 * (size_t)ndim)) {
      fail(WS_ERR_INVALID_ARG, "truncated tensor header");
      break;
    }
    // every dimension checked on its own and the product before it is formed: a negative pair or a wrapped
    // product must not pass as a small positive element count
    int64_t numel = 1;
    bool shape_ok = true;
    for (int d = 0; d < ndim && shape_ok; ++d) {
      if (shape[d] < 0 || shape[d] > ((int64_t)1 << 31) || (shape[d] > 0 && numel > ((int64_t)1 << 31) / shape[d]))
        shape_ok = false;
      else
        numel *= shape[d];
    }
    if (!shape_ok) { fail(WS_ERR_SHAPE, "implausible tensor size"); break; }
    data.resize((size_t)numel);
    if (numel && !rd(data.data(), 4 * (size_t)numel)) { fail(WS_ERR_INVALID_ARG, "truncated tensor data"); break; }
    const int r = ws_engine_set_tensor(eng, key.c_str(), data.data(), ndim, shape);
    if (r < 0) rc = r;
  }
  std::fclose(f);
  if (rc == WS_OK) rc = ws_engine_finalize(eng, max_batch, max_frames);
  if (rc != WS_OK) {
    ws_engine_destroy(eng);
    return rc;
  }
  *out = eng;
  return WS_OK;
}

int ws_engine_reserve(ws_engine* eng, int max_batch, int max_frames) {
  if (!eng || max_batch <= 0 || max_frames <= 0) {
    set_error("ws_engine_reserve: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (!eng->finalized) { set_error("ws_engine_reserve: engine not finalized"); return WS_ERR_STATE; }
  WS_HIP_CHECK(hipSetDevice(eng->device));
  return eng->model->reserve(max_batch, max_frames);
}

int ws_engine_max_batch(const ws_engine* eng) {
  return eng && eng->finalized ? eng->model->max_batch() : WS_ERR_INVALID_ARG;
}
int ws_engine_max_frames(const ws_engine* eng) {
  return eng && eng->finalized ? eng->model->max_frames() : WS_ERR_INVALID_ARG;
}

void ws_engine_destroy(ws_engine* eng) {
  if (!eng) return;
  delete eng->model;
  delete eng;
}

int ws_engine_embed_dim(const ws_engine* eng) { return eng ? eng->embed_dim : WS_ERR_INVALID_ARG; }
int ws_engine_feat_dim(const ws_engine* eng) { return eng ? eng->feat_dim : WS_ERR_INVALID_ARG; }

int ws_forward(ws_engine* eng, const float* feats, int batch, int num_frames, float* emb,
               ws_stream stream) {
  if (!eng || !feats || !emb || batch < 0 || num_frames <= 0) {
    set_error("ws_forward: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (!eng->finalized) { set_error("ws_forward: engine not finalized"); return WS_ERR_STATE; }
  if (batch == 0) return WS_OK;
  WS_HIP_CHECK(hipSetDevice(eng->device));
  return eng->model->forward(feats, batch, num_frames, emb, (hipStream_t)stream);
}

int ws_extract(ws_engine* eng, ws_frontend* fe, const void* wav, int wav_dtype, int batch,
               int num_samples, int64_t wav_stride, float scale, int window_type, float* emb,
               ws_stream stream) {
  if (!eng || !fe || !wav || !emb || batch < 0) {
    set_error("ws_extract: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (!eng->finalized) { set_error("ws_extract: engine not finalized"); return WS_ERR_STATE; }
  if (fe->num_bins != eng->feat_dim) {
    set_error("ws_extract: frontend has %d mel bins, model expects %d", fe->num_bins, eng->feat_dim);
    return WS_ERR_SHAPE;
  }
  const int T = ws_num_frames(num_samples, fe->sample_rate);
  if (T <= 0) { set_error("ws_extract: utterance shorter than one frame"); return WS_ERR_INVALID_ARG; }
  if (T > eng->model->max_frames()) {
    set_error("ws_extract: %d frames exceed the finalized capacity %d", T, eng->model->max_frames());
    return WS_ERR_CAPACITY;
  }
  const int chunk = eng->model->max_batch();
  const size_t esz = wav_dtype == WS_WAV_INT16 ? 2 : 4;
  float* fw = eng->model->feats_workspace();
  WS_HIP_CHECK(hipSetDevice(eng->device));
  for (int b0 = 0; b0 < batch; b0 += chunk) {
    const int nb = batch - b0 < chunk ? batch - b0 : chunk;
    const char* w = reinterpret_cast<const char*>(wav) + (size_t)b0 * wav_stride * esz;
    int r = ws_fbank(fe, w, wav_dtype, nb, num_samples, wav_stride, scale, window_type, 1, fw, stream);
    if (r) return r;
    r = eng->model->forward(fw, nb, T, emb + (size_t)b0 * eng->embed_dim, (hipStream_t)stream);
    if (r) return r;
  }
  return WS_OK;
}

int ws_forward_ragged(ws_engine* eng, const float* feats, int batch, int max_frames, const int32_t* num_frames,
                      float* emb, ws_stream stream) {
  if (!eng || !feats || !emb || !num_frames || batch < 0 || max_frames <= 0) {
    set_error("ws_forward_ragged: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (!eng->finalized) { set_error("ws_forward_ragged: engine not finalized"); return WS_ERR_STATE; }
  if (batch == 0) return WS_OK;
  WS_HIP_CHECK(hipSetDevice(eng->device));
  return eng->model->forward_ragged(feats, batch, max_frames, num_frames, emb, (hipStream_t)stream);
}

int ws_forward_ragged_cmvn(ws_engine* eng, const float* feats, int batch, int max_frames, const int32_t* num_frames,
                           int norm_mean, int norm_var, float* emb, ws_stream stream) {
  if (!eng || !feats || !emb || !num_frames || batch < 0 || max_frames <= 0) {
    set_error("ws_forward_ragged_cmvn: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (!eng->finalized) { set_error("ws_forward_ragged_cmvn: engine not finalized"); return WS_ERR_STATE; }
  if (batch == 0) return WS_OK;
  WS_HIP_CHECK(hipSetDevice(eng->device));
  return eng->model->forward_ragged(feats, batch, max_frames, num_frames, emb, (hipStream_t)stream,
                                    (norm_mean ? 1 : 0) | (norm_var ? 2 : 0));
}

int ws_extract_ragged(ws_engine* eng, ws_frontend* fe, const void* wav, int wav_dtype, int batch,
                      const int32_t* num_samples, int max_samples, int64_t wav_stride, float scale,
                      int window_type, float* emb, ws_stream stream) {
  if (!eng || !fe || !wav || !emb || !num_samples || batch < 0 || wav_stride < max_samples ||
      (wav_dtype != WS_WAV_INT16 && wav_dtype != WS_WAV_FLOAT32) ||
      (window_type != WS_WINDOW_HAMMING && window_type != WS_WINDOW_POVEY)) {
    set_error("ws_extract_ragged: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (!eng->finalized) { set_error("ws_extract_ragged: engine not finalized"); return WS_ERR_STATE; }
  if (fe->num_bins != eng->feat_dim) {
    set_error("ws_extract_ragged: frontend has %d mel bins, model expects %d", fe->num_bins, eng->feat_dim);
    return WS_ERR_SHAPE;
  }
  if (batch == 0) return WS_OK;
  const int T = ws_num_frames(max_samples, fe->sample_rate);
  if (T <= 0) { set_error("ws_extract_ragged: max_samples shorter than one frame"); return WS_ERR_INVALID_ARG; }
  if (T > eng->model->max_frames()) {
    set_error("ws_extract_ragged: %d frames exceed the finalized capacity %d", T, eng->model->max_frames());
    return WS_ERR_CAPACITY;
  }
  hipStream_t st = (hipStream_t)stream;
  WS_HIP_CHECK(hipSetDevice(eng->device));
  std::vector<int32_t> frames(batch);
  for (int b = 0; b < batch; ++b) {
    if (num_samples[b] < 0 || num_samples[b] > max_samples) {
      set_error("ws_extract_ragged: utterance %d has %d samples, max_samples is %d", b, num_samples[b], max_samples);
      return WS_ERR_INVALID_ARG;
    }
    frames[b] = ws_num_frames(num_samples[b], fe->sample_rate);
  }
  // one table for the whole call: fbank / CMN read level 0, the model every level (upload_lens rejects an
  // utterance shorter than the model's minimum)
  const int* lens = eng->model->upload_lens(frames.data(), batch, T, st);
  if (!lens) return WS_ERR_INVALID_ARG;
  const int chunk = eng->model->max_batch();
  const size_t esz = wav_dtype == WS_WAV_INT16 ? 2 : 4;
  float* fw = eng->model->feats_workspace();
  for (int b0 = 0; b0 < batch; b0 += chunk) {
    const int nb = batch - b0 < chunk ? batch - b0 : chunk;
    const char* w = reinterpret_cast<const char*>(wav) + (size_t)b0 * wav_stride * esz;
    WS_HIP_CHECK(launch_fbank(fe->tables, w, wav_dtype, nb, max_samples, wav_stride, scale, window_type, T, fw,
                              st, lens + b0));
    WS_HIP_CHECK(launch_cmn(fw, nb, T, fe->num_bins, st, lens + b0, fe->cmvn_mode));
    int r = eng->model->forward_chunk_ragged(fw, nb, T, lens, batch, b0, emb + (size_t)b0 * eng->embed_dim, st);
    if (r) return r;
  }
  return eng->model->finish_forward(emb, batch, st);
}

int ws_resample(const float* x, int64_t n_in, const float* kernel, int orig, int new_rate, int width,
                float* y, int64_t n_out, ws_stream stream) {
  if (n_out == 0) return WS_OK;
  if (!x || !kernel || !y || n_in <= 0 || n_out < 0 || orig <= 0 || new_rate <= 0 || width < 0) {
    set_error("ws_resample: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  WS_HIP_CHECK(launch_resample(x, n_in, kernel, orig, new_rate, width, y, n_out, (hipStream_t)stream));
  return WS_OK;
}

int ws_extract_chunked(ws_engine* eng, ws_frontend* fe, const void* wav, int wav_dtype,
                       int num_samples, int samples_per_chunk, float scale, int window_type,
                       float* emb, ws_stream stream) {
  if (!eng || !fe || !wav || !emb) {
    set_error("ws_extract_chunked: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (!eng->finalized) { set_error("ws_extract_chunked: engine not finalized"); return WS_ERR_STATE; }
  if (fe->num_bins != eng->feat_dim) {
    set_error("ws_extract_chunked: frontend has %d mel bins, model expects %d", fe->num_bins,
              eng->feat_dim);
    return WS_ERR_SHAPE;
  }
  const int F = fe->num_bins, E = eng->embed_dim;
  const int total = ws_num_frames(num_samples, fe->sample_rate);
  if (total <= 0) {
    set_error("ws_extract_chunked: utterance shorter than one frame");
    return WS_ERR_INVALID_ARG;
  }
  // speaker_engine.cc:101-103 (integer arithmetic as written there)
  int cf = total, n_full = 1, n_chunks = 1;
  if (samples_per_chunk > 0) {
    const int ms = fe->sample_rate / 1000;
    cf = 1 + (samples_per_chunk - ms * 25) / (ms * 10);
    if (samples_per_chunk < ms * 25 || cf <= 0) {
      set_error("ws_extract_chunked: samples_per_chunk %d is shorter than one frame", samples_per_chunk);
      return WS_ERR_INVALID_ARG;
    }
    n_full = total / cf;
    n_chunks = n_full + (total % cf ? 1 : 0);
  }
  if (cf > eng->model->max_frames()) {
    set_error("ws_extract_chunked: %d frames per chunk exceed the finalized capacity %d", cf,
              eng->model->max_frames());
    return WS_ERR_CAPACITY;
  }
  hipStream_t st = (hipStream_t)stream;
  WS_HIP_CHECK(hipSetDevice(eng->device));
  // scratch: [total][F] utterance feats | [n_chunks][cf][F] chunk tensor | [n_chunks][E] embeddings
  const size_t n_feats = ((size_t)total * F + 3) & ~size_t(3);
  const size_t n_chunk = (size_t)n_chunks * cf * F;
  const size_t need = (n_feats + n_chunk + (size_t)n_chunks * E) * sizeof(float);
  if (eng->chunk_scratch.bytes < need) {
    WS_HIP_CHECK(hipStreamSynchronize(st));       // earlier calls may still read the old buffer
    WS_HIP_CHECK(eng->chunk_scratch.alloc(need + need / 2));
  }
  float* feats = eng->chunk_scratch.as<float>();
  float* chunks = feats + n_feats;
  float* cemb = chunks + n_chunk;
  int r = ws_fbank(fe, wav, wav_dtype, 1, num_samples, num_samples, scale, window_type, 0, feats, stream);
  if (r) return r;
  WS_HIP_CHECK(launch_chunk_gather(feats, total, F, cf, samples_per_chunk > 0 ? n_full : 1, n_chunks,
                                   chunks, st));
  WS_HIP_CHECK(launch_cmn(chunks, n_chunks, cf, F, st));
  r = eng->model->forward(chunks, n_chunks, cf, cemb, st);
  if (r) return r;
  WS_HIP_CHECK(launch_chunk_average(cemb, n_chunks, E, emb, st));
  return n_chunks;
}

int ws_cmvn(float* feats, int batch, int num_frames, int feat_dim, int norm_mean, int norm_var, ws_stream stream) {
  if (!feats || batch < 0 || num_frames < 0 || feat_dim <= 0) { set_error("ws_cmvn: invalid argument"); return WS_ERR_INVALID_ARG; }
  if (batch == 0 || num_frames == 0) return WS_OK;
  WS_HIP_CHECK(launch_cmn(feats, batch, num_frames, feat_dim, (hipStream_t)stream, nullptr,
                          (norm_mean ? 1 : 0) | (norm_var ? 2 : 0)));
  return WS_OK;
}

int ws_cmn(float* feats, int batch, int num_frames, int feat_dim, ws_stream stream) {
  if (!feats || batch < 0 || num_frames < 0 || feat_dim <= 0) { set_error("ws_cmn: invalid argument"); return WS_ERR_INVALID_ARG; }
  if (batch == 0 || num_frames == 0) return WS_OK;
  WS_HIP_CHECK(launch_cmn(feats, batch, num_frames, feat_dim, (hipStream_t)stream));
  return WS_OK;
}

int ws_num_windows(int seg_length, int window_frames, int period_frames) {
  if (seg_length <= 0 || window_frames <= 0 || period_frames <= 0) return 0;
  if (seg_length <= window_frames) return 1;
  // len(range(0, seg_length - window + period, period))   (diar/extract_emb.py:74-75)
  return (seg_length - window_frames + period_frames + period_frames - 1) / period_frames;
}

int ws_extract_windows(ws_engine* eng, ws_frontend* fe, const void* wav, int wav_dtype, int num_samples,
                       int seg_length, int window_frames, int period_frames, float scale, int window_type,
                       int subseg_cmn, float* emb, int max_windows, ws_stream stream) {
  if (!eng || !fe || !wav || !emb || window_frames <= 0 || period_frames <= 0 || seg_length <= 0) {
    set_error("ws_extract_windows: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (!eng->finalized) { set_error("ws_extract_windows: engine not finalized"); return WS_ERR_STATE; }
  if (fe->num_bins != eng->feat_dim) {
    set_error("ws_extract_windows: frontend has %d mel bins, model expects %d", fe->num_bins, eng->feat_dim);
    return WS_ERR_SHAPE;
  }
  const int F = fe->num_bins, E = eng->embed_dim;
  const int total = ws_num_frames(num_samples, fe->sample_rate);
  if (total <= 0) { set_error("ws_extract_windows: segment shorter than one frame"); return WS_ERR_INVALID_ARG; }
  if (window_frames > eng->model->max_frames()) {
    set_error("ws_extract_windows: %d frames per window exceed the finalized capacity %d", window_frames,
              eng->model->max_frames());
    return WS_ERR_CAPACITY;
  }
  const int n_win = ws_num_windows(seg_length, window_frames, period_frames);
  if (n_win > max_windows) {
    set_error("ws_extract_windows: %d windows, the output holds %d", n_win, max_windows);
    return WS_ERR_CAPACITY;
  }
  hipStream_t st = (hipStream_t)stream;
  WS_HIP_CHECK(hipSetDevice(eng->device));
  // scratch: [total][F] segment feats | [n_win][window][F] window tensor
  const size_t n_feats = ((size_t)total * F + 3) & ~size_t(3);
  const size_t need = (n_feats + (size_t)n_win * window_frames * F) * sizeof(float);
  if (eng->chunk_scratch.bytes < need) {
    WS_HIP_CHECK(hipStreamSynchronize(st));       // earlier calls may still read the old buffer
    WS_HIP_CHECK(eng->chunk_scratch.alloc(need + need / 2));
  }
  float* feats = eng->chunk_scratch.as<float>();
  float* wins = feats + n_feats;
  int r = ws_fbank(fe, wav, wav_dtype, 1, num_samples, num_samples, scale, window_type, 0, feats, stream);
  if (r) return r;
  WS_HIP_CHECK(launch_window_gather(feats, total, F, window_frames, period_frames, seg_length, n_win, wins, st));
  if (subseg_cmn) WS_HIP_CHECK(launch_cmn(wins, n_win, window_frames, F, st));
  r = eng->model->forward(wins, n_win, window_frames, emb, st);
  (void)E;
  if (r) return r;
  return n_win;
}

int ws_engine_set_precision(ws_engine* eng, int mode) {
  if (!eng) { set_error("ws_engine_set_precision: invalid argument"); return WS_ERR_INVALID_ARG; }
  int r = eng->model->set_precision(mode);
  if (r) set_error("ws_engine_set_precision: unknown mode %d (0 = fp32 MFMA, 1 = split-f16 x3, 2 = f16)", mode);
  return r;
}

int ws_engine_check_range(ws_engine* eng, ws_stream stream) {
  if (!eng || !eng->finalized) { set_error("ws_engine_check_range: invalid argument"); return WS_ERR_INVALID_ARG; }
  WS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  const int n = eng->model->take_nonfinite();
  if (n > 0) {
    set_error("binary16 range exceeded: %d non-finite embedding values since the last check (an "
              "activation passed 65504 in the f16 / f16x3 back-end; use WS_PREC_FP32 for this model)", n);
    return WS_ERR_RANGE;
  }
  return eng->model->check_invariants();
}

int ws_debug_dispatch_log(int mode) {
  if (mode < 0 || mode > 2) { set_error("ws_debug_dispatch_log: mode %d (0 off, 1 on, 2 on + clear)", mode); return WS_ERR_INVALID_ARG; }
  dispatch_log_enable(mode != 0);
  if (mode == 2) dispatch_log_clear();
  return WS_OK;
}

int ws_debug_row_gather(const double* table, int row_len, const int32_t* idx, int64_t n, double* out, ws_stream stream) {
  if (!table || !idx || !out || row_len <= 0 || (row_len & 1) || n < 0) {
    set_error("ws_debug_row_gather: invalid argument (row_len must be even)");
    return WS_ERR_INVALID_ARG;
  }
  WS_HIP_CHECK(launch_row_gather_probe(table, row_len, idx, n, out, (hipStream_t)stream));
  return WS_OK;
}

int ws_debug_clock_probe(uint64_t* out, int samples, int64_t period_ticks, ws_stream stream) {
  // bounded: at most 4096 samples, 2^24 ticks between two of them, and 2^30 ticks (10.7 s at the counter's 100 MHz)
  // in total -- the probe is one spinning wavefront that nothing can cancel and that hipDeviceSynchronize and
  // ws_engine_reserve wait for (the two bounds alone allowed eleven minutes)
  if (!out || samples <= 0 || samples > 4096 || period_ticks <= 0 || period_ticks > (1LL << 24) ||
      (int64_t)samples * period_ticks > (1LL << 30)) {
    set_error("ws_debug_clock_probe: invalid argument (1..4096 samples, period 1..2^24 ticks, samples * period <= 2^30)");
    return WS_ERR_INVALID_ARG;
  }
  WS_HIP_CHECK(launch_clock_probe(reinterpret_cast<unsigned long long*>(out), samples,
                                  (unsigned long long)period_ticks, (hipStream_t)stream));
  return WS_OK;
}

int ws_debug_fbank_mode(int mode) {
  if (mode < 0 || mode > 2) {
    set_error("ws_debug_fbank_mode: mode %d (0 shipped kernels, 1 packed-fp32 reproducer build, 2 any-length kernel for "
              "every frontend)", mode);
    return WS_ERR_INVALID_ARG;
  }
  set_fbank_debug_mode(mode);
  return WS_OK;
}

long long ws_debug_dispatch_report(char* buf, long long cap) {
  if (cap < 0 || (cap > 0 && !buf)) { set_error("ws_debug_dispatch_report: invalid argument"); return WS_ERR_INVALID_ARG; }
  return (long long)dispatch_log_dump(buf, (size_t)cap);
}

int ws_engine_profile_enable(ws_engine* eng, int on) {
  if (!eng) { set_error("ws_engine_profile_enable: invalid argument"); return WS_ERR_INVALID_ARG; }
  eng->model->prof.enabled = on != 0;
  eng->model->prof.mask = on > 0 ? (unsigned)on : 0xFu;      // bit c = record kernel class c
  return WS_OK;
}

int ws_engine_profile_read(ws_engine* eng, double* ms, double* flops, double* bytes, int* launches) {
  if (!eng || !ms || !flops || !bytes || !launches) {
    set_error("ws_engine_profile_read: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  eng->model->prof.read(ms, flops, bytes, launches);
  return KernelProfiler::kClasses;
}

double ws_engine_flops(const ws_engine* eng, int batch, int num_frames) {
  return eng ? eng->model->flops(batch, num_frames) : 0.0;
}

// ---------------------------------------------------------------------------------------- PLDA
int ws_plda_create(int dim, const double* mu, const double* transform, const double* psi,
                   const double* offset, int normalize_length, int device_id, ws_plda** out) {
  if (!out || dim <= 0 || dim > 4096 || !mu || !transform || !psi || !offset) {
    set_error("ws_plda_create: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  WS_HIP_CHECK(hipSetDevice(device_id));
  ws_plda* p = new ws_plda;
  p->dim = dim; p->device = device_id; p->normalize_length = normalize_length ? 1 : 0;
  auto up = [&](DevBuf& d, const double* src, size_t n) -> hipError_t {
    hipError_t e = d.alloc(n * sizeof(double));
    if (e != hipSuccess) return e;
    return hipMemcpy(d.ptr, src, n * sizeof(double), hipMemcpyHostToDevice);
  };
  hipError_t e;
  if ((e = up(p->mu, mu, dim)) != hipSuccess ||
      (e = up(p->transform, transform, (size_t)dim * dim)) != hipSuccess ||
      (e = up(p->psi, psi, dim)) != hipSuccess || (e = up(p->offset, offset, dim)) != hipSuccess ||
      (e = p->mean_vec.alloc(dim * sizeof(double))) != hipSuccess) {
    delete p;
    set_error("ws_plda_create: device upload failed: %s", hipGetErrorString(e));
    return WS_ERR_HIP;
  }
  *out = p;
  return WS_OK;
}

void ws_plda_destroy(ws_plda* plda) { delete plda; }

static int plda_prepare(ws_plda* p, const void* emb, int is_f64, const int32_t* groups, int n_out,
                        const double* mean_vec, int pre_norm, double* out, hipStream_t st) {
  const double* mv = nullptr;
  WS_HIP_CHECK(hipSetDevice(p->device));
  if (mean_vec) {
    WS_HIP_CHECK(hipMemcpyAsync(p->mean_vec.ptr, mean_vec, p->dim * sizeof(double),
                                hipMemcpyHostToDevice, st));
    mv = p->mean_vec.as<double>();
  }
  if (n_out < 128) {            // few vectors: one workgroup per vector
    WS_HIP_CHECK(launch_plda_prepare(emb, is_f64, groups, n_out, p->dim, mv,
                                     p->transform.as<double>(), p->offset.as<double>(), pre_norm,
                                     p->normalize_length, out, st));
    return WS_OK;
  }
  // many vectors: rows -> V, Y = V transform^T + offset on the f64 MFMA GEMM, row re-normalisation
  if ((size_t)n_out > p->cap_v) {
    WS_HIP_CHECK(hipStreamSynchronize(st));
    WS_HIP_CHECK(p->V.alloc((size_t)n_out * p->dim * sizeof(double)));
    p->cap_v = n_out;
  }
  WS_HIP_CHECK(launch_plda_rows(emb, is_f64, groups, n_out, p->dim, mv, pre_norm, p->V.as<double>(), st));
  WS_HIP_CHECK(launch_plda_llr_gemm(p->V.as<double>(), nullptr, p->offset.as<double>(), n_out,
                                    p->transform.as<double>(), p->dim, p->dim, out, st));
  if (p->normalize_length) WS_HIP_CHECK(launch_plda_rownorm(out, n_out, p->dim, st));
  return WS_OK;
}

int ws_plda_prepare_enroll(ws_plda* plda, const float* emb, const int32_t* group_offsets,
                           int n_groups, const double* mean_vec, double* out, ws_stream stream) {
  if (!plda || !emb || !group_offsets || !out || n_groups < 0) {
    set_error("ws_plda_prepare_enroll: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  return plda_prepare(plda, emb, 0, group_offsets, n_groups, mean_vec, plda->normalize_length, out,
                      (hipStream_t)stream);
}

int ws_plda_prepare_test(ws_plda* plda, const float* emb, int n, const double* mean_vec,
                         double* out, ws_stream stream) {
  if (!plda || !emb || !out || n < 0) {
    set_error("ws_plda_prepare_test: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  return plda_prepare(plda, emb, 0, nullptr, n, mean_vec, plda->normalize_length, out,
                      (hipStream_t)stream);
}

int ws_plda_transform(ws_plda* plda, const double* x, int n, double* out, ws_stream stream) {
  if (!plda || !x || !out || n < 0) {
    set_error("ws_plda_transform: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  return plda_prepare(plda, x, 1, nullptr, n, nullptr, 0, out, (hipStream_t)stream);
}

// GEMM / gather operands.  Uniform n (n_sessions == NULL): K = dim, B operand = the test matrix
// itself, the t^2 term is the per-test constant colc.  Per-model n: K = 2 dim, B = [t | t^2].
struct PldaOps { const double* A; const double* Bm; const double* colc; int K; };

static int plda_terms(ws_plda* p, const double* enroll, const int32_t* n_sessions, int n_uniform,
                      int n_enroll, const double* test, int n_test, hipStream_t st, PldaOps* ops) {
  const size_t D2 = 2 * (size_t)p->dim;
  WS_HIP_CHECK(hipSetDevice(p->device));
  if ((size_t)n_enroll > p->cap_e) {
    WS_HIP_CHECK(hipStreamSynchronize(st));
    WS_HIP_CHECK(p->EA.alloc((size_t)n_enroll * D2 * sizeof(double)));
    WS_HIP_CHECK(p->rowc.alloc((size_t)n_enroll * sizeof(double)));
    p->cap_e = n_enroll;
  }
  if ((size_t)n_test > p->cap_t) {
    WS_HIP_CHECK(hipStreamSynchronize(st));
    WS_HIP_CHECK(p->TT.alloc((size_t)n_test * D2 * sizeof(double)));
    WS_HIP_CHECK(p->colc.alloc((size_t)n_test * sizeof(double)));
    p->cap_t = n_test;
  }
  WS_HIP_CHECK(launch_plda_enroll_terms(enroll, n_sessions, n_uniform, n_enroll, p->dim,
                                        p->psi.as<double>(), p->EA.as<double>(),
                                        p->rowc.as<double>(), st));
  ops->A = p->EA.as<double>();
  if (n_sessions) {
    WS_HIP_CHECK(launch_plda_test_terms(test, n_test, p->dim, p->TT.as<double>(), st));
    ops->Bm = p->TT.as<double>(); ops->colc = nullptr; ops->K = 2 * p->dim;
  } else {
    WS_HIP_CHECK(launch_plda_test_colc(test, n_test, p->dim, n_uniform, p->psi.as<double>(),
                                       p->colc.as<double>(), st));
    ops->Bm = test; ops->colc = p->colc.as<double>(); ops->K = p->dim;
  }
  return WS_OK;
}

int ws_plda_llr_matrix(ws_plda* plda, const double* enroll, const int32_t* n_sessions,
                       int n_uniform, int n_enroll, const double* test, int n_test, double* out,
                       ws_stream stream) {
  if (!plda || !enroll || (!n_sessions && n_uniform <= 0) || !test || !out || n_enroll < 0 ||
      n_test < 0) {
    set_error("ws_plda_llr_matrix: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (n_enroll == 0 || n_test == 0) return WS_OK;
  hipStream_t st = (hipStream_t)stream;
  PldaOps ops;
  int r = plda_terms(plda, enroll, n_sessions, n_uniform, n_enroll, test, n_test, st, &ops);
  if (r) return r;
  WS_HIP_CHECK(launch_plda_llr_gemm(ops.A, plda->rowc.as<double>(), ops.colc, n_enroll, ops.Bm,
                                    n_test, ops.K, out, st));
  return WS_OK;
}

int ws_plda_llr_pairs(ws_plda* plda, const double* enroll, const int32_t* n_sessions,
                      int n_uniform, int n_enroll, const double* test, int n_test,
                      const int32_t* idx_e, const int32_t* idx_t, int64_t num_trials, double* out,
                      ws_stream stream) {
  if (plda && num_trials == 0) return WS_OK;        // empty trial list: nothing to do
  if (!plda || !enroll || (!n_sessions && n_uniform <= 0) || !test || !out || !idx_e || !idx_t ||
      n_enroll < 0 || n_test < 0 || num_trials < 0) {
    set_error("ws_plda_llr_pairs: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (n_enroll == 0 || n_test == 0) {
    set_error("ws_plda_llr_pairs: trials given but an embedding table is empty");
    return WS_ERR_INVALID_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  PldaOps ops;
  int r = plda_terms(plda, enroll, n_sessions, n_uniform, n_enroll, test, n_test, st, &ops);
  if (r) return r;
  WS_HIP_CHECK(launch_plda_llr_pairs(ops.A, plda->rowc.as<double>(), ops.colc, ops.Bm, ops.K, idx_e,
                                     idx_t, num_trials, out, st));
  return WS_OK;
}

// ======================================================================= PLDA training statistics
int64_t ws_plda_stats_scratch(int n, int dim) {
  if (n <= 0 || dim <= 0) return 0;
  return plda_stats_scratch_doubles(n, dim);
}

int ws_plda_stats(const void* emb, int emb_is_f64, int n, int dim, const int32_t* group_offsets, int n_groups,
                  const double* mean_vec, int normalize_length, double* class_mean, double* scatter,
                  double* scratch, int64_t scratch_doubles, ws_stream stream) {
  if (!emb || !group_offsets || !class_mean || !scatter || !scratch || n <= 0 || dim <= 0 ||
      n_groups <= 0) {
    set_error("ws_plda_stats: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  if (scratch_doubles < plda_stats_scratch_doubles(n, dim)) {
    set_error("ws_plda_stats: scratch of %lld doubles is too small (need %lld)",
              (long long)scratch_doubles, (long long)plda_stats_scratch_doubles(n, dim));
    return WS_ERR_CAPACITY;
  }
  WS_HIP_CHECK(launch_plda_stats(emb, emb_is_f64, n, dim, group_offsets, n_groups, mean_vec, normalize_length,
                                 class_mean, scatter, scratch, (hipStream_t)stream));
  return WS_OK;
}

int ws_rows_affine(const void* x, int x_is_f64, int n, int d_in, const double* sub, const double* M,
                   int d_out, int normalize, double* out, ws_stream stream) {
  if (n == 0) return WS_OK;
  if (!x || !out || n < 0 || d_in <= 0 || d_out <= 0 || (!M && d_in != d_out)) {
    set_error("ws_rows_affine: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  WS_HIP_CHECK(launch_rows_affine(x, x_is_f64, n, d_in, sub, M, d_out, normalize, out,
                                  (hipStream_t)stream));
  return WS_OK;
}

// ============================================================================= cosine scoring
namespace {
// 256 B of zeros per device for the GEMM's masked loads (lives for the life of the process)
const float* scoring_zero_page(int device) {
  static float* pages[64] = {nullptr};
  if (device < 0 || device >= 64) return nullptr;
  if (!pages[device]) {
    float* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 256) != hipSuccess) { (void)hipFree(p); return nullptr; }
    pages[device] = p;
  }
  return pages[device];
}
int pointer_device(const void* ptr, int* device) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
    (void)hipGetLastError();
    set_error("pointer %p is not HIP device memory", ptr);
    return WS_ERR_INVALID_ARG;
  }
  *device = attr.device;
  return WS_OK;
}
// out[i][j] = <ua[i], ub[j]>, N = ws_cos_table_rows(n_b) columns
int cos_gemm(const float* ua, int n_a, const float* ub, int n_b, int ld, float* out, int ldo,
             hipStream_t st) {
  int dev = 0;
  int r = pointer_device(ua, &dev);
  if (r) return r;
  WS_HIP_CHECK(hipSetDevice(dev));
  const float* zeros = scoring_zero_page(dev);
  if (!zeros) { set_error("cosine scoring: zero page allocation failed"); return WS_ERR_HIP; }
  ConvGemmParams p;
  std::memset(&p, 0, sizeof(p));
  p.A = ua; p.lda = ld;
  p.W = ub; p.ldw = ld;
  p.prec = 0;                                  // exact fp32 products (the reference is float32 numpy)
  p.D = out; p.ldd = ldo;
  p.Hin = p.Win = p.Hout = p.Wout = 1;
  p.M = n_a; p.N = (n_b + 3) & ~3; p.K = ld; p.Cin = ld;
  p.stride_h = p.stride_w = p.kh = p.kw = p.dil_h = p.dil_w = 1;
  p.splitk = 1;
  p.zeros = zeros;
  WS_HIP_CHECK(launch_conv_gemm(p, st));
  return WS_OK;
}
}  // namespace

int ws_cos_table_rows(int n) { return n < 0 ? 0 : (n + 3) & ~3; }
int ws_cos_table_ld(int dim) { return dim < 0 ? 0 : (dim + 31) & ~31; }

int ws_cos_prepare(const float* emb, const float* mean_vec, int n, int dim, float* unit, float* mag,
                   ws_stream stream) {
  if (n == 0) return WS_OK;
  if (!emb || !unit || n < 0 || dim <= 0) {
    set_error("ws_cos_prepare: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  WS_HIP_CHECK(launch_cos_prepare(emb, mean_vec, n, dim, unit, mag, (hipStream_t)stream));
  return WS_OK;
}

int ws_cos_pairs(const float* unit_a, const float* unit_b, int dim, const int32_t* idx_a,
                 const int32_t* idx_b, int64_t num_trials, float* out, ws_stream stream) {
  if (num_trials == 0) return WS_OK;
  if (!unit_a || !unit_b || !idx_a || !idx_b || !out || dim <= 0 || num_trials < 0) {
    set_error("ws_cos_pairs: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  WS_HIP_CHECK(launch_cos_pairs(unit_a, unit_b, ws_cos_table_ld(dim), idx_a, idx_b, num_trials, out,
                                (hipStream_t)stream));
  return WS_OK;
}

int ws_cos_matrix(const float* unit_a, int n_a, const float* unit_b, int n_b, int dim, float* out,
                  int ldo, ws_stream stream) {
  if (n_a == 0 || n_b == 0) return WS_OK;
  if (!unit_a || !unit_b || !out || n_a < 0 || n_b < 0 || dim <= 0 ||
      ldo < ws_cos_table_rows(n_b) || (ldo & 3)) {
    set_error("ws_cos_matrix: invalid argument (ldo must be a multiple of 4, >= ws_cos_table_rows(n_b))");
    return WS_ERR_INVALID_ARG;
  }
  return cos_gemm(unit_a, n_a, unit_b, n_b, ws_cos_table_ld(dim), out, ldo, (hipStream_t)stream);
}

int ws_cohort_stats(const float* unit, int n, const float* unit_cohort, int n_cohort, int dim,
                    int top_n, float* scratch, int64_t scratch_floats, float* mean, float* sd,
                    ws_stream stream) {
  if (n == 0) return WS_OK;
  if (!unit || !unit_cohort || !scratch || !mean || !sd || n < 0 || n_cohort <= 0 || dim <= 0 ||
      top_n <= 0) {
    set_error("ws_cohort_stats: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  const int ld = ws_cos_table_ld(dim), lds = ws_cos_table_rows(n_cohort);
  int64_t rows_fit = scratch_floats / lds;
  if (rows_fit >= n) rows_fit = n;
  else rows_fit &= ~(int64_t)127;
  if (rows_fit < 128 && rows_fit < n) {
    set_error("ws_cohort_stats: scratch of %lld floats is too small (need >= %lld)",
              (long long)scratch_floats, (long long)128 * lds);
    return WS_ERR_CAPACITY;
  }
  hipStream_t st = (hipStream_t)stream;
  for (int64_t r0 = 0; r0 < n; r0 += rows_fit) {
    const int rows = (int)((n - r0) < rows_fit ? (n - r0) : rows_fit);
    int r = cos_gemm(unit + r0 * ld, rows, unit_cohort, n_cohort, ld, scratch, lds, st);
    if (r) return r;
    WS_HIP_CHECK(launch_topn_stats(scratch, lds, rows, n_cohort, top_n, mean + r0, sd + r0, st));
  }
  return WS_OK;
}

int ws_asnorm_pairs(const float* score, const int32_t* idx_e, const int32_t* idx_t,
                    const float* e_mean, const float* e_sd, const float* t_mean, const float* t_sd,
                    int64_t num_trials, float* out, ws_stream stream) {
  if (num_trials == 0) return WS_OK;
  if (!score || !idx_e || !idx_t || !e_mean || !e_sd || !t_mean || !t_sd || !out || num_trials < 0) {
    set_error("ws_asnorm_pairs: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  WS_HIP_CHECK(launch_asnorm_pairs(score, idx_e, idx_t, e_mean, e_sd, t_mean, t_sd, num_trials, out,
                                   (hipStream_t)stream));
  return WS_OK;
}

}  // extern "C"
