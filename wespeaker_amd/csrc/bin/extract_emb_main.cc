// extract_emb_main -- a C++ caller of the C-ABI (include/wespeaker_amd.h) with no Python and no torch:
// the native twin of the reference's runtime/core/bin/extract_emb_main.cc:43-117.
//
//   extract_emb_main --wav_scp wav.scp | --wav_path a.wav  [--result emb.txt]
//       --speaker_model_path model.wsamd   (wespeaker_amd.engine.save_native_model; the reference
//                                           takes an .onnx here)
//       [--samples_per_chunk 32000] [--precision fp32|f16x3|f16] [--device 0]
//
// Like the reference binary it prints one line per utterance, `key e0 e1 ...`, computed by the
// chunk-and-average rule of SpeakerEngine::ExtractEmbedding (ws_extract_chunked), then the real-time
// factor.  The reference spreads utterances over a host thread pool with one engine per task
// (extract_emb_main.cc:43-47); here one engine owns the GPU and utterances are streamed through it.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/wespeaker_amd.h"

namespace {

// RIFF/WAVE PCM16 reader (what wenet::WavReader does, runtime/core/frontend/wav.h:71-117): walks the
// chunks, takes channel 0.
bool read_wav(const std::string& path, std::vector<int16_t>* pcm, int* sample_rate) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  char id[4];
  uint32_t size = 0;
  if (!f.read(id, 4) || std::memcmp(id, "RIFF", 4) != 0) return false;
  f.read(reinterpret_cast<char*>(&size), 4);
  if (!f.read(id, 4) || std::memcmp(id, "WAVE", 4) != 0) return false;
  uint16_t fmt = 0, channels = 0, bits = 0;
  uint32_t rate = 0;
  while (f.read(id, 4) && f.read(reinterpret_cast<char*>(&size), 4)) {
    if (std::memcmp(id, "fmt ", 4) == 0) {
      if (size < 16 || size > 4096) return false;          // a PCM fmt chunk is 16 bytes (+ extension)
      std::vector<char> b(size);
      if (!f.read(b.data(), size)) return false;
      std::memcpy(&fmt, b.data(), 2); std::memcpy(&channels, b.data() + 2, 2);
      std::memcpy(&rate, b.data() + 4, 4); std::memcpy(&bits, b.data() + 14, 2);
    } else if (std::memcmp(id, "data", 4) == 0) {
      if (fmt != 1 || bits != 16 || channels == 0) return false;
      // the header may lie (0xFFFFFFFF from a streamed recording): never more than the file still holds
      const std::streamoff here = f.tellg();
      f.seekg(0, std::ios::end);
      const std::streamoff left = f.tellg() - here;
      f.seekg(here);
      if (left < 0) return false;
      if ((std::streamoff)size > left) size = (uint32_t)left;
      std::vector<int16_t> raw(size / 2);
      f.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)(raw.size() * 2));
      const size_t got = (size_t)f.gcount() / 2 / channels;
      pcm->resize(got);
      for (size_t i = 0; i < got; ++i) (*pcm)[i] = raw[i * channels];
      *sample_rate = (int)rate;
      return true;
    } else {
      f.seekg(size + (size & 1), std::ios::cur);
    }
  }
  return false;
}

int die(const char* what) {
  std::fprintf(stderr, "extract_emb_main: %s: %s\n", what, ws_last_error());
  return 1;
}

}  // namespace

int main(int argc, char** argv) {
  std::string wav_scp, wav_path, result, model_path, precision = "fp32";
  int samples_per_chunk = 32000, device = 0, fbank_dim = 80, sample_rate = 16000;
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string k = argv[i], v = argv[i + 1];
    if (k == "--wav_scp") wav_scp = v;
    else if (k == "--wav_path") wav_path = v;
    else if (k == "--result") result = v;
    else if (k == "--speaker_model_path") model_path = v;
    else if (k == "--samples_per_chunk") samples_per_chunk = std::atoi(v.c_str());
    else if (k == "--precision") precision = v;
    else if (k == "--device") device = std::atoi(v.c_str());
    else if (k == "--fbank_dim") fbank_dim = std::atoi(v.c_str());
    else if (k == "--sample_rate") sample_rate = std::atoi(v.c_str());
    else { std::fprintf(stderr, "unknown flag %s\n", k.c_str()); return 2; }
  }
  if ((wav_scp.empty() && wav_path.empty()) || model_path.empty()) {
    std::fprintf(stderr, "usage: extract_emb_main (--wav_scp F | --wav_path F) --speaker_model_path M "
                         "[--result F] [--samples_per_chunk N] [--precision fp32|f16x3|f16]\n");
    return 2;
  }
  std::vector<std::pair<std::string, std::string>> waves;
  if (!wav_path.empty()) {
    waves.emplace_back("test", wav_path);
  } else {
    std::ifstream scp(wav_scp);
    std::string line;
    while (std::getline(scp, line)) {
      std::istringstream ss(line);
      std::string key, path;
      if (ss >> key >> path) waves.emplace_back(key, path);
    }
    if (waves.empty()) { std::fprintf(stderr, "Please provide non-empty wav scp.\n"); return 2; }
  }

  if (hipSetDevice(device) != hipSuccess) { std::fprintf(stderr, "no HIP device %d\n", device); return 1; }
  const int chunk_frames = samples_per_chunk > 0 ? ws_num_frames(samples_per_chunk, sample_rate) : 3000;
  ws_engine* eng = nullptr;
  ws_frontend* fe = nullptr;
  if (ws_engine_load(model_path.c_str(), device, 64, chunk_frames > 0 ? chunk_frames : 198, &eng)) return die("load");
  if (ws_frontend_create(sample_rate, fbank_dim, device, &fe)) return die("frontend");
  const int mode = precision == "f16" ? WS_PREC_F16 : precision == "f16x3" ? WS_PREC_F16X3 : WS_PREC_FP32;
  if (ws_engine_set_precision(eng, mode)) return die("precision");
  const int E = ws_engine_embed_dim(eng);

  hipStream_t stream;
  if (hipStreamCreate(&stream) != hipSuccess) return 1;
  std::ofstream out_file;
  if (!result.empty()) out_file.open(result);
  std::ostream& out = result.empty() ? std::cout : out_file;

  int16_t* d_wav = nullptr;
  float* d_emb = nullptr;
  size_t cap = 0;
  if (hipMalloc(reinterpret_cast<void**>(&d_emb), sizeof(float) * E) != hipSuccess) return 1;
  std::vector<float> emb(E);
  long long total_ms_audio = 0;
  double total_ms_extract = 0;
  for (const auto& w : waves) {
    std::vector<int16_t> pcm;
    int sr = 0;
    if (!read_wav(w.second, &pcm, &sr) || sr != sample_rate) {
      std::fprintf(stderr, "cannot read %s as %d Hz PCM16\n", w.second.c_str(), sample_rate);
      return 1;
    }
    if (pcm.size() > cap) {
      if (d_wav) (void)hipFree(d_wav);
      cap = pcm.size() + pcm.size() / 2;
      if (hipMalloc(reinterpret_cast<void**>(&d_wav), cap * 2) != hipSuccess) return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (hipMemcpyAsync(d_wav, pcm.data(), pcm.size() * 2, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    if (samples_per_chunk <= 0) {                       // full mode: the whole utterance is one chunk
      const int frames = ws_num_frames((int)pcm.size(), sample_rate);
      if (frames > ws_engine_max_frames(eng) && ws_engine_reserve(eng, 1, frames + frames / 4)) return die("reserve");
    }
    const int n_chunks = ws_extract_chunked(eng, fe, d_wav, WS_WAV_INT16, (int)pcm.size(), samples_per_chunk,
                                            1.0f, WS_WINDOW_HAMMING, d_emb, stream);
    if (n_chunks < 0) return die(w.first.c_str());
    if (hipMemcpyAsync(emb.data(), d_emb, sizeof(float) * E, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    total_ms_extract += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    total_ms_audio += (long long)pcm.size() * 1000 / sample_rate;
    out << w.first;
    char buf[32];
    for (int i = 0; i < E; ++i) {
      std::snprintf(buf, sizeof(buf), " %.9g", emb[i]);
      out << buf;
    }
    out << std::endl;
  }
  std::fprintf(stderr, "Total: process %lld ms audio taken %.1f ms.  RTF: %.5f  (%zu utterances, %.1f embeddings/s)\n",
               total_ms_audio, total_ms_extract, total_ms_audio ? total_ms_extract / total_ms_audio : 0.0,
               waves.size(), total_ms_extract > 0 ? 1e3 * waves.size() / total_ms_extract : 0.0);
  (void)hipFree(d_wav);
  (void)hipFree(d_emb);
  ws_frontend_destroy(fe);
  ws_engine_destroy(eng);
  return 0;
}
