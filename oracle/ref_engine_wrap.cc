// ORACLE build glue (test infrastructure, never shipped).
// Drives the reference's OWN chunk-and-average engine -- SpeakerEngine::ExtractFeature /
// ApplyMean / ExtractEmbedding, runtime/core/speaker/speaker_engine.cc:63-159, on top of its own
// FeaturePipeline (frontend/feature_pipeline.cc) and Fbank -- compiled in place from /root/reference.
// The only thing replaced is the ONNX/MNN model back-end (needs ONNXRuntime / MNN, not buildable
// here): `SpeakerModel` is the reference's own abstract interface (speaker/speaker_model.h:25-32) and
// the stand-in forwards every chunk to a caller-supplied callback, so the cutting rule, the per-chunk
// CMN and the averaging that produce the result are the reference's code.
// No reference source is copied: this file only #includes it from where it lies.
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>
#ifndef M_2PI
#define M_2PI 6.283185307179586476925286766559005
#endif
// model_ is a private member that only the USE_ONNX / USE_MNN constructors fill
#define private public
#include "speaker/speaker_engine.h"
#undef private

typedef void (*ref_model_cb)(const float* feats, int frames, int feat_dim, float* emb, int emb_dim,
                             void* user);

namespace {
struct CallbackModel : wespeaker::SpeakerModel {
  ref_model_cb cb;
  int emb_dim;
  void* user;
  void ExtractEmbedding(const std::vector<std::vector<float>>& feats,
                        std::vector<float>* embed) override {
    const int T = (int)feats.size(), D = T ? (int)feats[0].size() : 0;
    std::vector<float> flat((size_t)T * D);
    for (int t = 0; t < T; ++t) std::memcpy(flat.data() + (size_t)t * D, feats[t].data(), sizeof(float) * D);
    embed->assign(emb_dim, 0.f);
    cb(flat.data(), T, D, embed->data(), emb_dim, user);
  }
};
}  // namespace

// Returns the number of chunks the reference engine cut (>= 0); avg_emb receives emb_dim floats.
extern "C" int ref_engine_extract(const int16_t* pcm, int num_samples, int feat_dim, int sample_rate,
                                  int emb_dim, int samples_per_chunk, ref_model_cb cb, void* user,
                                  float* avg_emb) {
  wespeaker::SpeakerEngine engine("", feat_dim, sample_rate, emb_dim, samples_per_chunk);
  auto model = std::make_shared<CallbackModel>();
  model->cb = cb; model->emb_dim = emb_dim; model->user = user;
  engine.model_ = model;
  std::vector<std::vector<std::vector<float>>> chunks;
  engine.ExtractFeature(pcm, num_samples, &chunks);          // only to report the chunk count
  std::vector<float> out;
  engine.ExtractEmbedding(pcm, num_samples, &out);
  std::memcpy(avg_emb, out.data(), sizeof(float) * emb_dim);
  return (int)chunks.size();
}
