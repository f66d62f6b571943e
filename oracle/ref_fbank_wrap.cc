// ORACLE build glue (test infrastructure, never shipped).
// Exposes the reference's OWN native fbank (runtime/core/frontend/fbank.h:31-218,
// fft.cc:59-119), compiled in place from /root/reference, through a tiny C entry
// point so that oracle/fbank.py can be pinned against reference-owned arithmetic.
// No reference source is copied: this file only #includes it from where it lies.
#include <vector>
#include <cstring>
#include <limits>
#include <cmath>
#ifndef M_2PI
#define M_2PI 6.283185307179586476925286766559005
#endif
#include "frontend/fbank.h"

extern "C" int ref_fbank(const float* wav, int num_samples, int num_bins, int sample_rate,
                         float* out, int max_frames) {
  int frame_length = sample_rate / 1000 * 25, frame_shift = sample_rate / 1000 * 10;
  wenet::Fbank fb(num_bins, sample_rate, frame_length, frame_shift);
  std::vector<float> w(wav, wav + num_samples);
  std::vector<std::vector<float>> feat;
  int n = fb.Compute(w, &feat);
  if (n > max_frames) n = max_frames;
  for (int i = 0; i < n; ++i) std::memcpy(out + (size_t)i * num_bins, feat[i].data(), sizeof(float) * num_bins);
  return n;
}
