// Small stand-in for glog so the reference's native frontend / engine sources
// (runtime/core/frontend/fbank.h, feature_pipeline.cc, speaker/speaker_engine.cc) compile without
// the glog dependency.  Test infrastructure only (oracle/_ref build); not part of the product.
#pragma once
#include <cstdio>
#include <cstdlib>
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s\n", #c); std::abort(); } } while (0)
#define CHECK_GE(a, b) CHECK((a) >= (b))
#define CHECK_EQ(a, b) CHECK((a) == (b))
namespace oracle_shim {
struct NullLog {
  template <typename T>
  NullLog& operator<<(const T&) { return *this; }
};
}  // namespace oracle_shim
#define LOG(severity) ::oracle_shim::NullLog()
