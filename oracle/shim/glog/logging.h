// Seven-line stand-in for glog so the reference's header-only Fbank
// (runtime/core/frontend/fbank.h) compiles without the glog dependency.
// Test infrastructure only (oracle/_ref build); not part of the product.
#pragma once
#include <cstdio>
#include <cstdlib>
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s\n", #c); std::abort(); } } while (0)
#define CHECK_GE(a, b) CHECK((a) >= (b))
