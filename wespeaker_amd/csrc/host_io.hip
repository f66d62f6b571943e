// Host-side wave-file loader of the batch driver (no device code): decode threads that read RIFF/WAVE PCM16
// files straight into the rows of a (pinned) batch buffer.
//
// Stands where the reference has DataLoader worker processes running processor.parse_raw / read_audio
// (wespeaker/dataset/processor.py:119-136, bin/extract.py:99-103 `num_workers`, `prefetch_factor`) and the native
// runtime's wenet::WavReader (runtime/core/frontend/wav.h:71-117).  A Python thread pool decodes ~7 k files/s
// on this path (GIL-bound: ~70 us of interpreter time per file), an eighth of what one MI355X embeds; here the
// per-file work is a read() + a chunk walk + one memcpy on a std::thread, and Python is touched once per batch.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace {

struct WavInfo {
  long data_off = 0;       // byte offset of the sample data
  long frames = 0;         // samples per channel
  int channels = 0, rate = 0;
};

// RIFF chunk walk over the header (wav.h:71-117 accepts the same canonical and LIST-extended layouts), on the
// first bytes of the file (one pread; a header that does not fit 4 KB is walked with further preads).
// Returns false for anything that is not 16-bit integer PCM.
bool parse_header(int fd, long file_size, WavInfo* w) {
  unsigned char h[4096];
  const long got = pread(fd, h, sizeof(h), 0);
  if (got < 12 || std::memcmp(h, "RIFF", 4) != 0 || std::memcmp(h + 8, "WAVE", 4) != 0) return false;
  bool have_fmt = false;
  long pos = 12;
  for (;;) {
    unsigned char cbuf[24];
    const unsigned char* c;
    long have = 24;                                 // bytes of this chunk (id, size, first 16 of the body) in hand
    if (pos + 24 <= got) c = h + pos;
    else {
      have = pread(fd, cbuf, 24, pos);
      if (have < 8) return false;
      c = cbuf;
    }
    const unsigned size = c[4] | (c[5] << 8) | (c[6] << 16) | ((unsigned)c[7] << 24);
    if (std::memcmp(c, "fmt ", 4) == 0) {
      if (size < 16 || have < 24) return false;     // (a format chunk cut off by the end of the file)
      const unsigned char* b = c + 8;
      const int fmt = b[0] | (b[1] << 8), bits = b[14] | (b[15] << 8);
      w->channels = b[2] | (b[3] << 8);
      w->rate = b[4] | (b[5] << 8) | (b[6] << 16) | ((int)b[7] << 24);
      if (fmt != 1 || bits != 16 || w->channels < 1) return false;
      have_fmt = true;
    } else if (std::memcmp(c, "data", 4) == 0) {
      if (!have_fmt) return false;
      const long avail = file_size - (pos + 8);
      const long bytes = (long)size < avail ? (long)size : avail;
      w->data_off = pos + 8;
      w->frames = bytes > 0 ? bytes / (2L * w->channels) : 0;
      return true;
    }
    pos += 8 + (long)size + (size & 1);
    if (pos + 8 > file_size) return false;
  }
}

bool open_wav(const char* path, int* fd, WavInfo* w) {
  *fd = path ? open(path, O_RDONLY | O_CLOEXEC) : -1;
  if (*fd < 0) return false;
  struct stat st;
  if (fstat(*fd, &st) != 0 || !parse_header(*fd, (long)st.st_size, w)) {
    close(*fd);
    *fd = -1;
    return false;
  }
  return true;
}

bool pread_all(int fd, void* dst, size_t bytes, long off) {
  char* p = reinterpret_cast<char*>(dst);
  while (bytes > 0) {
    const ssize_t r = pread(fd, p, bytes, off);
    if (r <= 0) return false;
    p += r; off += r; bytes -= (size_t)r;
  }
  return true;
}

// Persistent decode threads (spawning 16 threads per batch cost more than reading the batch).
class Pool {
 public:
  void run(int n, int threads, const std::function<void(int)>& body) {
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    if (threads <= 1) {
      for (int i = 0; i < n; ++i) body(i);
      return;
    }
    std::unique_lock<std::mutex> call(call_mu_);          // one parallel loop at a time
    {
      std::lock_guard<std::mutex> g(mu_);
      while ((int)workers_.size() < threads - 1) workers_.emplace_back([this]() { loop(); });
      body_ = &body; n_ = n; next_.store(0); active_ = threads - 1; wanted_ = threads - 1; ++epoch_;
    }
    cv_.notify_all();
    for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) body(i);
    std::unique_lock<std::mutex> g(mu_);
    done_cv_.wait(g, [this]() { return active_ == 0; });
    body_ = nullptr;
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }

 private:
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(int)>* body;
      int n;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&]() { return stop_ || (epoch_ != seen && wanted_ > 0); });
        if (stop_) return;
        seen = epoch_;
        --wanted_;
        body = body_; n = n_;
      }
      for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) (*body)(i);
      std::lock_guard<std::mutex> g(mu_);
      if (--active_ == 0) done_cv_.notify_one();
    }
  }
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> workers_;
  const std::function<void(int)>* body_ = nullptr;
  std::atomic<int> next_{0};
  int n_ = 0, active_ = 0, wanted_ = 0;
  unsigned long epoch_ = 0;
  bool stop_ = false;
};

Pool& pool() {
  static Pool* p = new Pool;      // intentionally leaked: no joins in static destructors of a dlopen'ed library
  return *p;
}

}  // namespace

extern "C" {

int ws_wav_probe(const char* const* paths, int n, int threads, int32_t* num_samples, int32_t* sample_rate) {
  if (n < 0 || (n > 0 && (!paths || !num_samples || !sample_rate))) {
    wsamd::set_error("ws_wav_probe: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  std::atomic<int> bad(0);
  pool().run(n, threads, [&](int i) {
    WavInfo w;
    int fd;
    const bool ok = open_wav(paths[i], &fd, &w) && w.frames <= 0x7fffffffL;
    if (fd >= 0) close(fd);
    num_samples[i] = ok ? (int32_t)w.frames : -1;
    sample_rate[i] = ok ? w.rate : 0;
    if (!ok) bad.fetch_add(1);
  });
  return bad.load();
}

int ws_wav_load_rows(const char* const* paths, int n, int threads, int16_t* dst, int64_t row_stride,
                     const int32_t* start, const int32_t* count) {
  if (n < 0 || (n > 0 && (!paths || !dst || !count)) || row_stride < 0) {
    wsamd::set_error("ws_wav_load_rows: invalid argument");
    return WS_ERR_INVALID_ARG;
  }
  std::atomic<int> first_bad(n);
  pool().run(n, threads, [&](int i) {
    const long s0 = start ? start[i] : 0, cnt = count[i];
    bool ok = cnt >= 0 && cnt <= row_stride && s0 >= 0;
    WavInfo w;
    int fd = -1;
    ok = ok && open_wav(paths[i], &fd, &w) && s0 + cnt <= w.frames;
    if (ok && cnt > 0) {
      int16_t* row = dst + (size_t)i * row_stride;
      const long off = w.data_off + 2L * w.channels * s0;
      if (w.channels == 1) {
        ok = pread_all(fd, row, 2 * (size_t)cnt, off);       // straight into the (pinned) batch row
      } else {                               // interleaved: keep channel 0 (cli/speaker.py:139 `wavform[0]`)
        std::vector<int16_t> tmp((size_t)cnt * w.channels);
        ok = pread_all(fd, tmp.data(), 2 * tmp.size(), off);
        for (long k = 0; ok && k < cnt; ++k) row[k] = tmp[(size_t)k * w.channels];
      }
    }
    if (fd >= 0) close(fd);
    if (!ok) {
      int cur = first_bad.load();
      while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
    }
  });
  if (first_bad.load() < n) {
    wsamd::set_error("ws_wav_load_rows: cannot read %d samples from '%s' (not 16-bit PCM, or shorter than asked)",
                     (int)count[first_bad.load()], paths[first_bad.load()] ? paths[first_bad.load()] : "(null)");
    return WS_ERR_INVALID_ARG;
  }
  return WS_OK;
}

}  // extern "C"
