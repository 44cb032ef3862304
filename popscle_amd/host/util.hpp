// util.hpp -- small host-side helpers of the popscle-compatible front end: gz line reader with the reference's
// whitespace tokeniser semantics, gz/plain writer, `--flag value` parser, error reporting.
#pragma once

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace pa {

// the reference's error(): print "FATAL ERROR -", throw (Error.cpp:29-43); main() catches and exits nonzero
[[noreturn]] inline void fatal(const char* fmt, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  fprintf(stderr, "\nFATAL ERROR - \n%s\n\n", buf);
  throw std::runtime_error(buf);
}

inline void notice(const char* fmt, ...) {  // Error.cpp notice(): timestamped stderr line
  time_t t = time(nullptr);
  char ts[64];
  strftime(ts, sizeof(ts), "%Y/%m/%d %H:%M:%S", localtime(&t));
  fprintf(stderr, "NOTICE [%s] - ", ts);
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}

// stage timer: prints "TIMING <what> <seconds>" to stderr when POPSCLE_AMD_TIMING is set
class StageTimer {
 public:
  StageTimer() : on_(getenv("POPSCLE_AMD_TIMING") != nullptr) { clock_gettime(CLOCK_MONOTONIC, &t0_); }
  void lap(const char* what) {
    timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (on_) fprintf(stderr, "TIMING %-28s %.3f s\n", what, (double)(t1.tv_sec - t0_.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0_.tv_nsec));
    t0_ = t1;
  }

 private:
  bool on_;
  timespec t0_;
};

// Line reader over plain or gzip text with the field semantics of the reference's tsv_reader (tsv_reader.cpp:28-51):
// fields are split on runs of whitespace (ksplit with delimiter 0), int fields are atoi, double fields are atof, and
// read_line() returns the number of fields, 0 at end of file.  A blank line therefore ends the file, as it does in the
// reference loops `while( tsv.read_line() > 0 )`.
class TsvReader {
 public:
  explicit TsvReader(const std::string& path) : path_(path) {
    fp_ = gzopen(path.c_str(), "rb");
    if (!fp_) fatal("Cannot open %s for reading", path.c_str());
    gzbuffer(fp_, 1 << 20);
  }
  ~TsvReader() { close(); }
  void close() {
    if (fp_) gzclose(fp_);
    fp_ = nullptr;
  }
  int read_line() {
    line_.clear();
    fields_.clear();
    if (!fp_) return 0;
    char buf[65536];
    bool got = false;
    while (gzgets(fp_, buf, sizeof(buf))) {
      got = true;
      size_t n = strlen(buf);
      line_.append(buf, n);
      if (n && buf[n - 1] == '\n') break;
    }
    if (!got) return 0;
    while (!line_.empty() && (line_.back() == '\n' || line_.back() == '\r')) line_.pop_back();
    ++nlines;
    char* s = line_.empty() ? nullptr : &line_[0];
    size_t i = 0, n = line_.size();
    while (i < n) {
      while (i < n && isspace((unsigned char)s[i])) ++i;
      if (i >= n) break;
      size_t b = i;
      while (i < n && !isspace((unsigned char)s[i])) ++i;
      fields_.push_back(std::make_pair(b, i - b));
    }
    for (auto& f : fields_) s[f.first + f.second] = '\0';  // in-place termination (the byte is whitespace or the end)
    nfields = (int)fields_.size();
    return nfields;
  }
  const char* str_field_at(int i) const { return line_.c_str() + fields_[(size_t)i].first; }
  int int_field_at(int i) const { return atoi(str_field_at(i)); }
  double double_field_at(int i) const { return atof(str_field_at(i)); }
  int nfields = 0;
  int nlines = 0;

 private:
  std::string path_;
  gzFile fp_ = nullptr;
  std::string line_;
  std::vector<std::pair<size_t, size_t>> fields_;
};

// worker threads of the host side (loader, writers); POPSCLE_AMD_THREADS overrides
inline int plp_threads() {
  if (const char* ev = getenv("POPSCLE_AMD_THREADS")) return std::max(1, atoi(ev));
  const unsigned hc = std::thread::hardware_concurrency();
  return (int)std::min(16u, std::max(1u, hc));
}

// threads of a pure compute pass over independent cells (the exact-call pass): no I/O thread to leave room for
inline int compute_threads() {
  if (const char* ev = getenv("POPSCLE_AMD_THREADS")) return std::max(1, atoi(ev));
  const unsigned hc = std::thread::hardware_concurrency();
  return (int)std::min(64u, std::max(1u, hc));
}

// fn(i) for i in [0, n), items handed out one at a time to the threads of a persistent pool.  The workers sleep on a
// condition variable between calls (OpenMP's spinning workers starve the inflating thread when every core is taken;
// threads spawned per call are not spread over the cores before a 5 ms job is over).
class WorkerPool {
 public:
  static WorkerPool& get() {
    static WorkerPool p;
    return p;
  }
  void run(int64_t n, int nth, const std::function<void(int64_t)>& fn) {
    if (n <= 0) return;
    nth = (int)std::min<int64_t>(nth, n);
    if (nth <= 1) {
      for (int64_t i = 0; i < n; ++i) fn(i);
      return;
    }
    std::lock_guard<std::mutex> serial(run_m_);
    {
      std::lock_guard<std::mutex> g(m_);
      while ((int)th_.size() < nth - 1) th_.emplace_back([this] { loop(); });
      fn_ = &fn;
      n_ = n;
      next_.store(0);
      busy_ = std::min<int>((int)th_.size(), nth - 1);
      want_ = busy_;
      ++gen_;
    }
    cv_.notify_all();
    for (int64_t i; (i = next_.fetch_add(1)) < n;) fn(i);
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this] { return busy_ == 0; });
    fn_ = nullptr;
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> g(m_);
      quit_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }

 private:
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> g(m_);
      cv_.wait(g, [&] { return quit_ || (gen_ != seen && want_ > 0); });
      if (quit_) return;
      seen = gen_;
      --want_;
      const std::function<void(int64_t)>* fn = fn_;
      const int64_t n = n_;
      g.unlock();
      for (int64_t i; (i = next_.fetch_add(1)) < n;) (*fn)(i);
      g.lock();
      if (--busy_ == 0) done_.notify_all();
    }
  }
  std::mutex m_, run_m_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> th_;
  const std::function<void(int64_t)>* fn_ = nullptr;
  std::atomic<int64_t> next_{0};
  int64_t n_ = 0;
  int busy_ = 0, want_ = 0;
  uint64_t gen_ = 0;
  bool quit_ = false;
};

// std::vector whose resize() leaves new elements of a trivial type uninitialised: a plain vector zero-fills them on one
// thread -- seconds of page faults for the multi-GB arrays of a packed pileup; here the parallel fill touches the pages
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = NoInitAlloc<U>;
  };
  NoInitAlloc() = default;
  template <class U>
  NoInitAlloc(const NoInitAlloc<U>&) {}
  template <class U>
  void construct(U* q) noexcept(std::is_nothrow_default_constructible<U>::value) {
    ::new (static_cast<void*>(q)) U;
  }
  template <class U, class... A>
  void construct(U* q, A&&... a) {
    ::new (static_cast<void*>(q)) U(std::forward<A>(a)...);
  }
};
template <class T>
using BigVec = std::vector<T, NoInitAlloc<T>>;

inline void parallel_for(int64_t n, int nth, const std::function<void(int64_t)>& fn) { WorkerPool::get().run(n, nth, fn); }

// fn(c) for c in [0, n) in blocks of `grain`
inline void parallel_for_blocked(int64_t n, int64_t grain, int nth, const std::function<void(int64_t)>& fn) {
  const int64_t nb = (n + grain - 1) / grain;
  parallel_for(nb, nth, [&](int64_t b) {
    const int64_t e = std::min(n, (b + 1) * grain);
    for (int64_t c = b * grain; c < e; ++c) fn(c);
  });
}

// ---- number formatting of the writers' inner loops (the cluster VCF of a 500 k x 500 k run has 32 M sample fields: with
// snprintf alone that is 20 CPU-seconds, two thirds of it in the three "%.3lg") ----
inline int fmt_int(int32_t v, char* o) {  // "%d"
  char tmp[12];
  int n = 0;
  uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
  do tmp[n++] = (char)('0' + u % 10), u /= 10; while (u);
  int k = 0;
  if (v < 0) o[k++] = '-';
  while (n) o[k++] = tmp[--n];
  return k;
}
// "%.3lg" of a finite x in [1e-290, 1e3), character for character what glibc prints -- or -1 (the caller then uses
// snprintf) where this code cannot be sure.  glibc rounds the EXACT binary value to three significant digits, half to
// even.  Here x is scaled to [100, 1000) in extended precision (relative error < 1e-17) and rounded to an integer; if the
// scaled value lies within 1e-6 of a half, which way the exact value rounds is not certain and the call declines.
inline int fmt_g3(double x, char* o) {
  if (!(x >= 1e-290 && x < 1e3)) return -1;
  static const struct Pow10 {
    long double p[640];  // 10^(i - 320)
    Pow10() {
      p[320] = 1.0L;
      for (int i = 321; i < 640; ++i) p[i] = p[i - 1] * 10.0L;
      for (int i = 319; i >= 0; --i) p[i] = p[i + 1] / 10.0L;
    }
  } P;
  int e2;
  (void)frexp(x, &e2);                       // x = m 2^e2, m in [0.5, 1)
  int E = (int)((e2 - 1) * 0.30102999566398120);  // floor(log10 x) or one more / less
  if (E > 3) E = 3;
  long double sc = (long double)x * P.p[320 + 2 - E];  // x / 10^(E - 2): in [100, 1000) when E = floor(log10 x)
  while (sc >= 1000.0L) ++E, sc = (long double)x * P.p[320 + 2 - E];
  while (sc < 100.0L) --E, sc = (long double)x * P.p[320 + 2 - E];
  const long double fl = floorl(sc), fr = sc - fl;
  if (fr > 0.499999L && fr < 0.500001L) return -1;
  int D = (int)fl + (fr > 0.5L ? 1 : 0);
  if (D == 1000) D = 100, ++E;
  if (E >= 3) return -1;  // ("%e" style with a positive exponent: not needed by the callers)
  const int d0 = D / 100, d1 = D / 10 % 10, d2 = D % 10;
  int k = 0;
  if (E < -4) {  // d.dde-XX, trailing zeros removed
    o[k++] = (char)('0' + d0);
    if (d1 || d2) {
      o[k++] = '.';
      o[k++] = (char)('0' + d1);
      if (d2) o[k++] = (char)('0' + d2);
    }
    o[k++] = 'e';
    o[k++] = '-';
    const int a = -E;
    if (a >= 100) o[k++] = (char)('0' + a / 100);
    o[k++] = (char)('0' + a / 10 % 10);
    o[k++] = (char)('0' + a % 10);
    return k;
  }
  if (E >= 0) {  // 1 <= x < 1000: E + 1 integer digits, the rest behind the point
    const char dg[3] = {(char)('0' + d0), (char)('0' + d1), (char)('0' + d2)};
    for (int i = 0; i <= E; ++i) o[k++] = dg[i];
    int last = 2;
    while (last > E && dg[last] == '0') --last;
    if (last > E) {
      o[k++] = '.';
      for (int i = E + 1; i <= last; ++i) o[k++] = dg[i];
    }
    return k;
  }
  o[k++] = '0';  // 0.000ddd
  o[k++] = '.';
  for (int i = 0; i < -E - 1; ++i) o[k++] = '0';
  o[k++] = (char)('0' + d0);
  if (d1 || d2) {
    o[k++] = (char)('0' + d1);
    if (d2) o[k++] = (char)('0' + d2);
  }
  return k;
}
// appends "%.3lg" of x
inline int fmt_g3_or_printf(double x, char* o) {
  const int n = fmt_g3(x, o);
  return n >= 0 ? n : sprintf(o, "%.3lg", x);
}

// hprintf()-style writer: plain file for mode "w", BGZF for "wz" -- what the reference's hts_open(..., "wz") produces:
// a series of independent gzip members of <= 64 KiB input each, with the BC extra field holding the member size and an
// empty member as end marker (SAM spec 4.1).  Independent members are also what makes the compression parallel: text is
// collected in 4 MiB batches whose blocks are deflated by the worker pool and written in order.
class OutFile {
 public:
  OutFile(const std::string& path, bool gz) : gz_(gz) {
    f_ = fopen(path.c_str(), gz ? "wb" : "w");
    if (!f_) fatal("Cannot open %s for writing", path.c_str());
    if (gz) buf_.reserve(kBatch + kBlock);
  }
  ~OutFile() { close(); }
  void printf(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    char small[4096];
    va_list ap2;
    va_copy(ap2, ap);
    int n = vsnprintf(small, sizeof(small), fmt, ap);
    va_end(ap);
    if (n < (int)sizeof(small)) {
      write(small, (size_t)n);
    } else {
      std::vector<char> big((size_t)n + 1);
      vsnprintf(big.data(), big.size(), fmt, ap2);
      write(big.data(), (size_t)n);
    }
    va_end(ap2);
  }
  void write(const char* p, size_t n) {
    if (!gz_) {
      if (fwrite(p, 1, n, f_) != n) fatal("write failed");
      return;
    }
    buf_.insert(buf_.end(), p, p + n);
    if (buf_.size() >= kBatch) flush_blocks(false);
  }
  void close() {
    if (!f_) return;
    if (gz_) {
      flush_blocks(true);
      static const unsigned char eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
                                            0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (fwrite(eof, 1, sizeof(eof), f_) != sizeof(eof)) fatal("write failed");
    }
    if (fclose(f_) != 0) fatal("write failed");
    f_ = nullptr;
  }

 private:
  // deflate level of the blocks: zlib's 6, which is htslib's default; POPSCLE_AMD_GZ_LEVEL = 1 .. 9 for something else (the
  // content is the same at any level; the writers' time is mostly number formatting, not deflate: level 4 measured ±0)
  static int gz_level() {
    static const int lvl = [] {
      const char* e = getenv("POPSCLE_AMD_GZ_LEVEL");
      const int v = e ? atoi(e) : 6;
      return v >= 1 && v <= 9 ? v : 6;
    }();
    return lvl;
  }
  static constexpr size_t kBlock = 0xff00;   // input bytes per BGZF block (htslib's BGZF_BLOCK_SIZE)
  static constexpr size_t kBatch = 4u << 20;
  // deflates whole blocks of buf_ (all of it when `all`), keeps the tail
  void flush_blocks(bool all) {
    const size_t nb = all ? (buf_.size() + kBlock - 1) / kBlock : buf_.size() / kBlock;
    if (nb == 0) return;
    std::vector<std::vector<unsigned char>> out(nb);
    std::atomic<bool> bad(false);
    parallel_for((int64_t)nb, plp_threads(), [&](int64_t i) {
      const size_t o = (size_t)i * kBlock, len = std::min(kBlock, buf_.size() - o);
      std::vector<unsigned char>& dst = out[(size_t)i];
      dst.resize(18 + compressBound((uLong)len) + 8);
      z_stream zs;
      memset(&zs, 0, sizeof(zs));
      if (deflateInit2(&zs, gz_level(), Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
        bad = true;
        return;
      }
      zs.next_in = (Bytef*)(buf_.data() + o);
      zs.avail_in = (uInt)len;
      zs.next_out = dst.data() + 18;
      zs.avail_out = (uInt)(dst.size() - 18 - 8);
      const int rc = deflate(&zs, Z_FINISH);
      const size_t clen = zs.total_out;
      deflateEnd(&zs);
      if (rc != Z_STREAM_END || 18 + clen + 8 > 0x10000) {
        bad = true;
        return;
      }
      static const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
      memcpy(dst.data(), hdr, 16);
      const size_t bsize = 18 + clen + 8 - 1;
      dst[16] = (unsigned char)(bsize & 0xff);
      dst[17] = (unsigned char)(bsize >> 8);
      const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)(buf_.data() + o), (uInt)len);
      unsigned char* tl = dst.data() + 18 + clen;
      for (int k = 0; k < 4; ++k) tl[k] = (unsigned char)(crc >> (8 * k));
      for (int k = 0; k < 4; ++k) tl[4 + k] = (unsigned char)((uint32_t)len >> (8 * k));
      dst.resize(18 + clen + 8);
    });
    if (bad) fatal("BGZF compression failed");
    for (const auto& d : out)
      if (fwrite(d.data(), 1, d.size(), f_) != d.size()) fatal("write failed");
    const size_t used = std::min(buf_.size(), nb * kBlock);
    buf_.erase(buf_.begin(), buf_.begin() + (long)used);
  }
  bool gz_;
  FILE* f_ = nullptr;
  std::vector<char> buf_;
};

// `--flag value` parser with the reference's conventions (params.cpp:167-171,449-485): long options only, boolean
// flags take no value, a repeated multi-value flag appends.
class Args {
 public:
  void add_string(const char* name, std::string* v) { strs_[name] = v; }
  void add_int(const char* name, int32_t* v) { ints_[name] = v; }
  void add_double(const char* name, double* v) { dbls_[name] = v; }
  void add_bool(const char* name, bool* v) { bools_[name] = v; }
  void add_multi_string(const char* name, std::vector<std::string>* v) { mstrs_[name] = v; }
  void add_multi_double(const char* name, std::vector<double>* v) { mdbls_[name] = v; }
  void parse(int argc, char** argv) {
    for (int i = 0; i < argc; ++i) {
      const char* a = argv[i];
      if (strncmp(a, "--", 2) != 0) fatal("Cannot recognize the argument %s", a);
      std::string key(a + 2);
      if (bools_.count(key)) {
        *bools_[key] = true;
        continue;
      }
      if (i + 1 >= argc) fatal("Missing value for the option --%s", key.c_str());
      const char* val = argv[++i];
      if (strs_.count(key)) *strs_[key] = val;
      else if (ints_.count(key)) *ints_[key] = atoi(val);
      else if (dbls_.count(key)) *dbls_[key] = atof(val);
      else if (mstrs_.count(key)) mstrs_[key]->push_back(val);
      else if (mdbls_.count(key)) mdbls_[key]->push_back(atof(val));
      else fatal("Cannot recognize the option --%s", key.c_str());
    }
  }

 private:
  std::map<std::string, std::string*> strs_;
  std::map<std::string, int32_t*> ints_;
  std::map<std::string, double*> dbls_;
  std::map<std::string, bool*> bools_;
  std::map<std::string, std::vector<std::string>*> mstrs_;
  std::map<std::string, std::vector<double>*> mdbls_;
};

}  // namespace pa
