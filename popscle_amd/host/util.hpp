// util.hpp -- small host-side helpers of the popscle-compatible front end: gz line reader with the reference's
// whitespace tokeniser semantics, gz/plain writer, `--flag value` parser, error reporting.
#pragma once

#include <zlib.h>

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

// hprintf()-style writer: plain file for mode "w", gzip for "wz" (the reference writes BGZF, which is a gzip stream)
class OutFile {
 public:
  OutFile(const std::string& path, bool gz) : gz_(gz) {
    if (gz) {
      g_ = gzopen(path.c_str(), "wb");
      if (!g_) fatal("Cannot open %s for writing", path.c_str());
    } else {
      f_ = fopen(path.c_str(), "w");
      if (!f_) fatal("Cannot open %s for writing", path.c_str());
    }
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
  void close() {
    if (g_) gzclose(g_);
    if (f_) fclose(f_);
    g_ = nullptr;
    f_ = nullptr;
  }

 private:
  void write(const char* p, size_t n) {
    if (gz_) {
      if (gzwrite(g_, p, (unsigned)n) != (int)n) fatal("write failed");
    } else if (fwrite(p, 1, n, f_) != n) {
      fatal("write failed");
    }
  }
  bool gz_;
  gzFile g_ = nullptr;
  FILE* f_ = nullptr;
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
