// plp_fast.hpp -- threaded reader of the .plp.gz table (the loader, not the kernels, bounds the
// end-to-end time; the reference parses ~0.5 M rows/s through tsv_reader + std::map<std::string,...>).
//
// Format (cmd_cram_dsc_pileup.cpp:438-523, read back at sc_drop_seq.cpp:339-370): a header line, then one row per
// (droplet, SNP) with four whitespace-separated fields DROPLET_ID SNP_ID ALLELES BASEQS; ALLELES holds one digit per
// base, BASEQS the phred+33 character.  Semantics kept from the reference's reader: fields split on whitespace runs,
// integers by atoi, a line without fields ends the file, rows of a filtered droplet are skipped, every kept base
// (bq >= minBQ) takes the next value of a global counter (its UMI name is that counter printed with "%x").
//
// Pipeline: a reader thread inflates the gzip stream into 8 MiB blocks cut at the last newline; the caller's thread
// cuts every block into line-aligned slices, parses them with OpenMP, and appends the slices in file order.
#pragma once


#include <sys/mman.h>
#include <sys/stat.h>
#include <memory>

#include "util.hpp"

namespace pa {

struct PlpRead {
  int32_t cell, snp;
  uint32_t numi;  // global kept-base counter
  uint8_t byte;   // allele<<7 | capped bq, or MUXGL_READ_OTHER
};

// Growable array of PlpRead without value-initialisation: a std::vector zero-fills on resize, and that single-threaded
// first touch of a few hundred MB (page faults) was a third of the load time; here the parallel copies touch the pages.
class PlpReadVec {
 public:
  PlpReadVec() {}
  explicit PlpReadVec(size_t n) { resize(n); }
  PlpReadVec(const PlpReadVec&) = delete;
  PlpReadVec& operator=(const PlpReadVec&) = delete;
  ~PlpReadVec() { free(p_); }
  size_t size() const { return n_; }
  size_t capacity() const { return cap_; }
  bool empty() const { return n_ == 0; }
  PlpRead* data() { return p_; }
  const PlpRead* data() const { return p_; }
  PlpRead* begin() { return p_; }
  PlpRead* end() { return p_ + n_; }
  const PlpRead* begin() const { return p_; }
  const PlpRead* end() const { return p_ + n_; }
  PlpRead& operator[](size_t i) { return p_[i]; }
  const PlpRead& operator[](size_t i) const { return p_[i]; }
  // a size HINT: like reserve(), but gives up quietly (the list then grows on demand) where the allocation is refused --
  // strict overcommit, a cgroup limit -- since a hint from the file size overshoots what a filtered load keeps
  bool try_reserve(size_t c) {
    if (c <= cap_) return true;
    void* probe = malloc(c * sizeof(PlpRead));
    if (!probe) return false;
    free(probe);
    reserve(c);
    return true;
  }
  void reserve(size_t c) {
    if (c <= cap_) return;
    PlpRead* q = (PlpRead*)malloc(c * sizeof(PlpRead));
    if (!q) fatal("out of memory (%zu reads)", c);
    if (c * sizeof(PlpRead) >= (64u << 20)) {  // fewer page faults where transparent huge pages are on request
      const uintptr_t a = ((uintptr_t)q + 4095) & ~(uintptr_t)4095, z = ((uintptr_t)q + c * sizeof(PlpRead)) & ~(uintptr_t)4095;
      if (z > a) (void)madvise((void*)a, z - a, MADV_HUGEPAGE);
    }
    if (n_) {  // rare (the size hint normally covers the file): copy with all threads
      const size_t nb = (n_ + (1u << 16) - 1) >> 16;
      PlpRead* src = p_;
      const size_t n = n_;
      parallel_for((int64_t)nb, plp_threads(), [&](int64_t b) {
        const size_t o = (size_t)b << 16;
        memcpy((void*)(q + o), (const void*)(src + o), std::min<size_t>(1u << 16, n - o) * sizeof(PlpRead));
      });
    }
    free(p_);
    p_ = q;
    cap_ = c;
  }
  void resize(size_t n) {  // new elements are NOT initialised
    if (n > cap_) reserve(std::max(n, cap_ * 2));
    n_ = n;
  }
  void swap(PlpReadVec& o) {
    std::swap(p_, o.p_);
    std::swap(n_, o.n_);
    std::swap(cap_, o.cap_);
  }

 private:
  PlpRead* p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

struct PlpParseOptions {
  int32_t minBQ = 13, capBQ = 20, S = 0;
  const std::vector<int32_t>* index_bcs = nullptr;  // DROPLET_ID -> cell id, -1 = filtered
  // slab filter (one rank of a sharded run): only the rows of cells [c0, c1) or markers [s0, s1) are kept; the bases of
  // the other rows are still COUNTED, because the name of a kept base -- and with it the order of the reads inside an
  // entry -- is its number among all kept bases of the file (sc_drop_seq.cpp:364)
  int32_t c0 = 0, c1 = INT32_MAX, s0 = 0, s1 = INT32_MAX;
};

// strcmp(sprintf("%x", a), sprintf("%x", b)) < 0 without the strings: left-align the hex digits; if one string is a
// prefix of the other the shorter one is smaller
inline bool hex_string_less(uint32_t a, uint32_t b) {
  auto ndig = [](uint32_t v) { return v ? (35 - __builtin_clz(v)) / 4 : 1; };
  const int la = ndig(a), lb = ndig(b);
  const uint64_t xa = (uint64_t)a << (4 * (8 - la)), xb = (uint64_t)b << (4 * (8 - lb));
  if (xa != xb) return xa < xb;
  return la < lb;
}

namespace detail {

inline bool is_ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

inline int32_t atoi_tok(const char* b, const char* e) {  // atoi on a token: optional sign, digits until a non-digit
  bool neg = false;
  if (b < e && (*b == '-' || *b == '+')) neg = *b++ == '-';
  int64_t v = 0;
  while (b < e && *b >= '0' && *b <= '9') v = v * 10 + (*b++ - '0');
  return (int32_t)(neg ? -v : v);
}

struct Slice {
  const char *b = nullptr, *e = nullptr;
  std::vector<PlpRead> rds;  // numi local to the slice
  uint32_t nkept = 0;        // kept bases of the slice, those of rows outside the slab filter included
  int64_t lines = 0;         // rows consumed (incl. the one that failed / the blank one)
  bool blank = false;        // a line without fields was met: the file ends there
  bool sorted = true;        // (cell, snp) non-decreasing inside the slice
  std::string err;           // first error; %lld in it is replaced by the global line number
  void reset() {
    b = e = nullptr;
    rds.clear();
    nkept = 0;
    lines = 0;
    blank = false;
    sorted = true;
    err.clear();
  }
};

inline void parse_slice(Slice& sl, const PlpParseOptions& po, const char* prefix) {
  const std::vector<int32_t>& index_bcs = *po.index_bcs;
  const char* p = sl.b;
  uint32_t& numi = sl.nkept;
  int64_t lastkey = -1;
  char msg[512];
  while (p < sl.e) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(sl.e - p));
    if (!nl) nl = sl.e;
    ++sl.lines;
    const char *fb[4], *fe[4];
    int nf = 0;
    const char* q = p;
    while (q < nl) {
      while (q < nl && is_ws((unsigned char)*q)) ++q;
      if (q >= nl) break;
      const char* t0 = q;
      while (q < nl && !is_ws((unsigned char)*q)) ++q;
      if (nf < 4) {
        fb[nf] = t0;
        fe[nf] = q;
      }
      ++nf;
    }
    p = nl + 1;
    if (nf == 0) {
      sl.blank = true;
      return;
    }
    if (nf < 4) {
      snprintf(msg, sizeof(msg), "%s.plp.gz: line %%lld has %d fields", prefix, nf);
      sl.err = msg;
      return;
    }
    const int32_t did = atoi_tok(fb[0], fe[0]);
    if (did < 0 || did >= (int32_t)index_bcs.size()) {
      snprintf(msg, sizeof(msg), "%s.plp.gz: DROPLET_ID %d out of range", prefix, did);
      sl.err = msg;
      return;
    }
    const int32_t ibc = index_bcs[(size_t)did];
    if (ibc < 0) continue;
    const int32_t snp = atoi_tok(fb[1], fe[1]);
    if (snp < 0 || snp >= po.S) {
      snprintf(msg, sizeof(msg), "%s.plp.gz: SNP_ID %d out of range", prefix, snp);
      sl.err = msg;
      return;
    }
    const char *pa = fb[2], *pq = fb[3];
    const int32_t l = (int32_t)(fe[3] - fb[3]);
    if ((int32_t)(fe[2] - fb[2]) != l) {
      snprintf(msg, sizeof(msg), "Length are different between %.*s and %.*s", (int)(fe[2] - fb[2]), pa, (int)l, pq);
      sl.err = msg;
      return;
    }
    if (!((ibc >= po.c0 && ibc < po.c1) || (snp >= po.s0 && snp < po.s1))) {  // outside the slabs: count, do not keep
      for (int32_t i = 0; i < l; ++i) numi += (int)(char)(pq[i] - (char)33) >= po.minBQ;
      continue;
    }
    const int64_t key = ((int64_t)ibc << 32) | (uint32_t)snp;
    if (key < lastkey) sl.sorted = false;
    lastkey = key;
    for (int32_t i = 0; i < l; ++i) {
      const int bq0 = (int)(char)(pq[i] - (char)33);
      if (bq0 >= po.minBQ) {
        const int bq = bq0 > po.capBQ ? po.capBQ : bq0;
        const int al = (int)(char)(pa[i] - (char)'0');
        const uint8_t byte = (al == 0) ? (uint8_t)bq : (al == 1) ? (uint8_t)(0x80 | bq) : (uint8_t)MUXGL_READ_OTHER;
        sl.rds.push_back(PlpRead{ibc, snp, numi++, byte});
      }
    }
  }
}

// source of inflated text blocks that end at a newline (the last block of the file may lack one)
struct BlockSource {
  virtual ~BlockSource() {}
  virtual bool next(std::vector<char>& blk) = 0;  // false at end of file
};

// plain gzip stream: one thread inflates (a gzip member cannot be split), the parser runs beside it
class GzBlockReader : public BlockSource {
 public:
  static constexpr size_t BLK = 8u << 20;
  explicit GzBlockReader(const std::string& path) {
    fp_ = gzopen(path.c_str(), "rb");
    if (!fp_) fatal("Cannot open %s for reading", path.c_str());
    gzbuffer(fp_, 1 << 20);
    th_ = std::thread([this] { run(); });
  }
  ~GzBlockReader() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
    }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
    if (fp_) gzclose(fp_);
  }
  // next block, false at end of file; the buffer stays valid until the following call
  bool next(std::vector<char>& blk) override {
    std::unique_lock<std::mutex> g(m_);
    cv_.wait(g, [this] { return have_ || done_; });
    if (!have_) {
      if (!err_.empty()) fatal("%s", err_.c_str());
      return false;
    }
    blk.swap(ready_);
    have_ = false;
    g.unlock();
    cv_.notify_all();
    return true;
  }

 private:
  void run() {
    std::vector<char> carry;
    for (;;) {
      std::vector<char> buf(carry.size() + BLK);
      if (!carry.empty()) memcpy(buf.data(), carry.data(), carry.size());
      const int n = gzread(fp_, buf.data() + carry.size(), (unsigned)BLK);
      if (n < 0) {
        std::lock_guard<std::mutex> g(m_);
        err_ = "gzread failed";
        done_ = true;
        cv_.notify_all();
        return;
      }
      const size_t tot = carry.size() + (size_t)n;
      const bool eof = n == 0;
      size_t cut = tot;
      if (!eof) {
        while (cut > 0 && buf[cut - 1] != '\n') --cut;
        if (cut == 0) {  // no newline in the whole block: keep accumulating
          carry.assign(buf.begin(), buf.begin() + (long)tot);
          continue;
        }
      }
      carry.assign(buf.begin() + (long)cut, buf.begin() + (long)tot);
      buf.resize(cut);
      if (!buf.empty()) {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !have_ || stop_; });
        if (stop_) return;
        ready_.swap(buf);
        have_ = true;
        g.unlock();
        cv_.notify_all();
      }
      if (eof) {
        std::lock_guard<std::mutex> g(m_);
        done_ = true;
        cv_.notify_all();
        return;
      }
    }
  }
  gzFile fp_ = nullptr;
  std::thread th_;
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<char> ready_;
  bool have_ = false, done_ = false, stop_ = false;
  std::string err_;
};

// BGZF (what dsc-pileup writes through hts_open(..., "wz"), cmd_cram_dsc_pileup.cpp:438-440): a chain of independent
// gzip members of <= 64 KiB, each announcing its compressed size in the BC extra field and its inflated size in the
// trailer -- so a batch of members is inflated by the whole worker pool straight to its place in the text block.
class BgzfBlockReader : public BlockSource {
 public:
  static constexpr size_t TEXT = 64u << 20;  // inflated bytes per block (about): 1024 members, so that the pool's hand-overs are rare
  static bool is_bgzf(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned char h[18];
    const bool ok = fread(h, 1, 18, f) == 18 && member_size(h, 18) > 0;
    fclose(f);
    return ok;
  }
  explicit BgzfBlockReader(const std::string& path) : path_(path) {
    f_ = fopen(path.c_str(), "rb");
    if (!f_) fatal("Cannot open %s for reading", path.c_str());
    // the members are inflated straight from a read-only mapping of the file (no copy of the compressed bytes through a
    // buffer on the one thread that walks the member headers); a file that cannot be mapped is read in 8 MB pieces
    struct stat st;
    if (fstat(fileno(f_), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
      void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f_), 0);
      if (m != MAP_FAILED) {
        map_ = (const unsigned char*)m;
        map_n_ = (size_t)st.st_size;
        (void)madvise(m, map_n_, MADV_SEQUENTIAL);
      }
    }
  }
  ~BgzfBlockReader() override {
    if (map_) munmap((void*)map_, map_n_);
    if (f_) fclose(f_);
  }
  bool next(std::vector<char>& blk) override {
    if (finished_) return false;
    struct Member {
      size_t off, csize, out;
      uint32_t isize;
    };
    std::vector<Member> mem;
    size_t text = carry_.size();
    if (!map_) {
      fpos_ += cpos_;  // drop what the previous call consumed; from here on the buffer only grows (members keep offsets)
      cbuf_.erase(cbuf_.begin(), cbuf_.begin() + (long)cpos_);
      cpos_ = 0;
    }
    auto have = [&]() { return (map_ ? map_n_ : cbuf_.size()) - cpos_; };
    auto refill = [&]() { return map_ ? false : this->refill(); };
    // gather whole members until the block is large enough
    for (;;) {
      if (have() < 18 && !refill()) break;
      if (have() == 0) break;
      const unsigned char* p = (map_ ? map_ : (const unsigned char*)cbuf_.data()) + cpos_;
      const size_t avail = have();
      const long ms = avail >= 18 ? member_size(p, avail) : -1;
      if (ms <= 0) fatal("%s: not a BGZF member at compressed offset %zu", path_.c_str(), fpos_ + cpos_);
      if ((size_t)ms > avail) {
        if (!refill()) fatal("%s: truncated BGZF member", path_.c_str());
        continue;
      }
      const uint32_t isize = (uint32_t)p[ms - 4] | ((uint32_t)p[ms - 3] << 8) | ((uint32_t)p[ms - 2] << 16) |
                             ((uint32_t)p[ms - 1] << 24);
      if (isize > 0x10000) fatal("%s: BGZF member claims %u bytes", path_.c_str(), isize);
      mem.push_back(Member{cpos_, (size_t)ms, text, isize});
      text += isize;
      cpos_ += (size_t)ms;
      if (text >= TEXT) break;
    }
    blk.resize(text);
    if (!carry_.empty()) memcpy(blk.data(), carry_.data(), carry_.size());
    std::atomic<bool> bad(false);
    const unsigned char* base = map_ ? map_ : (const unsigned char*)cbuf_.data();
    parallel_for_blocked((int64_t)mem.size(), 8, plp_threads(), [&](int64_t i) {
      const Member& m = mem[(size_t)i];
      if (m.isize == 0) return;  // e.g. the end-of-file marker
      const unsigned char* p = base + m.off;
      const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
      z_stream zs;
      memset(&zs, 0, sizeof(zs));
      if (inflateInit2(&zs, -15) != Z_OK) {
        bad = true;
        return;
      }
      zs.next_in = (Bytef*)(p + 12 + xlen);
      zs.avail_in = (uInt)(m.csize - 12 - xlen - 8);
      zs.next_out = (Bytef*)(blk.data() + m.out);
      zs.avail_out = m.isize;
      const int rc = inflate(&zs, Z_FINISH);
      const bool ok = rc == Z_STREAM_END && zs.total_out == m.isize;
      inflateEnd(&zs);
      const uint32_t want = (uint32_t)p[m.csize - 8] | ((uint32_t)p[m.csize - 7] << 8) |
                            ((uint32_t)p[m.csize - 6] << 16) | ((uint32_t)p[m.csize - 5] << 24);
      if (!ok || (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)(blk.data() + m.out), m.isize) != want) bad = true;
    });
    if (bad) fatal("%s: corrupt BGZF member", path_.c_str());
    const bool eof = map_ ? cpos_ == map_n_ : (cbuf_.size() == cpos_ && feof_);
    size_t cut = blk.size();
    if (!eof) {
      while (cut > 0 && blk[cut - 1] != '\n') --cut;
    }
    carry_.assign(blk.begin() + (long)cut, blk.end());
    blk.resize(cut);
    if (eof) finished_ = true;
    if (blk.empty()) return eof ? false : next(blk);
    return true;
  }

 private:
  // size of the gzip member starting at p if it is a BGZF block (gzip header with FEXTRA and a BC subfield), else -1
  static long member_size(const unsigned char* p, size_t avail) {
    if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return -1;
    const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
    if (avail < 12 + xlen) return xlen <= 0xffff ? 0x10000 : -1;  // need more bytes to see the subfields
    for (size_t o = 12; o + 4 <= 12 + xlen;) {
      const size_t slen = (size_t)p[o + 2] | ((size_t)p[o + 3] << 8);
      if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen)
        return (long)((size_t)p[o + 4] | ((size_t)p[o + 5] << 8)) + 1;
      o += 4 + slen;
    }
    return -1;
  }
  bool refill() {  // append the next chunk of the file
    if (feof_) return false;
    const size_t old = cbuf_.size(), want = 8u << 20;
    cbuf_.resize(old + want);
    const size_t n = fread(cbuf_.data() + old, 1, want, f_);
    cbuf_.resize(old + n);
    if (n < want) feof_ = true;
    return n > 0;
  }
  std::string path_;
  FILE* f_ = nullptr;
  std::vector<char> cbuf_, carry_;
  const unsigned char* map_ = nullptr;  // the whole file, when it could be mapped (cpos_ then counts from its start)
  size_t map_n_ = 0;
  size_t cpos_ = 0, fpos_ = 0;
  bool feof_ = false, finished_ = false;
};

}  // namespace detail

// Parses <prefix>.plp.gz; appends the kept bases to `rds` in file order, returns their number; *sorted tells whether the
// rows came in (cell, SNP) order.
inline uint64_t parse_plp_gz(const std::string& prefix, const PlpParseOptions& po, PlpReadVec& rds,
                             bool* sorted) {
  using namespace detail;
  {  // size hint: the gzip trailer holds the uncompressed length (mod 2^32; of the LAST member: nothing for a BGZF file,
     // which ends in an empty one), a kept base costs >= ~9 bytes of text and -- measured on dsc-pileup-like tables --
     // 4 to 6 compressed bytes: reserve for the larger guess (address space only: untouched pages cost nothing)
    FILE* f = fopen((prefix + ".plp.gz").c_str(), "rb");
    unsigned char t[4];
    if (f && fseek(f, -4, SEEK_END) == 0) {
      const uint64_t fsize = (uint64_t)ftell(f) + 4;
      uint64_t isize = 0;
      if (fread(t, 1, 4, f) == 4) isize = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
      (void)rds.try_reserve(rds.size() + (size_t)std::max<uint64_t>(isize / 9, fsize / 4 + (fsize >> 5)));
    }
    if (f) fclose(f);
  }
  std::unique_ptr<BlockSource> src;
  if (BgzfBlockReader::is_bgzf(prefix + ".plp.gz"))
    src.reset(new BgzfBlockReader(prefix + ".plp.gz"));
  else
    src.reset(new GzBlockReader(prefix + ".plp.gz"));
  BlockSource& rd = *src;
  const int nth = plp_threads();
  std::vector<char> blk;
  std::vector<Slice> sl;  // reused from block to block: their buffers stay warm
  bool header = false, file_sorted = true, ended = false;
  uint64_t numi = 0;
  int64_t nlines = 0, lastkey = -1;
  double t_wait = 0, t_parse = 0, t_app = 0;
  auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; };
  double tA = now();
  while (!ended && rd.next(blk)) {
    double tB = now(); t_wait += tB - tA;
    const char *b = blk.data(), *e = b + blk.size();
    if (!header) {  // sc_drop_seq.cpp:339-344
      const char* nl = (const char*)memchr(b, '\n', (size_t)(e - b));
      std::string h(b, nl ? nl : e);
      std::vector<std::string> f;
      for (size_t i = 0; i < h.size();) {
        while (i < h.size() && is_ws((unsigned char)h[i])) ++i;
        size_t j = i;
        while (j < h.size() && !is_ws((unsigned char)h[j])) ++j;
        if (j > i) f.push_back(h.substr(i, j - i));
        i = j;
      }
      if (f.empty()) fatal("Cannot read the first line of %s.plp.gz", prefix.c_str());
      if (f.size() != 4 || f[0] != "#DROPLET_ID" || f[1] != "SNP_ID" || f[2] != "ALLELES" || f[3] != "BASEQS")
        fatal("THe header line of %s.plp.gz is malformed or outdated. Expecting #DROPLET_ID SNP_ID ALLELES BASEQS",
              prefix.c_str());
      header = true;
      nlines = 1;
      b = nl ? nl + 1 : e;
    }
    // line-aligned slices
    const int want = (int)std::max<int64_t>(1, std::min<int64_t>(nth * 4, (e - b) / (64 << 10) + 1));
    if (sl.size() < (size_t)want) sl.resize((size_t)want);
    for (Slice& x : sl) x.reset();
    {
      const char* cur = b;
      for (int i = 0; i < want; ++i) {
        sl[(size_t)i].b = cur;
        const char* tgt = (i + 1 == want) ? e : b + (e - b) * (int64_t)(i + 1) / want;
        if (tgt < cur) tgt = cur;
        if (tgt < e) {
          const char* nl = (const char*)memchr(tgt, '\n', (size_t)(e - tgt));
          tgt = nl ? nl + 1 : e;
        }
        sl[(size_t)i].e = tgt;
        cur = tgt;
      }
    }
    parallel_for(want, nth, [&](int64_t i) {
      sl[(size_t)i].rds.reserve((size_t)(sl[(size_t)i].e - sl[(size_t)i].b) / 4 + 16);
      parse_slice(sl[(size_t)i], po, prefix.c_str());
    });
    double tC = now(); t_parse += tC - tB;
    // append in file order
    std::vector<size_t> off((size_t)want + 1, rds.size());
    {  // copy the slices' reads to their places (numi made global), in parallel
      std::vector<uint64_t> nbase((size_t)want + 1, numi);  // kept bases of the file in front of each slice
      for (size_t i = 0; i < (size_t)want; ++i) {
        off[i + 1] = off[i] + sl[i].rds.size();
        nbase[i + 1] = nbase[i] + sl[i].nkept;
      }
      rds.resize(off.back());
      parallel_for(want, nth, [&](int64_t i) {
        const std::vector<PlpRead>& src = sl[(size_t)i].rds;
        PlpRead* d = rds.data() + off[(size_t)i];
        const uint32_t base = (uint32_t)nbase[(size_t)i];
        for (size_t k = 0; k < src.size(); ++k) {
          d[k] = src[k];
          d[k].numi += base;
        }
      });
    }
    for (size_t si = 0; si < (size_t)want; ++si) {
      Slice& s = sl[si];
      if (!s.rds.empty()) {
        const int64_t first = ((int64_t)s.rds.front().cell << 32) | (uint32_t)s.rds.front().snp;
        const int64_t last = ((int64_t)s.rds.back().cell << 32) | (uint32_t)s.rds.back().snp;
        if (!s.sorted || first < lastkey) file_sorted = false;
        lastkey = last;
      }
      numi += s.nkept;
      if (!s.err.empty()) {
        std::string m = s.err;
        const size_t k = m.find("%lld");
        if (k != std::string::npos) m.replace(k, 4, std::to_string(nlines + s.lines));
        fatal("%s", m.c_str());
      }
      nlines += s.lines;
      if (s.blank) {  // the file ends here: drop what the later slices parsed
        rds.resize(off[si + 1]);
        ended = true;
        break;
      }
    }
    tA = now(); t_app += tA - tC;
  }
  if (getenv("POPSCLE_AMD_TIMING")) fprintf(stderr, "TIMING   plp: wait %.3f parse %.3f append %.3f sorted %d threads %d bgzf %d\n", t_wait, t_parse, t_app, (int)file_sorted, nth, (int)(dynamic_cast<BgzfBlockReader*>(src.get()) != nullptr));
  if (!header) fatal("Cannot read the first line of %s.plp.gz", prefix.c_str());
  *sorted = file_sorted;
  return numi;
}

// Brings the reads (file order) into (cell, SNP, "%x"-string of numi) order.  dsc-pileup writes the table SNP-major
// (cmd_cram_dsc_pileup.cpp:497-518), so the general case is a transpose: a stable bucket pass by cell keeps the file's
// SNP order inside every cell; a cell whose rows were not SNP-ascending in the file is sorted on its own.
inline void plp_order_by_cell(PlpReadVec& rds, int64_t C, bool already_sorted, std::vector<int64_t>& cell_rd0) {
  const int64_t n = (int64_t)rds.size();
  const int nth = plp_threads();
  const int64_t P = std::max<int64_t>(1, std::min<int64_t>(nth, n / (1 << 16) + 1));  // input parts
  cell_rd0.assign((size_t)C + 1, 0);
  if (!already_sorted) {
    std::vector<int64_t> cnt((size_t)(P * C), 0);  // [part][cell]
    parallel_for(P, nth, [&](int64_t t) {
      int64_t* c = cnt.data() + t * C;
      for (int64_t i = n * t / P; i < n * (t + 1) / P; ++i) ++c[rds[(size_t)i].cell];
    });
    int64_t run = 0;
    for (int64_t c = 0; c < C; ++c) {
      cell_rd0[(size_t)c] = run;
      for (int64_t t = 0; t < P; ++t) {
        const int64_t k = cnt[(size_t)(t * C + c)];
        cnt[(size_t)(t * C + c)] = run;
        run += k;
      }
    }
    cell_rd0[(size_t)C] = run;
    PlpReadVec dst((size_t)n);
    parallel_for(P, nth, [&](int64_t t) {
      int64_t* c = cnt.data() + t * C;
      for (int64_t i = n * t / P; i < n * (t + 1) / P; ++i) dst[(size_t)c[rds[(size_t)i].cell]++] = rds[(size_t)i];
    });
    rds.swap(dst);
  } else {
    for (int64_t i = 0; i < n; ++i) ++cell_rd0[(size_t)rds[(size_t)i].cell + 1];
    for (int64_t c = 0; c < C; ++c) cell_rd0[(size_t)c + 1] += cell_rd0[(size_t)c];
  }
  parallel_for_blocked(C, 64, nth, [&](int64_t c) {
    PlpRead* b = rds.data() + cell_rd0[(size_t)c];
    PlpRead* e = rds.data() + cell_rd0[(size_t)c + 1];
    bool ok = true;
    for (PlpRead* q = b; q + 1 < e; ++q)
      if (q[1].snp < q[0].snp) {
        ok = false;
        break;
      }
    if (!ok)
      std::stable_sort(b, e, [](const PlpRead& x, const PlpRead& y) { return x.snp < y.snp; });  // numi stays ascending
    for (PlpRead* q = b; q < e;) {  // entries are short: re-order each one by the "%x" string of its counters
      PlpRead* r = q + 1;
      while (r < e && r->snp == q->snp) ++r;
      if (r - q > 1) std::sort(q, r, [](const PlpRead& x, const PlpRead& y) { return hex_string_less(x.numi, y.numi); });
      q = r;
    }
  });
}

// ordered reads -> CSR (cell_ptr, entry_snp, entry_rptr, reads)
inline void plp_pack(const PlpReadVec& rds, int64_t C, const std::vector<int64_t>& cell_rd0,
                     std::vector<int64_t>& cell_ptr, BigVec<int32_t>& entry_snp, BigVec<int64_t>& entry_rptr,
                     BigVec<uint8_t>& reads) {
  const int nth = plp_threads();
  const int64_t n = (int64_t)rds.size();
  cell_ptr.assign((size_t)C + 1, 0);
  parallel_for_blocked(C, 64, nth, [&](int64_t c) {
    int64_t k = 0;
    for (int64_t i = cell_rd0[(size_t)c]; i < cell_rd0[(size_t)c + 1]; ++i)
      if (i == cell_rd0[(size_t)c] || rds[(size_t)i].snp != rds[(size_t)i - 1].snp) ++k;
    cell_ptr[(size_t)c + 1] = k;
  });
  for (int64_t c = 0; c < C; ++c) cell_ptr[(size_t)c + 1] += cell_ptr[(size_t)c];
  const int64_t nnz = cell_ptr[(size_t)C];
  entry_snp.resize((size_t)nnz);
  entry_rptr.resize((size_t)nnz + 1);
  reads.resize((size_t)n);
  entry_rptr[(size_t)nnz] = n;
  parallel_for_blocked(C, 64, nth, [&](int64_t c) {
    int64_t e = cell_ptr[(size_t)c] - 1;
    for (int64_t i = cell_rd0[(size_t)c]; i < cell_rd0[(size_t)c + 1]; ++i) {
      if (i == cell_rd0[(size_t)c] || rds[(size_t)i].snp != rds[(size_t)i - 1].snp) {
        ++e;
        entry_snp[(size_t)e] = rds[(size_t)i].snp;
        entry_rptr[(size_t)e] = i;
      }
      reads[(size_t)i] = rds[(size_t)i].byte;
    }
  });
}

}  // namespace pa
