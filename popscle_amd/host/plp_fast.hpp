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


#include "util.hpp"

namespace pa {

struct PlpRead {
  int32_t cell, snp;
  uint32_t numi;  // global kept-base counter
  uint8_t byte;   // allele<<7 | capped bq, or MUXGL_READ_OTHER
};

struct PlpParseOptions {
  int32_t minBQ = 13, capBQ = 20, S = 0;
  const std::vector<int32_t>* index_bcs = nullptr;  // DROPLET_ID -> cell id, -1 = filtered
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
  int64_t lines = 0;         // rows consumed (incl. the one that failed / the blank one)
  bool blank = false;        // a line without fields was met: the file ends there
  bool sorted = true;        // (cell, snp) non-decreasing inside the slice
  std::string err;           // first error; %lld in it is replaced by the global line number
  void reset() {
    b = e = nullptr;
    rds.clear();
    lines = 0;
    blank = false;
    sorted = true;
    err.clear();
  }
};

inline void parse_slice(Slice& sl, const PlpParseOptions& po, const char* prefix) {
  const std::vector<int32_t>& index_bcs = *po.index_bcs;
  const char* p = sl.b;
  uint32_t numi = 0;
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

// inflates `path` into blocks that end at a newline (the last block of the file may lack one)
class GzBlockReader {
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
  bool next(std::vector<char>& blk) {
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

}  // namespace detail

// Parses <prefix>.plp.gz; appends the kept bases to `rds` in file order, returns their number; *sorted tells whether the
// rows came in (cell, SNP) order.
inline uint64_t parse_plp_gz(const std::string& prefix, const PlpParseOptions& po, std::vector<PlpRead>& rds,
                             bool* sorted) {
  using namespace detail;
  {  // size hint: the gzip trailer holds the uncompressed length (mod 2^32); a kept base costs >= ~9 bytes of text
    FILE* f = fopen((prefix + ".plp.gz").c_str(), "rb");
    unsigned char t[4];
    if (f && fseek(f, -4, SEEK_END) == 0 && fread(t, 1, 4, f) == 4) {
      const uint64_t isize = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
      rds.reserve(rds.size() + (size_t)(isize / 9));
    }
    if (f) fclose(f);
  }
  GzBlockReader rd(prefix + ".plp.gz");
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
      for (size_t i = 0; i < (size_t)want; ++i) off[i + 1] = off[i] + sl[i].rds.size();
      if (off.back() > rds.capacity()) rds.reserve(std::max(off.back(), rds.capacity() * 2));
      rds.resize(off.back());
      const uint64_t base0 = numi - off[0];
      parallel_for(want, nth, [&](int64_t i) {
        const std::vector<PlpRead>& src = sl[(size_t)i].rds;
        PlpRead* d = rds.data() + off[(size_t)i];
        const uint32_t base = (uint32_t)(base0 + off[(size_t)i]);
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
        numi += s.rds.size();
      }
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
  if (getenv("POPSCLE_AMD_TIMING")) fprintf(stderr, "TIMING   plp: wait %.3f parse %.3f append %.3f sorted %d threads %d\n", t_wait, t_parse, t_app, (int)file_sorted, nth);
  if (!header) fatal("Cannot read the first line of %s.plp.gz", prefix.c_str());
  *sorted = file_sorted;
  return numi;
}

// Brings the reads (file order) into (cell, SNP, "%x"-string of numi) order.  dsc-pileup writes the table SNP-major
// (cmd_cram_dsc_pileup.cpp:497-518), so the general case is a transpose: a stable bucket pass by cell keeps the file's
// SNP order inside every cell; a cell whose rows were not SNP-ascending in the file is sorted on its own.
inline void plp_order_by_cell(std::vector<PlpRead>& rds, int64_t C, bool already_sorted, std::vector<int64_t>& cell_rd0) {
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
    std::vector<PlpRead> dst((size_t)n);
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
inline void plp_pack(const std::vector<PlpRead>& rds, int64_t C, const std::vector<int64_t>& cell_rd0,
                     std::vector<int64_t>& cell_ptr, std::vector<int32_t>& entry_snp, std::vector<int64_t>& entry_rptr,
                     std::vector<uint8_t>& reads) {
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
