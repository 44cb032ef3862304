// popscle-amd -- popscle-compatible front end for the demuxlet / freemuxlet genotype-likelihood path, with the hot path
// running on an MI355X through the C-ABI of libmuxgl (include/muxgl.h).
//
//   popscle-amd demuxlet   --plp P --vcf V [--field GT|GP|PL] --out O [...]     mirrors cmdCramDemuxlet (cmd_cram_demuxlet.cpp)
//   popscle-amd freemuxlet --plp P --nsample K --out O [...]                   mirrors cmdCramFreemux2 (cmd_cram_freemux2.cpp)
//   popscle-amd freemuxlet-old --plp P --nsample K --out O [...]               mirrors cmdCramFreemuxlet (cmd_cram_freemuxlet.cpp)
//   popscle-amd dump-plp   --plp P [--vcf V --field F] --out FILE              loader only: packed pileup to a binary file
//   popscle-amd synth-plp  --cells C --snps S --samples V --out P              a synthetic data set in the real file formats
//
// Everything here is host plumbing: flag surface, loaders (plp.hpp, vcf.hpp), the sequential control flow of
// the reference commands, and the text writers with the reference's printf formats.  All arithmetic of the path is
// behind muxgl_* calls; there is no CPU implementation of it in this program.
#include <cmath>

#include "plp.hpp"
#include "synthplp.hpp"

using namespace pa;

namespace {

void check(muxgl_handle* h, int rc, const char* what) {
  if (rc != 0) fatal("%s failed: %s", what, muxgl_last_error(h));
}

struct CommonFlags {
  std::string plpPrefix, outPrefix, groupList;
  LoadOptions lo;
  int32_t device = 0;
  std::string devices;  // --devices 0,1,2,...: a device group (muxgl_config.n_devices), the in-process counterpart of
                        // running one popscle per --group-list chunk (README.md:168)
  void add(Args& a) {
    a.add_string("plp", &plpPrefix);
    a.add_string("out", &outPrefix);
    a.add_int("cap-BQ", &lo.capBQ);
    a.add_int("min-BQ", &lo.minBQ);
    a.add_string("group-list", &lo.groupList);
    a.add_int("min-total", &lo.minRead);
    a.add_int("min-umi", &lo.minUMI);
    a.add_int("min-snp", &lo.minSNP);
    a.add_int("device", &device);
    a.add_string("devices", &devices);
  }
  // the handle the command computes on: one device, or the group named by --devices
  muxgl_config config(int32_t flags) const {
    muxgl_config cfg = MUXGL_CONFIG_INIT;
    cfg.device_id = device;
    cfg.flags = flags;
    if (!devices.empty()) {
      size_t pos = 0;
      while (pos <= devices.size()) {
        const size_t comma = std::min(devices.find(',', pos), devices.size());
        const std::string tok = devices.substr(pos, comma - pos);
        if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos)
          fatal("--devices expects a comma-separated list of device ordinals, got '%s'", devices.c_str());
        if (cfg.n_devices >= MUXGL_MAX_DEVICES) fatal("--devices names more than %d devices", MUXGL_MAX_DEVICES);
        cfg.device_ids[cfg.n_devices++] = atoi(tok.c_str());
        pos = comma + 1;
      }
      cfg.device_id = cfg.device_ids[0];
    }
    return cfg;
  }
  bool grouped() const { return devices.find(',') != std::string::npos; }
};

void upload(muxgl_handle* h, const Pileup& p) {
  check(h, muxgl_set_pileup(h, p.C(), p.S(), p.nnz(), (int64_t)p.reads.size(), p.cell_ptr.data(), p.entry_snp.data(),
                            p.entry_rptr.data(), p.reads.data()),
        "muxgl_set_pileup");
}

// The reference draws from the C library's rand().  Other code in this process (the HIP runtime) draws from and
// reseeds that shared generator, so the commands keep their own copy of it: glibc's rand() is random() on the default
// 128-byte additive-feedback state, which initstate_r / random_r reproduce value for value (srand(s) == seed s; a
// program that never calls srand() runs on seed 1).
struct RefRand {
  random_data rd;
  char state[128];
  explicit RefRand(unsigned seed) {
    memset(&rd, 0, sizeof(rd));
    memset(state, 0, sizeof(state));
    initstate_r(seed, state, sizeof(state), &rd);
  }
  int next() {
    int32_t r = 0;
    random_r(&rd, &r);
    return (int)r;
  }
};

const char* sid(const Pileup& p, int i) { return (i >= 0 && i < p.nv) ? p.sample_ids[(size_t)i].c_str() : "NA"; }

// ------------------------------------------------------------------------------------------------ demuxlet
int cmd_demuxlet(int argc, char** argv) {
  CommonFlags cf;
  VcfReader vr;
  std::vector<std::string> smIDs;
  std::string smList;
  std::vector<double> gridAlpha;
  double doublet_prior = 0.5;  // cmd_cram_demuxlet.cpp:32
  std::string sam, tagGroup, tagUMI;
  int32_t dummy_i = 0;
  bool deviceCalls = false;  // (ours) skip the host pass that settles mirrored alpha = 0.5 pairs and near-tie calls as the reference does
  Args a;
  cf.add(a);
  a.add_bool("device-calls", &deviceCalls);
  a.add_string("vcf", &vr.path);
  a.add_string("field", &cf.lo.field);
  a.add_double("geno-error-offset", &cf.lo.genoErrorOffset);
  a.add_double("geno-error-coeff", &cf.lo.genoErrorCoeffR2);
  a.add_string("r2-info", &cf.lo.r2info);
  a.add_int("min-mac", &vr.vfilt.minMAC);
  a.add_double("min-callrate", &vr.vfilt.minCallRate);
  a.add_multi_string("sm", &smIDs);
  a.add_string("sm-list", &smList);
  a.add_multi_double("alpha", &gridAlpha);
  a.add_double("doublet-prior", &doublet_prior);
  a.add_string("sam", &sam);  // BAM-direct mode needs htslib: accepted and rejected
  a.add_string("tag-group", &tagGroup);
  a.add_string("tag-UMI", &tagUMI);
  a.add_int("sam-verbose", &dummy_i);
  a.add_int("vcf-verbose", &dummy_i);
  a.add_int("min-MQ", &dummy_i);
  a.add_int("min-TD", &dummy_i);
  a.add_int("excl-flag", &dummy_i);
  a.parse(argc, argv);
  if (!sam.empty()) fatal("--sam (BAM-direct mode) is not available in this build: run dsc-pileup first and pass --plp");
  if (cf.plpPrefix.empty() || vr.path.empty() || cf.outPrefix.empty()) fatal("Missing required option(s) : --plp, --vcf, --out");
  if (gridAlpha.empty()) {  // :85-89
    gridAlpha.push_back(0);
    gridAlpha.push_back(0.5);
  }
  if ((int)gridAlpha.size() > MUXGL_MAX_ALPHA) fatal("at most %d --alpha values are supported", MUXGL_MAX_ALPHA);
  vr.wanted = smIDs;
  if (!smList.empty()) {
    TsvReader t(smList);
    while (t.read_line() > 0) vr.wanted.push_back(t.str_field_at(0));
  }
  vr.init();
  Pileup p;
  StageTimer tm;
  load_from_plp(cf.plpPrefix, cf.lo, &vr, p);
  tm.lap("demuxlet: load");

  muxgl_config cfg = cf.config(MUXGL_FLAG_DEMUX_ONLY);  // cells are independent: a group needs no column slabs
  muxgl_handle* h = nullptr;
  if (muxgl_create(&cfg, &h) != 0) fatal("%s", muxgl_last_error(nullptr));
  tm.lap("demuxlet: device init");
  upload(h, p);
  check(h, muxgl_demux_set_gp(h, p.nv, p.gp.data(), p.has_gp.data()), "muxgl_demux_set_gp");
  tm.lap("demuxlet: hand-over (H2D+plans)");
  muxgl_demux_params dp;
  memset(&dp, 0, sizeof(dp));
  dp.n_alpha = (int32_t)gridAlpha.size();
  for (size_t i = 0; i < gridAlpha.size(); ++i) dp.alpha[i] = gridAlpha[i];
  dp.doublet_prior = doublet_prior;
  notice("Starting to identify best matching individual IDs");
  std::vector<muxgl_demux_cell> cells((size_t)p.C());
  check(h, muxgl_demux_run(h, &dp, cells.data(), nullptr), "muxgl_demux_run");
  tm.lap("demuxlet: muxgl_demux_run");
  if (!deviceCalls) {
    // the calls rounding noise could decide -- the order of a mirrored alpha = 0.5 pair in DBL.BEST.GUESS / NEXT.GUESS
    // (cmd_cram_demuxlet.cpp:738-746,883-906) and near ties of the scans and thresholds (:827-837,925-988) -- settled in
    // the reference's own arithmetic (exact_calls.hpp)
    int64_t st[6];
    check(h, muxgl_demux_exact_calls(p.C(), p.nv, p.cell_ptr.data(), p.entry_snp.data(), p.entry_rptr.data(), p.reads.data(),
                                     p.gp.data(), p.has_gp.data(), &dp, cells.data(), compute_threads(), st),
          "muxgl_demux_exact_calls");
    notice("Exact-call pass: %lld droplets looked at (%lld near ties besides mirrored pairs, %lld needing every hypothesis, "
           "%lld calls changed)", (long long)st[0], (long long)st[3], (long long)st[4], (long long)st[5]);
    tm.lap("demuxlet: exact calls (host)");
  }

  // .best, cmd_cram_demuxlet.cpp:629,636-641,993-1013: rows in barcode-sorted order, INT_ID counts skipped cells too
  OutFile w(cf.outPrefix + ".best", false);
  w.printf("INT_ID\tBARCODE\tNUM.SNPS\tNUM.READS\tDROPLET.TYPE\tBEST.GUESS\tBEST.LLK\tNEXT.GUESS\tNEXT.LLK\t"
           "DIFF.LLK.BEST.NEXT\tBEST.POSTERIOR\tSNG.POSTERIOR\tSNG.BEST.GUESS\tSNG.BEST.LLK\tSNG.NEXT.GUESS\t"
           "SNG.NEXT.LLK\tSNG.ONLY.POSTERIOR\tDBL.BEST.GUESS\tDBL.BEST.LLK\tDIFF.LLK.SNG.DBL\n");
  std::map<std::string, int32_t> bc_map;
  for (int64_t i = 0; i < p.C(); ++i) bc_map[p.bcs[(size_t)i]] = (int32_t)i;
  static const char* tname[3] = {"SNG", "DBL", "AMB"};
  int ncells = 0;
  for (auto it = bc_map.begin(); it != bc_map.end(); ++it, ++ncells) {
    const int32_t i = it->second;
    const muxgl_demux_cell& c = cells[(size_t)i];
    if (p.cell_totl_reads[(size_t)i] < cf.lo.minRead || p.cell_uniq_reads[(size_t)i] < cf.lo.minUMI ||
        c.nsnps < cf.lo.minSNP)
      continue;  // :641
    if (!c.valid) continue;  // :653
    auto al = [&](int n) { return (n >= 0 && n < dp.n_alpha) ? dp.alpha[n] : NAN; };
    w.printf("%d\t%s\t%u\t%d\t%s\t%s,%s,%.2lf\t%.2lf\t%s,%s,%.2lf\t%.2lf\t%.2lf\t%.2lg\t%.2lg\t%s\t%.2lf\t%s\t%.2lf\t"
             "%.5lf\t%s,%s,%.2lf\t%.2lf\t%.2lf\n",
             ncells, it->first.c_str(), (unsigned)c.nsnps, p.cell_uniq_reads[(size_t)i], tname[c.type],
             sid(p, c.jBest), sid(p, c.kBest), al(c.aBest), c.bestLLK, sid(p, c.jNext), sid(p, c.kNext), al(c.aNext),
             c.nextLLK, c.bestLLK - c.nextLLK, c.bestPP, c.sngPP, sid(p, c.sBest), c.sngBestLLK, sid(p, c.sNext),
             c.sngNextLLK, c.sngOnlyPP, sid(p, c.dBest1), sid(p, c.dBest2), al(c.dBestA), c.dblBestLLK,
             c.sngBestLLK - c.dblBestLLK);
  }
  w.close();
  tm.lap("demuxlet: write .best");
  notice("Finished writing output files");
  muxgl_destroy(h);
  return 0;
}

// ------------------------------------------------------------------------------------------------ freemuxlet
// rows [0, n) of a table, formatted by the worker pool in batches and written in order
template <class F>
void write_rows_parallel(OutFile& w, int64_t n, F format_row) {
  constexpr int64_t GRAIN = 1024, BATCH = 64;  // rows per work item, work items per batch
  std::vector<std::string> parts((size_t)BATCH);
  for (int64_t r0 = 0; r0 < n; r0 += GRAIN * BATCH) {
    const int64_t nparts = std::min<int64_t>(BATCH, (n - r0 + GRAIN - 1) / GRAIN);
    parallel_for(nparts, plp_threads(), [&](int64_t i) {
      std::string& o = parts[(size_t)i];
      o.clear();
      for (int64_t r = r0 + i * GRAIN, re = std::min(n, r + GRAIN); r < re; ++r) format_row(r, o);
    });
    for (int64_t i = 0; i < nparts; ++i) w.write(parts[(size_t)i].data(), parts[(size_t)i].size());
  }
}

// old_clust0: the .clust0.vcf.gz of freemuxlet-old (cmd_cram_freemuxlet.cpp:377-431) differs from every other cluster
// VCF in two expressions: pps = gps * gls / maxGL (:409-411) and gq = (int)(-0.1*log10(..)) (:421)
void write_cluster_vcf(const std::string& path, const Pileup& p, int K, const double* gls /* [K][S][9] */,
                       const int32_t* cnt /* [K][S][3] */, const std::vector<uint8_t>& snps_observed, const tm* ltm,
                       bool old_clust0 = false) {
  // cmd_cram_freemux2.cpp:608-658 == cmd_cram_freemuxlet.cpp:655-707
  OutFile vc(path, true);
  vc.printf("##fileformat=VCFv4.2\n");
  vc.printf("##fileDate=%04d%02d%02d\n", 1970 + ltm->tm_year, 1 + ltm->tm_mon, ltm->tm_mday);  // sic: 1970+
  vc.printf("##source=cramore-freemuxlet\n");
  for (size_t i = 0; i < p.rid2chr.size(); ++i) vc.printf("##contig=<ID=%s>\n", p.rid2chr[i].c_str());
  vc.printf("##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele Frequency\">\n");
  vc.printf("##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n");
  vc.printf("##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Phred-scale Genotype Quality\">\n");
  vc.printf("##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Read Depth\">\n");
  vc.printf("##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"Allelic Read Depth\">\n");
  vc.printf("##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"Phred-scale genotype likelihood\">\n");
  vc.printf("##FORMAT=<ID=GP,Number=G,Type=Float,Description=\"Posterior probability using pooled allele frequencies\">\n");
  vc.printf("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT");
  for (int i = 0; i < K; ++i) vc.printf("\tCLUST%d", i);
  vc.printf("\n");
  const int64_t S = p.S();
  // rows are formatted by the worker pool in batches of SNPs and written in order
  auto appendf = [](std::string& o, const char* fmt, auto... args) {
    char buf[512];
    const int n = snprintf(buf, sizeof(buf), fmt, args...);
    o.append(buf, (size_t)(n < (int)sizeof(buf) ? n : (int)sizeof(buf) - 1));
  };
  std::atomic<int> checked(0);
  static const bool check_all = getenv("POPSCLE_AMD_CHECK_FORMAT") != nullptr;
  auto format_snp = [&](int64_t v, std::string& o) {
    const SnpInfo& s = p.snps[(size_t)v];
    appendf(o, "%s\t%d\t.\t%c\t%c\t.\tPASS\tAF=%.5lf\tGT:GQ:DP:AD:PL:GP", p.rid2chr[(size_t)s.rid].c_str(), s.pos,
            s.ref, s.alt, s.af);
    const double gps[3] = {(1. - s.af) * (1. - s.af), 2. * s.af * (1. - s.af), s.af * s.af};
    for (int i = 0; i < K; ++i) {
      const double* g = &gls[((size_t)i * S + v) * 9];
      const int32_t* c = &cnt[((size_t)i * S + v) * 3];
      double maxGL = g[0];
      if (maxGL < g[4]) maxGL = g[4];
      if (maxGL < g[8]) maxGL = g[8];
      int32_t pls[3];
      pls[0] = (int32_t)(-10.0 * log10(g[0] / maxGL));
      pls[1] = (int32_t)(-10.0 * log10(g[4] / maxGL));
      pls[2] = (int32_t)(-10.0 * log10(g[8] / maxGL));
      double pps[3];
      if (old_clust0) {
        pps[0] = gps[0] * g[0] / maxGL + 1e-100;
        pps[1] = gps[1] * g[4] / maxGL + 1e-100;
        pps[2] = gps[2] * g[8] / maxGL + 1e-100;
      } else {
        pps[0] = gps[0] * (g[0] / maxGL) + 1e-100;
        pps[1] = gps[1] * (g[4] / maxGL) + 1e-100;
        pps[2] = gps[2] * (g[8] / maxGL) + 1e-100;
      }
      const double sumPP = pps[0] + pps[1] + pps[2];
      pps[0] /= sumPP;
      pps[1] /= sumPP;
      pps[2] /= sumPP;
      const int bestG = (pps[0] > pps[1]) ? (pps[0] > pps[2] ? 0 : 2) : (pps[1] > pps[2] ? 1 : 2);
      int32_t gq = old_clust0 ? (int32_t)(-0.1 * log10(1 - pps[bestG] + 1e-100))
                              : (int32_t)(-10 * log10(1.0 - pps[bestG] + 1e-100));
      if (gq > 255) gq = 255;
      // "\t%d/%d:%d:%d:%d,%d:%d,%d,%d:%.3lg,%.3lg,%.3lg" (cmd_cram_freemux2.cpp:650-652), digit for digit (util.hpp:
      // fmt_int, fmt_g3_or_printf -- the latter hands any value it is not sure of to printf itself)
      char fb[192];
      int k = 0;
      fb[k++] = '\t';
      fb[k++] = bestG == 2 ? '1' : '0';
      fb[k++] = '/';
      fb[k++] = bestG > 0 ? '1' : '0';
      fb[k++] = ':';
      k += fmt_int(gq, fb + k);
      fb[k++] = ':';
      k += fmt_int(c[0], fb + k);
      fb[k++] = ':';
      k += fmt_int(c[1], fb + k);
      fb[k++] = ',';
      k += fmt_int(c[2], fb + k);
      fb[k++] = ':';
      k += fmt_int(pls[0], fb + k);
      fb[k++] = ',';
      k += fmt_int(pls[1], fb + k);
      fb[k++] = ',';
      k += fmt_int(pls[2], fb + k);
      fb[k++] = ':';
      k += fmt_g3_or_printf(pps[0], fb + k);
      fb[k++] = ',';
      k += fmt_g3_or_printf(pps[1], fb + k);
      fb[k++] = ',';
      k += fmt_g3_or_printf(pps[2], fb + k);
      // the reference's own conversion specification stays the authority: the first fields of every file (and every
      // field with POPSCLE_AMD_CHECK_FORMAT=1) are also printed with it and compared
      if (check_all || checked.load(std::memory_order_relaxed) < 4096) {
        checked.fetch_add(1, std::memory_order_relaxed);
        char ref[192];
        const int n = snprintf(ref, sizeof(ref), "\t%d/%d:%d:%d:%d,%d:%d,%d,%d:%.3lg,%.3lg,%.3lg", bestG == 2 ? 1 : 0,
                               bestG > 0 ? 1 : 0, gq, c[0], c[1], c[2], pls[0], pls[1], pls[2], pps[0], pps[1], pps[2]);
        if (n != k || memcmp(ref, fb, (size_t)k) != 0) {
          fb[k] = 0;
          fatal("cluster VCF: the fast number formatting wrote '%s' where printf writes '%s'", fb + 1, ref + 1);
        }
      }
      o.append(fb, (size_t)k);
    }
    o.push_back('\n');
  };
  constexpr int64_t GRAIN = 256, BATCH = 64;  // SNPs per work item, work items per batch
  std::vector<std::string> parts((size_t)BATCH);
  for (int64_t v0 = 0; v0 < S; v0 += GRAIN * BATCH) {
    const int64_t nparts = std::min<int64_t>(BATCH, (S - v0 + GRAIN - 1) / GRAIN);
    parallel_for(nparts, plp_threads(), [&](int64_t i) {
      std::string& o = parts[(size_t)i];
      o.clear();
      const int64_t vb = v0 + i * GRAIN, ve = std::min(S, vb + GRAIN);
      for (int64_t v = vb; v < ve; ++v)
        if (snps_observed[(size_t)v]) format_snp(v, o);
    });
    for (int64_t i = 0; i < nparts; ++i) vc.write(parts[(size_t)i].data(), parts[(size_t)i].size());
  }
  vc.close();
}

// snps_observed[v] = 1 for every marker some droplet covers (cmd_cram_freemux2.cpp:279-288).  Threads over the entries;
// the flags are accessed with relaxed atomics (several threads may store the same 1), and a set flag is only read, so that
// its cache line stays shared between the threads.
static void mark_observed_snps(const Pileup& p, std::vector<uint8_t>& snps_observed) {
  const int64_t nnz = p.nnz(), grain = 1 << 20;
  const int32_t* es = p.entry_snp.data();
  uint8_t* so = snps_observed.data();
  parallel_for((nnz + grain - 1) / grain, plp_threads(), [&](int64_t b) {
    for (int64_t e = b * grain, e1 = std::min(nnz, (b + 1) * grain); e < e1; ++e)
      if (!__atomic_load_n(so + es[e], __ATOMIC_RELAXED)) __atomic_store_n(so + es[e], (uint8_t)1, __ATOMIC_RELAXED);
  });
}

int cmd_freemuxlet(int argc, char** argv) {
  CommonFlags cf;
  std::string initClusterFile;
  double doublet_prior = 0.5, geno_error = 0.1, bfThres = 5.41, fracInitClust = 1.0;  // cmd_cram_freemux2.cpp:19-28
  double singletScoreThres = -1e300;
  int32_t nSamples = 0, initIteration = 10, randomSeed = 0, verbose = 0, maxIter = 10;
  bool auxFiles = false, keepInitMissing = false, randomizeSingletScore = false, noEarlyStop = false;
  Args a;
  cf.add(a);
  a.add_string("init-cluster", &initClusterFile);
  a.add_int("nsample", &nSamples);
  a.add_bool("aux-files", &auxFiles);
  a.add_int("verbose", &verbose);
  a.add_double("doublet-prior", &doublet_prior);
  a.add_double("geno-error", &geno_error);
  a.add_double("bf-thres", &bfThres);              // accepted, unused (as in the reference)
  a.add_double("frac-init-clust", &fracInitClust);
  a.add_int("iter-init", &initIteration);          // accepted, unused
  a.add_bool("keep-init-missing", &keepInitMissing);  // accepted, unused
  a.add_bool("randomize-singlet-score", &randomizeSingletScore);
  a.add_int("seed", &randomSeed);
  a.add_int("max-iter", &maxIter);                 // extension: the reference hard-codes 10 (:372)
  a.add_bool("no-early-stop", &noEarlyStop);       // extension for benchmarking fixed iteration counts
  a.parse(argc, argv);
  if (cf.plpPrefix.empty() || cf.outPrefix.empty() || nSamples == 0) fatal("Missing required option(s) : --plp, --out, --nsample");

  Pileup p;
  StageTimer tmr;
  load_from_plp(cf.plpPrefix, cf.lo, nullptr, p);
  tmr.lap("freemuxlet: load");
  const int64_t C = p.C(), S = p.S();
  const int K = nSamples;

  std::map<std::string, int32_t> initCluster;  // :90-104
  if (!initClusterFile.empty()) {
    TsvReader t(initClusterFile);
    while (t.read_line() > 0) {
      if (t.nfields != 2) fatal("ERROR: Initial clustering file %s has to have 2 columnes", initClusterFile.c_str());
      const int32_t ic = t.int_field_at(1);
      if (ic >= 0) {
        if (ic >= K)
          fatal("ERROR: --nsample %d parameter was set. The cluster ID must be between 0 to %d, or use negative values "
                "to not assign initial cluster (not implemented yet)", K, K - 1);
        initCluster[t.str_field_at(0)] = ic;
      }
    }
  }

  muxgl_config cfg = cf.config(0);
  muxgl_handle* h = nullptr;
  std::vector<double> af((size_t)S), llk0((size_t)C), llk2((size_t)C);
  std::vector<int32_t> nSNPs((size_t)C), nReads((size_t)C);
  for (int64_t s = 0; s < S; ++s) af[(size_t)s] = p.snps[(size_t)s].af;
  // A device group with a greedy start: the greedy procedure is sequential over all cells, each step scoring one cell
  // against the cluster pileups of all earlier ones, so it runs on ONE device holding the whole pileup (the group's
  // first) -- BEFORE the group exists, so that this device never holds the whole pileup next to its slabs.  The scores
  // it starts from (muxgl_fmx_prepare) are the same numbers on either kind of handle.
  const bool greedy_first = cf.grouped() && initClusterFile.empty();
  muxgl_handle* g = nullptr;
  if (greedy_first) {
    muxgl_config one = MUXGL_CONFIG_INIT;
    one.device_id = cfg.device_ids[0];
    if (muxgl_create(&one, &g) != 0) fatal("%s", muxgl_last_error(nullptr));
    upload(g, p);
    tmr.lap("freemuxlet: device init+hand-over (one device, for the greedy start)");
    check(g, muxgl_fmx_prepare(g, af.data(), llk0.data(), llk2.data(), nSNPs.data(), nReads.data()), "muxgl_fmx_prepare");
  } else {
    if (muxgl_create(&cfg, &h) != 0) fatal("%s", muxgl_last_error(nullptr));
    upload(h, p);
    tmr.lap("freemuxlet: device init+hand-over");
    check(h, muxgl_fmx_prepare(h, af.data(), llk0.data(), llk2.data(), nSNPs.data(), nReads.data()), "muxgl_fmx_prepare");
  }
  tmr.lap("freemuxlet: muxgl_fmx_prepare");

  std::vector<double> scores((size_t)C);
  {  // .lmix, :111-163
    OutFile wmix(cf.outPrefix + ".lmix", false);
    wmix.printf("INT_ID\tBARCODE\tNSNPs\tNREADs\tDBL.LLK\tSNG.LLK\tBF.SINGLET\tBF.SINGLET.PER.SNP\n");
    for (int64_t i = 0; i < C; ++i) {
      scores[(size_t)i] = llk2[(size_t)i] - llk0[(size_t)i];
      wmix.printf("%d\t%s\t%d\t%d\t%.2lf\t%.2lf\t%.2lf\t%.4lf\n", (int)i, p.bcs[(size_t)i].c_str(), nSNPs[(size_t)i],
                  nReads[(size_t)i], llk0[(size_t)i], llk2[(size_t)i], llk2[(size_t)i] - llk0[(size_t)i],
                  (llk2[(size_t)i] - llk0[(size_t)i]) / nSNPs[(size_t)i]);
    }
  }
  RefRand rng(randomSeed == 0 ? (unsigned)std::time(0) : (unsigned)randomSeed);  // srand(), :165-168
  if (randomizeSingletScore) {  // :171-181
    for (int64_t i = 0; i < C - 1; ++i) {
      const int64_t j = i + rng.next() % (C - i);
      if (i < j) std::swap(scores[(size_t)i], scores[(size_t)j]);
    }
  }

  std::vector<int32_t> clusts((size_t)C, -1);
  if (!initClusterFile.empty()) {  // :198-216
    int32_t nmiss = 0;
    for (int64_t i = 0; i < C; ++i) {
      auto it = initCluster.find(p.bcs[(size_t)i]);
      if (it == initCluster.end()) ++nmiss;
      else clusts[(size_t)i] = it->second;
    }
    if (nmiss > 0) notice("WARNING: %d of %d droplets do not have initial cluster assignment", nmiss, (int)C);
  } else if (!greedy_first) {  // greedy clustering, :217-261
    check(h, muxgl_fmx_greedy_init(h, K, scores.data(), fracInitClust, singletScoreThres, clusts.data()),
          "muxgl_fmx_greedy_init");
  } else {  // on the one-device handle; then the group is made and takes over
    check(g, muxgl_fmx_greedy_init(g, K, scores.data(), fracInitClust, singletScoreThres, clusts.data()),
          "muxgl_fmx_greedy_init");
    muxgl_destroy(g);
    g = nullptr;
    if (muxgl_create(&cfg, &h) != 0) fatal("%s", muxgl_last_error(nullptr));
    upload(h, p);
    check(h, muxgl_fmx_prepare(h, af.data(), nullptr, nullptr, nullptr, nullptr), "muxgl_fmx_prepare");
    tmr.lap("freemuxlet: device group init+hand-over");
  }
  tmr.lap("freemuxlet: .lmix + initial clusters");
  notice("Finished assigning initial identity of the cluster..");
  if (auxFiles) {  // :265-274
    OutFile wc0(cf.outPrefix + ".clust0.samples.gz", true);
    wc0.printf("INT_ID\tBARCODE\tCLUST0\n");
    for (int64_t i = 0; i < C; ++i) wc0.printf("%d\t%s\t%d\n", (int)i, p.bcs[(size_t)i].c_str(), clusts[(size_t)i]);
  }
  std::vector<uint8_t> snps_observed((size_t)S, 0);  // :279-288
  mark_observed_snps(p, snps_observed);
  check(h, muxgl_fmx_set_clusters(h, K, clusts.data()), "muxgl_fmx_set_clusters");
  time_t now = std::time(nullptr);
  tm* ltm = localtime(&now);
  BigVec<double> cgls((size_t)K * S * 9);  // (not zero-filled: 2.3 GB at configs[4]; muxgl_fmx_get_cluster_pileup writes all of it)
  BigVec<int32_t> ccnt((size_t)K * S * 3);
  if (auxFiles) {
    check(h, muxgl_fmx_get_cluster_pileup(h, cgls.data(), ccnt.data()), "muxgl_fmx_get_cluster_pileup");
    write_cluster_vcf(cf.outPrefix + ".clust0.vcf.gz", p, K, cgls.data(), ccnt.data(), snps_observed, ltm);
  }

  std::vector<muxgl_fmx_cell> cells((size_t)C);
  muxgl_fmx_params fp{doublet_prior, geno_error};
  tmr.lap("freemuxlet: set_clusters");
  for (int32_t iter = 0; iter < maxIter; ++iter) {  // :373-605
    notice("Inferring doublets and refining clusters.., iter = %d", iter + 1);
    int32_t nsingle = 0, namb = 0, nchanged = 0;
    check(h, muxgl_fmx_iterate(h, &fp, cells.data(), &nsingle, &namb, &nchanged, nullptr), "muxgl_fmx_iterate");
    notice("Refining per-cluster genotype likelihoods.... %d singlets, %d doublets, %d ambiguous, and %d changed", nsingle,
           (int)C - nsingle - namb, namb, nchanged);
    if (nchanged == 0 && !noEarlyStop) {
      notice("No more changes in cluster assginment and singlet identities. Finishing iterations early");
      break;
    }
  }
  tmr.lap("freemuxlet: EM iterations");
  check(h, muxgl_fmx_get_cluster_pileup(h, cgls.data(), ccnt.data()), "muxgl_fmx_get_cluster_pileup");
  write_cluster_vcf(cf.outPrefix + ".clust1.vcf.gz", p, K, cgls.data(), ccnt.data(), snps_observed, ltm);
  tmr.lap("freemuxlet: write .clust1.vcf.gz");

  OutFile wc1(cf.outPrefix + ".clust1.samples.gz", true);  // :660-665
  wc1.printf("INT_ID\tBARCODE\tNUM.SNPS\tNUM.READS\tDROPLET.TYPE\tBEST.GUESS\tBEST.LLK\tNEXT.GUESS\tNEXT.LLK\t"
             "DIFF.LLK.BEST.NEXT\tBEST.POSTERIOR\tSNG.POSTERIOR\tSNG.BEST.GUESS\tSNG.BEST.LLK\tSNG.NEXT.GUESS\t"
             "SNG.NEXT.LLK\tSNG.ONLY.POSTERIOR\tDBL.BEST.GUESS\tDBL.BEST.LLK\tDIFF.LLK.SNG.DBL\n");
  write_rows_parallel(wc1, C, [&](int64_t i, std::string& o) {  // rows formatted by the worker pool, written in order
    const muxgl_fmx_cell& c = cells[(size_t)i];
    char buf[1024];
    const int n = snprintf(buf, sizeof(buf),
               "%d\t%s\t%d\t%d\t%s\t%d,%d\t%.2lf\t%d,%d\t%.2lf\t%.2lf\t%.5lf\t%.2lg\t%d\t%.2lf\t%d\t%.2lf\t%.5lf\t%d,%d\t"
               "%.2lf\t%.2lf\n",
               (int)i, p.bcs[(size_t)i].c_str(), nSNPs[(size_t)i], nReads[(size_t)i],
               (c.type == 2) ? "AMB" : ((c.type == 0) ? "SNG" : "DBL"), c.jBest, c.kBest, c.bestLLK, c.jNext, c.kNext,
               c.nextLLK, c.bestLLK - c.nextLLK, c.bestPP, c.sngPP, c.sBest, c.sngBestLLK, c.sNext, c.sngNextLLK,
               c.sngOnlyPP, c.dBest1, c.dBest2, c.dblBestLLK, c.sngBestLLK - c.dblBestLLK);
    if (n < (int)sizeof(buf)) {
      o.append(buf, (size_t)n);
    } else {  // (a barcode of a kilobyte)
      std::vector<char> big((size_t)n + 1);
      snprintf(big.data(), big.size(),
               "%d\t%s\t%d\t%d\t%s\t%d,%d\t%.2lf\t%d,%d\t%.2lf\t%.2lf\t%.5lf\t%.2lg\t%d\t%.2lf\t%d\t%.2lf\t%.5lf\t%d,%d\t"
               "%.2lf\t%.2lf\n",
               (int)i, p.bcs[(size_t)i].c_str(), nSNPs[(size_t)i], nReads[(size_t)i],
               (c.type == 2) ? "AMB" : ((c.type == 0) ? "SNG" : "DBL"), c.jBest, c.kBest, c.bestLLK, c.jNext, c.kNext,
               c.nextLLK, c.bestLLK - c.nextLLK, c.bestPP, c.sngPP, c.sBest, c.sngBestLLK, c.sNext, c.sngNextLLK,
               c.sngOnlyPP, c.dBest1, c.dBest2, c.dblBestLLK, c.sngBestLLK - c.dblBestLLK);
      o.append(big.data(), (size_t)n);
    }
  });
  wc1.close();
  tmr.lap("freemuxlet: write .clust1.samples.gz");
  muxgl_destroy(h);
  return 0;
}

// ------------------------------------------------------------------------------------------------ freemuxlet-old
// mirrors cmdCramFreemuxlet (cmd_cram_freemuxlet.cpp), registered as "freemuxlet-old" (cramore.cpp:44).  Differences
// from freemux2 that are reproduced as written:
//   * the read/cell filter flags are parsed and then never handed to the loader (:55-62 vs :83), so the pileup is loaded
//     with sc_dropseq_lib_t's own defaults minBQ = 1, capBQ = 60 and no droplet filters (sc_drop_seq.h:181);
//   * no srand(): rand() runs from the C library's default seed, so the command is deterministic (RefRand(1));
//   * initial clusters from votes over a pairwise distance matrix (:176-343) instead of the greedy pass;
//   * ten EM iterations without early stop; --geno-error only enters the last one (:485,500).
int cmd_freemuxlet_old(int argc, char** argv) {
  CommonFlags cf;
  std::string initClusterFile;
  double doublet_prior = 0.5, geno_error = 0.0, bfThres = 5.41, fracInitClust = 1.0;  // :19-28
  int32_t nSamples = 0, initIteration = 10, verbose = 0, minUniq = 0;
  bool auxFiles = false, keepInitMissing = false;
  Args a;
  cf.add(a);
  cf.lo.capBQ = 40;  // the flag's default (:16); see above: it never reaches the loader
  a.add_int("min-uniq", &minUniq);
  a.add_string("init-cluster", &initClusterFile);
  a.add_int("nsample", &nSamples);
  a.add_bool("aux-files", &auxFiles);
  a.add_int("verbose", &verbose);
  a.add_double("doublet-prior", &doublet_prior);
  a.add_double("geno-error", &geno_error);
  a.add_double("bf-thres", &bfThres);
  a.add_double("frac-init-clust", &fracInitClust);
  a.add_int("iter-init", &initIteration);
  a.add_bool("keep-init-missing", &keepInitMissing);
  a.parse(argc, argv);
  if (cf.plpPrefix.empty() || cf.outPrefix.empty() || nSamples == 0) fatal("Missing required option(s) : --plp, --out, --nsample");
  if (nSamples > 64) fatal("freemuxlet-old: --nsample %d exceeds the 64 clusters the vote kernel supports", nSamples);

  Pileup p;
  StageTimer tmr;
  LoadOptions lo;  // sc_drop_seq.h:181
  lo.minBQ = 1;
  lo.capBQ = 60;
  load_from_plp(cf.plpPrefix, lo, nullptr, p);
  tmr.lap("freemuxlet-old: load");
  const int64_t C = p.C(), S = p.S();
  const int K = nSamples;

  std::map<std::string, int32_t> initCluster;  // :87-99
  if (!initClusterFile.empty()) {
    TsvReader t(initClusterFile);
    while (t.read_line() > 0) {
      if (t.nfields != 2) fatal("ERROR: Initial clustering file %s has to have 2 columnes", initClusterFile.c_str());
      const int32_t ic = t.int_field_at(1);
      if (ic >= 0) {
        if (ic >= K)
          fatal("ERROR: --nsample %d parameter was set. The cluster ID must be between 0 to %d, or use negative values "
                "to not assign initial cluster (not implemented yet)", K, K - 1);
        initCluster[t.str_field_at(0)] = ic;
      }
    }
  }

  if (cf.grouped()) fatal("freemuxlet-old: --devices is not supported (its pairwise matrix and votes run on one device)");
  muxgl_config cfg = cf.config(0);
  muxgl_handle* h = nullptr;
  if (muxgl_create(&cfg, &h) != 0) fatal("%s", muxgl_last_error(nullptr));
  upload(h, p);
  std::vector<double> af((size_t)S), llk0((size_t)C), llk2((size_t)C);
  std::vector<int32_t> nSNPs((size_t)C), nReads((size_t)C);
  for (int64_t s = 0; s < S; ++s) af[(size_t)s] = p.snps[(size_t)s].af;
  check(h, muxgl_fmx_prepare(h, af.data(), llk0.data(), llk2.data(), nSNPs.data(), nReads.data()), "muxgl_fmx_prepare");
  tmr.lap("freemuxlet-old: hand-over + muxgl_fmx_prepare");

  std::vector<double> scores((size_t)C);
  {  // .lmix, :110-165
    OutFile wmix(cf.outPrefix + ".lmix", false);
    wmix.printf("INT_ID\tBARCODE\tNSNPs\tNREADs\tDBL.LLK\tSNG.LLK\tLOG.BF\tBFpSNP\n");
    for (int64_t i = 0; i < C; ++i) {
      scores[(size_t)i] = llk2[(size_t)i] - llk0[(size_t)i];
      wmix.printf("%d\t%s\t%d\t%d\t%.2lf\t%.2lf\t%.2lf\t%.4lf\n", (int)i, p.bcs[(size_t)i].c_str(), nSNPs[(size_t)i],
                  nReads[(size_t)i], llk0[(size_t)i], llk2[(size_t)i], llk0[(size_t)i] - llk2[(size_t)i],
                  (llk0[(size_t)i] - llk2[(size_t)i]) / nSNPs[(size_t)i]);
    }
  }
  std::vector<int32_t> drops_srted((size_t)C);  // :167-173, comparator sc_drop_seq.h:190-198
  for (int64_t i = 0; i < C; ++i) drops_srted[(size_t)i] = (int32_t)i;
  std::sort(drops_srted.begin(), drops_srted.end(), [&](int32_t lhs, int32_t rhs) {
    const double cmp = scores[(size_t)lhs] - scores[(size_t)rhs];
    if (cmp != 0) return cmp > 0;
    return lhs > rhs;
  });

  notice("Calculate pairwise genetic distance matrix..");
  const bool wantDist = auxFiles && initClusterFile.empty();  // .ldist.gz rows are printed by the voting loop only (:262-265)
  std::vector<muxgl_dropd> dropDs(wantDist ? (size_t)(C * (C - 1) / 2) : 0);
  check(h, muxgl_fmxold_pair_dist(h, bfThres, wantDist ? dropDs.data() : nullptr), "muxgl_fmxold_pair_dist");
  tmr.lap("freemuxlet-old: pairwise distance matrix");

  std::vector<int32_t> clusts((size_t)C, -1), ccounts((size_t)K, 0);
  std::vector<double> jitter;
  RefRand rng(1);
  auto draw_jitter = [&](int64_t rows) {  // `votes[j] = rand()/(RAND_MAX+1.)/1000.` per visited cell (:257-259,304-306)
    jitter.resize((size_t)rows * K);
    for (size_t x = 0; x < jitter.size(); ++x) jitter[x] = rng.next() / (RAND_MAX + 1.) / 1000.;
  };
  auto counts_str = [&]() {
    std::string buf;
    for (int j = 0; j < K; ++j) buf += " " + std::to_string(ccounts[(size_t)j]);
    return buf;
  };
  {
    OutFile* wdist = nullptr;
    if (auxFiles) {  // :178-182
      wdist = new OutFile(cf.outPrefix + ".ldist.gz", true);
      wdist->printf("ID1\tID2\tNSNP\tREAD1\tREAD2\tREADMIN\tLLK0\tLLK2\tLDIFF\tDIFF.SNP\n");
    }
    if (!initClusterFile.empty()) {  // :227-243
      int32_t nmiss = 0;
      for (int64_t i = 0; i < C; ++i) {
        auto it = initCluster.find(p.bcs[(size_t)i]);
        if (it == initCluster.end()) ++nmiss;
        else {
          clusts[(size_t)i] = it->second;
          ++ccounts[(size_t)it->second];
        }
      }
      if (nmiss > 0) notice("WARNING: %d of %d droplets do not have initial cluster assignment", nmiss, (int)C);
    } else {  // :245-291
      int64_t nvis = 0;
      for (int64_t i = 0; i < C; ++i)
        if (!((double)i > (double)C * fracInitClust)) ++nvis;
      draw_jitter(nvis);
      check(h, muxgl_fmxold_vote_init(h, K, drops_srted.data(), jitter.data(), fracInitClust, clusts.data(), ccounts.data()),
            "muxgl_fmxold_vote_init");
      if (wdist) {  // :262-265
        for (int64_t i = 0; i < nvis; ++i) {
          const int32_t si = drops_srted[(size_t)i];
          for (int64_t j = 0; j < i; ++j) {
            const int32_t sj = drops_srted[(size_t)j];
            const int64_t hi = si > sj ? si : sj, lo2 = si > sj ? sj : si;
            const muxgl_dropd& dd = dropDs[(size_t)(hi * (hi - 1) / 2 + lo2)];
            wdist->printf("%d\t%d\t%d\t%d\t%d\t%d\t%.2lf\t%.2lf\t%.2lf\t%.4lf\n", si, sj, dd.nsnps, dd.nread1, dd.nread2,
                          dd.nread1 > dd.nread2 ? dd.nread2 : dd.nread1, dd.llk0, dd.llk2, dd.llk2 - dd.llk0,
                          (dd.llk2 - dd.llk0) / (dd.nsnps + 1e-6));
          }
        }
      }
    }
    if (wdist) {
      wdist->close();
      delete wdist;
    }
  }
  notice("Finished calculating pairwise distance between the droplets..");
  tmr.lap("freemuxlet-old: first voting pass");

  if (initIteration > 0) {  // :295-351: ten passes whatever the value
    std::vector<int32_t> orand((size_t)C);
    for (int32_t iter = 0; iter < 10; ++iter) {
      for (int64_t i = 0; i < C; ++i) orand[(size_t)i] = (int32_t)i;
      // std::random_shuffle(orand.begin(), orand.end()) as libstdc++ implements it (bits/stl_algo.h): the reference
      // binary's behaviour; spelled out because the function is gone from C++17
      for (int64_t i = 1; i < C; ++i) {
        const int64_t j = rng.next() % (i + 1);
        if (i != j) std::swap(orand[(size_t)i], orand[(size_t)j]);
      }
      draw_jitter(C);
      int32_t changed = 0;
      check(h, muxgl_fmxold_vote_refine(h, K, orand.data(), jitter.data(), keepInitMissing ? 1 : 0, clusts.data(), &changed,
                                        ccounts.data()),
            "muxgl_fmxold_vote_refine");
      notice("Iteration %d, # changed = %d, cluster counts:%s", iter, changed, counts_str().c_str());
    }
  }
  tmr.lap("freemuxlet-old: refinement passes");

  if (auxFiles) {  // :353-363
    OutFile wc0(cf.outPrefix + ".clust0.samples.gz", true);
    wc0.printf("INT_ID\tBARCODE\tCLUST0\n");
    for (int64_t i = 0; i < C; ++i) wc0.printf("%d\t%s\t%d\n", (int)i, p.bcs[(size_t)i].c_str(), clusts[(size_t)i]);
  }
  std::vector<uint8_t> snps_observed((size_t)S, 0);  // :367-376
  mark_observed_snps(p, snps_observed);
  check(h, muxgl_fmx_set_clusters(h, K, clusts.data()), "muxgl_fmx_set_clusters");
  time_t now = std::time(nullptr);
  tm* ltm = localtime(&now);
  BigVec<double> cgls((size_t)K * S * 9);  // (not zero-filled: 2.3 GB at configs[4]; muxgl_fmx_get_cluster_pileup writes all of it)
  BigVec<int32_t> ccnt((size_t)K * S * 3);
  if (auxFiles) {
    check(h, muxgl_fmx_get_cluster_pileup(h, cgls.data(), ccnt.data()), "muxgl_fmx_get_cluster_pileup");
    write_cluster_vcf(cf.outPrefix + ".clust0.vcf.gz", p, K, cgls.data(), ccnt.data(), snps_observed, ltm, true);
  }

  std::vector<muxgl_fmx_cell> cells((size_t)C);
  const int32_t max_iter = 10;  // :457
  for (int32_t iter = 0; iter < max_iter; ++iter) {
    notice("Inferring doublets and refining clusters.., iter = %d", iter + 1);
    muxgl_fmx_params fp{doublet_prior, (geno_error > 0 && iter + 1 == max_iter) ? geno_error : 0.0};  // :485,500
    int32_t nsingle = 0, namb = 0, nchanged = 0;
    check(h, muxgl_fmx_iterate(h, &fp, cells.data(), &nsingle, &namb, &nchanged, nullptr), "muxgl_fmx_iterate");
    notice("Refining per-cluster genotype likelihoods.... %d singlets, %d doublets, and %d ambiguous", nsingle,
           (int)C - nsingle - namb, namb);
  }
  tmr.lap("freemuxlet-old: EM iterations");
  check(h, muxgl_fmx_get_cluster_pileup(h, cgls.data(), ccnt.data()), "muxgl_fmx_get_cluster_pileup");
  write_cluster_vcf(cf.outPrefix + ".clust1.vcf.gz", p, K, cgls.data(), ccnt.data(), snps_observed, ltm);

  OutFile wc1(cf.outPrefix + ".clust1.samples.gz", true);  // :709-714
  wc1.printf("INT_ID\tBARCODE\tNUM.SNPS\tNUM.READS\tDROPLET.TYPE\tBEST.GUESS\tBEST.LLK\tNEXT.GUESS\tNEXT.LLK\t"
             "DIFF.LLK.BEST.NEXT\tBEST.POSTERIOR\tSNG.POSTERIOR\tSNG.BEST.GUESS\tSNG.BEST.LLK\tSNG.NEXT.GUESS\t"
             "SNG.NEXT.LLK\tSNG.ONLY.POSTERIOR\tDBL.BEST.GUESS\tDBL.BEST.LLK\tDIFF.LLK.SNG.DBL\n");
  write_rows_parallel(wc1, C, [&](int64_t i, std::string& o) {  // rows formatted by the worker pool, written in order
    const muxgl_fmx_cell& c = cells[(size_t)i];
    char buf[1024];
    const int n = snprintf(buf, sizeof(buf),
               "%d\t%s\t%d\t%d\t%s\t%d,%d\t%.2lf\t%d,%d\t%.2lf\t%.2lf\t%.5lf\t%.2lg\t%d\t%.2lf\t%d\t%.2lf\t%.5lf\t%d,%d\t"
               "%.2lf\t%.2lf\n",
               (int)i, p.bcs[(size_t)i].c_str(), nSNPs[(size_t)i], nReads[(size_t)i],
               (c.type == 2) ? "AMB" : ((c.type == 0) ? "SNG" : "DBL"), c.jBest, c.kBest, c.bestLLK, c.jNext, c.kNext,
               c.nextLLK, c.bestLLK - c.nextLLK, c.bestPP, c.sngPP, c.sBest, c.sngBestLLK, c.sNext, c.sngNextLLK,
               c.sngOnlyPP, c.dBest1, c.dBest2, c.dblBestLLK, c.sngBestLLK - c.dblBestLLK);
    if (n < (int)sizeof(buf)) {
      o.append(buf, (size_t)n);
    } else {  // (a barcode of a kilobyte)
      std::vector<char> big((size_t)n + 1);
      snprintf(big.data(), big.size(),
               "%d\t%s\t%d\t%d\t%s\t%d,%d\t%.2lf\t%d,%d\t%.2lf\t%.2lf\t%.5lf\t%.2lg\t%d\t%.2lf\t%d\t%.2lf\t%.5lf\t%d,%d\t"
               "%.2lf\t%.2lf\n",
               (int)i, p.bcs[(size_t)i].c_str(), nSNPs[(size_t)i], nReads[(size_t)i],
               (c.type == 2) ? "AMB" : ((c.type == 0) ? "SNG" : "DBL"), c.jBest, c.kBest, c.bestLLK, c.jNext, c.kNext,
               c.nextLLK, c.bestLLK - c.nextLLK, c.bestPP, c.sngPP, c.sBest, c.sngBestLLK, c.sNext, c.sngNextLLK,
               c.sngOnlyPP, c.dBest1, c.dBest2, c.dblBestLLK, c.sngBestLLK - c.dblBestLLK);
      o.append(big.data(), (size_t)n);
    }
  });
  wc1.close();
  tmr.lap("freemuxlet-old: writers");
  muxgl_destroy(h);
  return 0;
}

// ------------------------------------------------------------------------------------------------ dump-plp
// loader-only command for the CPU tests: writes the packed pileup as a flat little-endian file
//   magic "MUXGLPLP", int64 C,S,nnz,R,nv, then cell_ptr, entry_snp, entry_rptr, reads, af[S], has_gp[S], gp[S*nv*3],
//   cell_totl_reads[C], cell_uniq_reads[C], then C barcodes and nv sample ids as NUL-terminated strings
int cmd_dump_plp(int argc, char** argv) {
  CommonFlags cf;
  VcfReader vr;
  std::vector<std::string> smIDs;
  std::string smList;
  Args a;
  cf.add(a);
  a.add_string("vcf", &vr.path);
  a.add_string("field", &cf.lo.field);
  a.add_double("geno-error-offset", &cf.lo.genoErrorOffset);
  a.add_double("geno-error-coeff", &cf.lo.genoErrorCoeffR2);
  a.add_string("r2-info", &cf.lo.r2info);
  a.add_int("min-mac", &vr.vfilt.minMAC);
  a.add_double("min-callrate", &vr.vfilt.minCallRate);
  a.add_multi_string("sm", &smIDs);
  a.add_string("sm-list", &smList);
  a.add_int("rank", &cf.lo.rank);    // one rank's two slabs of a sharded run (LoadOptions): the file then holds the row
  a.add_int("world", &cf.lo.world);  // slab in place of the pileup and the column slab behind it
  a.parse(argc, argv);
  if (cf.plpPrefix.empty() || cf.outPrefix.empty()) fatal("Missing required option(s) : --plp, --out");
  Pileup p;
  if (!vr.path.empty()) {
    vr.wanted = smIDs;
    if (!smList.empty()) {
      TsvReader t(smList);
      while (t.read_line() > 0) vr.wanted.push_back(t.str_field_at(0));
    }
    vr.init();
    load_from_plp(cf.plpPrefix, cf.lo, &vr, p);
  } else {
    load_from_plp(cf.plpPrefix, cf.lo, nullptr, p);
  }
  FILE* f = fopen(cf.outPrefix.c_str(), "wb");
  if (!f) fatal("Cannot open %s for writing", cf.outPrefix.c_str());
  if (p.slabbed) {  // --rank / --world: magic "MUXGLSLB", int64 C, S, c0, c1, s0, s1, nnz_r, R_r, nnz_c, R_c; the row slab's
                    // cell_ptr[c1-c0+1], entry_snp, entry_rptr, reads; the column slab's cell_ptr[C+1], ...; af[S]; the
                    // two read counts [C]
    const int64_t hdr[10] = {p.C(), p.S(), p.slab_c0, p.slab_c1, p.slab_s0, p.slab_s1, p.nnz(), (int64_t)p.reads.size(),
                             (int64_t)p.col_entry_snp.size(), (int64_t)p.col_reads.size()};
    fwrite("MUXGLSLB", 1, 8, f);
    fwrite(hdr, sizeof(int64_t), 10, f);
    fwrite(p.cell_ptr.data(), sizeof(int64_t), p.cell_ptr.size(), f);
    fwrite(p.entry_snp.data(), sizeof(int32_t), p.entry_snp.size(), f);
    fwrite(p.entry_rptr.data(), sizeof(int64_t), p.entry_rptr.size(), f);
    fwrite(p.reads.data(), 1, p.reads.size(), f);
    fwrite(p.col_cell_ptr.data(), sizeof(int64_t), p.col_cell_ptr.size(), f);
    fwrite(p.col_entry_snp.data(), sizeof(int32_t), p.col_entry_snp.size(), f);
    fwrite(p.col_entry_rptr.data(), sizeof(int64_t), p.col_entry_rptr.size(), f);
    fwrite(p.col_reads.data(), 1, p.col_reads.size(), f);
    std::vector<double> af((size_t)p.S());
    for (int64_t s = 0; s < p.S(); ++s) af[(size_t)s] = p.snps[(size_t)s].af;
    fwrite(af.data(), sizeof(double), af.size(), f);
    fwrite(p.cell_totl_reads.data(), sizeof(int32_t), p.cell_totl_reads.size(), f);
    fwrite(p.cell_uniq_reads.data(), sizeof(int32_t), p.cell_uniq_reads.size(), f);
    fclose(f);
    return 0;
  }
  const int64_t hdr[5] = {p.C(), p.S(), p.nnz(), (int64_t)p.reads.size(), p.nv};
  fwrite("MUXGLPLP", 1, 8, f);
  fwrite(hdr, sizeof(int64_t), 5, f);
  fwrite(p.cell_ptr.data(), sizeof(int64_t), p.cell_ptr.size(), f);
  fwrite(p.entry_snp.data(), sizeof(int32_t), p.entry_snp.size(), f);
  fwrite(p.entry_rptr.data(), sizeof(int64_t), p.entry_rptr.size(), f);
  fwrite(p.reads.data(), 1, p.reads.size(), f);
  std::vector<double> af((size_t)p.S());
  for (int64_t s = 0; s < p.S(); ++s) af[(size_t)s] = p.snps[(size_t)s].af;
  fwrite(af.data(), sizeof(double), af.size(), f);
  std::vector<uint8_t> hg = p.has_gp;
  hg.resize((size_t)p.S(), 0);
  fwrite(hg.data(), 1, hg.size(), f);
  fwrite(p.gp.data(), sizeof(double), p.gp.size(), f);
  fwrite(p.cell_totl_reads.data(), sizeof(int32_t), p.cell_totl_reads.size(), f);
  fwrite(p.cell_uniq_reads.data(), sizeof(int32_t), p.cell_uniq_reads.size(), f);
  for (const std::string& s : p.bcs) fwrite(s.c_str(), 1, s.size() + 1, f);
  for (const std::string& s : p.sample_ids) fwrite(s.c_str(), 1, s.size() + 1, f);
  fclose(f);
  return 0;
}

// ------------------------------------------------------------------------------------------------ selftest-fmt
// fmt_g3 / fmt_int against printf on N random values plus the hard ones (exact ties of the third digit and their
// neighbours, powers of ten, both ends of the "%e" / "%f" switch); prints the counts, exit status 1 on any difference
int cmd_selftest_fmt(int argc, char** argv) {
  const long N = argc > 0 ? atol(argv[0]) : 1000000;
  uint64_t st = 0x9e3779b97f4a7c15ull;
  auto rnd = [&]() {  // xorshift64*
    st ^= st >> 12, st ^= st << 25, st ^= st >> 27;
    return st * 2685821657736338717ull;
  };
  auto unif = [&]() { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); };
  long n = 0, declined = 0, bad = 0;
  char a[64], b[64];
  auto check = [&](double x) {
    ++n;
    const int k = fmt_g3(x, a);
    if (k < 0) {
      ++declined;
      return;
    }
    a[k] = 0;
    snprintf(b, sizeof(b), "%.3lg", x);
    if (strcmp(a, b)) {
      if (bad++ < 20) fprintf(stderr, "fmt_g3 %.17g: %s, printf: %s\n", x, a, b);
    }
  };
  for (long i = 0; i < N; ++i) {
    const double u = unif();
    check(u);
    check(u * pow(10.0, -(double)(rnd() % 20)));
    check(pow(10.0, -100.0 * unif()) * (1 + u));
    const int D = 100 + (int)(rnd() % 900), E = -(int)(rnd() % 12);
    const double t = (D + 0.5) * pow(10.0, E - 2), r = D * pow(10.0, E - 2);
    check(t), check(nextafter(t, 0)), check(nextafter(t, 1e9));
    check(r), check(nextafter(r, 0)), check(nextafter(r, 1e9));
    const int32_t v = (int32_t)rnd();
    const int k = fmt_int(v, a);
    a[k] = 0;
    snprintf(b, sizeof(b), "%d", v);
    if (strcmp(a, b)) ++bad;
  }
  for (double x : {1.0, 0.9995, 0.99949999999999994, 0.9996, 1e-5, 9.995e-5, 9.9949999e-5, 1e-4, 0.0001235, 0.5, 0.125, 1e-100,
                   1e-99, 9.99e-100, 2.5e-7, 999.4, 999.5, 12.35, 100.0, 1e-290, 0.0, -1.0, 1e3, 1e-300})
    check(x);
  for (int32_t v : {0, 1, -1, 9, 10, 255, INT32_MAX, INT32_MIN}) {
    const int k = fmt_int(v, a);
    a[k] = 0;
    snprintf(b, sizeof(b), "%d", v);
    if (strcmp(a, b)) ++bad;
  }
  printf("%ld values, %ld left to printf, %ld differences\n", n, declined, bad);
  return bad != 0;
}

// ------------------------------------------------------------------------------------------------ bgzf
// writer-only command for the CPU tests: copies a file through the BGZF writer of util.hpp (line by line, as the
// commands' printf calls do)
int cmd_bgzf(int argc, char** argv) {
  std::string in, outp;
  Args a;
  a.add_string("in", &in);
  a.add_string("out", &outp);
  a.parse(argc, argv);
  if (in.empty() || outp.empty()) fatal("Missing required option(s) : --in, --out");
  FILE* f = fopen(in.c_str(), "rb");
  if (!f) fatal("Cannot open %s for reading", in.c_str());
  OutFile w(outp, true);
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, 777, f)) > 0) w.write(buf, n);  // odd-sized pieces across block borders
  fclose(f);
  w.close();
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: popscle-amd <demuxlet|freemuxlet|freemuxlet-old|dump-plp> [options]\n");
    return 1;
  }
  try {
    const std::string cmd = argv[1];
    if (cmd == "demuxlet") return cmd_demuxlet(argc - 2, argv + 2);
    if (cmd == "freemuxlet") return cmd_freemuxlet(argc - 2, argv + 2);
    if (cmd == "freemuxlet-old") return cmd_freemuxlet_old(argc - 2, argv + 2);
    if (cmd == "dump-plp") return cmd_dump_plp(argc - 2, argv + 2);
    if (cmd == "bgzf") return cmd_bgzf(argc - 2, argv + 2);
    if (cmd == "selftest-fmt") return cmd_selftest_fmt(argc - 2, argv + 2);
    if (cmd == "synth-plp") return cmd_synth_plp(argc - 2, argv + 2);
    fprintf(stderr, "Cannot recognize the command %s\n", argv[1]);
    return 1;
  } catch (const std::exception&) {
    return 1;
  }
}
