// synthplp.hpp -- `popscle-amd synth-plp`: writes a synthetic dsc-pileup data set (CEL / VAR / PLP as BGZF, the format of
// cmd_cram_dsc_pileup.cpp:438-523) and a matching genotype VCF, of the shapes BASELINE.json's configs name.  A
// measurement utility: the Python writer (popscle_amd/plpio.py) needs minutes for 10^8 PLP rows, this one seconds, so
// the loader and the commands can be timed end to end at the north_star's size.  Same generator family as
// popscle_amd/synth.py: AF ~ U(0.05, 0.95); genotypes ~ Binomial(2, AF); markers per droplet
// clip(round(LogNormal(ln mean, 0.6)), 50, 8000), drawn with replacement and de-duplicated; reads per entry
// 1 + Poisson(0.3); raw base quality ~ UniformInt[13, 40]; a tenth of the droplets mix two donors evenly; allele =
// Bernoulli(g / 2) of the source donor with 1 % flips and 0.5 % "other" alleles.  Its own counter-based random streams
// (one per droplet / marker), so the bytes do not depend on the thread count.
#pragma once
#include <cmath>

#include "util.hpp"

namespace pa {

struct SynthRng {  // splitmix64
  uint64_t s;
  explicit SynthRng(uint64_t seed, uint64_t stream) : s(seed * 0x9E3779B97F4A7C15ull ^ (stream + 1) * 0xD1B54A32D192ED03ull) { next(); }
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
  double normal() {
    const double u1 = 1.0 - uniform(), u2 = uniform();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  }
};

inline int cmd_synth_plp(int argc, char** argv) {
  int32_t C = 10000, S = 50000, V = 16, seed = 1, gzPlain = 0;
  double meanEntries = 800.0, sigma = 0.6, lambda = 0.3, doubletFrac = 0.10, flip = 0.01, other = 0.005;
  std::string out;
  Args a;
  a.add_int("cells", &C);
  a.add_int("snps", &S);
  a.add_int("samples", &V);
  a.add_int("seed", &seed);
  a.add_double("mean-entries", &meanEntries);
  a.add_string("out", &out);
  a.add_int("plain-gzip", &gzPlain);
  a.parse(argc, argv);
  if (out.empty() || C < 1 || S < 1 || V < 1) fatal("Missing required option(s) : --out PREFIX [--cells --snps --samples --seed]");
  const int nth = plp_threads();
  StageTimer tmr;

  // donors
  std::vector<double> af((size_t)S);
  std::vector<uint8_t> G((size_t)S * V);
  parallel_for_blocked(S, 4096, nth, [&](int64_t s) {
    SynthRng r((uint64_t)seed, (uint64_t)s * 2 + 1);
    af[(size_t)s] = 0.05 + 0.9 * r.uniform();
    for (int v = 0; v < V; ++v) G[(size_t)s * V + v] = (uint8_t)((r.uniform() < af[(size_t)s]) + (r.uniform() < af[(size_t)s]));
  });
  // droplets: markers, reads (allele digit + quality character per read)
  struct Cell {
    std::vector<int32_t> snp;
    std::vector<uint32_t> roff;  // reads of entry i = [roff[i], roff[i+1])
    std::string al, bq;
  };
  std::vector<Cell> cells((size_t)C);
  const int maxE = std::min<int>(8000, S), minE = std::min<int>(50, maxE);
  parallel_for_blocked(C, 64, nth, [&](int64_t c) {
    SynthRng r((uint64_t)seed, (uint64_t)c * 2);
    Cell& x = cells[(size_t)c];
    const int L = (int)std::min<double>(maxE, std::max<double>(minE, std::nearbyint(std::exp(std::log(meanEntries) + sigma * r.normal()))));
    x.snp.resize((size_t)L);
    for (int i = 0; i < L; ++i) x.snp[(size_t)i] = (int32_t)r.below((uint32_t)S);
    std::sort(x.snp.begin(), x.snp.end());
    x.snp.erase(std::unique(x.snp.begin(), x.snp.end()), x.snp.end());
    const bool dbl = r.uniform() < doubletFrac;
    const int s1 = (int)r.below((uint32_t)V), s2 = V > 1 ? (s1 + 1 + (int)r.below((uint32_t)(V - 1))) % V : s1;
    x.roff.assign(x.snp.size() + 1, 0);
    for (size_t i = 0; i < x.snp.size(); ++i) {
      int n = 1;  // 1 + Poisson(lambda), by inversion
      for (double pk = std::exp(-lambda), cum = pk, u = r.uniform(); u > cum && n < 64; ++n) {
        pk *= lambda / n;
        cum += pk;
      }
      for (int k = 0; k < n; ++k) {
        const int src = (dbl && r.uniform() < 0.5) ? s2 : s1;
        const int g = G[(size_t)x.snp[i] * V + src];
        int alle = r.uniform() < 0.5 * g ? 1 : 0;
        if (r.uniform() < flip) alle ^= 1;
        if (r.uniform() < other) alle = 2;
        x.al.push_back((char)('0' + alle));
        x.bq.push_back((char)(33 + 13 + (int)r.below(28)));
      }
      x.roff[i + 1] = (uint32_t)x.al.size();
    }
  });
  tmr.lap("synth-plp: generate");

  auto barcode = [&](int64_t c) {  // distinct 16-mers: the base-4 digits of a bijection of the droplet index
    uint64_t v = ((uint64_t)c * 0x9E3779B1ull + 0x1234567ull) & 0xFFFFFFFFull;
    std::string b(16, 'A');
    for (int k = 0; k < 16; ++k, v >>= 2) b[(size_t)k] = "ACGT"[v & 3];
    return b + "-1";
  };
  {  // CEL
    OutFile w(out + ".cel.gz", true);
    w.printf("#DROPLET_ID\tBARCODE\tNUM.READ\tNUM.UMI\tNUM.UMIwSNP\tNUM.SNP\n");
    for (int64_t c = 0; c < C; ++c) {
      const Cell& x = cells[(size_t)c];
      const int nr = (int)x.al.size();
      w.printf("%d\t%s\t%d\t%d\t%d\t%d\n", (int)c, barcode(c).c_str(), nr + 100, nr + 10, nr, (int)x.snp.size());
    }
  }
  {  // VAR
    OutFile w(out + ".var.gz", true);
    w.printf("#SNP_ID\tCHROM\tPOS\tREF\tALT\tAF\n");
    for (int64_t s = 0; s < S; ++s) w.printf("%d\t1\t%d\tA\tG\t%.5f\n", (int)s, (int)(1000 + 10 * s), af[(size_t)s]);
  }
  {  // VCF with the donors' hard calls
    OutFile w(out + ".vcf.gz", true);
    w.printf("##fileformat=VCFv4.2\n##contig=<ID=1>\n##INFO=<ID=R2,Number=1,Type=Float,Description=\"imputation r2\">\n"
             "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT");
    for (int v = 0; v < V; ++v) w.printf("\tS%d", v);
    w.printf("\n");
    std::string line;
    static const char* gts[3] = {"0/0", "0/1", "1/1"};
    for (int64_t s = 0; s < S; ++s) {
      char head[96];
      snprintf(head, sizeof(head), "1\t%d\t.\tA\tG\t.\tPASS\tR2=%.3f\tGT", (int)(1000 + 10 * s), 0.5 + 0.5 * af[(size_t)s]);
      line = head;
      for (int v = 0; v < V; ++v) {
        line += '\t';
        line += gts[G[(size_t)s * V + v]];
      }
      line += '\n';
      w.write(line.data(), line.size());
    }
  }
  tmr.lap("synth-plp: CEL, VAR, VCF");

  // PLP: rows sorted by marker, then droplet (cmd_cram_dsc_pileup.cpp:497-518): transpose, then format ranges of markers
  std::vector<int64_t> sptr((size_t)S + 1, 0);
  for (const Cell& x : cells)
    for (int32_t s : x.snp) ++sptr[(size_t)s + 1];
  for (int64_t s = 0; s < S; ++s) sptr[(size_t)s + 1] += sptr[(size_t)s];
  const int64_t nnz = sptr[(size_t)S];
  std::vector<int32_t> rcell((size_t)nnz), rpos((size_t)nnz);
  {
    std::vector<int64_t> fill(sptr.begin(), sptr.end() - 1);
    for (int64_t c = 0; c < C; ++c) {  // droplets ascending: rows of a marker come out in droplet order
      const Cell& x = cells[(size_t)c];
      for (size_t i = 0; i < x.snp.size(); ++i) {
        const int64_t o = fill[(size_t)x.snp[i]]++;
        rcell[(size_t)o] = (int32_t)c;
        rpos[(size_t)o] = (int32_t)i;
      }
    }
  }
  tmr.lap("synth-plp: transpose");
  {
    OutFile w(out + ".plp.gz", true);
    w.printf("#DROPLET_ID\tSNP_ID\tALLELES\tBASEQS\n");
    const int64_t nranges = std::max<int64_t>(1, std::min<int64_t>(S, nnz / 200000 + 1));
    const int64_t wave = 4 * nth;
    std::vector<std::string> txt((size_t)wave);
    for (int64_t r0 = 0; r0 < nranges; r0 += wave) {
      const int64_t nr = std::min(wave, nranges - r0);
      parallel_for(nr, nth, [&](int64_t k) {
        std::string& t = txt[(size_t)k];
        t.clear();
        const int64_t sa = S * (r0 + k) / nranges, sb = S * (r0 + k + 1) / nranges;
        char num[32];
        for (int64_t s = sa; s < sb; ++s)
          for (int64_t o = sptr[(size_t)s]; o < sptr[(size_t)s + 1]; ++o) {
            const Cell& x = cells[(size_t)rcell[(size_t)o]];
            const uint32_t a0 = x.roff[(size_t)rpos[(size_t)o]], a1 = x.roff[(size_t)rpos[(size_t)o] + 1];
            t.append(num, (size_t)snprintf(num, sizeof(num), "%d\t%d\t", (int)rcell[(size_t)o], (int)s));
            t.append(x.al, a0, a1 - a0);
            t += '\t';
            t.append(x.bq, a0, a1 - a0);
            t += '\n';
          }
      });
      for (int64_t k = 0; k < nr; ++k) w.write(txt[(size_t)k].data(), txt[(size_t)k].size());
    }
  }
  tmr.lap("synth-plp: PLP");
  int64_t bases = 0;
  for (const Cell& x : cells) bases += (int64_t)x.al.size();
  notice("synth-plp: %d droplets, %d markers, %d donors: %lld PLP rows, %lld bases -> %s.{cel,var,plp,vcf}.gz", (int)C, (int)S,
         (int)V, (long long)nnz, (long long)bases, out.c_str());
  return 0;
}

}  // namespace pa
