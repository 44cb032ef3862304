// vcf.hpp -- text/gz VCF reader producing per-sample genotype posteriors the way popscle's BCFFilteredReader does
// (bcf_filtered_reader.cpp), without htslib.  Scope: what `demuxlet --plp` needs from the VCF:
//   * sample selection (--sm / --sm-list; default all)                         bcf_filtered_reader.cpp:98-141
//   * variant filter: n_allele <= 2, min MAC, min call rate on GT               :505-581 (vfilt.require_GT path)
//   * parse_posteriors for --field GT / GP (or any float FORMAT key) / PL       :367-461, PL EM :250-327
//   * INFO float (R2) for --geno-error-coeff                                    sc_drop_seq.cpp:300-305
// Contig ids follow the header's ##contig order (htslib's rid), new names get the next id on first appearance.
// BCF (binary) input is not supported: convert with `bcftools view` first.
#pragma once

#include <cmath>
#include <memory>
#include <set>

#include "util.hpp"

namespace pa {

struct VcfFilter {
  int32_t minMAC = 1;        // cmd_cram_demuxlet.cpp:27
  double minCallRate = 0.5;  // :28
  int32_t maxAlleles = 2;    // :29
};

class VcfReader {
 public:
  std::string path;
  VcfFilter vfilt;
  std::vector<std::string> wanted;  // --sm / --sm-list; empty = all
  std::vector<std::string> sample_ids;  // selected: VCF column order, or sorted ID order when a subset was requested
  bool eof = false;

  // cursor
  int32_t rid = -1;
  int32_t pos = 0;  // 1-based
  std::vector<std::string> alleles;
  std::vector<float> gps;  // [nsel][3] after parse_posteriors

  void init() {
    rd_.reset(new TsvLine(path));
    std::string line;
    bool have_hdr = false;
    while (rd_->getline(line)) {
      if (line.rfind("##contig=<", 0) == 0) {
        size_t p = line.find("ID=");
        if (p != std::string::npos) {
          size_t e = line.find_first_of(",>", p + 3);
          contig_id(line.substr(p + 3, e - p - 3));
        }
      } else if (line.rfind("#CHROM", 0) == 0) {
        std::vector<std::string> f = split_tab(line);
        if (wanted.empty()) {
          for (size_t i = 9; i < f.size(); ++i) {
            sm_cols_.push_back((int)(i - 9));
            sample_ids.push_back(f[i]);
          }
        } else {
          // bcf_filtered_reader.cpp:105-122: the requested IDs are walked as a std::set, i.e. in sorted order, and that
          // order -- not the VCF's column order -- numbers the samples (it decides ties in the best/next scans and which
          // sample is "sample 0", whose GP factor every singlet carries, cmd_cram_demuxlet.cpp:806); an ID the header
          // does not have is an error there (:110-111), not a warning.
          std::set<std::string> want(wanted.begin(), wanted.end());
          for (const std::string& id : want) {
            size_t col = 0;
            for (size_t i = 9; i < f.size() && !col; ++i)
              if (f[i] == id) col = i;
            if (!col) fatal("Cannot find sample ID %s from the BCF file", id.c_str());
            sm_cols_.push_back((int)(col - 9));
            sample_ids.push_back(id);
          }
        }
        n_vcf_samples_ = f.size() > 9 ? (int)f.size() - 9 : 0;
        have_hdr = true;
        break;
      } else if (line.rfind("##", 0) != 0) {
        fatal("%s: malformed VCF header line: %.60s", path.c_str(), line.c_str());
      }
    }
    if (!have_hdr) fatal("%s: no #CHROM header line", path.c_str());
    if (sample_ids.empty()) fatal("%s: no sample to compare with", path.c_str());
  }
  int nsamples() const { return (int)sample_ids.size(); }
  const char* chrom_name(int r) const { return (r >= 0 && r < (int)contigs_.size()) ? contigs_[(size_t)r].c_str() : "."; }

  // advance to the next record that passes the variant filter; false at end of file (bcf_filtered_reader read())
  bool read() {
    std::string line;
    while (rd_->getline(line)) {
      if (line.empty() || line[0] == '#') continue;
      split_reuse(line, '\t', fields_);  // (the strings of the previous record are overwritten in place: no allocation)
      if (fields_.size() < 8) fatal("%s: VCF record with %zu columns", path.c_str(), fields_.size());
      alleles.clear();
      alleles.push_back(fields_[3]);
      if (fields_[4] != ".") {
        size_t b = 0;
        const std::string& alt = fields_[4];
        while (true) {
          size_t e = alt.find(',', b);
          alleles.push_back(alt.substr(b, e == std::string::npos ? std::string::npos : e - b));
          if (e == std::string::npos) break;
          b = e + 1;
        }
      }
      rid = contig_id(fields_[0]);
      pos = atoi(fields_[1].c_str());
      if (fields_.size() > 8) {
        if (fields_[8] != fmt_str_) {  // (nearly every record repeats the previous FORMAT string)
          fmt_str_ = fields_[8];
          fmt_keys_ = split_char(fmt_str_, ':');
        }
      } else {
        fmt_str_.clear();
        fmt_keys_.clear();
      }
      if (!passed_vfilter()) continue;
      return true;
    }
    eof = true;
    return false;
  }

  // INFO float value (first element); false if absent
  bool info_float(const std::string& key, float* out) const {
    const std::string& info = fields_[7];
    size_t b = 0;
    while (b < info.size()) {
      size_t e = info.find(';', b);
      if (e == std::string::npos) e = info.size();
      if (info.compare(b, key.size(), key) == 0 && b + key.size() < e && info[b + key.size()] == '=') {
        *out = (float)strtod(info.c_str() + b + key.size() + 1, nullptr);
        return true;
      }
      b = e + 1;
    }
    return false;
  }

  // bcf_filtered_reader.cpp:367-461 with gt_error = 0 (the only value load_from_plp passes, sc_drop_seq.cpp:285)
  bool parse_posteriors(const std::string& field) {
    const int nalleles = (int)alleles.size();
    const int ngenos = (nalleles + 1) * nalleles / 2;
    const int ns = nsamples();
    gps.assign((size_t)ns * ngenos, 0.f);
    if (field == "GT") {
      if (!parse_genotypes()) return false;
      for (int i = 0; i < ns; ++i) {
        const int g = genotype_at(i);
        float* o = &gps[(size_t)i * ngenos];
        if (g < 0) {  // missing genotype: HWE from allele counts with pseudocounts (:385-393)
          int l = 0;
          for (int j = 0; j < nalleles; ++j)
            for (int k = 0; k <= j; ++k, ++l)
              o[l] = (float)((j == k ? 1.0 : 2.0) * (acs_[(size_t)j] + 1.0 / nalleles) / (an_ + 1.0) *
                             (acs_[(size_t)k] + 1.0 / nalleles) / (an_ + 1.0));
        } else {
          for (int j = 0; j < ngenos; ++j) o[j] = (g == j) ? 1.0f : 0.0f;  // gt_error == 0 (:405-407)
        }
      }
      return true;
    }
    const int fi = fmt_index(field);
    if (fi < 0) return false;
    if (field == "PL") return parse_likelihoods(fi, nalleles, ngenos);
    // GP (or any float FORMAT key) as posterior (:417-459), float arithmetic throughout
    std::vector<float> gpSums((size_t)ngenos, 0.f);
    for (int i = 0; i < nalleles; ++i)
      for (int j = 0; j <= i; ++j) gpSums[(size_t)((i + 1) * i / 2 + j)] = (float)(((i == j) ? 1.0 : 2.0) / (float)(nalleles * nalleles));
    for (int i = 0; i < ns; ++i) {
      std::vector<std::string> v = split_char(sample_field(i, fi), ',');
      float* o = &gps[(size_t)i * ngenos];
      float sumgp = 0;
      for (int j = 0; j < ngenos; ++j) {
        // a missing value is htslib's bcf_float_missing / vector_end, both NaN bit patterns
        o[j] = (j < (int)v.size() && v[(size_t)j] != ".") ? (float)strtod(v[(size_t)j].c_str(), nullptr) : NAN;
        sumgp += o[j];
      }
      for (int j = 0; j < ngenos; ++j) {
        o[j] /= sumgp;
        gpSums[(size_t)j] += o[j];
      }
    }
    for (int j = 0; j < ngenos; ++j) gpSums[(size_t)j] /= (int32_t)(ns + 1.0);
    // :452-457 with gt_error = 0: (1-0)*gp + 0*gpSums changes nothing -- unless gpSums is NaN (a sample of this record
    // had a missing value), which 0*NaN spreads to every sample, as in the reference
    const double gt_error = 0.0;
    for (int i = 0; i < ns; ++i)
      for (int j = 0; j < ngenos; ++j) {
        float& x = gps[(size_t)i * ngenos + j];
        x = (float)((1.0 - gt_error) * x + gt_error * gpSums[(size_t)j]);
      }
    return true;
  }

 private:
  // minimal line reader (VCF lines can be very long)
  struct TsvLine {
    gzFile fp;
    explicit TsvLine(const std::string& p) {
      fp = gzopen(p.c_str(), "rb");
      if (!fp) fatal("Cannot open %s for reading", p.c_str());
      gzbuffer(fp, 1 << 20);
    }
    ~TsvLine() {
      if (fp) gzclose(fp);
    }
    bool getline(std::string& line) {
      line.clear();
      char buf[65536];
      bool got = false;
      while (gzgets(fp, buf, sizeof(buf))) {
        got = true;
        size_t n = strlen(buf);
        line.append(buf, n);
        if (n && buf[n - 1] == '\n') break;
      }
      while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
      return got;
    }
  };
  static std::vector<std::string> split_char(const std::string& s, char c) {
    std::vector<std::string> out;
    size_t b = 0;
    while (true) {
      size_t e = s.find(c, b);
      out.push_back(s.substr(b, e == std::string::npos ? std::string::npos : e - b));
      if (e == std::string::npos) break;
      b = e + 1;
    }
    return out;
  }
  static std::vector<std::string> split_tab(const std::string& s) { return split_char(s, '\t'); }
  static void split_reuse(const std::string& s, char c, std::vector<std::string>& out) {
    size_t n = 0, b = 0;
    while (true) {
      const size_t e = s.find(c, b);
      if (n == out.size()) out.emplace_back();
      out[n++].assign(s, b, e == std::string::npos ? std::string::npos : e - b);
      if (e == std::string::npos) break;
      b = e + 1;
    }
    out.resize(n);
  }
  bool want_all() const { return wanted.empty(); }
  int contig_id(const std::string& name) {
    auto it = contig_ids_.find(name);
    if (it != contig_ids_.end()) return it->second;
    int id = (int)contigs_.size();
    contig_ids_[name] = id;
    contigs_.push_back(name);
    return id;
  }
  int fmt_index(const std::string& key) const {
    for (size_t i = 0; i < fmt_keys_.size(); ++i)
      if (fmt_keys_[i] == key) return (int)i;
    return -1;
  }
  std::string sample_field(int isel, int fi) const {
    const size_t col = 9 + (size_t)sm_cols_[(size_t)isel];
    if (col >= fields_.size()) return ".";
    const std::string& f = fields_[col];  // the fi-th of its ':'-separated values, without splitting all of them
    size_t b = 0;
    for (int k = 0; k < fi; ++k) {
      b = f.find(':', b);
      if (b == std::string::npos) return ".";
      ++b;
    }
    const size_t e = f.find(':', b);
    return f.substr(b, e == std::string::npos ? std::string::npos : e - b);
  }
  // bcf_filtered_reader.cpp:192-248: allele counts over the selected samples; gts_[2i], gts_[2i+1] = allele or -1
  bool parse_genotypes() {
    const int fi = fmt_index("GT");
    if (fi < 0) return false;
    const int ns = nsamples();
    gts_.assign((size_t)ns * 2, -1);
    acs_.assign(alleles.size(), 0.0);
    an_ = 0;
    for (int i = 0; i < ns; ++i) {
      const std::string g = sample_field(i, fi);
      size_t b = 0;
      for (int j = 0; j < 2 && b <= g.size(); ++j) {
        size_t e = g.find_first_of("/|", b);
        std::string a = g.substr(b, e == std::string::npos ? std::string::npos : e - b);
        int al = (a.empty() || a == ".") ? -1 : atoi(a.c_str());
        if (al >= (int)alleles.size()) al = -1;
        gts_[(size_t)i * 2 + j] = al;
        if (al >= 0) {
          ++an_;
          acs_[(size_t)al] += 1;
        }
        if (e == std::string::npos) break;
        b = e + 1;
      }
    }
    return true;
  }
  int genotype_at(int i) const {  // bcf_filtered_reader.h:151-156, bcf_alleles2gt
    const int a1 = gts_[(size_t)i * 2], a2 = gts_[(size_t)i * 2 + 1];
    if (a1 < 0 || a2 < 0) return -1;
    return a1 <= a2 ? a2 * (a2 + 1) / 2 + a1 : a1 * (a1 + 1) / 2 + a2;
  }
  // bcf_filtered_reader.cpp:505-581 restricted to the filters demuxlet sets (maxAlleles, minMAC, minCallRate)
  bool passed_vfilter() {
    if ((int)alleles.size() > vfilt.maxAlleles) return false;
    const bool require_GT = (vfilt.minMAC > 0) || (vfilt.minCallRate > 0);  // bcf_filter_arg.h:110-113
    if (!require_GT) return true;
    if (!parse_genotypes())
      fatal("Cannot find the field GT from the VCF file at position %s:%d", fields_[0].c_str(), pos);
    if (vfilt.minCallRate > (double)an_ / (2.0 * (double)nsamples())) return false;
    const int ac = an_ - (int)acs_[0];
    if ((ac < vfilt.minMAC) || (an_ - ac < vfilt.minMAC)) return false;
    return true;
  }
  // PL -> GP by 10 EM iterations on the allele frequencies (bcf_filtered_reader.cpp:250-327), diploid samples
  bool parse_likelihoods(int fi, int nalleles, int ngenos) {
    const int ns = nsamples();
    std::vector<int> pls((size_t)ns * ngenos, 0);
    for (int i = 0; i < ns; ++i) {
      std::vector<std::string> v = split_char(sample_field(i, fi), ',');
      for (int l = 0; l < ngenos; ++l) pls[(size_t)i * ngenos + l] = (l < (int)v.size() && v[(size_t)l] != ".") ? atoi(v[(size_t)l].c_str()) : 255;
    }
    std::vector<double> acs((size_t)nalleles, 1.0 / nalleles), gp((size_t)ngenos);
    const int niter = 10;
    int an = 0;
    for (int it = 0; it < niter; ++it) {
      std::vector<double> newacs((size_t)nalleles, 0.0);
      an = 0;
      for (int i = 0; i < ns; ++i) {
        double sumgp = 0;
        int l = 0;
        for (int j = 0; j < nalleles; ++j)
          for (int k = 0; k <= j; ++k, ++l) {
            const int pl = pls[(size_t)i * ngenos + l];
            sumgp += (gp[(size_t)l] = (j == k ? 1 : 2) * acs[(size_t)j] * acs[(size_t)k] * pow(0.1, (pl > 255 ? 255 : pl) * 0.1));
          }
        l = 0;
        for (int j = 0; j < nalleles; ++j)
          for (int k = 0; k <= j; ++k, ++l) {
            gp[(size_t)l] /= sumgp;
            newacs[(size_t)j] += gp[(size_t)l];
            newacs[(size_t)k] += gp[(size_t)l];
          }
        an += 2;
        if (it + 1 == niter)
          for (l = 0; l < ngenos; ++l) gps[(size_t)i * ngenos + l] = (float)gp[(size_t)l];
      }
      for (int a = 0; a < nalleles; ++a) acs[(size_t)a] = newacs[(size_t)a] / an;
    }
    return true;
  }

  std::unique_ptr<TsvLine> rd_;
  std::vector<std::string> contigs_;
  std::map<std::string, int> contig_ids_;
  std::vector<int> sm_cols_;
  int n_vcf_samples_ = 0;
  std::vector<std::string> fields_, fmt_keys_;
  std::string fmt_str_;  // the FORMAT string fmt_keys_ was split from
  std::vector<int> gts_;
  std::vector<double> acs_;
  int an_ = 0;
};

}  // namespace pa
