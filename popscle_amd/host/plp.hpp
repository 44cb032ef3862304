// plp.hpp -- CEL/VAR/PLP loader: the producer side of the libmuxgl boundary.
//
// Mirrors sc_dropseq_lib_t::load_from_plp (sc_drop_seq.cpp:103-384) but parses straight into the packed CSR pileup of
// include/muxgl.h instead of the reference's map<cell, map<snp, map<string UMI, uint32>>> model:
//   .cel.gz  header check (:143-151), --group-list / min-total / min-umi / min-snp filters (:164-180), dense ids in
//            file order, DROPLET_ID consistency check (:184-185)
//   .var.gz  header check (:210-217), SNP id = row index, chromosome index by first appearance (:228-233), merge-join
//            with the VCF cursor (:258-327) and GP row construction (:287-315)
//   .plp.gz  header check (:339-344), per base: bq = q-33, keep if bq >= minBQ, cap at capBQ, allele = digit (:350-370)
// Read order inside an entry is the reference's std::map<std::string> iteration order: every kept base gets the UMI
// sprintf("%x", numi++) (:361-368), i.e. reads are ordered by the hexadecimal STRING of a global kept-base counter.
#pragma once

#include <algorithm>
#include <memory>

#include "../../include/muxgl.h"
#include "plp_fast.hpp"
#include <exception>
#include <unordered_map>
#include <thread>

#include "vcf.hpp"

namespace pa {

struct SnpInfo {
  int32_t rid, pos;
  char ref, alt;
  double af;
};

struct Pileup {
  // cells
  std::vector<std::string> bcs;       // barcode of each cell id (file order of the kept cells)
  std::vector<int32_t> cell_totl_reads, cell_uniq_reads;
  // SNPs
  std::vector<SnpInfo> snps;
  std::vector<std::string> rid2chr;
  // packed pileup
  std::vector<int64_t> cell_ptr;
  BigVec<int64_t> entry_rptr;
  BigVec<int32_t> entry_snp;
  BigVec<uint8_t> reads;
  // demuxlet only
  int32_t nv = 0;
  std::vector<std::string> sample_ids;
  std::vector<double> gp;       // [S][nv][3]
  std::vector<uint8_t> has_gp;  // [S]
  // Slabs of one rank of a sharded run (LoadOptions::world > 1): the packed arrays above then hold the ROW slab -- the
  // cells [slab_c0, slab_c1) of the file with every marker, renumbered from 0 (bcs and the two read counts still span
  // all C cells; the counts are filled for the slab's cells) -- and col_* the COLUMN slab: every cell, the entries with
  // slab_s0 <= marker < slab_s1, marker ids unchanged (the arguments of muxgl_fmx_set_column_slab).
  bool slabbed = false;
  int64_t slab_c0 = 0, slab_c1 = 0, slab_s0 = 0, slab_s1 = 0;
  std::vector<int64_t> col_cell_ptr;
  BigVec<int64_t> col_entry_rptr;
  BigVec<int32_t> col_entry_snp;
  BigVec<uint8_t> col_reads;
  int64_t C() const { return (int64_t)bcs.size(); }
  int64_t S() const { return (int64_t)snps.size(); }
  int64_t nnz() const { return (int64_t)entry_snp.size(); }
};

struct LoadOptions {
  int32_t minRead = 0, minUMI = 0, minSNP = 0;  // --min-total --min-umi --min-snp
  int32_t minBQ = 13, capBQ = 20;               // cmd_cram_demuxlet.cpp:17-18, cmd_cram_freemux2.cpp:16-17
  std::string groupList;                        // --group-list
  // VCF side (demuxlet)
  std::string field = "GP";                     // --field
  double genoErrorOffset = 0.10, genoErrorCoeffR2 = 0.0;
  std::string r2info = "R2";
  // One rank of a sharded run: keep only this rank's two slabs of the pileup -- equal slices of ceil(n / world) cells
  // and markers (popscle_amd/shard.py: equal_ranges; the reference's own way to cut a job is by droplets at file level,
  // --group-list, README.md:168).  The .plp.gz rows of other cells AND other markers are tokenised and counted, not kept:
  // memory, ordering and packing are 2 / world of the whole.
  int32_t rank = 0, world = 1;
};

inline void load_from_plp(const std::string& prefix, const LoadOptions& opt, VcfReader* pvr, Pileup& out) {
  int nv = 0;
  if (pvr) {
    if (!pvr->read()) fatal("Cannot read any single variant from %s", pvr->path.c_str());
    if (!pvr->parse_posteriors(opt.field))
      fatal("Cannot parse posterior probability at %s:%d", pvr->chrom_name(pvr->rid), pvr->pos);
    nv = pvr->nsamples();
    out.nv = nv;
    out.sample_ids = pvr->sample_ids;
  }
  std::set<std::string> valid_bcs;
  if (!opt.groupList.empty()) {  // load_valid_barcodes, sc_drop_seq.cpp:93-101
    TsvReader t(opt.groupList);
    while (t.read_line() > 0) valid_bcs.insert(t.str_field_at(0));
    notice("Loaded %zu valid barcodes from %s", valid_bcs.size(), opt.groupList.c_str());
  }

  notice("Loading pileup information with prefix %s", prefix.c_str());
  StageTimer tm;
  // ---- .var.gz: read on a thread of its own while this one reads the .cel.gz (two small files, ~0.35 s each at 500 k
  // rows; they share nothing).  The reference reads the droplets first: an error there is the one reported, an error
  // in the markers is rethrown after the droplets went through.
  struct VarJoin {
    std::thread t;
    std::exception_ptr err;
    void join() {
      if (t.joinable()) t.join();
    }
    ~VarJoin() { join(); }
  } varj;
  varj.t = std::thread([&]() {
    try {
  {
    TsvReader t(prefix + ".var.gz");
    if (t.read_line() > 0) {
      if (t.nfields != 6 || strcmp("#SNP_ID", t.str_field_at(0)) || strcmp("CHROM", t.str_field_at(1)) ||
          strcmp("POS", t.str_field_at(2)) || strcmp("REF", t.str_field_at(3)) || strcmp("ALT", t.str_field_at(4)) ||
          strcmp("AF", t.str_field_at(5)))
        fatal("THe header line of %s.var.gz is malformed or outdated. Expecting #SNP_ID CHROM POS REF ALT AF",
              prefix.c_str());
    } else {
      fatal("Cannot read the first line of %s.var.gz", prefix.c_str());
    }
    std::map<std::string, int32_t> chr2rid;
    while (t.read_line() > 0) {
      if (t.nfields < 6) fatal("%s.var.gz: line %d has %d fields", prefix.c_str(), t.nlines, t.nfields);
      const char* chr = t.str_field_at(1);
      if (!chr2rid.count(chr)) {
        const int32_t newrid = (int32_t)chr2rid.size();
        chr2rid[chr] = newrid;
        out.rid2chr.push_back(chr);
      }
      SnpInfo s;
      s.rid = chr2rid[chr];
      s.pos = t.int_field_at(2);
      s.ref = t.str_field_at(3)[0];
      s.alt = t.str_field_at(4)[0];
      s.af = t.double_field_at(5);
      out.snps.push_back(s);
      if ((int)out.snps.size() + 1 != t.nlines)
        fatal("Expected SNP nID = %d but observed %zu", t.nlines - 1, out.snps.size() - 1);
    }
  }
    } catch (...) {
      varj.err = std::current_exception();
    }
  });
  // ---- .cel.gz
  std::vector<int32_t> index_bcs, tmp_totl, tmp_uniq, tmp_nsnp;
  {
    TsvReader t(prefix + ".cel.gz");
    if (t.read_line() > 0) {
      if (t.nfields != 6 || strcmp("#DROPLET_ID", t.str_field_at(0)) || strcmp("BARCODE", t.str_field_at(1)) ||
          strcmp("NUM.READ", t.str_field_at(2)) || strcmp("NUM.UMI", t.str_field_at(3)) ||
          strcmp("NUM.UMIwSNP", t.str_field_at(4)) || strcmp("NUM.SNP", t.str_field_at(5)))
        fatal("The header line of %s.cel.gz is malformed or outdated. Expecting #DROPLET_ID BARCODE NUM.READ NUM.UMI "
              "NUM.UMIwSNP NUM.SNP", prefix.c_str());
    } else {
      fatal("Cannot read the first line of %s.cel.gz", prefix.c_str());
    }
    int32_t nskip = 0;
    std::unordered_map<std::string, int32_t> bc_map;  // (add_cell's std::map is only ever searched and extended: sc_drop_seq.cpp:28-45)
    bc_map.reserve(1 << 16);
    while (t.read_line() > 0) {
      if (t.nfields < 6) fatal("%s.cel.gz: line %d has %d fields", prefix.c_str(), t.nlines, t.nfields);
      if (!valid_bcs.empty() && !valid_bcs.count(t.str_field_at(1))) {
        ++nskip;
        index_bcs.push_back(-1);
        continue;
      }
      const int32_t n_reads = t.int_field_at(2), n_umis = t.int_field_at(3), n_umi_w_snps = t.int_field_at(4),
                    n_snps = t.int_field_at(5);
      if (n_reads < opt.minRead || n_umis < opt.minUMI || n_snps < opt.minSNP) {
        index_bcs.push_back(-1);
        ++nskip;
        continue;
      }
      int32_t new_id;  // add_cell, sc_drop_seq.cpp:28-45
      auto it = bc_map.find(t.str_field_at(1));
      if (it == bc_map.end()) {
        new_id = (int32_t)out.bcs.size();
        bc_map[t.str_field_at(1)] = new_id;
        out.bcs.push_back(t.str_field_at(1));
      } else {
        new_id = it->second;
      }
      index_bcs.push_back(new_id);
      if (new_id + nskip != t.int_field_at(0))
        fatal("Observed DROPLET_ID %d is different from expected DROPLET_ID %d. Did you modify the digital pileup "
              "files by yourself?", t.int_field_at(0), new_id + nskip);
      tmp_totl.push_back(n_reads);
      tmp_uniq.push_back(n_umi_w_snps);
      tmp_nsnp.push_back(n_snps);
    }
    notice("Finished loading %zu droplets, skipping %d", out.bcs.size(), nskip);
  }
  const int64_t C = out.C();

  varj.join();
  if (varj.err) std::rethrow_exception(varj.err);
  notice("Finished loading %zu variants..", out.snps.size());  // (after the droplets' line, as the reference prints them)
  const int64_t S = out.S();
  // The VCF merge-join (sc_drop_seq.cpp:258-327) walks the markers in file order next to the VCF cursor and touches
  // nothing the .plp.gz stage reads or writes (a marker without genotypes stays in the pileup, gps == NULL): it runs on
  // a thread of its own while the pileup is inflated and parsed -- at the north_star's size the VCF's 200 k x 64
  // genotypes were 1.2 s of a 4.7 s run, all of it in front of the big file.
  // fatal() throws: an error inside the thread is parked and rethrown on the caller's thread after the join, and the
  // guard joins on every way out of this function (unwinding past a joinable std::thread is std::terminate).
  struct VcfJoin {
    std::thread t;
    std::exception_ptr err;
    void join() {
      if (t.joinable()) t.join();
    }
    void finish() {
      join();
      if (err) std::rethrow_exception(err);
    }
    ~VcfJoin() { join(); }
  } vj;
  if (opt.world > 1 && (opt.rank < 0 || opt.rank >= opt.world)) fatal("--rank must be in [0, --world)");
  if (pvr) {
    out.gp.reserve((size_t)S * nv * 3);
    out.has_gp.reserve((size_t)S);
    vj.t = std::thread([&out, &vj, pvr, &opt, nv, S]() {
     try {
      for (int64_t si = 0; si < S; ++si) {
        const SnpInfo& s = out.snps[(size_t)si];
        const char* chr = out.rid2chr[(size_t)s.rid].c_str();
        bool with_gp = false;
        {
          bool found = false, passed = false;  // sc_drop_seq.cpp:258-327
          while (!(found || passed)) {
            if (pvr->eof) {
              passed = true;
            } else if (pvr->rid > s.rid) {
              passed = true;
            } else if (pvr->rid == s.rid) {
              if (pvr->pos > s.pos) {
                passed = true;
              } else if (pvr->pos == s.pos) {
                const char vref = pvr->alleles[0].empty() ? '.' : pvr->alleles[0][0];
                const char valt = pvr->alleles.size() > 1 && !pvr->alleles[1].empty() ? pvr->alleles[1][0] : '.';
                if (vref != s.ref || valt != s.alt) passed = true;
                else found = true;
              }
            }
            if (passed) break;
            if (found) {
              if (!pvr->parse_posteriors(opt.field))
                fatal("Cannot parse posterior probability at %s:%d", pvr->chrom_name(pvr->rid), pvr->pos);
              std::vector<double> gps((size_t)nv * 3);
              double avgGPs[3] = {1e-10, 1e-10, 1e-10};
              for (int32_t i = 0; i < nv * 3; ++i) avgGPs[i % 3] += (gps[(size_t)i] = pvr->gps[(size_t)i]);
              const double sumGP = avgGPs[0] + avgGPs[1] + avgGPs[2];
              avgGPs[0] /= sumGP;
              avgGPs[1] /= sumGP;
              avgGPs[2] /= sumGP;
              double err = opt.genoErrorOffset;
              if (opt.genoErrorCoeffR2 > 0) {
                float r2 = 0;
                if (!pvr->info_float(opt.r2info, &r2))
                  fatal("Cannot extract %s (1 float value) from INFO field at %s:%d. Cannot use --geno-error-coeff",
                        opt.r2info.c_str(), chr, s.pos);
                err += (1 - opt.genoErrorOffset) * (1 - r2) * opt.genoErrorCoeffR2;
              }
              if (err > 0.999) err = 0.999;
              if (err < 0) err = 0;
              if (err > 0)
                for (int32_t i = 0; i < nv * 3; ++i) gps[(size_t)i] = (1 - err) * gps[(size_t)i] + err * avgGPs[i % 3];
              out.gp.insert(out.gp.end(), gps.begin(), gps.end());
              with_gp = true;
              break;
            }
            pvr->read();
          }
          if (!with_gp) out.gp.insert(out.gp.end(), (size_t)nv * 3, 0.0);
          out.has_gp.push_back(with_gp ? 1 : 0);
        }
      }
     } catch (...) {
      vj.err = std::current_exception();
     }
    });
  }

  tm.lap("cel+var");
  // the reference meets a VCF error while it reads the .var.gz, i.e. before it opens the .plp.gz: when both stages
  // fail, the VCF's error is the one reported
  auto plp_stage = [&]() {
  // ---- .plp.gz: the big file.  One thread inflates, the others parse line-aligned slices of each inflated block
  // (plp_fast.hpp); rows come back in file order with the global kept-base counter that names each read's UMI.
  PlpReadVec rds;
  out.cell_totl_reads.assign((size_t)C, 0);
  out.cell_uniq_reads.assign((size_t)C, 0);
  bool sorted = true;
  {
    PlpParseOptions po;
    po.minBQ = opt.minBQ;
    po.capBQ = opt.capBQ > 127 ? 127 : opt.capBQ;
    po.S = (int32_t)S;
    po.index_bcs = &index_bcs;
    if (opt.world > 1) {
      const int64_t pc = C > 0 ? (C + opt.world - 1) / opt.world : 0, ps = S > 0 ? (S + opt.world - 1) / opt.world : 0;
      out.slabbed = true;
      out.slab_c0 = std::min<int64_t>(opt.rank * pc, C);
      out.slab_c1 = std::min<int64_t>((opt.rank + 1) * pc, C);
      out.slab_s0 = std::min<int64_t>(opt.rank * ps, S);
      out.slab_s1 = std::min<int64_t>((opt.rank + 1) * ps, S);
      po.c0 = (int32_t)out.slab_c0, po.c1 = (int32_t)out.slab_c1, po.s0 = (int32_t)out.slab_s0, po.s1 = (int32_t)out.slab_s1;
    }
    const uint64_t numi = parse_plp_gz(prefix, po, rds, &sorted);
    tm.lap("plp inflate+parse");
    notice("Finished loading %llu UMIs in total..", (unsigned long long)numi);
  }
  // order: cell, SNP, then the reference's std::map<std::string> order of the "%x" UMI strings
  std::vector<int64_t> cell_rd0;  // first read of every cell in the ordered list
  if (out.slabbed) {  // the kept rows go to the row slab (cells renumbered), to the column slab, or to both
    PlpReadVec rows, cols;
    size_t nr = 0, nc = 0;
    for (const PlpRead& r : rds) {
      nr += r.cell >= out.slab_c0 && r.cell < out.slab_c1;
      nc += r.snp >= out.slab_s0 && r.snp < out.slab_s1;
    }
    rows.resize(nr);
    cols.resize(nc);
    nr = nc = 0;
    for (const PlpRead& r : rds) {
      if (r.snp >= out.slab_s0 && r.snp < out.slab_s1) cols[nc++] = r;
      if (r.cell >= out.slab_c0 && r.cell < out.slab_c1) {
        rows[nr] = r;
        rows[nr++].cell -= (int32_t)out.slab_c0;
      }
    }
    PlpReadVec().swap(rds);
    plp_order_by_cell(cols, C, sorted, cell_rd0);
    plp_pack(cols, C, cell_rd0, out.col_cell_ptr, out.col_entry_snp, out.col_entry_rptr, out.col_reads);
    PlpReadVec().swap(cols);
    const int64_t Cr = out.slab_c1 - out.slab_c0;
    plp_order_by_cell(rows, Cr, sorted, cell_rd0);
    tm.lap("plp order (two slabs)");
    plp_pack(rows, Cr, cell_rd0, out.cell_ptr, out.entry_snp, out.entry_rptr, out.reads);
    tm.lap("plp pack");
    for (int64_t c = 0; c < C; ++c) {  // read counts: the slab's cells only
      const bool mine = c >= out.slab_c0 && c < out.slab_c1;
      if (!mine) {
        out.cell_totl_reads[(size_t)c] = out.cell_uniq_reads[(size_t)c] = 0;
        continue;
      }
      const int64_t lc = c - out.slab_c0, nent = out.cell_ptr[(size_t)lc + 1] - out.cell_ptr[(size_t)lc];
      int32_t kept = 0;
      for (int64_t e = out.cell_ptr[(size_t)lc]; e < out.cell_ptr[(size_t)lc + 1]; ++e)
        kept += (int32_t)(out.entry_rptr[(size_t)e + 1] - out.entry_rptr[(size_t)e]);
      out.cell_totl_reads[(size_t)c] = out.cell_uniq_reads[(size_t)c] = kept;
      if (kept == tmp_uniq[(size_t)c] && tmp_nsnp[(size_t)c] == (int32_t)nent) out.cell_totl_reads[(size_t)c] = tmp_totl[(size_t)c];
    }
    vj.finish();
    return;
  }
  plp_order_by_cell(rds, C, sorted, cell_rd0);
  for (int64_t c = 0; c < C; ++c)  // kept bases per cell; every kept base is its own UMI (:361-368)
    out.cell_totl_reads[(size_t)c] = out.cell_uniq_reads[(size_t)c] = (int32_t)(cell_rd0[(size_t)c + 1] - cell_rd0[(size_t)c]);
  tm.lap("plp order");
  plp_pack(rds, C, cell_rd0, out.cell_ptr, out.entry_snp, out.entry_rptr, out.reads);
  tm.lap("plp pack");
  if (vj.t.joinable()) {
    vj.finish();
    tm.lap("vcf merge-join (waited for)");
  }
  // sanity check on the observed counts (:375-380)
  for (int64_t c = 0; c < C; ++c) {
    const int64_t nent = out.cell_ptr[(size_t)c + 1] - out.cell_ptr[(size_t)c];
    if (out.cell_uniq_reads[(size_t)c] == tmp_uniq[(size_t)c] && tmp_nsnp[(size_t)c] == (int32_t)nent)
      out.cell_totl_reads[(size_t)c] = tmp_totl[(size_t)c];
  }
  };
  try {
    plp_stage();
  } catch (...) {
    vj.join();
    if (vj.err) std::rethrow_exception(vj.err);
    throw;
  }
}

}  // namespace pa
