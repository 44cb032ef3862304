// exact_calls.hpp -- demuxlet calls that rounding noise could decide, decided in the reference's own arithmetic (host).
//
// The kernels form a cell's log-likelihoods in another association than the reference (products with exponent
// bookkeeping and one log per hypothesis instead of a sum of logs): equal to ~1e-12, not to the last bit.  Every decision
// of cmd_cram_demuxlet.cpp:827-837 (best/next singlet), :883-906 (best/next doublet) and :925-988 (the +2 thresholds
// between them) is a comparison of two such numbers, so a comparison whose margin is within EPS = 1e-9 x max(1, |LL|)
// is not the kernels' to make.  This pass finds those cells from the records (next to best and next the call kernel leaves
// two bits in `valid`: the third-largest value of a scan is in reach of the runner-up), recomputes the contested hypotheses exactly as the reference does --
// IEEE doubles in the reference's operation order with nothing contracted, glibc's log; per-read update, floor and
// normalisation of :655-725 over the whole alpha grid -- and makes the scans and the call on those numbers:
//
//   * an alpha = 0.5 pair is symmetric in its two samples; the reference evaluates (j,k) and (k,j) with transposed
//     summation orders (:738-746), so its scan reports whichever order came out larger in the last bits.  The kernels
//     evaluate the pair once and name it (lo, hi) with its mirror as runner-up: EVERY cell whose best (or next) doublet is
//     such a pair has this one tie to settle (two hypotheses per cell: the bulk of the pass's work);
//   * best and next within EPS, or a threshold margin within EPS: the named hypotheses are recomputed and compared exactly;
//   * next and third within EPS (three or more hypotheses in reach of each other, e.g. a droplet with two entries against
//     samples that share their genotypes there): ALL hypotheses of that scan are recomputed for the cell.
//
// The result is the reference's record for the cell: its integer fields exactly, the log-likelihoods of the recomputed
// hypotheses exactly, the evidence sums as the kernels formed them (1e-16 relative; they enter no decision).
//
// Host C++ of the product (it shares nothing with the CPU checker the tests use): used by popscle-amd demuxlet
// before it writes .best, exported from libmuxgl as muxgl_demux_exact_calls.  Threaded over cells.
#ifndef POPSCLE_AMD_EXACT_CALLS_HPP
#define POPSCLE_AMD_EXACT_CALLS_HPP

#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "muxgl.h"

namespace exact_calls {

enum { ST_CELLS = 0, ST_MIRROR_TURNED = 1, ST_MIRROR_EXACT_TIES = 2, ST_NEAR_TIES = 3, ST_DEEP = 4, ST_CHANGED = 5, ST_N = 6 };

struct Tables {
  double err[256], mat[256];
  Tables() {  // PhredHelper.cpp:24-41
    for (int i = 0; i < 256; ++i) {
      err[i] = (i > 1) ? pow(0.1, i * 0.1) : 0.75;
      mat[i] = 1. - err[i];
    }
  }
};

struct Hyp {
  int32_t j, k, n;  // llksAB[j][k][n]; a singlet is (j, 0, 0) (:799,806)
  double ll;
};

// pGs[nAlpha * 9] of one entry from its reads [r0, r1): per-read update with division by the running maximum, floor,
// division by the maximum (cmd_cram_demuxlet.cpp:655-725)
inline void entry_pgs(const Tables& t, const uint8_t* reads, int64_t r0, int64_t r1, int32_t nAlpha, const double* gridAlpha,
                      double* pGs) {
#if defined(__clang__)
#pragma clang fp contract(off)  // this block only: every a*b+c below is two roundings, as in the reference's build
#elif defined(__FMA__)
#error "build exact_calls.hpp without -mfma / -march=native: g++ would contract a*b+c and change the last bit"
#endif
  for (int32_t i = 0; i < nAlpha * 9; ++i) pGs[i] = 1.0;
  for (int64_t r = r0; r < r1; ++r) {  // :659-700
    const uint8_t b = reads[r];
    if (b == MUXGL_READ_OTHER) continue;  // al == 2
    const int al = b >> 7, bq = b & 0x7f;
    const double pR = (al == 0) ? t.mat[bq] : t.err[bq] / 3.0;
    const double pA = (al == 1) ? t.mat[bq] : t.err[bq] / 3.0;
    double maxpG = 0;
    for (int32_t a = 0; a < nAlpha; ++a)
      for (int32_t l = 0; l < 3; ++l)
        for (int32_t m = 0; m < 3; ++m) {
          const double p = 0.5 * l + (m - l) * 0.5 * gridAlpha[a];
          double& pG = pGs[a * 9 + l * 3 + m];
          pG *= (pR * (1.0 - p) + pA * p);
          if (maxpG < pG) maxpG = pG;
        }
    for (int32_t i = 0; i < nAlpha * 9; ++i) pGs[i] /= maxpG;
  }
  double maxpG = 0;  // :703-725
  for (int32_t i = 0; i < nAlpha * 9; ++i) {
    pGs[i] += 1e-10;
    if (maxpG < pGs[i]) maxpG = pGs[i];
  }
  for (int32_t i = 0; i < nAlpha * 9; ++i) pGs[i] /= maxpG;
}

// The pGs of an entry with at most one usable read depend on that read's byte alone (three quarters of the entries of a
// droplet data set): the same operations on the same numbers, done once per byte instead of once per entry.
// Row 255 (MUXGL_READ_OTHER is never a usable read's byte): no usable read.
struct OneReadTable {
  int32_t nAlpha = 0;
  std::vector<double> rows;  // [256][nAlpha * 9]
  OneReadTable(const Tables& t, int32_t nA, const double* gridAlpha) : nAlpha(nA), rows((size_t)256 * nA * 9) {
    for (int b = 0; b < 256; ++b) {
      const uint8_t byte = (uint8_t)b;
      entry_pgs(t, &byte, 0, b == MUXGL_READ_OTHER ? 0 : 1, nA, gridAlpha, &rows[(size_t)b * nA * 9]);
    }
  }
  const double* row(uint8_t b) const { return &rows[(size_t)b * nAlpha * 9]; }
};

// log-likelihoods of the listed hypotheses over one cell, cmd_cram_demuxlet.cpp:655-747 restricted to those slots
inline void cell_lls(const Tables& t, const OneReadTable& one, int64_t e0, int64_t e1, const int32_t* entry_snp,
                     const int64_t* entry_rptr, const uint8_t* reads, int32_t V, const double* gp, const uint8_t* has_gp,
                     int32_t nAlpha, const double* gridAlpha, Hyp* hyps, size_t nh) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  double buf[MUXGL_MAX_ALPHA * 9];
  for (size_t h = 0; h < nh; ++h) hyps[h].ll = 0;
  for (int64_t e = e0; e < e1; ++e) {
    if (e + 8 < e1) {  // the genotype rows of an entry are a random gather from a tensor of tens of MB: ask for them early
      const double* g8 = gp + (size_t)entry_snp[e + 8] * V * 3;
      for (size_t h = 0; h < (nh < 4 ? nh : 4); ++h) {
        __builtin_prefetch(g8 + hyps[h].j * 3);
        __builtin_prefetch(g8 + hyps[h].k * 3);
      }
    }
    const int32_t s = entry_snp[e];
    if (!has_gp[s]) continue;  // :733 (the entry's pGs are formed by the reference, and read by nobody)
    int usable = 0;
    uint8_t byte = MUXGL_READ_OTHER;
    for (int64_t r = entry_rptr[e]; r < entry_rptr[e + 1] && usable < 2; ++r)
      if (reads[r] != MUXGL_READ_OTHER) {
        byte = reads[r];
        ++usable;
      }
    const double* pGs;
    if (usable <= 1) {
      pGs = one.row(byte);
    } else {
      entry_pgs(t, reads, entry_rptr[e], entry_rptr[e + 1], nAlpha, gridAlpha, buf);
      pGs = buf;
    }
    const double* g = gp + (size_t)s * V * 3;
    for (size_t h = 0; h < nh; ++h) {
      const double* gj = g + hyps[h].j * 3;
      const double* gk = g + hyps[h].k * 3;
      const double* pg = pGs + hyps[h].n * 9;
      double sum = 0;
      for (int32_t l = 0; l < 3; ++l)
        for (int32_t m = 0; m < 3; ++m) {  // :738-744
          const double p = gj[l] * gk[m];
          sum += (p * pg[l * 3 + m]);
        }
      hyps[h].ll += log(sum);  // :746
    }
  }
}

struct Top2 {
  double bv = -1e300, nv = -1e300;
  int32_t b = -1, n = -1;  // index into the candidate list
  void push(double v, int32_t i) {  // the reference's update rule (:827-837, :884-905)
    if (bv < v) {
      nv = bv; n = b;
      bv = v; b = i;
    } else if (nv < v) {
      nv = v; n = i;
    }
  }
};

// Settles the cells of cells[0 .. C) whose calls are within rounding reach.  stats: NULL or int64[ST_N].
inline void exact_calls(int64_t C, int32_t V, const int64_t* cell_ptr, const int32_t* entry_snp, const int64_t* entry_rptr,
                        const uint8_t* reads, const double* gp, const uint8_t* has_gp, int32_t nAlpha,
                        const double* gridAlpha, double doublet_prior, muxgl_demux_cell* cells, int nthreads,
                        int64_t* stats) {
  static const Tables tables;
  const OneReadTable one(tables, nAlpha, gridAlpha);
  if (nthreads < 1) nthreads = 1;
  nthreads = (int)std::min<int64_t>(nthreads, std::max<int64_t>(1, C / 16));
  std::vector<int64_t> st((size_t)nthreads * ST_N, 0);
  const int64_t nnz = cell_ptr[C];
  // :793-795
  const double log_single_prior = log((1.0 - doublet_prior) / V);
  const double log_doublet_prior1 = log(doublet_prior / V / (V - 1.) / (nAlpha - 1.));
  const double log_doublet_prior2 = log(doublet_prior / V / (V - 1.) / (nAlpha - 1.) * 2);
  auto work = [&](int tid) {
    // ranges of cells balanced by entries
    const int64_t lo_e = nnz * tid / nthreads, hi_e = nnz * (tid + 1) / nthreads;
    int64_t c0 = std::lower_bound(cell_ptr, cell_ptr + C, lo_e) - cell_ptr;
    int64_t c1 = (tid + 1 == nthreads) ? C : std::lower_bound(cell_ptr, cell_ptr + C, hi_e) - cell_ptr;
    int64_t* s = &st[(size_t)tid * ST_N];
    std::vector<Hyp> hs, hd;
    for (int64_t c = c0; c < c1; ++c) {
      muxgl_demux_cell& x = cells[c];
      if (!(x.valid & 1)) continue;
      double mag = 1.0;
      for (double v : {x.sngBestLLK, x.sngNextLLK, x.dblBestLLK, x.dblNextLLK})
        if (v > -1e299) mag = std::max(mag, fabs(v));
      const double eps = 1e-9 * mag;
      auto some = [](double v) { return v > -1e299; };
      auto near = [&](double a, double b) { return some(a) && some(b) && fabs(a - b) <= eps; };
      auto sym = [&](int32_t n) { return n >= 1 && n < nAlpha && gridAlpha[n] == 0.5; };
      const bool mirror = x.dBest1 >= 0 && sym(x.dBestA) && x.dNext1 == x.dBest2 && x.dNext2 == x.dBest1 && x.dNextA == x.dBestA;
      const bool next_sym = !mirror && x.dNext1 >= 0 && sym(x.dNextA);
      const bool s2 = near(x.sngBestLLK, x.sngNextLLK), s3 = (x.valid & MUXGL_CELL_DEEP_SNG) != 0;
      // (set by the call kernel from the third-largest value of the scan; for a mirrored best: from the best hypothesis that
      //  is neither order of it, demux_call_body.hpp)
      const bool d2 = !mirror && near(x.dblBestLLK, x.dblNextLLK);
      const bool d3 = (x.valid & MUXGL_CELL_DEEP_DBL) != 0;
      const bool th = near(x.dblBestLLK, x.sngBestLLK + 2) || near(x.dblNextLLK, x.sngBestLLK + 2) ||
                      near(x.sngBestLLK, x.sngNextLLK + 2) || near(x.dblBestLLK, x.sngNextLLK + 2);
      const bool do_s = s2 || s3 || th, do_d = mirror || next_sym || d2 || d3 || th;
      if (!do_s && !do_d) continue;
      ++s[ST_CELLS];
      if (s2 || s3 || d2 || d3 || th) ++s[ST_NEAR_TIES];
      if (s3 || d3) ++s[ST_DEEP];
      x.valid = 1;
      const muxgl_demux_cell before = x;

      if (do_s) {  // singlet scan (:827-837) over the contenders, ascending sample = the reference's scan order
        hs.clear();
        if (s3) {
          for (int32_t j = 0; j < V; ++j) hs.push_back(Hyp{j, 0, 0, 0.0});
        } else {
          if (x.sBest >= 0) hs.push_back(Hyp{x.sBest, 0, 0, 0.0});
          if (x.sNext >= 0) hs.push_back(Hyp{x.sNext, 0, 0, 0.0});
          std::sort(hs.begin(), hs.end(), [](const Hyp& a, const Hyp& b) { return a.j < b.j; });
        }
        cell_lls(tables, one, cell_ptr[c], cell_ptr[c + 1], entry_snp, entry_rptr, reads, V, gp, has_gp, nAlpha, gridAlpha,
                 hs.data(), hs.size());
        Top2 t;
        for (size_t i = 0; i < hs.size(); ++i) t.push(hs[i].ll, (int32_t)i);
        x.sBest = t.b >= 0 ? hs[(size_t)t.b].j : -1;
        x.sNext = t.n >= 0 ? hs[(size_t)t.n].j : -1;
        x.sngBestLLK = t.bv;
        x.sngNextLLK = t.nv;
      }
      if (do_d && x.dBest1 >= 0) {  // doublet scan (:883-906): j, then k, then n ascending
        hd.clear();
        if (d3) {
          for (int32_t j = 0; j < V; ++j)
            for (int32_t k = 0; k < V; ++k)
              if (k != j)
                for (int32_t n = 1; n < nAlpha; ++n) hd.push_back(Hyp{j, k, n, 0.0});
        } else {
          hd.push_back(Hyp{x.dBest1, x.dBest2, x.dBestA, 0.0});
          if (mirror) {
            hd.push_back(Hyp{x.dBest2, x.dBest1, x.dBestA, 0.0});
          } else if (x.dNext1 >= 0) {
            hd.push_back(Hyp{x.dNext1, x.dNext2, x.dNextA, 0.0});
            if (next_sym) hd.push_back(Hyp{x.dNext2, x.dNext1, x.dNextA, 0.0});
          }
          std::sort(hd.begin(), hd.end(), [&](const Hyp& a, const Hyp& b) {
            return ((int64_t)a.j * V + a.k) * nAlpha + a.n < ((int64_t)b.j * V + b.k) * nAlpha + b.n;
          });
        }
        cell_lls(tables, one, cell_ptr[c], cell_ptr[c + 1], entry_snp, entry_rptr, reads, V, gp, has_gp, nAlpha, gridAlpha,
                 hd.data(), hd.size());
        Top2 t;
        for (size_t i = 0; i < hd.size(); ++i) t.push(hd[i].ll, (int32_t)i);
        const Hyp& b = hd[(size_t)t.b];
        if (mirror && !d3) {
          if (b.j > b.k) ++s[ST_MIRROR_TURNED];
          if (t.bv == t.nv) ++s[ST_MIRROR_EXACT_TIES];
        }
        x.dBest1 = b.j; x.dBest2 = b.k; x.dBestA = b.n;
        x.dblBestLLK = t.bv;
        if (t.n >= 0) {
          const Hyp& n2 = hd[(size_t)t.n];
          x.dNext1 = n2.j; x.dNext2 = n2.k; x.dNextA = n2.n;
          x.dblNextLLK = t.nv;
        }
      }
      // the call on those numbers, :921-991 (sumLLK / sngLLK as the kernels summed them)
      if (x.dblBestLLK > x.sngBestLLK + 2) {
        x.type = MUXGL_DBL;
        x.bestPP = exp(x.dblBestLLK + ((gridAlpha[x.dBestA] == 0.5) ? log_doublet_prior2 : log_doublet_prior1) - x.sumLLK);
        x.jBest = x.dBest1; x.kBest = x.dBest2; x.aBest = x.dBestA;
        x.bestLLK = x.dblBestLLK;
        if (x.dblNextLLK > x.sngBestLLK + 2) {
          x.next_type = MUXGL_DBL;
          x.jNext = x.dNext1; x.kNext = x.dNext2; x.aNext = x.dNextA;
          x.nextLLK = x.dblNextLLK;
        } else {
          x.next_type = MUXGL_SNG;
          x.jNext = x.kNext = x.sBest; x.aNext = 0;
          x.nextLLK = x.sngBestLLK;
        }
      } else {
        x.type = (x.sngBestLLK > x.sngNextLLK + 2) ? MUXGL_SNG : MUXGL_AMB;
        x.bestPP = x.sngBestLLK + log_single_prior - x.sumLLK;
        x.jBest = x.kBest = x.sBest; x.aBest = 0;
        x.bestLLK = x.sngBestLLK;
        if (x.dblBestLLK > x.sngNextLLK + 2) {
          x.next_type = MUXGL_DBL;
          x.jNext = x.dBest1; x.kNext = x.dBest2; x.aNext = x.dBestA;
          x.nextLLK = x.dblBestLLK;
        } else {
          x.next_type = MUXGL_SNG;
          x.jNext = x.kNext = x.sNext; x.aNext = 0;
          x.nextLLK = x.sngNextLLK;
        }
      }
      x.sngOnlyPP = exp(x.sngBestLLK + log_single_prior - x.sngLLK);
      // what changed beyond the order of a mirrored pair
      auto unordered = [](int32_t a, int32_t b2) { return std::make_pair(std::min(a, b2), std::max(a, b2)); };
      if (x.type != before.type || x.next_type != before.next_type || x.sBest != before.sBest || x.sNext != before.sNext ||
          unordered(x.dBest1, x.dBest2) != unordered(before.dBest1, before.dBest2) || x.dBestA != before.dBestA ||
          unordered(x.dNext1, x.dNext2) != unordered(before.dNext1, before.dNext2) || x.dNextA != before.dNextA)
        ++s[ST_CHANGED];
    }
  };
  if (nthreads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto& t : th) t.join();
  }
  if (stats) {
    for (int i = 0; i < ST_N; ++i) stats[i] = 0;
    for (int t = 0; t < nthreads; ++t)
      for (int i = 0; i < ST_N; ++i) stats[i] += st[(size_t)t * ST_N + i];
  }
}

}  // namespace exact_calls
#endif
