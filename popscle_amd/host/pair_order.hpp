// pair_order.hpp -- the reference's own order of a mirrored alpha = 0.5 doublet pair, decided on the host.
//
// At alpha == 0.5 the doublet likelihood is symmetric in the two samples.  The reference still evaluates both orders,
// llksAB[j][k][n] and llksAB[k][j][n], with transposed summation orders (cmd_cram_demuxlet.cpp:738-746: p =
// gps[j*3+l] * gps[k*3+m], sum over l then m), so the two log-likelihoods differ by rounding noise, and its strict-'<'
// scan (:883-906: j ascending, k ascending, n ascending) reports whichever order came out larger as DBL.BEST.GUESS and
// the other as the runner-up.  The device computes the pair once and mirrors the value (which is exact), so its scan
// always names (lo, hi) first.  To print what the reference prints, this pass recomputes the two log-likelihoods of
// exactly those pairs -- the best and/or next doublet of a cell when its alpha is 0.5 -- in the reference's
// association (IEEE doubles, no contraction, glibc log; per-read update, floor and normalisation of :655-725 over the
// whole alpha grid, because the division is by the maximum over all alphas) and orders the pair by the reference's
// rule: the second-scanned order (hi, lo) wins only if its log-likelihood is strictly larger.
//
// Host C++ of the product itself (it shares nothing with the CPU checker under tests/): used by popscle-amd demuxlet before it writes .best, and exported from
// libmuxgl as muxgl_demux_reference_pair_order for callers of the C-ABI.  Cost: one pass over the entries of the cells
// concerned, two nine-term sums and two logs per entry -- threaded over cells.
#ifndef POPSCLE_AMD_PAIR_ORDER_HPP
#define POPSCLE_AMD_PAIR_ORDER_HPP

#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "muxgl.h"

namespace pair_order {

struct Tables {
  double err[256], mat[256];
  Tables() {  // PhredHelper.cpp:24-41
    for (int i = 0; i < 256; ++i) {
      err[i] = (i > 1) ? pow(0.1, i * 0.1) : 0.75;
      mat[i] = 1. - err[i];
    }
  }
};

// log-likelihoods of (j, k, n) and (k, j, n) over one cell, cmd_cram_demuxlet.cpp:655-747 restricted to the two slots
inline void cell_pair_ll(const Tables& t, int64_t e0, int64_t e1, const int32_t* entry_snp, const int64_t* entry_rptr,
                         const uint8_t* reads, int32_t V, const double* gp, const uint8_t* has_gp, int32_t nAlpha,
                         const double* gridAlpha, int32_t j, int32_t k, int32_t n, double* ll_jk, double* ll_kj) {
#if defined(__clang__)
#pragma clang fp contract(off)  // this block only: every a*b+c below is two roundings, as in the reference's build
#elif defined(__FMA__)
#error "build pair_order.hpp without -mfma / -march=native: g++ would contract a*b+c and change the last bit"
#endif
  double pGs[MUXGL_MAX_ALPHA * 9];
  double ajk = 0, akj = 0;
  for (int64_t e = e0; e < e1; ++e) {
    for (int32_t i = 0; i < nAlpha * 9; ++i) pGs[i] = 1.0;
    for (int64_t r = entry_rptr[e]; r < entry_rptr[e + 1]; ++r) {  // :659-700
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
    const int32_t s = entry_snp[e];
    if (!has_gp[s]) continue;  // :733
    const double* g = gp + (size_t)s * V * 3;
    double sjk = 0, skj = 0;
    for (int32_t l = 0; l < 3; ++l)
      for (int32_t m = 0; m < 3; ++m) {  // :738-744
        const double pg = pGs[n * 9 + l * 3 + m];
        double p = g[j * 3 + l] * g[k * 3 + m];
        sjk += (p * pg);
        p = g[k * 3 + l] * g[j * 3 + m];
        skj += (p * pg);
      }
    ajk += log(sjk);  // :746
    akj += log(skj);
  }
  *ll_jk = ajk;
  *ll_kj = akj;
}

// Orders the alpha = 0.5 pairs of cells[0 .. C) as the reference's scan would.  stats (may be NULL): [0] cells looked
// at, [1] pairs reordered to (hi, lo), [2] pairs whose two orders came out exactly equal.
inline void reference_pair_order(int64_t C, int32_t V, const int64_t* cell_ptr, const int32_t* entry_snp,
                                 const int64_t* entry_rptr, const uint8_t* reads, const double* gp,
                                 const uint8_t* has_gp, int32_t nAlpha, const double* gridAlpha, muxgl_demux_cell* cells,
                                 int nthreads, int64_t* stats) {
  static const Tables tables;
  if (nthreads < 1) nthreads = 1;
  nthreads = (int)std::min<int64_t>(nthreads, std::max<int64_t>(1, C / 16));
  std::vector<int64_t> st((size_t)nthreads * 3, 0);
  const int64_t nnz = cell_ptr[C];
  auto work = [&](int tid) {
    // ranges of cells balanced by entries
    const int64_t lo_e = nnz * tid / nthreads, hi_e = nnz * (tid + 1) / nthreads;
    int64_t c0 = std::lower_bound(cell_ptr, cell_ptr + C, lo_e) - cell_ptr;
    int64_t c1 = (tid + 1 == nthreads) ? C : std::lower_bound(cell_ptr, cell_ptr + C, hi_e) - cell_ptr;
    int64_t* s = &st[(size_t)tid * 3];
    for (int64_t c = c0; c < c1; ++c) {
      muxgl_demux_cell& x = cells[c];
      if (!x.valid) continue;
      auto sym = [&](int32_t a, int32_t b, int32_t n) {
        return a >= 0 && b >= 0 && a != b && n >= 1 && n < nAlpha && gridAlpha[n] == 0.5;
      };
      const bool symB = sym(x.dBest1, x.dBest2, x.dBestA), symN = sym(x.dNext1, x.dNext2, x.dNextA);
      if (!symB && !symN) continue;
      ++s[0];
      const bool mirror = symB && symN && x.dBestA == x.dNextA &&
                          std::min(x.dBest1, x.dBest2) == std::min(x.dNext1, x.dNext2) &&
                          std::max(x.dBest1, x.dBest2) == std::max(x.dNext1, x.dNext2);
      auto order = [&](int32_t a, int32_t b, int32_t n, int32_t* first, int32_t* second) {
        const int32_t lo = std::min(a, b), hi = std::max(a, b);
        double l_lohi, l_hilo;
        cell_pair_ll(tables, cell_ptr[c], cell_ptr[c + 1], entry_snp, entry_rptr, reads, V, gp, has_gp, nAlpha,
                     gridAlpha, lo, hi, n, &l_lohi, &l_hilo);
        // scan order (:883-906): (j = lo, k = hi) is met first; (hi, lo) replaces it only if strictly larger
        if (l_hilo > l_lohi) { *first = hi; *second = lo; ++s[1]; }
        else { *first = lo; *second = hi; if (l_hilo == l_lohi) ++s[2]; }
      };
      if (symB) {
        int32_t f, g2;
        order(x.dBest1, x.dBest2, x.dBestA, &f, &g2);
        x.dBest1 = f; x.dBest2 = g2;
        if (mirror) { x.dNext1 = g2; x.dNext2 = f; }  // the other order is the runner-up
      }
      if (symN && !mirror) {
        int32_t f, g2;
        order(x.dNext1, x.dNext2, x.dNextA, &f, &g2);
        x.dNext1 = f; x.dNext2 = g2;
      }
      // the derived guesses, as :921-988 copy them
      if (x.type == MUXGL_DBL) {
        x.jBest = x.dBest1; x.kBest = x.dBest2;
        if (x.next_type == MUXGL_DBL) { x.jNext = x.dNext1; x.kNext = x.dNext2; }
      } else if (x.next_type == MUXGL_DBL) {
        x.jNext = x.dBest1; x.kNext = x.dBest2;
      }
    }
  };
  if (nthreads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto& t : th) t.join();
  }
  if (stats) {
    stats[0] = stats[1] = stats[2] = 0;
    for (int t = 0; t < nthreads; ++t)
      for (int i = 0; i < 3; ++i) stats[i] += st[(size_t)t * 3 + i];
  }
}

}  // namespace pair_order
#endif
