// fmx_exact.hip -- freemuxlet calls that rounding noise could decide, decided in the reference's own arithmetic.
//
// The E-step kernels form a cell's K (K + 1) / 2 log-likelihoods from cluster posteriors that went through reciprocal
// multiplies and in another association than the reference (products with exponent bookkeeping, one log per hypothesis):
// equal to ~1e-12, not to the last bit.  Every decision of an EM iteration -- best / next singlet and doublet
// (cmd_cram_freemux2.cpp:469-497), the four `> x + 2` thresholds of the re-assignment (:521-584) -- compares two such
// numbers, and the re-assignment feeds the ordered M-step, i.e. the next iteration.  fmx_call_kernel therefore lists the
// cells where a comparison's margin is within 1e-9 x max(1, |LL|) (muxgl_fmx_cell carries the third-largest value of each
// scan for that), and this file settles them before the iteration's results leave the library:
//
//   gp_kernel   thread = (entry of a listed cell, cluster): the state of (cluster, SNP) as the reference's maps hold it
//               when the E-step reads it -- merge() (sc_drop_seq.h:77-101) of the entries of the cells assigned to the
//               cluster BEFORE this iteration that cover the SNP, in ascending cell id (:277-288, :590-596), every entry's
//               pileup recomputed from its read bytes (sc_drop_seq.cpp:452-509); then the cluster's genotype posterior
//               gp1s of :402-415.  IEEE operations in the reference's order, nothing contracted (exact_arith.hpp).
//   lk_kernel   thread = (entry, contested hypothesis): lk of :440-452 from the entry's own pileup and two posteriors.
//   host        llks += log(lk) over the cell's entries in ascending SNP (:454-455, glibc's log), the scans over the
//               contested hypotheses in the reference's scan order, the re-assignment, the nchanged rules; the records,
//               assignments and counters on the device are patched, and when an assignment changed the ordered M-step of
//               the iteration is run again from the corrected assignments.
//
// "Contested" = the named best and next of a scan, or every hypothesis of the scan when next and third are within reach
// too (e.g. clusters without cells: identical posteriors, exact ties in the reference, which keeps the first in scan
// order).  By induction over the iterations the assignments are the reference's, so the chains are over the reference's
// member sets.  One device, whole pileup (muxgl_fmx_iterate): a rank of a sharded run holds neither the other ranks'
// entries nor their assignments of the previous iteration in SNP-major form; there the listed cells are counted
// (muxgl_fmx_exact_stats) and left as the kernels decided them.
#include <math.h>

#include <algorithm>
#include <vector>

#include "common.hpp"
#include "exact_arith.hpp"

namespace {

using exact_arith::entry_pileup;
using exact_arith::merge;

// gp[(t * K + c) * 3 ..] = gp1s of cluster c at the SNP of entry ent[t] (:402-415); glis[t * 9 ..] = the entry's pileup
__global__ void __launch_bounds__(256)
    gp_kernel(int64_t nT, int K, const int64_t* __restrict__ ent, const int32_t* __restrict__ entry_snp,
              const int64_t* __restrict__ entry_rptr, const uint8_t* __restrict__ reads, const double* __restrict__ lut,
              const double* __restrict__ af, const int64_t* __restrict__ snp_ptr, const int64_t* __restrict__ snp_entry,
              const int32_t* __restrict__ snp_cell, const int32_t* __restrict__ prev_clust, double geno_error,
              double* __restrict__ gp, double* __restrict__ glis) {
#pragma clang fp contract(off)
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= nT * K) return;
  const int64_t t = tid / K;
  const int c = (int)(tid % K);
  const int64_t e = ent[t];
  const int32_t snp = entry_snp[e];
  double g[9];
  for (int i = 0; i < 9; ++i) g[i] = 1.0;  // snp_droplet_pileup() (sc_drop_seq.h:72-75): what operator[] default-constructs
  for (int64_t p = snp_ptr[snp], p1 = snp_ptr[snp + 1]; p < p1; ++p) {  // ascending cell id
    if (prev_clust[snp_cell[p]] != c) continue;
    const int64_t pe = snp_entry[p];
    double o[9];
    entry_pileup(reads, entry_rptr[pe], entry_rptr[pe + 1], lut, o);
    merge(g, o);
  }
  const double a = af[snp];
  double gp0s[3], gp1s[3];
  gp0s[0] = (1.0 - a) * (1.0 - a);  // :388-390
  gp0s[1] = 2 * a * (1.0 - a);
  gp0s[2] = a * a;
  gp1s[0] = (1.0 - a) * (1.0 - a) * g[0];  // :402-404
  gp1s[1] = 2 * a * (1.0 - a) * g[4];
  gp1s[2] = a * a * g[8];
  const double sum1 = gp1s[0] + gp1s[1] + gp1s[2];
  gp1s[0] /= sum1;
  gp1s[1] /= sum1;
  gp1s[2] /= sum1;
  if (geno_error > 0) {  // :410-415
    gp1s[0] = (1 - geno_error) * gp1s[0] + geno_error * gp0s[0];
    gp1s[1] = (1 - geno_error) * gp1s[1] + geno_error * gp0s[1];
    gp1s[2] = (1 - geno_error) * gp1s[2] + geno_error * gp0s[2];
  }
  double* o = gp + (size_t)tid * 3;
  o[0] = gp1s[0], o[1] = gp1s[1], o[2] = gp1s[2];
  if (c == 0) entry_pileup(reads, entry_rptr[e], entry_rptr[e + 1], lut, glis + (size_t)t * 9);
}

struct item {
  int64_t t;     // entry position in the batch
  int32_t j, k;  // hypothesis: k < j a pair, k == j the singlet of j
};

__global__ void __launch_bounds__(256)
    lk_kernel(int64_t n, int K, const item* __restrict__ items, const double* __restrict__ gp, const double* __restrict__ glis,
              double* __restrict__ lk_out) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const item it = items[i];
  const double* gl = glis + (size_t)it.t * 9;
  const double* gp1s = gp + ((size_t)it.t * K + it.j) * 3;
  double lk = 0;
  if (it.k == it.j) {  // :448-451
    for (int g1 = 0; g1 < 3; ++g1) lk += (gl[g1 * 3 + g1] * gp1s[g1]);
  } else {  // :440-445
    const double* gp2s = gp + ((size_t)it.t * K + it.k) * 3;
    for (int g1 = 0; g1 < 3; ++g1)
      for (int g2 = 0; g2 < 3; ++g2) lk += (gl[g1 * 3 + g2] * gp1s[g1] * gp2s[g2]);
  }
  lk_out[i] = lk;
}

__global__ void patch_kernel(int n, const int32_t* __restrict__ idx, const muxgl_fmx_cell* __restrict__ rec,
                             muxgl_fmx_cell* __restrict__ cells, int32_t* __restrict__ clust) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  cells[idx[i]] = rec[i];
  clust[idx[i]] = rec[i].clust;
}

struct hyp {
  int32_t j, k;
  double ll;
};

struct top2 {  // the reference's update rule (:469-497): strict >, first come first kept
  double bv = -1e300, nv = -1e300;
  int32_t b = -1, n = -1;
  void push(double v, int32_t i) {
    if (v > bv) {
      nv = bv, n = b;
      bv = v, b = i;
    } else if (v > nv) {
      nv = v, n = i;
    }
  }
};

template <class T>
int upload(muxgl_handle* h, T** d, const std::vector<T>& v) {
  if (dev_alloc(h, d, v.size() ? v.size() : 1)) return 1;
  if (!v.empty()) HIPCHK(h, hipMemcpyAsync(*d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, h->stream));
  return 0;
}

}  // namespace

// Settles the nflag cells fmx_call_kernel listed in h->d_flagged (the stream is drained, h->h_fstat holds the counters).
// *reassigned = an assignment (clust) changed: the caller runs the M-step again.
int fmx_exact_resolve(muxgl_handle* h, const muxgl_fmx_params* p, int32_t nflag, bool* reassigned) {
  *reassigned = false;
  const int K = h->K;
  const int64_t C = h->C;
  std::vector<int32_t> cells((size_t)nflag);
  HIPCHK(h, hipMemcpy(cells.data(), h->d_flagged, sizeof(int32_t) * (size_t)nflag, hipMemcpyDeviceToHost));
  std::sort(cells.begin(), cells.end());  // (atomic order -> cell order: the result must not depend on it)
  // the listed cells' records, previous states and entry ranges
  std::vector<muxgl_fmx_cell> all((size_t)C);
  std::vector<int32_t> prev((size_t)C);
  std::vector<int64_t> cptr((size_t)C + 1);
  HIPCHK(h, hipMemcpy(all.data(), h->d_fcells, sizeof(muxgl_fmx_cell) * (size_t)C, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(prev.data(), h->d_prev_state, sizeof(int32_t) * (size_t)C, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(cptr.data(), h->d_cell_ptr, sizeof(int64_t) * ((size_t)C + 1), hipMemcpyDeviceToHost));
  const double log_single_prior = log((1.0 - p->doublet_prior) / K);          // :379
  const double log_double_prior = log(p->doublet_prior / K / (K - 1) * 2.0);  // :380

  std::vector<int32_t> pidx;
  std::vector<muxgl_fmx_cell> prec;
  int64_t d_single = 0, d_amb = 0, d_changed = 0;
  // batches of listed cells: at most ~2^27 doubles of posteriors on the device at a time
  const int64_t cap_t = std::max<int64_t>(1, (int64_t(1) << 27) / (3 * (int64_t)K));
  size_t f0 = 0;
  while (f0 < cells.size()) {
    size_t f1 = f0;
    int64_t nT = 0;
    while (f1 < cells.size() && (f1 == f0 || nT + (cptr[(size_t)cells[f1] + 1] - cptr[(size_t)cells[f1]]) <= cap_t)) {
      nT += cptr[(size_t)cells[f1] + 1] - cptr[(size_t)cells[f1]];
      ++f1;
    }
    std::vector<int64_t> ent;
    ent.reserve((size_t)nT);
    std::vector<int64_t> toff;  // first batch position of every cell of the batch
    for (size_t f = f0; f < f1; ++f) {
      toff.push_back((int64_t)ent.size());
      for (int64_t e = cptr[(size_t)cells[f]]; e < cptr[(size_t)cells[f] + 1]; ++e) ent.push_back(e);
    }
    toff.push_back((int64_t)ent.size());
    // hypotheses of every cell of the batch, in the reference's scan order within each scan
    std::vector<std::vector<hyp>> hs(f1 - f0), hd(f1 - f0);
    std::vector<item> items;
    std::vector<int64_t> ioff;
    for (size_t f = f0; f < f1; ++f) {
      const muxgl_fmx_cell& x = all[(size_t)cells[f]];
      double mag = 1.0;
      for (double v : {x.sngBestLLK, x.sngNextLLK, x.dblBestLLK, x.dblNextLLK})
        if (v > -1e299) mag = std::max(mag, fabs(v));
      const double eps = 1e-9 * mag;
      auto near = [&](double a, double b) { return a > -1e299 && b > -1e299 && fabs(a - b) <= eps; };
      std::vector<hyp>& s = hs[f - f0];
      std::vector<hyp>& d = hd[f - f0];
      if (near(x.sngNextLLK, x.sngThirdLLK)) {
        for (int32_t j = 0; j < K; ++j) s.push_back(hyp{j, j, 0.0});
      } else {
        if (x.sBest >= 0) s.push_back(hyp{x.sBest, x.sBest, 0.0});
        if (x.sNext >= 0) s.push_back(hyp{x.sNext, x.sNext, 0.0});
        std::sort(s.begin(), s.end(), [](const hyp& a, const hyp& b) { return a.j < b.j; });
      }
      if (near(x.dblNextLLK, x.dblThirdLLK)) {
        for (int32_t j = 0; j < K; ++j)
          for (int32_t k = 0; k < j; ++k) d.push_back(hyp{j, k, 0.0});
      } else {
        if (x.dBest1 >= 0) d.push_back(hyp{x.dBest1, x.dBest2, 0.0});
        if (x.dNext1 >= 0) d.push_back(hyp{x.dNext1, x.dNext2, 0.0});
        std::sort(d.begin(), d.end(), [](const hyp& a, const hyp& b) { return a.j != b.j ? a.j < b.j : a.k < b.k; });
      }
      ioff.push_back((int64_t)items.size());
      for (int64_t t = toff[f - f0]; t < toff[f - f0 + 1]; ++t) {
        for (const hyp& q : s) items.push_back(item{t, q.j, q.j});
        for (const hyp& q : d) items.push_back(item{t, q.j, q.k});
      }
    }
    ioff.push_back((int64_t)items.size());

    int64_t* d_ent = nullptr;
    item* d_items = nullptr;
    double *d_gp = nullptr, *d_glis = nullptr, *d_lk = nullptr;
    auto cleanup = [&]() {
      dev_free(&d_ent);
      dev_free(&d_items);
      dev_free(&d_gp);
      dev_free(&d_glis);
      dev_free(&d_lk);
    };
    std::vector<double> lk(items.size());
    int rc = upload(h, &d_ent, ent) || upload(h, &d_items, items) || dev_alloc(h, &d_gp, (size_t)std::max<int64_t>(nT, 1) * K * 3) ||
             dev_alloc(h, &d_glis, (size_t)std::max<int64_t>(nT, 1) * 9) || dev_alloc(h, &d_lk, items.size() ? items.size() : 1);
    if (!rc && nT > 0) {
      const int64_t n = nT * K;
      hipLaunchKernelGGL(gp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, nT, K, d_ent, h->d_entry_snp,
                         h->d_entry_rptr, h->d_reads, h->d_lut, h->d_af, h->d_snp_ptr, h->d_snp_entry, h->d_snp_cell,
                         h->d_prev_clust, p->geno_error, d_gp, d_glis);
      if (!items.empty())
        hipLaunchKernelGGL(lk_kernel, dim3((unsigned)((items.size() + 255) / 256)), dim3(256), 0, h->stream,
                           (int64_t)items.size(), K, d_items, d_gp, d_glis, d_lk);
      hipError_t e = hipGetLastError();
      if (e == hipSuccess && !items.empty())
        e = hipMemcpyAsync(lk.data(), d_lk, sizeof(double) * lk.size(), hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      if (e != hipSuccess) {
        cleanup();
        MUXGL_FAIL(h, "fmx_exact_resolve: %s", hipGetErrorString(e));
      }
    }
    cleanup();
    if (rc) return 1;

    for (size_t f = f0; f < f1; ++f) {
      const int32_t ci = cells[f];
      std::vector<hyp>& s = hs[f - f0];
      std::vector<hyp>& d = hd[f - f0];
      const size_t nh = s.size() + d.size();
      const int64_t L = toff[f - f0 + 1] - toff[f - f0];
      const double* q = lk.data() + ioff[f - f0];
      for (int64_t t = 0; t < L; ++t) {  // :454-455: entries in ascending SNP, one log each
        for (size_t i = 0; i < s.size(); ++i) s[i].ll += log(q[(size_t)t * nh + i]);
        for (size_t i = 0; i < d.size(); ++i) d[i].ll += log(q[(size_t)t * nh + s.size() + i]);
      }
      muxgl_fmx_cell c = all[(size_t)ci];
      const muxgl_fmx_cell before = c;
      top2 ts, td;
      for (size_t i = 0; i < s.size(); ++i) ts.push(s[i].ll, (int32_t)i);
      for (size_t i = 0; i < d.size(); ++i) td.push(d[i].ll, (int32_t)i);
      c.sBest = ts.b >= 0 ? s[(size_t)ts.b].j : -1;
      c.sNext = ts.n >= 0 ? s[(size_t)ts.n].j : -1;
      c.sngBestLLK = ts.bv, c.sngNextLLK = ts.nv;
      c.dBest1 = td.b >= 0 ? d[(size_t)td.b].j : -1;
      c.dBest2 = td.b >= 0 ? d[(size_t)td.b].k : -1;
      c.dNext1 = td.n >= 0 ? d[(size_t)td.n].j : -1;
      c.dNext2 = td.n >= 0 ? d[(size_t)td.n].k : -1;
      c.dblBestLLK = td.bv, c.dblNextLLK = td.nv;
      // state before this iteration (what the nchanged rules compare with, :523,543-544,566)
      const int32_t ps = prev[(size_t)ci];
      const int32_t ptype = (int8_t)(ps & 0xff), pj = (int8_t)((ps >> 8) & 0xff) == -1 ? -1 : ((ps >> 8) & 0xff),
                    pk = (int8_t)((ps >> 16) & 0xff) == -1 ? -1 : ((ps >> 16) & 0xff);
      int chg;
      c.clust = -1;  // :520
      if (c.dblBestLLK > c.sngBestLLK + 2) {  // :521
        chg = ptype != 1;
        c.type = 1;
        c.bestPP = (c.dblBestLLK + log_double_prior - c.sumLLK);
        c.jBest = c.dBest1, c.kBest = c.dBest2;
        c.bestLLK = c.dblBestLLK;
        if (c.dblNextLLK > c.sngBestLLK + 2) {
          c.jNext = c.dNext1, c.kNext = c.dNext2;
          c.nextLLK = c.dblNextLLK;
        } else {
          c.jNext = c.kNext = c.sBest;
          c.nextLLK = c.sngBestLLK;
        }
      } else if (c.sngBestLLK > c.sngNextLLK + 2) {  // :542
        chg = (ptype != 0) || (pj != c.sBest) || (pk != c.sBest);
        c.type = 0;
        c.bestPP = (c.sngBestLLK + log_single_prior - c.sumLLK);
        c.jBest = c.kBest = c.sBest;
        c.bestLLK = c.sngBestLLK;
        c.clust = c.sBest;
        if (c.dblBestLLK > c.sngNextLLK + 2) {
          c.jNext = c.dBest1, c.kNext = c.dBest2;
          c.nextLLK = c.dblBestLLK;
        } else {
          c.jNext = c.kNext = c.sNext;
          c.nextLLK = c.sngNextLLK;
        }
      } else {  // :565
        chg = ptype != 2;
        c.type = 2;
        c.bestPP = (c.sngBestLLK + log_single_prior - c.sumLLK);
        c.jBest = c.kBest = c.sBest;
        c.bestLLK = c.sngBestLLK;
        if (c.dblBestLLK > c.sngNextLLK + 2) {
          c.jNext = c.dBest1, c.kNext = c.dBest2;
          c.nextLLK = c.dblNextLLK;  // sic, :577
        } else {
          c.jNext = c.kNext = c.sNext;
          c.nextLLK = c.sngNextLLK;
        }
      }
      // (sngPP, sngOnlyPP and sumLLK stay as the kernel summed them: the recomputed values move them by ~1e-12 relative)
      // what the kernel counted for this cell
      const int was_chg = (before.type == 1) ? (ptype != 1)
                          : (before.type == 0) ? ((ptype != 0) || (pj != before.sBest) || (pk != before.sBest))
                                               : (ptype != 2);
      d_single += (c.type == 0) - (before.type == 0);
      d_amb += (c.type == 2) - (before.type == 2);
      d_changed += chg - was_chg;
      if (c.clust != before.clust) *reassigned = true;
      if (c.type != before.type || c.clust != before.clust || c.sBest != before.sBest || c.sNext != before.sNext ||
          c.dBest1 != before.dBest1 || c.dBest2 != before.dBest2 || c.dNext1 != before.dNext1 || c.dNext2 != before.dNext2)
        ++h->fmx_exact_changed;
      pidx.push_back(ci);
      prec.push_back(c);
    }
    f0 = f1;
  }
  h->fmx_exact_cells += nflag;
  // records, assignments and counters as the reference has them
  int32_t* d_idx = nullptr;
  muxgl_fmx_cell* d_rec = nullptr;
  if (upload(h, &d_idx, pidx) || upload(h, &d_rec, prec)) {
    dev_free(&d_idx);
    dev_free(&d_rec);
    return 1;
  }
  hipLaunchKernelGGL(patch_kernel, dim3((unsigned)((pidx.size() + 255) / 256)), dim3(256), 0, h->stream, (int)pidx.size(), d_idx,
                     d_rec, h->d_fcells, h->d_clust);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  dev_free(&d_idx);
  dev_free(&d_rec);
  if (e != hipSuccess) MUXGL_FAIL(h, "fmx_exact_resolve (patch): %s", hipGetErrorString(e));
  h->h_fstat[0] += (int32_t)d_single;
  h->h_fstat[1] += (int32_t)d_amb;
  h->h_fstat[2] += (int32_t)d_changed;
  HIPCHK(h, hipMemcpy(h->d_fstat, h->h_fstat, 3 * sizeof(int32_t), hipMemcpyHostToDevice));
  return 0;
}
