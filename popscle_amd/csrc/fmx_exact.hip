// fmx_exact.hip -- freemuxlet calls that rounding noise could decide, decided in the reference's own arithmetic.
//
// The E-step kernels form a cell's K (K + 1) / 2 log-likelihoods from cluster posteriors that went through reciprocal
// multiplies and in another association than the reference (products with exponent bookkeeping, one log per hypothesis):
// equal to ~1e-12, not to the last bit.  Every decision of an EM iteration -- best / next singlet and doublet
// (cmd_cram_freemux2.cpp:469-497), the four `> x + 2` thresholds of the re-assignment (:521-584) -- compares two such
// numbers, and the re-assignment feeds the ordered M-step, i.e. the next iteration.  fmx_call_kernel therefore lists the
// cells where a comparison's margin is within 1e-9 x max(1, |LL|) (muxgl_fmx_cell carries the third-largest value of each
// scan for that), and the three steps below settle them before the iteration's results leave the library:
//
//   A  fmx_exact_snps    the SNPs the listed cells' entries cover (host list; the handle's own cells)
//   B  fmx_exact_rows    for the SNPs of such a list that this handle OWNS (its M-step range): the cluster posteriors
//                        gp1s of :402-415 from the state of (cluster, SNP) as the reference's maps hold it when the E-step
//                        reads it -- merge() (sc_drop_seq.h:77-101) of the entries of the cells assigned to the cluster
//                        BEFORE this iteration that cover the SNP, in ascending cell id (:277-288, :590-596), every
//                        entry's pileup recomputed from its read bytes (sc_drop_seq.cpp:452-509).  Thread = (SNP, cluster);
//                        IEEE operations in the reference's order, nothing contracted (exact_arith.hpp)
//   C  fmx_exact_finish  given the rows of ALL listed SNPs: lk of :440-452 for every (entry, contested hypothesis) on the
//                        device; on the host llks += log(lk) over the cell's entries in ascending SNP (:454-455, glibc's
//                        log), the scans over the contested hypotheses in the reference's scan order, the re-assignment,
//                        the nchanged rules; records and assignments on the device are patched
//
// "Contested" = the named best and next of a scan, or every hypothesis of the scan when next and third are within reach
// too (e.g. clusters without cells: identical posteriors, exact ties in the reference, which keeps the first in scan
// order).  By induction over the iterations the assignments are the reference's, so the chains run over the reference's
// member sets.
//
// One handle with the whole pileup (muxgl_fmx_iterate): A, B, C back to back, then the ordered M-step again if an
// assignment changed.  Several ranks (the muxgl_fmx_iter_* phases; a device group): a cell's SNPs are spread over the
// ranks' M-step ranges, so between A and B the lists are united and between B and C the rows are gathered by the caller
// (popscle_amd/freemuxlet.py run_em; muxgl_group.hip) -- two small exchanges in the iterations that have such cells, none
// otherwise; the assignments are then exchanged again and the M-step repeated.  The results do not depend on the number
// of ranks.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.hpp"
#include "exact_arith.hpp"

namespace {

using exact_arith::entry_pileup;
using exact_arith::merge;

// rows[(i * K + c) * 3 ..] = gp1s of cluster c at SNP snps[i] (:402-415), for the SNPs inside [s0, s1); owned[i] = 1 there
__global__ void __launch_bounds__(256)
    rows_kernel(int64_t n, int K, const int32_t* __restrict__ snps, int64_t s0, int64_t s1,
                const int64_t* __restrict__ entry_rptr, const uint8_t* __restrict__ reads, const double* __restrict__ lut,
                const double* __restrict__ af, const int64_t* __restrict__ snp_ptr, const int64_t* __restrict__ snp_entry,
                const int32_t* __restrict__ snp_cell, const int32_t* __restrict__ prev_clust, double geno_error,
                double* __restrict__ rows, uint8_t* __restrict__ owned) {
#pragma clang fp contract(off)
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n * K) return;
  const int64_t i = tid / K;
  const int c = (int)(tid % K);
  const int32_t snp = snps[i];
  if (snp < s0 || snp >= s1) return;
  double g[9];
  for (int q = 0; q < 9; ++q) g[q] = 1.0;  // snp_droplet_pileup() (sc_drop_seq.h:72-75): what operator[] default-constructs
  // Ascending cell id.  The lanes of a wave walk different chains (SNP, cluster) and a member of a SNP's list belongs to
  // ONE cluster: merging "when the member is mine" would run the merge -- a pileup from the read bytes and 36 true
  // divisions -- once per member with a sixteenth of the lanes active.  So every lane first moves its own cursor to its
  // NEXT member (a cheap scan, eight members' dependent loads in flight at a time), then all lanes merge together: the
  // heavy part runs as often as the longest chain of the wave has members, not as often as the list is long.
  int64_t p = snp_ptr[snp];
  const int64_t p1 = snp_ptr[snp + 1];
  for (;;) {
    int64_t hit = -1;
    while (hit < 0 && p < p1) {
      int32_t cl[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cl[u] = p + u < p1 ? snp_cell[p + u] : -1;
#pragma unroll
      for (int u = 0; u < 8; ++u) cl[u] = cl[u] >= 0 ? prev_clust[cl[u]] : -2;
      int u0 = 8;
#pragma unroll
      for (int u = 7; u >= 0; --u)
        if (cl[u] == c) u0 = u;
      if (u0 < 8) {
        hit = p + u0;
        p = hit + 1;
      } else {
        p += 8;
      }
    }
    if (!__any(hit >= 0)) break;  // (wave-uniform exit: every lane's list is exhausted)
    if (hit >= 0) {
      const int64_t pe = snp_entry[hit];
      double o[9];
      entry_pileup(reads, entry_rptr[pe], entry_rptr[pe + 1], lut, o);
      merge(g, o);
    }
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
  double* o = rows + (size_t)tid * 3;
  o[0] = gp1s[0], o[1] = gp1s[1], o[2] = gp1s[2];
  if (c == 0) owned[i] = 1;
}

// glis[t * 9 ..] = the pileup of entry ent[t], recomputed from its reads
__global__ void __launch_bounds__(256)
    glis_kernel(int64_t nT, const int64_t* __restrict__ ent, const int64_t* __restrict__ entry_rptr,
                const uint8_t* __restrict__ reads, const double* __restrict__ lut, double* __restrict__ glis) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nT) return;
  const int64_t e = ent[t];
  entry_pileup(reads, entry_rptr[e], entry_rptr[e + 1], lut, glis + (size_t)t * 9);
}

struct item {
  int32_t t;     // entry position in the batch
  int32_t r;     // its SNP's position in the list of rows
  int32_t j, k;  // hypothesis: k < j a pair, k == j the singlet of j
};

__global__ void __launch_bounds__(256)
    lk_kernel(int64_t n, int K, const item* __restrict__ items, const double* __restrict__ rows, const double* __restrict__ glis,
              double* __restrict__ lk_out) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const item it = items[i];
  const double* gl = glis + (size_t)it.t * 9;
  const double* gp1s = rows + ((size_t)it.r * K + it.j) * 3;
  double lk = 0;
  if (it.k == it.j) {  // :448-451
    for (int g1 = 0; g1 < 3; ++g1) lk += (gl[g1 * 3 + g1] * gp1s[g1]);
  } else {  // :440-445
    const double* gp2s = rows + ((size_t)it.r * K + it.k) * 3;
    for (int g1 = 0; g1 < 3; ++g1)
      for (int g2 = 0; g2 < 3; ++g2) lk += (gl[g1 * 3 + g2] * gp1s[g1] * gp2s[g2]);
  }
  lk_out[i] = lk;
}

// cells[idx[i]] = rec[i]; the assignment also at its place in the job-wide array the M-step reads (clust_all may be NULL);
// the exact scan results into the table fmx_call_kernel reads while the cell's inputs last
__global__ void patch_kernel(int n, const int32_t* __restrict__ idx, const muxgl_fmx_cell* __restrict__ rec,
                             muxgl_fmx_cell* __restrict__ cells, int32_t* __restrict__ clust, int32_t* __restrict__ clust_all,
                             fmx_xc* __restrict__ xc, int32_t* __restrict__ xc_epoch, int32_t epoch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const muxgl_fmx_cell r = rec[i];
  cells[idx[i]] = r;
  clust[idx[i]] = r.clust;
  if (clust_all) clust_all[idx[i]] = r.clust;
  xc[idx[i]] = fmx_xc{r.sBest, r.sNext, r.dBest1, r.dBest2, r.dNext1, r.dNext2, r.sngBestLLK, r.sngNextLLK, r.dblBestLLK, r.dblNextLLK};
  xc_epoch[idx[i]] = epoch;
}

// what the exact path needs of the listed cells, gathered: record, previous state, entry range
struct cell_info {
  muxgl_fmx_cell rec;
  int64_t e0, e1;
  int32_t prev, pad;
};
__global__ void gather_kernel(int n, const int32_t* __restrict__ idx, const muxgl_fmx_cell* __restrict__ cells,
                              const int32_t* __restrict__ prev_state, const int64_t* __restrict__ cell_ptr,
                              cell_info* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t c = idx[i];
  out[i].rec = cells[c];
  out[i].e0 = cell_ptr[c];
  out[i].e1 = cell_ptr[c + 1];
  out[i].prev = prev_state[c];
  out[i].pad = 0;
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

// dst[dst_off[b] + i] = src[src_off[b] + i] for i < dst_off[b + 1] - dst_off[b]: spans of a device array side by side (one
// workgroup per span), so that what the host needs of many cells comes over in one copy
template <class T>
__global__ void __launch_bounds__(256)
    gather_spans_kernel(const int64_t* __restrict__ src_off, const int64_t* __restrict__ dst_off, const T* __restrict__ src,
                        T* __restrict__ dst) {
  const int64_t s0 = src_off[blockIdx.x], d0 = dst_off[blockIdx.x], n = dst_off[blockIdx.x + 1] - d0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) dst[d0 + i] = src[s0 + i];
}

template <class T>
int upload(muxgl_handle* h, T** d, const T* v, size_t n) {
  if (dev_alloc(h, d, n ? n : 1)) return 1;
  if (n) HIPCHK(h, hipMemcpyAsync(*d, v, sizeof(T) * n, hipMemcpyHostToDevice, h->stream));
  return 0;
}

// out[dst_off[b] ...) = src[src_off[b] ...) for the nb spans (dst_off has nb + 1 entries); one launch, one copy back
template <class T>
int gather_spans(muxgl_handle* h, const std::vector<int64_t>& src_off, const std::vector<int64_t>& dst_off, const T* d_src,
                 T* out) {
  const size_t nb = src_off.size();
  const int64_t tot = dst_off.back();
  if (nb == 0 || tot == 0) return 0;
  int64_t *d_so = nullptr, *d_do = nullptr;
  T* d_out = nullptr;
  int rc = upload(h, &d_so, src_off.data(), nb) || upload(h, &d_do, dst_off.data(), nb + 1) || dev_alloc(h, &d_out, (size_t)tot);
  hipError_t e = hipSuccess;
  if (!rc) {
    hipLaunchKernelGGL(gather_spans_kernel<T>, dim3((unsigned)nb), dim3(256), 0, h->stream, d_so, d_do, d_src, d_out);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(T) * (size_t)tot, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  }
  dev_free(&d_so);
  dev_free(&d_do);
  dev_free(&d_out);
  if (rc) return 1;
  if (e != hipSuccess) MUXGL_FAIL(h, "exact calls (gather): %s", hipGetErrorString(e));
  return 0;
}

}  // namespace

// what step A leaves for step C
struct fmx_exact_state {
  std::vector<int32_t> cells;   // listed cells (local ids), ascending
  std::vector<cell_info> info;  // their records, previous states and entry ranges
  std::vector<int64_t> cptr;    // [cells + 1] positions of their entries in ent / esnp
  std::vector<int64_t> ent;     // entry ids
  std::vector<int32_t> esnp;    // their SNPs
};

void fmx_exact_release(muxgl_handle* h) {
  delete h->xs;
  h->xs = nullptr;
}

// A.  The stream is drained and h->h_fstat[3] cells are listed in h->d_flagged.  snps: unique, ascending.
int fmx_exact_snps(muxgl_handle* h, std::vector<int32_t>* snps) {
  snps->clear();
  fmx_exact_release(h);
  const int32_t nflag = h->h_fstat[3];
  h->xs = new fmx_exact_state;
  fmx_exact_state& xs = *h->xs;
  if (nflag <= 0) return 0;
  xs.cells.resize((size_t)nflag);
  HIPCHK(h, hipMemcpy(xs.cells.data(), h->d_flagged, sizeof(int32_t) * (size_t)nflag, hipMemcpyDeviceToHost));
  std::sort(xs.cells.begin(), xs.cells.end());  // (atomic order -> cell order: nothing may depend on it)
  {
    int32_t* d_idx = nullptr;
    cell_info* d_info = nullptr;
    xs.info.resize((size_t)nflag);
    int rc = upload(h, &d_idx, xs.cells.data(), xs.cells.size()) || dev_alloc(h, &d_info, (size_t)nflag);
    hipError_t e = hipSuccess;
    if (!rc) {
      hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((nflag + 255) / 256)), dim3(256), 0, h->stream, (int)nflag, d_idx, h->d_fcells,
                         h->d_prev_state, h->d_cell_ptr, d_info);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(xs.info.data(), d_info, sizeof(cell_info) * (size_t)nflag, hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    dev_free(&d_idx);
    dev_free(&d_info);
    if (rc) return 1;
    if (e != hipSuccess) MUXGL_FAIL(h, "fmx_exact_snps: %s", hipGetErrorString(e));
  }
  xs.cptr.push_back(0);
  for (const cell_info& ci : xs.info) {
    for (int64_t e = ci.e0; e < ci.e1; ++e) xs.ent.push_back(e);
    xs.cptr.push_back((int64_t)xs.ent.size());
  }
  xs.esnp.resize(xs.ent.size());
  {  // the listed cells' SNPs (a cell's entries are contiguous), gathered on the device: one copy
    std::vector<int64_t> so(xs.cells.size());
    for (size_t f = 0; f < xs.cells.size(); ++f) so[f] = xs.info[f].e0;
    if (gather_spans<int32_t>(h, so, xs.cptr, h->d_entry_snp, xs.esnp.data())) return 1;
  }
  *snps = xs.esnp;
  std::sort(snps->begin(), snps->end());
  snps->erase(std::unique(snps->begin(), snps->end()), snps->end());
  return 0;
}

// B.  rows[n][K][3] and owned[n] are filled for the SNPs of the handle's M-step range; the others are left untouched.
int fmx_exact_rows(muxgl_handle* h, const muxgl_fmx_params* p, const int32_t* snps, int64_t n, double* rows, uint8_t* owned) {
  if (n <= 0) return 0;
  muxgl_handle* m = h->col ? h->col : h;  // who holds the SNP-major view, its reads and the job-wide assignments
  const int K = h->K;
  int32_t* d_snps = nullptr;
  double* d_rows = nullptr;
  uint8_t* d_owned = nullptr;
  auto cleanup = [&]() {
    dev_free(&d_snps);
    dev_free(&d_rows);
    dev_free(&d_owned);
  };
  std::vector<double> tmp((size_t)n * K * 3);
  std::vector<uint8_t> own((size_t)n, 0);
  int rc = upload(h, &d_snps, snps, (size_t)n) || dev_alloc(h, &d_rows, (size_t)n * K * 3) || dev_alloc(h, &d_owned, (size_t)n);
  if (!rc) {
    hipError_t e = hipMemsetAsync(d_owned, 0, (size_t)n, h->stream);
    const int64_t nt = n * K;
    if (e == hipSuccess) {
      hipLaunchKernelGGL(rows_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, h->stream, n, K, d_snps, m->fs0, m->fs1,
                         m->d_entry_rptr, m->d_reads, h->d_lut, h->d_af, m->d_snp_ptr, m->d_snp_entry, m->d_snp_cell,
                         m->d_prev_clust, p->geno_error, d_rows, d_owned);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(tmp.data(), d_rows, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(own.data(), d_owned, own.size(), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
      cleanup();
      MUXGL_FAIL(h, "fmx_exact_rows: %s", hipGetErrorString(e));
    }
  }
  cleanup();
  if (rc) return 1;
  for (int64_t i = 0; i < n; ++i)
    if (own[(size_t)i]) {
      memcpy(rows + (size_t)i * K * 3, tmp.data() + (size_t)i * K * 3, sizeof(double) * (size_t)K * 3);
      if (owned) owned[i] = 1;
    }
  return 0;
}

// C.  rows[n][K][3] of the SNPs snps[n] (ascending), complete.  deltas[3]: what the listed cells' new calls add to
// (nsingle, namb, nchanged); *reassigned: an assignment changed (the M-step must run again).
int fmx_exact_finish(muxgl_handle* h, const muxgl_fmx_params* p, const int32_t* snps, int64_t n, const double* rows,
                     int64_t* deltas, int32_t* reassigned) {
  deltas[0] = deltas[1] = deltas[2] = 0;
  *reassigned = 0;
  if (!h->xs || h->xs->cells.empty()) {
    fmx_exact_release(h);
    return 0;
  }
  fmx_exact_state& xs = *h->xs;
  const int K = h->K;
  const size_t nf = xs.cells.size();
  const int64_t nT = (int64_t)xs.ent.size();
  const double log_single_prior = log((1.0 - p->doublet_prior) / K);          // :379
  const double log_double_prior = log(p->doublet_prior / K / (K - 1) * 2.0);  // :380

  // hypotheses of every listed cell, in the reference's scan order within each scan; one item per (entry, hypothesis)
  std::vector<std::vector<hyp>> hs(nf), hd(nf);
  const int64_t npairs = (int64_t)K * (K + 1) / 2;
  std::vector<item> items;
  std::vector<int64_t> ioff;
  auto eps_of = [](const muxgl_fmx_cell& x) {
    double mag = 1.0;
    for (double v : {x.sngBestLLK, x.sngNextLLK, x.dblBestLLK, x.dblNextLLK})
      if (v > -1e299) mag = std::max(mag, fabs(v));
    return 1e-9 * mag;
  };
  auto near_by = [](double a, double b, double eps) { return a > -1e299 && b > -1e299 && fabs(a - b) <= eps; };
  // the E-step's rows of the cells with a deep tie (below), gathered on the device: one copy
  std::vector<int64_t> deep_at(nf, -1);
  std::vector<double> deep_rows;
  {
    std::vector<int64_t> so, dof{0};
    for (size_t f = 0; f < nf; ++f) {
      const muxgl_fmx_cell& x = xs.info[f].rec;
      const double eps = eps_of(x);
      if (near_by(x.sngNextLLK, x.sngThirdLLK, eps) || near_by(x.dblNextLLK, x.dblThirdLLK, eps)) {
        deep_at[f] = dof.back();
        so.push_back((int64_t)xs.cells[f] * npairs);
        dof.push_back(dof.back() + npairs);
      }
    }
    deep_rows.resize((size_t)dof.back());
    if (gather_spans<double>(h, so, dof, h->d_fll, deep_rows.data())) return 1;
  }
  for (size_t f = 0; f < nf; ++f) {
    const muxgl_fmx_cell& x = xs.info[f].rec;
    const double eps = eps_of(x);
    auto near = [&](double a, double b) { return near_by(a, b, eps); };
    std::vector<hyp>& s = hs[f];
    std::vector<hyp>& d = hd[f];
    // three or more hypotheses of a scan within reach of each other: every hypothesis of the scan that the kernels' own
    // numbers (the cell's row of the E-step's result) do not put clearly below the runner-up -- 2 EPS below is beyond
    // what rounding can bridge, the kernels' deviation being orders of magnitude smaller than EPS
    const bool deep_s = near(x.sngNextLLK, x.sngThirdLLK), deep_d = near(x.dblNextLLK, x.dblThirdLLK);
    const double* fllrow = (deep_s || deep_d) ? deep_rows.data() + deep_at[f] : nullptr;
    if (deep_s) {
      for (int32_t j = 0; j < K; ++j)
        if (fllrow[(size_t)j * (j + 1) / 2 + j] >= x.sngNextLLK - 2 * eps) s.push_back(hyp{j, j, 0.0});
    } else {
      if (x.sBest >= 0) s.push_back(hyp{x.sBest, x.sBest, 0.0});
      if (x.sNext >= 0) s.push_back(hyp{x.sNext, x.sNext, 0.0});
      std::sort(s.begin(), s.end(), [](const hyp& a, const hyp& b) { return a.j < b.j; });
    }
    if (deep_d) {
      for (int32_t j = 0; j < K; ++j)
        for (int32_t k = 0; k < j; ++k)
          if (fllrow[(size_t)j * (j + 1) / 2 + k] >= x.dblNextLLK - 2 * eps) d.push_back(hyp{j, k, 0.0});
    } else {
      if (x.dBest1 >= 0) d.push_back(hyp{x.dBest1, x.dBest2, 0.0});
      if (x.dNext1 >= 0) d.push_back(hyp{x.dNext1, x.dNext2, 0.0});
      std::sort(d.begin(), d.end(), [](const hyp& a, const hyp& b) { return a.j != b.j ? a.j < b.j : a.k < b.k; });
    }
    ioff.push_back((int64_t)items.size());
    for (int64_t t = xs.cptr[f]; t < xs.cptr[f + 1]; ++t) {
      const int32_t* it = std::lower_bound(snps, snps + n, xs.esnp[(size_t)t]);
      if (it == snps + n || *it != xs.esnp[(size_t)t]) MUXGL_FAIL(h, "fmx_exact_finish: SNP %d is not in the list of rows", xs.esnp[(size_t)t]);
      const int32_t r = (int32_t)(it - snps);
      for (const hyp& q : s) items.push_back(item{(int32_t)t, r, q.j, q.j});
      for (const hyp& q : d) items.push_back(item{(int32_t)t, r, q.j, q.k});
    }
  }
  ioff.push_back((int64_t)items.size());

  int64_t* d_ent = nullptr;
  item* d_items = nullptr;
  double *d_rows = nullptr, *d_glis = nullptr, *d_lk = nullptr;
  auto cleanup = [&]() {
    dev_free(&d_ent);
    dev_free(&d_items);
    dev_free(&d_rows);
    dev_free(&d_glis);
    dev_free(&d_lk);
  };
  std::vector<double> lk(items.size());
  int rc = upload(h, &d_ent, xs.ent.data(), xs.ent.size()) || upload(h, &d_items, items.data(), items.size()) ||
           upload(h, &d_rows, rows, (size_t)n * K * 3) || dev_alloc(h, &d_glis, (size_t)std::max<int64_t>(nT, 1) * 9) ||
           dev_alloc(h, &d_lk, items.size() ? items.size() : 1);
  if (!rc && nT > 0) {
    hipLaunchKernelGGL(glis_kernel, dim3((unsigned)((nT + 255) / 256)), dim3(256), 0, h->stream, nT, d_ent, h->d_entry_rptr,
                       h->d_reads, h->d_lut, d_glis);
    if (!items.empty())
      hipLaunchKernelGGL(lk_kernel, dim3((unsigned)((items.size() + 255) / 256)), dim3(256), 0, h->stream, (int64_t)items.size(),
                         K, d_items, d_rows, d_glis, d_lk);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !items.empty())
      e = hipMemcpyAsync(lk.data(), d_lk, sizeof(double) * lk.size(), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
      cleanup();
      MUXGL_FAIL(h, "fmx_exact_finish: %s", hipGetErrorString(e));
    }
  }
  cleanup();
  if (rc) return 1;

  std::vector<int32_t> pidx;
  std::vector<muxgl_fmx_cell> prec;
  for (size_t f = 0; f < nf; ++f) {
    const int32_t ci = xs.cells[f];
    std::vector<hyp>& s = hs[f];
    std::vector<hyp>& d = hd[f];
    const size_t nh = s.size() + d.size();
    const int64_t L = xs.cptr[f + 1] - xs.cptr[f];
    const double* q = lk.data() + ioff[f];
    for (int64_t t = 0; t < L; ++t) {  // :454-455: entries in ascending SNP, one log each
      for (size_t i = 0; i < s.size(); ++i) s[i].ll += log(q[(size_t)t * nh + i]);
      for (size_t i = 0; i < d.size(); ++i) d[i].ll += log(q[(size_t)t * nh + s.size() + i]);
    }
    muxgl_fmx_cell c = xs.info[f].rec;
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
    const int32_t ps = xs.info[f].prev;
    auto byte = [](int32_t v) { return v == 0xff ? -1 : v; };
    const int32_t ptype = (int8_t)(ps & 0xff), pj = byte((ps >> 8) & 0xff), pk = byte((ps >> 16) & 0xff);
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
    deltas[0] += (c.type == 0) - (before.type == 0);
    deltas[1] += (c.type == 2) - (before.type == 2);
    deltas[2] += chg - was_chg;
    if (c.clust != before.clust) *reassigned = 1;
    if (c.type != before.type || c.clust != before.clust || c.sBest != before.sBest || c.sNext != before.sNext ||
        c.dBest1 != before.dBest1 || c.dBest2 != before.dBest2 || c.dNext1 != before.dNext1 || c.dNext2 != before.dNext2)
      ++h->fmx_exact_changed;
    pidx.push_back(ci);
    prec.push_back(c);
  }
  h->fmx_exact_cells += (int64_t)nf;
  fmx_exact_release(h);
  // records and assignments as the reference has them
  int32_t* d_idx = nullptr;
  muxgl_fmx_cell* d_rec = nullptr;
  if (upload(h, &d_idx, pidx.data(), pidx.size()) || upload(h, &d_rec, prec.data(), prec.size())) {
    dev_free(&d_idx);
    dev_free(&d_rec);
    return 1;
  }
  hipLaunchKernelGGL(patch_kernel, dim3((unsigned)((pidx.size() + 255) / 256)), dim3(256), 0, h->stream, (int)pidx.size(), d_idx,
                     d_rec, h->d_fcells, h->d_clust, h->col ? h->col->d_clust + h->cell_base : nullptr, h->d_xc, h->d_xc_epoch,
                     h->xs_epoch);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  dev_free(&d_idx);
  dev_free(&d_rec);
  if (e != hipSuccess) MUXGL_FAIL(h, "fmx_exact_finish (patch): %s", hipGetErrorString(e));
  return 0;
}

// the three steps on a handle that holds the whole job (muxgl_fmx_iterate)
int fmx_exact_resolve(muxgl_handle* h, const muxgl_fmx_params* p, bool* reassigned) {
  *reassigned = false;
  host_timer tm;  // MUXGL_TIMING=1: the three steps on stderr
  std::vector<int32_t> snps;
  if (fmx_exact_snps(h, &snps)) return 1;
  tm.lap("exact calls: listed cells and their SNPs");
  std::vector<double> rows(snps.size() * (size_t)h->K * 3);
  int64_t deltas[3];
  int32_t re = 0;
  if (fmx_exact_rows(h, p, snps.data(), (int64_t)snps.size(), rows.data(), nullptr)) return 1;
  tm.lap("exact calls: posterior rows (ordered chains)");
  if (fmx_exact_finish(h, p, snps.data(), (int64_t)snps.size(), rows.data(), deltas, &re)) return 1;
  tm.lap("exact calls: hypotheses, decisions, patch");
  for (int i = 0; i < 3; ++i) h->h_fstat[i] += (int32_t)deltas[i];
  h->h_fstat[3] = 0;
  HIPCHK(h, hipMemcpy(h->d_fstat, h->h_fstat, 4 * sizeof(int32_t), hipMemcpyHostToDevice));
  *reassigned = re != 0;
  return 0;
}
