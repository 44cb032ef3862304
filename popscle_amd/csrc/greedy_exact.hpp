// greedy_exact.hpp -- the decision of ONE step of freemux2's greedy initial clustering in the reference's own arithmetic
// (cmd_cram_freemux2.cpp:217-261, sc_drop_seq.cpp:452-509,544-578, sc_drop_seq.h:77-101), for the cells the fast kernels
// flag as near ties (fmx_greedy.hip: greedy_near_tie).
//
// What the reference's numbers are made of: IEEE double multiplications, additions, divisions and comparisons -- which a
// gfx950 lane reproduces bit for bit when nothing is contracted or re-associated -- and, at the very end, glibc's log().
// So the device forms, for the flagged cell and every (entry, cluster), the two likelihood terms lk2 and lk0 exactly as
// the reference does:
//   * the pileup of an entry from its reads (calculate_snp_droplet_pileup, alpha = 0.5: factor = Mat x frac + Err / 4.,
//     division by the sum after every read, clamp at 1e-6, division by the sum) -- recomputed from the read bytes, because
//     the library's entry table is made with reciprocal multiplies and may differ from the reference's in the last bit;
//   * the state of (cluster, SNP): merge() of the earlier cells of the processing order that joined the cluster and
//     cover the SNP, in that order, from the all-ones state std::map::operator[] default-constructs; "present" iff there
//     is at least one (the key exists, sc_drop_seq.cpp:549-550);
//   * lk2 += (glis[gi*3+gi] * gljs[gi*3+gi] * gps[gi]),  lk0 += (glis[gi*3+gi] * gljs[gj*3+gj] * gps[gi] * gps[gj]) in
//     the reference's loop order (:563-568) with gps from af as (:555-558);
// and the host adds log(lk2), log(lk0) over the cell's entries in ascending SNP order (:574-575) with glibc's log and takes
// the reference's argmax (strict '>' from cluster 0, cmd_cram_freemux2.cpp:233-242).  No tolerance is involved: the
// result is the reference's decision given the earlier ones.
//
// A thread = one (entry of the cell, cluster).  The SNP's other entries come from the SNP-major view (ascending cell id);
// the chain needs them in processing order, so the thread repeatedly picks the member with the smallest step index
// above the last one (lists are a few hundred long and this path runs for a handful of cells per run, if any).
#pragma once
#include <math.h>

#include <vector>

#include "common.hpp"
#include "exact_arith.hpp"

namespace greedy_exact {

using exact_arith::entry_pileup;
using exact_arith::merge;

// out[(t * K + c) * 2] = {lk2, lk0} of entry e0 + t against cluster c; lk2 = -1 marks "the cluster does not hold the SNP"
__global__ void __launch_bounds__(256)
    terms_kernel(int64_t e0, int L, int K, int32_t step, const int32_t* __restrict__ entry_snp,
                 const int64_t* __restrict__ entry_rptr, const uint8_t* __restrict__ reads, const double* __restrict__ lut,
                 const double* __restrict__ af, const int64_t* __restrict__ snp_ptr, const int64_t* __restrict__ snp_entry,
                 const int32_t* __restrict__ snp_cell, const int32_t* __restrict__ step_of_cell,
                 const int32_t* __restrict__ clust, double* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (int64_t)L * K) return;
  const int t = (int)(tid / K), c = (int)(tid % K);
  const int64_t e = e0 + t;
  const int32_t snp = entry_snp[e];
  double gljs[9];
  for (int i = 0; i < 9; ++i) gljs[i] = 1.0;  // snp_droplet_pileup() (sc_drop_seq.h:72-75)
  bool present = false;
  int32_t last = -1;
  const int64_t p0 = snp_ptr[snp], p1 = snp_ptr[snp + 1];
  for (;;) {  // members in processing order: the smallest step index above `last`
    int32_t nxt = 0x7fffffff;
    int64_t pe = -1;
    for (int64_t p = p0; p < p1; ++p) {
      const int32_t cell = snp_cell[p];
      const int32_t st = step_of_cell[cell];
      if (st > last && st < step && st < nxt && clust[cell] == c) {
        nxt = st;
        pe = snp_entry[p];
      }
    }
    if (pe < 0) break;
    double o[9];
    entry_pileup(reads, entry_rptr[pe], entry_rptr[pe + 1], lut, o);
    merge(gljs, o);
    present = true;
    last = nxt;
  }
  double lk2 = -1.0, lk0 = -1.0;
  if (present) {
    double glis[9];
    entry_pileup(reads, entry_rptr[e], entry_rptr[e + 1], lut, glis);
    const double a = af[snp];
    double gps[3];
    gps[0] = (1.0 - a) * (1.0 - a);
    gps[1] = 2.0 * a * (1.0 - a);
    gps[2] = a * a;
    lk0 = 0;
    lk2 = 0;
    for (int gi = 0; gi < 3; ++gi) {
      lk2 += (glis[gi * 3 + gi] * gljs[gi * 3 + gi] * gps[gi]);
      for (int gj = 0; gj < 3; ++gj) lk0 += (glis[gi * 3 + gi] * gljs[gj * 3 + gj] * gps[gi] * gps[gj]);
    }
  }
  out[tid * 2] = lk2;
  out[tid * 2 + 1] = lk0;
}

// The reference's decision for step `step` (cell with entries [e0, e0 + L)) given the decisions of the earlier steps in
// d_clust.  d_step_of_cell[C]: step index of a cell in the processing order, or a value >= the number of steps for cells
// that are not clustered.  scores_out: NULL or [K] (llk2 - llk0 per cluster).  Returns the cluster, or -1 on a HIP error.
inline int decide(muxgl_handle* h, int64_t e0, int L, int K, int32_t step, const int32_t* d_step_of_cell,
                  const int32_t* d_clust, double* scores_out) {
  if (L == 0) return 0;  // every distance is a sum over nothing: maxClust = 0 (:232-233)
  double* d_out = nullptr;
  if (dev_alloc(h, &d_out, (size_t)L * K * 2)) return -1;
  const int64_t n = (int64_t)L * K;
  hipLaunchKernelGGL(terms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, e0, L, K, step,
                     h->d_entry_snp, h->d_entry_rptr, h->d_reads, h->d_lut, h->d_af, h->d_snp_ptr, h->d_snp_entry,
                     h->d_snp_cell, d_step_of_cell, d_clust, d_out);
  std::vector<double> out((size_t)n * 2);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out.data(), d_out, sizeof(double) * out.size(), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  dev_free(&d_out);
  if (e != hipSuccess) return -1;
  int maxClust = 0;
  double maxScore = 0;
  for (int c = 0; c < K; ++c) {
    double llk2 = 0, llk0 = 0;  // dropD (sc_drop_seq.h:45-52)
    for (int t = 0; t < L; ++t) {
      const double lk2 = out[((size_t)t * K + c) * 2], lk0 = out[((size_t)t * K + c) * 2 + 1];
      if (lk2 < 0) continue;  // jt == clustPileup.end()
      llk2 += log(lk2);
      llk0 += log(lk0);
    }
    const double sc = llk2 - llk0;
    if (scores_out) scores_out[c] = sc;
    if (c == 0) {
      maxScore = sc;
    } else if (sc > maxScore) {  // :235-242
      maxClust = c;
      maxScore = sc;
    }
  }
  return maxClust;
}

}  // namespace greedy_exact
