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
// A thread = one (entry of a flagged cell, cluster); all flagged steps of a pass share launches.  The chain needs the SNP's other entries in PROCESSING order, while the
// SNP-major view lists them by cell id: the view is therefore sorted once per run by (SNP, step index) -- by_step, built at
// the first near tie of a muxgl_fmx_greedy_init call (one radix sort of the pileup's entries) -- and a thread walks its
// SNP's stretch up to the current step: one pass over the list instead of a rescan per chain link.
#pragma once
#include <math.h>

#include <algorithm>

#include <thread>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.hpp"
#include "exact_arith.hpp"

namespace greedy_exact {

using exact_arith::entry_pileup;
using exact_arith::merge;

// the SNP-major view in (SNP, step) order: key = SNP << 32 | step index of the entry's cell (0x7fffffff: not clustered),
// pos = the entry's position in the SNP-major arrays (snp_entry / snp_cell); SNP s owns [snp_ptr[s], snp_ptr[s + 1]) here
// as there
struct by_step {
  uint64_t* key = nullptr;
  int64_t* pos = nullptr;
};

__global__ void __launch_bounds__(256)
    by_step_keys_kernel(int64_t nnz, const int64_t* __restrict__ snp_entry, const int32_t* __restrict__ snp_cell,
                        const int32_t* __restrict__ entry_snp, const int32_t* __restrict__ step_of_cell,
                        uint64_t* __restrict__ key, int64_t* __restrict__ pos) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x) {
    key[p] = ((uint64_t)(uint32_t)entry_snp[snp_entry[p]] << 32) | (uint32_t)step_of_cell[snp_cell[p]];
    pos[p] = p;
  }
}

inline void release(by_step* b) {
  dev_free(&b->key);
  dev_free(&b->pos);
}

// d_step_of_cell[C]: a cell's index in the processing order, or a value >= the number of steps for cells that are not
// clustered.  0, or 1 with h->err set.
inline int build_by_step(muxgl_handle* h, const int32_t* d_step_of_cell, by_step* b) {
  const int64_t nnz = h->nnz;
  release(b);
  if (nnz == 0) return 0;
  uint64_t* key_in = nullptr;
  int64_t* pos_in = nullptr;
  void* tmp = nullptr;
  size_t tb = 0;
  hipError_t e = hipSuccess;
  int rc = dev_alloc(h, &key_in, (size_t)nnz) || dev_alloc(h, &pos_in, (size_t)nnz) || dev_alloc(h, &b->key, (size_t)nnz) ||
           dev_alloc(h, &b->pos, (size_t)nnz);
  if (!rc) {
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(by_step_keys_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, nnz, h->d_snp_entry, h->d_snp_cell,
                       h->d_entry_snp, d_step_of_cell, key_in, pos_in);
    e = hipGetLastError();
    if (e == hipSuccess)
      e = rocprim::radix_sort_pairs(nullptr, tb, key_in, b->key, pos_in, b->pos, (size_t)nnz, 0u, 64u, h->stream);
    if (e == hipSuccess) e = dev_malloc_retry(&tmp, tb ? tb : 1);
    if (e == hipSuccess)
      e = rocprim::radix_sort_pairs(tmp, tb, key_in, b->key, pos_in, b->pos, (size_t)nnz, 0u, 64u, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  }
  if (tmp) (void)hipFree(tmp);
  dev_free(&key_in);
  dev_free(&pos_in);
  if (rc || e != hipSuccess) {
    release(b);
    if (h->err.empty()) h->err = std::string("muxgl_fmx_greedy_init (exact path, members by step): ") +
                                 (rc ? "device allocation failed" : hipGetErrorString(e));
    return 1;
  }
  return 0;
}

// One request = one step of the processing order: the cell's entries [e0, e0 + L), its step index, and where its
// L x K pairs of terms go.
struct step_req {
  int64_t e0;
  int64_t off;  // first (t, c) slot of the request in the launch's output
  int32_t L;
  int32_t step;
};

// out[(off + t * K + c) * 2] = {lk2, lk0} of entry e0 + t against cluster c; lk2 = -1 marks "the cluster does not hold the
// SNP".  Workgroup b works on request blk_req[b], threads blk_first[b] ... of its L x K.
__global__ void __launch_bounds__(256)
    terms_kernel(const step_req* __restrict__ req, const int32_t* __restrict__ blk_req, const int32_t* __restrict__ blk_first,
                 int K, const int32_t* __restrict__ entry_snp, const int64_t* __restrict__ entry_rptr,
                 const uint8_t* __restrict__ reads, const double* __restrict__ lut, const double* __restrict__ af,
                 const int64_t* __restrict__ snp_ptr, const int64_t* __restrict__ snp_entry,
                 const int32_t* __restrict__ snp_cell, const uint64_t* __restrict__ bs_key, const int64_t* __restrict__ bs_pos,
                 const int32_t* __restrict__ clust, double* __restrict__ out) {
#pragma clang fp contract(off)
  const step_req r = req[blk_req[blockIdx.x]];
  const int64_t tid = (int64_t)blk_first[blockIdx.x] + threadIdx.x;
  if (tid >= (int64_t)r.L * K) return;
  const int t = (int)(tid / K), c = (int)(tid % K);
  const int64_t e = r.e0 + t;
  const int32_t snp = entry_snp[e];
  double gljs[9];
  for (int i = 0; i < 9; ++i) gljs[i] = 1.0;  // snp_droplet_pileup() (sc_drop_seq.h:72-75)
  bool present = false;
  const int64_t p1 = snp_ptr[snp + 1];
  for (int64_t q = snp_ptr[snp]; q < p1; ++q) {  // members in processing order, up to the current step
    if ((uint32_t)bs_key[q] >= (uint32_t)r.step) break;
    const int64_t p = bs_pos[q];
    if (clust[snp_cell[p]] != c) continue;
    const int64_t pe = snp_entry[p];
    double o[9];
    entry_pileup(reads, entry_rptr[pe], entry_rptr[pe + 1], lut, o);
    merge(gljs, o);
    present = true;
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
  out[(r.off + tid) * 2] = lk2;
  out[(r.off + tid) * 2 + 1] = lk0;
}

// device buffers of decide_many, kept between calls (a repair decides steps one at a time: no allocation per step)
struct scratch {
  step_req* d_req = nullptr;
  int32_t *d_br = nullptr, *d_bf = nullptr;
  double* d_out = nullptr;
  size_t n_req = 0, n_blk = 0, n_out = 0;
};
inline void release(scratch* s) {
  dev_free(&s->d_req);
  dev_free(&s->d_br);
  dev_free(&s->d_bf);
  dev_free(&s->d_out);
  s->n_req = s->n_blk = s->n_out = 0;
}
inline int reserve(muxgl_handle* h, scratch* s, size_t n_req, size_t n_blk, size_t n_out) {
  if (n_req > s->n_req) {
    dev_free(&s->d_req);
    s->n_req = 0;
    const size_t want = std::max<size_t>(n_req, 256);
    if (dev_alloc(h, &s->d_req, want)) return 1;
    s->n_req = want;
  }
  if (n_blk > s->n_blk) {
    dev_free(&s->d_br);
    dev_free(&s->d_bf);
    s->n_blk = 0;
    const size_t want = std::max<size_t>(n_blk + n_blk / 2, 4096);
    if (dev_alloc(h, &s->d_br, want) || dev_alloc(h, &s->d_bf, want)) return 1;
    s->n_blk = want;
  }
  if (n_out > s->n_out) {
    dev_free(&s->d_out);
    s->n_out = 0;
    const size_t want = std::max<size_t>(n_out + n_out / 2, (size_t)1 << 18);
    if (dev_alloc(h, &s->d_out, want)) return 1;
    s->n_out = want;
  }
  return 0;
}

// The reference's decisions for the steps rq[] (cells with entries [e0, e0 + L)), each given the decisions of the steps
// before it as they stand in d_clust -- the requests do not see each other's results; the caller uses a decision only
// where that is right (muxgl_fmx_greedy_init: steps that share no SNP with an overruled one).  bs: build_by_step() of the
// run's step indices.  win[n]: the cluster per request; scores[n * K]: llk2 - llk0 per cluster.  All requests of a call
// share launches (at most 2^24 pairs of terms each) and one copy back per launch; scr keeps the device buffers between
// calls.  0, or 1 with h->err set.
inline int decide_many(muxgl_handle* h, std::vector<step_req>& rq, int K, const by_step& bs, const int32_t* d_clust,
                       std::vector<int32_t>& win, std::vector<double>& scores, scratch* scr) {
  const size_t n = rq.size();
  win.assign(n, 0);
  scores.assign(n * (size_t)K, 0.0);
  constexpr int64_t MAX_SLOTS = (int64_t)1 << 24;
  size_t k0 = 0;
  while (k0 < n) {
    std::vector<int32_t> blk_req, blk_first;
    int64_t slots = 0;
    size_t k1 = k0;
    while (k1 < n && (k1 == k0 || slots + (int64_t)rq[k1].L * K <= MAX_SLOTS)) {
      rq[k1].off = slots;
      const int64_t m = (int64_t)rq[k1].L * K;
      for (int64_t f = 0; f < m; f += 256) {
        blk_req.push_back((int32_t)(k1 - k0));
        blk_first.push_back((int32_t)f);
      }
      slots += m;
      ++k1;
    }
    std::vector<double> out((size_t)slots * 2);
    if (slots > 0) {
      const size_t nb = blk_req.size();
      const int rc = reserve(h, scr, k1 - k0, nb, (size_t)slots * 2);
      step_req* d_req = scr->d_req;
      int32_t *d_br = scr->d_br, *d_bf = scr->d_bf;
      double* d_out = scr->d_out;
      hipError_t e = rc ? hipErrorOutOfMemory : hipSuccess;
      if (e == hipSuccess) e = hipMemcpyAsync(d_req, rq.data() + k0, sizeof(step_req) * (k1 - k0), hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(d_br, blk_req.data(), sizeof(int32_t) * nb, hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(d_bf, blk_first.data(), sizeof(int32_t) * nb, hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(terms_kernel, dim3((unsigned)nb), dim3(256), 0, h->stream, d_req, d_br, d_bf, K, h->d_entry_snp,
                           h->d_entry_rptr, h->d_reads, h->d_lut, h->d_af, h->d_snp_ptr, h->d_snp_entry, h->d_snp_cell, bs.key,
                           bs.pos, d_clust, d_out);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipMemcpyAsync(out.data(), d_out, sizeof(double) * out.size(), hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      if (e != hipSuccess) {
        if (h->err.empty()) h->err = std::string("muxgl_fmx_greedy_init (exact path): ") + hipGetErrorString(e);
        return 1;
      }
    }
    auto sum_requests = [&](size_t ka, size_t kb) {
      for (size_t k = ka; k < kb; ++k) {
        const int L = rq[k].L;
        const double* o = out.data() + (size_t)rq[k].off * 2;
        int maxClust = 0;  // (L == 0: every distance is a sum over nothing, maxClust = 0, :232-233)
        double maxScore = 0;
        for (int c = 0; c < K; ++c) {
          double llk2 = 0, llk0 = 0;  // dropD (sc_drop_seq.h:45-52)
          for (int t = 0; t < L; ++t) {
            const double lk2 = o[((size_t)t * K + c) * 2], lk0 = o[((size_t)t * K + c) * 2 + 1];
            if (lk2 < 0) continue;  // jt == clustPileup.end()
            llk2 += log(lk2);
            llk0 += log(lk0);
          }
          const double sc = llk2 - llk0;
          scores[k * (size_t)K + c] = sc;
          if (c == 0) {
            maxScore = sc;
          } else if (sc > maxScore) {  // :235-242
            maxClust = c;
            maxScore = sc;
          }
        }
        win[k] = maxClust;
      }
    };
    // (two logs per slot: with many slots the requests are shared out to host threads, by slots)
    const int nt = slots > (1 << 15) && k1 - k0 > 1 ? (int)std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (nt <= 1) {
      sum_requests(k0, k1);
    } else {
      std::vector<std::thread> th;
      size_t a = k0;
      for (int t = 0; t < nt && a < k1; ++t) {
        const int64_t want = slots * (t + 1) / nt;
        size_t b = a;
        while (b < k1 && (b + 1 == k1 ? slots : rq[b + 1].off) <= want) ++b;
        if (t == nt - 1) b = k1;
        if (b > a) th.emplace_back(sum_requests, a, b);
        a = b;
      }
      for (auto& x : th) x.join();
    }
    k0 = k1;
  }
  return 0;
}

}  // namespace greedy_exact
