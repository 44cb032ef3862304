// score_exact.hpp -- the singlet scores of muxgl_fmx_prepare, bit for bit the reference's where their ORDER depends on it.
//
// cmd_cram_freemux2.cpp:116-159 gives every cell llk0 = sum log(lk0), llk2 = sum log(lk2) over its entries and sorts the
// cells by llk2 - llk0, descending, ties by descending index (sc_drop_seq.h:190-198, cmd_cram_freemux2.cpp:183-189); the
// greedy start (:217-261) processes them in that order, so the order is part of every later decision.  The device sums
// (fmx_entry_kernel, fmx_cell_score_kernel: per-entry logs added as a tree) equal the reference's to ~1e-13, not to the
// last bit -- and real pileups do hold cells whose scores are equal up to rounding noise: with one read per entry lk2 = lk0
// mathematically, so a droplet of single-read entries scores 0 +- 1e-14 and its place among its likes is decided by that
// noise.  Therefore: every cell whose score is within EPS = 1e-9 x max(1, |llk0|, |llk2|) of a neighbour in the sorted
// order gets its two sums recomputed exactly as the reference forms them --
//   * per entry, on the device with nothing contracted (exact_arith.hpp): the pileup from the read bytes
//     (calculate_snp_droplet_pileup, sc_drop_seq.cpp:452-509), gps from af (:133-136), lk2 / lk0 in the reference's
//     loop order (:138-143);
//   * on the host: llk += log(lk) with glibc's log over the cell's entries in ascending SNP order (:147-148);
// -- and the pass repeats until no cell without exact sums has such a neighbour (the exact values move by ~1e-13, so one
// round and a check is the rule).  Cells beyond EPS of everyone keep the device's sums: their place cannot differ.
#pragma once
#include <math.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "common.hpp"
#include "exact_arith.hpp"

namespace score_exact {

// out[(off[k] + i) * 2] = {lk0, lk2} of entry i of cell cells[k]; one workgroup per listed cell
// (static: the header is part of two translation units)
static __global__ void __launch_bounds__(256)
    terms_kernel(const int32_t* __restrict__ cells, const int64_t* __restrict__ off, const int64_t* __restrict__ cell_ptr,
                 const int32_t* __restrict__ entry_snp, const int64_t* __restrict__ entry_rptr,
                 const uint8_t* __restrict__ reads, const double* __restrict__ lut, const double* __restrict__ af,
                 double* __restrict__ out) {
#pragma clang fp contract(off)
  const int32_t c = cells[blockIdx.x];
  const int64_t e0 = cell_ptr[c], e1 = cell_ptr[c + 1], o = off[blockIdx.x];
  for (int64_t e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
    double gls[9];
    exact_arith::entry_pileup(reads, entry_rptr[e], entry_rptr[e + 1], lut, gls);
    const double a = af[entry_snp[e]];
    double gps[3];
    gps[0] = (1.0 - a) * (1.0 - a);
    gps[1] = 2.0 * a * (1.0 - a);
    gps[2] = a * a;
    double lk0 = 0, lk2 = 0;
    for (int gi = 0; gi < 3; ++gi) {
      lk2 += (gls[gi * 3 + gi] * gps[gi]);
      for (int gj = 0; gj < 3; ++gj) lk0 += (gls[gi * 3 + gj] * gps[gi] * gps[gj]);
    }
    out[(o + (e - e0)) * 2] = lk0;
    out[(o + (e - e0)) * 2 + 1] = lk2;
  }
}

// exact llk0 / llk2 of the listed cells into l0 / l2 (indexed by cell).  cp = host copy of cell_ptr.
inline int compute(muxgl_handle* h, const std::vector<int32_t>& cells, const std::vector<int64_t>& cp, double* l0, double* l2) {
  constexpr int64_t MAX_ENTRIES = (int64_t)1 << 25;  // per launch: 512 MB of terms
  size_t k0 = 0;
  while (k0 < cells.size()) {
    std::vector<int64_t> off;
    int64_t tot = 0;
    size_t k1 = k0;
    while (k1 < cells.size() && (k1 == k0 || tot + (cp[(size_t)cells[k1] + 1] - cp[(size_t)cells[k1]]) <= MAX_ENTRIES)) {
      off.push_back(tot);
      tot += cp[(size_t)cells[k1] + 1] - cp[(size_t)cells[k1]];
      ++k1;
    }
    const size_t nc = k1 - k0;
    std::vector<double> out((size_t)tot * 2);
    if (tot > 0) {
      int32_t* d_cells = nullptr;
      int64_t* d_off = nullptr;
      double* d_out = nullptr;
      int rc = dev_alloc(h, &d_cells, nc) || dev_alloc(h, &d_off, nc) || dev_alloc(h, &d_out, (size_t)tot * 2);
      hipError_t e = rc ? hipErrorOutOfMemory : hipSuccess;
      if (e == hipSuccess) e = hipMemcpyAsync(d_cells, cells.data() + k0, sizeof(int32_t) * nc, hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(d_off, off.data(), sizeof(int64_t) * nc, hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(terms_kernel, dim3((unsigned)nc), dim3(256), 0, h->stream, d_cells, d_off, h->d_cell_ptr,
                           h->d_entry_snp, h->d_entry_rptr, h->d_reads, h->d_lut, h->d_af, d_out);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipMemcpyAsync(out.data(), d_out, sizeof(double) * out.size(), hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      dev_free(&d_cells);
      dev_free(&d_off);
      dev_free(&d_out);
      if (e != hipSuccess) {
        if (h->err.empty()) h->err = std::string("muxgl_fmx_prepare (exact scores): ") + hipGetErrorString(e);
        return 1;
      }
    }
    auto sum_cells = [&](size_t a, size_t b) {
      for (size_t k = a; k < b; ++k) {
        const int32_t c = cells[k0 + k];
        const int64_t n = cp[(size_t)c + 1] - cp[(size_t)c];
        const double* t = out.data() + (size_t)off[k] * 2;
        double llk0 = 0, llk2 = 0;  // :126
        for (int64_t i = 0; i < n; ++i) {
          llk0 += log(t[2 * i]);      // :147
          llk2 += log(t[2 * i + 1]);  // :148
        }
        l0[c] = llk0;
        l2[c] = llk2;
      }
    };
    const int nt = tot > (1 << 18) ? (int)std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (nt <= 1) {
      sum_cells(0, nc);
    } else {  // shares of about equal numbers of entries
      std::vector<std::thread> th;
      size_t a = 0;
      for (int t = 0; t < nt; ++t) {
        const int64_t want = tot * (t + 1) / nt;
        size_t b = a;
        while (b < nc && (b + 1 == nc ? tot : off[b + 1]) <= want) ++b;
        if (t == nt - 1) b = nc;
        th.emplace_back(sum_cells, a, b);
        a = b;
      }
      for (auto& x : th) x.join();
    }
    k0 = k1;
  }
  return 0;
}

// l0 / l2: [C] sums as the device made them; on return the cells whose place in the reference's order could depend on the
// last bits hold the reference's own sums.  *n_exact: how many cells that were.  exact_sums(cells): fills l0 / l2 of the
// listed cells (ascending) with the reference's sums; 0 = ok.
template <class F>
int settle_with(int64_t C, double* l0, double* l2, int64_t* n_exact, std::string* err, F&& exact_sums) {
  *n_exact = 0;
  if (C < 2) return 0;
  for (int64_t i = 0; i < C; ++i)
    if (!std::isfinite(l0[i]) || !std::isfinite(l2[i])) return 0;  // (no strict weak order to reproduce: std::sort's whim)
  std::vector<int32_t> ord((size_t)C);
  std::vector<uint8_t> exact((size_t)C, 0);
  auto before = [&](int32_t a, int32_t b) {  // sc_drop_seq.h:193-197
    const double cmp = (l2[a] - l0[a]) - (l2[b] - l0[b]);
    if (cmp != 0) return cmp > 0;
    return a > b;
  };
  for (int round = 0; round < 8; ++round) {
    for (int64_t i = 0; i < C; ++i) ord[(size_t)i] = (int32_t)i;
    std::sort(ord.begin(), ord.end(), before);
    std::vector<int32_t> todo;
    for (int64_t k = 0; k + 1 < C; ++k) {
      const int32_t a = ord[(size_t)k], b = ord[(size_t)k + 1];
      if (exact[a] && exact[b]) continue;
      const double mag = fmax(fmax(1.0, fmax(fabs(l0[a]), fabs(l2[a]))), fmax(fabs(l0[b]), fabs(l2[b])));
      if ((l2[a] - l0[a]) - (l2[b] - l0[b]) > 1e-9 * mag) continue;
      if (!exact[a]) {
        exact[a] = 1;
        todo.push_back(a);
      }
      if (!exact[b]) {
        exact[b] = 1;
        todo.push_back(b);
      }
    }
    if (todo.empty()) return 0;
    std::sort(todo.begin(), todo.end());
    if (exact_sums(todo)) return 1;
    *n_exact += (int64_t)todo.size();
  }
  *err = "muxgl_fmx_prepare (exact scores): the set of near-tied scores did not close in 8 rounds";
  return 1;
}

// host copy of a handle's cell_ptr
inline int fetch_cell_ptr(muxgl_handle* h, std::vector<int64_t>* cp) {
  cp->resize((size_t)h->C + 1);
  const hipError_t e = hipMemcpy(cp->data(), h->d_cell_ptr, sizeof(int64_t) * (size_t)(h->C + 1), hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    h->err = std::string("muxgl_fmx_prepare (exact scores): ") + hipGetErrorString(e);
    return 1;
  }
  return 0;
}

// one handle holding the whole pileup
inline int settle(muxgl_handle* h, double* l0, double* l2, int64_t* n_exact) {
  std::vector<int64_t> cp;
  return settle_with(h->C, l0, l2, n_exact, &h->err, [&](const std::vector<int32_t>& cells) -> int {
    if (cp.empty() && fetch_cell_ptr(h, &cp)) return 1;
    return compute(h, cells, cp, l0, l2);
  });
}

}  // namespace score_exact
