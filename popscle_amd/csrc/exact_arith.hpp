// exact_arith.hpp -- the reference's per-entry and per-merge arithmetic as device functions that reproduce it BIT FOR BIT:
// IEEE double multiplications, additions and (true) divisions in the reference's operation order, nothing contracted or
// re-associated.  Shared by the exact paths (greedy_exact.hpp: near ties of the greedy start; fmx_exact.hip: near-tie
// calls of an EM iteration).  The fast kernels do NOT use these: they work with reciprocal multiplies and other
// associations, equal to ~1e-13 relative.
#pragma once
#include "common.hpp"

namespace exact_arith {

// calculate_snp_droplet_pileup(ssd, sdp, 0.5), sc_drop_seq.cpp:452-509 (logdenom is never read outside the struct)
__device__ inline void entry_pileup(const uint8_t* __restrict__ reads, int64_t r0, int64_t r1,
                                    const double* __restrict__ lut /* [0,128) Err, [128,256) Mat */, double* gls) {
#pragma clang fp contract(off)
  for (int i = 0; i < 9; ++i) gls[i] = 1.0;
  for (int64_t r = r0; r < r1; ++r) {
    const uint8_t b = reads[r];
    if (b == MUXGL_READ_OTHER) continue;  // al > 1 (:470)
    const int al = b >> 7, bq = b & 0x7f;
    const double mat = lut[128 + bq], e4 = lut[bq] / 4.;
    // (al == 0 ? x : y) with alpha = 0.5 (:482-490): 1, 1-a/2, 1-a, (1+a)/2, .5, (1-a)/2, a, a/2, 0 and the complements
    const double f0[9] = {1.0, 0.75, 0.5, 0.75, 0.5, 0.25, 0.5, 0.25, 0.0};
    const double f1[9] = {0.0, 0.25, 0.5, 0.25, 0.5, 0.75, 0.5, 0.75, 1.0};
    for (int i = 0; i < 9; ++i) gls[i] *= (mat * (al == 0 ? f0[i] : f1[i]) + e4);
    double tmp = 0;
    for (int i = 0; i < 9; ++i) tmp += gls[i];
    for (int i = 0; i < 9; ++i) gls[i] /= tmp;
  }
  for (int i = 0; i < 9; ++i)
    if (gls[i] < 1e-6) gls[i] = 1e-6;
  double tmp = 0;
  for (int i = 0; i < 9; ++i) tmp += gls[i];
  for (int i = 0; i < 9; ++i) gls[i] /= tmp;
}

// snp_droplet_pileup::merge, sc_drop_seq.h:77-101 (the likelihoods only)
__device__ inline void merge(double* gls, const double* o) {
#pragma clang fp contract(off)
  for (int i = 0; i < 9; ++i) gls[i] *= o[i];
  double tmp = 0;
  for (int i = 0; i < 9; ++i) tmp += gls[i];
  for (int i = 0; i < 9; ++i) gls[i] /= tmp;
  for (int i = 0; i < 9; ++i)
    if (gls[i] < 1e-6) gls[i] = 1e-6;
  tmp = 0;
  for (int i = 0; i < 9; ++i) tmp += gls[i];
  for (int i = 0; i < 9; ++i) gls[i] /= tmp;
}

}  // namespace exact_arith
