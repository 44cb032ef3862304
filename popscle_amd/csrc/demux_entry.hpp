// demux_entry.hpp -- the per-entry doublet-genotype likelihoods pG of demuxlet (cmd_cram_demuxlet.cpp:655-725) in the
// formulation the row kernel and the pG table pass of the wave kernels share.
#pragma once
#include "common.hpp"

// cmd_cram_demuxlet.cpp:655-725 for one entry.  lut: [0,128) phred2Err, [128,256) phred2Mat, [256,384) Err/3
// (muxgl_create); first4 = the entry's first four read bytes (byte k = read k), the rest is read from `reads`.
template <int NA>
__device__ __forceinline__ void row_entry_pg(const uint8_t* __restrict__ reads, int64_t r0, int64_t r1, uint32_t first4,
                                             const double* __restrict__ alpha, const double* lut, double (&pG)[NA * 9]) {
  // The reference multiplies every element by pR*(1-p) + pA*p with p = 0.5*l + (m-l)*0.5*alpha (:673,685) and divides
  // by the running maximum after every read (:692-699).  Here:
  //   * the factor is evaluated as A_l + B_m with A_l = pR + d*0.5*l*(1-alpha), B_m = d*0.5*m*alpha, d = pA - pR for
  //     l < 2 and as pA - B_(2-m) for l = 2 (algebraically the same number; 9 instead of 27 FP64 operations per alpha
  //     and read);
  //   * the common rescaling only guards against underflow, it cancels in the final q/q_max -- it is applied every
  //     32 reads instead of every read;
  //   * the tail  (q/q_max + 1e-10) / (1 + 1e-10)  (:703-725) is one FMA per element.
#pragma unroll
  for (int i = 0; i < NA * 9; ++i) pG[i] = 1.0;
  int since_norm = 0;
  for (int64_t r = r0; r < r1; ++r) {
    const int64_t kk = r - r0;
    const uint32_t b = (kk < 4) ? ((first4 >> (8 * (int)kk)) & 0xffu) : (uint32_t)reads[r];  // first 4 come prefetched
    if (b == MUXGL_READ_OTHER) continue;  // :664
    const uint32_t al = b >> 7, bq = b & 0x7f;
    const double e3 = lut[256 + bq], mt = lut[128 + bq];  // Err/3.0, Mat
    const double pR = (al == 0) ? mt : e3;  // :666
    const double pA = (al == 0) ? e3 : mt;  // :667
    const double d = pA - pR;
#pragma unroll
    for (int n = 0; n < NA; ++n) {
      // rows l = 0, 1 are built up from pR, row l = 2 down from pA: the homozygous corners p = 0 and p = 1 then carry
      // pR and pA themselves.  (pR + d) instead of pA would lose the small one of the two against the rounding of the
      // large one -- 6e-7 relative at base quality 93 -- and that corner is all a sample homozygous for the other
      // allele sees of the read.)
      const double ha = 0.5 * alpha[n], hb = 0.5 - ha;
      const double A1 = fma(d, hb, pR);
      const double B1 = d * ha, B2 = d * (ha + ha);
      double* q = &pG[n * 9];
      q[0] *= pR;
      q[1] *= pR + B1;
      q[2] *= pR + B2;
      q[3] *= A1;
      q[4] *= A1 + B1;
      q[5] *= A1 + B2;
      q[6] *= pA - B2;
      q[7] *= pA - B1;
      q[8] *= pA;
    }
    if (++since_norm == 32) {
      since_norm = 0;
      double mx = 0.0;
#pragma unroll
      for (int i = 0; i < NA * 9; ++i) mx = fmax(mx, pG[i]);
      const double inv = 1.0 / mx;
#pragma unroll
      for (int i = 0; i < NA * 9; ++i) pG[i] *= inv;
    }
  }
  double mx = 0.0;
#pragma unroll
  for (int i = 0; i < NA * 9; ++i) mx = fmax(mx, pG[i]);
  const double c = 1.0 / (1.0 + 1e-10);
  const double s = c / mx, t = 1e-10 * c;
#pragma unroll
  for (int i = 0; i < NA * 9; ++i) pG[i] = fma(pG[i], s, t);
}

