// demux_call16.hip -- evidence sums, best/next scans and the SNG/DBL/AMB call (cmd_cram_demuxlet.cpp:788-991) for
// V <= 16, sixteen lanes per cell instead of one.
//
// The reference walks the V*V*A hypotheses of a cell sequentially.  demux_call_kernel (demux_kernels.hip) keeps that
// walk, one lane per cell, which leaves a 10 k-cell batch with 157 waves of ~270 serial logAdd's each (0.1 ms, as long
// as a fifth of the whole sweep).  Here lane j of a 16-lane group owns row j of llksAB:
//   * scans: the reference's update rule (strict <, first-seen wins; :827-837, :883-906) leaves the two largest
//     hypotheses under the total order (value descending, scan position ascending); that order is associative, so each
//     lane scans its row and a 4-step butterfly merges the sixteen top-2 lists -- the result is the reference's, tie
//     for tie.
//   * evidence (:804-821): each lane chains logAdd over its row in the reference's order, the sixteen row sums are then
//     chained in row order.  Same terms, same operator, a different association of a sum whose terms differ by
//     hundreds of log units: the result moves by <= 1 ulp (the bar is 1e-5).
#include "common.hpp"

namespace {

struct call_alpha {
  double a[MUXGL_MAX_ALPHA];
};

struct top2 {
  double bv, nv;  // best / next value
  int32_t bp, np; // scan positions (-1: none); position encodes the hypothesis
};

// reference update rule for one more element at a later position
__device__ __forceinline__ void top2_push(top2& t, double v, int32_t pos) {
  if (t.bv < v) {
    t.nv = t.bv;
    t.np = t.bp;
    t.bv = v;
    t.bp = pos;
  } else if (t.nv < v) {
    t.nv = v;
    t.np = pos;
  }
}

// key order: value descending, then position ascending; "none" entries (pos < 0) carry -1e300 and never win
__device__ __forceinline__ bool key_before(double va, int32_t pa, double vb, int32_t pb) {
  if (pb < 0) return true;
  if (pa < 0) return false;
  if (va > vb) return true;
  if (va < vb) return false;
  return pa < pb;
}

__device__ __forceinline__ top2 top2_merge(const top2& a, const top2& b) {
  top2 r;
  if (key_before(a.bv, a.bp, b.bv, b.bp)) {
    r.bv = a.bv;
    r.bp = a.bp;
    if (key_before(a.nv, a.np, b.bv, b.bp)) {
      r.nv = a.nv;
      r.np = a.np;
    } else {
      r.nv = b.bv;
      r.np = b.bp;
    }
  } else {
    r.bv = b.bv;
    r.bp = b.bp;
    if (key_before(b.nv, b.np, a.bv, a.bp)) {
      r.nv = b.nv;
      r.np = b.np;
    } else {
      r.nv = a.bv;
      r.np = a.bp;
    }
  }
  return r;
}

__device__ __forceinline__ top2 top2_xor(const top2& t, int m) {
  top2 o;
  o.bv = __shfl_xor(t.bv, m, 64);
  o.nv = __shfl_xor(t.nv, m, 64);
  o.bp = __shfl_xor(t.bp, m, 64);
  o.np = __shfl_xor(t.np, m, 64);
  return o;
}

template <int G>  // lanes per cell: 16 (V <= 16) or 64 (V <= 64)
__global__ void __launch_bounds__(64)
    demux_callg_kernel(int64_t C, const int64_t* __restrict__ cell_ptr, int nv, int nAlpha, call_alpha al,
                        double doublet_prior, const double* __restrict__ ll, muxgl_demux_cell* __restrict__ out) {
  const int lane = threadIdx.x;
  const int j = lane & (G - 1);
  const int64_t i = (int64_t)blockIdx.x * (64 / G) + lane / G;
  const bool cell_ok = i < C;
  const bool live = cell_ok && j < nv;
  const double* gridAlpha = al.a;
  const double log_single_prior = log((1.0 - doublet_prior) / nv);
  const double log_doublet_prior1 = log(doublet_prior / nv / (nv - 1.) / (nAlpha - 1.));
  const double log_doublet_prior2 = log(doublet_prior / nv / (nv - 1.) / (nAlpha - 1.) * 2);

  top2 sng = {-1e300, -1e300, -1, -1}, dbl = {-1e300, -1e300, -1, -1};
  double rowsum = 0.0, sterm = 0.0;
  bool have = false;
  if (live) {
    const double* row = ll + ((size_t)i * nv + j) * nv * nAlpha;
    const double s = row[0];  // llksAB[j][0][0]
    top2_push(sng, s, j);
    sterm = s + log_single_prior;
    rowsum = sterm;
    have = true;
    for (int k = 0; k < nv; ++k) {
      if (k == j) continue;
      for (int n = 1; n < nAlpha; ++n) {
        const double v = row[k * nAlpha + n];
        if (gridAlpha[n] == 0.5) {
          if (k < j) rowsum = dev_logadd(rowsum, v + log_doublet_prior2);  // :812-815
        } else {
          rowsum = dev_logadd(rowsum, v + log_doublet_prior1);
        }
        top2_push(dbl, v, (j * nv + k) * nAlpha + n);
      }
    }
  }
  // merge the sixteen rows
#pragma unroll
  for (int m = 1; m < G; m <<= 1) {
    sng = top2_merge(sng, top2_xor(sng, m));
    dbl = top2_merge(dbl, top2_xor(dbl, m));
  }
  double sumLLK = -1e-300, sngLLK = -1e-300;  // :791 (sic)
  const int base = lane & ~(G - 1);
  for (int t = 0; t < G; ++t) {
    const double rs = __shfl(rowsum, base + t, 64);
    const double st = __shfl(sterm, base + t, 64);
    const int hv = __shfl((int)have, base + t, 64);
    if (hv) {
      sumLLK = dev_logadd(sumLLK, rs);
      sngLLK = dev_logadd(sngLLK, st);
    }
  }
  if (!cell_ok || j != 0) return;

  muxgl_demux_cell o;
  memset(&o, 0, sizeof(o));
  o.nsnps = (int32_t)(cell_ptr[i + 1] - cell_ptr[i]);
  if (o.nsnps == 0) {  // :653
    out[i] = o;
    return;
  }
  o.valid = 1;
  const int32_t sBest = sng.bp, sNext = sng.np;
  const double sngBestLLK = sng.bv, sngNextLLK = sng.nv;
  const double dblBestLLK = dbl.bv, dblNextLLK = dbl.nv;
  int32_t dBest1 = -1, dBest2 = -1, dblBestAlpha = -1, dNext1 = -1, dNext2 = -1, dblNextAlpha = -1;
  if (dbl.bp >= 0) {
    dblBestAlpha = dbl.bp % nAlpha;
    dBest2 = (dbl.bp / nAlpha) % nv;
    dBest1 = dbl.bp / (nAlpha * nv);
  }
  if (dbl.np >= 0) {
    dblNextAlpha = dbl.np % nAlpha;
    dNext2 = (dbl.np / nAlpha) % nv;
    dNext1 = dbl.np / (nAlpha * nv);
  }
  int32_t bestType, nextType, jBest, kBest, jNext, kNext, alphaBest, alphaNext;
  double bestLLK, nextLLK, bestPP;
  if (dblBestLLK > sngBestLLK + 2) {  // :925
    bestType = MUXGL_DBL;
    bestPP = exp(dblBestLLK + ((gridAlpha[dblBestAlpha] == 0.5) ? log_doublet_prior2 : log_doublet_prior1) - sumLLK);
    jBest = dBest1;
    kBest = dBest2;
    bestLLK = dblBestLLK;
    alphaBest = dblBestAlpha;
    if (dblNextLLK > sngBestLLK + 2) {
      nextType = MUXGL_DBL;
      jNext = dNext1;
      kNext = dNext2;
      nextLLK = dblNextLLK;
      alphaNext = dblNextAlpha;
    } else {
      nextType = MUXGL_SNG;
      jNext = kNext = sBest;
      nextLLK = sngBestLLK;
      alphaNext = 0;
    }
  } else {
    bestType = (sngBestLLK > sngNextLLK + 2) ? MUXGL_SNG : MUXGL_AMB;  // :947 / :968
    bestPP = sngBestLLK + log_single_prior - sumLLK;                   // log value, as the reference (:949,970)
    jBest = kBest = sBest;
    bestLLK = sngBestLLK;
    alphaBest = 0;
    if (dblBestLLK > sngNextLLK + 2) {
      nextType = MUXGL_DBL;
      jNext = dBest1;
      kNext = dBest2;
      nextLLK = dblBestLLK;
      alphaNext = dblBestAlpha;
    } else {
      nextType = MUXGL_SNG;
      jNext = kNext = sNext;
      nextLLK = sngNextLLK;
      alphaNext = 0;
    }
  }
  o.type = bestType;
  o.next_type = nextType;
  o.sBest = sBest;
  o.sNext = sNext;
  o.dBest1 = dBest1;
  o.dBest2 = dBest2;
  o.dBestA = dblBestAlpha;
  o.dNext1 = dNext1;
  o.dNext2 = dNext2;
  o.dNextA = dblNextAlpha;
  o.jBest = jBest;
  o.kBest = kBest;
  o.aBest = alphaBest;
  o.jNext = jNext;
  o.kNext = kNext;
  o.aNext = alphaNext;
  o.sngBestLLK = sngBestLLK;
  o.sngNextLLK = sngNextLLK;
  o.dblBestLLK = dblBestLLK;
  o.dblNextLLK = dblNextLLK;
  o.sumLLK = sumLLK;
  o.sngLLK = sngLLK;
  o.bestLLK = bestLLK;
  o.nextLLK = nextLLK;
  o.bestPP = bestPP;
  o.sngPP = exp(sngLLK - sumLLK);                             // :990
  o.sngOnlyPP = exp(sngBestLLK + log_single_prior - sngLLK);  // :991
  out[i] = o;
}

}  // namespace

int demux_call16_launch(muxgl_handle* h, const muxgl_demux_params* p) {
  call_alpha al;
  for (int i = 0; i < MUXGL_MAX_ALPHA; ++i) al.a[i] = (i < p->n_alpha) ? p->alpha[i] : 0.0;
  if (h->V <= 16) {
    const unsigned blocks = (unsigned)((h->C + 3) / 4);
    hipLaunchKernelGGL(demux_callg_kernel<16>, dim3(blocks ? blocks : 1), dim3(64), 0, h->stream, h->C, h->d_cell_ptr,
                       h->V, p->n_alpha, al, p->doublet_prior, h->d_ll, h->d_dcells);
  } else {
    const unsigned blocks = (unsigned)h->C;
    hipLaunchKernelGGL(demux_callg_kernel<64>, dim3(blocks ? blocks : 1), dim3(64), 0, h->stream, h->C, h->d_cell_ptr,
                       h->V, p->n_alpha, al, p->doublet_prior, h->d_ll, h->d_dcells);
  }
  HIPCHK(h, hipGetLastError());
  return 0;
}
