// demux_call_body.hpp -- the per-cell call (evidence sums, best/next scans, SNG/DBL/AMB decision;
// cmd_cram_demuxlet.cpp:788-991) as a device function over G lanes, shared by demux_callg_kernel (LL tensor in HBM)
// and the fused finish kernel of the quad path (LL tile in LDS).  See demux_call16.hip for the argument why the
// lane-parallel scans and evidence chains give the reference's result.
#pragma once
#include "common.hpp"

namespace muxgl_call {

// the alpha grid and the three log priors of cmd_cram_demuxlet.cpp:793-795, which depend on the run's parameters only:
// taken on the host (glibc's log, the reference's own) instead of once per lane
struct call_alpha {
  double a[MUXGL_MAX_ALPHA];
  double log_single_prior, log_doublet_prior1, log_doublet_prior2;
};
inline call_alpha make_call_alpha(const muxgl_demux_params* p, int nv) {
  call_alpha al;
  for (int i = 0; i < MUXGL_MAX_ALPHA; ++i) al.a[i] = (i < p->n_alpha) ? p->alpha[i] : 0.0;
  const double dp = p->doublet_prior, nA = p->n_alpha;
  al.log_single_prior = log((1.0 - dp) / nv);
  al.log_doublet_prior1 = log(dp / nv / (nv - 1.) / (nA - 1.));
  al.log_doublet_prior2 = log(dp / nv / (nv - 1.) / (nA - 1.) * 2);
  return al;
}

// exp(x) for x <= 0, as the evidence sums use it (terms relative to their maximum): n = rint(x log2 e),
// r = x - n ln 2 in two pieces, a degree-13 Taylor polynomial on |r| <= 0.347 (truncation 4e-18) and ldexp -- about 20
// instructions against the library exp's 250 (it has no special cases to serve here); within 2 ulp.
// Domain: x <= 0 by construction (callers subtract the maximum); a NaN (-inf minus -inf when every term is -inf)
// is passed through as the library exp would, anything below -708 is 0, and a positive x is treated as 0.
__device__ __forceinline__ double exp_nonpos(double x) {
  if (!(x >= -708.0)) return x != x ? x : 0.0;
  x = fmin(x, 0.0);
  const double n = rint(x * 1.4426950408889634074);
  double r = fma(-n, 6.93147180369123816490e-01, x);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

// A lane's partner in step M of a butterfly over a wave.  The merges below are symmetric in their two arguments and every
// step starts with the merged value uniform over the group it ended with, so any pairing of the two halves does -- and
// within a row of sixteen lanes a pairing that a DPP modifier can express costs one move per dword (quad permutations for
// M = 1, 2; the mirrors of eight and of sixteen lanes for M = 4, 8) where a shuffle is a trip through the LDS crossbar
// (ds_bpermute: an address VGPR, ~100 cycles of latency each, and the finish kernel is nothing but such latencies).
template <int M>
__device__ __forceinline__ int32_t lane_partner(int32_t x) {
  if constexpr (M == 1) return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);        // quad_perm [1, 0, 3, 2]
  else if constexpr (M == 2) return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);   // quad_perm [2, 3, 0, 1]
  else if constexpr (M == 4) return __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true);  // row_half_mirror: i <-> 7 - i
  else if constexpr (M == 8) return __builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true);  // row_mirror: i <-> 15 - i
  else return __shfl_xor(x, M, 64);
}
template <int M>
__device__ __forceinline__ double lane_partner(double x) {
  if constexpr (M >= 16) return __shfl_xor(x, M, 64);
  return __hiloint2double(lane_partner<M>(__double2hiint(x)), lane_partner<M>(__double2loint(x)));
}

struct top2 {
  double bv, nv;  // best / next value
  int32_t bp, np; // scan positions (-1: none); position encodes the hypothesis
  double tv;  // third-largest VALUE of the scan (no position): tells the exact-call pass (host/exact_calls.hpp)
                       // whether best and next are the only hypotheses within rounding reach of each other
};

// reference update rule for one more element at a later position (selects, no branches: the scans are lock-step loops
// over sixteen lanes' rows, and a divergent branch per element serialises them)
template <bool T3 = true>  // T3: keep the third-largest value along (false: the caller finds it in a pass of its own)
__device__ __forceinline__ void top2_push(top2& t, double v, int32_t pos) {
  const bool b = t.bv < v;
  const bool n = !b && t.nv < v;
  if (T3) t.tv = (b || n) ? t.nv : fmax(t.tv, v);
  t.nv = b ? t.bv : (n ? v : t.nv);
  t.np = b ? t.bp : (n ? pos : t.np);
  t.bv = b ? v : t.bv;
  t.bp = b ? pos : t.bp;
}

// key order: value descending, then position ascending; "none" entries (pos < 0) carry -1e300 and never win
__device__ __forceinline__ bool key_before(double va, int32_t pa, double vb, int32_t pb) {
  if (pb < 0) return true;
  if (pa < 0) return false;
  if (va > vb) return true;
  if (va < vb) return false;
  return pa < pb;
}

template <bool T3 = true>
__device__ __forceinline__ top2 top2_merge(const top2& a, const top2& b) {
  // the best of the two lists, then the better of (the winner's runner-up, the loser's best) -- as selects
  const bool ab = key_before(a.bv, a.bp, b.bv, b.bp);
  const double wv = ab ? a.bv : b.bv, wn = ab ? a.nv : b.nv, lv = ab ? b.bv : a.bv;
  const int32_t wp = ab ? a.bp : b.bp, wnp = ab ? a.np : b.np, lp = ab ? b.bp : a.bp;
  const bool keep = key_before(wn, wnp, lv, lp);
  top2 r;
  r.bv = wv;
  r.bp = wp;
  r.nv = keep ? wn : lv;
  r.np = keep ? wnp : lp;
  // third value of the union of two descending triples x, y: max(x3, y3, min(x2, y1), min(x1, y2))
  r.tv = T3 ? fmax(fmax(a.tv, b.tv), fmax(fmin(a.nv, b.bv), fmin(a.bv, b.nv))) : -1e300;
  return r;
}

template <int M, bool T3 = true>
__device__ __forceinline__ top2 top2_partner(const top2& t) {
  top2 o;
  o.bv = lane_partner<M>(t.bv);
  o.nv = lane_partner<M>(t.nv);
  o.bp = lane_partner<M>(t.bp);
  o.np = lane_partner<M>(t.np);
  o.tv = T3 ? lane_partner<M>(t.tv) : -1e300;
  return o;
}

// order-independent insertion (the scans above rely on ascending positions; this one does not)
__device__ __forceinline__ void top2_insert(top2& t, double v, int32_t pos) {
  if (key_before(v, pos, t.bv, t.bp)) {
    t.tv = t.nv;
    t.nv = t.bv;
    t.np = t.bp;
    t.bv = v;
    t.bp = pos;
  } else if (key_before(v, pos, t.nv, t.np)) {
    t.tv = t.nv;
    t.nv = v;
    t.np = pos;
  } else {
    t.tv = fmax(t.tv, v);
  }
}

// A cell's scans after the merge over its lanes: top-two lists, the evidence as (largest term M, sum S of the terms
// relative to M), the same for the singlet terms alone.
struct call_partial {
  top2 sng, dbl;
  double M, S, Ms, Ss;
};

// merges the per-lane row results (top-2 lists, the largest evidence term rowmax of the lane's rows, their evidence sum
// racc relative to rowmax, the largest singlet term sterm and the singlet sum sacc relative to it) over the G lanes of
// the cell; every lane of the group ends with the cell's partial
template <int G, bool T3 = true>
__device__ __forceinline__ call_partial demux_call_merge(top2 sng, top2 dbl, double sterm, double rowmax, double racc,
                                                         double sacc) {
  const double NEG_INF = -__builtin_huge_val();
  double M = rowmax, Ms = sterm;
  constexpr int STEPS = G <= 1 ? 0 : (G <= 2 ? 1 : (G <= 4 ? 2 : (G <= 8 ? 3 : (G <= 16 ? 4 : (G <= 32 ? 5 : 6)))));
  static_assert((1 << STEPS) == G, "groups of 2^n lanes");
  wave_for<0, STEPS>([&](auto sc) {
    constexpr int m = 1 << decltype(sc)::value;
    sng = top2_merge<T3>(sng, top2_partner<m, T3>(sng));
    dbl = top2_merge<T3>(dbl, top2_partner<m, T3>(dbl));
    M = fmax(M, lane_partner<m>(M));
    Ms = fmax(Ms, lane_partner<m>(Ms));
  });
  double S = (racc > 0.0) ? racc * exp_nonpos(rowmax - M) : 0.0;
  double Ss = (sterm > NEG_INF) ? sacc * exp_nonpos(sterm - Ms) : 0.0;
  wave_for<0, STEPS>([&](auto sc) {
    constexpr int m = 1 << decltype(sc)::value;
    S += lane_partner<m>(S);
    Ss += lane_partner<m>(Ss);
  });
  return call_partial{sng, dbl, M, S, Ms, Ss};
}

// Second half of the call, ONE lane per cell: the evidence sums and the decision of cmd_cram_demuxlet.cpp:921-991.  (Its
// nine exponentials and logarithms are library calls of ~150 instructions each: callers that hold several cells give
// them to consecutive lanes of one wave instead of to lane 0 of a wave per cell.)
__device__ __forceinline__ void demux_call_decide(const call_partial& c, int32_t nsnps, int nv, int nAlpha,
                                                  const call_alpha& al, muxgl_demux_cell* out) {
  const double* gridAlpha = al.a;
  const double log_single_prior = al.log_single_prior;
  const double log_doublet_prior1 = al.log_doublet_prior1;
  const double log_doublet_prior2 = al.log_doublet_prior2;
  const top2& sng = c.sng;
  const top2& dbl = c.dbl;
  // :791 (sic): the reference starts both sums at -1e-300, i.e. with one more term exp(-1e-300)
  // (This lane's chain of transcendental calls is the finish kernel's critical path -- one lock-step round of workgroups, a
  //  handful of lanes each -- so it uses the short forms: exp_nonpos for the non-positive arguments all of them have,
  //  pos_log for the positive ones; ~25 instructions each against the library's 150-250, within 2 ulp, on quantities
  //  whose bar is 1e-5.)
  auto logadd = [](double la, double lb) {  // sc_drop_seq.cpp:5-8
    const double hi = la > lb ? la : lb, lo = la > lb ? lb : la;
    return hi + pos_log(1.0 + exp_nonpos(lo - hi), 0.0);
  };
  // posteriors the writer prints with %.2lg / %.5lf: the library exp only where the short form would flush a denormal
  auto pp_exp = [](double x) { return x < -700.0 ? exp(x) : exp_nonpos(x); };
  double sumLLK = -1e-300, sngLLK = -1e-300;
  if (c.S > 0.0) sumLLK = logadd(sumLLK, c.M + pos_log(c.S, 0.0));
  if (c.Ss > 0.0) sngLLK = logadd(sngLLK, c.Ms + pos_log(c.Ss, 0.0));

  muxgl_demux_cell o;
  memset(&o, 0, sizeof(o));
  o.nsnps = nsnps;
  if (o.nsnps == 0) {  // :653
    *out = o;
    return;
  }
  o.valid = 1;
  const int32_t sBest = sng.bp, sNext = sng.np;
  const double sngBestLLK = sng.bv, sngNextLLK = sng.nv;
  const double dblBestLLK = dbl.bv;
  int32_t dBest1 = -1, dBest2 = -1, dblBestAlpha = -1, dNext1 = -1, dNext2 = -1, dblNextAlpha = -1;
  if (dbl.bp >= 0) {
    dblBestAlpha = dbl.bp % nAlpha;
    dBest2 = (dbl.bp / nAlpha) % nv;
    dBest1 = dbl.bp / (nAlpha * nv);
  }
  // The scans list an alpha = 0.5 pair ONCE, as (lo, hi): its mirror (hi, lo) has the same likelihood (the reference's two
  // evaluations differ by rounding noise only, cmd_cram_demuxlet.cpp:738-746), so where such a pair is the best doublet
  // the runner-up is its mirror, as the reference's scan finds it.  Which of the two orders the reference names first is
  // decided by that noise: host/exact_calls.hpp recomputes it.  `third` = the best hypothesis that is neither of them.
  double dblNextLLK = dbl.nv, dblThird = dbl.tv;
  if (dbl.bp >= 0 && gridAlpha[dblBestAlpha] == 0.5) {
    dNext1 = dBest2;
    dNext2 = dBest1;
    dblNextAlpha = dblBestAlpha;
    dblNextLLK = dbl.bv;
    dblThird = dbl.nv;
  } else if (dbl.np >= 0) {
    dblNextAlpha = dbl.np % nAlpha;
    dNext2 = (dbl.np / nAlpha) % nv;
    dNext1 = dbl.np / (nAlpha * nv);
  }
  int32_t bestType, nextType, jBest, kBest, jNext, kNext, alphaBest, alphaNext;
  double bestLLK, nextLLK, bestPP;
  if (dblBestLLK > sngBestLLK + 2) {  // :925
    bestType = MUXGL_DBL;
    bestPP = pp_exp(dblBestLLK + ((gridAlpha[dblBestAlpha] == 0.5) ? log_doublet_prior2 : log_doublet_prior1) - sumLLK);
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
  o.sngPP = pp_exp(sngLLK - sumLLK);                             // :990
  o.sngOnlyPP = pp_exp(sngBestLLK + log_single_prior - sngLLK);  // :991
  {  // three or more hypotheses of a scan within rounding reach of each other: the exact-call pass must look at all of them
    double mag = 1.0;
    if (sngBestLLK > -1e299) mag = fmax(mag, fabs(sngBestLLK));
    if (sngNextLLK > -1e299) mag = fmax(mag, fabs(sngNextLLK));
    if (dblBestLLK > -1e299) mag = fmax(mag, fabs(dblBestLLK));
    if (dblNextLLK > -1e299) mag = fmax(mag, fabs(dblNextLLK));
    const double eps = 1e-9 * mag;
    if (sngNextLLK > -1e299 && sng.tv > -1e299 && sngNextLLK - sng.tv <= eps) o.valid |= MUXGL_CELL_DEEP_SNG;
    if (dblNextLLK > -1e299 && dblThird > -1e299 && dblNextLLK - dblThird <= eps) o.valid |= MUXGL_CELL_DEEP_DBL;
  }
  *out = o;
}

template <int G>
__device__ __forceinline__ void demux_call_finish(int lane, bool cell_ok, int32_t nsnps, int nv, int nAlpha,
                                                  const call_alpha& al, double doublet_prior, top2 sng, top2 dbl,
                                                  double sterm, double rowmax, double racc, muxgl_demux_cell* out,
                                                  double sacc = 1.0) {
  const call_partial c = demux_call_merge<G>(sng, dbl, sterm, rowmax, racc, sacc);
  if (!cell_ok || (lane & (G - 1)) != 0) return;
  demux_call_decide(c, nsnps, nv, nAlpha, al, out);
}



// lane = lane id in the wave; the G * Q lanes [base, base + G Q), base = lane & ~(G Q - 1), work on one cell: lane
// base + q G + j takes the rows j, j + G, ... and of each row the columns [q KQ, (q + 1) KQ), KQ = ceil(nv / Q) (Q = 1: whole
// rows).  ll_cell points at that cell's [nv][nv][nAlpha] hypotheses (global or LDS); lane base + 0 writes *out when
// cell_ok.  (Q = 4 at nv <= 16: a cell's scan is a chain of dependent selects and exponentials per lane; with sixteen
// lanes per cell and one calling wave per four-cell workgroup the oct path's finish kernel spent 35 of its 69 us in
// it -- measured by compiling the scans out -- whatever the number of cells.)
// demux_call_scan: the scans and their merge -- every lane of the group returns the cell's partial; demux_call_group: the
// scans and the decision by the group's first lane.
template <int G, int Q = 1>
// (ld: distance of two rows in doubles, nv * nAlpha unless the caller pads its tile -- the lanes of a group read the same
//  column of their rows at the same time, and rows a multiple of 64 dwords apart all sit in one LDS bank)
__device__ __forceinline__ call_partial demux_call_scan(int lane, bool cell_ok, int nv, int nAlpha, const call_alpha& al,
                                                        const double* ll_cell, int ld = 0) {
  if (ld == 0) ld = nv * nAlpha;
  const int j = lane & (G - 1);
  const int q = (lane / G) & (Q - 1);
  const int kq = (nv + Q - 1) / Q, k0 = q * kq, k1 = (k0 + kq < nv) ? k0 + kq : nv;
  const bool live = cell_ok && j < nv;
  const double* gridAlpha = al.a;
  const double log_single_prior = al.log_single_prior;
  const double log_doublet_prior1 = al.log_doublet_prior1;
  const double log_doublet_prior2 = al.log_doublet_prior2;

  top2 sng = {-1e300, -1e300, -1, -1, -1e300}, dbl = {-1e300, -1e300, -1, -1, -1e300};
  const double NEG_INF = -__builtin_huge_val();
  double sterm = NEG_INF, rowmax = NEG_INF, racc = 0.0, sacc = 0.0;
  if (live) {
    // a lane takes the rows j, j + G, ..: ascending rows are ascending scan positions, so the reference's update rule
    // (top2_push) applies across them as it does inside a row (and inside a lane's range of columns)
    // pass 1: scans, and the largest evidence term of the lane's rows
    for (int jr = j; jr < nv; jr += G) {
      const double* row = ll_cell + (size_t)jr * ld;
      if (q == 0) {
        const double s = row[0];  // llksAB[j][0][0]
        top2_push<!(G == 16 && Q == 4)>(sng, s, jr);
        const double st = s + log_single_prior;
        sterm = fmax(sterm, st);
        rowmax = fmax(rowmax, st);
      }
      for (int k = k0; k < k1; ++k) {
        if (k == jr) continue;
        for (int n = 1; n < nAlpha; ++n) {
          const double v = row[k * nAlpha + n];
          const bool sym = gridAlpha[n] == 0.5;
          if (sym) {
            if (k < jr) rowmax = fmax(rowmax, v + log_doublet_prior2);  // :812-815
          } else {
            rowmax = fmax(rowmax, v + log_doublet_prior1);
          }
          // (hi, lo) of an alpha = 0.5 pair is listed as (lo, hi), see demux_call_decide: pushed as "nothing" -- a select, not
          // a branch around the push: the lanes of a wave differ in jr and k, and a divergent branch serialises the loop
          top2_push<!(G == 16 && Q == 4)>(dbl, (sym && k < jr) ? -1e300 : v, (jr * nv + k) * nAlpha + n);
        }
      }
    }
    // pass 2: the evidence terms relative to that maximum (independent exp's instead of a logAdd chain)
    if (rowmax > NEG_INF) {
      for (int jr = j; jr < nv; jr += G) {
        const double* row = ll_cell + (size_t)jr * ld;
        if (q == 0) {
          const double st = row[0] + log_single_prior;
          racc += exp_nonpos(st - rowmax);
          sacc += exp_nonpos(st - sterm);
        }
        for (int k = k0; k < k1; ++k) {
          if (k == jr) continue;
          for (int n = 1; n < nAlpha; ++n) {
            const double v = row[k * nAlpha + n];
            if (gridAlpha[n] == 0.5) {
              if (k < jr) racc += exp_nonpos(v + log_doublet_prior2 - rowmax);
            } else {
              racc += exp_nonpos(v + log_doublet_prior1 - rowmax);
            }
          }
        }
      }
    }
  }
  // The third-largest value of each scan (what host/exact_calls.hpp needs to know whether best and next are the only
  // contenders).  Sixteen lanes x four column ranges (a handful of elements per lane, the tile in LDS): cheaper found
  // after the merge -- the largest element that is neither best nor next, one max-butterfly per scan -- than carried
  // through every update and every step of the two merges (the finish kernel of the oct path is one lock-step round of
  // workgroups: every instruction of it is on the step's critical path).
  constexpr bool T3 = !(G == 16 && Q == 4);
  call_partial cp = demux_call_merge<G * Q, T3>(sng, dbl, sterm, rowmax, racc, sacc);
  if (!T3) {
    double s3 = -1e300, d3 = -1e300;
    if (live) {
      for (int jr = j; jr < nv; jr += G) {
        const double* row = ll_cell + (size_t)jr * ld;
        s3 = fmax(s3, (q == 0 && jr != cp.sng.bp && jr != cp.sng.np) ? row[0] : -1e300);
        for (int k = k0; k < k1; ++k) {
          for (int n = 1; n < nAlpha; ++n) {
            const int32_t pos = (jr * nv + k) * nAlpha + n;
            const bool out = k == jr || (gridAlpha[n] == 0.5 && k < jr) || pos == cp.dbl.bp || pos == cp.dbl.np;
            d3 = fmax(d3, out ? -1e300 : row[k * nAlpha + n]);
          }
        }
      }
    }
    wave_for<0, 6>([&](auto sc) {
      constexpr int m = 1 << decltype(sc)::value;
      s3 = fmax(s3, lane_partner<m>(s3));
      d3 = fmax(d3, lane_partner<m>(d3));
    });
    cp.sng.tv = s3;
    cp.dbl.tv = d3;
  }
  return cp;
}

template <int G, int Q = 1>
__device__ __forceinline__ void demux_call_group(int lane, bool cell_ok, int32_t nsnps, int nv, int nAlpha,
                                                 const call_alpha& al, double doublet_prior, const double* ll_cell,
                                                 muxgl_demux_cell* out, int ld = 0) {
  const call_partial c = demux_call_scan<G, Q>(lane, cell_ok, nv, nAlpha, al, ll_cell, ld);
  if (cell_ok && (lane & (G * Q - 1)) == 0) demux_call_decide(c, nsnps, nv, nAlpha, al, out);
}

}  // namespace muxgl_call
