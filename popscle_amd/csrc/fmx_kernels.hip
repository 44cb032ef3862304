// fmx_kernels.hip -- freemuxlet hot path on gfx950.
//
// Reference being replaced (statgen/popscle): calculate_snp_droplet_pileup (sc_drop_seq.cpp:452-509), the per-cell
// singlet scores (cmd_cram_freemux2.cpp:117-160), the cluster pileup build (:277-288), and one EM iteration
// (:375-597): E-step pair likelihoods (:383-456), scans (:458-513), re-assignment (:515-584) and the ordered,
// clamped M-step (snp_droplet_pileup::merge, sc_drop_seq.h:77-101, driven by :590-596).
//
// Decomposition (new design; the reference is one thread walking std::maps):
//   fmx_entry_kernel      lane <-> entry: 9 genotype-pair likelihoods + counts + log lk0/lk2 of the entry
//   fmx_cell_score_kernel wave <-> cell: sums of the entry logs
//   fmx_cgp_kernel        lane <-> (SNP, cluster): cluster genotype posterior row used by the E-step
//   fmx_estep_pair_kernel workgroup <-> cell, lane <-> cluster pair: the plain E-step (the shaped ones live in fmx_oct.hip,
//                         fmx_row2.hip, fmx_wave.hip)
//   fmx_call_kernel       scans, evidence, re-assignment, change counters: lane <-> cell (K <= 24) or wave <-> cell
//   fmx_mstep_kernel      lane <-> (SNP, cluster): walks the SNP's entries in ascending cell id (SNP-major view) and
//                         applies merge() for the cells assigned to the cluster -- the exact sequential order of the
//                         reference, parallel over the S*K independent chains
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "demux_call_body.hpp"
#include "score_exact.hpp"

namespace {

constexpr double kMinNormGL = 1e-6;  // sc_drop_seq.h:14

// 1/x by v_rcp_f64 and two Newton steps: within 1 ulp of the division at a fifth of its instructions
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  return fma(r, fma(-x, r, 1.0), r);
}

// ------------------------------------------------------------------------------------------------ b1 / b2

__global__ void __launch_bounds__(256)
    fmx_entry_kernel(int64_t nnz, const int64_t* __restrict__ entry_rptr, const uint8_t* __restrict__ reads,
                     const int32_t* __restrict__ entry_snp, const double* __restrict__ af,
                     const double* __restrict__ lut_g, double* __restrict__ egls, double* __restrict__ egls6,
                     int32_t* __restrict__ ecnt, double* __restrict__ l0, double* __restrict__ l2,
                     uint32_t* __restrict__ flin) {
  __shared__ double lut[256];
  lut[threadIdx.x] = lut_g[threadIdx.x];
  __syncthreads();
  // (the trip count is wave-uniform: the ballot below needs every lane of the wave, also those beyond the last entry)
  for (int64_t eb = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); eb < nnz; eb += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = eb + (threadIdx.x & 63);
    const bool in = e < nnz;
    bool linear = false;
    if (in) {
    double gls[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) gls[i] = 1.0;
    int32_t nreads = 0, nref = 0, nalt = 0;
    const int64_t r1 = entry_rptr[e + 1];
    for (int64_t r = entry_rptr[e]; r < r1; ++r) {
      const uint32_t b = reads[r];
      ++nreads;                               // sc_drop_seq.cpp:466
      if (b == MUXGL_READ_OTHER) continue;    // :468
      const uint32_t al = b >> 7, bq = b & 0x7f;
      if (al == 0) ++nref;
      else ++nalt;
      const double mat = lut[128 + bq], e4 = lut[bq] / 4.;
      // fraction of reference reads expected for genotype pair (g1,g2) at alpha = 0.5: 1 - (g1+g2)/4 (:472-491)
      const double fr[9] = {1.0, .75, .5, .75, .5, .25, .5, .25, 0.0};
      double tmp = 0.0;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double f = (al == 0) ? fr[i] : fr[8 - i];
        gls[i] *= (mat * f + e4);
        tmp += gls[i];
      }
      const double inv = 1.0 / tmp;
#pragma unroll
      for (int i = 0; i < 9; ++i) gls[i] *= inv;
    }
    double tmp = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      if (gls[i] < kMinNormGL) gls[i] = kMinNormGL;  // :498-501
      tmp += gls[i];
    }
    const double inv = 1.0 / tmp;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      gls[i] *= inv;
      egls[(size_t)e * 9 + i] = gls[i];
    }
    // the matrix is symmetric (fraction 1-(g1+g2)/4 depends on g1+g2 only): six distinct values for the quad E-step
    if (egls6) {  // (a column slab has no E-step: no six-value copy, no scores)
      egls6[(size_t)e * 6 + 0] = gls[0];
      egls6[(size_t)e * 6 + 1] = gls[4];
      egls6[(size_t)e * 6 + 2] = gls[8];
      egls6[(size_t)e * 6 + 3] = gls[1];
      egls6[(size_t)e * 6 + 4] = gls[2];
      egls6[(size_t)e * 6 + 5] = gls[5];
    }
    ecnt[(size_t)e * 3 + 0] = nreads;
    ecnt[(size_t)e * 3 + 1] = nref;
    ecnt[(size_t)e * 3 + 2] = nalt;
    // With at most one usable read and no clamp the nine likelihoods are f(g1 + g2) with f LINEAR (one factor
    // mat * frac + e4 with frac = 1 - (g1+g2)/4, then a common scale): the E-step of such an entry is a two-term form
    // (fmx_wave.hip).  Checked on the values themselves, so whatever made them non-linear (a clamp at a high quality
    // cap, a second read) keeps the entry on the general path.
    {
      const double c0 = gls[0], c1 = gls[1] - gls[0];
      // a few roundings apart at most, measured at the SMALL end of the line (a value moved by the 1e-6 clamp must not
      // pass as a point of the line: 1e-14 of the maximum would be 1e-7 of a minimum that is 1e-7 of it)
      const double tol = fmin(1e-14 * fmax(gls[0], gls[8]), 1e-9 * fmin(gls[0], gls[8]));
      linear = nref + nalt <= 1 && fabs(gls[2] - fma(2.0, c1, c0)) <= tol && fabs(gls[5] - fma(3.0, c1, c0)) <= tol &&
               fabs(gls[8] - fma(4.0, c1, c0)) <= tol && gls[3] == gls[1] && gls[6] == gls[2] && gls[4] == gls[2] &&
               gls[7] == gls[5] &&
               // the moment forms reach the small end of the line through cancellation: relative error ~1e-16 * max / min
               fmin(gls[0], gls[8]) >= 1e-7 * fmax(gls[0], gls[8]);
    }
    if (l0) {
    // cmd_cram_freemux2.cpp:138-149
    const double a = af[entry_snp[e]];
    const double gps[3] = {(1.0 - a) * (1.0 - a), 2.0 * a * (1.0 - a), a * a};
    double lk0 = 0.0, lk2 = 0.0;
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) {
      lk2 += (gls[gi * 3 + gi] * gps[gi]);
#pragma unroll
      for (int gj = 0; gj < 3; ++gj) lk0 += (gls[gi * 3 + gj] * gps[gi] * gps[gj]);
    }
    l0[e] = log(lk0);
    l2[e] = log(lk2);
    }
    }
    if (flin) {
      const uint64_t m = __ballot(linear);
      if ((threadIdx.x & 63) == 0) flin[eb >> 5] = (uint32_t)m;
      if ((threadIdx.x & 63) == 32 && eb + 32 < nnz) flin[(eb >> 5) + 1] = (uint32_t)(m >> 32);
    }
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ void __launch_bounds__(64)
    fmx_cell_score_kernel(const int64_t* __restrict__ cell_ptr, const double* __restrict__ l0,
                          const double* __restrict__ l2, const int32_t* __restrict__ ecnt, double* __restrict__ llk0,
                          double* __restrict__ llk2, int32_t* __restrict__ nsnps, int32_t* __restrict__ nreads) {
  const int64_t c = blockIdx.x;
  const int64_t e0 = cell_ptr[c], e1 = cell_ptr[c + 1];
  double s0 = 0.0, s2 = 0.0, nr = 0.0;
  for (int64_t e = e0 + threadIdx.x; e < e1; e += 64) {
    s0 += l0[e];
    s2 += l2[e];
    nr += (double)ecnt[(size_t)e * 3];
  }
  s0 = wave_sum(s0);
  s2 = wave_sum(s2);
  nr = wave_sum(nr);
  if (threadIdx.x == 0) {
    llk0[c] = s0;
    llk2[c] = s2;
    nsnps[c] = (int32_t)(e1 - e0);
    nreads[c] = (int32_t)nr;
  }
}

// ------------------------------------------------------------------------------------------------ cluster GP

// cmd_cram_freemux2.cpp:402-415: gp = HWE(af) * diag(cluster gls), normalised, mixed with the prior by geno_error
__global__ void __launch_bounds__(256)
    fmx_cgp_kernel(int64_t S, int64_t s0, int64_t s1, int K, const double* __restrict__ af,
                   const double* __restrict__ cgls, double geno_error, double* __restrict__ cgp) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (s1 - s0) * K) return;
  const int64_t s = s0 + tid / K;
  const int k = (int)(tid % K);
  const double a = af[s];
  const double* g = cgls + ((size_t)k * S + s) * 9;
  const double p0 = (1.0 - a) * (1.0 - a), p1 = 2 * a * (1.0 - a), p2 = a * a;
  double g0 = p0 * g[0], g1 = p1 * g[4], g2 = p2 * g[8];
  const double sum = g0 + g1 + g2;
  g0 /= sum;
  g1 /= sum;
  g2 /= sum;
  if (geno_error > 0) {
    g0 = (1 - geno_error) * g0 + geno_error * p0;
    g1 = (1 - geno_error) * g1 + geno_error * p1;
    g2 = (1 - geno_error) * g2 + geno_error * p2;
  }
  double* o = cgp + ((size_t)s * K + k) * 3;
  o[0] = g0;
  o[1] = g1;
  o[2] = g2;
}

// ------------------------------------------------------------------------------------------------ E-step, any K

// workgroup <-> (cell, pair tile), lane <-> PPT pair slots; products with periodic renormalisation, one log at the end
template <int PPT>
__global__ void __launch_bounds__(256)
    fmx_estep_pair_kernel(const int64_t* __restrict__ cell_ptr, const int32_t* __restrict__ entry_snp,
                          const double* __restrict__ egls, const double* __restrict__ cgp, int K, int64_t c_off,
                          double* __restrict__ fll) {
  const int64_t c = c_off + blockIdx.x;
  const int64_t e0 = cell_ptr[c], e1 = cell_ptr[c + 1];
  const int npairs = K * (K + 1) / 2;
  const int T = blockDim.x;
  int pj[PPT], pk[PPT];
  double acc[PPT];
  int32_t ex[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int p = blockIdx.y * T * PPT + threadIdx.x + i * T;
    // invert p = j(j+1)/2 + k
    int j = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
    while ((j + 1) * (j + 2) / 2 <= p) ++j;
    while (j * (j + 1) / 2 > p) --j;
    pj[i] = j;
    pk[i] = p - j * (j + 1) / 2;
    acc[i] = 1.0;
    ex[i] = 0;
  }
  const int K3 = K * 3;
  int cnt = 0;
  for (int64_t e = e0; e < e1; ++e) {
    const double* gl = egls + (size_t)e * 9;
    const double* row = cgp + (size_t)entry_snp[e] * K3;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      if (blockIdx.y * T * PPT + threadIdx.x + i * T < npairs) {
        const double* a = row + pj[i] * 3;
        const double* b = row + pk[i] * 3;
        double lk;
        if (pj[i] == pk[i]) {
          lk = fma(gl[8], a[2], fma(gl[4], a[1], gl[0] * a[0]));
        } else {
          const double u0 = fma(a[2], gl[6], fma(a[1], gl[3], a[0] * gl[0]));
          const double u1 = fma(a[2], gl[7], fma(a[1], gl[4], a[0] * gl[1]));
          const double u2 = fma(a[2], gl[8], fma(a[1], gl[5], a[0] * gl[2]));
          lk = fma(b[2], u2, fma(b[1], u1, b[0] * u0));
        }
        acc[i] *= lk;
      }
    }
    if (++cnt == 16) {
      cnt = 0;
#pragma unroll
      for (int i = 0; i < PPT; ++i) prodacc_renorm(acc[i], ex[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int p = blockIdx.y * T * PPT + threadIdx.x + i * T;
    if (p < npairs) fll[(size_t)c * npairs + p] = prodacc_log(acc[i], ex[i]);
  }
}

// ------------------------------------------------------------------------------------------------ scans + re-assignment

// cmd_cram_freemux2.cpp:458-584 for one cell per lane; stat[0..2] = nsingle, namb, nchanged
// top two of a scan under the reference's update rule (strict >, first come first kept): the two largest under the
// total order (value descending, scan position ascending), which is associative -- lanes scan strided positions and
// merge their lists
struct fmx_top2 {
  double v1, v2;
  int32_t p1, p2;
  double v3;  // third-largest value (no position), for the exact-call pass: see muxgl_fmx_cell
};
__device__ __forceinline__ bool fmx_better(double va, int32_t pa, double vb, int32_t pb) {
  return va > vb || (va == vb && pa < pb);
}
__device__ __forceinline__ void fmx_top2_push(fmx_top2& t, double v, int32_t p) {
  if (fmx_better(v, p, t.v1, t.p1)) {
    t.v3 = t.v2;
    t.v2 = t.v1, t.p2 = t.p1;
    t.v1 = v, t.p1 = p;
  } else if (fmx_better(v, p, t.v2, t.p2)) {
    t.v3 = t.v2;
    t.v2 = v, t.p2 = p;
  } else {
    t.v3 = fmax(t.v3, v);
  }
}
__device__ __forceinline__ fmx_top2 fmx_top2_wave(fmx_top2 t) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    fmx_top2 o;
    o.v1 = __shfl_xor(t.v1, off, 64), o.p1 = __shfl_xor(t.p1, off, 64);
    o.v2 = __shfl_xor(t.v2, off, 64), o.p2 = __shfl_xor(t.p2, off, 64);
    o.v3 = __shfl_xor(t.v3, off, 64);
    fmx_top2_push(t, o.v1, o.p1);
    fmx_top2_push(t, o.v2, o.p2);
    t.v3 = fmax(t.v3, o.v3);  // (o.v1 >= o.v2 >= o.v3 are all in the union: o.v3 can be its third at best)
  }
  return t;
}
__device__ __forceinline__ double fmx_wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ double fmx_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// WAVE: one wave per cell -- the K (K + 1) / 2 log-likelihoods of a cell are read once, coalesced, by 64 lanes (one lane
// per cell reads them 16 KB apart from its neighbours' at K = 64: 0.7 TB/s); the evidence sums become max-shifted
// log-sum-exps (independent exps, one log) instead of the reference's serial logAdd chain -- the same value to ~1e-16
// relative.  !WAVE: one lane per cell, the reference's loop as written (kept behind MUXGL_FLAG_FORCE_TILE_SWEEP).
template <bool WAVE>
__global__ void __launch_bounds__(64)
    fmx_call_kernel(int64_t c0, int64_t c1, int K, double log_single_prior, double log_double_prior, const double* __restrict__ fll,
                    muxgl_fmx_cell* __restrict__ cells, int32_t* __restrict__ clust, int32_t* __restrict__ stat,
                    int32_t* __restrict__ prev_state, int32_t* __restrict__ flagged, const int32_t* __restrict__ xc_epoch,
                    const fmx_xc* __restrict__ xc, int32_t epoch) {
  const int64_t i = WAVE ? c0 + (int64_t)blockIdx.x : c0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c1) return;
  const int nSamples = K;
  const int npairs = K * (K + 1) / 2;
  // (log_single_prior, log_double_prior of :379-380: taken on the host with glibc's log, the reference's own, so that this
  //  kernel and the host side of the exact path, fmx_exact.hip, add the same two numbers)
  const double* llks = fll + (size_t)i * npairs;
  int32_t sBest = -1, sNext = -1, dBest1 = -1, dBest2 = -1, dNext1 = -1, dNext2 = -1;
  double sngBestLLK = -1e300, sngNextLLK = -1e300, dblBestLLK = -1e300, dblNextLLK = -1e300;
  double sumLLK = -1e300, sngLLK = -1e300;
  double sngThird = -1e300, dblThird = -1e300;
  if constexpr (WAVE) {
    const int lane = threadIdx.x;
    fmx_top2 ts = {-1e300, -1e300, 0x7fffffff, 0x7fffffff, -1e300}, td = ts;
    double mall = -1e300, msng = -1e300;
    // position p = j (j + 1) / 2 + k of the scan order; the row j of a position is followed through the strides
    int j = (int)((sqrt(8.0 * lane + 1.0) - 1.0) * 0.5);
    while ((j + 1) * (j + 2) / 2 <= lane) ++j;
    while (j * (j + 1) / 2 > lane) --j;
    int jj = j;
    for (int p = lane; p < npairs; p += 64) {
      while ((jj + 1) * (jj + 2) / 2 <= p) ++jj;
      const double v = llks[p];
      if (p - jj * (jj + 1) / 2 == jj) {
        fmx_top2_push(ts, v, p);
        msng = fmax(msng, v);
        mall = fmax(mall, v + log_single_prior);
      } else {
        fmx_top2_push(td, v, p);
        mall = fmax(mall, v + log_double_prior);
      }
    }
    ts = fmx_top2_wave(ts);
    td = fmx_top2_wave(td);
    mall = fmx_wave_max(mall);
    msng = fmx_wave_max(msng);
    double sall = 0.0, ssng = 0.0;
    jj = j;
    for (int p = lane; p < npairs; p += 64) {  // (second pass: the row is in the cache)
      while ((jj + 1) * (jj + 2) / 2 <= p) ++jj;
      const double v = llks[p];
      if (p - jj * (jj + 1) / 2 == jj) {
        sall += muxgl_call::exp_nonpos(v + log_single_prior - mall);  // (arguments <= 0: ~20 instructions against the
        ssng += muxgl_call::exp_nonpos(v - msng);                     //  library exp's 250; this loop was 8 of configs[4]'s 9 ms)
      } else {
        sall += muxgl_call::exp_nonpos(v + log_double_prior - mall);
      }
    }
    sall = fmx_wave_sum(sall);
    ssng = fmx_wave_sum(ssng);
    if (lane != 0) return;
    sumLLK = mall + log(sall);
    sngLLK = msng + log_single_prior + log(ssng);
    auto row_of = [](int p) {
      int r = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
      while ((r + 1) * (r + 2) / 2 <= p) ++r;
      while (r * (r + 1) / 2 > p) --r;
      return r;
    };
    if (ts.p1 != 0x7fffffff) sBest = row_of(ts.p1), sngBestLLK = ts.v1;
    if (ts.p2 != 0x7fffffff) sNext = row_of(ts.p2), sngNextLLK = ts.v2;
    if (td.p1 != 0x7fffffff) dBest1 = row_of(td.p1), dBest2 = td.p1 - dBest1 * (dBest1 + 1) / 2, dblBestLLK = td.v1;
    if (td.p2 != 0x7fffffff) dNext1 = row_of(td.p2), dNext2 = td.p2 - dNext1 * (dNext1 + 1) / 2, dblNextLLK = td.v2;
    sngThird = ts.v3;
    dblThird = td.v3;
  } else {
  // one lane per cell: the reference's scans as written (:469-497); its evidence chain -- logAdd per hypothesis, a library
  // exp and log each, 400 of them per cell at K = 16: nine tenths of this kernel's instructions -- as a running
  // (largest term, sum of the terms relative to it), rescaled when a larger term turns up: independent exponentials of
  // non-positive arguments and one log per sum, the same value to ~1e-16 relative (the sums start at -1e300 in the
  // reference, :466-467, which logAdd absorbs exactly)
  const double NEG_INF = -__builtin_huge_val();
  double mall = NEG_INF, sall = 0.0, msng = NEG_INF, ssng = 0.0;
  auto evid = [](double t, double& m, double& sacc) {
    if (t > m) {
      sacc = sacc * muxgl_call::exp_nonpos(m - t) + 1.0;
      m = t;
    } else {
      sacc += muxgl_call::exp_nonpos(t - m);
    }
  };
  for (int j = 0; j < nSamples; ++j) {
    for (int k = 0; k < j; ++k) {
      const double v = llks[j * (j + 1) / 2 + k];
      if (v > dblBestLLK) {
        dblThird = dblNextLLK;
        dNext1 = dBest1;
        dNext2 = dBest2;
        dblNextLLK = dblBestLLK;
        dBest1 = j;
        dBest2 = k;
        dblBestLLK = v;
      } else if (v > dblNextLLK) {
        dblThird = dblNextLLK;
        dNext1 = j;
        dNext2 = k;
        dblNextLLK = v;
      } else {
        dblThird = fmax(dblThird, v);
      }
      evid(v + log_double_prior, mall, sall);
    }
    const double v = llks[j * (j + 1) / 2 + j];
    if (v > sngBestLLK) {
      sngThird = sngNextLLK;
      sNext = sBest;
      sngNextLLK = sngBestLLK;
      sBest = j;
      sngBestLLK = v;
    } else if (v > sngNextLLK) {
      sngThird = sngNextLLK;
      sNext = j;
      sngNextLLK = v;
    } else {
      sngThird = fmax(sngThird, v);
    }
    evid(v + log_single_prior, mall, sall);
    evid(v + log_single_prior, msng, ssng);
  }
  if (sall > 0.0) sumLLK = mall + log(sall);
  if (ssng > 0.0) sngLLK = msng + log(ssng);
  }
  // (round 6) A decision whose margin is within rounding reach of the kernels' numbers -- best / next of a scan, next /
  // third, one of the four +2 thresholds -- is not this kernel's to make: the cell goes on the list fmx_exact.hip settles
  // in the reference's own arithmetic.  What that needs of the state BEFORE this iteration is kept aside (the previous
  // (type, jBest, kBest) of the nchanged rules below; the assignments the cluster pileups were built from: the launcher).
  // A cell settled in an earlier iteration whose inputs have not changed since (no assignment changed anywhere: same
  // `epoch`) takes the exact scan results from the table instead of being listed again: a converged job pays nothing.
  bool listed = false;
  double sngBestDev = -1e300;  // the kernels' own value where the table overrides it (sngOnlyPP stays what a listed cell keeps)
  bool from_table = false;
  if (prev_state) {
    double mag = 1.0;
    if (sngBestLLK > -1e299) mag = fmax(mag, fabs(sngBestLLK));
    if (dblBestLLK > -1e299) mag = fmax(mag, fabs(dblBestLLK));
    if (sngNextLLK > -1e299) mag = fmax(mag, fabs(sngNextLLK));
    if (dblNextLLK > -1e299) mag = fmax(mag, fabs(dblNextLLK));
    const double eps = 1e-9 * mag;
    auto near = [eps](double a, double b) { return a > -1e299 && b > -1e299 && fabs(a - b) <= eps; };
    if (near(sngBestLLK, sngNextLLK) || near(sngNextLLK, sngThird) || near(dblBestLLK, dblNextLLK) ||
        near(dblNextLLK, dblThird) || near(dblBestLLK, sngBestLLK + 2) || near(dblNextLLK, sngBestLLK + 2) ||
        near(sngBestLLK, sngNextLLK + 2) || near(dblBestLLK, sngNextLLK + 2)) {
      if (xc_epoch && xc_epoch[i] == epoch) {
        const fmx_xc e = xc[i];
        sngBestDev = sngBestLLK;
        from_table = true;
        sBest = e.sBest, sNext = e.sNext, dBest1 = e.dBest1, dBest2 = e.dBest2, dNext1 = e.dNext1, dNext2 = e.dNext2;
        sngBestLLK = e.sngBestLLK, sngNextLLK = e.sngNextLLK, dblBestLLK = e.dblBestLLK, dblNextLLK = e.dblNextLLK;
      } else {
        listed = true;
      }
    }
  }
  muxgl_fmx_cell c = cells[i];
  const int32_t prev_type = c.type, prev_j = c.jBest, prev_k = c.kBest;
  c.sBest = sBest;
  c.sngBestLLK = sngBestLLK;
  c.sNext = sNext;
  c.sngNextLLK = sngNextLLK;
  c.dBest1 = dBest1;
  c.dBest2 = dBest2;
  c.dblBestLLK = dblBestLLK;
  c.dNext1 = dNext1;
  c.dNext2 = dNext2;
  c.dblNextLLK = dblNextLLK;
  c.sngPP = exp(sngLLK - sumLLK);
  c.sngOnlyPP = exp((from_table ? sngBestDev : sngBestLLK) + log_single_prior - sngLLK);
  c.sumLLK = sumLLK;
  c.sngThirdLLK = sngThird;
  c.dblThirdLLK = dblThird;

  int32_t dsingle = 0, damb = 0, dchanged = 0;
  c.clust = -1;                           // :520
  if (dblBestLLK > sngBestLLK + 2) {      // :521
    if (c.type != 1) dchanged = 1;
    c.type = 1;
    c.bestPP = (dblBestLLK + log_double_prior - sumLLK);
    c.jBest = dBest1;
    c.kBest = dBest2;
    c.bestLLK = dblBestLLK;
    if (dblNextLLK > sngBestLLK + 2) {
      c.jNext = dNext1;
      c.kNext = dNext2;
      c.nextLLK = dblNextLLK;
    } else {
      c.jNext = c.kNext = sBest;
      c.nextLLK = sngBestLLK;
    }
  } else if (sngBestLLK > sngNextLLK + 2) {  // :542
    if ((c.type != 0) || (c.jBest != sBest) || (c.kBest != sBest)) dchanged = 1;
    c.type = 0;
    dsingle = 1;
    c.bestPP = (sngBestLLK + log_single_prior - sumLLK);
    c.jBest = c.kBest = sBest;
    c.bestLLK = sngBestLLK;
    c.clust = sBest;
    if (dblBestLLK > sngNextLLK + 2) {
      c.jNext = dBest1;
      c.kNext = dBest2;
      c.nextLLK = dblBestLLK;
    } else {
      c.jNext = c.kNext = sNext;
      c.nextLLK = sngNextLLK;
    }
  } else {  // :565
    if (c.type != 2) dchanged = 1;
    c.type = 2;
    damb = 1;
    c.bestPP = (sngBestLLK + log_single_prior - sumLLK);
    c.jBest = c.kBest = sBest;
    c.bestLLK = sngBestLLK;
    if (dblBestLLK > sngNextLLK + 2) {
      c.jNext = dBest1;
      c.kNext = dBest2;
      c.nextLLK = dblNextLLK;  // sic, :577
    } else {
      c.jNext = c.kNext = sNext;
      c.nextLLK = sngNextLLK;
    }
  }
  if (prev_state) prev_state[i] = (prev_type & 0xff) | ((prev_j & 0xff) << 8) | ((prev_k & 0xff) << 16);
  if (listed) flagged[atomicAdd(&stat[3], 1)] = (int32_t)i;
  cells[i] = c;
  clust[i] = c.clust;
  if (dsingle) atomicAdd(&stat[0], 1);
  if (damb) atomicAdd(&stat[1], 1);
  if (dchanged) atomicAdd(&stat[2], 1);
}

// ------------------------------------------------------------------------------------------------ M-step

// snp_droplet_pileup::merge (sc_drop_seq.h:77-101), applied in ascending cell id for every (cluster, SNP) chain.
// Division by the running sum is a multiplication by its reciprocal; logdenom is never read and is not kept.
__global__ void __launch_bounds__(256)
    fmx_mstep_kernel(int64_t S, int64_t s0, int64_t s1, int K, const int64_t* __restrict__ snp_ptr,
                     const int64_t* __restrict__ snp_entry,
                     const int32_t* __restrict__ entry_cell, const int32_t* __restrict__ clust,
                     const double* __restrict__ egls, const double* __restrict__ segls6,
                     const int32_t* __restrict__ ecnt, double* __restrict__ cgls, int32_t* __restrict__ ccnt) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (s1 - s0) * K) return;
  const int64_t s = s0 + tid / K;
  const int k = (int)(tid % K);
  double g[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) g[i] = 1.0;
  int32_t nreads = 0, nref = 0, nalt = 0;
  const int64_t p1 = snp_ptr[s + 1];
  for (int64_t p = snp_ptr[s]; p < p1; ++p) {
    const int64_t e = snp_entry[p];
    if (clust[entry_cell[e]] != k) continue;
    double o[9];
    if (egls) {
#pragma unroll
      for (int i = 0; i < 9; ++i) o[i] = egls[(size_t)e * 9 + i];
    } else {  // a column slab keeps the SNP-major six-value copy only (fmx_prepare_cols): the matrix is symmetric
      const double* a = segls6 + (size_t)p * 6;
      o[0] = a[0], o[4] = a[1], o[8] = a[2];
      o[1] = o[3] = a[3];
      o[2] = o[6] = a[4];
      o[5] = o[7] = a[5];
    }
    const int32_t* oc = ecnt + (size_t)e * 3;
    nreads += oc[0];
    nref += oc[1];
    nalt += oc[2];
    double tmp = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      g[i] *= o[i];
      tmp += g[i];
    }
    double inv = 1.0 / tmp;
    tmp = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      g[i] *= inv;
      if (g[i] < kMinNormGL) g[i] = kMinNormGL;
      tmp += g[i];
    }
    inv = 1.0 / tmp;
#pragma unroll
    for (int i = 0; i < 9; ++i) g[i] *= inv;
  }
  double* og = cgls + ((size_t)k * S + s) * 9;
#pragma unroll
  for (int i = 0; i < 9; ++i) og[i] = g[i];
  int32_t* oc = ccnt + ((size_t)k * S + s) * 3;
  oc[0] = nreads;
  oc[1] = nref;
  oc[2] = nalt;
}

// nreads / nref / nalt of the cluster pileups (the integer part of merge(), sc_drop_seq.h:78-80): order-independent
// sums over the entries of the cells currently assigned to a cluster, entry-parallel with integer atomics
__global__ void __launch_bounds__(256)
    fmx_counts_kernel(int64_t nnz, int64_t S, const int32_t* __restrict__ entry_snp,
                      const int32_t* __restrict__ entry_cell, const int32_t* __restrict__ clust,
                      const int32_t* __restrict__ ecnt, int32_t* __restrict__ ccnt) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t k = clust[entry_cell[e]];
    if (k < 0) continue;
    int32_t* oc = ccnt + ((size_t)k * S + entry_snp[e]) * 3;
    atomicAdd(oc, ecnt[(size_t)e * 3]);
    atomicAdd(oc + 1, ecnt[(size_t)e * 3 + 1]);
    atomicAdd(oc + 2, ecnt[(size_t)e * 3 + 2]);
  }
}

// Code of every element of the SNP-major view for the M-step's stream (fmx_mstep.hip): the byte of the entry's one usable
// read (0xFF: none -- a read byte is never 0xFF, that is the "other" allele) when it has at most one, 0x100 otherwise.
// "Other" reads do not enter the likelihoods (sc_drop_seq.cpp:470), so an entry's six values are then those of a one-read
// entry with that byte: row `code` of the table fmx_entry_kernel makes of the 256 one-read entries.
__global__ void __launch_bounds__(256)
    fmx_scode_kernel(int64_t nnz, const int64_t* __restrict__ snp_entry, const int64_t* __restrict__ entry_rptr,
                     const uint8_t* __restrict__ reads, uint16_t* __restrict__ scode) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = snp_entry[p];
    uint32_t code = 0xFF;
    int n = 0;
    for (int64_t r = entry_rptr[e], r1 = entry_rptr[e + 1]; r < r1 && n < 2; ++r) {
      const uint8_t b = reads[r];
      if (b == MUXGL_READ_OTHER) continue;
      code = b;
      ++n;
    }
    scode[p] = (uint16_t)(n <= 1 ? code : 0x100u);
  }
}

// one-time gather of the entry likelihoods and counts into SNP-major order
__global__ void __launch_bounds__(256)
    fmx_snp_major_kernel(int64_t nnz, const int64_t* __restrict__ snp_entry, const double* __restrict__ egls,
                         const int32_t* __restrict__ ecnt, double* __restrict__ segls, double* __restrict__ segls6,
                         int32_t* __restrict__ secnt) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = snp_entry[p];
    double g[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) g[i] = egls[(size_t)e * 9 + i];
    if (segls) {
#pragma unroll
      for (int i = 0; i < 9; ++i) segls[(size_t)p * 9 + i] = g[i];
    }
    // the six distinct values {00, 11, 22, 01, 02, 12} of the symmetric matrix: what the ordered M-step streams
    if (segls6) {
      double* o6 = segls6 + (size_t)p * 6;
      o6[0] = g[0], o6[1] = g[4], o6[2] = g[8], o6[3] = g[1], o6[4] = g[2], o6[5] = g[5];
    }
    if (secnt) {
#pragma unroll
      for (int i = 0; i < 3; ++i) secnt[(size_t)p * 3 + i] = ecnt[(size_t)e * 3 + i];
    }
  }
}

}  // namespace

static int fmx_mstep_launch(muxgl_handle* h) {
  const int64_t n = (h->fs1 - h->fs0) * h->K;
  if (n <= 0) return 0;
  if (!(h->flags & MUXGL_FLAG_FORCE_TILE_SWEEP)) {  // K <= 64: lane = chain, the SNP's list as a stream (fmx_mstep.hip)
    const int rc = fmx_mstep_stream_launch(h);
    if (rc >= 0) return rc;
  }
  // beyond 64 clusters (no BASELINE shape), or under MUXGL_FLAG_FORCE_TILE_SWEEP: lane = (SNP, cluster), every chain on its own
  hipLaunchKernelGGL(fmx_mstep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->S, h->fs0, h->fs1,
                     h->K, h->d_snp_ptr, h->d_snp_entry, h->d_entry_cell, h->d_clust, h->d_egls, h->d_segls6, h->d_ecnt,
                     h->d_cgls, h->d_ccnt);
  HIPCHK(h, hipGetLastError());
  return 0;
}

static int fmx_build_snp_major(muxgl_handle* h, host_timer& tm, bool keep_segls6 = false /* allocated by the caller */) {
  const int64_t nnz = h->nnz;
  if (plan_build_snp_major(h)) return 1;
  tm.lap("fmx_prepare: SNP-major view (device sort)");
  // (the nine-value copy and the counts are read by freemuxlet-old's pair kernel only: fmx_snp_major_full builds them on
  //  its first call -- 84 B per entry that a freemux2 run neither allocates nor writes: 40 GB at configs[4])
  dev_free(&h->d_segls);
  dev_free(&h->d_secnt);
  if (!keep_segls6 && dev_alloc(h, &h->d_segls6, (size_t)nnz * 6)) return 1;
  if (nnz) {
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(fmx_snp_major_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, nnz, h->d_snp_entry,
                       h->d_egls, h->d_ecnt, (double*)nullptr, h->d_segls6, (int32_t*)nullptr);
    HIPCHK(h, hipGetLastError());
    // the codes of the stream's elements and the table behind them: the entry kernel on 256 one-read entries, read byte
    // = entry number (entry 255's read is the "other" allele: no usable read)
    dev_free(&h->d_scode);
    dev_free(&h->d_mtab);
    static const bool no_codes = getenv("MUXGL_MSTEP_NO_CODES") != nullptr;  // (timing / tests: every row from the stream)
    if (!no_codes) {
      int64_t* d_rp = nullptr;
      uint8_t* d_rd = nullptr;
      double* d_g9 = nullptr;
      int32_t* d_cn = nullptr;
      std::vector<int64_t> rp(257);
      std::vector<uint8_t> rd(256);
      for (int i = 0; i < 257; ++i) rp[(size_t)i] = i;
      for (int i = 0; i < 256; ++i) rd[(size_t)i] = (uint8_t)i;
      int rc = dev_alloc(h, &h->d_scode, (size_t)nnz) || dev_alloc(h, &h->d_mtab, (size_t)256 * 6) || dev_alloc(h, &d_rp, 257) ||
               dev_alloc(h, &d_rd, 256) || dev_alloc(h, &d_g9, (size_t)256 * 9) || dev_alloc(h, &d_cn, (size_t)256 * 3);
      hipError_t e = hipSuccess;
      if (!rc) e = hipMemcpyAsync(d_rp, rp.data(), sizeof(int64_t) * 257, hipMemcpyHostToDevice, h->stream);
      if (!rc && e == hipSuccess) e = hipMemcpyAsync(d_rd, rd.data(), 256, hipMemcpyHostToDevice, h->stream);
      if (!rc && e == hipSuccess) {
        hipLaunchKernelGGL(fmx_entry_kernel, dim3(1), dim3(256), 0, h->stream, (int64_t)256, d_rp, d_rd, (const int32_t*)nullptr,
                           (const double*)nullptr, h->d_lut, d_g9, h->d_mtab, d_cn, (double*)nullptr, (double*)nullptr,
                           (uint32_t*)nullptr);
        hipLaunchKernelGGL(fmx_scode_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, nnz, h->d_snp_entry,
                           h->d_entry_rptr, h->d_reads, h->d_scode);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);  // (rp / rd are host vectors)
      dev_free(&d_rp);
      dev_free(&d_rd);
      dev_free(&d_g9);
      dev_free(&d_cn);
      if (rc) return 1;
      if (e != hipSuccess) MUXGL_FAIL(h, "M-step codes: %s", hipGetErrorString(e));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  tm.lap("fmx_prepare: SNP-major gather of likelihoods");
  return 0;
}

// SNP-major copies of the nine-value likelihoods and of the read counts (freemuxlet-old's pair kernel), on first use
int fmx_snp_major_full(muxgl_handle* h) {
  if (h->d_segls && h->d_secnt) return 0;
  const int64_t nnz = h->nnz;
  if (!h->d_egls || !h->d_snp_entry) MUXGL_FAIL(h, "the cell-major likelihoods or the SNP-major view are gone");
  if (dev_alloc(h, &h->d_segls, (size_t)nnz * 9)) return 1;
  if (dev_alloc(h, &h->d_secnt, (size_t)nnz * 3)) return 1;
  if (nnz) {
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(fmx_snp_major_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, nnz, h->d_snp_entry,
                       h->d_egls, h->d_ecnt, h->d_segls, (double*)nullptr, h->d_secnt);
    HIPCHK(h, hipGetLastError());
  }
  return 0;
}

// b1 for a column slab: entry likelihoods of its entries, straight into the SNP-major order the M-step streams; the
// cell-major copy is dropped again (nothing reads it: the E-step works on the row slab)
static int fmx_prepare_cols(muxgl_handle* h, const double* af) {
  HIPCHK(h, hipSetDevice(h->device));
  host_timer tm;
  const int64_t S = h->S, nnz = h->nnz;
  if (dev_alloc(h, &h->d_af, (size_t)S)) return 1;
  if (S) HIPCHK(h, hipMemcpyAsync(h->d_af, af, sizeof(double) * S, hipMemcpyHostToDevice, h->stream));
  if (dev_alloc(h, &h->d_egls, (size_t)nnz * 9)) return 1;
  if (dev_alloc(h, &h->d_ecnt, (size_t)nnz * 3)) return 1;
  if (nnz) {
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fmx_entry_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, nnz, h->d_entry_rptr,
                       h->d_reads, h->d_entry_snp, h->d_af, h->d_lut, h->d_egls, (double*)nullptr, h->d_ecnt,
                       (double*)nullptr, (double*)nullptr, (uint32_t*)nullptr);
    HIPCHK(h, hipGetLastError());
  }
  if (fmx_build_snp_major(h, tm)) return 1;
  dev_free(&h->d_egls);  // (both M-step kernels read the SNP-major six-value copy on a column slab)
  h->fmx_prepared = true;
  h->K = 0;
  h->fc0 = 0;
  h->fc1 = h->C;
  h->fs0 = h->slab_s0;
  h->fs1 = h->slab_s1;
  return 0;
}

extern "C" {

int muxgl_fmx_prepare(muxgl_handle* h, const double* af, double* cell_llk0, double* cell_llk2, int32_t* cell_nsnps,
                      int32_t* cell_nreads) {
  if (!h) return 1;
  if (h->group) return group_fmx_prepare(h, af, cell_llk0, cell_llk2, cell_nsnps, cell_nreads);
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->d_cell_ptr) MUXGL_FAIL(h, "muxgl_fmx_prepare: no pileup set (muxgl_set_pileup)");
  if (!af) MUXGL_FAIL(h, "muxgl_fmx_prepare: af is NULL");
  clear_timing(h);
  host_timer tm;
  const int64_t C = h->C, S = h->S, nnz = h->nnz;
  if (dev_alloc(h, &h->d_af, (size_t)S)) return 1;
  if (S) HIPCHK(h, hipMemcpyAsync(h->d_af, af, sizeof(double) * S, hipMemcpyHostToDevice, h->stream));
  if (dev_alloc(h, &h->d_egls, (size_t)nnz * 9)) return 1;
  if (dev_alloc(h, &h->d_ecnt, (size_t)nnz * 3)) return 1;
  if (dev_alloc(h, &h->d_flin, (size_t)((nnz + 31) / 32))) return 1;
  fmx_wave_streams_release(h);  // they are made of the likelihoods computed below
  double *d_l0 = nullptr, *d_l2 = nullptr, *d_c0 = nullptr, *d_c2 = nullptr;
  int32_t *d_ns = nullptr, *d_nr = nullptr;
  auto cleanup = [&]() {
    dev_free(&d_c0);
    dev_free(&d_c2);
    dev_free(&d_ns);
    dev_free(&d_nr);
  };
  // the two per-entry log-likelihoods live until the cells' scores are summed: in the buffer of the SNP-major six-value
  // copy, which is filled after that (16 B per entry less to allocate)
  if (dev_alloc(h, &h->d_segls6, (size_t)nnz * 6)) return 1;
  d_l0 = h->d_segls6;
  d_l2 = h->d_segls6 + nnz;
  if (dev_alloc(h, &d_c0, (size_t)C) ||
      dev_alloc(h, &d_c2, (size_t)C) || dev_alloc(h, &d_ns, (size_t)C) || dev_alloc(h, &d_nr, (size_t)C)) {
    cleanup();
    return 1;
  }
  tic(h, MUXGL_T_FMX_ENTRY);
  if (nnz) {
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fmx_entry_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, nnz, h->d_entry_rptr,
                       h->d_reads, h->d_entry_snp, h->d_af, h->d_lut, h->d_egls, (double*)nullptr, h->d_ecnt, d_l0, d_l2, h->d_flin);
  }
  if (C)
    hipLaunchKernelGGL(fmx_cell_score_kernel, dim3((unsigned)C), dim3(64), 0, h->stream, h->d_cell_ptr, d_l0, d_l2,
                       h->d_ecnt, d_c0, d_c2, d_ns, d_nr);
  toc(h, MUXGL_T_FMX_ENTRY);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess && C && cell_llk0) e = hipMemcpy(cell_llk0, d_c0, sizeof(double) * C, hipMemcpyDeviceToHost);
  if (e == hipSuccess && C && cell_llk2) e = hipMemcpy(cell_llk2, d_c2, sizeof(double) * C, hipMemcpyDeviceToHost);
  if (e == hipSuccess && C && cell_nsnps) e = hipMemcpy(cell_nsnps, d_ns, sizeof(int32_t) * C, hipMemcpyDeviceToHost);
  if (e == hipSuccess && C && cell_nreads) e = hipMemcpy(cell_nreads, d_nr, sizeof(int32_t) * C, hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) MUXGL_FAIL(h, "muxgl_fmx_prepare: %s", hipGetErrorString(e));
  collect_timing(h);
  tm.lap("fmx_prepare: entry kernels + scores D2H");
  // the cells whose place in the reference's order hangs on the last bits of their score: the reference's own sums
  h->fmx_exact_scores = 0;
  if (cell_llk0 && cell_llk2 && !h->col) {
    if (score_exact::settle(h, cell_llk0, cell_llk2, &h->fmx_exact_scores)) return 1;
    if (tm.on) fprintf(stderr, "[muxgl] fmx_prepare: %lld cells' scores recomputed exactly\n", (long long)h->fmx_exact_scores);
    tm.lap("fmx_prepare: near-tied scores in the reference's arithmetic");
  }

  if (h->col) {  // slabbed: the SNP-major side lives in the column slab
    if (fmx_prepare_cols(h->col, af)) {
      h->err = h->col->err;
      return 1;
    }
    dev_free(&h->d_segls);
    dev_free(&h->d_segls6);
    dev_free(&h->d_scode);
    dev_free(&h->d_mtab);
    dev_free(&h->d_secnt);
  } else if (fmx_build_snp_major(h, tm, true)) {
    return 1;
  }
  h->fmx_prepared = true;
  h->K = 0;
  h->fc0 = 0;
  h->fc1 = C;
  h->fs0 = h->col ? h->col->slab_s0 : 0;
  h->fs1 = h->col ? h->col->slab_s1 : S;
  demux_row_release(&h->frow);
  demux_row_release(&h->fqrow);
  return 0;
}

int muxgl_fmx_get_entry_gls(muxgl_handle* h, double* gls, int32_t* counts) {
  if (!h) return 1;
  if (h->group) return group_fmx_get_entry_gls(h, gls, counts);
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->fmx_prepared) MUXGL_FAIL(h, "muxgl_fmx_get_entry_gls: call muxgl_fmx_prepare first");
  if (gls && h->nnz) HIPCHK(h, hipMemcpy(gls, h->d_egls, sizeof(double) * 9 * h->nnz, hipMemcpyDeviceToHost));
  if (counts && h->nnz) HIPCHK(h, hipMemcpy(counts, h->d_ecnt, sizeof(int32_t) * 3 * h->nnz, hipMemcpyDeviceToHost));
  return 0;
}

// Attaches the column slab of a rank to a handle whose pileup (muxgl_set_pileup) is the rank's row slab.
int muxgl_fmx_set_column_slab(muxgl_handle* h, int64_t C_total, int64_t c0, int64_t s0, int64_t s1, int64_t nnz_s,
                              int64_t R_s, const int64_t* cell_ptr_s, const int32_t* entry_snp_s,
                              const int64_t* entry_rptr_s, const uint8_t* reads_s) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmx_set_column_slab");
  return fmx_attach_column_slab(h, C_total, c0, s0, s1, nnz_s, R_s, cell_ptr_s, entry_snp_s, entry_rptr_s, reads_s, false);
}

}  // extern "C"

int fmx_attach_column_slab(muxgl_handle* h, int64_t C_total, int64_t c0, int64_t s0, int64_t s1, int64_t nnz_s, int64_t R_s,
                           const int64_t* cell_ptr_s, const int32_t* entry_snp_s, const int64_t* entry_rptr_s,
                           const uint8_t* reads_s, bool trusted) {
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->d_cell_ptr || h->role == MUXGL_ROLE_COLS)
    MUXGL_FAIL(h, "muxgl_fmx_set_column_slab: hand the row slab over first (muxgl_set_pileup)");
  if (c0 < 0 || C_total < 0 || c0 + h->C > C_total)
    MUXGL_FAIL(h, "muxgl_fmx_set_column_slab: cells [%lld, %lld) do not fit %lld cells", (long long)c0,
               (long long)(c0 + h->C), (long long)C_total);
  if (s0 < 0 || s1 < s0 || s1 > h->S) MUXGL_FAIL(h, "muxgl_fmx_set_column_slab: bad SNP range");
  if (nnz_s < 0 || !cell_ptr_s || !entry_rptr_s || (nnz_s > 0 && !entry_snp_s))
    MUXGL_FAIL(h, "muxgl_fmx_set_column_slab: bad arrays");
  for (int64_t e = 0; e < nnz_s && !trusted; ++e)
    if (entry_snp_s[e] < s0 || entry_snp_s[e] >= s1)
      MUXGL_FAIL(h, "muxgl_fmx_set_column_slab: entry %lld has SNP id %d outside the slab's range [%lld, %lld)",
                 (long long)e, entry_snp_s[e], (long long)s0, (long long)s1);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->col) muxgl_destroy(h->col);
  h->col = nullptr;
  muxgl_handle* col = nullptr;
  if (muxgl_handle_create(h->device, h->flags, h->stream, &col, &h->err)) return 1;
  col->slab_s0 = s0;
  col->slab_s1 = s1;
  if (muxgl_set_pileup_role(col, MUXGL_ROLE_COLS, C_total, h->S, nnz_s, R_s, cell_ptr_s, entry_snp_s, entry_rptr_s,
                            reads_s, trusted)) {
    h->err = col->err;
    muxgl_destroy(col);
    return 1;
  }
  h->col = col;
  h->role = MUXGL_ROLE_ROWS;
  h->C_total = C_total;
  h->cell_base = c0;
  h->fmx_prepared = false;
  h->K = 0;
  return 0;
}

int fmx_cluster_counts_device(muxgl_handle* m) {
  const size_t n = (size_t)m->K * m->S;
  HIPCHK(m, hipMemsetAsync(m->d_ccnt, 0, sizeof(int32_t) * 3 * n, m->stream));
  if (m->nnz) {
    int64_t blocks = (m->nnz + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(fmx_counts_kernel, dim3((unsigned)blocks), dim3(256), 0, m->stream, m->nnz, m->S, m->d_entry_snp,
                       m->d_entry_cell, m->d_clust, m->d_ecnt, m->d_ccnt);
    HIPCHK(m, hipGetLastError());
  }
  return 0;
}

extern "C" {

int muxgl_fmx_set_clusters(muxgl_handle* h, int32_t K, const int32_t* clust) {
  if (!h) return 1;
  if (h->group) return group_fmx_set_clusters(h, K, clust);
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->fmx_prepared) MUXGL_FAIL(h, "muxgl_fmx_set_clusters: call muxgl_fmx_prepare first");
  if (K < 1 || K > 255) MUXGL_FAIL(h, "muxgl_fmx_set_clusters: K=%d outside [1,255]", K);
  const int64_t C = h->C, S = h->S;
  const int64_t CT = h->col ? h->C_total : C, cb = h->col ? h->cell_base : 0;  // clust[] spans the whole job
  if (!clust && CT) MUXGL_FAIL(h, "muxgl_fmx_set_clusters: clust is NULL");
  for (int64_t i = 0; i < CT; ++i)
    if (clust[i] >= K) MUXGL_FAIL(h, "muxgl_fmx_set_clusters: cell %lld has cluster %d >= K", (long long)i, clust[i]);
  h->K = K;
  muxgl_handle* m = h->col ? h->col : h;  // who holds the cluster pileups and runs the ordered merge
  m->K = K;
  if (dev_alloc(h, &m->d_cgls, (size_t)K * S * 9)) return 1;
  if (dev_alloc(h, &m->d_ccnt, (size_t)K * S * 3)) return 1;
  if (dev_alloc(h, &h->d_cgp, (size_t)(S + MUXGL_XCHG_PAD) * K * 3)) return 1;
  if (dev_alloc(h, &h->d_fll, (size_t)fmx_wave_fll_rows(h) * K * (K + 1) / 2)) return 1;
  if (h->col) {  // rows of SNPs outside the own range stay zero until a caller gathers them
    HIPCHK(h, hipMemsetAsync(m->d_cgls, 0, sizeof(double) * (size_t)K * S * 9, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_cgp, 0, sizeof(double) * (size_t)(S + MUXGL_XCHG_PAD) * K * 3, h->stream));
  }
  // state before the first iteration: cmd_cram_freemux2.cpp:191-194,213,244 and :345-367
  std::vector<int32_t> cl((size_t)CT);
  for (int64_t i = 0; i < CT; ++i) cl[(size_t)i] = clust[i] >= 0 ? clust[i] : -1;
  for (int64_t i = 0; i < C; ++i) {
    muxgl_fmx_cell& c = h->h_fcells[i];
    memset(&c, 0, sizeof(c));
    c.type = (cl[(size_t)(cb + i)] >= 0) ? 0 : -1;
    c.clust = cl[(size_t)(cb + i)];
    c.jBest = c.kBest = c.jNext = c.kNext = -1;
    c.sBest = c.sNext = c.dBest1 = c.dBest2 = c.dNext1 = c.dNext2 = -1;
    c.bestLLK = c.nextLLK = c.sngBestLLK = c.sngNextLLK = c.dblBestLLK = c.dblNextLLK = -1e300;
    c.bestPP = c.sngPP = c.sngOnlyPP = c.sumLLK = -1e300;
  }
  if (C) {
    HIPCHK(h, hipMemcpyAsync(h->d_fcells, h->h_fcells, sizeof(muxgl_fmx_cell) * C, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_clust, cl.data() + cb, sizeof(int32_t) * C, hipMemcpyHostToDevice, h->stream));
  }
  if (h->col && CT)
    HIPCHK(h, hipMemcpyAsync(m->d_clust, cl.data(), sizeof(int32_t) * CT, hipMemcpyHostToDevice, h->stream));
  if (dev_alloc(h, &m->d_prev_clust, (size_t)(CT ? CT : 1)) || dev_alloc(h, &h->d_prev_state, (size_t)(C ? C : 1)) ||
      dev_alloc(h, &h->d_flagged, (size_t)(C ? C : 1)) || dev_alloc(h, &h->d_xc_epoch, (size_t)(C ? C : 1)) ||
      dev_alloc(h, &h->d_xc, (size_t)(C ? C : 1)))
    return 1;
  HIPCHK(h, hipMemsetAsync(h->d_xc_epoch, 0, sizeof(int32_t) * (size_t)(C ? C : 1), h->stream));
  h->xs_epoch = 1;
  h->xs_keep = false;
  h->fmx_exact_cells = h->fmx_exact_changed = h->fmx_exact_unresolved = 0;
  clear_timing(h);
  tic(h, MUXGL_T_FMX_MSTEP);
  if (fmx_mstep_launch(m)) {  // :277-288: every assigned cell, ascending cell id
    if (m != h) h->err = m->err;
    return 1;
  }
  toc(h, MUXGL_T_FMX_MSTEP);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  collect_timing(h);
  return 0;
}

static int fmx_check_iter(muxgl_handle* h, const muxgl_fmx_params* p, const char* who) {
  if (!p) MUXGL_FAIL(h, "%s: params NULL", who);
  if (!h->fmx_prepared || h->K < 1) MUXGL_FAIL(h, "%s: call muxgl_fmx_prepare and muxgl_fmx_set_clusters first", who);
  return 0;
}

}  // extern "C"

// cluster genotype posteriors of the SNP shard [fs0,fs1) from the cluster pileups (cmd_cram_freemux2.cpp:402-415)
int fmx_phase_gp(muxgl_handle* h, const muxgl_fmx_params* p) {
  tic(h, MUXGL_T_FMX_GP);
  const muxgl_handle* m = h->col ? h->col : h;
  const int64_t n = (m->fs1 - m->fs0) * h->K;
  if (n > 0)
    hipLaunchKernelGGL(fmx_cgp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->S, m->fs0, m->fs1,
                       h->K, h->d_af, m->d_cgls, p->geno_error, h->d_cgp);
  toc(h, MUXGL_T_FMX_GP);
  HIPCHK(h, hipGetLastError());
  return 0;
}

// entries per chunk of the oct E-step (MUXGL_FMX_CH: tuning).  256 since the end of round 5: with the sweep moving bytes
// as fast as the fabric gives them, fewer chunk partials for the sweep to write and the reduce to read
// beat the shorter tail of 192-entry chunks (configs[3]: 2.05-2.08 -> 2.01-2.02 ms an iteration; 320 level, 384 +0.5 %,
// 128 +7 %); demuxlet's oct sweep keeps MUXGL_QUAD_CH
static int fmx_oct_chunk() {
  static const int ch = [] {
    const char* ev = getenv("MUXGL_FMX_CH");
    return ev && atoi(ev) >= 16 ? atoi(ev) / 4 * 4 : 256;
  }();
  return ch;
}

// E-step, scans and re-assignment of the cell shard [fc0,fc1) (:383-584); needs the whole cgp tensor
int fmx_phase_estep(muxgl_handle* h, const muxgl_fmx_params* p) {
  const int64_t c0 = h->fc0, c1 = h->fc1, nc = c1 - c0;
  const int K = h->K;
  const int npairs = K * (K + 1) / 2;
  tic(h, MUXGL_T_FMX_ESTEP);
  muxgl_row_state* st = h->frow ? h->frow : h->row;
  int qrc = -1;
  // (the chunk tables of the quad E-step are its own: demuxlet's oct kernel cuts cells into longer chunks, and a sharded
  //  run must cut a cell exactly as the whole-pileup run does)
  if (nc > 0 && K <= 16 && !h->fqrow && h->qrow && !(h->flags & (MUXGL_FLAG_FORCE_TILE_SWEEP | MUXGL_FLAG_FORCE_ROW_KERNEL)) &&
      demux_row_build(h, &h->fqrow, 0, h->C, fmx_oct_chunk()))
    return 1;
  if (nc > 0) qrc = fmx_oct_estep_launch(h, h->fqrow, c0, nc);  // K <= 16: eight lanes per entry
  if (nc > 0 && qrc < 0) qrc = fmx_row2_estep_launch(h, st, c0, nc);  // 16 < K <= 32: two clusters per lane
  if (nc > 0 && qrc < 0) qrc = fmx_wave_estep_launch(h, c0, nc);  // 32 < K: one wave per cell (part) and block
  if (qrc > 0) return 1;
  if (nc > 0 && qrc < 0) {  // the plain kernel: workgroup <-> (cell, tile of pairs).  K > 255 never happens (muxgl_fmx_set_clusters):
                            // reached under MUXGL_FLAG_FORCE_TILE_SWEEP / _FORCE_ROW_KERNEL, which tests use to cover it
    const int T = 256, PPT = 4;
    const unsigned tiles = (unsigned)((npairs + T * PPT - 1) / (T * PPT));
    hipLaunchKernelGGL(fmx_estep_pair_kernel<PPT>, dim3((unsigned)nc, tiles), dim3(T), 0, h->stream, h->d_cell_ptr,
                       h->d_entry_snp, h->d_egls, h->d_cgp, K, c0, h->d_fll);
  }
  toc(h, MUXGL_T_FMX_ESTEP);
  tic(h, MUXGL_T_FMX_CALL);
  HIPCHK(h, hipMemsetAsync(h->d_fstat, 0, 4 * sizeof(int32_t), h->stream));
  // The table of settled near-tie cells stays valid while the E-step's inputs do: no assignment changed in the previous
  // iteration (xs_keep: set by muxgl_fmx_iterate / the group from the counters, by muxgl_fmx_exact_hint for callers of the
  // phases) and the same parameters.
  if (!h->xs_keep || !h->xs_have_p || h->xs_p.doublet_prior != p->doublet_prior || h->xs_p.geno_error != p->geno_error) ++h->xs_epoch;
  h->xs_keep = false;
  h->xs_p = *p;
  h->xs_have_p = true;
  {  // the assignments the running iteration's cluster pileups were built from, for the exact path (fmx_exact.hip)
    muxgl_handle* m = h->col ? h->col : h;
    const int64_t CT = h->col ? h->C_total : h->C;
    if (m->d_prev_clust && CT)
      HIPCHK(h, hipMemcpyAsync(m->d_prev_clust, m->d_clust, sizeof(int32_t) * (size_t)CT, hipMemcpyDeviceToDevice, h->stream));
  }
  const double lsp = log((1.0 - p->doublet_prior) / K);          // cmd_cram_freemux2.cpp:379
  const double ldp = log(p->doublet_prior / K / (K - 1) * 2.0);  // :380
  if (nc > 0)
  {
    if ((h->flags & MUXGL_FLAG_FORCE_TILE_SWEEP) || K <= 24)  // (few hypotheses per cell: a wave per cell is mostly overhead,
                                                             //  0.54 against 0.13 ms at configs[3])
      hipLaunchKernelGGL(fmx_call_kernel<false>, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, h->stream, c0, c1, K,
                         lsp, ldp, h->d_fll, h->d_fcells, h->d_clust, h->d_fstat, h->d_prev_state, h->d_flagged,
                         h->d_xc_epoch, h->d_xc, h->xs_epoch);
    else
      hipLaunchKernelGGL(fmx_call_kernel<true>, dim3((unsigned)nc), dim3(64), 0, h->stream, c0, c1, K, lsp, ldp,
                         h->d_fll, h->d_fcells, h->d_clust, h->d_fstat, h->d_prev_state, h->d_flagged, h->d_xc_epoch, h->d_xc,
                         h->xs_epoch);
  }
  toc(h, MUXGL_T_FMX_CALL);
  HIPCHK(h, hipGetLastError());
  if (h->col && h->C)  // the new assignments of the own cells, at their place in the job-wide array the M-step reads
    HIPCHK(h, hipMemcpyAsync(h->col->d_clust + h->cell_base, h->d_clust, sizeof(int32_t) * (size_t)h->C,
                             hipMemcpyDeviceToDevice, h->stream));
  // the three counters, on their way to the host while the caller enqueues the next phase
  HIPCHK(h, hipMemcpyAsync(h->h_fstat, h->d_fstat, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_stat, h->stream));
  return 0;
}

// ordered clamped merge for the SNP shard [fs0,fs1) (:516-517 clear + :590-596); needs every cell's assignment
int fmx_phase_mstep(muxgl_handle* h) {
  tic(h, MUXGL_T_FMX_MSTEP);
  muxgl_handle* m = h->col ? h->col : h;
  if (fmx_mstep_launch(m)) {
    if (m != h) h->err = m->err;
    return 1;
  }
  toc(h, MUXGL_T_FMX_MSTEP);
  return 0;
}

extern "C" {

static int fmx_phase_fetch(muxgl_handle* h, muxgl_fmx_cell* out, int32_t* nsingle, int32_t* namb, int32_t* nchanged,
                           double* full_ll) {
  const int64_t C = h->C;
  const int npairs = h->K * (h->K + 1) / 2;
  // the per-cell records travel only when asked for: an EM loop needs the three counters per iteration and the records once
  if (!out && !full_ll) {
    HIPCHK(h, hipEventSynchronize(h->ev_stat));  // counters only: whatever was enqueued behind the E-step keeps running
  } else {
    if (C && out)
      HIPCHK(h, hipMemcpyAsync(h->h_fcells, h->d_fcells, sizeof(muxgl_fmx_cell) * C, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  if (out && C) memcpy(out, h->h_fcells, sizeof(muxgl_fmx_cell) * C);
  if (nsingle) *nsingle = h->h_fstat[0];
  if (namb) *namb = h->h_fstat[1];
  if (nchanged) *nchanged = h->h_fstat[2];
  // cells listed for the exact path that are still open (muxgl_fmx_iterate settled its own before it came here): counted as
  // unresolved until muxgl_fmx_exact_finish takes them off again
  h->fmx_listed = h->h_fstat[3];
  h->fmx_exact_unresolved += h->h_fstat[3];
  h->h_fstat[3] = 0;
  if (full_ll && C) HIPCHK(h, hipMemcpy(full_ll, h->d_fll, sizeof(double) * (size_t)C * npairs, hipMemcpyDeviceToHost));
  return 0;
}

int muxgl_fmx_iterate(muxgl_handle* h, const muxgl_fmx_params* p, muxgl_fmx_cell* out, int32_t* nsingle, int32_t* namb,
                      int32_t* nchanged, double* full_ll) {
  if (!h) return 1;
  if (h->group) return group_fmx_iterate(h, p, out, nsingle, namb, nchanged, full_ll);
  HIPCHK(h, hipSetDevice(h->device));
  if (fmx_check_iter(h, p, "muxgl_fmx_iterate")) return 1;
  if (h->col || h->fc0 != 0 || h->fc1 != h->C || h->fs0 != 0 || h->fs1 != h->S)
    MUXGL_FAIL(h, "muxgl_fmx_iterate: handle is sharded (muxgl_fmx_set_shard / muxgl_fmx_set_column_slab); use the "
                  "muxgl_fmx_iter_* phases");
  clear_timing(h);
  if (fmx_phase_gp(h, p) || fmx_phase_estep(h, p) || fmx_phase_mstep(h)) return 1;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->h_fstat[3] > 0) {  // calls within rounding reach: settled in the reference's arithmetic (fmx_exact.hip)
    bool reassigned = false;
    if (fmx_exact_resolve(h, p, &reassigned)) return 1;
    if (reassigned) {  // the ordered M-step again, from the corrected assignments
      if (fmx_phase_mstep(h)) return 1;
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
  }
  h->xs_keep = h->h_fstat[2] == 0;  // no assignment changed: the next E-step sees the same cluster pileups
  if (fmx_phase_fetch(h, out, nsingle, namb, nchanged, full_ll)) return 1;
  collect_timing(h);
  return 0;
}

// For callers of the phases: the job-wide number of changed cells of the iteration just finished (after the exact path).
// 0 keeps the table of settled near-tie cells valid for the next E-step; without this call every E-step starts afresh.
int muxgl_fmx_exact_hint(muxgl_handle* h, int32_t nchanged_jobwide) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmx_exact_hint");
  h->xs_keep = nchanged_jobwide == 0;
  return 0;
}

// The exact path for the sharded phases (see fmx_exact.hip; popscle_amd/freemuxlet.py run_em drives it): after
// muxgl_fmx_iter_estep + muxgl_fmx_iter_fetch, when the job-wide count of listed cells is not zero.
int muxgl_fmx_exact_snps(muxgl_handle* h, int32_t* out, int64_t cap, int64_t* n) {
  if (!h || !n) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmx_exact_snps");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (!h->xs_snps_valid) {
    if (h->fmx_listed > h->h_fstat[3]) h->h_fstat[3] = h->fmx_listed;  // (a muxgl_fmx_iter_fetch moved the count there)
    if (fmx_exact_snps(h, &h->xs_snps)) return 1;
    h->xs_snps_valid = true;
  }
  *n = (int64_t)h->xs_snps.size();
  if (out) {
    if (cap < *n) MUXGL_FAIL(h, "muxgl_fmx_exact_snps: room for %lld SNPs, %lld needed", (long long)cap, (long long)*n);
    memcpy(out, h->xs_snps.data(), sizeof(int32_t) * h->xs_snps.size());
    h->xs_snps_valid = false;
    h->xs_snps.clear();
  }
  return 0;
}

int muxgl_fmx_exact_rows(muxgl_handle* h, const muxgl_fmx_params* p, const int32_t* snps, int64_t n, double* rows,
                         uint8_t* owned) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmx_exact_rows");
  HIPCHK(h, hipSetDevice(h->device));
  if (fmx_check_iter(h, p, "muxgl_fmx_exact_rows")) return 1;
  if (n > 0 && (!snps || !rows)) MUXGL_FAIL(h, "muxgl_fmx_exact_rows: NULL argument");
  return fmx_exact_rows(h, p, snps, n, rows, owned);
}

int muxgl_fmx_exact_finish(muxgl_handle* h, const muxgl_fmx_params* p, const int32_t* snps, int64_t n, const double* rows,
                           int64_t* deltas, int32_t* reassigned) {
  if (!h || !deltas || !reassigned) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmx_exact_finish");
  HIPCHK(h, hipSetDevice(h->device));
  if (fmx_check_iter(h, p, "muxgl_fmx_exact_finish")) return 1;
  if (fmx_exact_finish(h, p, snps, n, rows, deltas, reassigned)) return 1;
  for (int i = 0; i < 3; ++i) h->h_fstat[i] += (int32_t)deltas[i];  // the rank's own counters, corrected
  h->h_fstat[3] = 0;
  h->fmx_exact_unresolved -= h->fmx_listed;
  h->fmx_listed = 0;
  HIPCHK(h, hipMemcpy(h->d_fstat, h->h_fstat, 4 * sizeof(int32_t), hipMemcpyHostToDevice));
  return 0;
}

// cells of the last muxgl_fmx_iter_fetch that are listed for the exact path and not settled yet
int muxgl_fmx_exact_pending(const muxgl_handle* h, int64_t* cells) {
  if (!h || !cells) return 1;
  *cells = h->group ? 0 : h->fmx_listed;
  return 0;
}

int muxgl_fmx_score_stats(const muxgl_handle* h, int64_t* exact_scores) {
  if (!h) return 1;
  if (exact_scores) *exact_scores = h->fmx_exact_scores;
  return 0;
}

int muxgl_fmx_exact_stats(const muxgl_handle* h, int64_t* near_tie_cells, int64_t* calls_changed, int64_t* unresolved) {
  if (!h) return 1;
  int64_t a = h->fmx_exact_cells, b = h->fmx_exact_changed, c = h->fmx_exact_unresolved;
  if (h->group) group_fmx_exact_stats(h, &a, &b, &c);
  if (near_tie_cells) *near_tie_cells = a;
  if (calls_changed) *calls_changed = b;
  if (unresolved) *unresolved = c;
  return 0;
}

// ---- sharded EM (multi-GPU): the same three phases, restricted to a cell range and a SNP range, with the exchanges
//      (all-gather of the cluster-GP rows after iter_gp, of the assignments after iter_estep) left to the caller ------

int muxgl_fmx_set_shard(muxgl_handle* h, int64_t c0, int64_t c1, int64_t s0, int64_t s1) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmx_set_shard");
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->fmx_prepared) MUXGL_FAIL(h, "muxgl_fmx_set_shard: call muxgl_fmx_prepare first");
  if (h->col) MUXGL_FAIL(h, "muxgl_fmx_set_shard: the handle holds slabs (muxgl_fmx_set_column_slab), which fix its ranges");
  if (c0 < 0 || c1 < c0 || c1 > h->C || s0 < 0 || s1 < s0 || s1 > h->S) MUXGL_FAIL(h, "muxgl_fmx_set_shard: bad range");
  h->fc0 = c0;
  h->fc1 = c1;
  h->fs0 = s0;
  h->fs1 = s1;
  demux_row_release(&h->frow);
  demux_row_release(&h->fqrow);
  if (c0 != 0 || c1 != h->C) {  // chunk tables of the cell shard for the row / quad E-step
    if (demux_row_build(h, &h->frow, c0, c1, MUXGL_ROW_CH)) return 1;
    if (demux_row_build(h, &h->fqrow, c0, c1, fmx_oct_chunk())) return 1;
  }
  return 0;
}

// end of a phase call: drained stream and timings, or (MUXGL_FLAG_ASYNC_PHASES) nothing -- the kernels are enqueued on
// muxgl_stream(h) and the caller orders its collectives against that stream
static int fmx_phase_done(muxgl_handle* h) {
  if (h->flags & MUXGL_FLAG_ASYNC_PHASES) return 0;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  collect_timing(h);
  return 0;
}

int muxgl_fmx_iter_gp(muxgl_handle* h, const muxgl_fmx_params* p) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmx_iter_gp");
  HIPCHK(h, hipSetDevice(h->device));
  if (fmx_check_iter(h, p, "muxgl_fmx_iter_gp")) return 1;
  clear_timing(h);
  if (fmx_phase_gp(h, p)) return 1;
  return fmx_phase_done(h);
}

int muxgl_fmx_iter_estep(muxgl_handle* h, const muxgl_fmx_params* p) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmx_iter_estep");
  HIPCHK(h, hipSetDevice(h->device));
  if (fmx_check_iter(h, p, "muxgl_fmx_iter_estep")) return 1;
  if (fmx_phase_estep(h, p)) return 1;  // (timings accumulate over the phases of an iteration: iter_gp clears them)
  return fmx_phase_done(h);
}

int muxgl_fmx_iter_mstep(muxgl_handle* h) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmx_iter_mstep");
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->fmx_prepared || h->K < 1) MUXGL_FAIL(h, "muxgl_fmx_iter_mstep: no clusters set");
  if (fmx_phase_mstep(h)) return 1;
  return fmx_phase_done(h);
}

int muxgl_fmx_iter_fetch(muxgl_handle* h, muxgl_fmx_cell* out, int32_t* nsingle, int32_t* namb, int32_t* nchanged,
                         double* full_ll) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmx_iter_fetch");
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->fmx_prepared || h->K < 1) MUXGL_FAIL(h, "muxgl_fmx_iter_fetch: no clusters set");
  if (fmx_phase_fetch(h, out, nsingle, namb, nchanged, full_ll)) return 1;
  if ((h->flags & MUXGL_FLAG_ASYNC_PHASES) && (out || full_ll)) collect_timing(h);  // the stream is drained here
  return 0;
}

int muxgl_fmx_buffer(muxgl_handle* h, int32_t which, void** dev_ptr, int64_t* n_elems) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmx_buffer");
  if (!dev_ptr || !n_elems) MUXGL_FAIL(h, "muxgl_fmx_buffer: NULL output");
  if (!h->fmx_prepared || h->K < 1) MUXGL_FAIL(h, "muxgl_fmx_buffer: no clusters set");
  switch (which) {
    case MUXGL_BUF_CGP:
      *dev_ptr = h->d_cgp;
      *n_elems = h->S * h->K * 3;
      return 0;
    case MUXGL_BUF_CLUST:  // job-wide: a slabbed handle keeps it with the column slab, whose merge reads it
      *dev_ptr = h->col ? h->col->d_clust : h->d_clust;
      *n_elems = h->col ? h->C_total : h->C;
      return 0;
    case MUXGL_BUF_CELLS:
      *dev_ptr = h->d_fcells;
      *n_elems = h->C;
      return 0;
    case MUXGL_BUF_STAT:
      *dev_ptr = h->d_fstat;
      *n_elems = 4;
      return 0;
    default:
      MUXGL_FAIL(h, "muxgl_fmx_buffer: unknown buffer %d", which);
  }
}

int muxgl_memcpy_dev(muxgl_handle* h, void* dst_dev, const void* src_dev, int64_t bytes) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_memcpy_dev");
  HIPCHK(h, hipSetDevice(h->device));
  if (bytes > 0) HIPCHK(h, hipMemcpyAsync(dst_dev, src_dev, (size_t)bytes, hipMemcpyDeviceToDevice, h->stream));
  if (!(h->flags & MUXGL_FLAG_ASYNC_PHASES)) HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int muxgl_fmx_get_cluster_pileup(muxgl_handle* h, double* gls, int32_t* counts) {
  if (!h) return 1;
  if (h->group) return group_fmx_get_cluster_pileup(h, gls, counts);
  HIPCHK(h, hipSetDevice(h->device));
  if (h->K < 1) MUXGL_FAIL(h, "muxgl_fmx_get_cluster_pileup: no clusters set");
  muxgl_handle* m = h->col ? h->col : h;  // a slabbed handle: rows of its SNP range, zeros elsewhere
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const size_t n = (size_t)h->K * h->S;
  if (gls && n) HIPCHK(h, hipMemcpy(gls, m->d_cgls, sizeof(double) * 9 * n, hipMemcpyDeviceToHost));
  if (counts && n) {
    if (fmx_cluster_counts_device(m)) {
      if (m != h) h->err = m->err;
      return 1;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(counts, m->d_ccnt, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToHost));
  }
  return 0;
}

}  // extern "C"
