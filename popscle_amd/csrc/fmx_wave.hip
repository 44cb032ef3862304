// fmx_wave.hip -- freemuxlet E-step (cmd_cram_freemux2.cpp:383-456) for 16 < K <= 255 clusters: one wave per cell, one
// lane per cluster, the design of demux_wave.hip with the entry's nine genotype-pair likelihoods in the role of pG.
//
//   * lane j keeps cluster j's genotype posterior gp_j (three doubles from the per-iteration tensor cgp[S][K][3], loaded
//     one entry ahead) and u[m] = sum_l gp_j[l] * glis[l][m]; the entry's likelihoods are wave-uniform and come through
//     the scalar cache;
//   * the partner posterior reaches lane j by rotating the wave one lane per step (DPP wave_ror:1); the pair likelihood
//     (:440-446) is symmetric in the two clusters, so 32 steps meet every unordered pair (the last step meets each pair
//     from both sides: one writer); the singlet (:448-452) uses the diagonal of glis only;
//   * products as mantissa * 2^exponent, one log per (cell, hypothesis), written to llks[j(j+1)/2 + k] directly.
// The general pair kernel this replaces for K = 64 ran at 5 % of the FP64 issue rate (one thread per pair, every thread
// fetching both posteriors of every entry).
// Beyond 64 clusters the K x K pair matrix is cut into 64 x 64 blocks as in demux_wave.hip: a diagonal block is the
// kernel above on clusters 64X .. 64X+63; an off-diagonal block (CROSS, X > Y only: the likelihood is symmetric) keeps
// cluster 64X + j in lane j and rotates the posteriors of clusters 64Y + k past it, all 64 rotations being pairs.
#include "common.hpp"

namespace {

__device__ __forceinline__ double fw_wror1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x13C, 0xF, 0xF, false);  // wave_ror:1 : lane j <- lane (j-1) mod 64
  hi = __builtin_amdgcn_mov_dpp(hi, 0x13C, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// LIN: entries flagged in `lin` (fmx_entry_kernel: at most one usable read, no clamp) have likelihoods that are linear
// in g1 + g2, glis[g1][g2] = c0 + c1 (g1 + g2), so that
//     sum_{l,m} P_j[l] P_k[m] glis[l][m] = s_k (c0 s_j + c1 E_j) + E_k (c1 s_j),   s = sum_l P[l],  E = P[1] + 2 P[2]:
// two moments of the partner rotate instead of its three posteriors (4 DPP moves instead of 6) and a pair costs a
// multiply, an FMA and the product update instead of a multiply, two FMAs and the update -- 7 vector instructions per
// pair instead of 10.  Three quarters of the entries of a typical pileup are such entries.  The branch is wave-uniform
// (a wave walks one cell, one entry at a time), so the order of a cell's factors is unchanged.
template <bool CROSS, bool LIN>
__global__ void __launch_bounds__(64, (LIN && !CROSS) ? 3 : 2)
    fmx_estep_wave_kernel(const wave_item* __restrict__ items, int64_t n_items, int64_t c0, int64_t c1,
                          const int64_t* __restrict__ cell_ptr, const int32_t* __restrict__ entry_snp,
                          const double* __restrict__ egls, const uint32_t* __restrict__ lin,
                          const double* __restrict__ cgp, int K, int jbase, int kbase, double* __restrict__ fll) {
  constexpr int NS = CROSS ? 64 : 32;
  if ((int64_t)blockIdx.x >= n_items) return;
  const wave_item it = items[blockIdx.x];  // a cell, or a part of a long one (common.hpp)
  if (it.cell < c0 || it.cell >= c1) return;  // not in this rank's cell shard
  const int64_t c = it.slab;  // row of fll: the cell, or an overflow row behind the C cell rows
  const int64_t e0 = it.e0, e1 = it.e1;
  const int j = threadIdx.x;
  const int sj = jbase + j;
  const bool live = sj < K, live2 = kbase + j < K;
  const int K3 = K * 3;
  const int npairs = K * (K + 1) / 2;

  // CROSS: 64 accumulators per lane; their integer exponents live in LDS (touched once per 16 entries), 16 KB per wave
  constexpr bool EXL = CROSS || LIN;  // exponents in LDS: the two sweep bodies of LIN leave no room for them either
  __shared__ int32_t exs[EXL ? NS : 1][64];
  double acc[NS], accS = 1.0;
  int32_t ex[EXL ? 1 : NS], exS = 0;
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    acc[t] = 1.0;
    if (EXL) exs[t][j] = 0;
    else ex[t] = 0;
  }
  double ng0 = 1.0, ng1 = 0.0, ng2 = 0.0;
  double np0 = 1.0, np1 = 0.0, np2 = 0.0;  // CROSS: posterior of cluster kbase + j
  if (e0 < e1 && live) {
    const double* row = cgp + (size_t)entry_snp[e0] * K3 + sj * 3;
    ng0 = row[0], ng1 = row[1], ng2 = row[2];
  }
  if (CROSS && e0 < e1 && live2) {
    const double* row = cgp + (size_t)entry_snp[e0] * K3 + (kbase + j) * 3;
    np0 = row[0], np1 = row[1], np2 = row[2];
  }
  // Software pipeline over the entries: the posterior triples AND the entry's nine (wave-uniform, scalar-loaded)
  // likelihoods of entry e + 1 are requested before entry e is swept, and the per-entry factors that do not depend on
  // the rotation (u, the singlet term) are formed for e + 1 right after the sweep of e: the scalar loads then have a
  // whole sweep to land instead of stalling the wave at the top of every entry.
  // u0..u2 of the general form; for a linear entry u0 = c0 s_j + c1 E_j, u1 = c1 s_j and the ring carries (s, E)
  double u0 = 0, u1 = 0, u2 = 0, sing = 1.0;
  double c0r = ng0, c1r = ng1, c2r = ng2;  // ring start of the current entry (own triple, or the partner's with CROSS)
  bool lin_cur = false;
  auto is_lin = [&](int64_t e) { return LIN && ((lin[e >> 5] >> (e & 31)) & 1u); };
  if (e0 < e1) {
    const double* q = egls + (size_t)e0 * 9;
    sing = fma(ng2, q[8], fma(ng1, q[4], ng0 * q[0]));
    lin_cur = is_lin(e0);
    if (lin_cur) {
      const double s = (ng0 + ng1) + ng2, E = fma(2.0, ng2, ng1), cc1 = q[1] - q[0];
      u0 = fma(cc1, E, q[0] * s);
      u1 = cc1 * s;
      c0r = CROSS ? (np0 + np1) + np2 : s;
      c1r = CROSS ? fma(2.0, np2, np1) : E;
    } else {
      u0 = fma(ng2, q[6], fma(ng1, q[3], ng0 * q[0]));
      u1 = fma(ng2, q[7], fma(ng1, q[4], ng0 * q[1]));
      u2 = fma(ng2, q[8], fma(ng1, q[5], ng0 * q[2]));
      if (CROSS) c0r = np0, c1r = np1, c2r = np2;
    }
  }
  int cnt = 0;
  for (int64_t e = e0; e < e1; ++e) {
    const bool more = e + 1 < e1;
    // requests for entry e + 1
    ng0 = 1.0, ng1 = 0.0, ng2 = 0.0;
    if (more && live) {
      const double* row = cgp + (size_t)entry_snp[e + 1] * K3 + sj * 3;
      ng0 = row[0], ng1 = row[1], ng2 = row[2];
    }
    if (CROSS) {
      np0 = 1.0, np1 = 0.0, np2 = 0.0;
      if (more && live2) {
        const double* row = cgp + (size_t)entry_snp[e + 1] * K3 + (kbase + j) * 3;
        np0 = row[0], np1 = row[1], np2 = row[2];
      }
    }
    const int64_t en = more ? e + 1 : e;
    const double* q = egls + (size_t)en * 9;  // wave-uniform: glis[g1*3+g2] of the next entry
    const bool lin_nx = is_lin(en);
    const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7], q8 = q[8];
    // sweep of entry e
    if (!CROSS) accS *= sing;  // singlet: sum_g glis[g][g] * gp_j[g] (:448-452)
    if (lin_cur) {
      double r0 = c0r, r1 = c1r;  // the partner's (s, E)
      if (CROSS) {
        r0 = __shfl(r0, (j + 1) & 63, 64);
        r1 = __shfl(r1, (j + 1) & 63, 64);
      }
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        r0 = fw_wror1(r0);
        r1 = fw_wror1(r1);
        acc[t] *= fma(r0, u0, r1 * u1);
        if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from forming all 32 sums first (64 VGPRs)
      }
    } else {
      double r0 = c0r, r1 = c1r, r2 = c2r;
      if (CROSS) {  // one lane ahead: the first rotation then brings cluster kbase + j itself
        r0 = __shfl(r0, (j + 1) & 63, 64);
        r1 = __shfl(r1, (j + 1) & 63, 64);
        r2 = __shfl(r2, (j + 1) & 63, 64);
      }
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        r0 = fw_wror1(r0);
        r1 = fw_wror1(r1);
        r2 = fw_wror1(r2);
        acc[t] *= fma(r2, u2, fma(r1, u1, r0 * u0));  // :440-446 as a product
        if (LIN && (t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    // factors of entry e + 1
    sing = fma(ng2, q8, fma(ng1, q4, ng0 * q0));
    lin_cur = lin_nx;
    if (lin_nx) {
      const double s = (ng0 + ng1) + ng2, E = fma(2.0, ng2, ng1), cc1 = q1 - q0;
      u0 = fma(cc1, E, q0 * s);
      u1 = cc1 * s;
      c0r = CROSS ? (np0 + np1) + np2 : s;
      c1r = CROSS ? fma(2.0, np2, np1) : E;
    } else {
      u0 = fma(ng2, q6, fma(ng1, q3, ng0 * q0));
      u1 = fma(ng2, q7, fma(ng1, q4, ng0 * q1));
      u2 = fma(ng2, q8, fma(ng1, q5, ng0 * q2));
      c0r = CROSS ? np0 : ng0, c1r = CROSS ? np1 : ng1, c2r = CROSS ? np2 : ng2;
    }
    if (++cnt == 16) {  // a factor is >= ~1e-13 (clamped likelihoods, mixed posteriors): sixteen cannot underflow
      cnt = 0;
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        if (EXL) {
          int ee;
          acc[t] = frexp(acc[t], &ee);
          exs[t][j] += ee;
        } else {
          prodacc_renorm(acc[t], ex[t]);
        }
      }
      if (!CROSS) prodacc_renorm(accS, exS);
    }
  }

  double* out = fll + (size_t)c * npairs;
  int kk = CROSS ? ((j + 1) & 63) : j;  // lane whose posterior this lane holds, followed through the rotations
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    kk = __builtin_amdgcn_mov_dpp(kk, 0x13C, 0xF, 0xF, false);
    const int sk = kbase + kk;
    if (CROSS) {
      if (live && sk < K) out[sj * (sj + 1) / 2 + sk] = prodacc_log(acc[t], exs[t][j]);  // jbase > kbase: sj > sk
    } else if (live && sk < K && kk != j && (t < 31 || j > kk)) {  // step 32 of 64 lanes meets every pair twice: one writer
      const int hi = sj > sk ? sj : sk, lo = sj > sk ? sk : sj;
      out[hi * (hi + 1) / 2 + lo] = prodacc_log(acc[t], EXL ? exs[t][j] : ex[EXL ? 0 : t]);
    }
  }
  if (!CROSS && live) out[sj * (sj + 1) / 2 + sj] = prodacc_log(accS, exS);
}

// 16 < K <= 32: the wave as a ring of 32.  Both 32-lane halves hold the same 32 posteriors (lane j and j + 32: cluster
// j & 31), so wave_ror:1 rotates the ring inside each half; lane j of the upper half works for cluster (j + 8) & 31, i.e.
// it sees the ring eight positions further on.  The pair likelihood is symmetric, so the ring offsets 1..16 are all the
// unordered pairs: the lower half meets offsets 1..8, the upper half 9..16 -- eight steps instead of 32 (offset 16 meets
// a pair from both sides: one writer).
__global__ void __launch_bounds__(64, 4)
    fmx_estep_wave32_kernel(const wave_item* __restrict__ items, int64_t n_items, int64_t c0, int64_t c1,
                            const int32_t* __restrict__ entry_snp, const double* __restrict__ egls,
                            const double* __restrict__ cgp, int K, double* __restrict__ fll) {
  constexpr int NS = 8;
  if ((int64_t)blockIdx.x >= n_items) return;
  const wave_item it = items[blockIdx.x];
  if (it.cell < c0 || it.cell >= c1) return;
  const int64_t c = it.slab;
  const int64_t e0 = it.e0, e1 = it.e1;
  const int j = threadIdx.x;
  const int half = j >> 5, sj = j & 31;  // ring position
  const int so = (sj + 8 * half) & 31;   // the cluster this lane works for
  const bool live = so < K, rlive = sj < K;
  const int K3 = K * 3;
  const int npairs = K * (K + 1) / 2;

  double acc[NS], accS = 1.0;
  int32_t ex[NS], exS = 0;
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    acc[t] = 1.0;
    ex[t] = 0;
  }
  double ng0 = 1.0, ng1 = 0.0, ng2 = 0.0;  // own posterior (cluster so)
  double nr0 = 1.0, nr1 = 0.0, nr2 = 0.0;  // ring posterior (cluster sj)
  if (e0 < e1) {
    const double* row = cgp + (size_t)entry_snp[e0] * K3;
    if (live) ng0 = row[so * 3], ng1 = row[so * 3 + 1], ng2 = row[so * 3 + 2];
    if (rlive) nr0 = row[sj * 3], nr1 = row[sj * 3 + 1], nr2 = row[sj * 3 + 2];
  }
  int cnt = 0;
  for (int64_t e = e0; e < e1; ++e) {
    const double g0 = ng0, g1 = ng1, g2 = ng2;
    double r0 = nr0, r1 = nr1, r2 = nr2;
    ng0 = 1.0, ng1 = 0.0, ng2 = 0.0;
    nr0 = 1.0, nr1 = 0.0, nr2 = 0.0;
    if (e + 1 < e1) {
      const double* row = cgp + (size_t)entry_snp[e + 1] * K3;
      if (live) ng0 = row[so * 3], ng1 = row[so * 3 + 1], ng2 = row[so * 3 + 2];
      if (rlive) nr0 = row[sj * 3], nr1 = row[sj * 3 + 1], nr2 = row[sj * 3 + 2];
    }
    const double* q = egls + (size_t)e * 9;  // wave-uniform: glis[g1*3+g2]
    const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7], q8 = q[8];
    accS *= fma(g2, q8, fma(g1, q4, g0 * q0));  // singlet: sum_g glis[g][g] * gp_j[g] (:448-452)
    const double u0 = fma(g2, q6, fma(g1, q3, g0 * q0));
    const double u1 = fma(g2, q7, fma(g1, q4, g0 * q1));
    const double u2 = fma(g2, q8, fma(g1, q5, g0 * q2));
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      r0 = fw_wror1(r0);
      r1 = fw_wror1(r1);
      r2 = fw_wror1(r2);
      acc[t] *= fma(r2, u2, fma(r1, u1, r0 * u0));  // :440-446 as a product
    }
    if (++cnt == 16) {  // a factor is >= ~1e-13 (clamped likelihoods, mixed posteriors): sixteen cannot underflow
      cnt = 0;
#pragma unroll
      for (int t = 0; t < NS; ++t) prodacc_renorm(acc[t], ex[t]);
      prodacc_renorm(accS, exS);
    }
  }

  double* out = fll + (size_t)c * npairs;
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const int k = (sj - t - 1) & 31;  // wave_ror:1 brings lane j the value of lane j - 1: here inside the ring of 32
    if (!live || k >= K || k == so) continue;
    if (half == 1 && t == NS - 1 && so < k) continue;  // ring offset 16: both ends meet the pair
    const int hi = so > k ? so : k, lo = so > k ? k : so;
    out[hi * (hi + 1) / 2 + lo] = prodacc_log(acc[t], ex[t]);
  }
  if (half == 0 && live) out[so * (so + 1) / 2 + so] = prodacc_log(accS, exS);
}

// rows of the parts of a cut cell added, in entry order, into the cell's row of fll
__global__ void __launch_bounds__(256)
    fmx_wave_combine_kernel(const wave_cut* __restrict__ cuts, int64_t c0, int64_t c1, int npairs, double* __restrict__ fll) {
  const wave_cut cu = cuts[blockIdx.x];
  if (cu.cell < c0 || cu.cell >= c1) return;
  double* dst = fll + (size_t)cu.cell * npairs;
  for (int i = threadIdx.x; i < npairs; i += blockDim.x) {
    double v = dst[i];
    for (int64_t q = 0; q < cu.count; ++q) v += fll[(size_t)(cu.first + q) * npairs + i];
    dst[i] = v;
  }
}

}  // namespace

// rows of d_fll the wave E-step needs: one per cell plus one per extra part of a long cell
int64_t fmx_wave_fll_rows(const muxgl_handle* h) {
  const wave_item* items;
  const wave_cut* cuts;
  int64_t n_items, n_cuts, n_over = 0;
  if (demux_wave_items(h, &items, &n_items, &cuts, &n_cuts, &n_over)) return h->C;
  return h->C + n_over;
}

// returns -1 when this path does not apply, 0 ok, 1 error
int fmx_wave_estep_launch(muxgl_handle* h, int64_t c0, int64_t nc) {
  if (h->K <= 16 || (h->flags & MUXGL_FLAG_FORCE_TILE_SWEEP)) return -1;
  const wave_item* items;
  const wave_cut* cuts;
  int64_t n_items, n_cuts, n_over;
  if (demux_wave_items(h, &items, &n_items, &cuts, &n_cuts, &n_over) || n_items == 0) return -1;
  const int nblk = (h->K + 63) / 64;
  const bool use_lin = h->d_flin && !(h->flags & MUXGL_FLAG_NO_LINEAR_ENTRIES);  // two-term form for linear entries
  if (h->K <= 32)
    hipLaunchKernelGGL(fmx_estep_wave32_kernel, dim3((unsigned)n_items), dim3(64), 0, h->stream, items, n_items, c0,
                       c0 + nc, h->d_entry_snp, h->d_egls, h->d_cgp, h->K, h->d_fll);
  for (int X = 0; X < (h->K <= 32 ? 0 : nblk); ++X) {
#define FW_ARGS(XB, YB) \
  items, n_items, c0, c0 + nc, h->d_cell_ptr, h->d_entry_snp, h->d_egls, h->d_flin, h->d_cgp, h->K, 64 * (XB), 64 * (YB), h->d_fll
    if (use_lin) hipLaunchKernelGGL((fmx_estep_wave_kernel<false, true>), dim3((unsigned)n_items), dim3(64), 0, h->stream, FW_ARGS(X, X));
    else hipLaunchKernelGGL((fmx_estep_wave_kernel<false, false>), dim3((unsigned)n_items), dim3(64), 0, h->stream, FW_ARGS(X, X));
    for (int Y = 0; Y < X; ++Y)  // (off-diagonal blocks hold 64 accumulators per lane: no room for a second sweep body)
      hipLaunchKernelGGL((fmx_estep_wave_kernel<true, false>), dim3((unsigned)n_items), dim3(64), 0, h->stream, FW_ARGS(X, Y));
#undef FW_ARGS
  }
  if (n_cuts)
    hipLaunchKernelGGL(fmx_wave_combine_kernel, dim3((unsigned)n_cuts), dim3(256), 0, h->stream, cuts, c0, c0 + nc,
                       h->K * (h->K + 1) / 2, h->d_fll);
  HIPCHK(h, hipGetLastError());
  return 0;
}
