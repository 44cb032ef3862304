// fmx_wave.hip -- freemuxlet E-step (cmd_cram_freemux2.cpp:383-456) for 16 < K <= 64 clusters: one wave per cell, one
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
#include "common.hpp"

namespace {

__device__ __forceinline__ double fw_wror1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x13C, 0xF, 0xF, false);  // wave_ror:1 : lane j <- lane (j-1) mod 64
  hi = __builtin_amdgcn_mov_dpp(hi, 0x13C, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(64, 2)
    fmx_estep_wave_kernel(const int32_t* __restrict__ order, int64_t n_cells, int64_t c0, int64_t c1,
                          const int64_t* __restrict__ cell_ptr, const int32_t* __restrict__ entry_snp,
                          const double* __restrict__ egls, const double* __restrict__ cgp, int K,
                          double* __restrict__ fll) {
  if ((int64_t)blockIdx.x >= n_cells) return;
  const int64_t c = order[blockIdx.x];
  if (c < c0 || c >= c1) return;  // not in this rank's cell shard
  const int64_t e0 = cell_ptr[c], e1 = cell_ptr[c + 1];
  const int j = threadIdx.x;
  const bool live = j < K;
  const int K3 = K * 3;
  const int npairs = K * (K + 1) / 2;

  double acc[32], accS = 1.0;
  int32_t ex[32], exS = 0;
#pragma unroll
  for (int t = 0; t < 32; ++t) {
    acc[t] = 1.0;
    ex[t] = 0;
  }
  double ng0 = 1.0, ng1 = 0.0, ng2 = 0.0;
  if (e0 < e1 && live) {
    const double* row = cgp + (size_t)entry_snp[e0] * K3 + j * 3;
    ng0 = row[0], ng1 = row[1], ng2 = row[2];
  }
  int cnt = 0;
  for (int64_t e = e0; e < e1; ++e) {
    const double g0 = ng0, g1 = ng1, g2 = ng2;
    ng0 = 1.0, ng1 = 0.0, ng2 = 0.0;
    if (e + 1 < e1 && live) {
      const double* row = cgp + (size_t)entry_snp[e + 1] * K3 + j * 3;
      ng0 = row[0], ng1 = row[1], ng2 = row[2];
    }
    const double* q = egls + (size_t)e * 9;  // wave-uniform: glis[g1*3+g2]
    const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7], q8 = q[8];
    accS *= fma(g2, q8, fma(g1, q4, g0 * q0));  // singlet: sum_g glis[g][g] * gp_j[g] (:448-452)
    const double u0 = fma(g2, q6, fma(g1, q3, g0 * q0));
    const double u1 = fma(g2, q7, fma(g1, q4, g0 * q1));
    const double u2 = fma(g2, q8, fma(g1, q5, g0 * q2));
    double r0 = g0, r1 = g1, r2 = g2;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      r0 = fw_wror1(r0);
      r1 = fw_wror1(r1);
      r2 = fw_wror1(r2);
      acc[t] *= fma(r2, u2, fma(r1, u1, r0 * u0));  // :440-446 as a product
    }
    if (++cnt == 16) {  // a factor is >= ~1e-13 (clamped likelihoods, mixed posteriors): sixteen cannot underflow
      cnt = 0;
#pragma unroll
      for (int t = 0; t < 32; ++t) prodacc_renorm(acc[t], ex[t]);
      prodacc_renorm(accS, exS);
    }
  }

  double* out = fll + (size_t)c * npairs;
  int kk = j;
#pragma unroll
  for (int t = 0; t < 32; ++t) {
    kk = __builtin_amdgcn_mov_dpp(kk, 0x13C, 0xF, 0xF, false);
    if (live && kk < K && kk != j && (t < 31 || j > kk)) {  // step 32 of 64 lanes meets every pair twice: one writer
      const int hi = j > kk ? j : kk, lo = j > kk ? kk : j;
      out[hi * (hi + 1) / 2 + lo] = prodacc_log(acc[t], ex[t]);
    }
  }
  if (live) out[j * (j + 1) / 2 + j] = prodacc_log(accS, exS);
}

}  // namespace

// returns -1 when this path does not apply, 0 ok, 1 error
int fmx_wave_estep_launch(muxgl_handle* h, int64_t c0, int64_t nc) {
  if (h->K <= 16 || h->K > 64 || (h->flags & MUXGL_FLAG_FORCE_TILE_SWEEP)) return -1;
  const int32_t* order = demux_wave_order(h);
  if (!order) return -1;
  hipLaunchKernelGGL(fmx_estep_wave_kernel, dim3((unsigned)h->C), dim3(64), 0, h->stream, order, h->C, c0, c0 + nc,
                     h->d_cell_ptr, h->d_entry_snp, h->d_egls, h->d_cgp, h->K, h->d_fll);
  HIPCHK(h, hipGetLastError());
  return 0;
}
