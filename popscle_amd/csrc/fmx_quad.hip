// fmx_quad.hip -- freemuxlet E-step (cmd_cram_freemux2.cpp:383-456) for K <= 16 clusters with the quad tiling of
// demux_quad.hip: 4 lanes per entry, lane r owns clusters 4r..4r+3, 16 entries per wave iteration.
//
// Per entry the reference evaluates, for every cluster pair k < j,  lk = sum_{g1,g2} glis[g1][g2] gp_j[g1] gp_k[g2]
// (:440-446) and for every cluster  lk = sum_g glis[g][g] gp_j[g]  (:448-452), and adds log(lk) to llks (:454-455).
// glis (calculate_snp_droplet_pileup, alpha = 0.5) is symmetric, so unordered pairs suffice.  Here:
//   u[c][m] = sum_l gp_c[l] * glis[l][m]  for the lane's four clusters, then every pair costs 3 FMA + 1 multiply into a
//   product accumulator (6 pairs inside the lane, 16 with the neighbouring tile via row_ror:4, 10 with the opposite tile
//   via row_ror:8).  Products leave as (mantissa, exponent) per chunk; fmx_quad_reduce_kernel takes one log per
//   (cell, pair).  The cluster-GP rows are re-laid per quad ([S][6][4][2], fmx_cgpq_kernel) once per iteration.
// Entries whose likelihoods are linear in the genotypes (fmx_entry_kernel: glis[g1][g2] = c0 + c1 (g1 + g2)) come first in
// every chunk (fq_partition_kernel) and are swept by a loop of their own from the clusters' moments E = g1 + 2 g2
// (fmx_ceq_kernel): pair (c0 + c1 E_j) + c1 E_k, singlet c0 + 2 c1 E_j -- an FMA and the product update per hypothesis.
#include "common.hpp"

namespace {

constexpr int FQ_ACC = 36;          // 4 singlets, 6 in-lane pairs, 16 + 10 cross-tile pairs
constexpr int FQ_ENTRY = 6;         // doubles per entry in LDS: the six distinct likelihoods {00,11,22,01,02,12}
constexpr int FQ_SLOT_STRIDE = 26;  // 4 entries x 6 + 2: slot regions 208 B apart => distinct LDS banks for 16 slots

__device__ __forceinline__ double fq_ror4(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x124, 0xF, 0xF, false);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x124, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fq_ror8(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x128, 0xF, 0xF, false);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x128, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__global__ void fq_tmap_kernel(int32_t* tmap /*[2][4]*/) {
  const int lane = threadIdx.x;
  const int t = (lane >> 2) & 3;
  const int t4 = __builtin_amdgcn_mov_dpp(t, 0x124, 0xF, 0xF, false);
  const int t8 = __builtin_amdgcn_mov_dpp(t, 0x128, 0xF, 0xF, false);
  if (lane < 16 && (lane & 3) == 0) {
    tmap[t] = t4;
    tmap[4 + t] = t8;
  }
}

__host__ __device__ constexpr int fq_within(int c1, int c2) { return 4 + (c1 == 0 ? c2 - 1 : (c1 == 1 ? 3 + c2 - 2 : 5)); }
__host__ __device__ constexpr int fq_t1(int c, int d) { return 10 + c * 4 + d; }
__host__ __device__ constexpr int fq_t2(int c, int d) { return 26 + (c == 0 ? d : (c == 1 ? 4 + d - 1 : (c == 2 ? 7 + d - 2 : 9))); }

// cluster-GP rows [S][K][3] -> [S][6][4][2]; clusters >= K are padded with (1,0,0) (their factors are exactly 1)
__global__ void __launch_bounds__(256)
    fmx_cgpq_kernel(int64_t S, int K, const double* __restrict__ cgp, double* __restrict__ cgpq) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= S * 48) return;
  const int64_t s = tid / 48;
  const int w = (int)(tid - s * 48);  // position inside the quad row: ((t*4 + r)*2 + half)
  const int half = w & 1, r = (w >> 1) & 3, t = w >> 3;
  const int d = 2 * t + half, j = 4 * r + d / 3, l = d % 3;
  cgpq[tid] = (j < K) ? cgp[((size_t)s * K + j) * 3 + l] : (l == 0 ? 1.0 : 0.0);
}

// E = g1 + 2 g2 of the cluster posteriors in the quad layout, [S + 1][2][4][2]: lane r reads its four clusters' moments
// as two 16-byte pieces, piece t of the four lanes contiguous.  Clusters >= K and the neutral row S: 0.
__global__ void __launch_bounds__(256) fmx_ceq_kernel(int64_t S, int K, const double* __restrict__ cgp, double* __restrict__ ceq) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (S + 1) * 16) return;
  const int64_t s = tid >> 4;
  const int w = (int)(tid & 15), half = w & 1, r = (w >> 1) & 3, t = w >> 3;
  const int j = 4 * r + 2 * t + half;
  double E = 0.0;
  if (s < S && j < K) {
    const double* g = cgp + ((size_t)s * K + j) * 3;
    E = fma(2.0, g[2], g[1]);
  }
  ceq[tid] = E;
}

// A chunk's linear entries (fmx_entry_kernel: glis[g1][g2] = c0 + c1 (g1 + g2)) as records {c0, c1, snp} in front, the SNP
// ids and six likelihoods of its other entries behind them, both kinds in entry order; one thread per chunk, once per
// muxgl_fmx_prepare.
__global__ void __launch_bounds__(64)
    fq_partition_kernel(int n_chunks, const row_chunk* __restrict__ chunks, const uint32_t* __restrict__ flin,
                        const int32_t* __restrict__ entry_snp, const double* __restrict__ egls6, fmx_lrec* __restrict__ lrec,
                        int32_t* __restrict__ gsnp, double* __restrict__ gl6, int32_t* __restrict__ nlin) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_chunks) return;
  const int64_t e0 = chunks[q].e0;
  const int len = chunks[q].len;
  int w = 0;
  for (int i = 0; i < len; ++i) {
    const int64_t e = e0 + i;
    if ((flin[e >> 5] >> (e & 31)) & 1u) {
      const double g00 = egls6[(size_t)e * 6];
      lrec[e0 + w++] = fmx_lrec{g00, egls6[(size_t)e * 6 + 3] - g00, entry_snp[e], 0};
    }
  }
  nlin[q] = w;
  for (int i = 0; i < len; ++i) {
    const int64_t e = e0 + i;
    if (!((flin[e >> 5] >> (e & 31)) & 1u)) {
      gsnp[e0 + w] = entry_snp[e];
#pragma unroll
      for (int t = 0; t < 6; ++t) gl6[(size_t)(e0 + w) * 6 + t] = egls6[(size_t)e * 6 + t];
      ++w;
    }
  }
}

__global__ void __launch_bounds__(64, 2)
    fmx_estep_quad_kernel(const row_chunk* __restrict__ chunks, int n_chunks, const int32_t* __restrict__ entry_snp,
                          const double* __restrict__ egls6, const fmx_lrec* __restrict__ lrec,
                          const int32_t* __restrict__ chunk_nlin, const int32_t* __restrict__ order,
                          const double* __restrict__ cgpq,
                          const double* __restrict__ ceq, int32_t S_dummy, double* __restrict__ part_m,
                          int32_t* __restrict__ part_e) {
  __shared__ __align__(16) double gl[16 * FQ_SLOT_STRIDE];
  __shared__ int32_t snps[64], snps_nx[64];

  const int lane = threadIdx.x;
  const int r = (lane >> 2) & 3;
  const int slot = ((lane >> 4) << 2) | (lane & 3);
  const int wq = xcd_swizzle(blockIdx.x, gridDim.x >> 3) * 16 + slot;  // place in the launch order (quad_order_key_kernel)
  const int q = wq < n_chunks ? (order ? order[wq] : wq) : n_chunks;
  int64_t e0 = 0;
  int len = 0;
  if (q < n_chunks) {
    e0 = chunks[q].e0;
    len = chunks[q].len;
  }
  // The chunk's first nl entries are its linear ones (fq_partition_kernel; 0 when that form is off; entry_snp / egls6
  // are then the partitioned copies): a loop of their own below, the nine-term loop for the rest.
  const int nl = (chunk_nlin && q < n_chunks) ? chunk_nlin[q] : 0;
  const int nbL = (wave_max_i32(nl) + 3) >> 2;
  const int nb = (wave_max_i32(len - nl) + 3) >> 2;

  double acc[FQ_ACC];
  int32_t ex[FQ_ACC];
#pragma unroll
  for (int a = 0; a < FQ_ACC; ++a) {
    acc[a] = 1.0;
    ex[a] = 0;
  }

  // the entry of the batch to come (SNP id + its nine likelihoods) is fetched one batch ahead
  int32_t psnp = -1;
  double pgl[6];
  auto fetch_entry = [&](int b) {
    const int idx = nl + b * 4 + r;
    psnp = -1;
#pragma unroll
    for (int i = 0; i < 6; ++i) pgl[i] = 1.0;  // dead entry: with g = (1,0,0) every factor is exactly 1
    if (idx < len) {
      const int64_t e = e0 + idx;
      psnp = entry_snp[e];
      const double2* src = reinterpret_cast<const double2*>(egls6 + (size_t)e * 6);  // 48 B, 16-byte aligned
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double2 v = src[i];
        pgl[2 * i] = v.x;
        pgl[2 * i + 1] = v.y;
      }
    }
  };
  double nG[4][3];
  auto load_row = [&](int32_t s) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      nG[c][0] = 1.0;
      nG[c][1] = 0.0;
      nG[c][2] = 0.0;
    }
    if (s >= 0) {
      const double2* pc = reinterpret_cast<const double2*>(cgpq + (size_t)s * 48) + r;
      double f[12];
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const double2 v = pc[t * 4];
        f[2 * t] = v.x;
        f[2 * t + 1] = v.y;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        nG[c][0] = f[3 * c];
        nG[c][1] = f[3 * c + 1];
        nG[c][2] = f[3 * c + 2];
      }
    }
  };
  // ---- the linear entries: glis[g1][g2] = c0 + c1 (g1 + g2), taken at s = 1 (posteriors normalised in FP64, see
  //      fmx_wave.hip): singlet c0 + 2 c1 E_j, pair (c0 + c1 E_j) + c1 E_k with E = g1 + 2 g2 -- a row is 4 doubles
  //      instead of 12, one double per cluster rotates instead of three, a pair is an FMA and the product update.
  if (nbL > 0) {
    struct rowl_t {
      double E[4];
    };
    auto load_rowl = [&](rowl_t& R, int32_t sidx) {
      const double2* pc = reinterpret_cast<const double2*>(ceq + (size_t)sidx * 16) + r;
      const double2 v0 = pc[0], v1 = pc[4];
      R.E[0] = v0.x, R.E[1] = v0.y, R.E[2] = v1.x, R.E[3] = v1.y;
    };
    const int lastL = nl > 0 ? nl - 1 : 0;
    auto fetchL = [&](int b) {
      const int idx = b * 4 + r;
      return lrec[e0 + (idx < lastL ? idx : lastL)];  // (unconditional; invalidated where used)
    };
    fmx_lrec pa = fetchL(0);
    rowl_t L[4];
    snps_nx[slot * 4 + r] = (r < nl) ? pa.snp : S_dummy;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) load_rowl(L[i], snps_nx[slot * 4 + i]);
    __syncthreads();
    for (int b = 0; b < nbL; ++b) {
      {  // phase 1: lane <-> entry
        const bool in = b * 4 + r < nl;
        double* dst = gl + slot * FQ_SLOT_STRIDE + r * FQ_ENTRY;
        dst[0] = in ? pa.c0 : 1.0;  // dead entry: a factor of exactly 1
        dst[1] = in ? pa.c1 : 0.0;
        pa = fetchL(b + 1);
        snps_nx[slot * 4 + r] = ((b + 1) * 4 + r < nl) ? pa.snp : S_dummy;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double* p = gl + slot * FQ_SLOT_STRIDE + i * FQ_ENTRY;
        const double c0v = p[0], c1v = p[1], c2v = c1v + c1v;
        double X[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[c] *= fma(c2v, L[i].E[c], c0v);  // singlet (:448-452)
          X[c] = fma(c1v, L[i].E[c], c0v);
        }
#pragma unroll
        for (int c1 = 0; c1 < 4; ++c1)
#pragma unroll
          for (int c2 = c1 + 1; c2 < 4; ++c2) acc[fq_within(c1, c2)] *= fma(c1v, L[i].E[c2], X[c1]);
        {
          double P[4];
#pragma unroll
          for (int d = 0; d < 4; ++d) P[d] = fq_ror4(L[i].E[d]);
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = 0; d < 4; ++d) acc[fq_t1(c, d)] *= fma(c1v, P[d], X[c]);  // :440-446
        }
        {
          double Q[4];
#pragma unroll
          for (int d = 0; d < 4; ++d) Q[d] = fq_ror8(L[i].E[d]);
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = c; d < 4; ++d) acc[fq_t2(c, d)] *= fma(c1v, Q[d], X[c]);
        }
        load_rowl(L[i], snps_nx[slot * 4 + i]);  // entry i of the next batch
      }
      if ((b & 3) == 3) {
#pragma unroll
        for (int a = 0; a < FQ_ACC; ++a) prodacc_renorm(acc[a], ex[a]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < FQ_ACC; ++a) prodacc_renorm(acc[a], ex[a]);
  }

  fetch_entry(0);
  snps_nx[slot * 4 + r] = psnp;
  __syncthreads();
  load_row(snps_nx[slot * 4]);

  for (int b = 0; b < nb; ++b) {
    {  // phase 1: lane <-> entry, likelihoods of 64 entries into LDS
      double* dst = gl + slot * FQ_SLOT_STRIDE + r * FQ_ENTRY;
#pragma unroll
      for (int i = 0; i < 6; ++i) dst[i] = pgl[i];
      snps[slot * 4 + r] = psnp;
      if (b + 1 < nb) fetch_entry(b + 1);
    }
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      double G[4][3];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        G[c][0] = nG[c][0];
        G[c][1] = nG[c][1];
        G[c][2] = nG[c][2];
      }
      if (i == 2 && b + 1 < nb) snps_nx[slot * 4 + r] = psnp;
      if (i + 1 < 4) load_row(snps[slot * 4 + i + 1]);
      else if (b + 1 < nb) load_row(snps_nx[slot * 4]);

      const double* p = gl + slot * FQ_SLOT_STRIDE + i * FQ_ENTRY;
      const double p0 = p[0], p4 = p[1], p8 = p[2], p1 = p[3], p2 = p[4], p5 = p[5];
      const double p3 = p1, p6 = p2, p7 = p5;  // glis[g1][g2] == glis[g2][g1]
      double u[4][3];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[c] *= fma(G[c][2], p8, fma(G[c][1], p4, G[c][0] * p0));  // singlet (:448-452)
        u[c][0] = fma(G[c][2], p6, fma(G[c][1], p3, G[c][0] * p0));
        u[c][1] = fma(G[c][2], p7, fma(G[c][1], p4, G[c][0] * p1));
        u[c][2] = fma(G[c][2], p8, fma(G[c][1], p5, G[c][0] * p2));
      }
#pragma unroll
      for (int c1 = 0; c1 < 4; ++c1)
#pragma unroll
        for (int c2 = c1 + 1; c2 < 4; ++c2)
          acc[fq_within(c1, c2)] *= fma(G[c2][2], u[c1][2], fma(G[c2][1], u[c1][1], G[c2][0] * u[c1][0]));
      {
        double P[4][3];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          P[d][0] = fq_ror4(G[d][0]);
          P[d][1] = fq_ror4(G[d][1]);
          P[d][2] = fq_ror4(G[d][2]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int d = 0; d < 4; ++d)
            acc[fq_t1(c, d)] *= fma(P[d][2], u[c][2], fma(P[d][1], u[c][1], P[d][0] * u[c][0]));  // :440-446
      }
      {
        double Q[4][3];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          Q[d][0] = fq_ror8(G[d][0]);
          Q[d][1] = fq_ror8(G[d][1]);
          Q[d][2] = fq_ror8(G[d][2]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int d = c; d < 4; ++d)
            acc[fq_t2(c, d)] *= fma(Q[d][2], u[c][2], fma(Q[d][1], u[c][1], Q[d][0] * u[c][0]));
      }
    }
    if ((b & 3) == 3) {
#pragma unroll
      for (int a = 0; a < FQ_ACC; ++a) prodacc_renorm(acc[a], ex[a]);
    }
    __syncthreads();
  }
  if (q < n_chunks) {
#pragma unroll
    for (int a = 0; a < FQ_ACC; ++a) {
      prodacc_renorm(acc[a], ex[a]);
      part_m[((size_t)q * FQ_ACC + a) * 4 + r] = acc[a];
      part_e[((size_t)q * FQ_ACC + a) * 4 + r] = ex[a];
    }
  }
}

__global__ void __launch_bounds__(192)
    fmx_quad_reduce_kernel(const int64_t* __restrict__ cell_chunk_ptr, const int32_t* __restrict__ cell_chunks,
                           const double* __restrict__ part_m, const int32_t* __restrict__ part_e,
                           const int32_t* __restrict__ tmap, int K, int64_t c_off, double* __restrict__ fll) {
  const int64_t c = c_off + blockIdx.x;
  const int64_t c0 = cell_chunk_ptr[c], c1 = cell_chunk_ptr[c + 1];
  const int npairs = K * (K + 1) / 2;
  const int idx = threadIdx.x;
  if (idx >= FQ_ACC * 4) return;
  const int a = idx >> 2, r = idx & 3;
  int j, k;
  bool publish = true;
  if (a < 4) {
    j = k = 4 * r + a;
  } else if (a < 10) {
    const int w = a - 4;
    const int c1i = w < 3 ? 0 : (w < 5 ? 1 : 2);
    const int c2i = w < 3 ? w + 1 : (w < 5 ? w - 1 : 3);
    j = 4 * r + c2i;
    k = 4 * r + c1i;
  } else if (a < 26) {
    j = 4 * r + ((a - 10) >> 2);
    k = 4 * tmap[r] + ((a - 10) & 3);
  } else {
    const int w = a - 26;
    const int cc = w < 4 ? 0 : (w < 7 ? 1 : (w < 9 ? 2 : 3));
    const int dd = w < 4 ? w : (w < 7 ? w - 3 : (w < 9 ? w - 5 : 3));
    const int ro = tmap[4 + r];
    j = 4 * r + cc;
    k = 4 * ro + dd;
    if (cc == dd && r > ro) publish = false;
  }
  if (!publish || j >= K || k >= K) return;
  double m = 1.0;
  int64_t e = 0;
  int cnt = 0;
  for (int64_t ci = c0; ci < c1; ++ci) {
    const size_t o = (size_t)cell_chunks[ci] * FQ_ACC * 4 + idx;
    m *= part_m[o];
    e += part_e[o];
    if (++cnt == 512) {
      cnt = 0;
      int ee;
      m = frexp(m, &ee);
      e += ee;
    }
  }
  const int hi = j > k ? j : k, lo = j > k ? k : j;
  fll[(size_t)c * npairs + hi * (hi + 1) / 2 + lo] = (c0 == c1) ? 0.0 : log(m) + (double)e * 0.6931471805599453094;
}

}  // namespace

// quad E-step for the cell shard [c0, c0+nc) described by the chunk tables st; -1 if not applicable
int fmx_quad_estep_launch(muxgl_handle* h, muxgl_row_state* st, int64_t c0, int64_t nc) {
  if (h->K > 16 || !st || (h->flags & (MUXGL_FLAG_FORCE_TILE_SWEEP | MUXGL_FLAG_FORCE_ROW_KERNEL))) return -1;
  if (!st->d_tmap) {
    if (dev_alloc(h, &st->d_tmap, 8)) return 1;
    hipLaunchKernelGGL(fq_tmap_kernel, dim3(1), dim3(64), 0, h->stream, st->d_tmap);
    HIPCHK(h, hipGetLastError());
  }
  const size_t need = (size_t)st->n_chunks * FQ_ACC * 4;
  if (need > st->part_cap) {
    if (dev_alloc(h, &st->d_part, need)) return 1;
    st->part_cap = need;
  }
  if (need > st->part_e_cap) {
    if (dev_alloc(h, &st->d_part_e, need)) return 1;
    st->part_e_cap = need;
  }
  const size_t nq = (size_t)h->S * 48;
  if (nq > h->cgpq_cap) {
    if (dev_alloc(h, &h->d_cgpq, nq)) return 1;
    h->cgpq_cap = nq;
  }
  if (nq)
    hipLaunchKernelGGL(fmx_cgpq_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, h->stream, h->S, h->K, h->d_cgp,
                       h->d_cgpq);
  const bool use_lin = h->d_flin && h->nnz > 0 && st->n_chunks > 0 && !(h->flags & MUXGL_FLAG_NO_LINEAR_ENTRIES);
  if (use_lin && !st->d_fq_nlin) {  // once per muxgl_fmx_prepare and chunk table: every chunk's linear entries in front
    if (dev_alloc(h, &st->d_fq_lrec, (size_t)h->nnz) || dev_alloc(h, &st->d_fq_gsnp, (size_t)h->nnz) ||
        dev_alloc(h, &st->d_fq_gl6, (size_t)h->nnz * 6) || dev_alloc(h, &st->d_fq_nlin, (size_t)st->n_chunks))
      return 1;
    hipLaunchKernelGGL(fq_partition_kernel, dim3((unsigned)((st->n_chunks + 63) / 64)), dim3(64), 0, h->stream,
                       (int)st->n_chunks, st->d_chunks, h->d_flin, h->d_entry_snp, h->d_egls6, st->d_fq_lrec, st->d_fq_gsnp,
                       st->d_fq_gl6, st->d_fq_nlin);
    HIPCHK(h, hipGetLastError());
    if (quad_launch_order(h, st->d_chunks, st->d_fq_nlin, st->n_chunks, &st->d_fq_order)) return 1;
  }
  if (use_lin) {
    const size_t ne = ((size_t)h->S + 1) * 16;
    if (ne > h->ceq_cap) {
      if (dev_alloc(h, &h->d_ceq, ne)) return 1;
      h->ceq_cap = ne;
    }
    hipLaunchKernelGGL(fmx_ceq_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, h->stream, h->S, h->K, h->d_cgp, h->d_ceq);
  }
  const unsigned blocks = (unsigned)((((st->n_chunks + 15) / 16) + 7) / 8 * 8);
  tic(h, MUXGL_T_FMX_ESTEP_SWEEP);
  if (blocks)
    hipLaunchKernelGGL(fmx_estep_quad_kernel, dim3(blocks), dim3(64), 0, h->stream, st->d_chunks, (int)st->n_chunks,
                       use_lin ? st->d_fq_gsnp : h->d_entry_snp, use_lin ? st->d_fq_gl6 : h->d_egls6,
                       use_lin ? st->d_fq_lrec : (const fmx_lrec*)nullptr, use_lin ? st->d_fq_nlin : (const int32_t*)nullptr,
                       use_lin ? st->d_fq_order : (const int32_t*)nullptr, h->d_cgpq, h->d_ceq, (int32_t)h->S, st->d_part, st->d_part_e);
  toc(h, MUXGL_T_FMX_ESTEP_SWEEP);
  if (nc > 0)
    hipLaunchKernelGGL(fmx_quad_reduce_kernel, dim3((unsigned)nc), dim3(192), 0, h->stream, st->d_cell_chunk_ptr,
                       st->d_cell_chunks, st->d_part, st->d_part_e, st->d_tmap, h->K, c0, h->d_fll);
  HIPCHK(h, hipGetLastError());
  return 0;
}
