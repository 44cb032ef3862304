// fmx_rowx.hip -- freemuxlet E-step for 17 .. 24 clusters: the row E-step (fmx_estep_row_kernel, fmx_kernels.hip) for
// the first sixteen clusters plus the remaining NB = K - 16 <= 8 clusters as broadcast operands -- the counterpart of
// demux_rowx.hip.
//
// Reference being replaced: cmd_cram_freemux2.cpp:383-456 (pair and singlet likelihoods of every droplet).
//
// A wave is 4 slots x 16 lanes, a slot owns one chunk (<= 128 entries of one cell), lane j owns cluster j < 16.  The
// posterior triples of the clusters 16 .. K-1 ("b" clusters) are staged in LDS by phase 1 next to the entry's 3 x 3
// likelihoods (lane <-> entry: NB x 24 contiguous bytes of the SNP's posterior row); pair (j, 16+m) multiplies b_m's
// triple, read as an LDS broadcast, into u_j; lanes j < NB also own b_j for its singlet and the pairs among the b
// clusters (ring offsets 1 .. NB/2, per-lane LDS reads); the pairs among the first sixteen come from eight row_ror:t
// rotations of the lane's own triple.  10 + NB + NB/2 product accumulators per lane (layout: row2.hpp), one log per
// chunk (:454-455), chunk partials added per cell in chunk order.
#include "common.hpp"
#include "row2.hpp"

namespace {

constexpr int FX2_PGS = 10;  // 9 likelihoods + 1 pad (16-byte aligned rows)
constexpr int FX2_SLOT_STRIDE = 16 * FX2_PGS + 4;

template <int NB>
__global__ void __launch_bounds__(64, 2)
    fmx_estep_rowx_kernel(const row_chunk* __restrict__ chunks, int n_chunks, const int32_t* __restrict__ entry_snp,
                          const double* __restrict__ egls, const double* __restrict__ cgp,
                          double* __restrict__ part) {
  constexpr int K = 16 + NB, K3 = K * 3, HB = NB / 2, NACC = rowx_nacc(NB);
  constexpr int BT = NB * 3;  // doubles of the b triples of one entry
  constexpr int BSLOT = 16 * BT;
  __shared__ __align__(16) double gl[4 * FX2_SLOT_STRIDE];
  __shared__ double bts[4 * BSLOT];
  __shared__ int32_t snps[64];
  const int lane = threadIdx.x;
  const int slot = lane >> 4, j = lane & 15;
  const int q = xcd_swizzle(blockIdx.x, gridDim.x >> 3) * 4 + slot;
  int64_t e0 = 0;
  int len = 0;
  if (q < n_chunks) {
    e0 = chunks[q].e0;
    len = chunks[q].len;
  }
  const int nb = (wave_max_i32(len) + 15) >> 4;
  double acc[NACC];
  int32_t ex[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    acc[a] = 1.0;
    ex[a] = 0;
  }
  const int jb = j < NB ? j : 0;  // the b cluster this lane owns (lanes beyond NB shadow lane 0; never published)

  for (int b = 0; b < nb; ++b) {
    {  // phase 1: lane <-> entry, the 3x3 likelihoods and the b clusters' posteriors of 64 entries into LDS
      const int idx = b * 16 + j;
      double* dst = gl + slot * FX2_SLOT_STRIDE + j * FX2_PGS;
      double* bd = bts + slot * BSLOT + j * BT;
      int32_t s = -1;
      if (idx < len) {
        const int64_t e = e0 + idx;
        s = entry_snp[e];
        const double* src = egls + (size_t)e * 9;
        const double* row = cgp + (size_t)s * K3 + 48;
#pragma unroll
        for (int i = 0; i < 9; ++i) dst[i] = src[i];
#pragma unroll
        for (int i = 0; i < BT; ++i) bd[i] = row[i];
      } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) dst[i] = 1.0;  // dead entry: with g = (1,0,0) every factor is exactly 1
#pragma unroll
        for (int i = 0; i < BT; ++i) bd[i] = (i % 3 == 0) ? 1.0 : 0.0;
      }
      snps[lane] = s;
    }
    __syncthreads();
    int32_t s_next = snps[slot * 16];
    double na0, na1, na2;
    auto fetch_row = [&]() {
      na0 = 1.0, na1 = 0.0, na2 = 0.0;
      if (s_next >= 0) {
        const double* row = cgp + (size_t)s_next * K3 + j * 3;
        na0 = row[0], na1 = row[1], na2 = row[2];
      }
    };
    fetch_row();
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
      const double a0 = na0, a1 = na1, a2 = na2;
      s_next = (i + 1 < 16) ? snps[slot * 16 + i + 1] : -1;
      fetch_row();  // the next entry's posteriors
      const double* p = gl + slot * FX2_SLOT_STRIDE + i * FX2_PGS;
      const double* bq = bts + slot * BSLOT + i * BT;
      const double b0 = bq[jb * 3], b1 = bq[jb * 3 + 1], b2 = bq[jb * 3 + 2];  // the lane's own b cluster
      // singlets: sum_g glis[g,g] * gp_x[g]   (cmd_cram_freemux2.cpp:448-452)
      acc[0] *= fma(a2, p[8], fma(a1, p[4], a0 * p[0]));
      acc[9 + NB] *= fma(b2, p[8], fma(b1, p[4], b0 * p[0]));
      // pairs: sum_{g1,g2} glis[g1,g2] gp_x[g1] gp_y[g2]   (:440-446)
      const double ua0 = fma(a2, p[6], fma(a1, p[3], a0 * p[0]));
      const double ua1 = fma(a2, p[7], fma(a1, p[4], a0 * p[1]));
      const double ua2 = fma(a2, p[8], fma(a1, p[5], a0 * p[2]));
#pragma unroll
      for (int m = 0; m < NB; ++m)  // (j, 16 + m): b_m as an LDS broadcast
        acc[9 + m] *= fma(bq[m * 3 + 2], ua2, fma(bq[m * 3 + 1], ua1, bq[m * 3] * ua0));
      if (HB > 0) {  // pairs among the b clusters, from the lanes that own one
        const double ub0 = fma(b2, p[6], fma(b1, p[3], b0 * p[0]));
        const double ub1 = fma(b2, p[7], fma(b1, p[4], b0 * p[1]));
        const double ub2 = fma(b2, p[8], fma(b1, p[5], b0 * p[2]));
#pragma unroll
        for (int d = 1; d <= HB; ++d) {
          const int pm = (jb + d >= NB) ? jb + d - NB : jb + d;
          acc[9 + NB + d] *= fma(bq[pm * 3 + 2], ub2, fma(bq[pm * 3 + 1], ub1, bq[pm * 3] * ub0));
        }
      }
      // pairs among the first sixteen: eight rotations of the lane's own triple
#define FX2_STEP(T)                                                                         \
  {                                                                                         \
    const double ra0 = row2_ror<T>(a0), ra1 = row2_ror<T>(a1), ra2 = row2_ror<T>(a2);       \
    acc[T] *= fma(ra2, ua2, fma(ra1, ua1, ra0 * ua0));                                      \
  }
      FX2_STEP(1) FX2_STEP(2) FX2_STEP(3) FX2_STEP(4) FX2_STEP(5) FX2_STEP(6) FX2_STEP(7) FX2_STEP(8)
#undef FX2_STEP
    }
#pragma unroll
    for (int a = 0; a < NACC; ++a) prodacc_renorm(acc[a], ex[a]);
    __syncthreads();
  }
  if (q < n_chunks) {
    double* out = part + (size_t)q * NACC * 16;
#pragma unroll
    for (int a = 0; a < NACC; ++a) out[a * 16 + j] = prodacc_log(acc[a], ex[a]);  // :454-455 as one log per chunk
  }
}

// adds the chunk partials of one cell, in chunk order, into the packed triangle fll[c][hi (hi+1)/2 + lo]
__global__ void __launch_bounds__(192)
    fmx_estep_rowx_reduce_kernel(const int64_t* __restrict__ cell_chunk_ptr, const int32_t* __restrict__ cell_chunks,
                                 const double* __restrict__ part, const int32_t* __restrict__ kmap, int K,
                                 int64_t c_off, double* __restrict__ fll) {
  const int64_t c = c_off + blockIdx.x;
  const int64_t c0 = cell_chunk_ptr[c], c1 = cell_chunk_ptr[c + 1];
  const int npairs = K * (K + 1) / 2, NB = K - 16, nacc = rowx_nacc(NB);
  for (int idx = threadIdx.x; idx < nacc * 16; idx += blockDim.x) {
    const int a = idx >> 4, j = idx & 15;
    int x, y;
    if (!rowx_pair_of(a, j, NB, kmap, x, y)) continue;
    if (y < 0) y = x;  // singlet: the diagonal of the triangle
    double s = 0.0;
    for (int64_t ci = c0; ci < c1; ++ci) s += part[(size_t)cell_chunks[ci] * nacc * 16 + idx];
    const int hi = x > y ? x : y, lo = x > y ? y : x;
    fll[(size_t)c * npairs + hi * (hi + 1) / 2 + lo] = s;
  }
}

template <int NB>
void fmx_rowx_sweep(muxgl_handle* h, muxgl_row_state* st, unsigned blocks) {
  hipLaunchKernelGGL(fmx_estep_rowx_kernel<NB>, dim3(blocks), dim3(64), 0, h->stream, st->d_chunks, (int)st->n_chunks,
                     h->d_entry_snp, h->d_egls, h->d_cgp, st->d_part);
}

}  // namespace

// E-step for 17..24 clusters and the cell shard [c0, c0+nc) described by the chunk tables st; -1 if not applicable
int fmx_rowx_estep_launch(muxgl_handle* h, muxgl_row_state* st, int64_t c0, int64_t nc) {
  if (h->K <= 16 || h->K > 16 + ROWX_MAX_NB || !st) return -1;
  if (h->flags & (MUXGL_FLAG_FORCE_TILE_SWEEP | MUXGL_FLAG_FORCE_WAVE_KERNEL | MUXGL_FLAG_FORCE_ROW_KERNEL)) return -1;
  const int NB = h->K - 16;
  const size_t need = (size_t)st->n_chunks * rowx_nacc(NB) * 16;
  if ((double)need * 8.0 > ROW2_PART_LIMIT) return -1;
  if (!st->d_tmap) {  // (the quad tile-map slot of this table set is unused beyond 16 clusters) lane map of the rotations
    if (dev_alloc(h, &st->d_tmap, 9 * 16)) return 1;
    hipLaunchKernelGGL(row2_kmap_kernel, dim3(1), dim3(64), 0, h->stream, st->d_tmap);
    HIPCHK(h, hipGetLastError());
  }
  if (need > st->part_cap) {
    if (dev_alloc(h, &st->d_part, need)) return 1;
    st->part_cap = need;
  }
  const unsigned blocks = (unsigned)((((st->n_chunks + 3) / 4) + 7) / 8 * 8);  // multiple of 8 for xcd_swizzle
  if (blocks) {
    switch (NB) {
      case 1: fmx_rowx_sweep<1>(h, st, blocks); break;
      case 2: fmx_rowx_sweep<2>(h, st, blocks); break;
      case 3: fmx_rowx_sweep<3>(h, st, blocks); break;
      case 4: fmx_rowx_sweep<4>(h, st, blocks); break;
      case 5: fmx_rowx_sweep<5>(h, st, blocks); break;
      case 6: fmx_rowx_sweep<6>(h, st, blocks); break;
      case 7: fmx_rowx_sweep<7>(h, st, blocks); break;
      default: fmx_rowx_sweep<8>(h, st, blocks); break;
    }
  }
  if (nc > 0)
    hipLaunchKernelGGL(fmx_estep_rowx_reduce_kernel, dim3((unsigned)nc), dim3(192), 0, h->stream, st->d_cell_chunk_ptr,
                       st->d_cell_chunks, st->d_part, st->d_tmap, h->K, c0, h->d_fll);
  HIPCHK(h, hipGetLastError());
  return 0;
}
