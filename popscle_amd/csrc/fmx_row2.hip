// fmx_row2.hip -- freemuxlet E-step for 16 < K <= 32 clusters: the row E-step (fmx_estep_row_kernel, fmx_kernels.hip)
// with TWO clusters per lane, the counterpart of demux_row2.hip.
//
// Reference being replaced: cmd_cram_freemux2.cpp:383-456 (pair and singlet likelihoods of every droplet).
//
// A wave is 4 slots x 16 lanes, a slot owns one chunk (<= 128 entries of one cell), lane j of the slot owns the
// clusters j and j + 16.  With u_x[m] = sum_l gp_x[l] * glis[l][m] for its two clusters, a rotation of the slot's 16
// lanes (DPP row_ror:t on the partner's two posterior triples, 12 moves) brings four pairs; the entry's 3 x 3 likelihood
// matrix is symmetric, so eight rotations cover the unordered pairs (the eighth visits every pair of lanes twice: one
// writer in the reduce kernel) and the pair (j, j + 16) is formed inside the lane.  35 product accumulators per lane,
// one log per chunk (:454-455), chunk partials added per cell in chunk order.  The ring-of-32 wave kernel it replaces
// (fmx_wave.hip) walks one entry per wave and pays 6 moves per pair.
#include "common.hpp"
#include "row2.hpp"

namespace {

constexpr int F2_NACC = ROW2_NACC;  // accumulator layout: row2.hpp
constexpr int F2_PGS = 10;  // 9 likelihoods + 1 pad (16-byte aligned rows)
constexpr int F2_SLOT_STRIDE = 16 * F2_PGS + 4;

__global__ void __launch_bounds__(64, 2)
    fmx_estep_row2_kernel(const row_chunk* __restrict__ chunks, int n_chunks, const int32_t* __restrict__ entry_snp,
                          const double* __restrict__ egls, const double* __restrict__ cgp, int K,
                          double* __restrict__ part) {
  __shared__ __align__(16) double gl[4 * F2_SLOT_STRIDE];
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
  double acc[F2_NACC];
  int32_t ex[F2_NACC];
#pragma unroll
  for (int a = 0; a < F2_NACC; ++a) {
    acc[a] = 1.0;
    ex[a] = 0;
  }
  const int K3 = K * 3;
  const bool live_b = j + 16 < K;  // cluster j always exists (K > 16)

  for (int b = 0; b < nb; ++b) {
    {  // phase 1: lane <-> entry, the 3x3 likelihoods of 64 entries into LDS
      const int idx = b * 16 + j;
      double* dst = gl + slot * F2_SLOT_STRIDE + j * F2_PGS;
      int32_t s = -1;
      if (idx < len) {
        const int64_t e = e0 + idx;
        s = entry_snp[e];
        const double* src = egls + (size_t)e * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) dst[i] = src[i];
      } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) dst[i] = 1.0;  // dead entry: with g = (1,0,0) every factor is exactly 1
      }
      snps[lane] = s;
    }
    __syncthreads();
    int32_t s_next = snps[slot * 16];
    double na0, na1, na2, nb0, nb1, nb2;
    auto fetch_row = [&]() {
      na0 = 1.0, na1 = 0.0, na2 = 0.0, nb0 = 1.0, nb1 = 0.0, nb2 = 0.0;
      if (s_next >= 0) {
        const double* row = cgp + (size_t)s_next * K3 + j * 3;
        na0 = row[0], na1 = row[1], na2 = row[2];
        if (live_b) nb0 = row[48], nb1 = row[49], nb2 = row[50];
      }
    };
    fetch_row();
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
      const double a0 = na0, a1 = na1, a2 = na2, b0 = nb0, b1 = nb1, b2 = nb2;
      s_next = (i + 1 < 16) ? snps[slot * 16 + i + 1] : -1;
      fetch_row();  // the next entry's posteriors
      const double* p = gl + slot * F2_SLOT_STRIDE + i * F2_PGS;
      // singlets: sum_g glis[g,g] * gp_x[g]   (cmd_cram_freemux2.cpp:448-452)
      acc[0] *= fma(a2, p[8], fma(a1, p[4], a0 * p[0]));
      acc[1] *= fma(b2, p[8], fma(b1, p[4], b0 * p[0]));
      // pairs: sum_{g1,g2} glis[g1,g2] gp_x[g1] gp_y[g2]   (:440-446)
      const double ua0 = fma(a2, p[6], fma(a1, p[3], a0 * p[0]));
      const double ua1 = fma(a2, p[7], fma(a1, p[4], a0 * p[1]));
      const double ua2 = fma(a2, p[8], fma(a1, p[5], a0 * p[2]));
      const double ub0 = fma(b2, p[6], fma(b1, p[3], b0 * p[0]));
      const double ub1 = fma(b2, p[7], fma(b1, p[4], b0 * p[1]));
      const double ub2 = fma(b2, p[8], fma(b1, p[5], b0 * p[2]));
      acc[2] *= fma(b2, ua2, fma(b1, ua1, b0 * ua0));
#define F2_STEP(T)                                                                  \
  {                                                                                 \
    const double ra0 = row2_ror<T>(a0), ra1 = row2_ror<T>(a1), ra2 = row2_ror<T>(a2);     \
    const double rb0 = row2_ror<T>(b0), rb1 = row2_ror<T>(b1), rb2 = row2_ror<T>(b2);     \
    acc[3 + 4 * (T - 1) + 0] *= fma(ra2, ua2, fma(ra1, ua1, ra0 * ua0));            \
    acc[3 + 4 * (T - 1) + 1] *= fma(rb2, ua2, fma(rb1, ua1, rb0 * ua0));            \
    acc[3 + 4 * (T - 1) + 2] *= fma(ra2, ub2, fma(ra1, ub1, ra0 * ub0));            \
    acc[3 + 4 * (T - 1) + 3] *= fma(rb2, ub2, fma(rb1, ub1, rb0 * ub0));            \
  }
      F2_STEP(1) F2_STEP(2) F2_STEP(3) F2_STEP(4) F2_STEP(5) F2_STEP(6) F2_STEP(7) F2_STEP(8)
#undef F2_STEP
    }
#pragma unroll
    for (int a = 0; a < F2_NACC; ++a) prodacc_renorm(acc[a], ex[a]);
    __syncthreads();
  }
  if (q < n_chunks) {
    double* out = part + (size_t)q * F2_NACC * 16;
#pragma unroll
    for (int a = 0; a < F2_NACC; ++a) out[a * 16 + j] = prodacc_log(acc[a], ex[a]);  // :454-455 as one log per chunk
  }
}

// adds the chunk partials of one cell, in chunk order, into the packed triangle fll[c][hi (hi+1)/2 + lo]
__global__ void __launch_bounds__(192)
    fmx_estep_row2_reduce_kernel(const int64_t* __restrict__ cell_chunk_ptr, const int32_t* __restrict__ cell_chunks,
                                 const double* __restrict__ part, const int32_t* __restrict__ kmap, int K,
                                 int64_t c_off, double* __restrict__ fll) {
  const int64_t c = c_off + blockIdx.x;
  const int64_t c0 = cell_chunk_ptr[c], c1 = cell_chunk_ptr[c + 1];
  const int npairs = K * (K + 1) / 2;
  for (int idx = threadIdx.x; idx < F2_NACC * 16; idx += blockDim.x) {
    const int a = idx >> 4, j = idx & 15;
    int x, y;
    if (!row2_pair_of(a, j, kmap, x, y)) continue;
    if (x >= K || y >= K) continue;
    double s = 0.0;
    for (int64_t ci = c0; ci < c1; ++ci) s += part[(size_t)cell_chunks[ci] * F2_NACC * 16 + idx];
    const int hi = x > y ? x : y, lo = x > y ? y : x;
    fll[(size_t)c * npairs + hi * (hi + 1) / 2 + lo] = s;
  }
}

}  // namespace

// E-step with two clusters per lane for the cell shard [c0, c0+nc) described by the chunk tables st; -1 if not applicable
int fmx_row2_estep_launch(muxgl_handle* h, muxgl_row_state* st, int64_t c0, int64_t nc) {
  if (h->K <= 16 || h->K > 32 || !st) return -1;
  if (h->flags & (MUXGL_FLAG_FORCE_TILE_SWEEP | MUXGL_FLAG_FORCE_WAVE_KERNEL)) return -1;
  if (!st->d_tmap) {  // (the quad tile-map slot of this table set is unused beyond 16 clusters) lane map of the rotations
    if (dev_alloc(h, &st->d_tmap, 9 * 16)) return 1;
    hipLaunchKernelGGL(row2_kmap_kernel, dim3(1), dim3(64), 0, h->stream, st->d_tmap);
    HIPCHK(h, hipGetLastError());
  }
  const size_t need = (size_t)st->n_chunks * F2_NACC * 16;
  if ((double)need * 8.0 > ROW2_PART_LIMIT) return -1;
  if (need > st->part_cap) {
    if (dev_alloc(h, &st->d_part, need)) return 1;
    st->part_cap = need;
  }
  const unsigned blocks = (unsigned)((((st->n_chunks + 3) / 4) + 7) / 8 * 8);  // multiple of 8 for xcd_swizzle
  if (blocks)
    hipLaunchKernelGGL(fmx_estep_row2_kernel, dim3(blocks), dim3(64), 0, h->stream, st->d_chunks, (int)st->n_chunks,
                       h->d_entry_snp, h->d_egls, h->d_cgp, h->K, st->d_part);
  if (nc > 0)
    hipLaunchKernelGGL(fmx_estep_row2_reduce_kernel, dim3((unsigned)nc), dim3(192), 0, h->stream, st->d_cell_chunk_ptr,
                       st->d_cell_chunks, st->d_part, st->d_tmap, h->K, c0, h->d_fll);
  HIPCHK(h, hipGetLastError());
  return 0;
}
