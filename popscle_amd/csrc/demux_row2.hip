// demux_row2.hip -- the demuxlet pair sweep for 16 < V <= 32 samples on the reference's default grid {a0, 0.5}:
// the row kernel (demux_row.hip) with TWO samples per lane.
//
// Reference being replaced: cmd_cram_demuxlet.cpp:655-747 (per-read pG update, floor/normalise, pair sweep).
//
// Why: the ring-of-32 wave kernel (demux_wave.hip) walks one entry per wave and pays six DPP moves for every
// hypothesis of a lone alpha (138 VALU instructions per entry, plus the pG table pass in front of it).  Here a wave is
// 4 slots x 16 lanes as in the row kernel, a slot owns one chunk (<= 128 entries of one cell), and lane j of the slot
// owns the samples j and j + 16:
//
//   * with u_x[m] = sum_l g_x[l] * pG[1][l][m] for its two samples (18 FP64 operations), a rotation of the slot's 16
//     lanes (DPP row_ror:t on the partner's two triples, 12 moves) brings FOUR pairs (a,ka) (a,kb) (b,ka) (b,kb), i.e.
//     3 moves per hypothesis instead of 6; alpha = 0.5 is symmetric, so 8 rotations cover the 496 unordered pairs
//     (the eighth visits every pair twice: one writer, see the reduce kernel) and the pair (j, j+16) is formed in the lane;
//   * 35 product accumulators per lane (mantissa * 2^exponent, prodacc), one log per chunk, chunk partials added per
//     cell in chunk order: by demux_row2_finish_kernel into a tile in LDS where the call is made at once, or by
//     demux_row2_reduce_kernel into the [C][V][V][A] tensor when the caller asks for it;
//   * phase 1 of every 16-entry batch is lane <-> entry (a4, a5), exactly the row kernel's.
//
// 75 VALU instructions per entry (258 per four entries in the inner loop, 96 of them DPP moves) and no table pass.
// Measured at 10 k cells x 50 k SNPs: sweep 1.8-1.9 ms, pass 2.05 ms at V = 17 and 2.14 ms at V = 32, against 4.0 ms for
// the ring of 32 (3.25 ms sweep + 0.54 ms table pass).  246 VGPRs, two waves per SIMD; three waves (168 VGPRs, the
// epilogue spilling) measured slower (2.4 ms).
#include "common.hpp"
#include "demux_call_body.hpp"
#include "demux_entry.hpp"
#include "row2.hpp"

namespace {

struct row2_alpha {
  double a[2];  // [0] = the singlet slot's alpha, [1] = 0.5
};

constexpr int R2_NACC = ROW2_NACC;  // accumulator layout: row2.hpp
constexpr int R2_PGS = 18;                    // doubles per entry in LDS (two alphas x 9)
constexpr int R2_SLOT_STRIDE = 16 * R2_PGS + 4;  // +4 doubles: the 4 slots' broadcast reads fall on distinct banks

__global__ void __launch_bounds__(64, 2)
    demux_row2_kernel(const row_chunk* __restrict__ chunks, int n_chunks, const int32_t* __restrict__ entry_snp,
                      const int64_t* __restrict__ entry_rptr, const uint8_t* __restrict__ reads,
                      const double* __restrict__ gp, const uint8_t* __restrict__ has_gp,
                      const double* __restrict__ lut_g, int V, row2_alpha al, double* __restrict__ part) {
  __shared__ double lut[384];
  __shared__ __align__(16) double pgs[4 * R2_SLOT_STRIDE];
  __shared__ int32_t snps[64];

  const int lane = threadIdx.x;
  const int slot = lane >> 4, j = lane & 15;
  for (int i = lane; i < 384; i += 64) lut[i] = lut_g[i];

  const int q = xcd_swizzle(blockIdx.x, gridDim.x >> 3) * 4 + slot;
  int64_t e0 = 0;
  int len = 0;
  if (q < n_chunks) {
    e0 = chunks[q].e0;
    len = chunks[q].len;
  }
  const int nb = (wave_max_i32(len) + 15) >> 4;  // trip count of the wave = its longest chunk

  double acc[R2_NACC];
  int32_t ex[R2_NACC];
#pragma unroll
  for (int a = 0; a < R2_NACC; ++a) {
    acc[a] = 1.0;
    ex[a] = 0;
  }
  const int V3 = V * 3;
  const bool live_b = j + 16 < V;  // sample j always exists (V > 16)
  const bool a0_zero = (al.a[0] == 0.0);  // wave-uniform

  // metadata of the batch to come, fetched one batch ahead in two dependent stages (as in demux_row_kernel)
  int32_t ps = -1;
  int64_t pr0 = 0, pr1 = 0;
  uint32_t pbytes = 0;
  int32_t phg = 0;
  auto fetch_meta = [&](int b) {
    const int idx = b * 16 + j;
    ps = -1;
    pr0 = pr1 = 0;
    if (idx < len) {
      const int64_t e = e0 + idx;
      ps = entry_snp[e];
      pr0 = entry_rptr[e];
      pr1 = entry_rptr[e + 1];
    }
  };
  auto fetch_dependent = [&]() {
    phg = 0;
    pbytes = 0;
    if (ps >= 0) {
      phg = has_gp[ps];
      const int64_t n = pr1 - pr0;
      if (n > 0) pbytes = reads[pr0];
      if (n > 1) pbytes |= (uint32_t)reads[pr0 + 1] << 8;
      if (n > 2) pbytes |= (uint32_t)reads[pr0 + 2] << 16;
      if (n > 3) pbytes |= (uint32_t)reads[pr0 + 3] << 24;
    }
  };
  fetch_meta(0);
  fetch_dependent();
  __syncthreads();

  for (int b = 0; b < nb; ++b) {
    // ---- phase 1: lane <-> entry (a4, a5) ----
    {
      double pG[18];
      int32_t s = ps;
      const int64_t r0 = pr0, r1 = pr1;
      const uint32_t first4 = pbytes;
      const int32_t hg = phg;
      if (b + 1 < nb) fetch_meta(b + 1);
      if (s >= 0) {
        if (hg) {
          row_entry_pg<2>(reads, r0, r1, first4, al.a, lut, pG);
        } else {
          s = -1;  // :733  marker without genotypes: contributes nothing
        }
      }
      if (s < 0) {
#pragma unroll
        for (int i = 0; i < 18; ++i) pG[i] = 1.0;  // with g = (1,0,0) every factor of a dead entry is exactly 1
      }
      double* dst = pgs + slot * R2_SLOT_STRIDE + j * R2_PGS;
#pragma unroll
      for (int i = 0; i < 18; ++i) dst[i] = pG[i];
      snps[lane] = s;
    }
    __syncthreads();

    // ---- phase 2: lane <-> two samples, 16 entries of the slot's chunk ----
    int32_t s_next = snps[slot * 16];
    double na0 = 1.0, na1 = 0.0, na2 = 0.0, nb0 = 1.0, nb1 = 0.0, nb2 = 0.0, nh0 = 1.0, nh1 = 0.0, nh2 = 0.0;
    auto fetch_row = [&]() {
      na0 = 1.0, na1 = 0.0, na2 = 0.0, nb0 = 1.0, nb1 = 0.0, nb2 = 0.0, nh0 = 1.0, nh1 = 0.0, nh2 = 0.0;
      if (s_next >= 0) {
        const double* row = gp + (size_t)s_next * V3;
        nh0 = row[0], nh1 = row[1], nh2 = row[2];
        na0 = row[j * 3], na1 = row[j * 3 + 1], na2 = row[j * 3 + 2];
        if (live_b) nb0 = row[j * 3 + 48], nb1 = row[j * 3 + 49], nb2 = row[j * 3 + 50];
      }
    };
    fetch_row();
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
      const double a0 = na0, a1 = na1, a2 = na2, b0 = nb0, b1 = nb1, b2 = nb2, h0 = nh0, h1 = nh1, h2 = nh2;
      if (i == 8 && b + 1 < nb) fetch_dependent();  // stage 2 of the next batch
      s_next = (i + 1 < 16) ? snps[slot * 16 + i + 1] : -1;
      fetch_row();  // the next entry's triples
      const double* qn = pgs + slot * R2_SLOT_STRIDE + i * R2_PGS;
      // singlet slots: llksAB[x][0][0] (:806,828) = sum_{l,m} g_x[l] g_0[m] pG[0][l][m]
      {
        const double ua0 = fma(a2, qn[6], fma(a1, qn[3], a0 * qn[0]));
        const double ub0 = fma(b2, qn[6], fma(b1, qn[3], b0 * qn[0]));
        if (a0_zero) {
          // alpha[0] == 0: the three columns of pG[0] are the same numbers (:673) and the sum factorises
          const double hs = h0 + h1 + h2;
          acc[0] *= ua0 * hs;
          acc[1] *= ub0 * hs;
        } else {
          const double ua1 = fma(a2, qn[7], fma(a1, qn[4], a0 * qn[1]));
          const double ua2 = fma(a2, qn[8], fma(a1, qn[5], a0 * qn[2]));
          const double ub1 = fma(b2, qn[7], fma(b1, qn[4], b0 * qn[1]));
          const double ub2 = fma(b2, qn[8], fma(b1, qn[5], b0 * qn[2]));
          acc[0] *= fma(h2, ua2, fma(h1, ua1, h0 * ua0));
          acc[1] *= fma(h2, ub2, fma(h1, ub1, h0 * ub0));
        }
      }
      const double* p = qn + 9;
      const double ua0 = fma(a2, p[6], fma(a1, p[3], a0 * p[0]));
      const double ua1 = fma(a2, p[7], fma(a1, p[4], a0 * p[1]));
      const double ua2 = fma(a2, p[8], fma(a1, p[5], a0 * p[2]));
      const double ub0 = fma(b2, p[6], fma(b1, p[3], b0 * p[0]));
      const double ub1 = fma(b2, p[7], fma(b1, p[4], b0 * p[1]));
      const double ub2 = fma(b2, p[8], fma(b1, p[5], b0 * p[2]));
      acc[2] *= fma(b2, ua2, fma(b1, ua1, b0 * ua0));  // the lane's own two samples (:738-746)
#define R2_STEP(T)                                                                                       \
  {                                                                                                      \
    const double ra0 = row2_ror<T>(a0), ra1 = row2_ror<T>(a1), ra2 = row2_ror<T>(a2);                          \
    const double rb0 = row2_ror<T>(b0), rb1 = row2_ror<T>(b1), rb2 = row2_ror<T>(b2);                          \
    acc[3 + 4 * (T - 1) + 0] *= fma(ra2, ua2, fma(ra1, ua1, ra0 * ua0)); /* :738-746 as a product */     \
    acc[3 + 4 * (T - 1) + 1] *= fma(rb2, ua2, fma(rb1, ua1, rb0 * ua0));                                 \
    acc[3 + 4 * (T - 1) + 2] *= fma(ra2, ub2, fma(ra1, ub1, ra0 * ub0));                                 \
    acc[3 + 4 * (T - 1) + 3] *= fma(rb2, ub2, fma(rb1, ub1, rb0 * ub0));                                 \
  }
      R2_STEP(1) R2_STEP(2) R2_STEP(3) R2_STEP(4) R2_STEP(5) R2_STEP(6) R2_STEP(7) R2_STEP(8)
#undef R2_STEP
    }
#pragma unroll
    for (int a = 0; a < R2_NACC; ++a) prodacc_renorm(acc[a], ex[a]);
    __syncthreads();
  }

  if (q < n_chunks) {
    double* out = part + (size_t)q * R2_NACC * 16;
#pragma unroll
    for (int a = 0; a < R2_NACC; ++a) out[a * 16 + j] = prodacc_log(acc[a], ex[a]);
  }
}

// adds the chunk partials of one cell, in chunk order, into ll[c][x][y][n] (+ mirror for alpha 0.5)
__global__ void __launch_bounds__(192)
    demux_row2_reduce_kernel(const int64_t* __restrict__ cell_chunk_ptr, const int32_t* __restrict__ cell_chunks,
                             const double* __restrict__ part, const int32_t* __restrict__ kmap, int V, int A,
                             double* __restrict__ ll) {
  const int64_t c = blockIdx.x;
  const int64_t c0 = cell_chunk_ptr[c], c1 = cell_chunk_ptr[c + 1];
  if (c0 == c1) return;
  double* out = ll + (size_t)c * V * V * A;
  for (int idx = threadIdx.x; idx < R2_NACC * 16; idx += blockDim.x) {
    const int a = idx >> 4, j = idx & 15;
    int x, y;
    if (!row2_pair_of(a, j, kmap, x, y)) continue;
    const bool singlet = a < 2;
    if (singlet) y = 0;  // llksAB[x][0][0]
    if (x >= V || y >= V) continue;
    double s = 0.0;
    for (int64_t ci = c0; ci < c1; ++ci) s += part[(size_t)cell_chunks[ci] * R2_NACC * 16 + idx];
    if (singlet) {
      out[((size_t)x * V) * A] = s;
    } else {
      out[((size_t)x * V + y) * A + 1] = s;
      out[((size_t)y * V + x) * A + 1] = s;
    }
  }
}

// reduce + call fused (the LL tensor is not requested): the cell's [V][V][2] tile is put together in LDS from the chunk
// partials, one wave makes the call on it (demux_call_body.hpp, 64 lanes per cell) and the record goes straight to the
// caller's pinned host buffer -- no round trip of the 164 MB tensor (10 k cells, V = 32) through HBM
__global__ void __launch_bounds__(256)
    demux_row2_finish_kernel(const int64_t* __restrict__ cell_ptr, const int64_t* __restrict__ cell_chunk_ptr,
                             const int32_t* __restrict__ cell_chunks, const double* __restrict__ part,
                             const int32_t* __restrict__ kmap, int V, muxgl_call::call_alpha al, double doublet_prior,
                             muxgl_demux_cell* __restrict__ out) {
  __shared__ double llt[32 * 32 * 2];
  __shared__ __align__(16) muxgl_demux_cell rec;
  static_assert(sizeof(muxgl_demux_cell) % 16 == 0, "records are copied out in 16-byte pieces");
  const int64_t c = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t c0 = cell_chunk_ptr[c], c1 = cell_chunk_ptr[c + 1];
  for (int t = tid; t < V * V * 2; t += 256) llt[t] = 0.0;
  __syncthreads();
  for (int idx = tid; idx < R2_NACC * 16 && c0 != c1; idx += 256) {
    const int a = idx >> 4, j = idx & 15;
    int x, y;
    if (!row2_pair_of(a, j, kmap, x, y)) continue;
    if (a < 2) y = 0;  // llksAB[x][0][0]
    if (x >= V || y >= V) continue;
    double s = 0.0;
    for (int64_t ci = c0; ci < c1; ++ci) s += part[(size_t)cell_chunks[ci] * R2_NACC * 16 + idx];
    if (a < 2) {
      llt[(x * V) * 2] = s;
    } else {
      llt[(x * V + y) * 2 + 1] = s;
      llt[(y * V + x) * 2 + 1] = s;
    }
  }
  __syncthreads();
  if (tid < 64)
    muxgl_call::demux_call_group<64>(tid, true, (int32_t)(cell_ptr[c + 1] - cell_ptr[c]), V, 2, al, doublet_prior, llt,
                                     &rec);
  __syncthreads();
  constexpr int NQ = (int)(sizeof(muxgl_demux_cell) / 16);
  if (tid < NQ) reinterpret_cast<uint4*>(out + c)[tid] = reinterpret_cast<const uint4*>(&rec)[tid];
}

}  // namespace

// returns -1 when the two-samples-per-lane row path does not apply, 0 ok, 1 error
int demux_row2_launch(muxgl_handle* h, const muxgl_demux_params* p) {
  if (h->V <= 16 || h->V > 32 || !h->row || h->C == 0) return -1;
  if (h->flags & (MUXGL_FLAG_FORCE_TILE_SWEEP | MUXGL_FLAG_FORCE_WAVE_KERNEL)) return -1;
  if (p->n_alpha != 2 || p->alpha[1] != 0.5 || p->alpha[0] == 0.5) return -1;
  muxgl_row_state* st = h->row;
  const size_t need = (size_t)st->n_chunks * R2_NACC * 16;
  if ((double)need * 8.0 > ROW2_PART_LIMIT) return -1;
  if (need > st->part_cap) {
    if (dev_alloc(h, &st->d_part, need)) return 1;
    st->part_cap = need;
  }
  if (!st->d_tmap) {  // (the quad kernel's tile map slot: unused beyond 16 samples) lane map of the nine rotations
    if (dev_alloc(h, &st->d_tmap, 9 * 16)) return 1;
    hipLaunchKernelGGL(row2_kmap_kernel, dim3(1), dim3(64), 0, h->stream, st->d_tmap);
    HIPCHK(h, hipGetLastError());
  }
  row2_alpha al = {{p->alpha[0], p->alpha[1]}};
  tic(h, MUXGL_T_DEMUX_SWEEP);
  const unsigned blocks = (unsigned)((((st->n_chunks + 3) / 4) + 7) / 8 * 8);  // multiple of 8 for xcd_swizzle
  if (blocks) {
    hipLaunchKernelGGL(demux_row2_kernel, dim3(blocks), dim3(64), 0, h->stream, st->d_chunks, (int)st->n_chunks,
                       h->d_entry_snp, h->d_entry_rptr, h->d_reads, h->d_gp, h->d_has_gp, h->d_lut, h->V, al,
                       st->d_part);
    HIPCHK(h, hipGetLastError());
  }
  toc(h, MUXGL_T_DEMUX_SWEEP);
  tic(h, MUXGL_T_DEMUX_REDUCE);
  if (h->want_full_ll) {
    hipLaunchKernelGGL(demux_row2_reduce_kernel, dim3((unsigned)h->C), dim3(192), 0, h->stream, st->d_cell_chunk_ptr,
                       st->d_cell_chunks, st->d_part, st->d_tmap, h->V, p->n_alpha, h->d_ll);
  } else {  // reduce + call fused, records written to the pinned host buffer
    const muxgl_call::call_alpha ca = muxgl_call::make_call_alpha(p, h->V);
    hipLaunchKernelGGL(demux_row2_finish_kernel, dim3((unsigned)h->C), dim3(256), 0, h->stream, h->d_cell_ptr,
                       st->d_cell_chunk_ptr, st->d_cell_chunks, st->d_part, st->d_tmap, h->V, ca, p->doublet_prior,
                       h->h_dcells);
    h->records_on_host = true;
  }
  HIPCHK(h, hipGetLastError());
  toc(h, MUXGL_T_DEMUX_REDUCE);
  return 0;
}
