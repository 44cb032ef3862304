// demux_rowx.hip -- the demuxlet pair sweep for 17 .. 24 samples on the reference's default grid {0, 0.5}: the row
// kernel (demux_row.hip) for the first sixteen samples plus the remaining NB = V - 16 <= 8 samples as broadcast
// operands.
//
// Reference being replaced: cmd_cram_demuxlet.cpp:655-747 (per-read pG update, floor/normalise, pair sweep).
//
// Why: with two samples per lane (demux_row2.hip) 17 samples cost what 32 cost -- the lanes whose second sample does
// not exist still step.  Here a wave is 4 slots x 16 lanes as in the row kernel (a slot owns a chunk of <= 128 entries
// of one cell, lane j owns sample j < 16), and the samples 16 .. V-1 ("b" samples) never occupy a rotating register:
//   * their GP triples are staged in LDS by phase 1 (lane <-> entry: NB x 24 contiguous bytes of the entry's GP row);
//   * pair (j, 16+m): every lane reads b_m's triple as an LDS broadcast and multiplies it into u_j -- no DPP moves;
//   * pairs among the b samples and their singlets: lane j < NB also owns b_j (u of its own) and meets b_(j+d mod NB),
//     d = 1 .. NB/2, through per-lane LDS reads (for even NB the last offset visits every pair twice: one writer);
//   * pairs among the first sixteen: eight row_ror:t rotations of the lane's own triple, as in the row kernel.
// 10 + NB + NB/2 product accumulators per lane; ~110 (V = 17) .. ~155 (V = 24) VALU instructions per four entries
// against 258 with two samples per lane.  The staged triples cost 1.5 KB of LDS per b sample and wave: with eight of
// them a one-wave workgroup holds 19.8 KB, which still lets eight workgroups share a CU (two waves per SIMD); beyond
// that, and for a singlet alpha other than 0, demux_row2_kernel takes over.  As in the quad kernel the per-entry
// likelihoods are kept in their eight distinct values (alpha 0: p = l/2; alpha 0.5: p = (l+m)/4), which is what makes
// room for the triples.  Phase 1, chunk partials, reduce / fused finish: as in demux_row2.hip.
#include "common.hpp"
#include "demux_call_body.hpp"
#include "demux_entry.hpp"
#include "row2.hpp"

namespace {

struct rowx_alpha {
  double a[2];  // {0, 0.5}
};

constexpr int RX_MAX_NB = ROWX_MAX_NB;  // accumulator layout: row2.hpp
constexpr int RX_PGS = 8;                        // doubles per entry in LDS: q0[l] (alpha 0), q1[l+m] (alpha 0.5)
constexpr int RX_SLOT_STRIDE = 16 * RX_PGS + 4;  // +4 doubles: the 4 slots' broadcast reads fall on distinct banks

template <int NB>
__global__ void __launch_bounds__(64, 2)
    demux_rowx_kernel(const row_chunk* __restrict__ chunks, int n_chunks, const int32_t* __restrict__ entry_snp,
                      const int64_t* __restrict__ entry_rptr, const uint8_t* __restrict__ reads,
                      const double* __restrict__ gp, const uint8_t* __restrict__ has_gp,
                      const double* __restrict__ lut_g, rowx_alpha al, double* __restrict__ part) {
  constexpr int V = 16 + NB, V3 = V * 3, HB = NB / 2, NACC = rowx_nacc(NB);
  constexpr int BT = NB * 3;                     // doubles of the b triples of one entry
  constexpr int BSLOT = 16 * BT;
  __shared__ double lut[384];
  __shared__ __align__(16) double pgs[4 * RX_SLOT_STRIDE];
  __shared__ double bts[4 * BSLOT];
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

  double acc[NACC];
  int32_t ex[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    acc[a] = 1.0;
    ex[a] = 0;
  }
  const int jb = j < NB ? j : 0;          // the b sample this lane owns (lanes beyond NB shadow lane 0; never published)

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
    // ---- phase 1: lane <-> entry (a4, a5), and the entry's b triples into LDS ----
    {
      double pG[18];
      double bt[BT];
      int32_t s = ps;
      const int64_t r0 = pr0, r1 = pr1;
      const uint32_t first4 = pbytes;
      const int32_t hg = phg;
      if (b + 1 < nb) fetch_meta(b + 1);
      if (s >= 0 && !hg) s = -1;  // :733  marker without genotypes: contributes nothing
      if (s >= 0) {
        const double* row = gp + (size_t)s * V3 + 48;
#pragma unroll
        for (int i = 0; i < BT; ++i) bt[i] = row[i];  // (in flight while the reads are turned into likelihoods)
        row_entry_pg<2>(reads, r0, r1, first4, al.a, lut, pG);
      } else {
#pragma unroll
        for (int i = 0; i < 18; ++i) pG[i] = 1.0;  // with g = (1,0,0) every factor of a dead entry is exactly 1
#pragma unroll
        for (int i = 0; i < BT; ++i) bt[i] = (i % 3 == 0) ? 1.0 : 0.0;
      }
      // alpha 0: p = l/2 whatever m (:673), the three columns of pG[0] are the same numbers; alpha 0.5: p = (l+m)/4
      double* dst = pgs + slot * RX_SLOT_STRIDE + j * RX_PGS;
      dst[0] = pG[0];
      dst[1] = pG[3];
      dst[2] = pG[6];
      dst[3] = pG[9];
      dst[4] = pG[10];
      dst[5] = pG[11];
      dst[6] = pG[14];
      dst[7] = pG[17];
      double* bd = bts + slot * BSLOT + j * BT;
#pragma unroll
      for (int i = 0; i < BT; ++i) bd[i] = bt[i];
      snps[lane] = s;
    }
    __syncthreads();

    // ---- phase 2: lane <-> sample j (and b sample jb), 16 entries of the slot's chunk ----
    int32_t s_next = snps[slot * 16];
    double na0, na1, na2, nh0, nh1, nh2;
    auto fetch_row = [&]() {
      na0 = 1.0, na1 = 0.0, na2 = 0.0, nh0 = 1.0, nh1 = 0.0, nh2 = 0.0;
      if (s_next >= 0) {
        const double* row = gp + (size_t)s_next * V3;
        nh0 = row[0], nh1 = row[1], nh2 = row[2];
        na0 = row[j * 3], na1 = row[j * 3 + 1], na2 = row[j * 3 + 2];
      }
    };
    fetch_row();
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
      const double a0 = na0, a1 = na1, a2 = na2, h0 = nh0, h1 = nh1, h2 = nh2;
      if (i == 8 && b + 1 < nb) fetch_dependent();  // stage 2 of the next batch
      s_next = (i + 1 < 16) ? snps[slot * 16 + i + 1] : -1;
      fetch_row();  // the next entry's triples
      const double* qn = pgs + slot * RX_SLOT_STRIDE + i * RX_PGS;
      const double* bq = bts + slot * BSLOT + i * BT;
      const double b0 = bq[jb * 3], b1 = bq[jb * 3 + 1], b2 = bq[jb * 3 + 2];  // the lane's own b sample
      const double q0 = qn[0], q1 = qn[1], q2 = qn[2], p0 = qn[3], p1 = qn[4], p2 = qn[5], p3 = qn[6], p4 = qn[7];
      // singlet slots: llksAB[x][0][0] (:806,828) = sum_{l,m} g_x[l] g_0[m] pG[0][l][m] = (sum_l g_x[l] q_l) (sum_m g_0[m])
      {
        const double hs = h0 + h1 + h2;
        acc[0] *= fma(a2, q2, fma(a1, q1, a0 * q0)) * hs;
        acc[9 + NB] *= fma(b2, q2, fma(b1, q1, b0 * q0)) * hs;
      }
      // u_x[m] = sum_l g_x[l] pG[1][l][m], pG[1][l][m] = p_(l+m)
      const double ua0 = fma(a2, p2, fma(a1, p1, a0 * p0));
      const double ua1 = fma(a2, p3, fma(a1, p2, a0 * p1));
      const double ua2 = fma(a2, p4, fma(a1, p3, a0 * p2));
      // pairs (j, 16 + m): b_m as an LDS broadcast (:738-746 as a product)
#pragma unroll
      for (int m = 0; m < NB; ++m)
        acc[9 + m] *= fma(bq[m * 3 + 2], ua2, fma(bq[m * 3 + 1], ua1, bq[m * 3] * ua0));
      // pairs among the b samples, from the lanes that own one
      if (HB > 0) {
        const double ub0 = fma(b2, p2, fma(b1, p1, b0 * p0));
        const double ub1 = fma(b2, p3, fma(b1, p2, b0 * p1));
        const double ub2 = fma(b2, p4, fma(b1, p3, b0 * p2));
#pragma unroll
        for (int d = 1; d <= HB; ++d) {
          const int pm = (jb + d >= NB) ? jb + d - NB : jb + d;
          acc[9 + NB + d] *= fma(bq[pm * 3 + 2], ub2, fma(bq[pm * 3 + 1], ub1, bq[pm * 3] * ub0));
        }
      }
      // pairs among the first sixteen: eight rotations of the lane's own triple
#define RX_STEP(T)                                                                          \
  {                                                                                         \
    const double ra0 = row2_ror<T>(a0), ra1 = row2_ror<T>(a1), ra2 = row2_ror<T>(a2);       \
    acc[T] *= fma(ra2, ua2, fma(ra1, ua1, ra0 * ua0));                                      \
  }
      RX_STEP(1) RX_STEP(2) RX_STEP(3) RX_STEP(4) RX_STEP(5) RX_STEP(6) RX_STEP(7) RX_STEP(8)
#undef RX_STEP
    }
#pragma unroll
    for (int a = 0; a < NACC; ++a) prodacc_renorm(acc[a], ex[a]);
    __syncthreads();
  }

  if (q < n_chunks) {
    double* out = part + (size_t)q * NACC * 16;
#pragma unroll
    for (int a = 0; a < NACC; ++a) out[a * 16 + j] = prodacc_log(acc[a], ex[a]);
  }
}

// adds the chunk partials of one cell, in chunk order, into ll[c][x][y][n] (+ mirror for alpha 0.5)
__global__ void __launch_bounds__(192)
    demux_rowx_reduce_kernel(const int64_t* __restrict__ cell_chunk_ptr, const int32_t* __restrict__ cell_chunks,
                             const double* __restrict__ part, const int32_t* __restrict__ kmap, int V, int A,
                             double* __restrict__ ll) {
  const int64_t c = blockIdx.x;
  const int64_t c0 = cell_chunk_ptr[c], c1 = cell_chunk_ptr[c + 1];
  if (c0 == c1) return;
  const int NB = V - 16, nacc = rowx_nacc(NB);
  double* out = ll + (size_t)c * V * V * A;
  for (int idx = threadIdx.x; idx < nacc * 16; idx += blockDim.x) {
    const int a = idx >> 4, j = idx & 15;
    int x, y;
    if (!rowx_pair_of(a, j, NB, kmap, x, y)) continue;
    double s = 0.0;
    for (int64_t ci = c0; ci < c1; ++ci) s += part[(size_t)cell_chunks[ci] * nacc * 16 + idx];
    if (y < 0) {
      out[((size_t)x * V) * A] = s;  // llksAB[x][0][0]
    } else {
      out[((size_t)x * V + y) * A + 1] = s;
      out[((size_t)y * V + x) * A + 1] = s;
    }
  }
}

// reduce + call fused (the LL tensor is not requested), see demux_row2_finish_kernel
__global__ void __launch_bounds__(256)
    demux_rowx_finish_kernel(const int64_t* __restrict__ cell_ptr, const int64_t* __restrict__ cell_chunk_ptr,
                             const int32_t* __restrict__ cell_chunks, const double* __restrict__ part,
                             const int32_t* __restrict__ kmap, int V, muxgl_call::call_alpha al, double doublet_prior,
                             muxgl_demux_cell* __restrict__ out) {
  __shared__ double llt[(16 + RX_MAX_NB) * (16 + RX_MAX_NB) * 2];
  __shared__ __align__(16) muxgl_demux_cell rec;
  static_assert(sizeof(muxgl_demux_cell) % 16 == 0, "records are copied out in 16-byte pieces");
  const int64_t c = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t c0 = cell_chunk_ptr[c], c1 = cell_chunk_ptr[c + 1];
  const int NB = V - 16, nacc = rowx_nacc(NB);
  for (int t = tid; t < V * V * 2; t += 256) llt[t] = 0.0;
  __syncthreads();
  for (int idx = tid; idx < nacc * 16 && c0 != c1; idx += 256) {
    const int a = idx >> 4, j = idx & 15;
    int x, y;
    if (!rowx_pair_of(a, j, NB, kmap, x, y)) continue;
    double s = 0.0;
    for (int64_t ci = c0; ci < c1; ++ci) s += part[(size_t)cell_chunks[ci] * nacc * 16 + idx];
    if (y < 0) {
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

template <int NB>
void rowx_sweep(muxgl_handle* h, muxgl_row_state* st, unsigned blocks, const rowx_alpha& al) {
  hipLaunchKernelGGL(demux_rowx_kernel<NB>, dim3(blocks), dim3(64), 0, h->stream, st->d_chunks, (int)st->n_chunks,
                     h->d_entry_snp, h->d_entry_rptr, h->d_reads, h->d_gp, h->d_has_gp, h->d_lut, al, st->d_part);
}

}  // namespace

// returns -1 when the path does not apply, 0 ok, 1 error
int demux_rowx_launch(muxgl_handle* h, const muxgl_demux_params* p) {
  if (h->V <= 16 || h->V > 16 + RX_MAX_NB || !h->row || h->C == 0) return -1;
  if (h->flags & (MUXGL_FLAG_FORCE_TILE_SWEEP | MUXGL_FLAG_FORCE_WAVE_KERNEL | MUXGL_FLAG_FORCE_ROW_KERNEL)) return -1;
  if (p->n_alpha != 2 || p->alpha[0] != 0.0 || p->alpha[1] != 0.5) return -1;
  muxgl_row_state* st = h->row;
  const int NB = h->V - 16;
  const size_t need = (size_t)st->n_chunks * rowx_nacc(NB) * 16;
  if ((double)need * 8.0 > ROW2_PART_LIMIT) return -1;
  if (!st->d_tmap) {  // (the quad tile-map slot of this table set is unused beyond 16 samples) lane map of the rotations
    if (dev_alloc(h, &st->d_tmap, 9 * 16)) return 1;
    hipLaunchKernelGGL(row2_kmap_kernel, dim3(1), dim3(64), 0, h->stream, st->d_tmap);
    HIPCHK(h, hipGetLastError());
  }
  if (need > st->part_cap) {
    if (dev_alloc(h, &st->d_part, need)) return 1;
    st->part_cap = need;
  }
  rowx_alpha al = {{p->alpha[0], p->alpha[1]}};
  tic(h, MUXGL_T_DEMUX_SWEEP);
  const unsigned blocks = (unsigned)((((st->n_chunks + 3) / 4) + 7) / 8 * 8);  // multiple of 8 for xcd_swizzle
  if (blocks) {
    switch (NB) {
      case 1: rowx_sweep<1>(h, st, blocks, al); break;
      case 2: rowx_sweep<2>(h, st, blocks, al); break;
      case 3: rowx_sweep<3>(h, st, blocks, al); break;
      case 4: rowx_sweep<4>(h, st, blocks, al); break;
      case 5: rowx_sweep<5>(h, st, blocks, al); break;
      case 6: rowx_sweep<6>(h, st, blocks, al); break;
      case 7: rowx_sweep<7>(h, st, blocks, al); break;
      default: rowx_sweep<8>(h, st, blocks, al); break;
    }
    HIPCHK(h, hipGetLastError());
  }
  toc(h, MUXGL_T_DEMUX_SWEEP);
  tic(h, MUXGL_T_DEMUX_REDUCE);
  if (h->want_full_ll) {
    hipLaunchKernelGGL(demux_rowx_reduce_kernel, dim3((unsigned)h->C), dim3(192), 0, h->stream, st->d_cell_chunk_ptr,
                       st->d_cell_chunks, st->d_part, st->d_tmap, h->V, p->n_alpha, h->d_ll);
  } else {  // reduce + call fused, records written to the pinned host buffer
    const muxgl_call::call_alpha ca = muxgl_call::make_call_alpha(p, h->V);
    hipLaunchKernelGGL(demux_rowx_finish_kernel, dim3((unsigned)h->C), dim3(256), 0, h->stream, h->d_cell_ptr,
                       st->d_cell_chunk_ptr, st->d_cell_chunks, st->d_part, st->d_tmap, h->V, ca, p->doublet_prior,
                       h->h_dcells);
    h->records_on_host = true;
  }
  HIPCHK(h, hipGetLastError());
  toc(h, MUXGL_T_DEMUX_REDUCE);
  return 0;
}
