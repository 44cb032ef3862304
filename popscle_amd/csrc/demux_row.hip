// demux_row.hip -- the demuxlet pair sweep for V <= 16 samples: the "row" kernel.
//
// Reference being replaced: cmd_cram_demuxlet.cpp:655-747 (per-read pG update, floor/normalise, pair sweep).
//
// Why a second kernel: at V = 16 with the reference's default grid {0, 0.5} an entry carries only 16 singlet + 120
// unordered doublet hypotheses (~550 FP64 lane-instructions), so a workgroup-per-cell sweep that stages GP rows through
// LDS is dominated by staging, barriers and LDS reads.  Here nothing but the per-entry pG (72..432 B) goes through LDS:
//
//   * a 64-lane wave is 4 "slots" x 16 lanes; a slot owns one chunk (<= CH entries of one cell), lane j of the slot
//     owns sample j.  Lane j loads ITS OWN GP triple gp[snp][j][0..2] straight from global/L2 (the 16 lanes of a slot
//     read the 384-byte row contiguously) -- no row staging.
//   * with u[n][m] = sum_l g_j[l] * pG[n][l][m] (9 FMA per alpha, per lane), every pair term is
//     sumP[j,k,n] = sum_m g_k[m] * u[n][m]  (3 FMA) and one multiply into the product accumulator; the partner's g_k
//     arrives by rotating the slot's 16 lanes with DPP row_ror:1 (6 x v_mov_b32_dpp per shift, no LDS): after t shifts
//     lane j holds the triple of sample k = kmap[t][j].  alpha = 0.5 is symmetric in (j,k): 8 shifts instead of 15.
//   * per-hypothesis products are kept as mantissa * 2^exponent (prodacc) and turned into ONE log per chunk;
//     demux_row_reduce_kernel then adds the chunk partials of each cell in chunk order (deterministic) into the
//     [C][V][V][A] tensor the call kernel reads.
//   * phase 1 of every 16-entry batch is lane <-> entry: all 64 lanes turn 64 entries' reads into pG (a4,a5).
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "demux_entry.hpp"

namespace {


struct row_alpha {
  double a[MUXGL_MAX_ALPHA];     // internal order: [0] = the singlet slot's alpha, then non-symmetric, then 0.5
  int32_t orig[MUXGL_MAX_ALPHA]; // internal index -> index in the caller's grid
};

__device__ __forceinline__ double dpp_ror1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x121, 0xF, 0xF, false);  // row_ror:1
  hi = __builtin_amdgcn_mov_dpp(hi, 0x121, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// which sample's triple lane j holds after t rotations (measured, so the kernels never assume a rotation direction)
__global__ void row_kmap_kernel(int32_t* kmap /*[16][16] = [t][j]*/) {
  const int lane = threadIdx.x;
  int v = lane & 15;
  kmap[lane & 15] = v;
  for (int t = 1; t < 16; ++t) {
    v = __builtin_amdgcn_mov_dpp(v, 0x121, 0xF, 0xF, false);
    if (lane < 16) kmap[t * 16 + lane] = v;
  }
}

// NNS = number of non-symmetric doublet alphas, NSY = 1 if the grid holds alpha == 0.5.
// accumulators per lane: [0] singlet (j,0,n=0); then shift-major: t = 1..15 -> NNS slots (+1 symmetric slot if t <= 8)
template <int NNS, int NSY>
struct row_layout {
  static constexpr int NA = 1 + NNS + NSY;
  static constexpr int NSHIFT = (NNS > 0) ? 15 : (NSY ? 8 : 0);
  static constexpr int NACC = 1 + 15 * NNS + 8 * NSY;
  static constexpr int PGS = ((NA * 9 + 1) / 2) * 2;  // doubles per entry in LDS, even
  static constexpr int SLOT_STRIDE = 16 * PGS + 4;    // +4 doubles: the 4 slots' broadcast reads fall on distinct banks
  __host__ __device__ static constexpr int acc_index(int t, int n /*internal, >= 1*/) {
    // slots of shifts 1..t-1, then this shift's
    return 1 + (t - 1) * NNS + ((t - 1 < 8) ? (t - 1) : 8) * NSY + (n - 1);
  }
};

template <int NNS, int NSY>
__global__ void __launch_bounds__(64, (NNS == 0 ? 2 : 1))
    demux_row_kernel(const row_chunk* __restrict__ chunks, int n_chunks, const int32_t* __restrict__ entry_snp,
                     const int64_t* __restrict__ entry_rptr, const uint8_t* __restrict__ reads,
                     const double* __restrict__ gp, const uint8_t* __restrict__ has_gp,
                     const double* __restrict__ lut_g, int V, row_alpha al, double* __restrict__ part) {
  using L = row_layout<NNS, NSY>;
  constexpr int NA = L::NA, NACC = L::NACC, PGS = L::PGS;
  __shared__ double lut[384];
  __shared__ __align__(16) double pgs[4 * L::SLOT_STRIDE];
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
  const int V3 = V * 3;
  const bool live = j < V;
  const bool a0_zero = (al.a[0] == 0.0);  // wave-uniform

  // Metadata of the batch to come is fetched one batch ahead, in two dependent stages, so that phase 1 never waits
  // on HBM: (ps, pr0, pr1) at the top of the previous batch, then has_gp[ps] and the first four read bytes in the
  // middle of the previous batch's phase 2.
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
      double pG[NA * 9];
      int32_t s = ps;
      const int64_t r0 = pr0, r1 = pr1;
      const uint32_t first4 = pbytes;
      const int32_t hg = phg;
      if (b + 1 < nb) fetch_meta(b + 1);  // stage 1 of the next batch: independent loads, issued now
      if (s >= 0) {
        if (hg) {
          row_entry_pg<NA>(reads, r0, r1, first4, al.a, lut, pG);
        } else {
          s = -1;  // :733  marker without genotypes: contributes nothing
        }
      }
      if (s < 0) {
#pragma unroll
        for (int i = 0; i < NA * 9; ++i) pG[i] = 1.0;  // with g = (1,0,0) every factor of a dead entry is exactly 1
      }
      double* dst = pgs + slot * L::SLOT_STRIDE + j * PGS;
#pragma unroll
      for (int i = 0; i < NA * 9; ++i) dst[i] = pG[i];
      snps[lane] = s;
    }
    __syncthreads();

    // ---- phase 2: lane <-> sample, 16 entries of the slot's chunk ----
    int32_t s_next = snps[slot * 16];
    double ng0 = 1.0, ng1 = 0.0, ng2 = 0.0, nh0 = 1.0, nh1 = 0.0, nh2 = 0.0;
    if (s_next >= 0) {
      const double* row = gp + (size_t)s_next * V3;
      nh0 = row[0], nh1 = row[1], nh2 = row[2];
      if (live) ng0 = row[j * 3], ng1 = row[j * 3 + 1], ng2 = row[j * 3 + 2];
    }
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
      const double g0 = ng0, g1 = ng1, g2 = ng2, h0 = nh0, h1 = nh1, h2 = nh2;
      // prefetch the next entry's triples
      ng0 = 1.0, ng1 = 0.0, ng2 = 0.0, nh0 = 1.0, nh1 = 0.0, nh2 = 0.0;
      if (i == 8 && b + 1 < nb) fetch_dependent();  // stage 2 of the next batch
      if (i + 1 < 16) {
        s_next = snps[slot * 16 + i + 1];
        if (s_next >= 0) {
          const double* row = gp + (size_t)s_next * V3;
          nh0 = row[0], nh1 = row[1], nh2 = row[2];
          if (live) ng0 = row[j * 3], ng1 = row[j * 3 + 1], ng2 = row[j * 3 + 2];
        }
      }
      const double* qn = pgs + slot * L::SLOT_STRIDE + i * PGS;
      // singlet slot: llksAB[j][0][0] (:806,828) = sum_{l,m} g_j[l] g_0[m] pG[0][l][m]
      if (a0_zero) {
        // alpha[0] == 0: p does not depend on m (:673), the three columns of pG[0] are the same numbers, and the sum
        // factorises into (sum_l g_j[l] pG[0][l][0]) * (g_0[0] + g_0[1] + g_0[2])
        const double u0 = fma(g2, qn[6], fma(g1, qn[3], g0 * qn[0]));
        acc[0] *= u0 * (h0 + h1 + h2);
      } else {
        const double u0 = fma(g2, qn[6], fma(g1, qn[3], g0 * qn[0]));
        const double u1 = fma(g2, qn[7], fma(g1, qn[4], g0 * qn[1]));
        const double u2 = fma(g2, qn[8], fma(g1, qn[5], g0 * qn[2]));
        acc[0] *= fma(h2, u2, fma(h1, u1, h0 * u0));
      }
      if (L::NSHIFT > 0) {
        double u[(NA > 1 ? NA - 1 : 1)][3];
#pragma unroll
        for (int n = 1; n < NA; ++n) {
          const double* p = qn + n * 9;
          u[n - 1][0] = fma(g2, p[6], fma(g1, p[3], g0 * p[0]));
          u[n - 1][1] = fma(g2, p[7], fma(g1, p[4], g0 * p[1]));
          u[n - 1][2] = fma(g2, p[8], fma(g1, p[5], g0 * p[2]));
        }
        double r0 = g0, r1 = g1, r2 = g2;
#pragma unroll
        for (int t = 1; t <= L::NSHIFT; ++t) {
          r0 = dpp_ror1(r0);
          r1 = dpp_ror1(r1);
          r2 = dpp_ror1(r2);
#pragma unroll
          for (int n = 1; n < NA; ++n) {
            const bool sym = NSY && (n == NA - 1);
            if (!sym || t <= 8) {
              const double sp = fma(r2, u[n - 1][2], fma(r1, u[n - 1][1], r0 * u[n - 1][0]));  // :738-744
              acc[L::acc_index(t, n)] *= sp;                                                  // :746 as a product
            }
          }
        }
      }
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

// adds the chunk partials of one cell, in chunk order, into ll[c][j][k][n] (+ mirror for alpha 0.5)
template <int NNS, int NSY>
__global__ void __launch_bounds__(256)
    demux_row_reduce_kernel(const int64_t* __restrict__ cell_chunk_ptr, const int32_t* __restrict__ cell_chunks,
                            const double* __restrict__ part, const int32_t* __restrict__ kmap, int V, int A,
                            row_alpha al, double* __restrict__ ll) {
  using L = row_layout<NNS, NSY>;
  constexpr int NA = L::NA, NACC = L::NACC;
  const int64_t c = blockIdx.x;
  const int64_t c0 = cell_chunk_ptr[c], c1 = cell_chunk_ptr[c + 1];
  if (c0 == c1) return;
  double* out = ll + (size_t)c * V * V * A;
  for (int idx = threadIdx.x; idx < NACC * 16; idx += blockDim.x) {
    const int a = idx >> 4, j = idx & 15;
    if (j >= V) continue;
    int t = 0, n = 0;
    if (a > 0) {
      // invert acc_index
      int rem = a - 1;
      for (t = 1; t <= 15; ++t) {
        const int cnt = NNS + ((t <= 8) ? NSY : 0);
        if (rem < cnt) break;
        rem -= cnt;
      }
      n = 1 + rem;
    }
    const int k = (a == 0) ? 0 : kmap[t * 16 + j];
    if (k >= V) continue;
    double s = 0.0;
    for (int64_t ci = c0; ci < c1; ++ci) s += part[(size_t)cell_chunks[ci] * NACC * 16 + idx];
    const int no = al.orig[n];
    const bool sym = NSY && (n == NA - 1) && a > 0;
    if (sym && t == 8 && j < k) continue;  // shift 8 visits every unordered pair twice: one writer
    out[((size_t)j * V + k) * A + no] = s;
    if (sym) out[((size_t)k * V + j) * A + no] = s;
  }
}

struct row_plan {
  std::vector<row_chunk> chunks;       // launch order: non-increasing length
  std::vector<int64_t> cell_chunk_ptr; // [C+1]
  std::vector<int32_t> cell_chunks;    // chunk ids (launch order positions) of each cell, in entry order
};

}  // namespace

void demux_row_release(muxgl_row_state** pst) {
  muxgl_row_state* st = *pst;
  if (!st) return;
  dev_free(&st->d_chunks);
  dev_free(&st->d_cell_chunk_ptr);
  dev_free(&st->d_cell_chunks);
  dev_free(&st->d_kmap);
  dev_free(&st->d_tmap);
  dev_free(&st->d_part);
  dev_free(&st->d_part_e);
  dev_free(&st->d_chunk_pos);
  dev_free(&st->d_qent_lin);
  dev_free(&st->d_chunk_nlin);
  dev_free(&st->d_orec);
  dev_free(&st->d_unit_ptr);
  dev_free(&st->d_quad_order);
  dev_free(&st->d_fo_lsteps);
  dev_free(&st->d_fo_gsteps);
  dev_free(&st->d_fo_lptr);
  dev_free(&st->d_fo_gptr);
  dev_free(&st->d_fo_loff);
  dev_free(&st->d_fo_goff);
  dev_free(&st->d_fo_lc);
  dev_free(&st->d_fo_ggl);
  dev_free(&st->d_fq_nlin);
  dev_free(&st->d_fq_order);
  delete st;
  *pst = nullptr;
}

void demux_row_free(muxgl_handle* h) {
  demux_row_release(&h->row);
  demux_row_release(&h->frow);
  demux_row_release(&h->fqrow);
  demux_row_release(&h->qrow);
}

// builds the chunk tables of every cell from the host copy of the CSR arrays (called by muxgl_set_pileup)
int demux_row_plan(muxgl_handle* h) {
  if (demux_row_build(h, &h->row, 0, h->C, MUXGL_ROW_CH)) return 1;
  int qch = MUXGL_OCT_CH;
  if (const char* ev = getenv("MUXGL_OCT_CH")) qch = atoi(ev) >= 16 ? atoi(ev) / 4 * 4 : qch;  // (tuning)
  return demux_row_build(h, &h->qrow, 0, h->C, qch);
}

// chunk tables of the cells [cb, ce): built on the device (plan_kernels.hip) from the device copy of the CSR arrays
int demux_row_build(muxgl_handle* h, muxgl_row_state** pst, int64_t cb, int64_t ce, int ch) {
  if (!*pst) *pst = new muxgl_row_state();
  muxgl_row_state* st = *pst;
  dev_free(&st->d_chunk_pos);  // (derived from the chunk tables: rebuilt on first use)
  if (plan_build_chunks(h, st, cb, ce, ch)) return 1;
  if (!st->d_kmap) {
    if (dev_alloc(h, &st->d_kmap, 256)) return 1;
    hipLaunchKernelGGL(row_kmap_kernel, dim3(1), dim3(64), 0, h->stream, st->d_kmap);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return 0;
}

template <int NNS, int NSY>
static int row_launch_t(muxgl_handle* h, const row_alpha& al, int A) {
  using L = row_layout<NNS, NSY>;
  muxgl_row_state* st = h->row;
  const size_t need = (size_t)st->n_chunks * L::NACC * 16;
  if (need > st->part_cap) {
    if (dev_alloc(h, &st->d_part, need)) return 1;
    st->part_cap = need;
  }
  const unsigned blocks = (unsigned)((((st->n_chunks + 3) / 4) + 7) / 8 * 8);  // multiple of 8 for xcd_swizzle
  if (blocks) {
    hipLaunchKernelGGL((demux_row_kernel<NNS, NSY>), dim3(blocks), dim3(64), 0, h->stream, st->d_chunks,
                       (int)st->n_chunks, h->d_entry_snp, h->d_entry_rptr, h->d_reads, h->d_gp, h->d_has_gp, h->d_lut,
                       h->V, al, st->d_part);
    HIPCHK(h, hipGetLastError());
  }
  toc(h, MUXGL_T_DEMUX_SWEEP);
  tic(h, MUXGL_T_DEMUX_REDUCE);
  if (h->C) {
    hipLaunchKernelGGL((demux_row_reduce_kernel<NNS, NSY>), dim3((unsigned)h->C), dim3(256), 0, h->stream,
                       st->d_cell_chunk_ptr, st->d_cell_chunks, st->d_part, st->d_kmap, h->V, A, al, h->d_ll);
    HIPCHK(h, hipGetLastError());
  }
  toc(h, MUXGL_T_DEMUX_REDUCE);
  return 0;
}

// returns -1 when the row path does not apply (caller falls back to the tile sweep), 0 ok, 1 error
int demux_row_launch(muxgl_handle* h, const muxgl_demux_params* p) {
  if (h->V > 16 || !h->row || h->C == 0 || (h->flags & (MUXGL_FLAG_FORCE_TILE_SWEEP | MUXGL_FLAG_FORCE_WAVE_KERNEL)))
    return -1;
  const int A = p->n_alpha;
  row_alpha al;
  int nns = 0, nsy = 0, pos = 0;
  al.a[pos] = p->alpha[0];
  al.orig[pos++] = 0;
  for (int n = 1; n < A; ++n)
    if (p->alpha[n] != 0.5) {
      al.a[pos] = p->alpha[n];
      al.orig[pos++] = n;
      ++nns;
    }
  for (int n = 1; n < A; ++n)
    if (p->alpha[n] == 0.5) {
      if (nsy) return -1;  // 0.5 listed twice: generic path
      al.a[pos] = p->alpha[n];
      al.orig[pos++] = n;
      ++nsy;
    }
  for (; pos < MUXGL_MAX_ALPHA; ++pos) {
    al.a[pos] = 0.0;
    al.orig[pos] = 0;
  }
  if (nns > 5) return -1;
  tic(h, MUXGL_T_DEMUX_SWEEP);
#define ROW_CASE(N, S) \
  if (nns == N && nsy == S) return row_launch_t<N, S>(h, al, A);
  ROW_CASE(0, 0) ROW_CASE(0, 1) ROW_CASE(1, 0) ROW_CASE(1, 1) ROW_CASE(2, 0) ROW_CASE(2, 1) ROW_CASE(3, 0)
  ROW_CASE(3, 1) ROW_CASE(4, 0) ROW_CASE(4, 1) ROW_CASE(5, 0) ROW_CASE(5, 1)
#undef ROW_CASE
  return -1;
}
