// demux_oct.hip -- the demuxlet sweep for the reference's DEFAULT configuration: V <= 16 samples and the alpha grid
// {0, 0.5} that cmdCramDemuxlet installs when no --alpha is given (cmd_cram_demuxlet.cpp:85-89).
//
// Reference being replaced: cmd_cram_demuxlet.cpp:655-747.  Same mathematics as demux_row.hip; what changes is the
// tiling: EIGHT lanes per entry, lane p of an entry owns the two samples p and p + 8.
//
//   * Why eight.  Measured on MI355X (round 3, tools/gpu_quadx.sh): the four-lanes-per-entry tiling of rounds 1-2 (lane =
//     4 samples, 36 accumulators, 250 VGPRs, two waves per SIMD) was bound by its row gathers, not by VALU issue -- its
//     loop over the linear entries took 0.257 ms, 0.092 ms with the row loads removed, 0.241 ms with every row an L1
//     hit, 0.123 ms with all sixteen slots of a wave reading the same row and 0.162 ms when every 16-lane group read
//     whole rows: what a gather costs in the vector cache is the number of distinct 128-byte lines an instruction
//     touches (64-byte pieces of sixteen different rows per instruction there), hits or misses alike.  With eight lanes
//     per entry an instruction's eight lanes read 128 contiguous bytes -- one whole line per entry -- so a row of
//     moments (256 B) costs two line accesses instead of four half lines and a row of triples (384 B) three instead of
//     six; and a lane holds 18 accumulators instead of 36, which lets three to four waves share a SIMD instead of two.
//   * Two entries share a 16-lane DPP row, interleaved (lane = 16 g + 2 p + h: entry h of row g, position p), so that
//     row_ror:2t rotates the eight positions of both entries by t.  Rotations t = 1, 2, 3 bring the partner's two
//     samples: four pairs each; t = 4 faces lane p with lane p + 4: of their four pairs each lane takes (a, b'),
//     and both take (a, a') and (b, b') (published once): 1 + 12 + 3 pairs + 2 singlets = 18 accumulators.
//   * the grid {0, 0.5} has structure the general code cannot assume: for alpha = 0 the mixing proportion is
//     p = l/2, independent of m (3 distinct likelihoods per entry instead of 9), and for alpha = 0.5 it is
//     p = (l+m)/4 (5 distinct instead of 9).  Phase 1 therefore carries 8 products per read instead of 18, and the
//     singlet slot factorises into (sum_l g_j[l] q0[l]) * (g_0[0]+g_0[1]+g_0[2]).
//   * products leave the kernel as (mantissa, exponent) pairs; the finish kernel multiplies the chunk partials
//     of a cell in chunk order and takes ONE log per hypothesis and cell.
//   * entries with at most one usable read (three quarters of a typical pileup) are linear in the genotypes: a chunk's
//     records are partitioned (oct_partition_kernel), the linear ones are swept first by a loop of their own from rows of
//     moments (s, rho) -- one FMA and the product update per hypothesis, the sums s folded in at the end.  Their
//     likelihoods depend on ONE read byte (quad_lrec::code, common.hpp; 256 values: a table in LDS), so that loop has
//     no phase 1: the lanes read the entry's 8-byte record themselves.
//   * the launch order sorts chunks of similar trip counts into the same waves (quad_order_key_kernel).  One wave per
//     unit of eight chunks, handed out by the hardware as slots free up.  (Measured and dropped: a persistent launch of
//     as many waves as the chip holds, each with a work list balanced by instruction counts (LPT), 0.32 ms against
//     0.28 ms -- co-resident waves do not run at equal speed, the dispatcher's greedy hand-out balances better than a
//     static plan; and units in descending cost order instead of SNP windows, +-0.)
#include <algorithm>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "demux_call_body.hpp"

namespace {

constexpr int O_NLUT = 129;        // {A, B, 2B} by (allele << 6 | base quality <= 63), and one neutral entry
constexpr int O_LPAD = 3;          // steps of neutral records behind a unit's longest linear list (the loop reads ahead)

// Geometry of the tiling, by the number P of lanes ("positions") an entry occupies: P = 8 for V <= 16 samples (lane p
// owns samples p and p + 8; two entries share a 16-lane DPP row, interleaved, so row_ror:2t rotates both rings of eight
// by t), P = 16 for 16 < V <= 32 (round 4: lane p owns samples p and p + 16, an entry IS a DPP row, row_ror:t).
// Rotations t = 1 .. P/2 - 1 bring the partner's two samples (four pairs each); at t = P/2 lane p faces lane p + P/2
// and takes (a, b'), both take (a, a') and (b, b') (published once).
template <int P>
struct og {
  static_assert(P == 8 || P == 16, "eight or sixteen lanes per entry");
  static constexpr int SLOTS = 64 / P;            // entry streams (chunks) per wave
  static constexpr int BATCH = P;                 // entries per slot and phase 1 (64 lanes <-> SLOTS x P entries)
  static constexpr int NROT = P / 2 - 1;          // full rotations
  static constexpr int N_ACC = 3 + 4 * NROT + 3;  // per lane: 2 singlets, the in-lane pair, 4 per rotation, 3 facing (18 / 34)
  static constexpr int N_HYP = N_ACC * P;         // accumulators of a chunk: 144 slots for 136 hypotheses / 544 for 528
  static constexpr int SLOT_STRIDE = 9 * P + 2;   // doubles: P entries x 8 likelihoods + P singlet factors + 2 pad (bank spread)
  static constexpr int ACC_AB = 2;                // (a, b)
  static constexpr int ACC_F_AB = 3 + 4 * NROT, ACC_F_AA = ACC_F_AB + 1, ACC_F_BB = ACC_F_AB + 2;  // (a, b'), (a, a'), (b, b')
  __host__ __device__ static constexpr int acc_single(int c) { return c; }  // c = 0: a, 1: b
  __host__ __device__ static constexpr int acc_rot(int t, int c, int d) { return 3 + (t - 1) * 4 + c * 2 + d; }  // t = 1 .. NROT
  __host__ __device__ static constexpr int ror(int t) { return 0x120 + (P == 8 ? 2 * t : t); }  // DPP control: rotation by t positions
  __device__ static __forceinline__ int pos(int lane) { return P == 8 ? (lane >> 1) & 7 : lane & 15; }
  __device__ static __forceinline__ int slot(int lane) { return P == 8 ? ((lane >> 4) << 1) | (lane & 1) : lane >> 4; }
  __device__ static __forceinline__ int lane0(int lane) { return P == 8 ? lane & ~14 : lane & ~15; }  // position 0 of the slot
  static constexpr uint32_t MROW = 32u * P;       // bytes of a marker's row of moments (s, rho) x 2P samples (unit sums: half)
  static constexpr int TROW = 6 * P;              // doubles of a marker's row of triples
};

template <int CTRL>
__device__ __forceinline__ double dpp_rot(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// position whose samples a lane sees after the rotation by t, t = 1 .. P/2, measured rather than assumed: pmap[P/2][P]
template <int P>
__global__ void oct_pmap_kernel(int32_t* pmap) {
  const int lane = threadIdx.x;
  const int p = og<P>::pos(lane);
  wave_for<1, P / 2 + 1>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const int pt = __builtin_amdgcn_mov_dpp(p, og<P>::ror(t), 0xF, 0xF, false);
    if (og<P>::slot(lane) == 0) pmap[(t - 1) * P + p] = pt;
  });
}

// A chunk's entry records with its linear entries (at most one usable read, plan_kernels.hip: lin_kernel) first, both
// kinds in entry order, and the number of linear ones; the linear ones also as quad_lrec {snp, the read byte that
// counts}.  One thread per chunk (<= 128 records), once per pileup and GP tensor (has_gp enters the codes).
__global__ void __launch_bounds__(64)
    oct_partition_kernel(int n_chunks, const row_chunk* __restrict__ chunks, const quad_entry* __restrict__ qent,
                         const uint32_t* __restrict__ lin, const uint8_t* __restrict__ reads,
                         const double* __restrict__ gp0s, quad_entry* __restrict__ out, quad_lrec* __restrict__ lrec,
                         int32_t* __restrict__ nlin) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_chunks) return;
  const int64_t e0 = chunks[q].e0;
  const int len = chunks[q].len;
  int w = 0;
  for (int i = 0; i < len; ++i) {
    const int64_t e = e0 + i;
    if ((lin[e >> 5] >> (e & 31)) & 1u) {
      const quad_entry p = qent[e];
      uint32_t code = MUXGL_READ_OTHER;
      for (uint32_t k = 0; k < p.nreads; ++k) {  // the usable read (:664)
        const uint32_t bb = k < 4 ? (p.first4 >> (8 * k)) & 0xffu : (uint32_t)reads[(int64_t)p.r0 + k];
        if (bb != MUXGL_READ_OTHER) code = bb;
      }
      if (gp0s[p.snp] < 0.0) code = MUXGL_READ_OTHER;  // no genotypes: the entry is skipped (:733)
      lrec[e0 + w] = quad_lrec{p.snp, code};
      out[e0 + w++] = p;
    }
  }
  nlin[q] = w;
  for (int i = 0; i < len; ++i) {
    const int64_t e = e0 + i;
    if (!((lin[e >> 5] >> (e & 31)) & 1u)) out[e0 + w++] = qent[e];
  }
}

// Launch order: a wave's trip counts are the maxima over its chunks, so within every bucket of QUAD_BUCKET
// consecutive chunks of the plan's order (ascending first SNP -- the rows gathered by co-resident workgroups stay a
// sliding window at that grain) the chunks are sorted by their number of non-linear batches.
constexpr int QUAD_BUCKET = 1024;
__global__ void __launch_bounds__(256)
    quad_order_key_kernel(int n_chunks, int bucket, const row_chunk* __restrict__ chunks,
                          const int32_t* __restrict__ chunk_nlin, uint64_t* __restrict__ key, int32_t* __restrict__ iota) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_chunks) return;
  const int nl = chunk_nlin[w], len = chunks[w].len;
  const uint64_t bg = (uint64_t)min(len - nl, 1023), bl = (uint64_t)min(nl, 1023);
  key[w] = ((uint64_t)(w / bucket) << 40) | ((1023u - bg) << 30) | ((1023u - bl) << 20) | (uint64_t)(w % bucket);
  iota[w] = w;
}

// Steps of a unit's linear loop: the longest linear list among its eight chunks, rounded to the loop's unrolling, plus
// the read-ahead (0 for a unit without linear entries).
__global__ void __launch_bounds__(256)
    oct_unit_steps_kernel(int n_units, int n_chunks, int slots, const int32_t* __restrict__ order,
                          const int32_t* __restrict__ nlin, int32_t* __restrict__ steps) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_units) return;
  int m = 0;
  for (int k = 0; k < slots; ++k) {
    const int w = u * slots + k;
    if (w < n_chunks) m = max(m, nlin[order[w]]);
  }
  steps[u] = m > 0 ? (m + 2) / 3 * 3 + O_LPAD : 0;
}

// The linear entries' records in the order the sweep reads them: orec[unit_ptr[u] + i * slots + slot], one 64-byte line
// (eight slots) or half of one (four) per step of a wave.  {byte offset of the marker's row of moments, byte offset of the entry's {A, B, 2B}}.
__global__ void __launch_bounds__(256)
    oct_repack_kernel(int n_units, int n_chunks, const row_chunk* __restrict__ chunks, const int32_t* __restrict__ order,
                      const int32_t* __restrict__ nlin, const quad_lrec* __restrict__ lrec,
                      const int32_t* __restrict__ steps, const int64_t* __restrict__ unit_ptr, uint32_t S_dummy,
                      int slots, uint32_t mrow /* bytes of a marker's row of moments */, uint2* __restrict__ orec) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int u = t / slots, slot = t % slots;
  if (u >= n_units) return;
  const int w = u * slots + slot;
  int64_t e0 = 0;
  int nl = 0;
  if (w < n_chunks) {
    const int q = order[w];
    e0 = chunks[q].e0;
    nl = nlin[q];
  }
  uint2* dst = orec + unit_ptr[u] + slot;
  const int n = steps[u];
  for (int i = 0; i < n; ++i) {
    uint2 r{S_dummy * mrow, (uint32_t)(O_NLUT - 1) * 32u};
    if (i < nl) {
      const quad_lrec x = lrec[e0 + i];
      const uint32_t idx = x.code == MUXGL_READ_OTHER ? (uint32_t)(O_NLUT - 1) : ((x.code >> 7) << 6) | (x.code & 0x3fu);
      r = uint2{(uint32_t)x.snp * mrow, idx * 32u};
    }
    dst[(size_t)i * slots] = r;
  }
}

// The sweep kernel.  Lane = 16 g + 2 p + h: slot 2 g + h of the wave (one chunk), position p (samples p, p + 8).
// UNIT_S: every triple of the GP tensor sums to 1 within 4 ulp (muxgl_demux_set_gp checks; hard calls through the
// reference's error mixing do, sc_drop_seq.cpp:287-315).  The rows of moments then carry no sums -- rho alone, 128
// bytes per marker: ONE line per linear entry -- and the products of sums (accW, accH) are 1.  The sweep is bound by
// the bytes its row gathers move (see the file header), so this halves the linear entries' cost; each factor changes
// by <= 8 ulp.  (Rows of triples stay whole: g0 = 1 - g1 - g2 has an absolute error of an ulp of 1, which is a
// relative error of 1e-4 where a genotype no sample carries has probability 0.1 * 1e-10 / V after the mixing -- tried,
// and caught by the test with qualities up to 93.)
template <int P, bool UNIT_S>
#ifndef OCT_WAVES
#define OCT_WAVES 3  // waves per SIMD the P = 8 kernel is allocated for (timing experiments: 4, 5)
#endif
__global__ void __launch_bounds__(64, P == 8 ? OCT_WAVES : 2)
    demux_oct_kernel(const row_chunk* __restrict__ chunks, int n_chunks, const quad_entry* __restrict__ qent,
                     const uint2* __restrict__ orec, const int64_t* __restrict__ unit_ptr,
                     const int32_t* __restrict__ chunk_nlin,
                     const int32_t* __restrict__ order, const uint8_t* __restrict__ reads,
                     const double* __restrict__ gpo, const double* __restrict__ gmo,
                     const double* __restrict__ gp0s, int32_t S_dummy, const double* __restrict__ lut_g,
                     const int32_t* __restrict__ chunk_pos, double* __restrict__ part_m, int32_t* __restrict__ part_e) {
  __shared__ double lut[384];
  __shared__ __align__(16) double ablut[O_NLUT * 4];
  using G = og<P>;
  constexpr int ON_ACC = G::N_ACC, O_SLOTS = G::SLOTS, O_BATCH = G::BATCH, O_SLOT_STRIDE = G::SLOT_STRIDE;
  constexpr int O_ACC_AB = G::ACC_AB, O_ACC_F_AB = G::ACC_F_AB, O_ACC_F_AA = G::ACC_F_AA, O_ACC_F_BB = G::ACC_F_BB;
  __shared__ __align__(16) double pgs[O_SLOTS * O_SLOT_STRIDE];
  __shared__ int32_t snps[64], snps_nx[64];

  const int lane = threadIdx.x;
  const int p = G::pos(lane);    // position: samples p and p + P
  const int slot = G::slot(lane);  // 64 / P entry streams per wave
  for (int i = lane; i < 384; i += 64) lut[i] = lut_g[i];
  // {A, B, 2B} of a linear entry by the read byte that counts (allele << 6 | base quality; O_NLUT - 1: none): the one
  // factor pR + (pA - pR) p of :673,685 through the tail (q / q_max + 1e-10) / (1 + 1e-10) of :703-725;
  // q0[l] = A + 2B l, q1[l+m] = A + B (l+m)
  for (int i = lane; i < O_NLUT; i += 64) {
    const uint32_t bq = (uint32_t)i & 0x3f;
    const bool ref = (i >> 6) == 0;
    const double e3 = lut_g[256 + bq], mt = lut_g[128 + bq];
    const double pR = ref ? mt : e3, pA = ref ? e3 : mt;  // :666-667
    const double mx = fmax(pR, pA);
    double x = __builtin_amdgcn_rcp(mx);
    x = fma(x, fma(-mx, x, 1.0), x);
    x = fma(x, fma(-mx, x, 1.0), x);
    const double cc = 1.0 / (1.0 + 1e-10);
    const double sc = cc * x, tt = 1e-10 * cc;
    const bool none = i == O_NLUT - 1;  // no usable read / no genotypes: factors of exactly 1
    const double B = none ? 0.0 : (pA - pR) * (0.25 * sc);
    ablut[4 * i] = none ? 1.0 : fma(pR, sc, tt);
    ablut[4 * i + 1] = B;
    ablut[4 * i + 2] = B + B;
    ablut[4 * i + 3] = 0.0;
  }
  __syncthreads();  // the tables are complete

  const int unit = xcd_swizzle(blockIdx.x, gridDim.x >> 3);  // eight chunks that are neighbours in the launch order
  const int wq = unit * O_SLOTS + slot;
  const uint32_t p16 = (uint32_t)p * 16u;
  const int q = wq < n_chunks ? (order ? order[wq] : wq) : n_chunks;
  int64_t e0 = 0;
  int len = 0;
  int32_t qpos = 0;  // where the chunk's partials go: its position in its cell's list
  if (q < n_chunks) {
    e0 = chunks[q].e0;
    len = chunks[q].len;
    qpos = chunk_pos[q];
  }
  // The chunk's first nl records are its linear entries (oct_partition_kernel; 0 when that form is off): they are swept
  // first, by a loop of their own (below), the others by the nine-term loop.  Trip counts of the wave = the longest
  // run of either kind among its chunks.
  const int nl = (chunk_nlin && q < n_chunks) ? chunk_nlin[q] : 0;
  const int nLmax = wave_max_i32(nl);
  const int ngmax = wave_max_i32(len - nl);
  const int nb = (ngmax + O_BATCH - 1) / O_BATCH;

  double acc[ON_ACC];
  int32_t exs[ON_ACC];
#pragma unroll
  for (int a = 0; a < ON_ACC; ++a) {
    acc[a] = 1.0;
    exs[a] = 0;
  }
  double accW[2] = {1.0, 1.0};  // linear entries: products of the sums s of the lane's two samples
  int32_t exW[2] = {0, 0};
  double accH = 1.0;            // nine-term entries: product of sample 0's sums, which every singlet carries (:806)
  int32_t exH = 0;

  auto renorm = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < ON_ACC; ++a) prodacc_renorm(acc[a], exs[a]);
    prodacc_renorm(accW[0], exW[0]);
    prodacc_renorm(accW[1], exW[1]);
    prodacc_renorm(accH, exH);
  };

  // ---- the linear entries (at most one usable read).  The single factor pR + (pA - pR) p, p = l/2 (alpha 0) or
  //      (l+m)/4 (alpha 0.5), stays linear through the tail (:703-725): q0[l] = A + 2B l, q1[l+m] = A + B (l+m).  With the
  //      moments s = g0 + g1 + g2 and rho = (g1 + 2 g2) / s of a triple (gmo),
  //          singlet  sum_l g_j[l] q0[l] * s_0      = s_j s_0 (A + 2B rho_j)          (s_0: sample 0's sum, :806),
  //          pair     sum_lm g_j[l] g_k[m] q1[l+m]  = s_j s_k (A + B rho_j + B rho_k):
  //      the sums s go into products of their own (accW, folded into the accumulators at the end) and a hypothesis costs
  //      an FMA and the product update instead of a three-term dot product and the update; a row is 2 x 16 bytes per
  //      lane, and one double per sample rotates instead of three.
  //      A pipeline over rings of three register sets (unrolled three times, so that ring positions are names): record
  //      three entries ahead, row two ahead, (A, B) one ahead; loads are unconditional and a slot behind the end of its
  //      list sweeps neutral entries (code 0xFF, the dummy row: every factor exactly 1).
  if (nLmax > 0) {
    struct rowl_t {
      double sa, ra, sb, rb;  // (s, rho) of samples p and p + 8
    };
    auto load_rowl = [&](rowl_t& R, uint32_t row_off) __attribute__((always_inline)) {
      if (UNIT_S) {  // rows of 2P x rho, (rho_p, rho_p+P) adjacent: [P][2]
        const double2 v = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(gmo) + (size_t)((row_off >> 1) + p16));
        R.sa = R.sb = 1.0;
        R.ra = v.x;
        R.rb = v.y;
        return;
      }
      const double2* pc = reinterpret_cast<const double2*>(reinterpret_cast<const char*>(gmo) + (size_t)(row_off + p16));
      const double2 va = pc[0], vb = pc[P];
      R.sa = va.x;
      R.ra = va.y;
      R.sb = vb.x;
      R.rb = vb.y;
    };
    // records of the unit, step-major: step i of the wave reads orec[i][slot] = {byte offset of the marker's row of
    // moments, byte offset of the entry's {A, B, 2B}}; a slot whose list has ended finds neutral records (the dummy
    // row, the table's neutral entry: every factor exactly 1), O_LPAD steps of them behind the longest list
    const uint2* lr = orec + unit_ptr[unit] + slot;
    struct ab_t {
      double A, B, B2;
    };
    auto ab_of = [&](const uint2& rc) __attribute__((always_inline)) {
      const double* t = reinterpret_cast<const double*>(reinterpret_cast<const char*>(ablut) + rc.y);
      const double2 v = *reinterpret_cast<const double2*>(t);
      return ab_t{v.x, v.y, t[2]};
    };
    auto sweepL = [&](const rowl_t& R, const ab_t& ab) __attribute__((always_inline)) {
      const double A = ab.A, B = ab.B, B2 = ab.B2;
      if (!UNIT_S) {
        accW[0] *= R.sa;
        accW[1] *= R.sb;
      }
      acc[G::acc_single(0)] *= fma(B2, R.ra, A);  // singlet slot llksAB[j][0][0] (:806,828), alpha = 0
      acc[G::acc_single(1)] *= fma(B2, R.rb, A);
      const double Xa = fma(B, R.ra, A), Xb = fma(B, R.rb, A);
      acc[O_ACC_AB] *= fma(B, R.rb, Xa);
      wave_for<1, G::NROT + 1>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const double Pa = dpp_rot<G::ror(t)>(R.ra), Pb = dpp_rot<G::ror(t)>(R.rb);
        acc[G::acc_rot(t, 0, 0)] *= fma(B, Pa, Xa);
        acc[G::acc_rot(t, 0, 1)] *= fma(B, Pb, Xa);
        acc[G::acc_rot(t, 1, 0)] *= fma(B, Pa, Xb);
        acc[G::acc_rot(t, 1, 1)] *= fma(B, Pb, Xb);
      });
      {  // the lane facing this one: (a, b') here and (a', b) over there; (a, a') and (b, b') on both sides
        const double Pa = dpp_rot<G::ror(P / 2)>(R.ra), Pb = dpp_rot<G::ror(P / 2)>(R.rb);
        acc[O_ACC_F_AB] *= fma(B, Pb, Xa);
        acc[O_ACC_F_AA] *= fma(B, Pa, Xa);
        acc[O_ACC_F_BB] *= fma(B, Pb, Xb);
      }
    };
    // entry i: its row in Rc and {A, B, 2B} in abc; rc0 held its record (free now), rc1 / rc2 hold those of i + 1 / i + 2
    auto step = [&](int i, const rowl_t& Rc, rowl_t& Rnn, const ab_t& abc, ab_t& abn, uint2& rc0, const uint2& rc1,
                    const uint2& rc2) __attribute__((always_inline)) {
      rc0 = lr[(i + 3) * O_SLOTS];
      // (issue order = completion order: the gathered row, which nobody reads for two sweeps, goes last, so that the
      //  sweep's wait for the next entry's {A, B, 2B} leaves it in flight -- fmx_oct.hip has the measurement)
      abn = ab_of(rc1);
      load_rowl(Rnn, rc2.x);
      __builtin_amdgcn_sched_barrier(0);  // the loads are issued in front of the sweep they hide behind
      sweepL(Rc, abc);
      __builtin_amdgcn_sched_barrier(0);
    };
    rowl_t L0, L1, L2;
    ab_t ab0, ab1, ab2;
    uint2 ra = lr[0], rb = lr[O_SLOTS], rc = lr[2 * O_SLOTS];
    load_rowl(L0, ra.x);
    load_rowl(L1, rb.x);
    ab0 = ab_of(ra);
    int since = 0;
    for (int i = 0; i < nLmax; i += 3) {  // (up to two neutral entries behind the longest list of the wave)
      step(i, L0, L2, ab0, ab1, ra, rb, rc);
      step(i + 1, L1, L0, ab1, ab2, rb, rc, ra);
      step(i + 2, L2, L1, ab2, ab0, rc, ra, rb);
      // a factor of a linear entry is > 2^-23 (one read of quality <= 60, lin_kernel): 30 of them between two
      // renormalisations cannot underflow
      if (++since == 10) {
        since = 0;
        renorm();
      }
    }
    renorm();
  }

  // ---- the other entries: phase 1 (lane <-> entry: 8 slots x 8 entries per batch), then phase 2 per entry ----
  // records {snp, read count, first four read bytes, read offset} of the lane's own entry: batch b in precA, b+1 in
  // precB, b+2 requested during phase 1 of batch b (loaded unconditionally from a clamped index and invalidated where
  // it is used: a predicated load followed by a merge with the defaults makes the compiler wait for the load on the spot)
  if (nb > 0) {
    const int last = len > 0 ? len - 1 : 0;
    auto fetch_meta = [&](int b) __attribute__((always_inline)) {
      const int idx = nl + b * O_BATCH + p;
      return qent[e0 + (idx < last ? idx : last)];
    };
    auto snp_of = [&](const quad_entry& r, int b) __attribute__((always_inline)) { return (nl + b * O_BATCH + p < len) ? r.snp : -1; };
    quad_entry precA = fetch_meta(0), precB = fetch_meta(1);
    // sum of sample 0's triple at the lane's own entry (negative: marker without genotypes), one batch ahead as well
    double hs_cur = gp0s[precA.snp];

    struct row_t {
      double a[3], b[3];  // triples of samples p and p + 8
    };
    auto load_row = [&](row_t& R, int32_t s) __attribute__((always_inline)) {
      // 16-byte pieces of this lane's doubles; a piece of the entry's eight lanes is 128 contiguous bytes.  Rows of
      // padding entries and of markers without genotypes are (1,0,0) -- the dummy row S_dummy resp. the host's fill --
      // which together with read likelihoods of 1 (phase 1) makes every factor of such an entry exactly 1 (:733).
      const double2* pc = reinterpret_cast<const double2*>(gpo + (size_t)s * G::TROW) + p;
      const double2 v0 = pc[0], v1 = pc[P], v2 = pc[2 * P];
      R.a[0] = v0.x;
      R.a[1] = v0.y;
      R.a[2] = v1.x;
      R.b[0] = v1.y;
      R.b[1] = v2.x;
      R.b[2] = v2.y;
    };

    // phase 1 of a batch.  cmd_cram_demuxlet.cpp:655-725 for alpha in {0, 0.5}:
    //      q0[l] = prod_reads (pR + d*l/2),  q1[t] = prod_reads (pR + d*t/4), t = l+m, d = pA - pR
    auto phase1 = [&](int b) __attribute__((always_inline)) {
      const int32_t sa = snp_of(precA, b);
      const int32_t s = (hs_cur >= 0.0) ? sa : -1;  // no genotypes: the entry is skipped (:733)
      const double hs_out = (s >= 0) ? hs_cur : 1.0;       // multiplies every singlet (:806)
      const int64_t r0 = precA.r0, r1 = (int64_t)precA.r0 + precA.nreads;
      const uint32_t first4 = precA.first4;
      precA = precB;
      precB = fetch_meta(b + 2);
      hs_cur = gp0s[precA.snp];
      // The first four reads (all of them for > 99 % of the entries) come out of the record and are handled without a
      // branch: a read that does not exist, an allele other than 0/1 (:664) or a skipped entry multiplies by exactly 1
      // (pR = pA = 1).  The eight LUT reads are issued together.
      double q0[3], q1[5];
      {
        double pRk[4], pAk[4];
        const uint32_t nr = (s >= 0) ? (uint32_t)(r1 - r0) : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t bb = (first4 >> (8 * k)) & 0xffu;
          const uint32_t bq = bb & 0x7f;
          const double e3 = lut[256 + bq], mt = lut[128 + bq];
          const bool use = (uint32_t)k < nr && bb != MUXGL_READ_OTHER;
          const bool ref = (bb >> 7) == 0;
          pRk[k] = use ? (ref ? mt : e3) : 1.0;  // :666-667
          pAk[k] = use ? (ref ? e3 : mt) : 1.0;
        }
        {
          const double d = pAk[0] - pRk[0];
          const double mid = fma(d, 0.5, pRk[0]);
          q0[0] = pRk[0];
          q0[1] = mid;
          q0[2] = pAk[0];
          q1[0] = pRk[0];
          q1[1] = fma(d, 0.25, pRk[0]);
          q1[2] = mid;
          q1[3] = fma(d, 0.75, pRk[0]);
          q1[4] = pAk[0];
        }
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          const double d = pAk[k] - pRk[k];
          const double mid = fma(d, 0.5, pRk[k]);
          q0[0] *= pRk[k];
          q0[1] *= mid;
          q0[2] *= pAk[k];
          q1[0] *= pRk[k];
          q1[1] *= fma(d, 0.25, pRk[k]);
          q1[2] *= mid;
          q1[3] *= fma(d, 0.75, pRk[k]);
          q1[4] *= pAk[k];
        }
        if (nr > 4) {  // deep entries: the rest from the read array
          int since = 4;
          for (int64_t rr = r0 + 4; rr < r1; ++rr) {
            const uint32_t bb = (uint32_t)reads[rr];
            if (bb == MUXGL_READ_OTHER) continue;  // :664
            const uint32_t al = bb >> 7, bq = bb & 0x7f;
            const double e3 = lut[256 + bq], mt = lut[128 + bq];
            const double pR = (al == 0) ? mt : e3, pA = (al == 0) ? e3 : mt;
            const double d = pA - pR;
            const double mid = fma(d, 0.5, pR);
            q0[0] *= pR;
            q0[1] *= mid;
            q0[2] *= pA;
            q1[0] *= pR;
            q1[1] *= fma(d, 0.25, pR);
            q1[2] *= mid;
            q1[3] *= fma(d, 0.75, pR);
            q1[4] *= pA;
            if (++since >= 32) {  // common rescaling, only against underflow (cancels in q/q_max)
              since = 0;
              double mx = fmax(fmax(q0[0], q0[1]), q0[2]);
#pragma unroll
              for (int t = 0; t < 5; ++t) mx = fmax(mx, q1[t]);
              const double inv = 1.0 / mx;
#pragma unroll
              for (int l = 0; l < 3; ++l) q0[l] *= inv;
#pragma unroll
              for (int t = 0; t < 5; ++t) q1[t] *= inv;
            }
          }
        }
        // (q/q_max + 1e-10) / (1 + 1e-10), :703-725; the maximum over all 18 elements is the maximum over these 8.
        // 1/q_max by v_rcp_f64 and two Newton steps (<= 1 ulp; the factor is common to the entry's likelihoods).  For
        // a skipped entry everything is 1 and must stay exactly 1.
        double mx = fmax(fmax(q0[0], q0[1]), q0[2]);
#pragma unroll
        for (int t = 0; t < 5; ++t) mx = fmax(mx, q1[t]);
        double x = __builtin_amdgcn_rcp(mx);
        x = fma(x, fma(-mx, x, 1.0), x);
        x = fma(x, fma(-mx, x, 1.0), x);
        const double cc = 1.0 / (1.0 + 1e-10);
        const double sc = (s >= 0) ? cc * x : 1.0, tt = (s >= 0) ? 1e-10 * cc : 0.0;
#pragma unroll
        for (int l = 0; l < 3; ++l) q0[l] = fma(q0[l], sc, tt);
#pragma unroll
        for (int t = 0; t < 5; ++t) q1[t] = fma(q1[t], sc, tt);
      }
      double* dst = pgs + slot * O_SLOT_STRIDE + p * 8;
      dst[0] = q0[0];
      dst[1] = q0[1];
      dst[2] = q0[2];
#pragma unroll
      for (int t = 0; t < 5; ++t) dst[3 + t] = q1[t];
      pgs[slot * O_SLOT_STRIDE + 8 * P + p] = hs_out;
      snps[slot * P + p] = (s >= 0) ? s : S_dummy;
      const int32_t sn = snp_of(precA, b + 1);
      snps_nx[slot * P + p] = (sn >= 0) ? sn : S_dummy;  // next batch; its no-genotype rows are (1,0,0)
    };

    // phase 2 for entry i of the slot's batch: lane <-> 2 samples
    auto sweep_entry = [&](const row_t& R, int i) __attribute__((always_inline)) {
      const double* qq = pgs + slot * O_SLOT_STRIDE + i * 8;
      const double a0 = qq[0], a1 = qq[1], a2 = qq[2];
      const double b0 = qq[3], b1 = qq[4], b2 = qq[5], b3 = qq[6], b4 = qq[7];
      if (!UNIT_S) accH *= pgs[slot * O_SLOT_STRIDE + 8 * P + i];
      // singlet slot llksAB[j][0][0] (:806,828), alpha = 0
      acc[G::acc_single(0)] *= fma(R.a[2], a2, fma(R.a[1], a1, R.a[0] * a0));
      acc[G::acc_single(1)] *= fma(R.b[2], a2, fma(R.b[1], a1, R.b[0] * a0));
      // u[m] = sum_l g_j[l] * pG[alpha=.5][l][m],  pG[l][m] = b[l+m]
      double ua[3], ub[3];
      ua[0] = fma(R.a[2], b2, fma(R.a[1], b1, R.a[0] * b0));
      ua[1] = fma(R.a[2], b3, fma(R.a[1], b2, R.a[0] * b1));
      ua[2] = fma(R.a[2], b4, fma(R.a[1], b3, R.a[0] * b2));
      ub[0] = fma(R.b[2], b2, fma(R.b[1], b1, R.b[0] * b0));
      ub[1] = fma(R.b[2], b3, fma(R.b[1], b2, R.b[0] * b1));
      ub[2] = fma(R.b[2], b4, fma(R.b[1], b3, R.b[0] * b2));
      auto dot = [](const double* g, const double* u) { return fma(g[2], u[2], fma(g[1], u[1], g[0] * u[0])); };  // :738-746
      acc[O_ACC_AB] *= dot(R.b, ua);
      wave_for<1, G::NROT + 1>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        double Pa[3], Pb[3];
        Pa[0] = dpp_rot<G::ror(t)>(R.a[0]);
        Pa[1] = dpp_rot<G::ror(t)>(R.a[1]);
        Pa[2] = dpp_rot<G::ror(t)>(R.a[2]);
        Pb[0] = dpp_rot<G::ror(t)>(R.b[0]);
        Pb[1] = dpp_rot<G::ror(t)>(R.b[1]);
        Pb[2] = dpp_rot<G::ror(t)>(R.b[2]);
        acc[G::acc_rot(t, 0, 0)] *= dot(Pa, ua);
        acc[G::acc_rot(t, 0, 1)] *= dot(Pb, ua);
        acc[G::acc_rot(t, 1, 0)] *= dot(Pa, ub);
        acc[G::acc_rot(t, 1, 1)] *= dot(Pb, ub);
      });
      {  // the lane facing this one: (a, b') here and (a', b) over there; (a, a') and (b, b') on both sides
        constexpr int CF = G::ror(P / 2);
        double Pa[3], Pb[3];
        Pa[0] = dpp_rot<CF>(R.a[0]);
        Pa[1] = dpp_rot<CF>(R.a[1]);
        Pa[2] = dpp_rot<CF>(R.a[2]);
        Pb[0] = dpp_rot<CF>(R.b[0]);
        Pb[1] = dpp_rot<CF>(R.b[1]);
        Pb[2] = dpp_rot<CF>(R.b[2]);
        acc[O_ACC_F_AB] *= dot(Pb, ua);
        acc[O_ACC_F_AA] *= dot(Pa, ua);
        acc[O_ACC_F_BB] *= dot(Pb, ub);
      }
    };

    // One batch: phase 1, then its P entries.  Rows are requested TWO entries ahead into three register sets that
    // rotate without copies; P entries against three sets give a period of three batches, which are written out, each
    // starting P mod 3 sets later.  On entry X holds the row of entry 0, Y of entry 1 (both requested earlier), Z is free.
    auto batch = [&](int b, row_t& X, row_t& Y, row_t& Z) __attribute__((always_inline)) {
      phase1(b);
      __syncthreads();
      const int32_t* sn = snps + slot * P;
      const int32_t* sx = snps_nx + slot * P;
      const int lim = (b == nb - 1) ? ngmax - b * O_BATCH : O_BATCH;  // the last batch ends with the longest list
      auto set = [&](auto ic) __attribute__((always_inline)) -> row_t& {
        constexpr int i = decltype(ic)::value % 3;
        if constexpr (i == 0) return X;
        else if constexpr (i == 1) return Y;
        else return Z;
      };
      bool go = true;  // (wave-uniform)
      wave_for<0, P>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        if (!go) return;
        load_row(set(std::integral_constant<int, e + 2>{}), e + 2 < P ? sn[e + 2 < P ? e + 2 : 0] : sx[e + 2 - P < 0 ? 0 : e + 2 - P]);
        __builtin_amdgcn_sched_barrier(0);
        sweep_entry(set(ec), e);
        __builtin_amdgcn_sched_barrier(0);
        if (lim <= e + 1) go = false;
      });
      if (P == 16 || (b & 1)) renorm();  // 16 entries per slot since the last renormalisation
      __syncthreads();
    };
    // (after a batch that started with (X, Y, Z) the next batch's entries 0 and 1 sit in the sets P mod 3 and P mod 3 + 1)

    row_t R0, R1, R2;
    snps_nx[slot * P + p] = (snp_of(precA, 0) >= 0) ? precA.snp : S_dummy;
    __syncthreads();
    load_row(R0, snps_nx[slot * P + 0]);
    load_row(R1, snps_nx[slot * P + 1]);
    __syncthreads();
    for (int b = 0; b < nb; b += 3) {
      batch(b, R0, R1, R2);
      if (b + 1 >= nb) break;
      if (P == 8) batch(b + 1, R2, R0, R1);
      else batch(b + 1, R1, R2, R0);
      if (b + 2 >= nb) break;
      if (P == 8) batch(b + 2, R1, R2, R0);
      else batch(b + 2, R2, R0, R1);
    }
    renorm();
  }

  // the sums of the linear entries: every hypothesis gets the products of its two samples (the partner's through the
  // same rotation that brought its values); every singlet the product of sample 0's sums, which for the linear entries
  // is accW of position 0's sample a, for the others accH (:806)
  {
    int32_t exa[ON_ACC];
    const double wa = accW[0], wb = accW[1];
    const int32_t ea = exW[0], eb = exW[1];
    const int src0 = G::lane0(lane);  // position 0 of the same slot
    const double W0 = __shfl(wa, src0, 64) * accH;
    const int32_t e0w = __shfl(ea, src0, 64) + exH;
    acc[G::acc_single(0)] *= wa * W0;
    exa[G::acc_single(0)] = ea + e0w;
    acc[G::acc_single(1)] *= wb * W0;
    exa[G::acc_single(1)] = eb + e0w;
    acc[O_ACC_AB] *= wa * wb;
    exa[O_ACC_AB] = ea + eb;
    wave_for<1, G::NROT + 1>([&](auto tc) {
      constexpr int T = decltype(tc)::value, CTRL = G::ror(T);
      const double pa = dpp_rot<CTRL>(wa), pb = dpp_rot<CTRL>(wb);
      const int32_t qa = __builtin_amdgcn_mov_dpp(ea, CTRL, 0xF, 0xF, false);
      const int32_t qb = __builtin_amdgcn_mov_dpp(eb, CTRL, 0xF, 0xF, false);
      acc[G::acc_rot(T, 0, 0)] *= wa * pa;
      exa[G::acc_rot(T, 0, 0)] = ea + qa;
      acc[G::acc_rot(T, 0, 1)] *= wa * pb;
      exa[G::acc_rot(T, 0, 1)] = ea + qb;
      acc[G::acc_rot(T, 1, 0)] *= wb * pa;
      exa[G::acc_rot(T, 1, 0)] = eb + qa;
      acc[G::acc_rot(T, 1, 1)] *= wb * pb;
      exa[G::acc_rot(T, 1, 1)] = eb + qb;
    });
    {
      constexpr int CF = G::ror(P / 2);
      const double pa = dpp_rot<CF>(wa), pb = dpp_rot<CF>(wb);
      const int32_t qa = __builtin_amdgcn_mov_dpp(ea, CF, 0xF, 0xF, false);
      const int32_t qb = __builtin_amdgcn_mov_dpp(eb, CF, 0xF, 0xF, false);
      acc[O_ACC_F_AB] *= wa * pb;
      exa[O_ACC_F_AB] = ea + qb;
      acc[O_ACC_F_AA] *= wa * pa;
      exa[O_ACC_F_AA] = ea + qa;
      acc[O_ACC_F_BB] *= wb * pb;
      exa[O_ACC_F_BB] = eb + qb;
    }
    if (q < n_chunks) {
#pragma unroll
      for (int a = 0; a < ON_ACC; ++a) {
        int e;
        const double m = frexp(acc[a], &e);
        part_m[((size_t)qpos * ON_ACC + a) * P + p] = m;
        part_e[((size_t)qpos * ON_ACC + a) * P + p] = exs[a] + e + exa[a];
      }
    }
  }
}

// Decodes accumulator idx = a * P + p into its hypothesis (j, k); false for slots nobody reads (pairs held twice,
// j / k >= V).
template <int P>
__device__ __forceinline__ bool oct_decode(int idx, const int32_t* __restrict__ pmap, int V, int& j, int& k) {
  using G = og<P>;
  const int a = idx / P, p = idx % P;
  bool publish = true;
  if (a < 2) {
    j = p + P * a;
    k = 0;
  } else if (a == G::ACC_AB) {
    j = p + P;
    k = p;
  } else if (a < G::ACC_F_AB) {
    const int t = (a - 3) >> 2, c = ((a - 3) >> 1) & 1, d = (a - 3) & 1;
    j = p + P * c;
    k = pmap[t * P + p] + P * d;
  } else {
    const int pf = pmap[G::NROT * P + p];
    if (a == G::ACC_F_AB) {
      j = p;
      k = pf + P;
    } else if (a == G::ACC_F_AA) {
      j = p;
      k = pf;
      publish = p < pf;  // the facing lane holds the same pair
    } else {
      j = p + P;
      k = pf + P;
      publish = p < pf;
    }
  }
  return publish && j < V && k < V;
}

// log(m 2^e), spelled with an explicit fma: the reduce kernel and the finish kernel must give the same bits, and whether
// "a + b * c" is contracted is the compiler's choice per site
__device__ __forceinline__ double oct_log(double m, int64_t e) { return pos_log(m, (double)e); }

// Multiplies the chunk partials of one cell in chunk order: ONE log per hypothesis.  A chunk's partials sit at the chunk's
// position ci in its cell's list (the sweep writes them there, oct_chunk_pos_kernel), so a cell's are consecutive and a
// reader needs no chunk ids.
template <int P>
__device__ __forceinline__ bool oct_hypothesis(int idx, int64_t c0, int64_t c1, const double* __restrict__ part_m,
                                               const int32_t* __restrict__ part_e, const int32_t* __restrict__ pmap, int V,
                                               int& j, int& k, double& v) {
  constexpr int O_NHYP = og<P>::N_HYP;
  if (!oct_decode<P>(idx, pmap, V, j, k)) return false;
  // eight chunks per trip: the sixteen loads are independent and in flight together, the products stay in chunk order
  double m = 1.0;
  int64_t e = 0;
  int cnt = 0;
  for (int64_t ci = c0; ci < c1; ci += 8) {
    double pm[8];
    int32_t pe[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool ok = ci + u < c1;
      const size_t o = (size_t)(ok ? ci + u : c0) * O_NHYP + idx;
      pm[u] = ok ? part_m[o] : 1.0;
      pe[u] = ok ? part_e[o] : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      m *= pm[u];
      e += pe[u];
    }
    if (++cnt == 64) {  // mantissas are in [0.5,1): 512 factors cannot underflow
      cnt = 0;
      int ee;
      m = frexp(m, &ee);
      e += ee;
    }
  }
  v = oct_log(m, e);
  return true;
}

// position of every chunk in its cell's list (the inverse of cell_chunks)
__global__ void __launch_bounds__(256)
    oct_chunk_pos_kernel(int64_t n_chunks, const int32_t* __restrict__ cell_chunks, int32_t* __restrict__ pos) {
  const int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ci < n_chunks) pos[cell_chunks[ci]] = (int32_t)ci;
}

// writes ll[c][j][k][n] (+ mirror) of one cell to the LL tensor in HBM (needed when the caller asks for the tensor)
template <int P>
__global__ void __launch_bounds__(192)
    demux_oct_reduce_kernel(const int64_t* __restrict__ cell_chunk_ptr, const double* __restrict__ part_m,
                            const int32_t* __restrict__ part_e, const int32_t* __restrict__ pmap, int V,
                            double* __restrict__ ll) {
  const int64_t c = blockIdx.x;
  const int64_t c0 = cell_chunk_ptr[c], c1 = cell_chunk_ptr[c + 1];
  if (c0 == c1) return;
  double* out = ll + (size_t)c * V * V * 2;
  for (int idx = threadIdx.x; idx < og<P>::N_HYP; idx += 192) {
    int j, k;
    double v;
    if (!oct_hypothesis<P>(idx, c0, c1, part_m, part_e, pmap, V, j, k, v)) continue;
    if (idx < 2 * P) {
      out[((size_t)j * V + k) * 2 + 0] = v;  // singlet: alpha index 0
    } else {
      out[((size_t)j * V + k) * 2 + 1] = v;
      out[((size_t)k * V + j) * 2 + 1] = v;
    }
  }
}

// The same reduction, but the hypotheses of the workgroup's cells (four at P = 8, one at P = 16) stay in LDS and the call
// (demux_call_body.hpp) follows at once: no LL tensor round trip through HBM, one launch less, and the 160-byte
// records go straight to the caller's pinned host buffer (16-byte stores of consecutive lanes), which removes the
// separate device-to-host copy.
#ifndef QF_CELLS_N
#define QF_CELLS_N 4
#endif
template <int P>
__global__ void __launch_bounds__(256)
    demux_oct_finish_kernel(int64_t C, const int64_t* __restrict__ cell_ptr, const int64_t* __restrict__ cell_chunk_ptr,
                            const double* __restrict__ part_m, const int32_t* __restrict__ part_e,
                            const int32_t* __restrict__ pmap, int V,
                            muxgl_call::call_alpha al, double doublet_prior, muxgl_demux_cell* __restrict__ out) {
  constexpr int QF_CELLS = P == 8 ? QF_CELLS_N : 1, THREADS = 256, O_NHYP = og<P>::N_HYP;
  static_assert(P == 16 || QF_CELLS * 64 == THREADS, "a wave per cell at P = 8");
  constexpr int LD = 4 * P + 1;  // row stride of the tiles in doubles (2P samples x 2 alphas): odd, so that the lanes of a cell's
                                 // call, which read the same column of their rows at once, meet different LDS banks
  __shared__ double llt[QF_CELLS][2 * P * LD];
  __shared__ __align__(16) muxgl_demux_cell rec[QF_CELLS];
  __shared__ int64_t ccp[QF_CELLS + 1];
  static_assert(sizeof(muxgl_demux_cell) % 16 == 0, "records are copied out in 16-byte pieces");
  const int64_t cbase = (int64_t)blockIdx.x * QF_CELLS;
  const int tid = threadIdx.x;
  if (tid <= QF_CELLS) ccp[tid] = cell_chunk_ptr[cbase + tid <= C ? cbase + tid : C];
  for (int t = tid; t < QF_CELLS * 2 * P * LD; t += THREADS) (&llt[0][0])[t] = 0.0;
  __syncthreads();
  // The kernel's time is a chain of dependent latencies (measured: the same 65 us for 5 000 cells as for 10 000), so the
  // chain is kept short: the chunk ranges of the workgroup's cells come from ONE load (above), a cell's partials are
  // consecutive (no chunk ids), and a thread's hypotheses -- NH of the workgroup's QF_CELLS x N_HYP -- are walked TOGETHER,
  // four chunks a trip: 2 x 4 x NH loads in flight instead of one hypothesis after the other (three dependent trips each).
  constexpr int NH = (QF_CELLS * O_NHYP + THREADS - 1) / THREADS;
  int hj[NH], hk[NH], hidx[NH], hlc[NH];
  int64_t h0[NH];
  int hn[NH];
  bool hv[NH];
  double hm[NH];
  int64_t he[NH];
  int nmax = 0;
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int w = tid + i * THREADS;
    const int lc = w < QF_CELLS * O_NHYP ? w / O_NHYP : 0;
    hlc[i] = lc;
    hidx[i] = w - lc * O_NHYP;
    h0[i] = ccp[lc];
    hn[i] = (int)(ccp[lc + 1] - ccp[lc]);
    hv[i] = w < QF_CELLS * O_NHYP && hn[i] > 0 && oct_decode<P>(hidx[i] < O_NHYP ? hidx[i] : 0, pmap, V, hj[i], hk[i]);
    if (!hv[i]) hn[i] = 0, hidx[i] = 0;
    hm[i] = 1.0;
    he[i] = 0;
    nmax = hn[i] > nmax ? hn[i] : nmax;
  }
  int cnt = 0;
  for (int ci = 0; ci < nmax; ci += 4) {
    double pm[NH][4];
    int32_t pe[NH][4];
#pragma unroll
    for (int i = 0; i < NH; ++i)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = ci + u < hn[i];
        const size_t o = (ok ? (size_t)(h0[i] + ci + u) * O_NHYP : 0) + hidx[i];  // (a valid address either way)
        const double vm = part_m[o];
        const int32_t ve = part_e[o];
        pm[i][u] = ok ? vm : 1.0;
        pe[i][u] = ok ? ve : 0;
      }
#pragma unroll
    for (int i = 0; i < NH; ++i)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        hm[i] *= pm[i][u];
        he[i] += pe[i][u];
      }
    if (++cnt == 128) {  // mantissas are in [0.5,1): 512 factors cannot underflow (as oct_hypothesis)
      cnt = 0;
#pragma unroll
      for (int i = 0; i < NH; ++i) {
        int ee;
        hm[i] = frexp(hm[i], &ee);
        he[i] += ee;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    if (!hv[i]) continue;
    const double v = oct_log(hm[i], he[i]);
    if (hidx[i] < 2 * P) {
      llt[hlc[i]][hj[i] * LD + hk[i] * 2 + 0] = v;
    } else {
      llt[hlc[i]][hj[i] * LD + hk[i] * 2 + 1] = v;
      llt[hlc[i]][hk[i] * LD + hj[i] * 2 + 1] = v;
    }
  }
  __syncthreads();
  if constexpr (P == 8) {  // a wave per cell for the scans (sixteen rows x four ranges of columns); the cells' decisions side by side
    __shared__ muxgl_call::call_partial parts[QF_CELLS];
    const int lc = tid >> 6;
    const int64_t c = cbase + lc;
    const muxgl_call::call_partial cp = muxgl_call::demux_call_scan<16, 4>(tid & 63, c < C, V, 2, al, llt[lc], LD);
    if ((tid & 63) == 0) parts[lc] = cp;
    __syncthreads();
    if (tid < QF_CELLS && cbase + tid < C)
      muxgl_call::demux_call_decide(parts[tid], (int32_t)(cell_ptr[cbase + tid + 1] - cell_ptr[cbase + tid]), V, 2, al,
                                    &rec[tid]);
  } else {  // one cell, wave 0, lane = row: the association of demux_callg_kernel, which makes the call from the tensor
    if (tid < 64)
      muxgl_call::demux_call_group<64>(tid, cbase < C, cbase < C ? (int32_t)(cell_ptr[cbase + 1] - cell_ptr[cbase]) : 0, V, 2,
                                       al, doublet_prior, llt[0], &rec[0], LD);
  }
  __syncthreads();
  constexpr int NQ = (int)(sizeof(muxgl_demux_cell) / 16);
  if (tid < QF_CELLS * NQ && cbase + tid / NQ < C)
    reinterpret_cast<uint4*>(out + cbase)[tid] = reinterpret_cast<const uint4*>(&rec[0])[tid];
}

}  // namespace

// the launch order of the chunks of the oct / quad kernels (quad_order_key_kernel), shared with fmx_quad.hip
int quad_launch_order(muxgl_handle* h, const row_chunk* d_chunks, const int32_t* d_nlin, int64_t n, int32_t** order) {
  int32_t* d_iota = nullptr;
  uint64_t *d_key = nullptr, *d_key2 = nullptr;
  void* d_tmp = nullptr;
  auto cleanup = [&]() {
    dev_free(&d_iota);
    dev_free(&d_key);
    dev_free(&d_key2);
    if (d_tmp) (void)hipFree(d_tmp);
    d_tmp = nullptr;
  };
  if (dev_alloc(h, order, (size_t)n) || dev_alloc(h, &d_iota, (size_t)n) || dev_alloc(h, &d_key, (size_t)n) ||
      dev_alloc(h, &d_key2, (size_t)n)) {
    cleanup();
    return 1;
  }
  int bucket = QUAD_BUCKET;
  if (const char* ev = getenv("MUXGL_QUAD_BUCKET")) bucket = atoi(ev) > 0 ? atoi(ev) : bucket;  // (tuning)
  if (bucket > (1 << 20)) bucket = 1 << 20;
  hipLaunchKernelGGL(quad_order_key_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (int)n, bucket, d_chunks,
                     d_nlin, d_key, d_iota);
  size_t tmp_bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_key, d_key2, d_iota, *order, (size_t)n, 0u, 64u, h->stream);
  if (e == hipSuccess) e = dev_malloc_retry((void**)&d_tmp, tmp_bytes ? tmp_bytes : 1);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_key, d_key2, d_iota, *order, (size_t)n, 0u, 64u, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  cleanup();
  if (e != hipSuccess) MUXGL_FAIL(h, "quad_launch_order: %s", hipGetErrorString(e));
  return 0;
}

namespace {
template <int P>
int oct_launch_t(muxgl_handle* h, const muxgl_demux_params* p) {
  constexpr int O_SLOTS = og<P>::SLOTS, O_NHYP = og<P>::N_HYP;
  muxgl_row_state* st = h->qrow;
  if (!st->d_tmap || st->tmap_p != P) {  // positions met by the rotations
    if (dev_alloc(h, &st->d_tmap, (size_t)(P / 2) * P)) return 1;
    hipLaunchKernelGGL(oct_pmap_kernel<P>, dim3(1), dim3(64), 0, h->stream, st->d_tmap);
    HIPCHK(h, hipGetLastError());
    st->tmap_p = P;
  }
  const size_t need = (size_t)st->n_chunks * O_NHYP;
  if (need > st->part_cap) {
    if (dev_alloc(h, &st->d_part, need)) return 1;
    st->part_cap = need;
  }
  if (need > st->part_e_cap) {
    if (dev_alloc(h, &st->d_part_e, need)) return 1;
    st->part_e_cap = need;
  }
  if (!st->d_chunk_pos) {  // where a chunk's partials go (oct_hypothesis)
    if (dev_alloc(h, &st->d_chunk_pos, (size_t)(st->n_chunks ? st->n_chunks : 1))) return 1;
    if (st->n_chunks)
      hipLaunchKernelGGL(oct_chunk_pos_kernel, dim3((unsigned)((st->n_chunks + 255) / 256)), dim3(256), 0, h->stream, st->n_chunks,
                         st->d_cell_chunks, st->d_chunk_pos);
    HIPCHK(h, hipGetLastError());
  }
  const bool use_lin = h->d_lin && h->d_gmq && !(h->flags & MUXGL_FLAG_NO_LINEAR_ENTRIES);
  const unsigned blocks = (unsigned)((((st->n_chunks + O_SLOTS - 1) / O_SLOTS) + 7) / 8 * 8);  // multiple of 8 for xcd_swizzle
  if (use_lin && !st->d_chunk_nlin && st->n_chunks) {  // once per pileup and GP tensor: every chunk's linear entries first
    quad_lrec* d_lrec = nullptr;  // chunk-major records, repacked below
    int32_t* d_steps = nullptr;
    auto cleanup = [&]() {
      dev_free(&d_lrec);
      dev_free(&d_steps);
    };
    if (dev_alloc(h, &st->d_qent_lin, (size_t)h->nnz) || dev_alloc(h, &st->d_chunk_nlin, (size_t)st->n_chunks) ||
        dev_alloc(h, &d_lrec, (size_t)h->nnz)) {
      cleanup();
      return 1;
    }
    hipLaunchKernelGGL(oct_partition_kernel, dim3((unsigned)((st->n_chunks + 63) / 64)), dim3(64), 0, h->stream,
                       (int)st->n_chunks, st->d_chunks, h->d_qent, h->d_lin, h->d_reads, h->d_gp0s, st->d_qent_lin,
                       d_lrec, st->d_chunk_nlin);
    if (hipGetLastError() != hipSuccess || quad_launch_order(h, st->d_chunks, st->d_chunk_nlin, st->n_chunks, &st->d_quad_order)) {
      cleanup();
      MUXGL_FAIL(h, "demux_oct_launch: partition failed");
    }
    // the linear entries' records, step-major per unit (what the sweep streams)
    const int n_units = (int)blocks;
    std::vector<int32_t> steps((size_t)n_units);
    std::vector<int64_t> uptr((size_t)n_units + 1, 0);
    if (dev_alloc(h, &d_steps, (size_t)n_units)) {
      cleanup();
      return 1;
    }
    hipLaunchKernelGGL(oct_unit_steps_kernel, dim3((unsigned)((n_units + 255) / 256)), dim3(256), 0, h->stream, n_units,
                       (int)st->n_chunks, O_SLOTS, st->d_quad_order, st->d_chunk_nlin, d_steps);
    hipError_t e = hipMemcpyAsync(steps.data(), d_steps, sizeof(int32_t) * (size_t)n_units, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    for (int u = 0; u < n_units; ++u) uptr[(size_t)u + 1] = uptr[(size_t)u] + (int64_t)steps[(size_t)u] * O_SLOTS;
    if (e != hipSuccess || dev_alloc(h, &st->d_unit_ptr, uptr.size()) || dev_alloc(h, &st->d_orec, (size_t)uptr.back() + 1)) {
      cleanup();
      MUXGL_FAIL(h, "demux_oct_launch: record tables");
    }
    e = hipMemcpyAsync(st->d_unit_ptr, uptr.data(), sizeof(int64_t) * uptr.size(), hipMemcpyHostToDevice, h->stream);
    hipLaunchKernelGGL(oct_repack_kernel, dim3((unsigned)(((size_t)n_units * O_SLOTS + 255) / 256)), dim3(256), 0, h->stream,
                       n_units, (int)st->n_chunks, st->d_chunks, st->d_quad_order, st->d_chunk_nlin, d_lrec, d_steps,
                       st->d_unit_ptr, (uint32_t)h->S, O_SLOTS, og<P>::MROW, st->d_orec);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);  // (uptr is a host buffer; d_lrec is freed)
    cleanup();
    if (e != hipSuccess) MUXGL_FAIL(h, "demux_oct_launch: %s", hipGetErrorString(e));
  }
  tic(h, MUXGL_T_DEMUX_SWEEP);
  if (blocks) {
    auto kern = h->gp_unit_sums ? demux_oct_kernel<P, true> : demux_oct_kernel<P, false>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0,
                       h->stream, st->d_chunks, (int)st->n_chunks,
                       use_lin ? st->d_qent_lin : h->d_qent, st->d_orec, st->d_unit_ptr,
                       use_lin ? st->d_chunk_nlin : (const int32_t*)nullptr,
                       use_lin ? st->d_quad_order : (const int32_t*)nullptr, h->d_reads,
                       h->d_gpq, h->d_gmq, h->d_gp0s, (int32_t)h->S, h->d_lut, st->d_chunk_pos, st->d_part, st->d_part_e);
    HIPCHK(h, hipGetLastError());
  }
  toc_tic(h, MUXGL_T_DEMUX_SWEEP, MUXGL_T_DEMUX_REDUCE);
  if (h->want_full_ll) {
    hipLaunchKernelGGL(demux_oct_reduce_kernel<P>, dim3((unsigned)h->C), dim3(192), 0, h->stream, st->d_cell_chunk_ptr,
                       st->d_part, st->d_part_e, st->d_tmap, h->V, h->d_ll);
  } else {  // reduce + call fused, records written to the pinned host buffer
    const muxgl_call::call_alpha al = muxgl_call::make_call_alpha(p, h->V);
    constexpr int QF_CELLS = P == 8 ? QF_CELLS_N : 1;
    hipLaunchKernelGGL(demux_oct_finish_kernel<P>, dim3((unsigned)((h->C + QF_CELLS - 1) / QF_CELLS)), dim3(256), 0,
                       h->stream, h->C, h->d_cell_ptr, st->d_cell_chunk_ptr, st->d_part, st->d_part_e, st->d_tmap, h->V, al,
                       p->doublet_prior, h->h_dcells);
    h->records_on_host = true;
  }
  HIPCHK(h, hipGetLastError());
  toc(h, MUXGL_T_DEMUX_REDUCE);
  return 0;
}
}  // namespace

// returns -1 when the path does not apply, 0 ok, 1 error
int demux_oct_launch(muxgl_handle* h, const muxgl_demux_params* p) {
  if (h->V > 32 || !h->qrow || !h->d_gpq || !h->d_qent || h->C == 0) return -1;
  const int P = h->V <= 16 ? 8 : 16;
  // (row offsets of the linear entries' records are 32-bit byte offsets)
  if ((uint64_t)(h->S + 1) * (32u * (unsigned)P) >= ((uint64_t)1 << 32)) return -1;
  if (h->flags & (MUXGL_FLAG_FORCE_TILE_SWEEP | MUXGL_FLAG_FORCE_ROW_KERNEL | MUXGL_FLAG_FORCE_WAVE_KERNEL)) return -1;
  if (p->n_alpha != 2 || p->alpha[0] != 0.0 || p->alpha[1] != 0.5) return -1;
  return P == 8 ? oct_launch_t<8>(h, p) : oct_launch_t<16>(h, p);
}
