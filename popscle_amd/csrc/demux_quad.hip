// demux_quad.hip -- the demuxlet sweep for the reference's DEFAULT configuration: V <= 16 samples and the alpha grid
// {0, 0.5} that cmdCramDemuxlet installs when no --alpha is given (cmd_cram_demuxlet.cpp:85-89).
//
// Reference being replaced: cmd_cram_demuxlet.cpp:655-747.  Same mathematics as demux_row.hip; what changes is the
// register tiling, chosen because the row kernel turned out VALU-issue bound (94 % VALU busy) with 44 % of its vector
// instructions being DPP moves and per-iteration bookkeeping rather than FP64 arithmetic:
//
//   * 4 lanes per entry instead of 16: lane r of a quad owns the FOUR samples 4r..4r+3 (12 doubles, one contiguous
//     96-byte piece of the GP row), so a wave sweeps 16 entries per iteration instead of 4 and every per-iteration
//     cost (LDS reads of the entry's likelihoods, address arithmetic, waits) is amortised over 4x more entries.
//   * pairs inside a lane need no data movement (6 pairs); the other tiles arrive with two DPP rotations of the
//     16-lane row (row_ror:4 -> neighbouring tile, 16 pairs; row_ror:8 -> opposite tile, 10 pairs with c <= d so that
//     the two lanes facing each other split the tile pair) : 3 DPP moves per entry instead of 12.
//   * the grid {0, 0.5} has structure the general code cannot assume: for alpha = 0 the mixing proportion is
//     p = l/2, independent of m (3 distinct likelihoods per entry instead of 9), and for alpha = 0.5 it is
//     p = (l+m)/4 (5 distinct instead of 9).  Phase 1 therefore carries 8 products per read instead of 18, and the
//     singlet slot factorises into (sum_l g_j[l] q0[l]) * (g_0[0]+g_0[1]+g_0[2]).
//   * products leave the kernel as (mantissa, exponent) pairs; demux_quad_reduce_kernel multiplies the chunk partials
//     of a cell in chunk order and takes ONE log per hypothesis and cell (the row kernel took one per chunk).
//   * entries with at most one usable read (three quarters of a typical pileup) are linear in the genotypes: a chunk's
//     records are partitioned (quad_partition_kernel), the linear ones are swept first by a loop of their own from rows of
//     moments (s, rho) -- one FMA and the product update per hypothesis, the sums s folded in at the end -- and the
//     launch order sorts chunks of similar trip counts into the same waves (quad_order_key_kernel).
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "demux_call_body.hpp"

namespace {

#ifndef QUAD_EXP
#define QUAD_EXP 0  // timing experiments (tools/gpu_quadx.sh): 1 no nine-term loop, 2 no linear loop, 4 hot rows, 8 no renorm
#endif
#ifndef QL_TOUCH
#define QL_TOUCH 0  // records ahead (0: off -- measured: one more load per entry costs 10 % of the sweep)
#endif
constexpr int QL_SLACK = 256;   // records the linear loop may read behind the last chunk's list
constexpr int QN_ACC = 36;      // per lane: 4 singlets, 6 in-lane pairs, 16 pairs with the neighbour tile, 10 with the opposite
constexpr int Q_SLOT_STRIDE = 38;  // doubles: 4 entries x 8 likelihoods + 4 singlet factors + 2 pad => the 16 slots of a
                                   // wave hit distinct banks (76 dwords apart)

__device__ __forceinline__ double dpp_ror4(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x124, 0xF, 0xF, false);  // row_ror:4
  hi = __builtin_amdgcn_mov_dpp(hi, 0x124, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_ror8(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x128, 0xF, 0xF, false);  // row_ror:8
  hi = __builtin_amdgcn_mov_dpp(hi, 0x128, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// tile (= lane>>2 & 3) whose samples a lane sees after row_ror:4 / row_ror:8, measured rather than assumed
__global__ void quad_tmap_kernel(int32_t* tmap /*[2][4]*/) {
  const int lane = threadIdx.x;
  int t = (lane >> 2) & 3;
  const int t4 = __builtin_amdgcn_mov_dpp(t, 0x124, 0xF, 0xF, false);
  const int t8 = __builtin_amdgcn_mov_dpp(t, 0x128, 0xF, 0xF, false);
  if (lane < 16 && (lane & 3) == 0) {
    tmap[t] = t4;
    tmap[4 + t] = t8;
  }
}

// The likelihoods of a linear entry depend on ONE read byte (quad_lrec::code, common.hpp; 256 values: a table in LDS),
// so the sweep needs neither the read array nor a per-entry phase 1 for such entries.

// A chunk's entry records with its linear entries (at most one usable read, plan_kernels.hip: lin_kernel) first, both
// kinds in entry order, and the number of linear ones; the linear ones also as quad_lrec.  One thread per chunk
// (<= 128 records), once per pileup and GP tensor (has_gp enters the codes).
__global__ void __launch_bounds__(64)
    quad_partition_kernel(int n_chunks, const row_chunk* __restrict__ chunks, const quad_entry* __restrict__ qent,
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

// Launch order: a wave's trip counts are the maxima over its 16 chunks, so within every bucket of QUAD_BUCKET
// consecutive chunks of the plan's order (non-increasing length, then first SNP -- the rows gathered by co-resident
// workgroups stay a sliding window at that grain) the chunks are sorted by their number of non-linear batches.
constexpr int QUAD_BUCKET = 1024;
__global__ void __launch_bounds__(256)
    quad_order_key_kernel(int n_chunks, int bucket, const row_chunk* __restrict__ chunks,
                          const int32_t* __restrict__ chunk_nlin, uint64_t* __restrict__ key, int32_t* __restrict__ iota) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_chunks) return;
  const int nl = chunk_nlin[w], len = chunks[w].len;
  const uint64_t bg = (uint64_t)((len - nl + 3) >> 2), bl = (uint64_t)((nl + 3) >> 2);  // <= 32 each
  key[w] = ((uint64_t)(w / bucket) << 36) | ((63u - bg) << 26) | ((63u - bl) << 20) | (uint64_t)(w % bucket);
  iota[w] = w;
}

// accumulator index layout
__host__ __device__ constexpr int q_acc_single(int c) { return c; }
__host__ __device__ constexpr int q_acc_within(int c1, int c2) {  // c1 < c2
  return 4 + (c1 == 0 ? c2 - 1 : (c1 == 1 ? 3 + c2 - 2 : 5));
}
__host__ __device__ constexpr int q_acc_t1(int c, int d) { return 10 + c * 4 + d; }
__host__ __device__ constexpr int q_acc_t2(int c, int d) {  // c <= d
  return 26 + (c == 0 ? d : (c == 1 ? 4 + d - 1 : (c == 2 ? 7 + d - 2 : 9)));
}

// The sweep kernel.  GP rows are requested TWO entries ahead into three register sets that rotate without copies (a
// copy would wait for the load it copies); four entries per batch against three sets gives a period of twelve entries,
// so three batches are written out, each starting one set later.  The 36 integer exponents of the accumulators live in
// LDS (touched once per 16 entries) to make room for the third set.  The (empty) asm statements pin program order:
// everything that reads a set is complete before the set is reloaded, and a reload is issued before the sweep behind
// it -- otherwise the scheduler renames registers to hoist loads, runs out of VGPRs and spills.
//
// What the time is made of (config 1, ablations of the nine-term loop on MI355X, round 1): FP64 arithmetic alone 0.28 ms,
// + DPP moves 0.06, + phase 1 0.09, + row gathers 0.12 = 0.55 ms at the clocks of a 20-launch run (0.49 ms sustained).
// The parts add up instead of overlapping, and neither a fifth fewer instructions (dummy row instead of predicated
// loads), nor deeper prefetch, nor a branch-free phase 1 moved the total by more than 2 %.  (Not the power cap: the
// launch runs at 2.4 GHz and 1.2 kW of 1.4.)  What did move it in round 2 is less work per entry: the loop of the
// linear entries (0.49 -> 0.43 ms) and the launch order (-> 0.41 ms); a whole batch of row look-ahead and the linear
// entries' phase 1 done once per pileup instead of per launch changed nothing again.
__global__ void __launch_bounds__(64, 2)
    demux_quad_kernel(const row_chunk* __restrict__ chunks, int n_chunks, const quad_entry* __restrict__ qent,
                       const quad_lrec* __restrict__ qlrec,
                       const int32_t* __restrict__ chunk_nlin, const int32_t* __restrict__ order,
                       const uint8_t* __restrict__ reads,
                       const double* __restrict__ gpq, const double* __restrict__ gmq,
                       const double* __restrict__ gp0s, int32_t S_dummy, const double* __restrict__ lut_g,
                       double* __restrict__ part_m, int32_t* __restrict__ part_e) {
  __shared__ double lut[384];
  __shared__ __align__(16) double ablut[256 * 2];
  __shared__ __align__(16) double pgs[16 * Q_SLOT_STRIDE];
  __shared__ int32_t snps[64], snps_nx[64];
  // exponents of the accumulators since the start of the chunk: a factor is >= 1e-10 / (1 + 1e-10) > 2^-34, a chunk
  // has at most MUXGL_QUAD_CH of them
  __shared__ int16_t exs[QN_ACC][64];
  static_assert(MUXGL_QUAD_CH * 36 < 32000, "accumulator exponents are kept as int16");

  const int lane = threadIdx.x;
  const int r = (lane >> 2) & 3;                       // tile: samples 4r..4r+3
  const int slot = ((lane >> 4) << 2) | (lane & 3);    // 16 entry streams per wave
  for (int i = lane; i < 384; i += 64) lut[i] = lut_g[i];
  // (A, B) of a linear entry by the read byte that counts (quad_lrec::code): the one factor pR + (pA - pR) p of
  // :673,685 through the tail (q / q_max + 1e-10) / (1 + 1e-10) of :703-725; q0[l] = A + 2B l, q1[l+m] = A + B (l+m)
  for (int i = lane; i < 256; i += 64) {
    const uint32_t bq = (uint32_t)i & 0x7f;
    const bool ref = (i >> 7) == 0;
    const double e3 = lut_g[256 + bq], mt = lut_g[128 + bq];
    const double pR = ref ? mt : e3, pA = ref ? e3 : mt;  // :666-667
    const double mx = fmax(pR, pA);
    double x = __builtin_amdgcn_rcp(mx);
    x = fma(x, fma(-mx, x, 1.0), x);
    x = fma(x, fma(-mx, x, 1.0), x);
    const double cc = 1.0 / (1.0 + 1e-10);
    const double sc = cc * x, tt = 1e-10 * cc;
    const bool none = i == MUXGL_READ_OTHER;  // no usable read / no genotypes: factors of exactly 1
    ablut[2 * i] = none ? 1.0 : fma(pR, sc, tt);
    ablut[2 * i + 1] = none ? 0.0 : (pA - pR) * (0.25 * sc);
  }
#pragma unroll
  for (int a = 0; a < QN_ACC; ++a) exs[a][lane] = 0;

  const int wq = xcd_swizzle(blockIdx.x, gridDim.x >> 3) * 16 + slot;  // place in the launch order
  const int q = wq < n_chunks ? (order ? order[wq] : wq) : n_chunks;
  int64_t e0 = 0;
  int len = 0;
  if (q < n_chunks) {
    e0 = chunks[q].e0;
    len = chunks[q].len;
  }
  // The chunk's first nl records are its linear entries (quad_partition_kernel; 0 when that form is off): they are swept
  // first, by a loop of their own (below), the others by the nine-term loop.  Trip counts of the wave = the longest
  // run of either kind among its 16 chunks.
  const int nl = (chunk_nlin && q < n_chunks) ? chunk_nlin[q] : 0;
  const int nLmax = (QUAD_EXP & 2) ? 0 : wave_max_i32(nl);
  const int nb = (QUAD_EXP & 1) ? 0 : (wave_max_i32(len - nl) + 3) >> 2;

  double acc[QN_ACC];
#pragma unroll
  for (int a = 0; a < QN_ACC; ++a) acc[a] = 1.0;
  double accW[4] = {1.0, 1.0, 1.0, 1.0};  // linear entries: products of the sums s of the lane's four samples
  int32_t exW[4] = {0, 0, 0, 0};

  // records {snp, read count, first four read bytes, read offset} of the lane's own entry: batch b in precA, b+1 in
  // precB, b+2 requested during phase 1 of batch b
  // (loaded unconditionally from a clamped index and invalidated where it is used: a predicated load followed by a
  // merge with the defaults makes the compiler wait for the load on the spot)
  const int last = len > 0 ? len - 1 : 0;
  auto fetch_meta = [&](int b) {
    const int idx = nl + b * 4 + r;
    return qent[e0 + (idx < last ? idx : last)];
  };
  auto snp_of = [&](const quad_entry& p, int b) { return (nl + b * 4 + r < len) ? p.snp : -1; };
  quad_entry precA = fetch_meta(0), precB = fetch_meta(1);
  // sum of sample 0's triple at the lane's own entry (negative: marker without genotypes), one batch ahead as well
  double hs_cur = gp0s[precA.snp];

  struct row_t {
    double G[4][3];
  };
  auto load_row = [&](row_t& R, int32_t s) {
    // six 16-byte pieces of this lane's 12 doubles; piece t of the quad's four lanes is 64 contiguous bytes.  Rows of
    // padding entries and of markers without genotypes are (1,0,0) -- the dummy row S_dummy resp. the host's fill --
    // which together with read likelihoods of 1 (phase 1) makes every factor of such an entry exactly 1 (:733).
    const double2* pc = reinterpret_cast<const double2*>(gpq + (size_t)s * 48) + r;
    double f[12];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const double2 v = pc[t * 4];
      f[2 * t] = v.x;
      f[2 * t + 1] = v.y;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      R.G[c][0] = f[3 * c];
      R.G[c][1] = f[3 * c + 1];
      R.G[c][2] = f[3 * c + 2];
    }
  };

  // ---- phase 1 of a batch: lane <-> entry.  cmd_cram_demuxlet.cpp:655-725 for alpha in {0, 0.5}:
  //      q0[l] = prod_reads (pR + d*l/2),  q1[t] = prod_reads (pR + d*t/4), t = l+m, d = pA - pR ----
  auto phase1 = [&](int b) {
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
    // (pR = pA = 1).  The eight LUT reads are issued together.  Measured: the branchy read loop made phase 1 two
    // fifths of the kernel time at a seventh of its instructions -- per-read LDS round trips on a divergent path.
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
    double* dst = pgs + slot * Q_SLOT_STRIDE + r * 8;
    dst[0] = q0[0];
    dst[1] = q0[1];
    dst[2] = q0[2];
#pragma unroll
    for (int t = 0; t < 5; ++t) dst[3 + t] = q1[t];
    pgs[slot * Q_SLOT_STRIDE + 32 + r] = hs_out;
    snps[slot * 4 + r] = (s >= 0) ? s : S_dummy;
    const int32_t sn = snp_of(precA, b + 1);
    snps_nx[slot * 4 + r] = (sn >= 0) ? sn : S_dummy;  // next batch; its no-genotype rows are (1,0,0)
  };

  // ---- phase 2 for entry i of the slot's batch: lane <-> 4 samples ----
  auto sweep_entry = [&](const row_t& R, int i) {
    const double* qq = pgs + slot * Q_SLOT_STRIDE + i * 8;
    const double a0 = qq[0], a1 = qq[1], a2 = qq[2];
    const double b0 = qq[3], b1 = qq[4], b2 = qq[5], b3 = qq[6], b4 = qq[7];
    const double hs = pgs[slot * Q_SLOT_STRIDE + 32 + i];
    double u[4][3];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // singlet slot llksAB[j][0][0] (:806,828), alpha = 0
      const double u0 = fma(R.G[c][2], a2, fma(R.G[c][1], a1, R.G[c][0] * a0));
      acc[q_acc_single(c)] *= u0 * hs;
      // u[c][m] = sum_l g_j[l] * pG[alpha=.5][l][m],  pG[l][m] = b[l+m]
      u[c][0] = fma(R.G[c][2], b2, fma(R.G[c][1], b1, R.G[c][0] * b0));
      u[c][1] = fma(R.G[c][2], b3, fma(R.G[c][1], b2, R.G[c][0] * b1));
      u[c][2] = fma(R.G[c][2], b4, fma(R.G[c][1], b3, R.G[c][0] * b2));
    }
#pragma unroll
    for (int c1 = 0; c1 < 4; ++c1)  // pairs inside the lane
#pragma unroll
      for (int c2 = c1 + 1; c2 < 4; ++c2)
        acc[q_acc_within(c1, c2)] *= fma(R.G[c2][2], u[c1][2], fma(R.G[c2][1], u[c1][1], R.G[c2][0] * u[c1][0]));
    {  // neighbouring tile
      double P[4][3];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        P[d][0] = dpp_ror4(R.G[d][0]);
        P[d][1] = dpp_ror4(R.G[d][1]);
        P[d][2] = dpp_ror4(R.G[d][2]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d)
          acc[q_acc_t1(c, d)] *= fma(P[d][2], u[c][2], fma(P[d][1], u[c][1], P[d][0] * u[c][0]));  // :738-746
    }
    {  // opposite tile: the two lanes facing each other split the 16 pairs (c <= d here, d < c over there)
      double Q[4][3];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        Q[d][0] = dpp_ror8(R.G[d][0]);
        Q[d][1] = dpp_ror8(R.G[d][1]);
        Q[d][2] = dpp_ror8(R.G[d][2]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = c; d < 4; ++d)
          acc[q_acc_t2(c, d)] *= fma(Q[d][2], u[c][2], fma(Q[d][1], u[c][1], Q[d][0] * u[c][0]));
    }
  };
  auto pin = [&]() {  // all accumulator updates so far are in program order before what follows
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                 "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]));
    asm volatile("" : "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]), "+v"(acc[16]), "+v"(acc[17]),
                 "+v"(acc[18]), "+v"(acc[19]), "+v"(acc[20]), "+v"(acc[21]), "+v"(acc[22]), "+v"(acc[23]));
    asm volatile("" : "+v"(acc[24]), "+v"(acc[25]), "+v"(acc[26]), "+v"(acc[27]), "+v"(acc[28]), "+v"(acc[29]),
                 "+v"(acc[30]), "+v"(acc[31]), "+v"(acc[32]), "+v"(acc[33]), "+v"(acc[34]), "+v"(acc[35])
                 :
                 : "memory");
  };
  auto renorm = [&]() {
#pragma unroll
    for (int a = 0; a < QN_ACC; ++a) {
      int e;
      acc[a] = frexp(acc[a], &e);
      exs[a][lane] = (int16_t)(exs[a][lane] + e);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) prodacc_renorm(accW[c], exW[c]);
  };

  // One batch: phase 1, then its four entries.  X holds the row of entry 0, Y of entry 1 (both requested earlier); Z is
  // free.  Entry i is swept while the row of stream position i+2 is requested into the set that was swept last.
  auto batch = [&](int b, row_t& X, row_t& Y, row_t& Z) {
    phase1(b);
    __syncthreads();
    load_row(Z, snps[slot * 4 + 2]);
    asm volatile("" ::: "memory");
    sweep_entry(X, 0);
    pin();
    load_row(X, snps[slot * 4 + 3]);
    asm volatile("" ::: "memory");
    sweep_entry(Y, 1);
    pin();
    load_row(Y, snps_nx[slot * 4 + 0]);
    asm volatile("" ::: "memory");
    sweep_entry(Z, 2);
    pin();
    load_row(Z, snps_nx[slot * 4 + 1]);
    asm volatile("" ::: "memory");
    sweep_entry(X, 3);
    pin();
    if ((b & 3) == 3) renorm();  // 16 entries per slot since the last renormalisation
    __syncthreads();
  };

  // ---- the linear entries (at most one usable read).  The single factor pR + (pA - pR) p, p = l/2 (alpha 0) or
  //      (l+m)/4 (alpha 0.5), stays linear through the tail (:703-725): q0[l] = A + 2B l, q1[l+m] = A + B (l+m).  With the
  //      moments s = g0 + g1 + g2 and rho = (g1 + 2 g2) / s of a triple (gmq),
  //          singlet  sum_l g_j[l] q0[l] * s_0      = s_j s_0 (A + 2B rho_j)          (s_0: sample 0's sum, :806),
  //          pair     sum_lm g_j[l] g_k[m] q1[l+m]  = s_j s_k (A + B rho_j + B rho_k):
  //      the sums s go into products of their own (accW, folded into the accumulators at the end) and a hypothesis costs
  //      an FMA and the product update instead of a three-term dot product and the update; a row is 8 doubles instead
  //      of 12, and one double per sample rotates instead of three.
  //      There is no phase 1 here: (A, B) depend on one read byte, so the four lanes of a slot read the entry's 8-byte
  //      record {snp, code} themselves and look (A, B) up in LDS.  A pipeline over rings of three register sets
  //      (unrolled three times, so that ring positions are names): record three entries ahead, row two ahead, (A, B)
  //      one ahead; loads are unconditional and a slot behind the end of its list sweeps neutral entries (code 0xFF,
  //      the dummy row: every factor exactly 1).
  if (nLmax > 0) {
    struct rowl_t {
      double m[4][2];
    };
    auto load_rowl = [&](rowl_t& R, int32_t sidx) {
      if (QUAD_EXP & 4) sidx &= 255;
      if (QUAD_EXP & 64) sidx &= 15;
      if (QUAD_EXP & 128) sidx = 0;
      if (QUAD_EXP & 256) {  // (experiment) half the row loads
        const double2* pc = reinterpret_cast<const double2*>(gmq + (size_t)sidx * 32) + r;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const double2 v = pc[c * 4];
          R.m[c][0] = R.m[c + 2][0] = v.x;
          R.m[c][1] = R.m[c + 2][1] = v.y;
        }
        return;
      }
      if (QUAD_EXP & 4096) {  // (experiment) every 16-lane group reads one whole row (two full lines) per instruction
        const int ln = lane & 15;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int sk = c == 0 ? __builtin_amdgcn_mov_dpp(sidx, 0x00, 0xF, 0xF, false)
                       : c == 1 ? __builtin_amdgcn_mov_dpp(sidx, 0x55, 0xF, 0xF, false)
                       : c == 2 ? __builtin_amdgcn_mov_dpp(sidx, 0xAA, 0xF, 0xF, false)
                                : __builtin_amdgcn_mov_dpp(sidx, 0xFF, 0xF, 0xF, false);
          const double2 v = (reinterpret_cast<const double2*>(gmq + (size_t)sk * 32))[ln];
          R.m[c][0] = v.x;
          R.m[c][1] = v.y;
        }
        return;
      }
      if (QUAD_EXP & 1024) {  // (experiment) the sixteen slots of an instruction spread over the four pieces of their rows
        const double2* pc = reinterpret_cast<const double2*>(gmq + (size_t)sidx * 32) + r;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double2 v = pc[((c + slot) & 3) * 4];
          R.m[c][0] = v.x;
          R.m[c][1] = v.y;
        }
        return;
      }
      if (QUAD_EXP & 2048) {  // (experiment) same, over sixteen 16-byte pieces
        const double2* pc = reinterpret_cast<const double2*>(gmq + (size_t)sidx * 32);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double2 v = pc[(c * 4 + r + slot) & 15];
          R.m[c][0] = v.x;
          R.m[c][1] = v.y;
        }
        return;
      }
      if (QUAD_EXP & 512) {  // (experiment) the lane's eight doubles as 64 contiguous bytes
        const double2* pc = reinterpret_cast<const double2*>(gmq + (size_t)sidx * 32) + r * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double2 v = pc[c];
          R.m[c][0] = v.x;
          R.m[c][1] = v.y;
        }
        return;
      }
      if (QUAD_EXP & 32) {  // (experiment) no row loads at all
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          R.m[c][0] = 1.0;
          R.m[c][1] = 1e-9 * (double)(sidx & 3);
        }
        return;
      }
      const double2* pc = reinterpret_cast<const double2*>(gmq + (size_t)sidx * 32) + r;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double2 v = pc[c * 4];
        R.m[c][0] = v.x;
        R.m[c][1] = v.y;
      }
    };
    const int2* lr = reinterpret_cast<const int2*>(qlrec + e0);
    auto settle = [&](int2& rc, int i) {  // a record behind the end of the slot's list: neutral
      const bool in = i < nl;
      rc.x = in ? rc.x : S_dummy;
      rc.y = in ? rc.y : (int)MUXGL_READ_OTHER;
    };
    auto ab_of = [&](const int2& rc) { return *reinterpret_cast<const double2*>(ablut + 2 * rc.y); };
    auto sweepL = [&](const rowl_t& R, const double2& ab) {
      const double A = ab.x, B = ab.y, B2 = B + B;
      double X[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        accW[c] *= R.m[c][0];
        acc[q_acc_single(c)] *= fma(B2, R.m[c][1], A);  // singlet slot llksAB[j][0][0] (:806,828), alpha = 0
        X[c] = fma(B, R.m[c][1], A);
      }
#pragma unroll
      for (int c1 = 0; c1 < 4; ++c1)  // pairs inside the lane
#pragma unroll
        for (int c2 = c1 + 1; c2 < 4; ++c2) acc[q_acc_within(c1, c2)] *= fma(B, R.m[c2][1], X[c1]);
      {  // neighbouring tile
        double P[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) P[d] = dpp_ror4(R.m[d][1]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int d = 0; d < 4; ++d) acc[q_acc_t1(c, d)] *= fma(B, P[d], X[c]);
      }
      {  // opposite tile: the two lanes facing each other split the 16 pairs (c <= d here, d < c over there)
        double Q[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) Q[d] = dpp_ror8(R.m[d][1]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int d = c; d < 4; ++d) acc[q_acc_t2(c, d)] *= fma(B, Q[d], X[c]);
      }
    };
    // entry i: its row in Rc and (A, B) in abc; rc0 held its record (free now), rc1 / rc2 hold those of i + 1 / i + 2
    // (tch: the line QL_TOUCH records ahead, requested so that its first touch -- an HBM miss, which a wave would meet
    //  in one of its sixteen record streams at almost every entry -- has two sweeps to land: loads return in order)
    auto step = [&](int i, const rowl_t& Rc, rowl_t& Rnn, const double2& abc, double2& abn, int2& rc0, const int2& rc1,
                    int2& rc2, int& tch) {
      if (QL_TOUCH) {
        asm volatile("" ::"v"(tch));
        tch = lr[i + 3 + QL_TOUCH].x;
      }
      rc0 = lr[i + 3];
      settle(rc2, i + 2);
      load_rowl(Rnn, rc2.x);
      abn = ab_of(rc1);
      __builtin_amdgcn_sched_barrier(0);  // the loads are issued in front of the sweep they hide behind
      sweepL(Rc, abc);
      __builtin_amdgcn_sched_barrier(0);
    };
    rowl_t L0, L1, L2;
    double2 ab0, ab1, ab2;
    int2 ra = lr[0], rb = lr[1], rc = lr[2];
    settle(ra, 0);
    settle(rb, 1);
    load_rowl(L0, ra.x);
    load_rowl(L1, rb.x);
    __syncthreads();  // ablut is complete
    ab0 = ab_of(ra);
    int since = 0, t0 = 0, t1 = 0, t2 = 0;
    const long long tl0 = (QUAD_EXP & 16) ? clock64() : 0;
    for (int i = 0; i < nLmax; i += 3) {  // (up to two neutral entries behind the longest list of the wave)
      step(i, L0, L2, ab0, ab1, ra, rb, rc, t0);
      step(i + 1, L1, L0, ab1, ab2, rb, rc, ra, t1);
      step(i + 2, L2, L1, ab2, ab0, rc, ra, rb, t2);
      if (++since == 5 && !(QUAD_EXP & 8)) {  // 15 entries per slot since the last renormalisation
        since = 0;
        renorm();
      }
    }
    renorm();
    if ((QUAD_EXP & 16) && (blockIdx.x % 331) == 0 && lane == 0)
      printf("wave %d: linear loop %d entries, %lld cycles per entry, wall start %lld us\n", (int)blockIdx.x, nLmax,
             (clock64() - tl0) / (nLmax > 0 ? nLmax : 1), wall_clock64() / 100);
  }

  row_t R0, R1, R2;
  snps_nx[slot * 4 + r] = (snp_of(precA, 0) >= 0) ? precA.snp : S_dummy;
  __syncthreads();
  load_row(R0, snps_nx[slot * 4 + 0]);
  load_row(R1, snps_nx[slot * 4 + 1]);
  __syncthreads();
  for (int b = 0; b < nb; b += 3) {
    batch(b, R0, R1, R2);  // leaves entry 0 of the next batch in R1, entry 1 in R2
    if (b + 1 >= nb) break;
    batch(b + 1, R1, R2, R0);
    if (b + 2 >= nb) break;
    batch(b + 2, R2, R0, R1);
  }

  // the sums of the linear entries: every hypothesis gets the products of its two samples (the partner's through the
  // same rotation that brought its values)
  int32_t exa[QN_ACC];
#pragma unroll
  for (int a = 0; a < QN_ACC; ++a) exa[a] = 0;
  if (nLmax > 0) {
    double Wn[4], Wo[4];
    int32_t en[4], eo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      prodacc_renorm(accW[c], exW[c]);
      Wn[c] = dpp_ror4(accW[c]);
      Wo[c] = dpp_ror8(accW[c]);
      en[c] = __builtin_amdgcn_mov_dpp(exW[c], 0x124, 0xF, 0xF, false);
      eo[c] = __builtin_amdgcn_mov_dpp(exW[c], 0x128, 0xF, 0xF, false);
    }
    // singlets: the lane's own sum and sample 0's (tile 0, c = 0 of the same slot), which every singlet carries (:806)
    const double W0 = __shfl(accW[0], lane & ~12, 64);
    const int32_t e0w = __shfl(exW[0], lane & ~12, 64);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[q_acc_single(c)] *= accW[c] * W0;
      exa[q_acc_single(c)] = exW[c] + e0w;
    }
#pragma unroll
    for (int c1 = 0; c1 < 4; ++c1)
#pragma unroll
      for (int c2 = c1 + 1; c2 < 4; ++c2) {
        acc[q_acc_within(c1, c2)] *= accW[c1] * accW[c2];
        exa[q_acc_within(c1, c2)] = exW[c1] + exW[c2];
      }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        acc[q_acc_t1(c, d)] *= accW[c] * Wn[d];
        exa[q_acc_t1(c, d)] = exW[c] + en[d];
      }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int d = c; d < 4; ++d) {
        acc[q_acc_t2(c, d)] *= accW[c] * Wo[d];
        exa[q_acc_t2(c, d)] = exW[c] + eo[d];
      }
  }
  if (q < n_chunks) {
#pragma unroll
    for (int a = 0; a < QN_ACC; ++a) {
      int e;
      acc[a] = frexp(acc[a], &e);
      part_m[((size_t)q * QN_ACC + a) * 4 + r] = acc[a];
      part_e[((size_t)q * QN_ACC + a) * 4 + r] = (int32_t)exs[a][lane] + e + exa[a];
    }
  }
}

// Decodes accumulator idx = a*4 + r of the quad kernel into its hypothesis (j, k) and multiplies the chunk partials of
// one cell in chunk order: ONE log per hypothesis.  Returns false for slots nobody reads (mirrors held twice, j/k >= V).
__device__ __forceinline__ bool quad_hypothesis(int idx, int64_t c0, int64_t c1, const int32_t* __restrict__ cell_chunks,
                                                const double* __restrict__ part_m, const int32_t* __restrict__ part_e,
                                                const int32_t* __restrict__ tmap, int V, int& j, int& k, double& v) {
  const int a = idx >> 2, r = idx & 3;
  bool publish = true;
  if (a < 4) {
    j = 4 * r + a;
    k = 0;
  } else if (a < 10) {
    int w = a - 4, c1i, c2i;
    if (w < 3) {
      c1i = 0;
      c2i = w + 1;
    } else if (w < 5) {
      c1i = 1;
      c2i = w - 1;
    } else {
      c1i = 2;
      c2i = 3;
    }
    j = 4 * r + c2i;
    k = 4 * r + c1i;
  } else if (a < 26) {
    const int cc = (a - 10) >> 2, dd = (a - 10) & 3;
    j = 4 * r + cc;
    k = 4 * tmap[r] + dd;
  } else {
    int w = a - 26, cc, dd;
    if (w < 4) {
      cc = 0;
      dd = w;
    } else if (w < 7) {
      cc = 1;
      dd = w - 3;
    } else if (w < 9) {
      cc = 2;
      dd = w - 5;
    } else {
      cc = 3;
      dd = 3;
    }
    const int ro = tmap[4 + r];
    j = 4 * r + cc;
    k = 4 * ro + dd;
    if (cc == dd && r > ro) publish = false;  // the facing lane holds the same pair
  }
  if (!publish || j >= V || k >= V) return false;
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
      const size_t o = (size_t)cell_chunks[ok ? ci + u : c0] * QN_ACC * 4 + idx;
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
  v = log(m) + (double)e * 0.6931471805599453094;
  return true;
}

// writes ll[c][j][k][n] (+ mirror) of one cell to the LL tensor in HBM (needed when the caller asks for the tensor)
__global__ void __launch_bounds__(192)
    demux_quad_reduce_kernel(const int64_t* __restrict__ cell_chunk_ptr, const int32_t* __restrict__ cell_chunks,
                             const double* __restrict__ part_m, const int32_t* __restrict__ part_e,
                             const int32_t* __restrict__ tmap, int V, double* __restrict__ ll) {
  const int64_t c = blockIdx.x;
  const int64_t c0 = cell_chunk_ptr[c], c1 = cell_chunk_ptr[c + 1];
  const int idx = threadIdx.x;
  if (c0 == c1 || idx >= QN_ACC * 4) return;
  int j, k;
  double v;
  if (!quad_hypothesis(idx, c0, c1, cell_chunks, part_m, part_e, tmap, V, j, k, v)) return;
  double* out = ll + (size_t)c * V * V * 2;
  if (idx < 16) {
    out[((size_t)j * V + k) * 2 + 0] = v;  // singlet: alpha index 0
  } else {
    out[((size_t)j * V + k) * 2 + 1] = v;
    out[((size_t)k * V + j) * 2 + 1] = v;
  }
}

// The same reduction, but the hypotheses of four cells stay in LDS and the call (demux_call_body.hpp) follows at once,
// sixteen lanes per cell in wave 0: no LL tensor round trip through HBM, one launch less, and the 160-byte records go
// straight to the caller's pinned host buffer (16-byte stores of consecutive lanes), which removes the separate
// device-to-host copy.
constexpr int QF_CELLS = 4;
__global__ void __launch_bounds__(256)
    demux_quad_finish_kernel(int64_t C, const int64_t* __restrict__ cell_ptr, const int64_t* __restrict__ cell_chunk_ptr,
                             const int32_t* __restrict__ cell_chunks, const double* __restrict__ part_m,
                             const int32_t* __restrict__ part_e, const int32_t* __restrict__ tmap, int V,
                             muxgl_call::call_alpha al, double doublet_prior, muxgl_demux_cell* __restrict__ out) {
  __shared__ double llt[QF_CELLS][16 * 16 * 2];
  __shared__ __align__(16) muxgl_demux_cell rec[QF_CELLS];
  static_assert(sizeof(muxgl_demux_cell) % 16 == 0, "records are copied out in 16-byte pieces");
  const int64_t cbase = (int64_t)blockIdx.x * QF_CELLS;
  const int tid = threadIdx.x;
  for (int t = tid; t < QF_CELLS * 512; t += 256) (&llt[0][0])[t] = 0.0;
  __syncthreads();
  for (int w = tid; w < QF_CELLS * QN_ACC * 4; w += 256) {
    const int lc = w / (QN_ACC * 4), idx = w - lc * (QN_ACC * 4);
    const int64_t c = cbase + lc;
    if (c >= C) break;
    const int64_t c0 = cell_chunk_ptr[c], c1 = cell_chunk_ptr[c + 1];
    int j, k;
    double v;
    if (c0 != c1 && quad_hypothesis(idx, c0, c1, cell_chunks, part_m, part_e, tmap, V, j, k, v)) {
      if (idx < 16) {
        llt[lc][(j * V + k) * 2 + 0] = v;
      } else {
        llt[lc][(j * V + k) * 2 + 1] = v;
        llt[lc][(k * V + j) * 2 + 1] = v;
      }
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int lc = tid >> 4;
    const int64_t c = cbase + lc;
    const bool ok = c < C;
    muxgl_call::demux_call_group<16>(tid, ok, ok ? (int32_t)(cell_ptr[c + 1] - cell_ptr[c]) : 0, V, 2, al.a,
                                     doublet_prior, llt[lc], &rec[lc]);
  }
  __syncthreads();
  constexpr int NQ = (int)(sizeof(muxgl_demux_cell) / 16);
  if (tid < QF_CELLS * NQ && cbase + tid / NQ < C)
    reinterpret_cast<uint4*>(out + cbase)[tid] = reinterpret_cast<const uint4*>(&rec[0])[tid];
}

}  // namespace

// the launch order of the chunks of a quad kernel (quad_order_key_kernel), shared with fmx_quad.hip
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
  if (e == hipSuccess) e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_key, d_key2, d_iota, *order, (size_t)n, 0u, 64u, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  cleanup();
  if (e != hipSuccess) MUXGL_FAIL(h, "quad_launch_order: %s", hipGetErrorString(e));
  return 0;
}

// returns -1 when the quad path does not apply, 0 ok, 1 error
int demux_quad_launch(muxgl_handle* h, const muxgl_demux_params* p) {
  if (h->V > 16 || !h->qrow || !h->d_gpq || !h->d_qent || h->C == 0) return -1;
  if (h->flags & (MUXGL_FLAG_FORCE_TILE_SWEEP | MUXGL_FLAG_FORCE_ROW_KERNEL | MUXGL_FLAG_FORCE_WAVE_KERNEL)) return -1;
  if (p->n_alpha != 2 || p->alpha[0] != 0.0 || p->alpha[1] != 0.5) return -1;
  muxgl_row_state* st = h->qrow;
  if (!st->d_tmap) {  // tile map of the two rotations
    if (dev_alloc(h, &st->d_tmap, 8)) return 1;
    hipLaunchKernelGGL(quad_tmap_kernel, dim3(1), dim3(64), 0, h->stream, st->d_tmap);
    HIPCHK(h, hipGetLastError());
  }
  const size_t need = (size_t)st->n_chunks * QN_ACC * 4;
  if (need > st->part_cap) {
    if (dev_alloc(h, &st->d_part, need)) return 1;
    st->part_cap = need;
  }
  if (need > st->part_e_cap) {
    if (dev_alloc(h, &st->d_part_e, need)) return 1;
    st->part_e_cap = need;
  }
  const bool use_lin = h->d_lin && h->d_gmq && !(h->flags & MUXGL_FLAG_NO_LINEAR_ENTRIES);
  if (use_lin && !st->d_chunk_nlin && st->n_chunks) {  // once per pileup and GP tensor: every chunk's linear entries first
    // (the sweep reads up to QL_SLACK records behind a chunk's list, unconditionally: slack behind the array)
    if (dev_alloc(h, &st->d_qent_lin, (size_t)h->nnz) || dev_alloc(h, &st->d_chunk_nlin, (size_t)st->n_chunks) ||
        dev_alloc(h, &st->d_qlrec, (size_t)h->nnz + QL_SLACK))
      return 1;
    HIPCHK(h, hipMemsetAsync(st->d_qlrec, 0xFF, sizeof(quad_lrec) * ((size_t)h->nnz + QL_SLACK), h->stream));
    hipLaunchKernelGGL(quad_partition_kernel, dim3((unsigned)((st->n_chunks + 63) / 64)), dim3(64), 0, h->stream,
                       (int)st->n_chunks, st->d_chunks, h->d_qent, h->d_lin, h->d_reads, h->d_gp0s, st->d_qent_lin,
                       st->d_qlrec, st->d_chunk_nlin);
    HIPCHK(h, hipGetLastError());
    if (quad_launch_order(h, st->d_chunks, st->d_chunk_nlin, st->n_chunks, &st->d_quad_order)) return 1;
  }
  tic(h, MUXGL_T_DEMUX_SWEEP);
  const unsigned blocks = (unsigned)((((st->n_chunks + 15) / 16) + 7) / 8 * 8);  // multiple of 8 for xcd_swizzle
  if (blocks) {
    hipLaunchKernelGGL(demux_quad_kernel, dim3(blocks), dim3(64), 0, h->stream, st->d_chunks, (int)st->n_chunks,
                       use_lin ? st->d_qent_lin : h->d_qent, st->d_qlrec,
                       use_lin ? st->d_chunk_nlin : (const int32_t*)nullptr,
                       use_lin ? st->d_quad_order : (const int32_t*)nullptr, h->d_reads,
                       h->d_gpq, h->d_gmq, h->d_gp0s, (int32_t)h->S, h->d_lut, st->d_part, st->d_part_e);
    HIPCHK(h, hipGetLastError());
  }
  toc(h, MUXGL_T_DEMUX_SWEEP);
  tic(h, MUXGL_T_DEMUX_REDUCE);
  if (h->want_full_ll) {
    hipLaunchKernelGGL(demux_quad_reduce_kernel, dim3((unsigned)h->C), dim3(192), 0, h->stream, st->d_cell_chunk_ptr,
                       st->d_cell_chunks, st->d_part, st->d_part_e, st->d_tmap, h->V, h->d_ll);
  } else {  // reduce + call fused, records written to the pinned host buffer
    muxgl_call::call_alpha al;
    for (int i = 0; i < MUXGL_MAX_ALPHA; ++i) al.a[i] = (i < p->n_alpha) ? p->alpha[i] : 0.0;
    hipLaunchKernelGGL(demux_quad_finish_kernel, dim3((unsigned)((h->C + QF_CELLS - 1) / QF_CELLS)), dim3(256), 0,
                       h->stream, h->C, h->d_cell_ptr, st->d_cell_chunk_ptr, st->d_cell_chunks, st->d_part, st->d_part_e, st->d_tmap, h->V, al,
                       p->doublet_prior, h->h_dcells);
    h->records_on_host = true;
  }
  HIPCHK(h, hipGetLastError());
  toc(h, MUXGL_T_DEMUX_REDUCE);
  return 0;
}
