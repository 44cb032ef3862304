// demux_ring.hip -- the linear entries of the demuxlet pair sweep for more than 32 samples (demux_wave.hip's diagonal
// blocks: samples 64 X .. 64 X + 63 against themselves).
//
// Reference being replaced: cmd_cram_demuxlet.cpp:733-747 for the entries with at most one usable read, whose
// likelihoods are pG[l][m] = A + Bl*l + Bm*m (cmd_cram_demuxlet.cpp:673,685 for a single factor; see the notes on
// EM_LINEAR in demux_wave.hip).  A pair hypothesis of such an entry is s_j s_k (A + Bl rho_j + Bm rho_k): one FMA and the
// product update per (pair, alpha).  At 64 samples and the north_star's six alphas that is 18 208 hypotheses per entry:
// the sweep is bound by FP64 issue, so every vector instruction that is not one of those two counts.  The kernels this
// file replaced spent 55 of 183 vector instructions per (wave, entry) around the sweep: every wave fetched the marker's
// row and the entry's (A, Bl, Bm) table row for itself through a register pipeline two entries deep, wrote a ring of its
// own, and the symmetric alpha walked all entries again in a launch of its own.  Here:
//   * workgroup = work unit (a cell, or a part of a long one: wave_item), four waves; wave w owns sixteen rotation
//     steps of up to four non-symmetric alphas -- partners at distances 8 w + 1 .. 8 w + 8 and 32 + 8 w + 1 .. 32 + 8 w + 8 --
//     AND the distances 8 w + 1 .. 8 w + 8 of the symmetric one (alpha = 0.5 meets every unordered pair within 32 steps),
//     whose partners are those of its first eight steps (round 5: they used to be read a second time): 72 product
//     accumulators per lane, all five pair alphas of the north_star grid in ONE walk;
//   * the marker rows (s, rho per sample: wave_gm_kernel) are STAGED: the entries of a unit are taken in batches of
//     eight; each wave fetches two rows of the next batch (one 16-byte load per lane and row) while the current batch is
//     swept and leaves them in LDS -- rho twice over, so that the staged row IS the ring all four waves read their
//     partners from at immediate offsets (demux_wave.hip: the ring), the sums s, and the factor of the singlet slot
//     (j, 0, n = 0), which the loader forms once.  One fetch per row and workgroup with a whole batch (~5 k cycles) of
//     cover, and per wave and entry nothing but two LDS reads and the lane's A + Bl rho_j per alpha;
//   * (A, Bl, Bm) come from a table of 130 rows indexed by (allele, base quality) of the entry's one read (row 128: no
//     usable read; row 129: a marker without genotypes, or padding -- factors of exactly 1), read through the scalar
//     cache: 8 bytes of record {snp, row} per entry instead of 24 bytes per entry and alpha;
//   * products as mantissa * 2^exponent, renormalised every 24 entries (each factor is >= 1e-10 / (1 + 1e-10), so 24 of
//     them cannot underflow); half of the exponents live in registers, half in LDS; the product of the lane's own sums
//     (wave 1) and the singlet slot (wave 0) are kept by one wave each.
// Results go to the wave layout llw[cell][block][alpha][step][lane] of demux_wave.hip, whose EM_GENERAL launches add the
// other entries on top.
#include "common.hpp"
#include "demux_entry.hpp"

namespace {

constexpr int RL_B = 8;       // entries per staged batch
constexpr int RL_ROW = 256;   // doubles per staged entry: rho of the 64 lanes twice over (the ring), s, singlet factors
constexpr int RL_LUTW = 24;   // doubles per table row: (A, Bl, Bm) of six slots, padded to 192 bytes
constexpr int RL_BITS = 18;   // ... and, as an int32 in the double behind them, an upper bound of the bits a factor of the row costs
constexpr int RL_NLUT = 130;  // rows: allele << 6 | quality; 128 = no usable read; 129 = neutral
constexpr int RL_PAD = 32;    // records readable behind the end of the stream
constexpr int RL_RENORM = 3;  // batches between renormalisations of the singlet slot / the product of the lane's sums
constexpr int RL_BUDGET = 600;  // bits the products may have lost before they are renormalised (a batch costs <= 8 * 35 more)

struct ring_alpha {
  double a[6];  // slot 0: alpha[0] (singlets), 1..4: the launch's non-symmetric alphas, 5: the symmetric one
};

// table row r, slot s: the likelihoods of a one-read entry as the per-entry kernel computes them (row_entry_pg), read off
// as A = pG[0][0], Bl = pG[1][0] - pG[0][0], Bm = pG[0][1] - pG[0][0]
__global__ void __launch_bounds__(64) ring_lut_kernel(ring_alpha al, const double* __restrict__ lut, double* __restrict__ out) {
  const int r = blockIdx.x, s = threadIdx.x;
  __shared__ double lo[6];
  double* o = out + (size_t)r * RL_LUTW + (s < 6 ? s : 0) * 3;
  if (s < 6) {
    if (r == RL_NLUT - 1) {
      o[0] = 1.0, o[1] = 0.0, o[2] = 0.0;
    } else {
      double pG[9];
      const uint32_t byte = (uint32_t)(((r >> 6) & 1) << 7) | (uint32_t)(r & 63);
      row_entry_pg<1>(nullptr, 0, r < 128 ? 1 : 0, byte, &al.a[s], lut, pG);
      o[0] = pG[0];
      o[1] = pG[3] - pG[0];
      o[2] = pG[1] - pG[0];
    }
    // the smallest factor A + Bl rho_j + Bm rho_k an accumulator can meet at this row (rho in [0, 2]: a corner)
    lo[s] = fmin(fmin(o[0], fma(2.0, o[1], o[0])), fmin(fma(2.0, o[2], o[0]), fma(2.0, o[1] + o[2], o[0])));
  }
  __syncthreads();
  if (s == 0) {  // RL_BITS: how many bits a product can lose to a factor of this row, at most (0 for the neutral row)
    double m = lo[0];
    for (int i = 1; i < 6; ++i) m = fmin(m, lo[i]);
    int e = 0;
    if (m < 1.0) {
      (void)frexp(m > 1e-300 ? m : 1e-300, &e);  // m >= 2^(e-1)
      e = 1 - e;
    }
    reinterpret_cast<int32_t*>(out + (size_t)r * RL_LUTW + RL_BITS)[0] = e > 0 ? e + 1 : 0;
  }
}

// the linear entries' records in stream order: {snp, byte offset of the table row}
__global__ void __launch_bounds__(256)
    ring_rec_kernel(int64_t n_lin, const fmx_grec* __restrict__ rec_lin, const int64_t* __restrict__ entry_rptr,
                    const uint8_t* __restrict__ reads, const uint8_t* __restrict__ has_gp, uint2* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_lin + RL_PAD) return;
  uint32_t snp = 0, row = RL_NLUT - 1;
  if (r < n_lin) {
    const fmx_grec x = rec_lin[r];
    snp = (uint32_t)x.snp;
    if (has_gp[x.snp]) {
      row = 128;
      for (int64_t q = entry_rptr[x.e]; q < entry_rptr[x.e + 1]; ++q) {
        const uint8_t b = reads[q];
        if (b == MUXGL_READ_OTHER) continue;
        row = ((uint32_t)(b >> 7) << 6) | (uint32_t)(b & 0x3f);  // (the class holds qualities <= 60 only: lin_kernel)
        break;
      }
    }
  }
  out[r] = uint2{snp, row * (uint32_t)(RL_LUTW * 8)};
}

// GEN: the same workgroup then walks the unit's OTHER entries (the clear bits' record stream, pG rows from the table of
// demux_entry_pg_kernel) with the same accumulators -- three FMAs and the product update per hypothesis
// (cmd_cram_demuxlet.cpp:738-746) -- so that a unit's hypotheses are written ONCE, as one log per hypothesis of all its
// entries.  (Round 3 walked them in launches of their own, demux_wave.hip's EM_GENERAL, which read the slab written
// here and wrote it again, per alpha set.)  The general walk is staged like the linear one, in batches of two entries:
// of the four waves, two prepare each entry of the NEXT batch -- wave 2 e the ring of triples, the singlet factor and
// the lane's u[m] = sum_l g_j[l] pG[n][l][m] for alpha[0] and the launch's first two alphas, wave 2 e + 1 the u of the
// other two and of the symmetric one -- and leave them in LDS; the sweeping waves read their u (15 numbers per lane,
// formed once per workgroup instead of once per wave) and the partners' triples from there and carry no scalar
// operands at all.
// (tells the compiler that the values, requested by opaque ds_read statements, are defined from here on)
__device__ __forceinline__ void ring_landed(dbl2& a, double& b) { asm volatile("" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void ring_landed(dbl2& a) { asm volatile("" : "+v"(a)); }

__device__ __forceinline__ void ring_sgpr_landed(double& a, double& b, double& c) { asm volatile("" : "+s"(a), "+s"(b), "+s"(c)); }

template <int NA, bool SYM, bool GEN>
__global__ void __launch_bounds__(256, 2)
    demux_ring_lin_kernel(const wave_item* __restrict__ items, int64_t n_items, const uint32_t* __restrict__ lin,
                          const int64_t* __restrict__ lin_rank, const uint2* __restrict__ rrec,
                          const double* __restrict__ lutg, const double* __restrict__ gm, int V, int nAlpha, ring_sel sel,
                          const fmx_grec* __restrict__ gen_rec, const double* __restrict__ gp,
                          const double* __restrict__ pgt, int pg_by_record, double* __restrict__ ll) {
  constexpr int NS = NA > 0 ? 16 : 0, NSY = SYM ? 8 : 0, NACC = NA * NS + NSY;
  constexpr int NXL = NACC / 2, NXV = NACC - NXL;  // exponents in LDS / in registers (GEN: 36 KB + the 44 KB stage = 80 KB per workgroup)
  // SHARE (round 5): with both kinds of alphas in the launch, wave w owns the rotation steps 8 w .. 8 w + 7 and
  // 32 + 8 w .. 39 + 8 w (slots t = 0 .. 7 and 8 .. 15) -- so that the partners of its symmetric distances 8 w + 1 .. 8 w + 8
  // are the partners of its first eight steps, and the symmetric alpha needs no ring reads of its own: 16 partner values
  // per wave and entry instead of 24 (what an LDS read costs the sweep is the bytes it returns).
  constexpr bool SHARE = NA > 0 && SYM;
  constexpr int GN = NS / 4, GS = SHARE ? 0 : NSY / 4, GT = GN + GS;  // groups of four ring reads per entry
  constexpr int NA1 = NA > 0 ? NA : 1;
  static_assert(GT % 2 == 0 && GT > 0, "the read buffers alternate per group");
  constexpr int RG_B = 2;        // general entries per staged batch
#ifndef RING_GEN_G
#define RING_GEN_G 1
#endif
  constexpr int RG_G = RING_GEN_G;  // rotation steps per group of ring reads in the general walk
  constexpr int RG_ROW = 1408;   // doubles per staged general entry: ring of (g0, g1) [128][2], ring of g2 [128], own pairs [8][64][2] (gstore)
  constexpr int STAGE_N = (GEN && 2 * RG_B * RG_ROW > 2 * RL_B * RL_ROW) ? 2 * RG_B * RG_ROW : 2 * RL_B * RL_ROW;
  __shared__ double stage_all[STAGE_N];
  double (*stage)[RL_B][RL_ROW] = reinterpret_cast<double (*)[RL_B][RL_ROW]>(stage_all);
  __shared__ int32_t exs_all[4][NXL][64];
  if ((int64_t)blockIdx.x >= n_items) return;
  const wave_item it = items[blockIdx.x];
  if (it.e0 == it.e1) return;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = threadIdx.x & 63;  // (w: known to be wave-uniform)
  const bool live = sel.jbase + j < V;
  int32_t (*exs)[64] = exs_all[w];
  int64_t i0, i1;
  wave_stream_range<EM_LINEAR>(lin, lin_rank, it.e0, it.e1, i0, i1);
  const int64_t n = i1 - i0;
  const int nb = (int)((n + RL_B - 1) / RL_B);
  const uint2* rr = rrec + i0;
  const double* grow = gm + (size_t)(live ? sel.jbase + j : V - 1) * 2;

  double acc[NACC], accX = 1.0;  // accX: wave 0 the singlet slot, wave 1 the product of the lane's own sums s
  int32_t ex[NXV], exX = 0;
#pragma unroll
  for (int t = 0; t < NACC; ++t) {
    acc[t] = 1.0;
    if (t < NXV) ex[t] = 0;
    else exs[t - NXV][j] = 0;
  }

  // LDS byte addresses inside a staged row: own rho at [j]; second own value: the singlet factor [192 + j] (wave 0) or the
  // sum s [128 + j]; the ring reads: below
  const uint32_t base0 = (uint32_t)(uintptr_t)&stage[0][0][0];
  constexpr uint32_t ROWB = RL_ROW * 8, BUFB = RL_B * ROWB;
  const uint32_t ownoff = (uint32_t)j * 8u, own2off = (uint32_t)((w == 0 ? 192 : 128) + j) * 8u;
  // ring reads: slot t < 8 is step 8 w + t, read at [j + 63 - 8 w - t] = rsoff + (7 - t) -- which is also the symmetric
  // distance 8 w + t + 1; slot t >= 8 is step 24 + 8 w + t, read at [j + 39 - 8 w - t] = rboff + (15 - t)
  const uint32_t rboff = (uint32_t)(j + 24 - 8 * w) * 8u, rsoff = (uint32_t)(j + 64 - 8 * w - 8) * 8u;

  // ---- loader: rows of the entries w and w + 4 of a batch
  uint32_t rcs[RL_B], rcl[RL_B], rns[RL_B], rnl[RL_B];  // records of the current / next batch: snp, table row offset
  auto load_recs = [&](uint32_t (&rs_)[RL_B], uint32_t (&rl_)[RL_B], int b) {  // (records behind the unit's end exist -- RL_PAD)
    const uint32_t* p = (const uint32_t*)(rr + (int64_t)b * RL_B);
#pragma unroll
    for (int k = 0; k < RL_B; ++k) {
      const bool in = (int64_t)b * RL_B + k < n;
      rs_[k] = in ? p[2 * k] : 0u;
      rl_[k] = in ? p[2 * k + 1] : (uint32_t)((RL_NLUT - 1) * RL_LUTW * 8);
    }
  };
  double2 g2[2], h2[2];  // (s, rho) of the lane's sample / of sample 0, whose moments every singlet carries (:806,828)
  double q0[2][3];       // (A, Bl, Bm) of alpha[0] for the two entries
  auto load_rows = [&](const uint32_t (&rs_)[RL_B], const uint32_t (&rl_)[RL_B]) {
    const uint32_t sa = w == 0 ? rs_[0] : (w == 1 ? rs_[1] : (w == 2 ? rs_[2] : rs_[3]));
    const uint32_t sb = w == 0 ? rs_[4] : (w == 1 ? rs_[5] : (w == 2 ? rs_[6] : rs_[7]));
    const uint32_t la = w == 0 ? rl_[0] : (w == 1 ? rl_[1] : (w == 2 ? rl_[2] : rl_[3]));
    const uint32_t lb = w == 0 ? rl_[4] : (w == 1 ? rl_[5] : (w == 2 ? rl_[6] : rl_[7]));
    g2[0] = *(const double2*)(grow + (size_t)sa * V * 2);
    g2[1] = *(const double2*)(grow + (size_t)sb * V * 2);
    h2[0] = *(const double2*)(gm + (size_t)sa * V * 2);
    h2[1] = *(const double2*)(gm + (size_t)sb * V * 2);
    const double* La = (const double*)((const char*)lutg + la);
    const double* Lb = (const double*)((const char*)lutg + lb);
#pragma unroll
    for (int i = 0; i < 3; ++i) q0[0][i] = La[i], q0[1][i] = Lb[i];
  };
  auto store_rows = [&](int buf, int b) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = w + 4 * q;
      const bool ok = live && (int64_t)b * RL_B + k < n;
      const double s = ok ? g2[q].x : 1.0, rho = ok ? g2[q].y : 0.0;
      // sum_m h[m] sum_l g[l] (A + Bl l + Bm m) of alpha[0] = s_h s_g (A + Bl rho_g + Bm rho_h)
      const double sv = ok ? (s * h2[q].x) * fma(q0[q][2], h2[q].y, fma(q0[q][1], rho, q0[q][0])) : 1.0;
      double* row = &stage[buf][k][0];
      row[j] = rho;
      row[j + 64] = rho;
      row[128 + j] = s;
      row[192 + j] = sv;
    }
  };

  unsigned pfv = 0, pfx = 0;  // touches of the record stream (see dw_walk in demux_wave.hip)
  load_recs(rcs, rcl, 0);
  load_rows(rcs, rcl);
  store_rows(0, 0);
  __syncthreads();

  double rd[2][4], ow[2][2];             // ring values of a group; own (rho, second value) of an entry
  double lA[2][6], lB[2][6], lM[2][6];   // table row of an entry, by slot (see ring_alpha)
  double u0[NA1], u0s = 0.0;             // the lane's A + Bl rho_j per alpha
  auto load_lut = [&](auto pc, uint32_t off) {
    constexpr int p = decltype(pc)::value;
    const double* L = (const double*)((const char*)lutg + off);
#pragma unroll
    for (int s = 1; s < 6; ++s) {
      const bool used = s <= 4 ? s - 1 < NA : SYM;
      if (used) lA[p][s] = L[s * 3], lB[p][s] = L[s * 3 + 1], lM[p][s] = L[s * 3 + 2];
    }
  };
  // The table row of the NEXT entry is requested while the current entry's first group is swept and must not be waited
  // for there: scalar loads and LDS reads share one counter that scalar loads leave out of order, so the wait the
  // compiler puts in front of the row's first use is lgkmcnt(0) -- placed right behind the request it exposed the scalar
  // cache's latency twice per entry with the next group's ring reads in flight.  lut_landed names the row's first use:
  // at the top of the entry's last group, a whole entry's sweep behind the request, where the only LDS reads in flight
  // are the ones that group waits for anyway.
  auto lut_landed = [&](auto pc) {
    constexpr int p = decltype(pc)::value;
#pragma unroll
    for (int s = 1; s < 6; ++s) {
      const bool used = s <= 4 ? s - 1 < NA : SYM;
      if (used) ring_sgpr_landed(lA[p][s], lB[p][s], lM[p][s]);
    }
  };
  load_lut(std::integral_constant<int, 0>{}, rcl[0]);

  int cnt = 0, bits = 0;
  for (int b = 0; b < nb; ++b) {
    const uint32_t bb = base0 + (uint32_t)(b & 1) * BUFB;
    const uint32_t own = bb + ownoff, own2 = bb + own2off, rb = bb + rboff, rs = bb + rsoff;
    load_recs(rns, rnl, b + 1);
    pfx ^= pfv;
    {
      const int64_t ta = (int64_t)(b + 4) * RL_B;
      pfv = ((const unsigned*)(rr + (ta < n ? ta : 0)))[j & 15];
    }
    load_rows(rns, rnl);

    // ---- the batch: 8 entries x GT groups of four ring reads as one flat pipeline, a group ahead of its use
    auto issue = [&](auto kc, auto gc) {
      constexpr int k = decltype(kc)::value, g = decltype(gc)::value;
      if constexpr (g == 0) {
        ow[k & 1][0] = wave_ring_rd<k * ROWB>(own);
        ow[k & 1][1] = wave_ring_rd<k * ROWB>(own2);
      }
      wave_for<0, 4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (g < GN && 4 * g + i >= 8) rd[g & 1][i] = wave_ring_rd<k * ROWB + (15 - (4 * g + i)) * 8>(rb);
        else if constexpr (g < GN) rd[g & 1][i] = wave_ring_rd<k * ROWB + (7 - (4 * g + i)) * 8>(rs);
        else rd[g & 1][i] = wave_ring_rd<k * ROWB + (7 - (4 * (g - GN) + i)) * 8>(rs);
      });
    };
    issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    wave_for<0, RL_B>([&](auto kc) {
      constexpr int k = decltype(kc)::value, p = k & 1;
      wave_for<0, GT>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr bool last = g + 1 == GT;
        constexpr bool more = !last || k + 1 < RL_B;
#ifndef RING_NO_LUT_LANDING
        if constexpr (last) lut_landed(std::integral_constant<int, 1 - p>{});
#endif
        if constexpr (!last) issue(kc, std::integral_constant<int, g + 1>{});
        else if constexpr (k + 1 < RL_B) issue(std::integral_constant<int, k + 1>{}, std::integral_constant<int, 0>{});
        constexpr int ahead = more ? (last ? 6 : 4) : 0;  // younger reads: they may stay in flight
        if constexpr (g == 0) {
          asm volatile("s_waitcnt lgkmcnt(%6)"
                       : "+v"(rd[0][0]), "+v"(rd[0][1]), "+v"(rd[0][2]), "+v"(rd[0][3]), "+v"(ow[p][0]), "+v"(ow[p][1])
                       : "n"(ahead));
          // the entry's factors of the lane
          const double rho = ow[p][0];
#pragma unroll
          for (int a = 0; a < NA; ++a) u0[a] = fma(lB[p][a + 1], rho, lA[p][a + 1]);
          if (SYM) u0s = fma(lB[p][5], rho, lA[p][5]);
          if (w < 2) accX *= ow[p][1];
          // table row of the next entry
          if constexpr (k + 1 < RL_B) load_lut(std::integral_constant<int, 1 - p>{}, rcl[k + 1]);
          else load_lut(std::integral_constant<int, 1 - p>{}, rnl[0]);
        } else {
          asm volatile("s_waitcnt lgkmcnt(%4)"
                       : "+v"(rd[g & 1][0]), "+v"(rd[g & 1][1]), "+v"(rd[g & 1][2]), "+v"(rd[g & 1][3])
                       : "n"(ahead));
        }
        wave_for<0, 4>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if constexpr (g < GN) {
            constexpr int t = 4 * g + i;
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[a * NS + t] *= fma(lM[p][a + 1], rd[g & 1][i], u0[a]);
            if constexpr (SHARE && t < 8) acc[NA * NS + t] *= fma(lM[p][5], rd[g & 1][i], u0s);
          } else {
            constexpr int t = 4 * (g - GN) + i;
            acc[NA * NS + t] *= fma(lM[p][5], rd[g & 1][i], u0s);
          }
        });
        __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from forming all the sums first
      });
    });

    // Renormalisation when the products may have lost RL_BUDGET bits since the last one, by the rows' own bounds (a
    // Q20 read costs 9 bits where the worst row costs 34: every dozen batches instead of every third).  The singlet
    // slot / the product of the lane's sums carry the samples' sums, which no table bounds: every third batch.
#pragma unroll
    for (int k = 0; k < RL_B; ++k) bits += *reinterpret_cast<const int32_t*>((const char*)lutg + rcl[k] + RL_BITS * 8);
    if (++cnt == RL_RENORM) {
      cnt = 0;
      prodacc_renorm(accX, exX);
    }
    if (bits > RL_BUDGET) {
      bits = 0;
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        if (t < NXV) {
          prodacc_renorm(acc[t], ex[t]);
        } else {
          int ee;
          acc[t] = frexp(acc[t], &ee);
          exs[t - NXV][j] += ee;
        }
      }
    }
    store_rows((b + 1) & 1, b + 1);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RL_B; ++k) rcs[k] = rns[k], rcl[k] = rnl[k];
  }
  pfx ^= pfv;
  asm volatile("" ::"v"(pfx));

  // ---- the unit's other entries (GEN) ----
  if constexpr (GEN) {
    int64_t gi0, gi1;
    wave_stream_range<EM_GENERAL>(lin, lin_rank, it.e0, it.e1, gi0, gi1);
    const int64_t ng = gi1 - gi0;
    if (ng > 0) {  // (workgroup-uniform)
      const fmx_grec* gr = gen_rec + gi0;
      const int nbg = (int)((ng + RG_B - 1) / RG_B);
      const int V3 = V * 3, PG = nAlpha * 9;
      const int jo = (live ? sel.jbase + j : V - 1) * 3;
      constexpr uint32_t ROWGB = RG_ROW * 8, BUFGB = RG_B * ROWGB;
      // loader role of this wave: entry le of a batch; half 0: ring, singlet factor, u of alpha[0]'s slot and of the
      // launch's alphas 0 and 1 (table rows n0 .. n2); half 1: u of alphas 2, 3 and of the symmetric one
      const int le = w >> 1, lh = w & 1;
      const int n0 = lh == 0 ? 0 : sel.n[2], n1 = lh == 0 ? sel.n[0] : sel.n[3], n2 = lh == 0 ? sel.n[1] : sel.nsym;
      double gl[3], hs[3], q[3][9];
      bool gin = false;
      // the record {entry, snp} of the wave's entry of the NEXT load, requested a batch before the rows that hang off it
      // (round 5: record -> genotype row was a chain of two dependent loads inside one batch of two entries; the record is
      // wave-uniform, so holding one more costs scalar registers only)
      fmx_grec rec_pf = gr[le < ng ? le : ng - 1];
      auto gload = [&](int b) {
        const int64_t i = (int64_t)b * RG_B + le;
        gin = i < ng;
        const fmx_grec r = rec_pf;  // (behind the end: a valid record whose values are not used)
        {
          const int64_t inx = i + RG_B;
          rec_pf = gr[inx < ng ? inx : ng - 1];
        }
        const double* row = gp + (size_t)r.snp * V3;
#pragma unroll
        for (int k = 0; k < 3; ++k) gl[k] = row[jo + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) hs[k] = row[k];  // sample 0's triple multiplies every singlet (:806,828)
        const double* t = pgt + (size_t)(pg_by_record ? gi0 + (gin ? i : ng - 1) : r.e) * PG;  // (rows by record / by entry)
#pragma unroll
        for (int k = 0; k < 9; ++k) q[0][k] = t[n0 * 9 + k], q[1][k] = t[n1 * 9 + k], q[2][k] = t[n2 * 9 + k];
      };
      auto gstore = [&](int buf) {
        double* base = stage_all + (size_t)(buf * RG_B + le) * RG_ROW;
        const bool ok = live && gin;
        const double g0 = ok ? gl[0] : 1.0, g1 = ok ? gl[1] : 0.0, g2 = ok ? gl[2] : 0.0;
        double uu[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int m = 0; m < 3; ++m) uu[a][m] = fma(g2, q[a][6 + m], fma(g1, q[a][3 + m], g0 * q[a][m]));
        if (!gin) {  // an entry behind the end of the stream: every factor exactly 1
#pragma unroll
          for (int a = 0; a < 3; ++a) uu[a][0] = 1.0, uu[a][1] = 0.0, uu[a][2] = 0.0;
        }
        // Layout of a staged row (RG_ROW doubles): the ring of (g0, g1) pairs twice over [128][2], the ring of g2 twice over
        // [128], then the lane's own sixteen values as eight 16-byte pairs P[8][64][2] -- value v = 3 a + m is u[a][m]
        // (a = 0 .. 3: the launch's alphas, 4: the symmetric one), v = 15 the singlet factor; pair v / 2, half v % 2.  One
        // ds_read_b128 then brings two values (round 5: what a DS instruction costs the sweep is per instruction, not per
        // byte -- tools/fp64_ilp_probe.hip -- and a general entry took 88 of them per wave; now 56).
        dbl2* r01 = reinterpret_cast<dbl2*>(base);
        dbl2* P = reinterpret_cast<dbl2*>(base + 384);
        if (lh == 0) {
          r01[j] = dbl2{g0, g1}, r01[j + 64] = dbl2{g0, g1};
          base[256 + j] = g2, base[320 + j] = g2;
          P[0 * 64 + j] = dbl2{uu[1][0], uu[1][1]};
          P[1 * 64 + j] = dbl2{uu[1][2], uu[2][0]};
          P[2 * 64 + j] = dbl2{uu[2][1], uu[2][2]};
          base[384 + (7 * 64 + j) * 2 + 1] = gin ? fma(hs[2], uu[0][2], fma(hs[1], uu[0][1], hs[0] * uu[0][0])) : 1.0;
        } else {
          P[3 * 64 + j] = dbl2{uu[0][0], uu[0][1]};
          P[4 * 64 + j] = dbl2{uu[0][2], uu[1][0]};
          P[5 * 64 + j] = dbl2{uu[1][1], uu[1][2]};
          P[6 * 64 + j] = dbl2{uu[2][0], uu[2][1]};
          base[384 + (7 * 64 + j) * 2] = uu[2][2];
        }
      };
      __syncthreads();  // (the linear walk's last batch has been read by every wave)
      gload(0);
      gstore(0);
      __syncthreads();
      const uint32_t gown = base0 + 3072u + (uint32_t)j * 16u;
      const uint32_t grb2 = base0 + rboff, grs2 = base0 + rsoff;  // (the same slots as the linear walk's)
      const uint32_t grb01 = 2u * grb2 - base0, grs01 = 2u * grs2 - base0;
      // Bits since the last renormalisation: the linear walk leaves at most RL_BUDGET + 8 * 35 = 880 behind; a general entry
      // costs < 37 (every factor >= 1.1e-11): renormalising once more than 870 are counted keeps a product above
      // 2^-(870 + 2 * 37) -- and above 2^-(880 + 74) in the first batch.  (A renormalisation block of its own in front of
      // this loop made the compiler spill inside both walks: 246 instead of 215 ms at configs[2].)
      int gbits = bits;
      for (int b = 0; b < nbg; ++b) {
        const uint32_t bo = (uint32_t)(b & 1) * BUFGB;
        gload(b + 1);
        wave_for<0, RG_B>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          // the lane's factors (and the singlet factor) of entry k: pairs P[q] that hold a value this launch uses
          dbl2 pv[8];
          wave_for<0, 8>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr bool need = (q < 6 && (2 * q) / 3 < NA) || (q < 6 && (2 * q + 1) / 3 < NA) || (q >= 6 && SYM) || q == 7;
            if constexpr (need) pv[q] = wave_ring_rd128<k * ROWGB + q * 1024>(gown + bo);
          });
          // partner triples GS_ steps at a time, a group ahead of their use: (g0, g1) as one 16-byte read, g2
          constexpr int GS_ = RG_G, NGN = NS / GS_, NGS = SHARE ? 0 : NSY / GS_, NGT = NGN + NGS;
          dbl2 rd01[2][GS_];
          double rd2[2][GS_];
          auto issue = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            wave_for<0, GS_>([&](auto ic) {
              constexpr int i = decltype(ic)::value;
              if constexpr (g < NGN && GS_ * g + i >= 8) {
                constexpr int x = 15 - (GS_ * g + i);
                rd01[g & 1][i] = wave_ring_rd128<k * ROWGB + x * 16>(grb01 + bo);
                rd2[g & 1][i] = wave_ring_rd<k * ROWGB + 2048 + x * 8>(grb2 + bo);
              } else if constexpr (g < NGN) {
                constexpr int x = 7 - (GS_ * g + i);
                rd01[g & 1][i] = wave_ring_rd128<k * ROWGB + x * 16>(grs01 + bo);
                rd2[g & 1][i] = wave_ring_rd<k * ROWGB + 2048 + x * 8>(grs2 + bo);
              } else {
                constexpr int x = 7 - (GS_ * (g - NGN) + i);
                rd01[g & 1][i] = wave_ring_rd128<k * ROWGB + x * 16>(grs01 + bo);
                rd2[g & 1][i] = wave_ring_rd<k * ROWGB + 2048 + x * 8>(grs2 + bo);
              }
            });
          };
          issue(std::integral_constant<int, 0>{});
          wave_for<0, NGT>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr bool lastg = g + 1 == NGT;
            if constexpr (!lastg) issue(std::integral_constant<int, g + 1>{});
            // (the lane's own pairs were requested in front of group 0's ring reads: waiting for these waits for them)
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(lastg ? 0 : 2 * GS_) : "memory");
            wave_for<0, GS_>([&](auto ic) {
              constexpr int i = decltype(ic)::value;
              ring_landed(rd01[g & 1][i], rd2[g & 1][i]);
            });
            if constexpr (g == 0) {
              // the lane's factors were requested by opaque statements too: nothing that uses them may be scheduled
              // in front of the wait above (a "memory" clobber does not order register-only instructions)
              wave_for<0, 8>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr bool need = (q < 6 && (2 * q) / 3 < NA) || (q < 6 && (2 * q + 1) / 3 < NA) || (q >= 6 && SYM) || q == 7;
                if constexpr (need) ring_landed(pv[q]);
              });
              if (w == 0) accX *= pv[7].y;
            }
            wave_for<0, GS_>([&](auto ic) {
              constexpr int i = decltype(ic)::value;
              const double r0 = rd01[g & 1][i].x, r1 = rd01[g & 1][i].y, r2 = rd2[g & 1][i];
              if constexpr (g < NGN) {
                constexpr int t = GS_ * g + i;
                wave_for<0, NA>([&](auto ac) {
                  constexpr int a = decltype(ac)::value;
                  constexpr int v0 = 3 * a, v1 = 3 * a + 1, v2 = 3 * a + 2;
                  const double u0_ = (v0 & 1) ? pv[v0 / 2].y : pv[v0 / 2].x, u1_ = (v1 & 1) ? pv[v1 / 2].y : pv[v1 / 2].x,
                               u2_ = (v2 & 1) ? pv[v2 / 2].y : pv[v2 / 2].x;
                  acc[a * NS + t] *= fma(r2, u2_, fma(r1, u1_, r0 * u0_));  // :738-746
                });
                if constexpr (SHARE && t < 8) acc[NA * NS + t] *= fma(r2, pv[7].x, fma(r1, pv[6].y, r0 * pv[6].x));
              } else {
                constexpr int t = GS_ * (g - NGN) + i;
                acc[NA * NS + t] *= fma(r2, pv[7].x, fma(r1, pv[6].y, r0 * pv[6].x));
              }
            });
            __builtin_amdgcn_sched_barrier(0);
          });
        });
        gbits += 37 * RG_B;
        if (gbits > 870) {
          gbits = 0;
#pragma unroll
          for (int t = 0; t < NACC; ++t) {
            if (t < NXV) {
              prodacc_renorm(acc[t], ex[t]);
            } else {
              int ee;
              acc[t] = frexp(acc[t], &ee);
              exs[t - NXV][j] += ee;
            }
          }
          prodacc_renorm(accX, exX);
        }
        gstore((b + 1) & 1);
        __syncthreads();
      }
    }
  }

  // ---- results: the wave layout llw[c][block][n][step t][lane j], lane j at step t holds (j, k = j - t - 1 mod 64); the
  //      logarithm of the product of the partner's sums comes out of the ring like the partner's rho did
  double* out = ll + ((size_t)it.slab * sel.nblk2 + sel.blk) * nAlpha * 4096;
  const double lx = prodacc_log(accX, exX);
  if (w == 1) stage[0][0][j] = lx, stage[0][0][j + 64] = lx;
  __syncthreads();
  const double logW = stage[0][0][j];
  const uint32_t rb = base0 + rboff, rs = base0 + rsoff;
  wave_for<0, NS>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const int T = t < 8 ? 8 * w + t : 24 + 8 * w + t;  // the slot's rotation step (see SHARE)
    if (T >= 63) return;                               // (the lane itself; wave-uniform)
    double lw = t < 8 ? wave_ring_rd<(7 - (t < 8 ? t : 0)) * 8>(rs) : wave_ring_rd<(15 - (t < 8 ? 8 : t)) * 8>(rb);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lw));
    lw += logW;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int xi = a * NS + t;
      const int32_t e = xi < NXV ? ex[xi < NXV ? xi : 0] : exs[xi < NXV ? 0 : xi - NXV][j];
      out[((size_t)sel.n[a] * 64 + T) * 64 + j] = prodacc_log(acc[xi], e) + lw;
    }
  });
  wave_for<0, NSY>([&](auto tc) {
    constexpr int t = decltype(tc)::value, xi = NA * NS + t;
    double lw = wave_ring_rd<(7 - t) * 8>(rs);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lw));
    const int tt = 8 * w + t, kk = (j - tt - 1) & 63;
    const int32_t e = xi < NXV ? ex[xi < NXV ? xi : 0] : exs[xi < NXV ? 0 : xi - NXV][j];
    const double v = prodacc_log(acc[xi], e) + lw + logW;
    if (tt < 31 || j > kk) {  // distance 32 meets every unordered pair from both ends: one writer
      out[((size_t)sel.nsym * 64 + tt) * 64 + j] = v;
      out[((size_t)sel.nsym * 64 + (62 - tt)) * 64 + kk] = v;
    }
  });
  if (w == 0 && sel.with_singlet) out[j] = lx;  // llw[c][0][0][j]
}

template <int NA, bool SYM>
void ring_launch(muxgl_handle* h, const wave_item* items, int64_t n_items, const double* lut, const double* gm, int A,
                 const ring_sel& sel, const double* pgt, bool pg_by_record, double* llw) {
  if (pgt)
    hipLaunchKernelGGL((demux_ring_lin_kernel<NA, SYM, true>), dim3((unsigned)n_items), dim3(256), 0, h->stream, items, n_items,
                       h->d_lin, h->d_lin_rank, h->d_ring_rec, lut, gm, h->V, A, sel, h->d_gen_rec, h->d_gp, pgt,
                       pg_by_record ? 1 : 0, llw);
  else
    hipLaunchKernelGGL((demux_ring_lin_kernel<NA, SYM, false>), dim3((unsigned)n_items), dim3(256), 0, h->stream, items, n_items,
                       h->d_lin, h->d_lin_rank, h->d_ring_rec, lut, gm, h->V, A, sel, h->d_gen_rec, h->d_gp, pgt, 0, llw);
}

}  // namespace

void demux_ring_release(muxgl_handle* h) {
  dev_free(&h->d_ring_rec);
  dev_free(&h->d_ring_lut);
  h->ring_rec_n = -1;
}

// One launch: the linear entries of every work unit for up to four non-symmetric alphas (sel.n[0 .. na)) and, with
// sel.nsym > 0, the symmetric one.  gm: wave_gm_kernel's moments.  Needs h->d_lin_rank / d_lin_rec (plan_build_bit_streams).
// pgt: the table of per-entry likelihoods [entry][alpha][9] -- the launch then also walks the units' other entries (GEN) and the
// slab it writes is final; NULL: the linear entries only (the caller adds the others with demux_wave.hip's EM_GENERAL).
int demux_ring_lin_launch(muxgl_handle* h, const muxgl_demux_params* p, const wave_item* items, int64_t n_items,
                          const double* gm, int na, const ring_sel& sel, double* llw, const double* pgt, bool pg_by_record) {
  if (h->ring_rec_n != h->n_lin_rec || !h->d_ring_rec) {  // per pileup and genotype set (rows of markers without genotypes)
    if (dev_alloc(h, &h->d_ring_rec, (size_t)h->n_lin_rec + RL_PAD)) return 1;
    const int64_t nr = h->n_lin_rec + RL_PAD;
    hipLaunchKernelGGL(ring_rec_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, h->stream, h->n_lin_rec, h->d_lin_rec,
                       h->d_entry_rptr, h->d_reads, h->d_has_gp, h->d_ring_rec);
    HIPCHK(h, hipGetLastError());
    h->ring_rec_n = h->n_lin_rec;
  }
  if (!h->d_ring_lut && dev_alloc(h, &h->d_ring_lut, (size_t)RL_NLUT * RL_LUTW)) return 1;
  ring_alpha al;
  al.a[0] = p->alpha[0];
  for (int a = 0; a < 4; ++a) al.a[1 + a] = a < na ? p->alpha[sel.n[a]] : p->alpha[0];
  al.a[5] = sel.nsym > 0 ? p->alpha[sel.nsym] : p->alpha[0];
  hipLaunchKernelGGL(ring_lut_kernel, dim3(RL_NLUT), dim3(64), 0, h->stream, al, h->d_lut, h->d_ring_lut);
  const bool sym = sel.nsym > 0;
#define RING(NA, SY) ring_launch<NA, SY>(h, items, n_items, h->d_ring_lut, gm, p->n_alpha, sel, pgt, pg_by_record, llw)
  if (na == 4) sym ? RING(4, true) : RING(4, false);
  else if (na == 2) sym ? RING(2, true) : RING(2, false);
  else if (na == 1) sym ? RING(1, true) : RING(1, false);
  else if (na == 0 && sym) RING(0, true);
  else MUXGL_FAIL(h, "demux_ring_lin_launch: %d alphas", na);
#undef RING
  HIPCHK(h, hipGetLastError());
  return 0;
}
