// demux_wave.hip -- the demuxlet pair sweep for more than 16 samples: one wave per cell, one lane per sample.
//   16 < V <= 32   demux_wave32_kernel: the wave as a ring of 32 (both halves hold the same triples)
//   32 < V <= 64   demux_wave_kernel / demux_wave_multi_kernel, described below
//   64 < V <= 255  the same kernels on 64 x 64 blocks of the pair matrix (wave_blk)
//
// Reference being replaced: cmd_cram_demuxlet.cpp:733-747 (pair sweep); the per-entry likelihoods pG (:655-725) come
// from demux_entry_pg_kernel (demux_kernels.hip), written once per run for all alphas ([nnz][A][9] doubles).
//
// At V = 64 with six alphas an entry carries 18 208 hypotheses (179 kflop): the path is FP64-issue bound, not bandwidth
// bound (SURVEY hard part 3), so the design minimises vector instructions per hypothesis:
//   * lane j keeps sample j's triple g_j (loaded straight from the contiguous 1.5 KB GP row) and, per alpha,
//     u[m] = sum_l g_j[l] * pG[alpha][l][m]; the entry's nine pG values are wave-uniform and are fetched through the
//     scalar cache into SGPRs, so they cost no vector instruction and no LDS traffic;
//   * at rotation step t lane j faces sample (j - t) mod 64; the partner's values are read from a copy of the 64 values
//     in LDS (the "ring", see dw_sweep_gen; round 1 rotated them with DPP moves, which cost as much issue time as the
//     arithmetic they fed): 3 FMA + 1 multiply per hypothesis, or 1 FMA + 1 multiply for the entries with one usable
//     read (demux_ring.hip);
//   * 64 accumulators per lane: alpha = 0.5 (symmetric in (j,k): 32 steps) and a lone alpha (63 steps) get a launch of
//     their own, other alphas go four (or two) at a time, the rotation steps cut into ranges of 16 (32) that are the
//     waves of one workgroup (demux_wave_multi_kernel: the ring reads are shared by the alphas); the singlet slot
//     (j,0,n=0) rides along with the first range;
//   * the walk over a cell's entries is a software pipeline over record streams (dw_walk);
//   * products are kept as mantissa * 2^exponent and turned into one log per (cell, hypothesis).
// Work unit = cell, or a part of a cell longer than 2048 entries (wave_item; 100 k waves at BASELINE configs[2]); the
// units are launched longest first.  Markers without genotypes are neutral by construction (wave_neutral_pg_kernel).
#include <algorithm>
#include <vector>

#include "common.hpp"

namespace {

// Markers without genotypes (gps == NULL, cmd_cram_demuxlet.cpp:733) are skipped by the reference.  The wave kernels do
// not test for them -- a test per entry is a chain of two dependent scalar loads in front of every sweep -- but make
// them neutral instead: their row of the device copy of the GP tensor is (1, 0, 0) for every sample (demux_set_gp) and
// their likelihoods in the pG table are all ones, so every factor they contribute is exactly 1.
__global__ void __launch_bounds__(256)
    wave_neutral_pg_kernel(int64_t nnz, int width, const int32_t* __restrict__ entry_snp,
                           const uint8_t* __restrict__ has_gp, double* __restrict__ pg) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnz || has_gp[entry_snp[e]]) return;
  for (int i = 0; i < width; ++i) pg[(size_t)e * width + i] = 1.0;
}

// The moments of every genotype triple the linear entries use: gm[marker][sample] = (s, rho), s = g0 + g1 + g2,
// rho = (g1 + 2 g2) / s (0 for an all-zero triple, whose hypotheses are 0 through s).
__global__ void __launch_bounds__(256) wave_gm_kernel(int64_t n, const double* __restrict__ gp, double* __restrict__ gm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double g0 = gp[3 * i], g1 = gp[3 * i + 1], g2 = gp[3 * i + 2];
  const double sm = (g0 + g1) + g2;
  gm[2 * i] = sm;
  gm[2 * i + 1] = sm > 0.0 ? fma(2.0, g2, g1) / sm : 0.0;
}

__device__ __forceinline__ double dpp_wror1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x13C, 0xF, 0xF, false);  // wave_ror:1 : lane j <- lane (j-1) mod 64
  hi = __builtin_amdgcn_mov_dpp(hi, 0x13C, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

#ifndef DW_DEPTH
#define DW_DEPTH 3  // entries the walk runs ahead (ring slots of the software pipeline)
#endif
#ifndef DW_TOUCH
#define DW_TOUCH 1  // touch the likelihood rows D - 1 entries ahead (see dw_walk)
#endif
#define DW_G_GEN 2  // ring reads (of three values) per group of the sweep

// The ring.  Rotation t + 1 of a launch that starts at offset s0 brings lane j the value of lane (j - s0 - t - 1) mod 64
// (one lane further with CROSS).  The 64 values are kept twice over in LDS, ring[i] = ring[i + 64] = value of lane i, so
// lane j reads ring[j + 64 - s0 - t - 1]: an immediate offset from ONE base address, &ring[j + 64 - s0 - NS].  A DPP
// rotation of a double is two vector moves, as much issue time as the FMA it feeds; the LDS read travels on the other
// pipe (ds_read_b64: 2 LDS cycles per wave).  Reads are issued a group ahead of their use.
// groups [GB, GE) of the sweep (a sweep is cut in two so that work for the next entry can be placed in its middle)
template <int NA, int NS, int CR, int GB, int GE>
__device__ __forceinline__ void dw_sweep_gen(uint32_t rb, const double (&u)[NA][3], double (&acc)[NA * NS]) {
  constexpr int G = DW_G_GEN;
  static_assert(G == 2, "the wait below names two triples");
  double rd[2][G][3];
  auto issue = [&](auto gc) {
    constexpr int g = decltype(gc)::value;
    wave_for<0, G>([&](auto ic) {
      constexpr int k = decltype(ic)::value, t = g * G + k;
      if constexpr (t < NS) {
        rd[g & 1][k][0] = wave_ring_rd<(NS - 1 - t + CR) * 8>(rb);
        rd[g & 1][k][1] = wave_ring_rd<(NS - 1 - t + CR) * 8 + 1024>(rb);
        rd[g & 1][k][2] = wave_ring_rd<(NS - 1 - t + CR) * 8 + 2048>(rb);
      } else {
        rd[g & 1][k][0] = rd[g & 1][k][1] = rd[g & 1][k][2] = 0.0;
      }
    });
  };
  if constexpr (GB < GE) issue(std::integral_constant<int, GB>{});
  wave_for<GB, GE>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    constexpr int left = NS - (g + 1) * G;
    constexpr int ahead = g + 1 < GE ? 3 * (left < G ? left : G) : 0;
    if constexpr (g + 1 < GE) issue(std::integral_constant<int, g + 1>{});
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(rd[g & 1][0][0]), "+v"(rd[g & 1][0][1]), "+v"(rd[g & 1][0][2]), "+v"(rd[g & 1][1][0]),
                   "+v"(rd[g & 1][1][1]), "+v"(rd[g & 1][1][2])
                 : "n"(ahead));
    wave_for<0, G>([&](auto ic) {
      constexpr int k = decltype(ic)::value, t = g * G + k;
      if constexpr (t < NS) {
#pragma unroll
        for (int a = 0; a < NA; ++a)
          acc[a * NS + t] *= fma(rd[g & 1][k][2], u[a][2], fma(rd[g & 1][k][1], u[a][1], rd[g & 1][k][0] * u[a][0]));  // :738-746
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// More than 64 samples: the V x V pair matrix is cut into 64 x 64 blocks (X, Y).  A diagonal block is the kernel as
// described above on the samples 64X .. 64X+63 (jbase).  An off-diagonal block (CROSS) keeps sample 64X + j in lane j and
// rotates the triples of the samples 64Y + k (kbase) past it: all 64 rotations are pairs, including the unrotated one,
// which is reached by starting one lane ahead.  One result slab llw[c][block][alpha][step][lane] per block.
struct wave_blk {
  int32_t jbase, kbase;  // first sample of the lane's block / of the rotating block
  int32_t blk, nblk2;    // slab of this launch, slabs per cell
};

// Entries with at most one usable read (bit set in `lin`, plan_kernels.hip) have likelihoods that are LINEAR in the two
// genotypes, pG[l][m] = A + Bl*l + Bm*m, and take the one-moment form s_j s_k (A + Bl rho_j + Bm rho_k) -- two vector
// instructions per hypothesis instead of four.  Three quarters of the entries of a typical pileup are such entries.
// In the diagonal blocks they are swept by a kernel of their own (demux_ring.hip), which WRITES the result slab;
// the kernels here then walk the others (EM_GENERAL: the clear bits' record stream) and ADD their log-likelihoods to
// what is there.  EM_ALL is the single launch without the distinction (off-diagonal blocks, 17..32 samples, or
// MUXGL_FLAG_NO_LINEAR_ENTRIES).  (wave_gm_kernel above makes the (s, rho) rows the ring kernel stages.)

// The walk of one work unit: entries [i0, i1) of a record stream {entry, snp} (STREAM: the non-linear entries of the
// cell, plan_build_bit_streams), or the entries i0 .. i1 - 1 themselves.  A cell's entries meet marker rows all
// over the genotype tensor and the table of likelihoods is 72 bytes per (entry, alpha): every entry is a chain of
// misses a microsecond long, and with 64 accumulators per lane only two waves share a SIMD.  So the walk runs ahead:
//   * records D entries ahead (scalar loads), genotype triples D - 1 ahead, in rings of D register slots addressed
//     statically (the loop is unrolled D times); the loads are unconditional -- a lane without a sample reads the last
//     sample's triple, an entry behind the end the last record -- because the compiler cannot count loads under a lane
//     mask and would wait for all of them at every use;
//   * the likelihoods of the NEXT entry are requested in two batches, one before the sweep and one in its middle, and
//     the lane's factors u for the next entry are formed from them in the middle of the sweep and right after it --
//     two batches because the nine likelihoods of four alphas are 72 scalar registers.
template <int NA, int NS, bool WITH_SINGLET, bool CROSS, int EM, bool EXL>
__device__ __forceinline__ void dw_walk(int64_t i0, int64_t i1, const fmx_grec* __restrict__ rec,
                                        const int32_t* __restrict__ entry_snp, const double* __restrict__ pg, int PG,
                                        const int32_t (&seln)[NA], const double* __restrict__ gp, int V3, int jo, int ko,
                                        bool live, bool live2, int j, int s0, double (*ring)[128], int32_t (*exs)[64],
                                        double (&acc)[NA * NS], int32_t (&ex)[EXL ? 1 : NA * NS], double& accS, int32_t& exS) {
  static_assert(EM == EM_ALL || EM == EM_GENERAL, "the linear entries' walk lives in demux_ring.hip");
  constexpr bool STREAM = EM != EM_ALL;
  constexpr int D = (CROSS || NA * NS > 32) ? 2 : DW_DEPTH;  // (64 accumulators per lane: two slots is what fits)
  constexpr int HA = (NA + 1) / 2, HB = NA - HA;  // alphas of the first / second batch of likelihoods
  constexpr int CR = CROSS ? 1 : 0;
  constexpr int NG = (NS + DW_G_GEN - 1) / DW_G_GEN, NGH = NG / 2;
  if (i0 >= i1) return;
  const uint32_t rb = (uint32_t)(uintptr_t)&ring[0][j + 64 - s0 - NS];
  int64_t ide[D];
  int32_t ids[D];
  double g[D][3], p[CROSS ? D : 1][3];  // triple of sample jbase + j (and of kbase + j) at the slot's marker
  auto load_id = [&](int64_t i, auto sc) {  // clamped: a valid record is read behind the end, and not used
    constexpr int s = decltype(sc)::value;
    const int64_t ic = i < i1 ? i : i1 - 1;
    if (STREAM) ide[s] = rec[ic].e, ids[s] = rec[ic].snp;
    else ide[s] = ic, ids[s] = entry_snp[ic];
  };
  auto load_gp = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    const double* row = gp + (size_t)ids[s] * V3 + jo;
#pragma unroll
    for (int k = 0; k < 3; ++k) g[s][k] = row[k];
    if (CROSS) {
      constexpr int sq = CROSS ? s : 0;
      const double* rowp = gp + (size_t)ids[s] * V3 + ko;
      p[sq][0] = rowp[0], p[sq][1] = rowp[1], p[sq][2] = rowp[2];
    }
  };
  // factors of the current entry, and those of the next entry that are formed in the middle of the sweep
  double u[NA][3], un[HA][3], sv = 1.0;
  double rv[3] = {1.0, 0.0, 0.0};  // the next entry's ring values
  double qa[HA][9], qb[HB > 0 ? HB : 1][9], qs[WITH_SINGLET ? 9 : 1], hs[WITH_SINGLET ? 3 : 1];
  auto row_of = [&](auto sc) {  // row of the table for the entry in slot s
    constexpr int s = decltype(sc)::value;
    return pg + (size_t)ide[s] * PG;
  };
  auto load_a = [&](auto sc) {
    const double* row = row_of(sc);
#pragma unroll
    for (int a = 0; a < HA; ++a) {
      const double* q = row + (size_t)seln[a] * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) qa[a][k] = q[k];
    }
  };
  auto load_b = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    const double* row = row_of(sc);
#pragma unroll
    for (int a = 0; a < HB; ++a) {
      const double* q = row + (size_t)seln[HA + a] * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) qb[a][k] = q[k];
    }
    if (WITH_SINGLET) {
      const double* h = gp + (size_t)ids[s] * V3;  // sample 0's triple multiplies every singlet (:806,828)
#pragma unroll
      for (int k = 0; k < 9; ++k) qs[k] = row[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) hs[k] = h[k];
    }
  };
  auto factor = [&](double (&uu)[3], const double (&q)[9], double g0, double g1, double g2) {
    uu[0] = fma(g2, q[6], fma(g1, q[3], g0 * q[0]));
    uu[1] = fma(g2, q[7], fma(g1, q[4], g0 * q[1]));
    uu[2] = fma(g2, q[8], fma(g1, q[5], g0 * q[2]));
  };
  auto comp_a = [&](auto sc) {  // first batch: the ring values and the factors of alphas 0 .. HA-1
    constexpr int s = decltype(sc)::value, sq = CROSS ? s : 0;
    const double g0 = live ? g[s][0] : 1.0, g1 = live ? g[s][1] : 0.0, g2 = live ? g[s][2] : 0.0;
    if constexpr (CROSS) {
      rv[0] = live2 ? p[sq][0] : 1.0, rv[1] = live2 ? p[sq][1] : 0.0, rv[2] = live2 ? p[sq][2] : 0.0;
    } else {
      rv[0] = g0, rv[1] = g1, rv[2] = g2;
    }
#pragma unroll
    for (int a = 0; a < HA; ++a) factor(un[a], qa[a], g0, g1, g2);
  };
  auto comp_b = [&](auto sc) {  // behind the sweep: the current entry's factors are free to be overwritten
    constexpr int s = decltype(sc)::value;
    const double g0 = live ? g[s][0] : 1.0, g1 = live ? g[s][1] : 0.0, g2 = live ? g[s][2] : 0.0;
#pragma unroll
    for (int a = 0; a < HA; ++a)
#pragma unroll
      for (int k = 0; k < 3; ++k) u[a][k] = un[a][k];
#pragma unroll
    for (int a = 0; a < HB; ++a) factor(u[HA + a], qb[a], g0, g1, g2);
    if (WITH_SINGLET) {
      const double v0 = fma(g2, qs[6], fma(g1, qs[3], g0 * qs[0]));
      const double v1 = fma(g2, qs[7], fma(g1, qs[4], g0 * qs[1]));
      const double v2 = fma(g2, qs[8], fma(g1, qs[5], g0 * qs[2]));
      sv = fma(hs[2], v2, fma(hs[1], v1, hs[0] * v0));
    }
  };
  auto ring_put = [&]() {
    ring[0][j] = rv[0], ring[0][j + 64] = rv[0];
    ring[1][j] = rv[1], ring[1][j + 64] = rv[1];
    ring[2][j] = rv[2], ring[2][j + 64] = rv[2];
  };
  // The table of likelihoods is read through the scalar cache one entry ahead -- not enough for a first touch of its
  // lines, which come from HBM.  So the row of the entry whose record has just arrived (D - 1 entries ahead) is touched
  // with one vector load, 8 bytes per lane; the value is folded into a word nobody reads (its consumer sits D steps
  // later, where the load has long landed) so that the compiler keeps and counts the load.
  unsigned long long pfv[D], pfx = 0;
  auto touch = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    pfx ^= pfv[s];
    pfv[s] = ((const unsigned long long*)row_of(sc))[j < PG ? j : PG - 1];
  };
  wave_for<0, D>([&](auto sc) { pfv[decltype(sc)::value] = 0; });
  wave_for<0, D>([&](auto sc) { load_id(i0 + decltype(sc)::value, sc); });
  wave_for<0, D - 1>([&](auto sc) { load_gp(sc); });
  {
    using S0 = std::integral_constant<int, 0>;
    load_a(S0{});
    load_b(S0{});
    comp_a(S0{});
    comp_b(S0{});
    ring_put();
  }
  int cnt = 0;
  for (int64_t ib = i0; ib < i1; ib += D) {
    wave_for<0, D>([&](auto sc) {
      constexpr int s = decltype(sc)::value, s1 = (s + 1) % D, sp = (s + D - 1) % D;
      using S1 = std::integral_constant<int, s1>;
      const int64_t i = ib + s;
      if (i >= i1) return;  // (the factors formed for the entry behind the last one are not used)
      load_gp(std::integral_constant<int, sp>{});  // entry i + D - 1: its record was read a step ago
      if (DW_TOUCH) touch(std::integral_constant<int, sp>{});
      load_a(S1{});
      load_id(i + D, sc);
      if (WITH_SINGLET) accS *= sv;
      dw_sweep_gen<NA, NS, CR, 0, NGH>(rb, u, acc);
      comp_a(S1{});
      load_b(S1{});
      dw_sweep_gen<NA, NS, CR, NGH, NG>(rb, u, acc);
      comp_b(S1{});
      ring_put();  // behind the sweep's reads: the LDS serves one wave's requests in order
      if (++cnt == 16) {  // every factor is >= 1.1e-11: sixteen of them cannot underflow
        cnt = 0;
#pragma unroll
        for (int t = 0; t < NA * NS; ++t) {
          if (EXL) {
            int ee;
            acc[t] = frexp(acc[t], &ee);
            exs[t][j] += ee;
          } else {
            prodacc_renorm(acc[t], ex[t]);
          }
        }
        if (WITH_SINGLET) prodacc_renorm(accS, exS);
      }
    });
  }
  if (DW_TOUCH) {
    wave_for<0, D>([&](auto sc) { pfx ^= pfv[decltype(sc)::value]; });
    if (pfx == 0x9E3779B97F4A7C15ull) accS *= 1.0;  // (never: a use the compiler cannot see through)
    asm volatile("" ::"v"(pfx));
  }
}

// NSHIFT = 32 for the symmetric alpha 0.5, 63 otherwise (64 with CROSS).  WITH_SINGLET: also accumulate llksAB[j][0][n=0].
template <int NSHIFT, bool WITH_SINGLET, bool CROSS = false, int EM = EM_ALL>
__global__ void __launch_bounds__(64, 2)
    demux_wave_kernel(const wave_item* __restrict__ items, int64_t n_items, const int64_t* __restrict__ cell_ptr,
                      const int32_t* __restrict__ entry_snp, const double* __restrict__ pg,
                      const uint32_t* __restrict__ lin, const int64_t* __restrict__ lin_rank,
                      const fmx_grec* __restrict__ rec_gen, const double* __restrict__ gp,
                      const uint8_t* __restrict__ has_gp, int V, int nAlpha, int n_sel, wave_blk wb, double* __restrict__ ll) {
  if ((int64_t)blockIdx.x >= n_items) return;
  const wave_item it = items[blockIdx.x];
  const int64_t c = it.slab;  // slab index: the cell id, or an overflow slab
  if (it.e0 == it.e1) return;
  const int j = threadIdx.x;
  const bool live = wb.jbase + j < V;
  const bool live2 = wb.kbase + j < V;
  const int V3 = V * 3;
  const int TW = nAlpha * 9;  // row width of the table the walk reads
  const int jo = (live ? wb.jbase + j : V - 1) * 3, ko = (live2 ? wb.kbase + j : V - 1) * 3;  // (no sample: any valid row)

  constexpr bool EXL = NSHIFT > 32;  // 63 / 64 accumulators per lane: their exponents live in LDS
  __shared__ double ring[3][128];    // the partner values of the current entry, see dw_sweep_gen
  __shared__ int32_t exs[EXL ? NSHIFT : 1][64];
  const uint32_t rb = (uint32_t)(uintptr_t)&ring[0][j + 64 - NSHIFT];
  double acc[NSHIFT], accS = 1.0;
  int32_t ex[EXL ? 1 : NSHIFT], exS = 0;
#pragma unroll
  for (int t = 0; t < NSHIFT; ++t) {
    acc[t] = 1.0;
    if (EXL) exs[t][j] = 0;
    else ex[t] = 0;
  }
  int64_t i0 = it.e0, i1 = it.e1;
  if constexpr (EM != EM_ALL) wave_stream_range<EM>(lin, lin_rank, it.e0, it.e1, i0, i1);
  const int32_t seln[1] = {n_sel};
  dw_walk<1, NSHIFT, WITH_SINGLET, CROSS, EM, EXL>(i0, i1, rec_gen, entry_snp, pg, TW, seln, gp, V3, jo, ko, live, live2, j, 0, ring,
                                                   exs, acc, ex, accS, exS);

  // Results go to the wave layout llw[c][n][step t][lane j] (coalesced; the partner of (t, j) is re-derived by the
  // readers with the same rotation): lane j, step t holds the hypothesis (j, k = j - t - 1 mod 64).  Alpha = 0.5 fills
  // the mirrored half too: the pair met at step t by lane j is met at step 62 - t by lane k.
  double* out = ll + ((size_t)c * wb.nblk2 + wb.blk) * nAlpha * 4096;
  int kk = j;
  wave_for<0, NSHIFT>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    kk = __builtin_amdgcn_mov_dpp(kk, 0x13C, 0xF, 0xF, false);
    double v = prodacc_log(acc[t], EXL ? exs[t][j] : ex[EXL ? 0 : t]);
    if (n_sel > 0) {
      if (NSHIFT >= 63) {
        double* o = out + ((size_t)n_sel * 64 + t) * 64 + j;
        if (EM == EM_GENERAL) v += *o;  // on top of the linear entries' launch
        *o = v;
      } else if (t < 31 || j > kk) {  // step 32 of 64 lanes meets every unordered pair twice: one writer
        double* o = out + ((size_t)n_sel * 64 + t) * 64 + j;
        if (EM == EM_GENERAL) v += *o;
        *o = v;
        out[((size_t)n_sel * 64 + (62 - t)) * 64 + kk] = v;
      }
    }
  });
  if (WITH_SINGLET) {
    double v = prodacc_log(accS, exS);  // llw[c][0][0][j]
    if (EM == EM_GENERAL) v += out[j];
    out[j] = v;
  }
}

// Several non-symmetric alphas in one launch.  The partner values of a step (three ring reads, one for a linear entry)
// do not depend on alpha, only u does: with NA alphas a step costs 3 reads + 4*NA vector instructions for NA hypotheses.
// The accumulator budget (64 per lane) is kept by giving a WAVE NS = 64/NA of the 63 rotation steps, starting at offset
// s0 = NS * wave: the ring is read at that offset.
struct wave_sel {
  int32_t n[4];
};
template <int NA, int NS, bool WITH_SINGLET, bool CROSS = false, int EM = EM_ALL>
__global__ void __launch_bounds__(64 * (64 / NS), 2)
    demux_wave_multi_kernel(const wave_item* __restrict__ items, int64_t n_items, const int64_t* __restrict__ cell_ptr,
                            const int32_t* __restrict__ entry_snp, const double* __restrict__ pg,
                            const uint32_t* __restrict__ lin, const int64_t* __restrict__ lin_rank,
                            const fmx_grec* __restrict__ rec_gen, const double* __restrict__ gp,
                            const uint8_t* __restrict__ has_gp, int V, int nAlpha, wave_sel sel, wave_blk wb, double* __restrict__ ll) {
  // One workgroup per work unit, one WAVE per range of NS rotation steps (s0 = 0, NS, 2 NS, ...).  The waves are
  // independent -- own accumulators, own ring and exponents in LDS, no barrier -- but they walk the same entries at the
  // same pace on one CU, so the marker rows and likelihood tables the first of them pulls from HBM are cache hits for
  // the others: a launch per step range instead read every row 64 / NS times from HBM.
  constexpr int NW = 64 / NS;
  if ((int64_t)blockIdx.x >= n_items) return;
  const wave_item it = items[blockIdx.x];
  const int64_t c = it.slab;  // slab index: the cell id, or an overflow slab
  if (it.e0 == it.e1) return;
  const int w = threadIdx.x >> 6, j = threadIdx.x & 63;
  const int s0 = w * NS;
  const bool live = wb.jbase + j < V;
  const bool live2 = wb.kbase + j < V;
  const int V3 = V * 3;
  const int TW = nAlpha * 9;  // row width of the table the walk reads
  const int jo = (live ? wb.jbase + j : V - 1) * 3, ko = (live2 ? wb.kbase + j : V - 1) * 3;  // (no sample: any valid row)

  // 64 accumulators per lane; their integer exponents live in LDS (touched once per 16 entries), 16 KB per wave
  __shared__ int32_t exs_all[NW][NA * NS][64];
  __shared__ double ring_all[NW][3][128];  // the partner values of the current entry, see dw_sweep_gen
  int32_t (*exs)[64] = exs_all[w];
  double (*ring)[128] = ring_all[w];
  const uint32_t rb = (uint32_t)(uintptr_t)&ring[0][j + 64 - s0 - NS];
  double acc[NA * NS], accS = 1.0;
  int32_t ex[1] = {0}, exS = 0;
#pragma unroll
  for (int t = 0; t < NA * NS; ++t) {
    acc[t] = 1.0;
    exs[t][j] = 0;
  }
  int64_t i0 = it.e0, i1 = it.e1;
  if constexpr (EM != EM_ALL) wave_stream_range<EM>(lin, lin_rank, it.e0, it.e1, i0, i1);
  int32_t seln[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) seln[a] = sel.n[a];
  if (WITH_SINGLET && w == 0)  // the singlet slot rides along with the first step range only
    dw_walk<NA, NS, WITH_SINGLET, CROSS, EM, true>(i0, i1, rec_gen, entry_snp, pg, TW, seln, gp, V3, jo, ko, live, live2, j, s0, ring,
                                                   exs, acc, ex, accS, exS);
  else
    dw_walk<NA, NS, false, CROSS, EM, true>(i0, i1, rec_gen, entry_snp, pg, TW, seln, gp, V3, jo, ko, live, live2, j, s0, ring, exs,
                                            acc, ex, accS, exS);

  double* out = ll + ((size_t)c * wb.nblk2 + wb.blk) * nAlpha * 4096;  // wave layout, see demux_wave_kernel
  wave_for<0, NS>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    if (s0 + t >= (CROSS ? 64 : 63)) return;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      double* o = out + ((size_t)sel.n[a] * 64 + s0 + t) * 64 + j;
      double v = prodacc_log(acc[a * NS + t], exs[a * NS + t][j]);
      if (EM == EM_GENERAL) v += *o;  // on top of the linear entries' launch
      *o = v;
    }
  });
  if (WITH_SINGLET && w == 0) {
    double v = prodacc_log(accS, exS);
    if (EM == EM_GENERAL) v += out[j];
    out[j] = v;
  }
}

// 16 < V <= 32: a ring of 32.  Both 32-lane halves of the wave hold the same 32 sample triples (lane j and lane j + 32:
// sample j & 31), so the 64-lane rotation wave_ror:1 IS the rotation of the ring of 32 in each half.  The stationary
// side differs: lane j of the upper half works for sample (j + 16) & 31, i.e. it sees the ring 16 positions further
// on, so 16 steps meet all 31 partners of every sample (lower half: offsets 1..16, upper half: 17..32, the last one
// being the sample itself) -- against 63 steps of which more than half face idle lanes when the wave is a ring of 64.
// NA alphas per launch share the rotation as in demux_wave_multi_kernel; an alpha of 0.5 is just one of them (both
// orders of a pair are computed, the one with the larger sample on the stationary side is stored twice).  Results go
// to the slab positions the 64-lane layout assigns to (sample, partner): the call kernel needs no second code path.
// ALLSYM: every alpha of the launch is 0.5 (the reference's default grid has just that one): the ring offsets 1..16 are
// all the unordered pairs, so the halves sit eight positions apart and eight steps do (offset 16 meets a pair from
// both ends: one writer).
template <int NA, bool WITH_SINGLET, bool ALLSYM = false>
__global__ void __launch_bounds__(64, 2)
    demux_wave32_kernel(const wave_item* __restrict__ items, int64_t n_items, const int64_t* __restrict__ cell_ptr,
                        const int32_t* __restrict__ entry_snp, const double* __restrict__ pg,
                        const uint32_t* __restrict__ /*lin: the rings of 32 keep the single launch*/,
                        const int64_t* __restrict__, const fmx_grec* __restrict__, const double* __restrict__ gp,
                        const uint8_t* __restrict__ has_gp, int V, int nAlpha,
                        wave_sel sel, uint32_t symmask, double* __restrict__ ll) {
  constexpr int NS = ALLSYM ? 8 : 16;
  if ((int64_t)blockIdx.x >= n_items) return;
  const wave_item it = items[blockIdx.x];
  const int64_t c = it.slab;  // slab index: the cell id, or an overflow slab
  const int64_t e0 = it.e0, e1 = it.e1;
  if (e0 == e1) return;
  const int j = threadIdx.x;
  const int half = j >> 5, sj = j & 31;  // ring position
  const int so = (sj + NS * half) & 31;  // the sample this lane works for
  const bool live = so < V, rlive = sj < V;
  const int V3 = V * 3;
  const int PG = nAlpha * 9;

  __shared__ int32_t exs[NA * NS][64];
  double acc[NA * NS], accS = 1.0;
  int32_t exS = 0;
#pragma unroll
  for (int t = 0; t < NA * NS; ++t) {
    acc[t] = 1.0;
    exs[t][j] = 0;
  }

  int64_t e = e0;
  double ng0 = 1.0, ng1 = 0.0, ng2 = 0.0;  // own triple (sample so)
  double nr0 = 1.0, nr1 = 0.0, nr2 = 0.0;  // ring triple (sample sj)
  if (e < e1) {
    const double* row = gp + (size_t)entry_snp[e] * V3;
    if (live) ng0 = row[so * 3], ng1 = row[so * 3 + 1], ng2 = row[so * 3 + 2];
    if (rlive) nr0 = row[sj * 3], nr1 = row[sj * 3 + 1], nr2 = row[sj * 3 + 2];
  }
  int cnt = 0;
  while (e < e1) {
    const int64_t ecur = e;
    const int32_t scur = entry_snp[ecur];
    const double g0 = ng0, g1 = ng1, g2 = ng2;
    double r0 = nr0, r1 = nr1, r2 = nr2;
    ++e;
    ng0 = 1.0, ng1 = 0.0, ng2 = 0.0;
    nr0 = 1.0, nr1 = 0.0, nr2 = 0.0;
    if (e < e1) {
      const double* row = gp + (size_t)entry_snp[e] * V3;
      if (live) ng0 = row[so * 3], ng1 = row[so * 3 + 1], ng2 = row[so * 3 + 2];
      if (rlive) nr0 = row[sj * 3], nr1 = row[sj * 3 + 1], nr2 = row[sj * 3 + 2];
    }
    if (WITH_SINGLET) {
      const double* s = pg + (size_t)ecur * PG;
      const double* h = gp + (size_t)scur * V3;  // sample 0's triple multiplies every singlet (:806,828)
      const double v0 = fma(g2, s[6], fma(g1, s[3], g0 * s[0]));
      const double v1 = fma(g2, s[7], fma(g1, s[4], g0 * s[1]));
      const double v2 = fma(g2, s[8], fma(g1, s[5], g0 * s[2]));
      accS *= fma(h[2], v2, fma(h[1], v1, h[0] * v0));
    }
    double u[NA][3];
#pragma unroll
    for (int a = 0; a < NA; ++a) {  // wave-uniform likelihoods through the scalar cache
      const double* q = pg + (size_t)ecur * PG + (size_t)sel.n[a] * 9;
      u[a][0] = fma(g2, q[6], fma(g1, q[3], g0 * q[0]));
      u[a][1] = fma(g2, q[7], fma(g1, q[4], g0 * q[1]));
      u[a][2] = fma(g2, q[8], fma(g1, q[5], g0 * q[2]));
    }
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      r0 = dpp_wror1(r0);
      r1 = dpp_wror1(r1);
      r2 = dpp_wror1(r2);
#pragma unroll
      for (int a = 0; a < NA; ++a) acc[a * NS + t] *= fma(r2, u[a][2], fma(r1, u[a][1], r0 * u[a][0]));  // :738-746
    }
    if (++cnt == 16) {  // every factor is >= 1.1e-11: sixteen of them cannot underflow
      cnt = 0;
#pragma unroll
      for (int t = 0; t < NA * NS; ++t) {
        int ee;
        acc[t] = frexp(acc[t], &ee);
        exs[t][j] += ee;
      }
      if (WITH_SINGLET) prodacc_renorm(accS, exS);
    }
  }

  double* out = ll + (size_t)c * nAlpha * 4096;  // wave layout, see demux_wave_kernel
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const int k = (sj - t - 1) & 31;   // wave_ror:1 brings lane j the value of lane j - 1: here inside the ring of 32
    const int tt = (so - k - 1) & 63;  // the step at which the 64-lane layout has sample so facing sample k
    if (live && k < V && k != so) {
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int n = sel.n[a];
        const double v = prodacc_log(acc[a * NS + t], exs[a * NS + t][j]);
        if ((symmask >> n) & 1u) {  // alpha 0.5: one writer per unordered pair, mirrored (as on the other paths)
          // (ALLSYM: a pair is met once, except at ring offset 16, the last step of the upper half)
          if (ALLSYM ? !(half == 1 && t == NS - 1 && so < k) : so > k) {
            out[((size_t)n * 64 + tt) * 64 + so] = v;
            out[((size_t)n * 64 + ((k - so - 1) & 63)) * 64 + k] = v;
          }
        } else {
          out[((size_t)n * 64 + tt) * 64 + so] = v;
        }
      }
    }
  }
  if (WITH_SINGLET && half == 0 && live) out[so] = prodacc_log(accS, exS);  // llw[c][0][0][j]
}

// adds the overflow slabs of a cut cell, in entry order, into the cell's slab.  grid = (cut cells, slab pieces)
__global__ void __launch_bounds__(256)
    wave_combine_kernel(const wave_cut* __restrict__ cuts, int64_t slab_doubles, double* __restrict__ llw) {
  const wave_cut cu = cuts[blockIdx.x];
  double* dst = llw + (size_t)cu.cell * slab_doubles;
  for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < slab_doubles; i += (int64_t)gridDim.y * blockDim.x) {
    double v = dst[i];
    for (int64_t q = 0; q < cu.count; ++q) v += llw[(size_t)(cu.first + q) * slab_doubles + i];
    dst[i] = v;  // positions no launch writes hold whatever they held: nobody reads them
  }
}

// wave layout -> the ABI's [C][V][V][A] tensor (when the caller asks for it, and for V > 64, where the call kernel reads
// the tensor).  grid = (C, blocks per cell); block (X, Y) of a symmetric alpha exists for X >= Y only and is mirrored.
__global__ void __launch_bounds__(64)
    demux_wave_to_full_kernel(const double* __restrict__ llw, const int64_t* __restrict__ cell_ptr, int V, int nAlpha,
                              int nblk, uint32_t symmask, double* __restrict__ ll) {
  const int64_t c = blockIdx.x;
  const int X = (int)blockIdx.y / nblk, Y = (int)blockIdx.y % nblk;
  const int j = threadIdx.x;
  const int sj = 64 * X + j;
  if (sj >= V || cell_ptr[c] == cell_ptr[c + 1]) return;  // the sweep leaves nothing behind for an empty cell
  const double* in = llw + ((size_t)c * nblk * nblk + blockIdx.y) * nAlpha * 4096;
  double* out = ll + (size_t)c * V * V * nAlpha;
  if (Y == 0 && X == Y) out[(size_t)sj * V * nAlpha] = in[j];
  if (Y == 0 && X != Y) {  // singlet slot (sj, 0, 0) of the samples beyond the first block: kept by the diagonal block
    const double* dg = llw + ((size_t)c * nblk * nblk + (size_t)X * nblk + X) * nAlpha * 4096;
    out[(size_t)sj * V * nAlpha] = dg[j];
  }
  const bool cross = X != Y;
  for (int n = 1; n < nAlpha; ++n) {
    const bool sym = (symmask >> n) & 1u;
    if (sym && X < Y) continue;  // filled by the mirror of (Y, X)
    for (int t = 0; t < (cross ? 64 : 63); ++t) {
      const int k = cross ? ((j - t) & 63) : ((j - t - 1) & 63);  // wave_ror:1 brings lane j the value of lane j - 1
      const int sk = 64 * Y + k;
      if (sk >= V) continue;
      const double v = in[((size_t)n * 64 + t) * 64 + j];
      out[((size_t)sj * V + sk) * nAlpha + n] = v;
      if (sym && cross) out[((size_t)sk * V + sj) * nAlpha + n] = v;
    }
  }
}

}  // namespace

struct muxgl_wave_state {
  int32_t* d_order = nullptr;  // cells, longest first
  wave_item* d_items = nullptr;  // work units of the demuxlet wave kernels, longest first
  wave_cut* d_cuts = nullptr;    // cells cut into several units
  int64_t n_items = 0, n_cuts = 0, n_over = 0;
  double* d_pg = nullptr;      // [nnz][A][9]
  double* d_gm = nullptr;      // [S][V][2], see wave_gm_kernel
  size_t gm_cap = 0;
  size_t pg_cap = 0;
};

namespace {
__global__ void __launch_bounds__(256)
    gp_neutral_rows_kernel(int64_t S, int V, const uint8_t* __restrict__ has_gp, double* __restrict__ gp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (marker, sample)
  if (i >= S * V) return;
  const int64_t s = i / V;
  if (has_gp[s]) return;
  gp[i * 3] = 1.0;
  gp[i * 3 + 1] = 0.0;
  gp[i * 3 + 2] = 0.0;
}
}  // namespace

// every kernel that reads d_gp either tests has_gp first or (wave kernels) relies on these neutral rows
int demux_gp_neutral_rows(muxgl_handle* h, int V) {
  const int64_t n = h->S * V;
  if (n > 0)
    hipLaunchKernelGGL(gp_neutral_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->S, V,
                       h->d_has_gp, h->d_gp);
  HIPCHK(h, hipGetLastError());
  return 0;
}

const int32_t* demux_wave_order(const muxgl_handle* h) { return h->wave ? h->wave->d_order : nullptr; }

int demux_wave_items(const muxgl_handle* h, const wave_item** items, int64_t* n_items, const wave_cut** cuts,
                     int64_t* n_cuts, int64_t* n_over) {
  if (!h->wave) return 1;
  *items = h->wave->d_items;
  *n_items = h->wave->n_items;
  *cuts = h->wave->d_cuts;
  *n_cuts = h->wave->n_cuts;
  *n_over = h->wave->n_over;
  return 0;
}

void demux_wave_free(muxgl_handle* h) {
  muxgl_wave_state* st = h->wave;
  if (!st) return;
  dev_free(&st->d_order);
  dev_free(&st->d_items);
  dev_free(&st->d_cuts);
  dev_free(&st->d_pg);
  dev_free(&st->d_gm);
  delete st;
  h->wave = nullptr;
}

int demux_wave_plan(muxgl_handle* h, const int64_t* cell_ptr) {
  if (!h->wave) h->wave = new muxgl_wave_state();
  muxgl_wave_state* st = h->wave;
  std::vector<int32_t> order((size_t)h->C);
  for (int64_t c = 0; c < h->C; ++c) order[(size_t)c] = (int32_t)c;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
    return cell_ptr[a + 1] - cell_ptr[a] > cell_ptr[b + 1] - cell_ptr[b];
  });
  if (dev_alloc(h, &st->d_order, (size_t)h->C)) return 1;
  if (h->C) HIPCHK(h, hipMemcpy(st->d_order, order.data(), sizeof(int32_t) * h->C, hipMemcpyHostToDevice));
  // work units: long cells in equal parts (see wave_item); longest unit first
  constexpr int64_t WAVE_ITEM = 2048;
  std::vector<wave_item> items;
  std::vector<wave_cut> cuts;
  items.reserve((size_t)h->C);
  int64_t n_over = 0;
  for (int64_t i = 0; i < h->C; ++i) {
    const int64_t c = order[(size_t)i], b = cell_ptr[c], n = cell_ptr[c + 1] - b;
    const int64_t parts = n > WAVE_ITEM ? (n + WAVE_ITEM - 1) / WAVE_ITEM : 1;
    if (parts > 1) cuts.push_back(wave_cut{c, h->C + n_over, parts - 1});
    for (int64_t q = 0; q < parts; ++q)
      items.push_back(wave_item{b + n * q / parts, b + n * (q + 1) / parts, q == 0 ? c : h->C + n_over + q - 1, c});
    n_over += parts - 1;
  }
  std::stable_sort(items.begin(), items.end(),
                   [](const wave_item& a, const wave_item& b) { return a.e1 - a.e0 > b.e1 - b.e0; });
  st->n_items = (int64_t)items.size();
  st->n_cuts = (int64_t)cuts.size();
  st->n_over = n_over;
  if (dev_alloc(h, &st->d_items, items.size()) || dev_alloc(h, &st->d_cuts, cuts.size())) return 1;
  if (!items.empty())
    HIPCHK(h, hipMemcpy(st->d_items, items.data(), sizeof(wave_item) * items.size(), hipMemcpyHostToDevice));
  if (!cuts.empty())
    HIPCHK(h, hipMemcpy(st->d_cuts, cuts.data(), sizeof(wave_cut) * cuts.size(), hipMemcpyHostToDevice));
  return 0;
}

// returns -1 when the wave path does not apply, 0 ok, 1 error
int demux_wave_launch(muxgl_handle* h, const muxgl_demux_params* p) {
  if (!h->wave || h->C == 0 || (h->flags & MUXGL_FLAG_FORCE_TILE_SWEEP)) return -1;
  if (p->n_alpha < 2) return -1;  // singlets only: left to the general path
  // a handful of samples beyond a block boundary fill the extra blocks so thinly that the tile sweep is faster
  // (measured: V = 65 tile 195 ms vs 249 ms, V = 96 tile 497 ms vs 253 ms, per 2000 cells)
  if (h->V > 64 && h->V % 64 != 0 && h->V % 64 <= 8 && h->V < 128) return -1;
  if (h->V <= 16 && !(h->flags & MUXGL_FLAG_FORCE_WAVE_KERNEL)) return -1;  // the row/quad kernels are better there
  muxgl_wave_state* st = h->wave;
  const int A = p->n_alpha, V = h->V;
  const int nblk = (V + 63) / 64, nblk2 = nblk * nblk;  // 64 x 64 blocks of the pair matrix
  size_t need = (size_t)h->nnz * A * 9;
  const size_t llw_need = (size_t)(h->C + st->n_over) * nblk2 * A * 4096;
  // pG table, result slabs and (V > 64) the tensor the call kernel reads must fit comfortably: else the tile sweep
  if (((double)need + (double)llw_need + (nblk > 1 ? (double)h->C * V * V * A : 0.0)) * 8.0 > 230e9) return -1;
  if (llw_need > h->llw_cap) {
    if (dev_alloc(h, &h->d_llw, llw_need)) return 1;
    h->llw_cap = llw_need;
  }
  // linear entries (one usable read) in a launch of their own with the one-moment form, the others on top: see the notes above dw_walk
  const bool use_lin = V > 32 && h->d_lin && !(h->flags & MUXGL_FLAG_NO_LINEAR_ENTRIES);
  if (use_lin && h->n_lin_rec < 0 &&
      plan_build_bit_streams(h, h->d_lin, &h->d_lin_rank, &h->d_lin_rec, &h->d_gen_rec, &h->n_lin_rec))
    return 1;
  if (use_lin) {
    const size_t need_g = (size_t)h->S * V * 2;
    if (need_g > st->gm_cap || !st->d_gm) {
      if (dev_alloc(h, &st->d_gm, need_g)) return 1;
      st->gm_cap = need_g;
    }
  }
  // one block of the pair matrix (V <= 64): nobody reads the rows of the linear entries (the ring kernel takes their
  // (A, Bl, Bm) from a table by allele and quality), so only the rows of the others are computed (lane <-> record of their
  // stream) -- the table stays indexed by entry: in the order of the stream (tried: a quarter of the memory) the
  // sweep that reads it was 5 ms slower at configs[2].  Beyond 64 samples the off-diagonal blocks walk every entry.
  const bool gen_only = use_lin && nblk == 1;
  // the ring kernel walks a unit's other entries too, with the same accumulators: one write of the slab
  // (MUXGL_FLAG_SPLIT_GENERAL_SWEEP: launches of their own on top of it, as in round 3 -- lets tests compare the two)
  const bool ring_gen = use_lin && !(h->flags & MUXGL_FLAG_SPLIT_GENERAL_SWEEP) && h->gp_min_sum >= 0.35;
  // ... and then nobody but its loader reads the table, record by record in the order of the stream: rows by record (a
  // quarter of the entry-indexed table: 10 GB instead of 41 at configs[2]; allocating those 41 GB was most of a first
  // call's extra second)
  const bool pg_by_record = ring_gen && gen_only;
  if (pg_by_record) need = (size_t)(h->nnz - h->n_lin_rec) * A * 9;
  if (need > st->pg_cap) {
    if (dev_alloc(h, &st->d_pg, need ? need : 1)) return 1;
    st->pg_cap = need;
  }
  tic(h, MUXGL_T_DEMUX_SWEEP);
  if (demux_entry_pg_launch(h, p, st->d_pg, gen_only, pg_by_record)) return 1;
  if (h->nnz && !gen_only)  // (gen_only: the kernel writes the neutral rows itself)
    hipLaunchKernelGGL(wave_neutral_pg_kernel, dim3((unsigned)((h->nnz + 255) / 256)), dim3(256), 0, h->stream, h->nnz,
                       A * 9, h->d_entry_snp, h->d_has_gp, st->d_pg);
  const unsigned blocks = (unsigned)st->n_items;
  std::vector<int> plain;  // non-symmetric alphas
  uint32_t symmask = 0;
  for (int n = 1; n < A; ++n) {
    if (p->alpha[n] != 0.5) plain.push_back(n);
    else symmask |= 1u << n;
  }
  if (use_lin) {
    const int64_t ng = h->S * (int64_t)V;
    if (ng) hipLaunchKernelGGL(wave_gm_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, h->stream, ng, h->d_gp, st->d_gm);
  }
#define KARGS                                                                                                    \
  st->d_items, st->n_items, h->d_cell_ptr, h->d_entry_snp, st->d_pg, h->d_lin, h->d_lin_rank, h->d_gen_rec, h->d_gp, \
      h->d_has_gp, V, A
#define MULTI_K(NA, NS, WS, CR, EMODE)                                                                                        \
  hipLaunchKernelGGL((demux_wave_multi_kernel<NA, NS, WS, CR, EMODE>), dim3(blocks), dim3(64 * (64 / NS)), 0, h->stream, KARGS, \
                     sel, wb, h->d_llw)
#define MULTI_LAUNCH(NA, NS, WS, CR)          \
  do {                                        \
    if (use_lin && !(CR)) {                   \
      if (!ring_gen) MULTI_K(NA, NS, WS, false, EM_GENERAL); \
    } else {                                  \
      MULTI_K(NA, NS, WS, CR, EM_ALL);        \
    }                                         \
  } while (0)
#define WAVE_K(NS, WS, CR, EMODE) \
  hipLaunchKernelGGL((demux_wave_kernel<NS, WS, CR, EMODE>), dim3(blocks), dim3(64), 0, h->stream, KARGS, n, wb, h->d_llw)
#define WAVE_LAUNCH(NS, WS, CR)          \
  do {                                   \
    if (use_lin && !(CR)) {              \
      if (!ring_gen) WAVE_K(NS, WS, false, EM_GENERAL); \
    } else {                             \
      WAVE_K(NS, WS, CR, EM_ALL);        \
    }                                    \
  } while (0)
  if (V <= 32) {  // ring of 32: 16 rotation steps, up to four alphas per launch, see demux_wave32_kernel
    std::vector<int> all;
    for (int n = 1; n < A; ++n) all.push_back(n);
    bool first = true;
#define W32_LAUNCH(NA, WS) \
  hipLaunchKernelGGL((demux_wave32_kernel<NA, WS>), dim3(blocks), dim3(64), 0, h->stream, KARGS, sel, symmask, h->d_llw)
    for (size_t done = 0; done < all.size();) {
      const size_t left = all.size() - done;
      const int na = left >= 4 ? 4 : (left >= 2 ? 2 : 1);
      wave_sel sel = {{0, 0, 0, 0}};
      for (int a = 0; a < na; ++a) sel.n[a] = all[done + a];
      if (na == 4) {
        if (first) W32_LAUNCH(4, true);
        else W32_LAUNCH(4, false);
      } else if (na == 2) {
        if (first) W32_LAUNCH(2, true);
        else W32_LAUNCH(2, false);
      } else if ((symmask >> sel.n[0]) & 1u) {  // a lone alpha of 0.5: eight steps
        if (first)
          hipLaunchKernelGGL((demux_wave32_kernel<1, true, true>), dim3(blocks), dim3(64), 0, h->stream, KARGS, sel,
                             symmask, h->d_llw);
        else
          hipLaunchKernelGGL((demux_wave32_kernel<1, false, true>), dim3(blocks), dim3(64), 0, h->stream, KARGS, sel,
                             symmask, h->d_llw);
      } else {
        if (first) W32_LAUNCH(1, true);
        else W32_LAUNCH(1, false);
      }
      HIPCHK(h, hipGetLastError());
      first = false;
      done += na;
    }
#undef W32_LAUNCH
  }
  for (int X = 0; X < (V <= 32 ? 0 : nblk); ++X) {
    // ---- diagonal block: samples 64X.. against themselves.  Non-symmetric alphas four (or two) at a time, see
    //      demux_wave_multi_kernel; the rest one per launch; the singlet slot rides along with the first launch
    {
      const wave_blk wb = {64 * X, 64 * X, X * nblk + X, nblk2};
      bool first = true;
      size_t done = 0;
      // the linear entries of the launch's alphas first (demux_ring.hip; the symmetric alpha rides with the first of
      // these launches), then the others on top
      uint32_t sym_left = symmask;
      auto ring = [&](int na, const int* idx, int nsym) -> int {
        if (!use_lin) return 0;
        ring_sel rs{};
        for (int a = 0; a < na; ++a) rs.n[a] = idx[a];
        if (nsym == 0 && sym_left) nsym = __builtin_ctz(sym_left);
        if (nsym) sym_left &= ~(1u << nsym);
        rs.nsym = nsym;
        rs.with_singlet = first ? 1 : 0;
        rs.jbase = 64 * X, rs.blk = wb.blk, rs.nblk2 = nblk2;
        return demux_ring_lin_launch(h, p, st->d_items, st->n_items, st->d_gm, na, rs, h->d_llw, ring_gen ? st->d_pg : nullptr,
                                     pg_by_record);
      };
      while (plain.size() - done >= 4) {
        wave_sel sel = {{plain[done], plain[done + 1], plain[done + 2], plain[done + 3]}};
        if (ring(4, &plain[done], 0)) return 1;
        if (first) MULTI_LAUNCH(4, 16, true, false);  // (all four step ranges: a wave each)
        else MULTI_LAUNCH(4, 16, false, false);
        HIPCHK(h, hipGetLastError());
        first = false;
        done += 4;
      }
      while (plain.size() - done >= 2) {
        wave_sel sel = {{plain[done], plain[done + 1], 0, 0}};
        if (ring(2, &plain[done], 0)) return 1;
        if (first) MULTI_LAUNCH(2, 32, true, false);
        else MULTI_LAUNCH(2, 32, false, false);
        HIPCHK(h, hipGetLastError());
        first = false;
        done += 2;
      }
      for (int n = 1; n < A; ++n) {
        const bool sym = (p->alpha[n] == 0.5);
        if (!sym && !(done < plain.size() && plain[done] == n)) continue;  // already covered by a multi-alpha launch
        if (sym && ((sym_left >> n) & 1u) && ring(0, nullptr, n)) return 1;  // (not yet taken along)
        if (!sym && ring(1, &plain[done], 0)) return 1;
        if (sym && first) WAVE_LAUNCH(32, true, false);
        else if (sym) WAVE_LAUNCH(32, false, false);
        else if (first) WAVE_LAUNCH(63, true, false);
        else WAVE_LAUNCH(63, false, false);
        HIPCHK(h, hipGetLastError());
        first = false;
        if (!sym) ++done;
      }
    }
    // ---- off-diagonal blocks: every rotation is a pair.  Symmetric alphas only for Y < X (mirrored by the converter)
    for (int Y = 0; Y < nblk; ++Y) {
      if (Y == X) continue;
      const wave_blk wb = {64 * X, 64 * Y, X * nblk + Y, nblk2};
      size_t done = 0;
      while (plain.size() - done >= 4) {
        wave_sel sel = {{plain[done], plain[done + 1], plain[done + 2], plain[done + 3]}};
        MULTI_LAUNCH(4, 16, false, true);
        HIPCHK(h, hipGetLastError());
        done += 4;
      }
      while (plain.size() - done >= 2) {
        wave_sel sel = {{plain[done], plain[done + 1], 0, 0}};
        MULTI_LAUNCH(2, 32, false, true);
        HIPCHK(h, hipGetLastError());
        done += 2;
      }
      for (int n = 1; n < A; ++n) {
        const bool sym = (p->alpha[n] == 0.5);
        if (sym && Y > X) continue;
        if (!sym && !(done < plain.size() && plain[done] == n)) continue;
        WAVE_LAUNCH(64, false, true);
        HIPCHK(h, hipGetLastError());
        if (!sym) ++done;
      }
    }
  }
#undef WAVE_LAUNCH
#undef WAVE_K
#undef MULTI_LAUNCH
#undef MULTI_K
#undef KARGS
  if (st->n_cuts) {  // cells walked in several parts: add the parts' log-likelihoods up
    hipLaunchKernelGGL(wave_combine_kernel, dim3((unsigned)st->n_cuts, 8), dim3(256), 0, h->stream, st->d_cuts,
                       (int64_t)nblk2 * A * 4096, h->d_llw);
    HIPCHK(h, hipGetLastError());
  }
  h->ll_wave = nblk == 1;  // the 64-lane call kernel reads the slab directly; beyond 64 samples it reads the tensor
  if (h->want_full_ll || nblk > 1) {
    if (demux_ensure_ll(h, p)) return 1;
    hipLaunchKernelGGL(demux_wave_to_full_kernel, dim3((unsigned)h->C, (unsigned)nblk2), dim3(64), 0, h->stream, h->d_llw,
                       h->d_cell_ptr, V, A, nblk, symmask, h->d_ll);
    HIPCHK(h, hipGetLastError());
  }
  toc(h, MUXGL_T_DEMUX_SWEEP);
  return 0;
}
