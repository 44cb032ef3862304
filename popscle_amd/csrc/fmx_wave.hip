// fmx_wave.hip -- freemuxlet E-step (cmd_cram_freemux2.cpp:383-456) for 32 < K <= 255 clusters: one wave per cell, one
// lane per cluster, the design of demux_wave.hip with the entry's nine genotype-pair likelihoods in the role of pG.
//
//   * lane j keeps cluster j's genotype posterior gp_j (three doubles from the per-iteration tensor cgp[S][K][3], loaded
//     two to three entries ahead) and u[m] = sum_l gp_j[l] * glis[l][m]; the entry's likelihoods are wave-uniform and
//     come through the scalar cache; entries whose likelihoods are linear in the genotypes need one moment per cluster
//     and one FMA per pair instead (LIN below);
//   * at rotation step t lane j faces cluster (j - t) mod 64: a few rotations are DPP moves (wave_ror:1), the others
//     reads of a copy of the 64 values in LDS (the ring, below); the pair likelihood (:440-446) is symmetric in the two
//     clusters, so 32 steps meet every unordered pair (the last step meets each pair from both sides: one writer); the
//     singlet (:448-452) uses the diagonal of glis only;
//   * products as mantissa * 2^exponent, one log per (cell, hypothesis), written to llks[j(j+1)/2 + k] directly.
// The general pair kernel this replaces for K = 64 ran at 5 % of the FP64 issue rate (one thread per pair, every thread
// fetching both posteriors of every entry).
// Beyond 64 clusters the K x K pair matrix is cut into 64 x 64 blocks as in demux_wave.hip: a diagonal block is the
// kernel above on clusters 64X .. 64X+63; an off-diagonal block (CROSS, X > Y only: the likelihood is symmetric) keeps
// cluster 64X + j in lane j and rotates the posteriors of clusters 64Y + k past it, all 64 rotations being pairs.
#include "common.hpp"

#include <rocprim/device/device_scan.hpp>

namespace {

// rotations of the ring done by DPP moves; the remaining ones are read from the copy in LDS (see the kernel)
#ifndef FMX_TD_LIN
#define FMX_TD_LIN 8
#endif
#ifndef FMX_TD_GEN
#define FMX_TD_GEN 6
#endif
// entries the walks run ahead of the sweep (ring slots of the software pipeline)
#ifndef FMX_DEPTH_LIN
#define FMX_DEPTH_LIN 4
#endif
#ifndef FMX_DEPTH_GEN
#define FMX_DEPTH_GEN 3
#endif
// accumulator exponents of the diagonal-block kernels in LDS (1) or in registers (0)
#ifndef FMX_EXP_LDS
#define FMX_EXP_LDS 1
#endif
// the pair sums of the general entries around the lane's smallest u (see PIV in fw_walk_gen); 0: the three-term sums
#ifndef FMX_PIVOT
#define FMX_PIVOT 1
#endif

__device__ __forceinline__ double fw_wror1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x13C, 0xF, 0xF, false);  // wave_ror:1 : lane j <- lane (j-1) mod 64
  hi = __builtin_amdgcn_mov_dpp(hi, 0x13C, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// LIN: entries flagged in `lin` (fmx_entry_kernel: at most one usable read, no clamp) have likelihoods that are linear
// in g1 + g2, glis[g1][g2] = c0 + c1 (g1 + g2), so that, with s = sum_l P[l] and E = P[1] + 2 P[2],
//     sum_{l,m} P_j[l] P_k[m] glis[l][m] = c0 s_j s_k + c1 (E_j s_k + s_j E_k).
// The posteriors are normalised by the kernel that writes them (s = 1 to within 2 ulp), and the form is taken at s = 1:
//     (c0 + c1 E_j) + c1 E_k,      singlet  c0 + 2 c1 E_j.
// That is a relative difference of <= ~4e-16 per factor from the nine-term sum, ~1e-12 on the log-likelihood of a cell
// of several thousand entries, against the 1e-5 the path is held to (decisions on re-associated arithmetic).
// What it buys: ONE number per (SNP, cluster) -- E, 8 bytes from the side tensor cE[S][K] instead of the 24-byte
// triple -- travels from HBM and round the ring, and a pair costs an FMA with a wave-uniform factor and the product
// update.  Three quarters of the entries of a typical pileup are such entries.  They are read from a stream of their
// own, 24-byte records {c0, c1, snp} in entry order (the other entries: {entry, snp}), so the wave walks its cell
// twice -- linear entries, then the others -- each walk with its own sweep body and software pipeline; a product's
// factors are therefore taken in that order in every run.
//
// The ring.  Rotation t brings lane j the value of lane (j - t) mod 64 (one lane further with CROSS, see below).  The
// first TD rotations are DPP moves of the value itself; the others are read from a copy of the 64 values in LDS, kept
// twice over so that the wrap needs no address arithmetic: lane j reads ring[j + 64 - t], an immediate offset from one
// base address.  A DPP rotation of a double costs two vector moves -- as much issue time as the FMA it feeds -- while
// the LDS read travels on the other pipe (ds_read_b64: 2 LDS cycles per wave, 256 B/clk per CU).
//
// The pipeline.  A cell's entries meet SNP rows all over the posterior tensor: every entry is an HBM miss of 512 B
// (linear) or 1.5 KB (general) per wave, ~1 us away.  The walks therefore run D entries ahead with the stream records
// (scalar loads) and D - 1 entries ahead with the posterior loads, in rings of D register slots addressed statically
// (the loop is unrolled D times); the entry's nine likelihoods and the rotation-independent factors (u, the singlet
// term) are formed one entry ahead, right after the previous sweep.

template <int NS, bool EXL>
__device__ __forceinline__ void fw_renorm(double (&acc)[NS], int32_t (&ex)[EXL ? 1 : NS], int32_t (*exs)[64], int j) {
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    if (EXL) {
      int ee;
      acc[t] = frexp(acc[t], &ee);
      exs[t][j] += ee;
    } else {
      prodacc_renorm(acc[t], ex[t]);
    }
  }
}

// the general sweep over entries [i0, i1) of the stream grec (STREAM), or over the entries i0 .. i1 - 1 themselves
template <bool CROSS, bool STREAM, int NS, bool EXL, bool PIVOT = true>
__device__ __forceinline__ void fw_walk_gen(int64_t i0, int64_t i1, const fmx_grec* __restrict__ grec,
                                            const int32_t* __restrict__ entry_snp, const double* __restrict__ egls,
                                            const double* __restrict__ cgp, int K3, int jo, int ko, bool live, bool live2, int j,
                                            double (*ring)[128], int32_t (*exs)[64], double (&acc)[NS],
                                            int32_t (&ex)[EXL ? 1 : NS], double& accS, int32_t& exS, int& cnt) {
  constexpr int D = CROSS ? 2 : FMX_DEPTH_GEN;  // (64 accumulators: two slots is what fits)
  constexpr int TD = FMX_TD_GEN < NS ? FMX_TD_GEN : NS;
  constexpr int RB = CROSS ? 65 : 64;  // CROSS: rotation 1 meets cluster kbase + j itself
  if (i0 >= i1) return;
  const uint32_t ring_a = (uint32_t)(uintptr_t)&ring[0][j];  // LDS byte address of the lane's ring slot
  int64_t ide[D];
  int32_t ids[D];
  double gp[D][3], gq[CROSS ? D : 1][3];  // posterior of cluster jbase + j (and of kbase + j) at the slot's SNP
  auto load_id = [&](int64_t i, auto sc) {  // clamped: a valid record is read behind the end, and not used
    constexpr int s = decltype(sc)::value;
    const int64_t ic = i < i1 ? i : i1 - 1;
    if (STREAM) ide[s] = grec[ic].e, ids[s] = grec[ic].snp;
    else ide[s] = ic, ids[s] = entry_snp[ic];
  };
  // Loads are unconditional -- a lane without a cluster reads the last cluster's triple, an entry behind the end reads
  // the last record's -- and a lane's value is made neutral where it is used: with loads under a lane mask the
  // compiler cannot count the ones in flight and waits for all of them (s_waitcnt vmcnt(0)) at every use.
  auto load_gp = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    const double* row = cgp + (size_t)ids[s] * K3 + jo;
    gp[s][0] = row[0], gp[s][1] = row[1], gp[s][2] = row[2];
    if (CROSS) {
      constexpr int sq = CROSS ? s : 0;
      const double* rowq = cgp + (size_t)ids[s] * K3 + ko;
      gq[sq][0] = rowq[0], gq[sq][1] = rowq[1], gq[sq][2] = rowq[2];
    }
  };
  double u0 = 0, u1 = 0, u2 = 0, sing = 1.0;
  double c0r = 1.0, c1r = 0, c2r = 0;  // what the ring carries for the current entry
  // PIV (diagonal blocks): the pair sum  sum_m P_k[m] u[m]  of a lane is taken around the lane's SMALLEST u,
  //     u_p + P_k[a] (u_a - u_p) + P_k[b] (u_b - u_p),   {a, b} = {0, 1, 2} \ {p},
  // which uses sum_m P_k[m] = 1 (the posteriors are normalised in FP64 by the kernel that writes them: 1 to within 2 ulp,
  // the assumption the linear form makes too).  Every term is >= 0, so nothing cancels whatever the entry looks like
  // (relative deviation from the three-term sum <= ~4e-16), and a hypothesis costs two FMAs and the product update
  // instead of three and one.  Which two of the partner's three numbers a lane needs depends on the lane: they are read
  // from the ring in LDS through two per-lane base addresses (no DPP rotations in this form), set once per entry.
  constexpr bool PIV = !CROSS && PIVOT && FMX_PIVOT;
  double up = 0, da = 0, db = 0;
  uint32_t ra_a = ring_a, ra_b = ring_a + 1024;
  auto factors = [&](auto sc, double q0, double q1, double q2, double q3, double q4, double q5, double q6, double q7, double q8) {
    constexpr int s = decltype(sc)::value, sq = CROSS ? s : 0;
    const double g0 = live ? gp[s][0] : 1.0, g1 = live ? gp[s][1] : 0.0, g2 = live ? gp[s][2] : 0.0;
    sing = fma(g2, q8, fma(g1, q4, g0 * q0));
    u0 = fma(g2, q6, fma(g1, q3, g0 * q0));
    u1 = fma(g2, q7, fma(g1, q4, g0 * q1));
    u2 = fma(g2, q8, fma(g1, q5, g0 * q2));
    if (CROSS) {
      c0r = live2 ? gq[sq][0] : 1.0, c1r = live2 ? gq[sq][1] : 0.0, c2r = live2 ? gq[sq][2] : 0.0;
    } else {
      c0r = g0, c1r = g1, c2r = g2;
    }
    if constexpr (PIV) {
      const bool p0 = u0 <= u1 && u0 <= u2, p1 = !p0 && u1 <= u2;  // pivot 0 / 1 / (else) 2
      up = p0 ? u0 : (p1 ? u1 : u2);
      da = (p0 ? u1 : u0) - up;  // a = 1 with pivot 0, else 0
      db = ((p0 || p1) ? u2 : u1) - up;  // b = 2 with pivot 0 or 1, else 1
      ra_a = ring_a + (p0 ? 1024u : 0u);
      ra_b = ring_a + ((p0 || p1) ? 2048u : 1024u);
    }
    ring[0][j] = c0r, ring[0][j + 64] = c0r;
    ring[1][j] = c1r, ring[1][j + 64] = c1r;
    ring[2][j] = c2r, ring[2][j + 64] = c2r;
  };
  wave_for<0, D>([&](auto sc) { load_id(i0 + decltype(sc)::value, sc); });
  wave_for<0, D - 1>([&](auto sc) { load_gp(sc); });
  {
    const double* q = egls + (size_t)ide[0] * 9;
    factors(std::integral_constant<int, 0>{}, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8]);
  }
  for (int64_t ib = i0; ib < i1; ib += D) {
    wave_for<0, D>([&](auto sc) {
      constexpr int s = decltype(sc)::value, s1 = (s + 1) % D, sp = (s + D - 1) % D;
      const int64_t i = ib + s;
      if (i >= i1) return;  // (the factors formed for the entry behind the last one are not used)
      load_gp(std::integral_constant<int, sp>{});   // entry i + D - 1: its record was read a step ago
      const double* q = egls + (size_t)ide[s1] * 9;  // wave-uniform: glis[g1*3+g2] of the next entry
      const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7], q8 = q[8];
      load_id(i + D, sc);
      // sweep of entry i
      if (!CROSS) accS *= sing;  // singlet: sum_g glis[g][g] * gp_j[g] (:448-452)
      if constexpr (PIV) {
        constexpr int G = 4;  // pairs of ring reads per group; the reads of group g + 1 are in flight while group g is consumed
        constexpr int NG = (NS + G - 1) / G;
        double rd[2][G][2];
        auto issue = [&](auto gc) {
          constexpr int g = decltype(gc)::value;
          wave_for<0, G>([&](auto ic) {
            constexpr int k = decltype(ic)::value, t = g * G + k;  // rotation t + 1
            if constexpr (t < NS) {
              rd[g & 1][k][0] = wave_ring_rd<(64 - (t + 1)) * 8>(ra_a);
              rd[g & 1][k][1] = wave_ring_rd<(64 - (t + 1)) * 8>(ra_b);
            } else {
              rd[g & 1][k][0] = rd[g & 1][k][1] = 0.0;
            }
          });
        };
        issue(std::integral_constant<int, 0>{});
        wave_for<0, NG>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          constexpr int left = NS - (g + 1) * G;
          constexpr int ahead = g + 1 < NG ? 2 * (left < G ? left : G) : 0;  // younger reads: they may stay in flight
          if constexpr (g + 1 < NG) issue(std::integral_constant<int, g + 1>{});
          asm volatile("s_waitcnt lgkmcnt(%8)"
                       : "+v"(rd[g & 1][0][0]), "+v"(rd[g & 1][0][1]), "+v"(rd[g & 1][1][0]), "+v"(rd[g & 1][1][1]),
                         "+v"(rd[g & 1][2][0]), "+v"(rd[g & 1][2][1]), "+v"(rd[g & 1][3][0]), "+v"(rd[g & 1][3][1])
                       : "n"(ahead));
          wave_for<0, G>([&](auto ic) {
            constexpr int k = decltype(ic)::value, t = g * G + k;
            if constexpr (t < NS) acc[t] *= fma(rd[g & 1][k][1], db, fma(rd[g & 1][k][0], da, up));  // :440-446 as a product
          });
          __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from forming all the sums first
        });
      } else {
      constexpr int G = CROSS ? 1 : 2;  // ring reads per group; the reads of group g + 1 are in flight while group g is consumed
      constexpr int NG = (NS - TD + G - 1) / G;
      double r0 = c0r, r1 = c1r, r2 = c2r;
      if (CROSS && TD > 0) {  // one lane ahead: the first rotation then brings cluster kbase + j itself
        r0 = __shfl(r0, (j + 1) & 63, 64);
        r1 = __shfl(r1, (j + 1) & 63, 64);
        r2 = __shfl(r2, (j + 1) & 63, 64);
      }
      double rd[2][G][3];
      auto issue = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        wave_for<0, G>([&](auto ic) {
          constexpr int k = decltype(ic)::value, t = TD + g * G + k;  // rotation t + 1
          if constexpr (t < NS) {
            rd[g & 1][k][0] = wave_ring_rd<(RB - (t + 1)) * 8>(ring_a);
            rd[g & 1][k][1] = wave_ring_rd<(RB - (t + 1)) * 8 + 1024>(ring_a);
            rd[g & 1][k][2] = wave_ring_rd<(RB - (t + 1)) * 8 + 2048>(ring_a);
          } else {
            rd[g & 1][k][0] = rd[g & 1][k][1] = rd[g & 1][k][2] = 0.0;
          }
        });
      };
      if constexpr (NG > 0) issue(std::integral_constant<int, 0>{});
      wave_for<0, NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int left = NS - TD - (g + 1) * G;
        constexpr int ahead = g + 1 < NG ? 3 * (left < G ? left : G) : 0;  // younger reads: they may stay in flight
        if constexpr (g + 1 < NG) issue(std::integral_constant<int, g + 1>{});
        // the DPP rotations are spread over the groups of ring reads: they fill the time the reads take
        wave_for<g * TD / NG, (g + 1) * TD / NG>([&](auto dc) {
          r0 = fw_wror1(r0);
          r1 = fw_wror1(r1);
          r2 = fw_wror1(r2);
          acc[decltype(dc)::value] *= fma(r2, u2, fma(r1, u1, r0 * u0));  // :440-446 as a product
        });
        if constexpr (G == 2)
          asm volatile("s_waitcnt lgkmcnt(%6)"
                       : "+v"(rd[g & 1][0][0]), "+v"(rd[g & 1][0][1]), "+v"(rd[g & 1][0][2]), "+v"(rd[g & 1][G - 1][0]),
                         "+v"(rd[g & 1][G - 1][1]), "+v"(rd[g & 1][G - 1][2])
                       : "n"(ahead));
        else
          asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(rd[g & 1][0][0]), "+v"(rd[g & 1][0][1]), "+v"(rd[g & 1][0][2]) : "n"(ahead));
        wave_for<0, G>([&](auto ic) {
          constexpr int k = decltype(ic)::value, t = TD + g * G + k;
          if constexpr (t < NS) acc[t] *= fma(rd[g & 1][k][2], u2, fma(rd[g & 1][k][1], u1, rd[g & 1][k][0] * u0));
        });
        __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from forming all the sums first
      });
      if constexpr (TD == NS) {
        wave_for<0, NS>([&](auto dc) {
          r0 = fw_wror1(r0);
          r1 = fw_wror1(r1);
          r2 = fw_wror1(r2);
          acc[decltype(dc)::value] *= fma(r2, u2, fma(r1, u1, r0 * u0));
        });
      }
      }
      // factors of the next entry and its ring (behind the sweep's reads: the LDS serves one wave's requests in order)
      factors(std::integral_constant<int, s1>{}, q0, q1, q2, q3, q4, q5, q6, q7, q8);
      if (++cnt == 16) {  // a factor is >= ~1e-13 (clamped likelihoods, mixed posteriors): sixteen cannot underflow
        cnt = 0;
        fw_renorm<NS, EXL>(acc, ex, exs, j);
        if (!CROSS) prodacc_renorm(accS, exS);
      }
    });
  }
}

// the linear sweep over records [l0, l1) of the stream lrec
template <int NS, bool EXL>
__device__ __forceinline__ void fw_walk_lin(int64_t l0, int64_t l1, const fmx_lrec* __restrict__ lrec,
                                            const double* __restrict__ cE, int K, int sj, bool live, int j,
                                            double (*ring)[128], int32_t (*exs)[64], double (&acc)[NS],
                                            int32_t (&ex)[EXL ? 1 : NS], double& accS, int32_t& exS, int& cnt) {
  constexpr int D = FMX_DEPTH_LIN;
  constexpr int TD = FMX_TD_LIN < NS ? FMX_TD_LIN : NS;
  if (l0 >= l1) return;
  const uint32_t ring_a = (uint32_t)(uintptr_t)&ring[0][j];
  double rc0[D], rc1[D];
  int32_t rsnp[D];
  double En[D];  // E of cluster jbase + j at the slot's SNP
  auto load_rec = [&](int64_t r, auto sc) {
    constexpr int s = decltype(sc)::value;
    const fmx_lrec* p = lrec + (r < l1 ? r : l1 - 1);
    rc0[s] = p->c0, rc1[s] = p->c1, rsnp[s] = p->snp;
  };
  auto load_E = [&](auto sc) {  // unconditional, see fw_walk_gen
    constexpr int s = decltype(sc)::value;
    En[s] = cE[(size_t)rsnp[s] * K + sj];
  };
  double u0 = 0, u1 = 0, sing = 1.0, c0r = 0.0;
  auto factors = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    c0r = live ? En[s] : 0.0;
    u1 = rc1[s];
    u0 = fma(u1, c0r, rc0[s]);
    sing = fma(2.0 * u1, c0r, rc0[s]);
    ring[0][j] = c0r, ring[0][j + 64] = c0r;
  };
  wave_for<0, D>([&](auto sc) { load_rec(l0 + decltype(sc)::value, sc); });
  wave_for<0, D - 1>([&](auto sc) { load_E(sc); });
  factors(std::integral_constant<int, 0>{});
  for (int64_t rb = l0; rb < l1; rb += D) {
    wave_for<0, D>([&](auto sc) {
      constexpr int s = decltype(sc)::value, s1 = (s + 1) % D, sp = (s + D - 1) % D;
      const int64_t r = rb + s;
      if (r >= l1) return;
      load_E(std::integral_constant<int, sp>{});  // entry r + D - 1: its record was read a step ago
      load_rec(r + D, sc);
      // sweep of entry r
      accS *= sing;
      constexpr int G = 4;  // ring reads per group; the reads of group g + 1 are in flight while group g is consumed
      constexpr int NG = (NS - TD + G - 1) / G;
      double r0 = c0r;  // the partner's E
      double rd[2][G];
      auto issue = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        wave_for<0, G>([&](auto ic) {
          constexpr int k = decltype(ic)::value, t = TD + g * G + k;  // rotation t + 1
          if constexpr (t < NS) rd[g & 1][k] = wave_ring_rd<(64 - (t + 1)) * 8>(ring_a);
          else rd[g & 1][k] = 0.0;
        });
      };
      if constexpr (NG > 0) issue(std::integral_constant<int, 0>{});
      wave_for<0, NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int left = NS - TD - (g + 1) * G;
        constexpr int ahead = g + 1 < NG ? (left < G ? left : G) : 0;
        if constexpr (g + 1 < NG) issue(std::integral_constant<int, g + 1>{});
        wave_for<g * TD / NG, (g + 1) * TD / NG>([&](auto dc) {
          r0 = fw_wror1(r0);
          acc[decltype(dc)::value] *= fma(u1, r0, u0);
        });
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(rd[g & 1][0]), "+v"(rd[g & 1][1]), "+v"(rd[g & 1][2]), "+v"(rd[g & 1][3]) : "n"(ahead));
        wave_for<0, G>([&](auto ic) {
          constexpr int k = decltype(ic)::value, t = TD + g * G + k;
          if constexpr (t < NS) acc[t] *= fma(u1, rd[g & 1][k], u0);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (TD == NS) {
        wave_for<0, NS>([&](auto dc) {
          r0 = fw_wror1(r0);
          acc[decltype(dc)::value] *= fma(u1, r0, u0);
        });
      }
      factors(std::integral_constant<int, s1>{});
      if (++cnt == 16) {
        cnt = 0;
        fw_renorm<NS, EXL>(acc, ex, exs, j);
        prodacc_renorm(accS, exS);
      }
    });
  }
}

template <bool CROSS, bool LIN, bool PIVOT = true>
__global__ void __launch_bounds__(64, CROSS ? 2 : 3)
    fmx_estep_wave_kernel(const wave_item* __restrict__ items, int64_t n_items, int64_t c0, int64_t c1,
                          const int32_t* __restrict__ entry_snp, const double* __restrict__ egls,
                          const uint32_t* __restrict__ lin, const int64_t* __restrict__ lin_rank,
                          const fmx_lrec* __restrict__ lrec, const fmx_grec* __restrict__ grec,
                          const double* __restrict__ cgp, const double* __restrict__ cE, int K, int jbase, int kbase,
                          double* __restrict__ fll) {
  static_assert(!(CROSS && LIN), "off-diagonal blocks hold 64 accumulators per lane: no room for a second sweep body");
  constexpr int NS = CROSS ? 64 : 32;
  if ((int64_t)blockIdx.x >= n_items) return;
  const wave_item it = items[blockIdx.x];  // a cell, or a part of a long one (common.hpp)
  if (it.cell < c0 || it.cell >= c1) return;  // not in this rank's cell shard
  const int64_t c = it.slab;  // row of fll: the cell, or an overflow row behind the C cell rows
  const int j = threadIdx.x;
  const int sj = jbase + j;
  const bool live = sj < K, live2 = kbase + j < K;
  const int sjc = live ? sj : K - 1, skc = live2 ? kbase + j : K - 1;  // where a lane without a cluster reads (unused)
  const int npairs = K * (K + 1) / 2;

  // CROSS: 64 accumulators per lane; their integer exponents live in LDS (touched once per 16 entries), 16 KB per wave
  constexpr bool EXL = CROSS || FMX_EXP_LDS;
  __shared__ int32_t exs[EXL ? NS : 1][64];
  __shared__ double ring[3][128];
  double acc[NS], accS = 1.0;
  int32_t ex[EXL ? 1 : NS], exS = 0;
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    acc[t] = 1.0;
    if (EXL) exs[t][j] = 0;
    else ex[t] = 0;
  }
  int cnt = 0;
  if constexpr (LIN) {
    // the part's place in the two streams: linear entries before e = lin_rank[e / 32] + set bits below e in its word
    auto rank = [&](int64_t e) {
      int64_t r = lin_rank[e >> 5];
      if (e & 31) r += __popc(lin[e >> 5] & ((1u << (e & 31)) - 1u));
      return r;
    };
    const int64_t l0 = rank(it.e0), l1 = rank(it.e1);
    fw_walk_lin<NS, EXL>(l0, l1, lrec, cE, K, sjc, live, j, ring, exs, acc, ex, accS, exS, cnt);
    fw_walk_gen<false, true, NS, EXL, PIVOT>(it.e0 - l0, it.e1 - l1, grec, entry_snp, egls, cgp, K * 3, sjc * 3, 0, live, false, j,
                                      ring, exs, acc, ex, accS, exS, cnt);
  } else {
    fw_walk_gen<CROSS, false, NS, EXL, PIVOT>(it.e0, it.e1, grec, entry_snp, egls, cgp, K * 3, sjc * 3, skc * 3, live, live2, j,
                                       ring, exs, acc, ex, accS, exS, cnt);
  }

  double* out = fll + (size_t)c * npairs;
  int kk = CROSS ? ((j + 1) & 63) : j;  // lane whose posterior this lane holds, followed through the rotations
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    kk = __builtin_amdgcn_mov_dpp(kk, 0x13C, 0xF, 0xF, false);
    const int sk = kbase + kk;
    if (CROSS) {
      if (live && sk < K) out[sj * (sj + 1) / 2 + sk] = prodacc_log(acc[t], exs[t][j]);  // jbase > kbase: sj > sk
    } else if (live && sk < K && kk != j && (t < 31 || j > kk)) {  // step 32 of 64 lanes meets every pair twice: one writer
      const int hi = sj > sk ? sj : sk, lo = sj > sk ? sk : sj;
      out[hi * (hi + 1) / 2 + lo] = prodacc_log(acc[t], EXL ? exs[t][j] : ex[EXL ? 0 : t]);
    }
  }
  if (!CROSS && live) out[sj * (sj + 1) / 2 + sj] = prodacc_log(accS, exS);
}

// ---- the streams ----
__global__ void __launch_bounds__(256) fw_popc_kernel(int64_t nwords, int64_t nnz, const uint32_t* __restrict__ lin, int64_t* __restrict__ cnt) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w > nwords) return;
  uint32_t v = 0;
  if (w < nwords) {
    v = lin[w];
    const int64_t left = nnz - w * 32;
    if (left < 32) v &= (1u << left) - 1u;
  }
  cnt[w] = __popc(v);
}

__global__ void __launch_bounds__(256)
    fw_stream_kernel(int64_t nnz, const uint32_t* __restrict__ lin, const int64_t* __restrict__ rank, const int32_t* __restrict__ entry_snp,
                     const double* __restrict__ egls, fmx_lrec* __restrict__ lrec, fmx_grec* __restrict__ grec) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnz) return;
  const uint32_t w = lin[e >> 5];
  const int64_t r = rank[e >> 5] + __popc(w & ((1u << (e & 31)) - 1u));
  if ((w >> (e & 31)) & 1u) {
    const double q0 = egls[(size_t)e * 9], q1 = egls[(size_t)e * 9 + 1];
    lrec[r] = fmx_lrec{q0, q1 - q0, entry_snp[e], 0};
  } else {
    grec[e - r] = fmx_grec{e, entry_snp[e], 0};
  }
}

__global__ void __launch_bounds__(256) fw_ce_kernel(int64_t n, const double* __restrict__ cgp, double* __restrict__ cE) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cE[i] = fma(2.0, cgp[3 * i + 2], cgp[3 * i + 1]);
}

// rows of the parts of a cut cell added, in entry order, into the cell's row of fll
__global__ void __launch_bounds__(256)
    fmx_wave_combine_kernel(const wave_cut* __restrict__ cuts, int64_t c0, int64_t c1, int npairs, double* __restrict__ fll) {
  const wave_cut cu = cuts[blockIdx.x];
  if (cu.cell < c0 || cu.cell >= c1) return;
  double* dst = fll + (size_t)cu.cell * npairs;
  for (int i = threadIdx.x; i < npairs; i += blockDim.x) {
    double v = dst[i];
    for (int64_t q = 0; q < cu.count; ++q) v += fll[(size_t)(cu.first + q) * npairs + i];
    dst[i] = v;
  }
}

}  // namespace

// rows of d_fll the wave E-step needs: one per cell plus one per extra part of a long cell
int64_t fmx_wave_fll_rows(const muxgl_handle* h) {
  const wave_item* items;
  const wave_cut* cuts;
  int64_t n_items, n_cuts, n_over = 0;
  if (demux_wave_items(h, &items, &n_items, &cuts, &n_cuts, &n_over)) return h->C;
  return h->C + n_over;
}

void fmx_wave_streams_release(muxgl_handle* h) {
  for (muxgl_row_state* st : {h->qrow, h->fqrow})  // the quad E-step's partitioned copies are made of the same data
    if (st) {
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
    }
  dev_free(&h->d_flin_rank);
  dev_free(&h->d_lrec);
  dev_free(&h->d_grec);
  dev_free(&h->d_cE);
  h->n_lrec = -1;
  h->cE_n = 0;
}

// the two entry streams of the linear-entry kernel, made once per fmx_prepare from d_flin and d_egls
static int fmx_wave_streams_build(muxgl_handle* h) {
  if (h->n_lrec >= 0) return 0;
  const int64_t nnz = h->nnz, nwords = (nnz + 31) / 32;
  if (dev_alloc(h, &h->d_flin_rank, (size_t)nwords + 1)) return 1;
  hipLaunchKernelGGL(fw_popc_kernel, dim3((unsigned)((nwords + 256) / 256)), dim3(256), 0, h->stream, nwords, nnz, h->d_flin,
                     h->d_flin_rank);
  size_t tb = 0;
  void* tmp = nullptr;
  HIPCHK(h, rocprim::exclusive_scan(nullptr, tb, h->d_flin_rank, h->d_flin_rank, (int64_t)0, (size_t)nwords + 1,
                                    rocprim::plus<int64_t>(), h->stream));
  HIPCHK(h, dev_malloc_retry((void**)&tmp, tb ? tb : 1));
  hipError_t e = rocprim::exclusive_scan(tmp, tb, h->d_flin_rank, h->d_flin_rank, (int64_t)0, (size_t)nwords + 1,
                                         rocprim::plus<int64_t>(), h->stream);
  int64_t n_lin = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&n_lin, h->d_flin_rank + nwords, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(tmp);
  if (e != hipSuccess) MUXGL_FAIL(h, "fmx_wave_streams_build: %s", hipGetErrorString(e));
  if (dev_alloc(h, &h->d_lrec, (size_t)n_lin) || dev_alloc(h, &h->d_grec, (size_t)(nnz - n_lin))) return 1;
  hipLaunchKernelGGL(fw_stream_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, h->stream, nnz, h->d_flin,
                     h->d_flin_rank, h->d_entry_snp, h->d_egls, h->d_lrec, h->d_grec);
  HIPCHK(h, hipGetLastError());
  h->n_lrec = n_lin;
  return 0;
}

// returns -1 when this path does not apply, 0 ok, 1 error
int fmx_wave_estep_launch(muxgl_handle* h, int64_t c0, int64_t nc) {
  if (h->K <= 32 || (h->flags & MUXGL_FLAG_FORCE_TILE_SWEEP)) return -1;  // (up to 32 clusters: fmx_row2.hip, fmx_oct.hip)
  const wave_item* items;
  const wave_cut* cuts;
  int64_t n_items, n_cuts, n_over;
  if (demux_wave_items(h, &items, &n_items, &cuts, &n_cuts, &n_over) || n_items == 0) return -1;
  const int nblk = (h->K + 63) / 64;
  const bool use_lin = h->d_flin && !(h->flags & MUXGL_FLAG_NO_LINEAR_ENTRIES);  // linear-entry stream
  if (use_lin) {
    if (fmx_wave_streams_build(h)) return 1;
    const int64_t n = h->S * (int64_t)h->K;
    if (h->cE_n != n) {
      if (dev_alloc(h, &h->d_cE, (size_t)n)) return 1;
      h->cE_n = n;
    }
    if (n) hipLaunchKernelGGL(fw_ce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, n, h->d_cgp, h->d_cE);
  }
  tic(h, MUXGL_T_FMX_ESTEP_SWEEP);
  for (int X = 0; X < nblk; ++X) {
#define FW_ARGS(XB, YB) \
  items, n_items, c0, c0 + nc, h->d_entry_snp, h->d_egls, h->d_flin, h->d_flin_rank, h->d_lrec, h->d_grec, h->d_cgp, h->d_cE, h->K, \
      64 * (XB), 64 * (YB), h->d_fll
    const bool piv = !(h->flags & MUXGL_FLAG_NO_PIVOT_SUMS);  // (the three-term sums: lets tests compare the two forms)
    if (use_lin && piv) hipLaunchKernelGGL((fmx_estep_wave_kernel<false, true, true>), dim3((unsigned)n_items), dim3(64), 0, h->stream, FW_ARGS(X, X));
    else if (use_lin) hipLaunchKernelGGL((fmx_estep_wave_kernel<false, true, false>), dim3((unsigned)n_items), dim3(64), 0, h->stream, FW_ARGS(X, X));
    else if (piv) hipLaunchKernelGGL((fmx_estep_wave_kernel<false, false, true>), dim3((unsigned)n_items), dim3(64), 0, h->stream, FW_ARGS(X, X));
    else hipLaunchKernelGGL((fmx_estep_wave_kernel<false, false, false>), dim3((unsigned)n_items), dim3(64), 0, h->stream, FW_ARGS(X, X));
    for (int Y = 0; Y < X; ++Y)  // (off-diagonal blocks hold 64 accumulators per lane: no room for a second sweep body)
      hipLaunchKernelGGL((fmx_estep_wave_kernel<true, false, false>), dim3((unsigned)n_items), dim3(64), 0, h->stream, FW_ARGS(X, Y));
#undef FW_ARGS
  }
  toc(h, MUXGL_T_FMX_ESTEP_SWEEP);
  if (n_cuts)
    hipLaunchKernelGGL(fmx_wave_combine_kernel, dim3((unsigned)n_cuts), dim3(256), 0, h->stream, cuts, c0, c0 + nc,
                       h->K * (h->K + 1) / 2, h->d_fll);
  HIPCHK(h, hipGetLastError());
  return 0;
}
