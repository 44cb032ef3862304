// fmx_greedy.hip -- freemux2's greedy initial clustering (cmd_cram_freemux2.cpp:217-261 with
// calculate_droplet_clust_distance, sc_drop_seq.cpp:544-578, and merge(), sc_drop_seq.h:77-101) on the device.
//
// The algorithm is sequential across cells by construction: cell i (in score order) joins the cluster that maximises
// sum_snp [log lk2 - log lk0] against the pileups built from cells 0..i-1, and is merged into it before cell i+1 is
// looked at.  What is parallel is the inside of one step -- L entries x K clusters independent likelihood terms, then L
// independent merges -- so ONE persistent 1024-thread workgroup walks the cell list (a grid-wide barrier per cell
// would cost as much as the step itself).  Per cell:
//   * stage (thread = entry): SNP id, allele frequency and the entry-only factors w_g = gl_i[g,g] * hwe[g] to LDS
//     (A = w_0 + w_1 + w_2), from registers that were loaded during the previous cell's step;
//   * distance (thread = (cluster j, entry stripe)): one 32-byte gather per term from the SNP-major table
//     diag[snp][j] = {gl_j[0,0], gl_j[1,1], gl_j[2,2], B} with B = sum_g gl_j[g,g] * hwe[g] kept up to date by the
//     merge (B > 0 doubles as "the (cluster, SNP) key exists", sc_drop_seq.cpp:549-550).  The K rows of a SNP are
//     contiguous, eight gathers per thread are in flight.  lk2 = sum_g w_g gl_j[g,g]; the reference's nine-term lk0
//     (:563-568) factorises exactly into A * B (same value up to the rounding of a different association).  Products
//     are kept as (mantissa, exponent) instead of two log's per term; stripes are combined with wave shuffles and one
//     pass through LDS; two log's per cluster;
//   * argmax: strict `>` from cluster 0 (:235-242);
//   * merge (thread = entry): the nine-value state of (SNP, winner) is updated in the reference's operation order
//     (multiply, normalise, clamp at 1e-6, normalise; divisions as reciprocal multiplies), B refreshed, five 16-byte
//     accesses each way.
// The per-(cluster, SNP) states built here are discarded afterwards, exactly as in the reference, which rebuilds the
// cluster pileups from the assignment in ascending cell order (:277-288 -> muxgl_fmx_set_clusters).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <thread>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.hpp"
#include "greedy_exact.hpp"

namespace {

// workgroup barrier that orders LDS traffic only: outstanding global loads / stores stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A decision is a NEAR TIE when the runner-up's score is within eps x max(1, |scores|) of the winner's.  The scores here
// are formed in another association than the reference's (products instead of sums of logs, lk0 as A x B, ratios of
// replayed states): they agree with it to ~1e-13 relative, so a margin beyond eps (default 1e-9) cannot come out the
// other way there; a margin within eps is not decided here but by greedy_exact.hpp in the reference's own arithmetic
// (muxgl_fmx_greedy_init).  Two scores of exactly 0 are the structural tie of clusters that share no SNP with the cell
// (sums over nothing on both sides, also in the reference): the first of them wins there as here, no flag.
__device__ __forceinline__ bool greedy_near_tie(double bs, double ss, double eps) {
  if (!(ss > -HUGE_VAL)) return false;  // a single cluster
  if (bs == 0.0 && ss == 0.0) return false;
  const double scale = fmax(1.0, fmax(fabs(bs), fabs(ss)));
  return !(bs - ss > eps * scale);  // (also true for NaN scores: let the exact path look at them)
}
// The score of a cluster from its two products.  Only the STRUCTURAL zero -- both products empty, i.e. exactly 1 -- may
// come out as 0.0.  A cluster that does share SNPs with the cell and whose score lands within eps of 0 (an entry whose
// reads are all of another allele has uniform likelihoods: lk2 = lk0 up to rounding, so the reference's score is noise of
// either sign, +-1e-16, or exactly 0 where this kernel has -1e-17) gets the smallest positive score instead: in front of
// the structural zeros and within reach of anything near 0, so that the step is a near tie and greedy_exact.hpp decides
// it -- greedy_near_tie's exemption of two zeros then only ever meets structural ones, with no third candidate hiding
// just below them.  (A non-empty product of terms < 1 cannot be exactly 1.)
__device__ __forceinline__ double greedy_score(double m2, int32_t e2, double m0, int32_t e0, double eps) {
  const double sc = prodacc_log(m2, e2) - prodacc_log(m0, e0);
  const bool empty = ldexp(m2, e2) == 1.0 && ldexp(m0, e0) == 1.0;
  return (!empty && fabs(sc) <= fmin(eps, 1e-6)) ? 1e-300 : sc;  // (tests raise eps to 1e300: flags, not scores, change)
}

constexpr double kMinNormGL = 1e-6;  // sc_drop_seq.h:14
constexpr int GT = 1024;             // threads of the persistent workgroup
constexpr int GU = 6;                // gathers a thread has in flight in the distance phase
constexpr int ST = GT;               // entries staged in LDS per pass (one per thread)

__global__ void __launch_bounds__(GT)
    fmx_greedy_kernel(const int64_t* __restrict__ hdr_e0, const int32_t* __restrict__ hdr_len,
                      const int32_t* __restrict__ hdr_cell, int64_t n_order, const int32_t* __restrict__ entry_snp, const double* __restrict__ egls,
                      const double* __restrict__ af, int K, int Kp /* K rounded up to a power of two */,
                      double* diag, double* offd, int32_t* __restrict__ clust, uint8_t* __restrict__ near_flag,
                      const int32_t* __restrict__ forced, double tie_eps, int64_t misdecide) {
  __shared__ int32_t s_snp[ST];
  __shared__ __align__(16) double s_w[ST][4];  // w0, w1, w2, allele frequency
  __shared__ double p_m2[GT], p_m0[GT];
  __shared__ int32_t p_x2[GT], p_x0[GT];
  __shared__ int winner;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int j = t & (Kp - 1);
  const int stripe = t / Kp, nstripes = GT / Kp;
  const bool small = Kp <= 64;                      // a wave holds 64/Kp whole stripes
  const int ngroups = small ? GT / 64 : nstripes;   // partials per cluster after the in-wave step
  // Software pipeline over the cell list, so that a step starts with its inputs already on chip.  Barriers inside a
  // step synchronise LDS only (lds_barrier): a full __syncthreads() drains every outstanding global access, which
  // would put each prefetch back on the critical path; the one full barrier per step is the one that publishes the
  // merged states, and the prefetches ride along with its store drain.
  //   * headers {first entry, length, cell id} of 64 steps sit in one register per lane of every wave (read with a
  //     wave-uniform shuffle), the next 64 are loaded a batch ahead;
  //   * the first ST entries' SNP ids of the next cell are requested at the start of a step, what hangs off them
  //     (allele frequency, the entry's diagonal likelihoods) before the merge.
  int64_t b_e0 = hdr_e0[lane], nb_e0 = hdr_e0[64 + lane];  // the host pads the header arrays to a multiple of 64, +64
  int32_t b_len = hdr_len[lane], nb_len = hdr_len[64 + lane];
  int32_t b_cell = hdr_cell[lane], nb_cell = hdr_cell[64 + lane];
  int64_t e0_n1 = __shfl(b_e0, 0, 64);
  int64_t e1_n1 = e0_n1 + __shfl(b_len, 0, 64);
  int32_t cell_n1 = __shfl(b_cell, 0, 64);
  int32_t pf_snp = 0;
  double pf_a = 0, pf_g0 = 0, pf_g4 = 0, pf_g8 = 0;
  if (e0_n1 + t < e1_n1) {
    pf_snp = entry_snp[e0_n1 + t];
    pf_a = af[pf_snp];
    const double* gl = egls + (size_t)(e0_n1 + t) * 9;
    pf_g0 = gl[0];
    pf_g4 = gl[4];
    pf_g8 = gl[8];
  }
  for (int64_t oi = 0; oi < n_order; ++oi) {
    const int32_t cell = cell_n1;
    const int64_t e0 = e0_n1, e1 = e1_n1;
    {  // header of step oi + 1
      const int l = (int)((oi + 1) & 63);
      if (l == 0) {  // batch boundary: the batch loaded 64 steps ago becomes current, the one after it is requested
        b_e0 = nb_e0;
        b_len = nb_len;
        b_cell = nb_cell;
        const int64_t nb = oi + 1 + 64 + lane;
        nb_e0 = hdr_e0[nb];
        nb_len = hdr_len[nb];
        nb_cell = hdr_cell[nb];
      }
      e0_n1 = __shfl(b_e0, l, 64);
      e1_n1 = e0_n1 + __shfl(b_len, l, 64);
      cell_n1 = __shfl(b_cell, l, 64);
    }
    const bool have_next = oi + 1 < n_order;
    const int32_t nx_snp = (have_next && e0_n1 + t < e1_n1) ? entry_snp[e0_n1 + t] : 0;
    // ---- distance to every cluster
    double m2 = 1.0, m0 = 1.0;
    int32_t x2 = 0, x0 = 0;
    for (int64_t cb = e0; cb < e1; cb += ST) {
      const int n = (int)((e1 - cb < ST) ? (e1 - cb) : ST);
      if (cb != e0) lds_barrier();  // the previous pass has been consumed
      for (int i = t; i < n; i += GT) {  // ST == GT: one entry per thread
        int32_t snp;
        double a, g0, g4, g8;
        if (cb == e0) {
          snp = pf_snp;
          a = pf_a;
          g0 = pf_g0;
          g4 = pf_g4;
          g8 = pf_g8;
        } else {
          const int64_t e = cb + i;
          snp = entry_snp[e];
          a = af[snp];
          const double* gl = egls + (size_t)e * 9;
          g0 = gl[0];
          g4 = gl[4];
          g8 = gl[8];
        }
        s_snp[i] = snp;
        *reinterpret_cast<double4*>(s_w[i]) =
            make_double4(g0 * ((1.0 - a) * (1.0 - a)), g4 * (2.0 * a * (1.0 - a)), g8 * (a * a), a);
      }
      lds_barrier();
      if (j < K) {
        for (int ib = stripe; ib < n; ib += nstripes * GU) {
          double4 d[GU];
#pragma unroll
          for (int u = 0; u < GU; ++u) {
            const int i = ib + u * nstripes;
            d[u] = (i < n) ? *reinterpret_cast<const double4*>(diag + ((size_t)s_snp[i] * K + j) * 4)
                           : make_double4(0, 0, 0, 0);
          }
#pragma unroll
          for (int u = 0; u < GU; ++u) {
            if (d[u].w == 0.0) continue;  // no such (cluster, SNP) yet, or past the end
            const double4 w = *reinterpret_cast<const double4*>(s_w[ib + u * nstripes]);
            m2 *= (w.x * d[u].x + w.y * d[u].y) + w.z * d[u].z;
            m0 *= ((w.x + w.y) + w.z) * d[u].w;
            if ((u & 3) == 3) {  // a term is >= ~1e-30 (clamped likelihoods x HWE priors): four cannot underflow
              prodacc_renorm(m2, x2);
              prodacc_renorm(m0, x0);
            }
          }
        }
      }
    }
    if (have_next && e0_n1 + t < e1_n1) {  // second half of the next cell's prefetch
      pf_snp = nx_snp;
      pf_a = af[nx_snp];
      const double* gl = egls + (size_t)(e0_n1 + t) * 9;
      pf_g0 = gl[0];
      pf_g4 = gl[4];
      pf_g8 = gl[8];
    }
    prodacc_renorm(m2, x2);
    prodacc_renorm(m0, x0);
    if (small) {  // stripes of one wave: lanes Kp apart
      for (int off = Kp; off < 64; off <<= 1) {
        m2 *= __shfl_xor(m2, off, 64);
        m0 *= __shfl_xor(m0, off, 64);
        x2 += __shfl_xor(x2, off, 64);
        x0 += __shfl_xor(x0, off, 64);
      }
      prodacc_renorm(m2, x2);
      prodacc_renorm(m0, x0);
    }
    if (!small || lane < Kp) {
      const int g = small ? wave : stripe;
      p_m2[g * Kp + j] = m2;
      p_m0[g * Kp + j] = m0;
      p_x2[g * Kp + j] = x2;
      p_x0[g * Kp + j] = x0;
    }
    lds_barrier();
    if (t < 64) {  // wave 0: scores of clusters t, t+64, ...; running argmax with strict `>` in cluster order (:233-242)
      double bs = -HUGE_VAL, ss = -HUGE_VAL;  // the lane's best and second-best score
      int best = -1;
      for (int c = t; c < K; c += 64) {
        double a2 = 1.0, a0 = 1.0;
        int32_t b2 = 0, b0 = 0;
        for (int g = 0; g < ngroups; ++g) {  // <= 16 mantissas in [0.5,1): no underflow
          a2 *= p_m2[g * Kp + c];
          a0 *= p_m0[g * Kp + c];
          b2 += p_x2[g * Kp + c];
          b0 += p_x0[g * Kp + c];
        }
        const double sc = greedy_score(a2, b2, a0, b0, tie_eps);
        if (best < 0 || sc > bs) {
          ss = bs;
          bs = sc;
          best = c;
        } else {
          ss = fmax(ss, sc);
        }
      }
      // the first maximum in cluster order == largest value, smallest index among equals; the runner-up's score with it
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double os = __shfl_xor(bs, off, 64), o2 = __shfl_xor(ss, off, 64);
        const int ob = __shfl_xor(best, off, 64);
        if (ob >= 0 && (best < 0 || os > bs || (os == bs && ob < best))) {
          ss = best >= 0 ? fmax(bs, o2) : o2;
          bs = os;
          best = ob;
        } else if (ob >= 0) {
          ss = fmax(ss, os);
        }
      }
      if (t == 0) {
        const int32_t fc = forced[oi];
        bool near = greedy_near_tie(bs, ss, tie_eps);
        if (fc >= 0) {
          best = fc;
          near = false;
        } else if (oi == misdecide) {
          best = (best + 1) % K;
          near = true;
        }
        near_flag[oi] = near ? 1 : 0;
        winner = best;
        clust[cell] = best;
      }
    }
    lds_barrier();
    // ---- merge the cell into the winner (:248-251)
    const int w = winner;
    const bool staged = e1 - e0 <= ST;  // SNP id and allele frequency are still in LDS
    for (int64_t e = e0 + t; e < e1; e += GT) {
      const int32_t snp = staged ? s_snp[e - e0] : entry_snp[e];
      // diag[snp][w] = {g00, g11, g22, B} (what the distance phase gathers), offd[snp][w] = {g01, g02, g10, g12, g20,
      // g21}: 16-byte accesses, five each way.  One CU moves 64 B per clock to and from L2, and that -- not latency --
      // bounds a step, so the tables are as compact as the arithmetic allows.
      double2* dg = reinterpret_cast<double2*>(diag + ((size_t)snp * K + w) * 4);
      double2* od = reinterpret_cast<double2*>(offd + ((size_t)snp * K + w) * 6);
      const double* o = egls + (size_t)e * 9;
      const double a = staged ? s_w[e - e0][3] : af[snp];
      const double2 r0 = dg[0], r1 = dg[1], r2 = od[0], r3 = od[1], r4 = od[2];
      const bool present = r1.y != 0.0;
      double v[9] = {r0.x, r2.x, r2.y, r3.x, r0.y, r3.y, r4.x, r4.y, r1.x};  // gls[g1*3+g2] order
      double tmp = 0;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        v[q] = (present ? v[q] : 1.0) * o[q];
        tmp += v[q];
      }
      double r = 1.0 / tmp;
      tmp = 0;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        v[q] *= r;
        if (v[q] < kMinNormGL) v[q] = kMinNormGL;
        tmp += v[q];
      }
      r = 1.0 / tmp;
#pragma unroll
      for (int q = 0; q < 9; ++q) v[q] *= r;
      const double B = (v[0] * ((1.0 - a) * (1.0 - a)) + v[4] * (2.0 * a * (1.0 - a))) + v[8] * (a * a);
      dg[0] = make_double2(v[0], v[4]);
      dg[1] = make_double2(v[8], B);
      od[0] = make_double2(v[1], v[2]);
      od[1] = make_double2(v[3], v[5]);
      od[2] = make_double2(v[6], v[7]);
    }
    __syncthreads();  // the workgroup's stores are visible to its own later loads (one CU, one L1)
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Batched variant (K <= 64): the same sequential algorithm with NOTHING of a cell's step left on a single compute unit's
// chain of memory round trips (the serial kernel above spends 22 us per cell at K = 16, 58 us at K = 64, in five
// dependent trips to L2 per step).
//
// Cells are taken in batches of GB.  Write w_i for the cluster cell i of a batch joins.  Cell i's distance to cluster c
// is a product over its entries of terms that depend on the state (c, SNP); relative to the state at the START of the
// batch only the states (w_b, SNP) for SNPs of EARLIER batch cells b < i have changed -- a few per cent of cell i's
// K x L terms.  The sequential rule  w_i = argmax_c D_i(c | w_0 .. w_{i-1})  is therefore solved as a FIXPOINT:
//   1. start products: every batch cell's products against the state at the start of the batch, and
//      w := argmax of those (the guess that ignores the batch's own merges);
//   2. ratios: for every entry of a batch cell whose SNP also occurs in an earlier cell of the batch (a "hot" entry; its
//      predecessors are a static chain, built once by a sort of (batch, SNP) keys), and every cluster c some predecessor
//      joined under the current guess: replay those predecessors' merges on top of the start state of (c, SNP) -- in cell
//      order, with the reference's merge() -- and leave term(replayed) / term(start);
//   3. scores = start products x the ratios, per cell the first maximum; where a guess changes, 2. and 3. are taken
//      again, until nothing changes.  By induction over i the fixpoint is the sequential result (cell 0 of the batch has
//      no predecessor, so it is final after the first pass; cell i is final once cells < i are) -- in exact arithmetic:
//      the scores here are start product x term(replayed)/term(start), not the product over the replayed state, so they
//      match the sequential rule up to rounding and a near tie within a few ulp could pick another cluster than the
//      reference's loop (strict '>' keeps the first maximum); tests/test_fmx_gpu.py::test_greedy_init_near_ties holds
//      12 000 low-margin cells against the serial kernel and the CPU restatement.  It is reached after one or two passes
//      except while the first clusters are being seeded;
//   4. apply: the batch's merges into the (cluster, SNP) states, one thread per chain of entries at the same SNP walking
//      it in cell order (the order matters only inside a chain: merge() clamps).
// No state is written while a batch is being decided, so the "snapshot" is simply the table itself.
//
// ONE launch walks all batches (greedy_batches_kernel, below): a batch is a chain of small dependent phases, and what it
// costs is the latency of their dependent trips to memory (~2 us each) -- so everything about a batch that does not
// depend on the states (which entries, SNPs, allele-frequency weights, likelihoods, chains) sits in tables in processing
// order, built once, and is fetched BEFORE the grid barrier that the phase waits on.
#ifndef MUXGL_GREEDY_GB
#define MUXGL_GREEDY_GB 32
#endif
constexpr int GB = MUXGL_GREEDY_GB;  // cells per batch
constexpr int GCH = 64;    // entries per chunk of the distance phase
constexpr int GA_T = 256;  // threads per chunk
constexpr int BT = 512;    // threads of a workgroup of greedy_batches_kernel (256 VGPRs a thread): two chunks side by side
constexpr int GSUB = BT / GA_T;
constexpr int GWAVES = BT / 64;
constexpr int ICAP = 2048;  // incidences of one cell kept in LDS (more: the rest through the global tables)

// "position" p = index of an entry in processing order (cells in score order, a cell's entries in SNP order); an
// "incidence" x = (hot entry, one of its predecessors in the batch), grouped by hot entry, the hot entries by cell.
struct greedy_tabs {
  const int64_t* pos_ptr;      // [n + 1] positions of step cell i
  const int64_t* chunk_first;  // [n + 1] chunks (64 positions of one cell) of step cell i
  const int64_t* chunk_p0;     // per chunk: first position, entries
  const int32_t* chunk_n;
  const int32_t* pos_snp;      // per position: SNP,
  const double4* pos_w;        //   {gl[0,0] hwe0, gl[1,1] hwe1, gl[2,2] hwe2, allele frequency},
  const int64_t* pos_e;        //   entry (egls row),
  const int32_t* pos_cell;     //   step cell,
  const int32_t* pos_q;        //   index in (batch, SNP) order: the members of its chain are neighbours there,
  const uchar2* pos_pl;        //   {place in the chain, length of the chain}
  const uint8_t* srt_cell;     // in (batch, SNP) order: cell inside the batch,
  const int64_t* srt_e;        //   entry
  const int64_t* cinc_ptr;     // [n + 1] incidences of step cell i
  const uchar4* inc_meta;      // per incidence: {cell of the predecessor inside the batch, place in its chain, chain length, -}
  const int32_t* inc_hp;       //   position of the hot entry
  const int64_t* inc_e;        //   entry of the predecessor
  const int32_t* hdr_cell;     // [n] step cell -> cell
  const double* egls;
  double* diag;                // [S][K][4] = {g00, g11, g22, B}
  double* offd;                // [S][K][6]
  int32_t* ic;                 // per incidence beyond ICAP of its cell: cluster its ratio belongs to (-1: none)
  double2* rat;                //   {term2, term0}(replayed) / (start)
  double2* pm;                 // chunk partials of the current batch
  int2* px;
  unsigned long long* guess;   // [GB] {batch + 1, first guess} of the batch's cells
  unsigned long long* passw;   // [GB + 1][GB] {batch + 1, changed << 8 | guess} after each pass of the batch
  unsigned* cflag;             // per pair of chunks of the batch: batch + 1 once its partials are in memory
  int32_t* clust;
  uint8_t* near;               // [n] by step: the decision was a near tie (greedy_near_tie)
  const int32_t* forced;       // [n] by step: >= 0: the cluster this cell joins whatever its scores say (decided by the exact path)
  double tie_eps;
  int64_t misdecide;           // -1, or (tests) the step at which the kernel takes the NEXT cluster and raises the flag
  unsigned* bar;               // [0] arrivals, [1] a workgroup gave up
  int32_t* pass_hist;          // NULL or [GB + 2] (MUXGL_TIMING)
  uint64_t* ticks;
  int64_t n;
  int K, Kp;
};

// ---- tables, built once per run ---------------------------------------------------------------------------------------

// per position: key (batch, SNP) and payload p for the chain sort, SNP, weights, entry, step cell.  One wave per cell.
__global__ void __launch_bounds__(256)
    greedy_pos_kernel(int64_t n, const int64_t* __restrict__ hdr_e0, const int64_t* __restrict__ pos_ptr,
                      const int32_t* __restrict__ entry_snp, const double* __restrict__ egls, const double* __restrict__ af,
                      uint64_t* __restrict__ key, uint32_t* __restrict__ val, int32_t* __restrict__ pos_cell,
                      int32_t* __restrict__ pos_snp, double4* __restrict__ pos_w, int64_t* __restrict__ pos_e) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int64_t p0 = pos_ptr[i], p1 = pos_ptr[i + 1], e0 = hdr_e0[i];
  for (int64_t p = p0 + (threadIdx.x & 63); p < p1; p += 64) {
    const int64_t e = e0 + (p - p0);
    const int32_t snp = entry_snp[e];
    const double a = af[snp];
    const double* gl = egls + (size_t)e * 9;
    key[p] = ((uint64_t)(i / GB) << 32) | (uint32_t)snp;
    val[p] = (uint32_t)p;
    pos_cell[p] = (int32_t)i;
    pos_snp[p] = snp;
    pos_e[p] = e;
    pos_w[p] = make_double4(gl[0] * ((1.0 - a) * (1.0 - a)), gl[4] * (2.0 * a * (1.0 - a)), gl[8] * (a * a), a);
  }
}

// chunks of a cell: 64 consecutive positions
__global__ void __launch_bounds__(256)
    greedy_chunks_kernel(int64_t n, const int64_t* __restrict__ pos_ptr, const int64_t* __restrict__ chunk_first,
                         int64_t* __restrict__ chunk_p0, int32_t* __restrict__ chunk_n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t p0 = pos_ptr[i], p1 = pos_ptr[i + 1];
  int64_t c = chunk_first[i];
  for (int64_t p = p0; p < p1; p += GCH, ++c) {
    chunk_p0[c] = p;
    chunk_n[c] = (int32_t)(p1 - p < GCH ? p1 - p : GCH);
  }
}

// after the stable sort, equal keys are the entries of one batch at one SNP, in cell order -- a chain: per position its
// predecessor, its place in the chain and the chain's length, its index in sorted order; in sorted order (a chain's
// members side by side) the cell inside the batch and the entry
__global__ void __launch_bounds__(256)
    greedy_links_kernel(int64_t P, const uint64_t* __restrict__ skey, const uint32_t* __restrict__ sval,
                        const int32_t* __restrict__ pos_cell, const int64_t* __restrict__ pos_e, int32_t* __restrict__ prev,
                        int32_t* __restrict__ pos_q, uchar2* __restrict__ pos_pl, uint8_t* __restrict__ srt_cell,
                        int64_t* __restrict__ srt_e) {
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < P; q += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t k = skey[q];
    const uint32_t p = sval[q];
    int64_t lo = q, hi = q + 1;
    while (lo > 0 && skey[lo - 1] == k) --lo;
    while (hi < P && skey[hi] == k) ++hi;  // (a chain has at most GB members)
    prev[p] = lo < q ? (int32_t)sval[q - 1] : -1;
    pos_q[p] = (int32_t)q;
    pos_pl[p] = make_uchar2((unsigned char)(q - lo), (unsigned char)(hi - lo));
    srt_cell[q] = (uint8_t)(pos_cell[p] % GB);
    srt_e[q] = pos_e[p];
  }
}

// hot entries (positions with a predecessor in their batch) per cell
__global__ void __launch_bounds__(256)
    greedy_hot_count_kernel(int64_t n, const int64_t* __restrict__ pos_ptr, const int32_t* __restrict__ prev,
                            int64_t* __restrict__ nhot) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  int64_t c = 0;
  for (int64_t p = pos_ptr[i] + (threadIdx.x & 63); p < pos_ptr[i + 1]; p += 64) c += prev[p] >= 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) nhot[i] = c;
}

// hot_pos[hot_ptr[i] ..] = the hot positions of cell i in ascending order; hot_len = number of predecessors of each
__global__ void __launch_bounds__(256)
    greedy_hot_fill_kernel(int64_t n, const int64_t* __restrict__ pos_ptr, const int32_t* __restrict__ prev,
                           const int64_t* __restrict__ hot_ptr, int32_t* __restrict__ hot_pos,
                           int64_t* __restrict__ hot_len) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  int64_t base = hot_ptr[i];
  for (int64_t pb = pos_ptr[i]; pb < pos_ptr[i + 1]; pb += 64) {
    const int64_t p = pb + lane;
    const bool hot = p < pos_ptr[i + 1] && prev[p] >= 0;
    const uint64_t m = __ballot(hot);
    if (hot) {
      const int64_t h = base + __popcll(m & ((1ull << lane) - 1ull));
      hot_pos[h] = (int32_t)p;
      int64_t len = 0;
      for (int32_t q = prev[p]; q >= 0; q = prev[q]) ++len;
      hot_len[h] = len;
    }
    base += __popcll(m);
  }
}

// incidences: for hot entry h, its predecessors in cell order
__global__ void __launch_bounds__(256)
    greedy_inc_fill_kernel(int64_t H, const int32_t* __restrict__ hot_pos, const int32_t* __restrict__ prev,
                           const int64_t* __restrict__ hinc_ptr, const int32_t* __restrict__ pos_cell,
                           const int64_t* __restrict__ pos_e, uchar4* __restrict__ inc_meta, int32_t* __restrict__ inc_hp,
                           int64_t* __restrict__ inc_e) {
  for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < H; h += (int64_t)gridDim.x * blockDim.x) {
    const int64_t xb = hinc_ptr[h], xe = hinc_ptr[h + 1];
    const int32_t hp = hot_pos[h];
    int32_t q = prev[hp];
    for (int64_t x = xe - 1; x >= xb; --x) {
      inc_meta[x] = make_uchar4((unsigned char)(pos_cell[q] % GB), (unsigned char)(x - xb), (unsigned char)(xe - xb), 0);
      inc_hp[x] = hp;
      inc_e[x] = pos_e[q];
      q = prev[q];
    }
  }
}

// incidences of step cell i: [cinc_ptr[i], cinc_ptr[i + 1])
__global__ void __launch_bounds__(256)
    greedy_cinc_kernel(int64_t n1, const int64_t* __restrict__ hot_ptr, const int64_t* __restrict__ hinc_ptr,
                       int64_t* __restrict__ cinc_ptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n1) cinc_ptr[i] = hinc_ptr[hot_ptr[i]];
}

// sc_drop_seq.h:77-101 on a nine-value state, divisions as reciprocal multiplies (as in the serial kernel)
__device__ __forceinline__ void greedy_merge9(double (&v)[9], bool present, const double (&o)[9]) {
  double tmp = 0;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    v[q] = (present ? v[q] : 1.0) * o[q];
    tmp += v[q];
  }
  double r = 1.0 / tmp;
  tmp = 0;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    v[q] *= r;
    if (v[q] < kMinNormGL) v[q] = kMinNormGL;
    tmp += v[q];
  }
  r = 1.0 / tmp;
#pragma unroll
  for (int q = 0; q < 9; ++q) v[q] *= r;
}

__device__ __forceinline__ void greedy_load9(double (&o)[9], const double* __restrict__ src) {
#pragma unroll
  for (int q = 0; q < 9; ++q) o[q] = src[q];
}

// What one workgroup of greedy_batches_kernel writes and another reads later in the same launch (MI355X: eight XCDs with
// private L2s; a compute unit's L1 is never refreshed by another's stores):
//   * the states (xwg_ld / xwg_st): plain 16-byte accesses under the grid barrier's agent-scope release / acquire.
//     (Measured alternative, HISTORY.md: 8-byte relaxed agent-scope accesses and a barrier without fences -- slower,
//     configs[3] 0.154 s against 0.132 s: every read then goes to memory);
//   * chunk partials and guesses (sc1_ld / sc1_st): 8-byte relaxed agent-scope accesses (sc1: the store writes through, the
//     load bypasses L1) handed over point to point -- a tag word stored after the data is drained (chunk partials), or a
//     word that carries tag and value at once (guesses) -- so that their readers need not wait at a grid barrier.
__device__ __forceinline__ double sc1_ld(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sc1_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double2 sc1_ld(const double2* p) { return make_double2(sc1_ld(&p->x), sc1_ld(&p->y)); }
__device__ __forceinline__ double4 sc1_ld(const double4* p) {
  return make_double4(sc1_ld(&p->x), sc1_ld(&p->y), sc1_ld(&p->z), sc1_ld(&p->w));
}
__device__ __forceinline__ int2 sc1_ld(const int2* p) {
  const unsigned long long u =
      __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_int2((int)(uint32_t)u, (int)(uint32_t)(u >> 32));
}
__device__ __forceinline__ void sc1_st(double2* p, double2 v) { sc1_st(&p->x, v.x), sc1_st(&p->y, v.y); }
__device__ __forceinline__ void sc1_st(int2* p, int2 v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)(uint32_t)v.x | ((unsigned long long)(uint32_t)v.y << 32),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class V>
__device__ __forceinline__ V xwg_ld(const V* p) { return *p; }
template <class V>
__device__ __forceinline__ void xwg_st(V* p, V v) { *p = v; }

// the nine values of state (SNP, cluster) out of the two tables; false: no such state yet
__device__ __forceinline__ bool greedy_state(const double* diag, const double* offd, size_t row, double (&v)[9], double& B) {
  const double2* dg = reinterpret_cast<const double2*>(diag + row * 4);
  const double2* od = reinterpret_cast<const double2*>(offd + row * 6);
  const double2 r0 = xwg_ld(dg), r1 = xwg_ld(dg + 1);
  // (the off-diagonal values with the diagonal, not behind the test on it: one trip)
  const double2 r2 = xwg_ld(od), r3 = xwg_ld(od + 1), r4 = xwg_ld(od + 2);
  B = r1.y;
  if (r1.y == 0.0) return false;
  v[0] = r0.x, v[1] = r2.x, v[2] = r2.y, v[3] = r3.x, v[4] = r0.y, v[5] = r3.y, v[6] = r4.x, v[7] = r4.y, v[8] = r1.x;
  return true;
}

__device__ __forceinline__ void greedy_state_store(double* diag, double* offd, size_t row, const double (&v)[9], double B) {
  double2* dg = reinterpret_cast<double2*>(diag + row * 4);
  double2* od = reinterpret_cast<double2*>(offd + row * 6);
  xwg_st(dg, make_double2(v[0], v[4]));
  xwg_st(dg + 1, make_double2(v[8], B));
  xwg_st(od, make_double2(v[1], v[2]));
  xwg_st(od + 1, make_double2(v[3], v[5]));
  xwg_st(od + 2, make_double2(v[6], v[7]));
  // (write-through stores here -- 8-byte atomics or global_store_dwordx4 sc1 -- do not make the barrier's release cheaper:
  //  measured 100 / 133 ms against 97 ms at configs[3])
}

// ---- LDS of a workgroup ------------------------------------------------------------------------------------------------
struct greedy_dist_lds {  // one 256-thread group working on a PAIR of chunks of the distance phase
  int32_t n[2];
  int32_t snp[2][GCH];
  double w[2][GCH][4];
  double2 r_m[2][GA_T];
  int2 r_x[2][GA_T];
};
struct greedy_score_lds {  // a workgroup deciding its cell
  double2 part_m[GWAVES][64];   // per wave: product of its share of the ratios, per cluster
  int2 part_x[GWAVES][64];
};
struct greedy_lds {
  union {
    greedy_dist_lds dist[GSUB];
    greedy_score_lds s;
  };
  double2 own_m[64];  // start products of the workgroup's cell
  int2 own_x[64];
  double2 rat[ICAP];  // the cell's incidences
  int32_t ic[ICAP];
  uchar4 meta[ICAP];
  int32_t g[GB];      // the guesses
  unsigned xv[64];    // what greedy_collect gathered
  int32_t cf[GB + 1]; // chunks of the batch's cells, from the batch's first
  int ok, any;
};

// ---- phase 1: products of one chunk per cluster against the states at the start of the batch -----------------------------
// a group of 256 threads = (cluster, entry stripe) takes two chunks at a time (c, c + 1; `two`: the second exists), so that a
// thread has eight gathers in flight; each chunk keeps its own products.  t = thread in the group
__device__ __forceinline__ void greedy_dist_stage(greedy_dist_lds& L, int t, const greedy_tabs& T, int64_t c, bool two) {
  const int h = t >> 7, tt = t & 127;  // half a group stages a chunk
  if (h == 0 || two) {
    const int64_t p = T.chunk_p0[c + h] + tt;
    const int32_t n = T.chunk_n[c + h];
    if (tt == 0) L.n[h] = n;
    if (tt < n) {
      L.snp[h][tt] = T.pos_snp[p];
      *reinterpret_cast<double4*>(L.w[h][tt]) = T.pos_w[p];
    }
  } else if (tt == 0) {
    L.n[1] = 0;
  }
}

__device__ __forceinline__ void greedy_dist_terms(greedy_dist_lds& L, int t, int K, int Kp, const double* diag0) {
  const int j = t & (Kp - 1), stripe = t / Kp, nstripes = GA_T / Kp;
  double m2[2] = {1.0, 1.0}, m0[2] = {1.0, 1.0};
  int32_t x2[2] = {0, 0}, x0[2] = {0, 0};
  if (j < K) {
    int cnt[2] = {0, 0};
    const int n0 = L.n[0], n1 = L.n[1], nmax = n0 > n1 ? n0 : n1;
    for (int i0 = stripe; i0 < nmax; i0 += 4 * nstripes) {  // eight gathers in flight; a chunk's terms in entry order
      double4 d[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * nstripes;
          d[h][u] = i < (h ? n1 : n0) ? xwg_ld(reinterpret_cast<const double4*>(diag0 + ((size_t)L.snp[h][i] * K + j) * 4))
                                      : make_double4(0, 0, 0, 0);
        }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * nstripes;
          if (i >= (h ? n1 : n0) || d[h][u].w == 0.0) continue;  // no such (cluster, SNP) yet
          const double4 w = *reinterpret_cast<const double4*>(L.w[h][i]);
          m2[h] *= (w.x * d[h][u].x + w.y * d[h][u].y) + w.z * d[h][u].z;
          m0[h] *= ((w.x + w.y) + w.z) * d[h][u].w;
          if (++cnt[h] == 4) {
            cnt[h] = 0;
            prodacc_renorm(m2[h], x2[h]);
            prodacc_renorm(m0[h], x0[h]);
          }
        }
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    prodacc_renorm(m2[h], x2[h]);
    prodacc_renorm(m0[h], x0[h]);
    L.r_m[h][t] = make_double2(m2[h], m0[h]);
    L.r_x[h][t] = make_int2(x2[h], x0[h]);
  }
}

// threads [0, Kp) fold the first chunk's stripes, threads [Kp, 2 Kp) the second's; pm / px: the first chunk's row
__device__ __forceinline__ void greedy_dist_fold(greedy_dist_lds& L, int t, int Kp, bool two, double2* __restrict__ pm,
                                                 int2* __restrict__ px) {
  const int nstripes = GA_T / Kp;
  const int h = t / Kp, j = t - h * Kp;
  if (h == 0 || (h == 1 && two)) {
    double a2 = 1.0, a0 = 1.0;
    int32_t b2 = 0, b0 = 0;
    for (int sidx = 0; sidx < nstripes; ++sidx) {
      const double2 m = L.r_m[h][sidx * Kp + j];
      const int2 x = L.r_x[h][sidx * Kp + j];
      a2 *= m.x;
      a0 *= m.y;
      b2 += x.x;
      b0 += x.y;
      if ((sidx & 7) == 7) {
        prodacc_renorm(a2, b2);
        prodacc_renorm(a0, b0);
      }
    }
    prodacc_renorm(a2, b2);
    prodacc_renorm(a0, b0);
    sc1_st(pm + t, make_double2(a2, a0));  // (row c + 1 follows row c: pm[h Kp + j])
    sc1_st(px + t, make_int2(b2, b0));
  }
}

// ---- phase 2 -------------------------------------------------------------------------------------------------------------
// start products of the workgroup's cell per cluster (chunk partials in entry order): thread jj < Kp = cluster jj; the
// products stay in LDS for the scores, returns log lk2 - log lk0
__device__ __forceinline__ double greedy_start_products(greedy_lds& L, int jj, const greedy_tabs& T, int c0, int c1) {
  const int Kp = T.Kp;
  double a2 = 1.0, a0 = 1.0;
  int32_t b2 = 0, b0 = 0;
  for (int cc = c0; cc < c1; cc += 16) {  // sixteen partials in flight, multiplied in chunk order
    double2 m[16];
    int2 x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (cc + u < c1) {
        m[u] = sc1_ld(T.pm + (size_t)(cc + u) * Kp + jj);
        x[u] = sc1_ld(T.px + (size_t)(cc + u) * Kp + jj);
      }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (cc + u < c1) {
        a2 *= m[u].x;
        a0 *= m[u].y;
        b2 += x[u].x;
        b0 += x[u].y;
      }
      if ((u & 7) == 7 && cc + u < c1) {  // (every eighth chunk from c0)
        prodacc_renorm(a2, b2);
        prodacc_renorm(a0, b0);
      }
    }
  }
  prodacc_renorm(a2, b2);
  prodacc_renorm(a0, b0);
  L.own_m[jj] = make_double2(a2, a0);
  L.own_x[jj] = make_int2(b2, b0);
  return prodacc_log(a2, b2) - prodacc_log(a0, b0);
}

// first maximum in cluster order (:233-242) over the lanes of a wave: largest value, smallest cluster among equals
__device__ __forceinline__ int greedy_wave_argmax(double sc, int c, int K) {
  double bs = sc;
  int best = c < K ? c : -1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double os = __shfl_xor(bs, off, 64);
    const int ob = __shfl_xor(best, off, 64);
    if (ob >= 0 && (best < 0 || os > bs || (os == bs && ob < best))) {
      bs = os;
      best = ob;
    }
  }
  return best;
}

// the same with the runner-up's score: near = the decision is a near tie (every lane of the wave returns the same)
__device__ __forceinline__ int greedy_wave_argmax2(double sc, int c, int K, double eps, bool& near) {
  double bs = c < K ? sc : -HUGE_VAL, ss = -HUGE_VAL;
  int best = c < K ? c : -1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double os = __shfl_xor(bs, off, 64), o2 = __shfl_xor(ss, off, 64);
    const int ob = __shfl_xor(best, off, 64);
    if (ob >= 0 && (best < 0 || os > bs || (os == bs && ob < best))) {
      ss = best >= 0 ? fmax(bs, o2) : o2;  // the loser's best and the other side's runner-up
      bs = os;
      best = ob;
    } else if (ob >= 0) {
      ss = fmax(ss, os);
    }
  }
  near = greedy_near_tie(bs, ss, eps);
  return best;
}

// what an applying thread keeps of "its" position, fetched before the decisions are known
struct greedy_pos_regs {
  int32_t snp, q;       // SNP; index in (batch, SNP) order
  int cell;             // cell inside the batch
  unsigned place, len;  // place in the chain, length of the chain
  uint64_t before;      // cells of the (up to eight) members before it, nearest first, a byte each
  uint64_t behind;      //   ... behind it
  double af;
  double gl[9];         // its likelihoods
  double gl1[9];        // those of the member behind it
};

// what a thread keeps of "its" incidence (the first ICAP... BT of the cell), fetched before the states are known
struct greedy_inc_regs {
  int32_t snp;
  double4 w;      // of the hot entry
  double gl[9];   // of the predecessor
};

// Ratio of incidence x (predecessor x of its hot entry) under the guess L.g[]: only the LAST predecessor that joined a
// given cluster carries that cluster's ratio, and it replays every earlier predecessor of the same cluster before itself.
template <class META>
__device__ __forceinline__ void greedy_ratio(greedy_lds& L, const greedy_tabs& T, int64_t x, int64_t X0, const greedy_inc_regs& R,
                                             META meta) {
  const int K = T.K;
  const uchar4 mx = meta(x);
  const int64_t xb = x - mx.y, xe = xb + mx.z;
  const int c = L.g[mx.x];
  bool active = true;
  for (int64_t y = x + 1; y < xe && active; ++y) active = L.g[meta(y).x] != c;
  int32_t icv = -1;
  double2 rv = make_double2(1.0, 1.0);
  if (active) {
    const double a = R.w.w;
    const double h0 = (1.0 - a) * (1.0 - a), h1 = 2.0 * a * (1.0 - a), h2 = a * a;
    const double wx = R.w.x, wy = R.w.y, wz = R.w.z, A = (wx + wy) + wz;
    double v[9], B0;
    bool present = greedy_state(T.diag, T.offd, (size_t)R.snp * K + c, v, B0);
    const double o2 = present ? (wx * v[0] + wy * v[4]) + wz * v[8] : 1.0, o0 = present ? A * B0 : 1.0;
    for (int64_t y = xb; y < x; ++y) {
      if (L.g[meta(y).x] != c) continue;
      double o[9];
      greedy_load9(o, T.egls + (size_t)T.inc_e[y] * 9);
      greedy_merge9(v, present, o);
      present = true;
    }
    greedy_merge9(v, present, R.gl);
    const double B = (v[0] * h0 + v[4] * h1) + v[8] * h2;
    icv = c;
    rv = make_double2(((wx * v[0] + wy * v[4]) + wz * v[8]) / o2, (A * B) / o0);
  }
  const int64_t k = x - X0;
  if (k < ICAP) {
    L.ic[k] = icv;
    L.rat[k] = rv;
  } else {
    T.ic[x] = icv;
    T.rat[x] = rv;
  }
}

// Scores of the workgroup's cell under the ratios.  The cell's incidences are cut into GWAVES * (64 / Kp) contiguous
// shares; lane (s, c) of wave w walks share w * (64 / Kp) + s and multiplies the ratios that belong to cluster c, in their
// order, into its partial product (a ratio of another cluster counts as 1.0 -- exact -- so nothing diverges; the reads are
// LDS broadcasts); the partials are folded in share order onto the start products.  Returns the first maximum.
__device__ __forceinline__ int greedy_cell_decide(greedy_lds& L, const greedy_tabs& T, int t, int64_t X0, int64_t X1, bool& near) {
  const int K = T.K, Kp = T.Kp, lane = t & 63, wave = t >> 6;
  const int ns = 64 / Kp, c = lane & (Kp - 1), sh = wave * ns + lane / Kp;
  double a2 = 1.0, a0 = 1.0;
  int32_t b2 = 0, b0 = 0;
  const int nx = (int)(X1 - X0), nshares = GWAVES * ns;
  const int share = ((nx + nshares - 1) / nshares + 3) & ~3;  // (a multiple of 4)
  const int k0 = sh * share, k1 = k0 + share < nx ? k0 + share : nx;
  for (int k = k0; k < k1; k += 4) {
    int32_t ci[4];
    double2 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // (-1 beyond the cell's last incidence: see the ratios)
      ci[u] = k + u < ICAP ? L.ic[k + u] : (k + u < nx ? T.ic[X0 + k + u] : -1);
      r[u] = k + u < ICAP ? L.rat[k + u] : (k + u < nx ? T.rat[X0 + k + u] : make_double2(1.0, 1.0));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a2 *= ci[u] == c ? r[u].x : 1.0;
      a0 *= ci[u] == c ? r[u].y : 1.0;
    }
    prodacc_renorm(a2, b2);  // (a ratio lies within 1e-30 .. 1e30)
    prodacc_renorm(a0, b0);
  }
  L.s.part_m[wave][lane] = make_double2(a2, a0);
  L.s.part_x[wave][lane] = make_int2(b2, b0);
  __syncthreads();
  int best = -1;
  if (wave == 0) {
    double sc = 0.0;
    if (lane < K) {
      double2 m = L.own_m[lane];
      int2 e = L.own_x[lane];
      const int nsh = share ? (nx + share - 1) / share : 0;
      for (int q = 0; q < nsh; ++q) {
        const double2 pm = L.s.part_m[q / ns][(q % ns) * Kp + lane];
        const int2 px = L.s.part_x[q / ns][(q % ns) * Kp + lane];
        m.x *= pm.x;
        m.y *= pm.y;
        e.x += px.x;
        e.y += px.y;
        prodacc_renorm(m.x, e.x);
        prodacc_renorm(m.y, e.y);
      }
      sc = greedy_score(m.x, e.x, m.y, e.y, T.tie_eps);
    }
    best = greedy_wave_argmax2(sc, lane, K, T.tie_eps, near);
  }
  return best;  // (wave 0)
}

// ---- point-to-point hand-overs ---------------------------------------------------------------------------------------------
constexpr unsigned GREEDY_SPIN_LIMIT = 1u << 23;  // polls of ~1 us

// every thread of the workgroup calls; lanes l < n of wave 0 wait until words[l]'s upper half is `tag` and leave the lower
// halves in L.xv.  false: gave up (bar[1] raised: everyone leaves, the host reports the failure)
__device__ __forceinline__ bool greedy_collect(greedy_lds& L, const unsigned long long* words, int n, unsigned tag, unsigned* bar) {
  const int t = threadIdx.x;
  if (t < 64) {
    bool ok = true;
    if (t < n) {
      unsigned long long v = 0;
      for (unsigned spins = 0;; ++spins) {
        v = __hip_atomic_load(words + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == tag) break;
        if (spins > GREEDY_SPIN_LIMIT ||
            ((spins & 255) == 255 && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      L.xv[t] = (unsigned)v;
    }
    ok = __all(ok);
    if (t == 0) {
      if (!ok) __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      L.ok = ok;
    }
  }
  __syncthreads();
  return L.ok != 0;
}

// every thread of the workgroup calls; wave 0 waits until flags[f0 .. f1) all carry `tag`
__device__ __forceinline__ bool greedy_await(greedy_lds& L, const unsigned* flags, int f0, int f1, unsigned tag, unsigned* bar) {
  const int t = threadIdx.x;
  if (t < 64) {
    bool ok = true;
    for (int f = f0 + t; f < f1 && ok; f += 64)
      for (unsigned spins = 0; __hip_atomic_load(flags + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag; ++spins) {
        if (spins > GREEDY_SPIN_LIMIT ||
            ((spins & 255) == 255 && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    ok = __all(ok);
    if (t == 0) {
      if (!ok) __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      L.ok = ok;
    }
  }
  __syncthreads();
  return L.ok != 0;
}

// ---- the grid barrier ------------------------------------------------------------------------------------------------------
// every wave drains its stores, the workgroup meets, one lane releases at agent scope (write-back of the XCD's L2),
// arrives on ONE monotonic counter (zeroed by the host before the launch), polls it relaxed, then acquires at agent scope.
// The spins are bounded: a workgroup that gives up raises bar[1], everyone leaves, the host reports the failure.
__device__ __forceinline__ bool greedy_grid_barrier(unsigned* bar, unsigned& epoch, int* s_ok) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ++epoch;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = epoch * gridDim.x;
    bool ok = true;
    for (unsigned spins = 0; __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target;) {
      if (++spins > GREEDY_SPIN_LIMIT ||
          ((spins & 255) == 0 && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *s_ok = ok;
  }
  __syncthreads();
  return *s_ok != 0;
}

// The same barrier by XCD (MI355X: eight XCDs, a private L2 each): a workgroup arrives on its XCD's counter; the XCD's
// last arriver -- every other workgroup of the XCD has drained its stores into the shared L2 by then -- writes that L2
// back ONCE, arrives on the top counter, waits for the other XCDs, acquires and releases its XCD's generation word, on
// which the others wait (and acquire: their L1s).  Eight write-backs instead of one per workgroup.  Which XCD a workgroup
// runs on is read from the hardware (HW_REG_XCC_ID) and the workgroups per XCD are counted at the start of the launch, so
// nothing is assumed about the placement.  bar[]: [0] flat counter (the census), [1] gave up, [16] top, then 16 words
// apart per XCD: [32 + 16 x] workgroups, [160 + 16 x] arrivals, [288 + 16 x] generation.
constexpr int GBAR_WORDS = 416, GBAR_TOP = 16, GBAR_CNT = 32, GBAR_ARR = 160, GBAR_GEN = 288;
struct greedy_xbar {
  unsigned x = 0, mine = 0, nx = 0, epoch = 0;  // this workgroup's XCD, its workgroups, XCDs in use, barriers passed
};

__device__ __forceinline__ bool greedy_xcd_barrier(unsigned* bar, greedy_xbar& X, int* s_ok) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ++X.epoch;
  if (threadIdx.x == 0) {
    bool ok = true;
    auto wait_for = [&](unsigned* word, unsigned target) {
      for (unsigned spins = 0; __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target;) {
        if (++spins > GREEDY_SPIN_LIMIT ||
            ((spins & 255) == 0 && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    };
    const unsigned old = __hip_atomic_fetch_add(bar + GBAR_ARR + 16 * X.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == X.epoch * X.mine) {  // the XCD's last arriver
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(bar + GBAR_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_for(bar + GBAR_TOP, X.epoch * X.nx);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      // the generation word releases this XCD's waiters: only when the other XCDs have arrived -- after a timeout they
      // leave through bar[1] (raised above, polled in wait_for) instead of running on with states nobody wrote back
      if (ok) __hip_atomic_store(bar + GBAR_GEN + 16 * X.x, X.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      wait_for(bar + GBAR_GEN + 16 * X.x, X.epoch);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    *s_ok = ok;
  }
  __syncthreads();
  return *s_ok != 0;
}

// ---- the batches, in ONE launch --------------------------------------------------------------------------------------------
// gridDim.x workgroups (at most one per compute unit, so all are resident; at least 2 GB; 192 by default -- measured best
// at configs[3] and configs[4] with the barrier by XCD) stay on the chip for the whole cell list.
// Workgroup b < GB DECIDES cell b of every batch; the others APPLY the batch's merges; all of them take the chunks of the
// distance phase.  Per batch:
//   (fetched before the grid barrier that ends the previous batch: the chunk pair's SNPs and weights to LDS; deciders:
//    the SNP, weights and predecessor likelihoods of their cell's incidences to registers, the chains' bytes to LDS)
//   phase 1: the chunks' start products against the states -> pm / px (written through), a tag word per chunk pair once
//            they are in memory;
//   phase 2: decider b waits for the tags of ITS cell's chunks, folds them into its cell's start products, publishes its
//            first guess {tag, guess} and collects the other deciders'; then per PASS: the ratios of its cell's incidences
//            under the guesses, its cell's score, its new guess -> {tag, changed, guess}, collected by everyone, until a
//            pass changed no guess; appliers meanwhile fetch what "their" position brings (SNP, weights, likelihoods, the
//            cells of the other members of its chain) and stage the next batch's chunks;
//   phase 3: appliers: the merges under the last pass's guesses;                                                GRID BARRIER
// -- the only all-to-all dependency of a batch is states -> next batch's products, so that is the one grid barrier (greedy_xcd_barrier); the
// other hand-overs are point to point (tags: batch + 1, so a word of an earlier batch never matches).  A decider's state
// reads are complete before it publishes (its guess depends on them), so no merge starts before every decider has read.
__global__ void __launch_bounds__(BT) greedy_batches_kernel(const greedy_tabs* __restrict__ Tp) {
  extern __shared__ __align__(16) unsigned char greedy_lds_raw[];
  greedy_lds& L = *reinterpret_cast<greedy_lds*>(greedy_lds_raw);
  const greedy_tabs& T = *Tp;  // (the table of pointers stays in memory: as an argument by value it spills)
  const int t = threadIdx.x, sub = t / GA_T, ts = t % GA_T;
  const int K = T.K, Kp = T.Kp;
  const bool decider = blockIdx.x < GB;
  const int napply = gridDim.x - GB;  // workgroups that apply
  unsigned epoch = 0;
  [[maybe_unused]] uint64_t tk = wall_clock64(), tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#ifdef MUXGL_GREEDY_TICKS  // (build knob: where a workgroup's time goes, printed under MUXGL_TIMING; the reads cost ~1 us each)
#define GTICK(i) { const uint64_t now = wall_clock64(); tacc[i] += now - tk; tk = now; }
#else
#define GTICK(i)
#endif

  // what is fetched ahead for batch `oi0`
  greedy_inc_regs IR;
  int64_t X0 = 0, X1 = 0, cbase = 0, nch = 0;
  auto meta_at = [&](int64_t x) { return x - X0 < ICAP ? L.meta[x - X0] : T.inc_meta[x]; };
  auto fetch_inc_regs = [&](greedy_inc_regs& R, int64_t x) {
    const int32_t hp = T.inc_hp[x];
    R.snp = T.pos_snp[hp];
    R.w = T.pos_w[hp];
    greedy_load9(R.gl, T.egls + (size_t)T.inc_e[x] * 9);
  };
  auto fetch_pos_regs = [&](greedy_pos_regs& R, int64_t p) {
    const uchar2 pl = T.pos_pl[p];
    R.snp = T.pos_snp[p];
    R.q = T.pos_q[p];
    R.cell = T.pos_cell[p] % GB;
    R.place = pl.x, R.len = pl.y;
    R.af = T.pos_w[p].w;
    greedy_load9(R.gl, T.egls + (size_t)T.pos_e[p] * 9);
    R.before = 0, R.behind = 0;
    if (R.len > 1) {
      const int nbehind = (int)R.len - (int)R.place - 1;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k < (int)R.place) R.before |= (uint64_t)T.srt_cell[R.q - 1 - k] << (8 * k);
        if (k < nbehind) R.behind |= (uint64_t)T.srt_cell[R.q + 1 + k] << (8 * k);
      }
      if (nbehind > 0) greedy_load9(R.gl1, T.egls + (size_t)T.srt_e[R.q + 1] * 9);
    }
  };
  auto fetch_ahead = [&](int64_t oi0) {
    if (oi0 >= T.n) return;
    const int nb = (int)(T.n - oi0 < GB ? T.n - oi0 : GB);
    cbase = T.chunk_first[oi0], nch = T.chunk_first[oi0 + nb] - cbase;
    const int64_t c = 2 * ((int64_t)blockIdx.x * GSUB + sub);
    if (c < nch) greedy_dist_stage(L.dist[sub], ts, T, cbase + c, c + 1 < nch);
    if (decider && (int)blockIdx.x < nb) {
      if (t <= nb) L.cf[t] = (int32_t)(T.chunk_first[oi0 + t] - cbase);
      X0 = T.cinc_ptr[oi0 + blockIdx.x], X1 = T.cinc_ptr[oi0 + blockIdx.x + 1];
      for (int64_t x = X0 + t; x < X1 && x - X0 < ICAP; x += BT) L.meta[x - X0] = T.inc_meta[x];
      if (X0 + t < X1) fetch_inc_regs(IR, X0 + t);
    }
  };
  // census: which XCD this workgroup runs on, how many workgroups each XCD got
  greedy_xbar XB;
  XB.x = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;  // HW_REG_XCC_ID, bits 3:0
  if (t == 0) __hip_atomic_fetch_add(T.bar + GBAR_CNT + 16 * XB.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!greedy_grid_barrier(T.bar, epoch, &L.ok)) return;
  XB.mine = __hip_atomic_load(T.bar + GBAR_CNT + 16 * XB.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int x = 0; x < 8; ++x)
    XB.nx += __hip_atomic_load(T.bar + GBAR_CNT + 16 * x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  fetch_ahead(0);

  for (int64_t oi0 = 0; oi0 < T.n; oi0 += GB) {
    const int nb = (int)(T.n - oi0 < GB ? T.n - oi0 : GB);
    const unsigned tag = (unsigned)(oi0 / GB + 1);  // of this batch's hand-overs
    GTICK(6)
    // ---- phase 1: chunk partials against the states at the start of the batch (first trip: staged before the barrier)
    for (int64_t cb = 2 * (int64_t)blockIdx.x * GSUB; cb < nch; cb += 2 * (int64_t)gridDim.x * GSUB) {
      const int64_t c = cb + 2 * sub;
      const bool two = c + 1 < nch;
      if (cb != 2 * (int64_t)blockIdx.x * GSUB) {
        __syncthreads();  // the group's LDS is free for its next chunks
        if (c < nch) greedy_dist_stage(L.dist[sub], ts, T, cbase + c, two);
      }
      __syncthreads();
      if (c < nch) greedy_dist_terms(L.dist[sub], ts, K, Kp, T.diag);
      __syncthreads();
      if (c < nch) greedy_dist_fold(L.dist[sub], ts, Kp, two, T.pm + (size_t)c * Kp, T.px + (size_t)c * Kp);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows are written through: drained = in memory
      __syncthreads();
      if (c < nch && ts == 0) __hip_atomic_store(T.cflag + c / 2, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    GTICK(0)

    // ---- phase 2
    // appliers: what "their" position p_a brings -- SNP, weights, likelihoods, the cells of its chain's other members (and
    // the likelihoods of the member behind it) -- fetched while the deciders work
    const int64_t p0 = T.pos_ptr[oi0], p1 = T.pos_ptr[oi0 + nb];
    const int64_t p_a = p0 + (int64_t)(blockIdx.x - GB) * BT + t;
    greedy_pos_regs A;
    if (!decider) {
      if (p_a < p1) fetch_pos_regs(A, p_a);
      fetch_ahead(oi0 + GB);  // (the chunk's LDS is free: only deciders use the union's other member)
    }
    const bool deciding = decider && (int)blockIdx.x < nb;
    const int32_t fc = deciding ? T.forced[oi0 + blockIdx.x] : -1;  // (requested here, used behind the waits below)
    if (decider) {
      // the own cell's start products (its chunks' partials: wait for their pairs' tags) and first guess (the one that
      // ignores the batch's own merges); the deciders tell each other their guesses through one 8-byte word each,
      // {tag, guess}: the data is the flag
      const int b = blockIdx.x;
      if (deciding && !greedy_await(L, T.cflag, L.cf[b] / 2, (L.cf[b + 1] + 1) / 2, tag, T.bar)) return;
      GTICK(1)
      if (deciding && t < 64) {
        const double sc = t < Kp && t < K ? greedy_start_products(L, t, T, L.cf[b], L.cf[b + 1]) : 0.0;
        int best = greedy_wave_argmax(sc, t, K);
        if (fc >= 0) best = fc;
        if (t == 0)
          __hip_atomic_store(T.guess + b, ((unsigned long long)tag << 32) | (unsigned)best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!greedy_collect(L, T.guess, nb, tag, T.bar)) return;
      if (t < nb) L.g[t] = (int32_t)L.xv[t];
      __syncthreads();
    }
    GTICK(8)
    int passes = 0;
    for (;;) {
      if (deciding) {
        for (int64_t x = X0 + t; x < X1; x += BT) {
          if (x == X0 + t) {
            greedy_ratio(L, T, x, X0, IR, meta_at);
          } else {
            greedy_inc_regs R2;
            fetch_inc_regs(R2, x);
            greedy_ratio(L, T, x, X0, R2, meta_at);
          }
        }
        if (t < 4) {  // the scores read the clusters four at a time
          const int64_t k = X1 - X0 + t;
          if (k < ((X1 - X0 + 3) & ~(int64_t)3) && k < ICAP) L.ic[k] = -1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // (one workgroup, one L1: ratios beyond ICAP are visible to the loads below)
        GTICK(9)
        bool near = false;
        int best = greedy_cell_decide(L, T, t, X0, X1, near);
        if (t < 64) {
          if (fc >= 0) {
            best = fc;
            near = false;
          } else if (oi0 + blockIdx.x == T.misdecide) {
            best = (best + 1) % K;
            near = true;
          }
          if (t == 0) T.near[oi0 + blockIdx.x] = near ? 1 : 0;  // (the last pass's flag stands)
        }
        if (t == 0)  // {tag, changed, guess} of this pass
          __hip_atomic_store(T.passw + (size_t)passes * GB + blockIdx.x,
                             ((unsigned long long)tag << 32) | (best != L.g[blockIdx.x] ? 0x100u : 0u) | (unsigned)best,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      GTICK(2)
      // everyone reads the pass's guesses (appliers: the last pass's are the decisions)
      if (!greedy_collect(L, T.passw + (size_t)passes * GB, nb, tag, T.bar)) return;
      GTICK(3)
      if (t < 64) {
        const bool chg = t < nb && (L.xv[t] & 0x100u);
        const bool any = __any(chg);
        if (t == 0) L.any = any;
      }
      if (t < nb) L.g[t] = (int32_t)(L.xv[t] & 0xffu);
      __syncthreads();
      ++passes;
      if (!L.any) break;  // (uniform over the grid)
      if (passes > GB) {  // cell i is final after pass i + 1: cannot happen
        if (t == 0) __hip_atomic_store(T.bar + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
    }
    if (T.pass_hist && blockIdx.x == 0 && t == 0) atomicAdd(T.pass_hist + (passes < GB + 1 ? passes : GB + 1), 1);
    GTICK(7)

    // ---- phase 3: the batch's merges.  The members of a chain (the batch's entries at one SNP) that join the same cluster
    // are merged into its state in cell order (merge() clamps), by the thread of the first of them; members that join
    // different clusters touch different states.
    if (blockIdx.x == 0 && t < nb) T.clust[T.hdr_cell[oi0 + t]] = L.g[t];
    if (!decider) {
      for (int64_t p = p_a; p < p1; p += (int64_t)napply * BT) {
        if (p != p_a) fetch_pos_regs(A, p);
        const int w = L.g[A.cell];
        bool first = true;
        for (int k = 0; k < (int)A.place && first; ++k) {
          const int c = k < 8 ? (int)((A.before >> (8 * k)) & 0xff) : (int)T.srt_cell[A.q - 1 - k];
          first = L.g[c] != w;
        }
        if (!first) continue;
        const double a = A.af;
        const double h0 = (1.0 - a) * (1.0 - a), h1 = 2.0 * a * (1.0 - a), h2 = a * a;
        const size_t row = (size_t)A.snp * K + w;
        double v[9], B0;
        bool present = greedy_state(T.diag, T.offd, row, v, B0);
        greedy_merge9(v, present, A.gl);
        const int nbehind = (int)A.len - (int)A.place - 1;
        for (int k = 0; k < nbehind; ++k) {
          const int c = k < 8 ? (int)((A.behind >> (8 * k)) & 0xff) : (int)T.srt_cell[A.q + 1 + k];
          if (L.g[c] != w) continue;
          if (k == 0) {
            greedy_merge9(v, true, A.gl1);
          } else {
            double o[9];
            greedy_load9(o, T.egls + (size_t)T.srt_e[A.q + 1 + k] * 9);
            greedy_merge9(v, true, o);
          }
        }
        const double B = (v[0] * h0 + v[4] * h1) + v[8] * h2;
        greedy_state_store(T.diag, T.offd, row, v, B);
      }
    }
    GTICK(4)
    if (decider) fetch_ahead(oi0 + GB);
    GTICK(5)
    if (!greedy_xcd_barrier(T.bar, XB, &L.ok)) return;
  }
  if (T.pass_hist && t == 0 && (blockIdx.x == GB - 1 || blockIdx.x == GB))
    for (int i = 0; i < 12; ++i) T.ticks[(blockIdx.x == GB ? 12 : 0) + i] = tacc[i];
}

}  // namespace

extern "C" int muxgl_fmx_greedy_init(muxgl_handle* h, int32_t K, const double* scores, double frac_init_clust,
                                     double singlet_score_thres, int32_t* clust_out) {
  if (!h) return 1;
  // the procedure is sequential over ALL cells (cmd_cram_freemux2.cpp:217-261): it runs on one device holding the
  // whole pileup; a multi-device run makes the initial clustering on a one-device handle first (popscle-amd does)
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmx_greedy_init");
  if (h->col) MUXGL_FAIL(h, "muxgl_fmx_greedy_init: needs the whole pileup on one handle (this one holds slabs)");
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->fmx_prepared) MUXGL_FAIL(h, "muxgl_fmx_greedy_init: call muxgl_fmx_prepare first");
  if (K < 1 || K > 255) MUXGL_FAIL(h, "muxgl_fmx_greedy_init: K=%d outside [1,255]", K);
  if ((!scores || !clust_out) && h->C) MUXGL_FAIL(h, "muxgl_fmx_greedy_init: NULL array");
  const int64_t C = h->C, S = h->S;
  host_timer tm;
  // sort: score descending, ties by id descending (sc_drop_seq.h:187-198); then the eligibility rules of :222-223
  struct scored {  // (score, id) side by side: the comparisons do not chase the ids into the score array
    double s;
    int32_t id;
  };
  std::vector<scored> keyed((size_t)C);
  for (int64_t i = 0; i < C; ++i) keyed[(size_t)i] = scored{scores[i], (int32_t)i};
  auto before = [](const scored& lhs, const scored& rhs) {
    const double cmp = lhs.s - rhs.s;
    if (cmp != 0) return cmp > 0;
    return lhs.id > rhs.id;
  };
  bool finite = true;
  for (int64_t i = 0; i < C && finite; ++i) finite = std::isfinite(scores[i]);
  if (finite && C >= (1 << 14)) {
    // a total order (the ids break every tie): any correct sort gives the one permutation, so eight threads sort an eighth
    // each and merge pairwise.  (With a NaN or an infinity among the scores the comparator is not a strict weak order and
    // the result is whatever ONE std::sort makes of it -- that case keeps the single call.)
    constexpr int NT = 8;
    size_t cut[NT + 1];
    for (int k = 0; k <= NT; ++k) cut[k] = (size_t)C * (size_t)k / NT;
    {
      std::vector<std::thread> th;
      for (int k = 0; k < NT; ++k) th.emplace_back([&, k] { std::sort(keyed.begin() + (long)cut[k], keyed.begin() + (long)cut[k + 1], before); });
      for (auto& t : th) t.join();
    }
    for (int w = 1; w < NT; w *= 2) {
      std::vector<std::thread> th;
      for (int k = 0; k + w < NT; k += 2 * w)
        th.emplace_back([&, k, w] {
          std::inplace_merge(keyed.begin() + (long)cut[k], keyed.begin() + (long)cut[k + w],
                             keyed.begin() + (long)cut[std::min(k + 2 * w, NT)], before);
        });
      for (auto& t : th) t.join();
    }
  } else {
    std::sort(keyed.begin(), keyed.end(), before);
  }
  std::vector<int32_t> order((size_t)C);
  for (int64_t i = 0; i < C; ++i) order[(size_t)i] = keyed[(size_t)i].id;
  std::vector<int32_t> todo;
  todo.reserve((size_t)C);
  for (int64_t i = 0; i < C; ++i) {
    const int32_t si = order[(size_t)i];
    if ((double)i > (double)C * frac_init_clust) continue;
    if (scores[si] < singlet_score_thres) continue;
    todo.push_back(si);
  }
  for (int64_t i = 0; i < C; ++i) clust_out[i] = -1;
  if (todo.empty()) return 0;

  int Kp = 1;
  while (Kp < K) Kp <<= 1;
  // step headers in processing order, padded so that the kernel's batch prefetch never reads past the end
  const size_t n = todo.size(), npad = (n + 63) / 64 * 64 + 128;
  std::vector<int64_t> cp((size_t)C + 1);
  HIPCHK(h, hipMemcpy(cp.data(), h->d_cell_ptr, sizeof(int64_t) * (size_t)(C + 1), hipMemcpyDeviceToHost));
  std::vector<int64_t> he0(npad, 0);
  std::vector<int32_t> hlen(npad, 0), hcell(npad, 0);
  for (size_t i = 0; i < n; ++i) {
    const int32_t c = todo[i];
    he0[i] = cp[(size_t)c];
    hlen[i] = (int32_t)(cp[(size_t)c + 1] - cp[(size_t)c]);
    hcell[i] = c;
  }
  int64_t* d_he0 = nullptr;
  int32_t *d_hlen = nullptr, *d_hcell = nullptr, *d_clust = nullptr;
  double *d_diag = nullptr, *d_offd = nullptr;  // [S][K][4], [S][K][6]
  // batched path (K <= 64): chunk tables and chunk partials of the distance kernel, chain tables of the fixpoint.
  // It takes the chain of memory round trips of a step off the critical path for every K it covers (1.x us per cell
  // against 22 us at K = 16 and 58 us at K = 64 for the serial kernel), so it is the default; the serial kernel remains
  // for K > 64, for jobs with 2^31 or more entries to cluster, and behind MUXGL_FLAG_FORCE_TILE_SWEEP.
  int64_t P = 0;
  std::vector<int64_t> pos_ptr(npad + 1, 0);
  for (size_t i = 0; i < npad; ++i) pos_ptr[i + 1] = pos_ptr[i] + (i < n ? hlen[i] : 0);
  P = pos_ptr[n];
  // (the batched kernel keeps 2 GB or more workgroups resident, one per compute unit: a device -- or partition -- with fewer
  //  takes the serial kernel)
  int cus = 0;
  HIPCHK(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
  const bool batched =
      K <= 64 && P > 0 && P < ((int64_t)1 << 31) && !(h->flags & MUXGL_FLAG_FORCE_TILE_SWEEP) && cus >= 2 * GB;
  int64_t *d_chunk_first = nullptr, *d_chunk_p0 = nullptr, *d_pos_e = nullptr, *d_cinc_ptr = nullptr, *d_inc_e = nullptr;
  int32_t *d_chunk_n = nullptr, *d_pos_snp = nullptr, *d_inc_hp = nullptr;
  unsigned long long* d_passw = nullptr;
  unsigned* d_cflag = nullptr;
  double4* d_pos_w = nullptr;
  double2* d_pm = nullptr;
  int2* d_px = nullptr;
  int64_t *d_pos_ptr = nullptr, *d_nhot = nullptr, *d_hot_ptr = nullptr, *d_hot_len = nullptr, *d_hinc_ptr = nullptr;
  uint64_t *d_key = nullptr, *d_key2 = nullptr;
  uint32_t *d_val = nullptr, *d_val2 = nullptr;
  int32_t *d_pos_cell = nullptr, *d_prev = nullptr, *d_pos_q = nullptr, *d_hot_pos = nullptr, *d_ic = nullptr;
  uchar2* d_pos_pl = nullptr;
  uint8_t* d_srt_cell = nullptr;
  int64_t* d_srt_e = nullptr;
  unsigned* d_bar = nullptr;
  unsigned long long* d_guess = nullptr;
  greedy_tabs* d_tabs = nullptr;
  uint64_t* d_ticks = nullptr;
  uchar4* d_inc_meta = nullptr;
  int32_t* d_hist = nullptr;
  double2* d_rat = nullptr;
  void* d_tmp = nullptr;
  std::vector<int64_t> chunk_first;
  int64_t max_batch_chunks = 0;
  if (batched) {
    chunk_first.assign(npad + 1, 0);
    for (size_t i = 0; i < npad; ++i) chunk_first[i + 1] = chunk_first[i] + (i < n ? (hlen[i] + GCH - 1) / GCH : 0);
    for (size_t i = 0; i < n; i += GB) {
      const size_t e = std::min(n, i + GB);
      max_batch_chunks = std::max(max_batch_chunks, chunk_first[e] - chunk_first[i]);
    }
  }
  int rc = 1;
  h->err.clear();
  tm.lap("greedy_init: host sort + step headers");
  uint8_t* d_near = nullptr;    // [npad] by step: near tie (greedy_near_tie)
  int32_t* d_forced = nullptr;  // [npad] by step: cluster decided by the exact path, or -1
  int32_t* d_step = nullptr;    // [C] step index of a cell (exact path, on first use)
  greedy_exact::by_step d_bystep;  // the SNP-major view in (SNP, step) order (exact path, on first use)
  greedy_exact::scratch d_xscr;    // device buffers of the exact path's launches
  bool use_batched = batched;
  int wgs = 0;
  double tie_eps = 1e-9;
  // test hooks, honoured only under MUXGL_TEST_HOOKS=1 (tests/test_fmx_gpu.py): a stray variable of one of these names in
  // a user's environment neither changes a decision nor opens a file
  const bool hooks = getenv("MUXGL_TEST_HOOKS") != nullptr && getenv("MUXGL_TEST_HOOKS")[0] == '1';
  const char* ev = nullptr;
  if (hooks && (ev = getenv("MUXGL_GREEDY_TIE_EPS"))) tie_eps = atof(ev);  // 1e300 sends every step through the exact path
  int64_t misdecide = -1;  // (the rerun path: the kernel decides this step wrongly and flags it)
  if (hooks && (ev = getenv("MUXGL_GREEDY_TEST_MISDECIDE"))) misdecide = atoll(ev);
  FILE* dump = nullptr;    // (the exact path's scores of every step it decides, as {int64 step, double[K]} records)
  if (hooks && (ev = getenv("MUXGL_GREEDY_DUMP_SCORES"))) dump = fopen(ev, "wb");
  h->greedy_near_ties = h->greedy_overruled = 0;
  do {
    if (dev_alloc(h, &d_he0, npad) || dev_alloc(h, &d_hlen, npad) || dev_alloc(h, &d_hcell, npad)) break;
    if (dev_alloc(h, &d_near, npad) || dev_alloc(h, &d_forced, npad)) break;
    if (dev_alloc(h, &d_clust, (size_t)C)) break;
    if (dev_alloc(h, &d_diag, (size_t)S * K * 4) || dev_alloc(h, &d_offd, (size_t)S * K * 6)) break;
    hipError_t e = hipMemcpyAsync(d_he0, he0.data(), sizeof(int64_t) * npad, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_hlen, hlen.data(), sizeof(int32_t) * npad, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_hcell, hcell.data(), sizeof(int32_t) * npad, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_clust, 0xFF, sizeof(int32_t) * (size_t)C, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_diag, 0, sizeof(double) * (size_t)S * K * 4, h->stream);
    if (e == hipSuccess && batched) {
      const size_t nP = (size_t)P, nchunks = (size_t)chunk_first[n];
      if (dev_alloc(h, &d_chunk_first, chunk_first.size()) || dev_alloc(h, &d_chunk_p0, nchunks + 1) ||
          dev_alloc(h, &d_chunk_n, nchunks + 1) || dev_alloc(h, &d_pm, (size_t)max_batch_chunks * Kp) ||
          dev_alloc(h, &d_px, (size_t)max_batch_chunks * Kp) || dev_alloc(h, &d_pos_ptr, npad + 1) || dev_alloc(h, &d_key, nP + 8) ||
          dev_alloc(h, &d_key2, nP) || dev_alloc(h, &d_val, nP) || dev_alloc(h, &d_val2, nP) || dev_alloc(h, &d_pos_cell, nP) ||
          dev_alloc(h, &d_pos_snp, nP) || dev_alloc(h, &d_pos_w, nP) || dev_alloc(h, &d_pos_e, nP) ||
          dev_alloc(h, &d_prev, nP) || dev_alloc(h, &d_pos_pl, nP) || dev_alloc(h, &d_srt_cell, nP + 8) || dev_alloc(h, &d_nhot, npad + 1) ||
          dev_alloc(h, &d_hot_ptr, npad + 1) || dev_alloc(h, &d_cinc_ptr, npad + 1) || dev_alloc(h, &d_passw, (size_t)(GB + 1) * GB) ||
          dev_alloc(h, &d_cflag, (size_t)max_batch_chunks / 2 + 2) ||
          dev_alloc(h, &d_hist, (size_t)GB + 2) || dev_alloc(h, &d_bar, (size_t)GBAR_WORDS) || dev_alloc(h, &d_guess, (size_t)GB) ||
          dev_alloc(h, &d_ticks, (size_t)24) || dev_alloc(h, &d_tabs, (size_t)1))
        break;
      (void)hipMemsetAsync(d_hist, 0, sizeof(int32_t) * (GB + 2), h->stream);
      (void)hipMemsetAsync(d_passw, 0, sizeof(unsigned long long) * (GB + 1) * GB, h->stream);
      (void)hipMemsetAsync(d_cflag, 0, sizeof(unsigned) * ((size_t)max_batch_chunks / 2 + 2), h->stream);
      (void)hipMemsetAsync(d_bar, 0, sizeof(unsigned) * GBAR_WORDS, h->stream);
      (void)hipMemsetAsync(d_guess, 0, sizeof(unsigned long long) * GB, h->stream);
      (void)hipMemsetAsync(d_ticks, 0, sizeof(uint64_t) * 24, h->stream);
      e = hipMemsetAsync(d_offd, 0, sizeof(double) * (size_t)S * K * 6, h->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(d_chunk_first, chunk_first.data(), sizeof(int64_t) * chunk_first.size(), hipMemcpyHostToDevice,
                           h->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(d_pos_ptr, pos_ptr.data(), sizeof(int64_t) * (npad + 1), hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) e = hipMemsetAsync(d_nhot, 0, sizeof(int64_t) * (npad + 1), h->stream);
      if (e != hipSuccess) break;
      if (tm.on) { (void)hipStreamSynchronize(h->stream); tm.lap("greedy_init:   allocations, copies"); }
      // ---- tables in processing order; chains: entries of a batch at the same SNP, linked in cell order (one stable
      //      sort of (batch, SNP))
      const unsigned cblocks = (unsigned)((n + 3) / 4);
      hipLaunchKernelGGL(greedy_pos_kernel, dim3(cblocks), dim3(256), 0, h->stream, (int64_t)n, d_he0, d_pos_ptr,
                         h->d_entry_snp, h->d_egls, h->d_af, d_key, d_val, d_pos_cell, d_pos_snp, d_pos_w, d_pos_e);
      hipLaunchKernelGGL(greedy_chunks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (int64_t)n, d_pos_ptr,
                         d_chunk_first, d_chunk_p0, d_chunk_n);
      if (tm.on) { (void)hipStreamSynchronize(h->stream); tm.lap("greedy_init:   position tables"); }
      unsigned bbits = 1;
      while (bbits < 31 && ((int64_t)1 << bbits) < (int64_t)(n / GB + 1)) ++bbits;
      size_t tmp_bytes = 0;
      e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_key, d_key2, d_val, d_val2, nP, 0u, 32u + bbits, h->stream);
      if (e == hipSuccess) e = dev_malloc_retry((void**)&d_tmp, tmp_bytes ? tmp_bytes : 1);
      if (e == hipSuccess)
        e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_key, d_key2, d_val, d_val2, nP, 0u, 32u + bbits, h->stream);
      if (e != hipSuccess) break;
      if (tm.on) { (void)hipStreamSynchronize(h->stream); tm.lap("greedy_init:   sort"); }
      // (the unsorted keys and payloads are done with: their buffers take the entries and indices in sorted order)
      d_srt_e = reinterpret_cast<int64_t*>(d_key);
      d_pos_q = reinterpret_cast<int32_t*>(d_val);
      const unsigned pblocks = (unsigned)std::min<int64_t>((P + 255) / 256, 16384);
      hipLaunchKernelGGL(greedy_links_kernel, dim3(pblocks), dim3(256), 0, h->stream, P, d_key2, d_val2, d_pos_cell, d_pos_e, d_prev,
                         d_pos_q, d_pos_pl, d_srt_cell, d_srt_e);
      if (tm.on) { (void)hipStreamSynchronize(h->stream); tm.lap("greedy_init:   links"); }
      hipLaunchKernelGGL(greedy_hot_count_kernel, dim3(cblocks), dim3(256), 0, h->stream, (int64_t)n, d_pos_ptr, d_prev,
                         d_nhot);
      auto scan = [&](int64_t* in, int64_t* out, size_t cnt) -> hipError_t {
        size_t tb = 0;
        hipError_t er = rocprim::exclusive_scan(nullptr, tb, in, out, (int64_t)0, cnt, rocprim::plus<int64_t>(), h->stream);
        void* tmp = nullptr;
        if (er == hipSuccess) er = dev_malloc_retry((void**)&tmp, tb ? tb : 1);
        if (er == hipSuccess)
          er = rocprim::exclusive_scan(tmp, tb, in, out, (int64_t)0, cnt, rocprim::plus<int64_t>(), h->stream);
        if (er == hipSuccess) er = hipStreamSynchronize(h->stream);
        if (tmp) (void)hipFree(tmp);
        return er;
      };
      e = scan(d_nhot, d_hot_ptr, npad + 1);  // (nhot[n..] = 0: hot_ptr[n] is the total)
      int64_t H = 0;
      if (e == hipSuccess) e = hipMemcpy(&H, d_hot_ptr + n, sizeof(int64_t), hipMemcpyDeviceToHost);
      if (e != hipSuccess) break;
      dev_free(&d_key2);
      dev_free(&d_val2);
      if (dev_alloc(h, &d_hot_pos, (size_t)H + 1) || dev_alloc(h, &d_hot_len, (size_t)H + 1) ||
          dev_alloc(h, &d_hinc_ptr, (size_t)H + 1))
        break;
      e = hipMemsetAsync(d_hot_len, 0, sizeof(int64_t) * ((size_t)H + 1), h->stream);
      if (e != hipSuccess) break;
      hipLaunchKernelGGL(greedy_hot_fill_kernel, dim3(cblocks), dim3(256), 0, h->stream, (int64_t)n, d_pos_ptr, d_prev,
                         d_hot_ptr, d_hot_pos, d_hot_len);
      e = scan(d_hot_len, d_hinc_ptr, (size_t)H + 1);
      int64_t I = 0;
      if (e == hipSuccess) e = hipMemcpy(&I, d_hinc_ptr + H, sizeof(int64_t), hipMemcpyDeviceToHost);
      if (e != hipSuccess) break;
      if (dev_alloc(h, &d_inc_hp, (size_t)I + 1) || dev_alloc(h, &d_inc_e, (size_t)I + 1) || dev_alloc(h, &d_ic, (size_t)I + 1) ||
          dev_alloc(h, &d_rat, (size_t)I + 1) || dev_alloc(h, &d_inc_meta, (size_t)I + 1))
        break;
      if (H > 0)
        hipLaunchKernelGGL(greedy_inc_fill_kernel, dim3((unsigned)std::min<int64_t>((H + 255) / 256, 16384)), dim3(256), 0,
                           h->stream, H, d_hot_pos, d_prev, d_hinc_ptr, d_pos_cell, d_pos_e, d_inc_meta, d_inc_hp, d_inc_e);
      hipLaunchKernelGGL(greedy_cinc_kernel, dim3((unsigned)((npad + 1 + 255) / 256)), dim3(256), 0, h->stream,
                         (int64_t)(npad + 1), d_hot_ptr, d_hinc_ptr, d_cinc_ptr);
      e = hipGetLastError();
      if (e != hipSuccess) break;
      if (tm.on) (void)hipStreamSynchronize(h->stream);
      tm.lap("greedy_init: tables in processing order (sort, links, hot lists)");
      if (tm.on)
        fprintf(stderr, "[muxgl] greedy_init: %lld positions, %lld hot entries, %lld incidences\n", (long long)P, (long long)H,
                (long long)I);
      greedy_tabs T;
      T.pos_ptr = d_pos_ptr;
      T.chunk_first = d_chunk_first;
      T.chunk_p0 = d_chunk_p0;
      T.chunk_n = d_chunk_n;
      T.pos_snp = d_pos_snp;
      T.pos_w = d_pos_w;
      T.pos_e = d_pos_e;
      T.pos_cell = d_pos_cell;
      T.pos_q = d_pos_q;
      T.pos_pl = d_pos_pl;
      T.srt_cell = d_srt_cell;
      T.srt_e = d_srt_e;
      T.cinc_ptr = d_cinc_ptr;
      T.inc_meta = d_inc_meta;
      T.inc_hp = d_inc_hp;
      T.inc_e = d_inc_e;
      T.hdr_cell = d_hcell;
      T.egls = h->d_egls;
      T.diag = d_diag;
      T.offd = d_offd;
      T.ic = d_ic;
      T.rat = d_rat;
      T.pm = d_pm;
      T.px = d_px;
      T.guess = d_guess;
      T.passw = d_passw;
      T.cflag = d_cflag;
      T.clust = d_clust;
      T.bar = d_bar;
      T.pass_hist = tm.on ? d_hist : nullptr;
      T.ticks = d_ticks;
      T.n = (int64_t)n;
      T.K = (int)K;
      T.Kp = Kp;
      T.near = d_near;
      T.forced = d_forced;
      T.tie_eps = tie_eps;
      T.misdecide = misdecide;
      // one workgroup per compute unit at most, and no more than the device can keep resident at once (the grid barrier
      // spins until every workgroup has arrived): GB of them decide a cell each, the others apply the merges
      wgs = 192;
      if (const char* s = getenv("MUXGL_GREEDY_WGS")) wgs = atoi(s);
      wgs = std::max(2 * GB, std::min(wgs, cus));
      e = hipFuncSetAttribute((const void*)greedy_batches_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)sizeof(greedy_lds));
      int per_cu = 0;
      if (e == hipSuccess)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)greedy_batches_kernel, BT, sizeof(greedy_lds));
      if (e == hipSuccess) e = hipMemcpy(d_tabs, &T, sizeof(T), hipMemcpyHostToDevice);
      if (e != hipSuccess || (int64_t)per_cu * cus < wgs) {
        // (an LDS limit below sizeof(greedy_lds), a partition that cannot hold the workgroups: the serial kernel does it)
        (void)hipGetLastError();
        e = hipSuccess;
        use_batched = false;
        if (tm.on) fprintf(stderr, "[muxgl] greedy_init: the batched kernel cannot be resident here, taking the serial one\n");
      }
    }
    tm.lap("greedy_init: tables ready");
    // ---- run, and let the exact path (greedy_exact.hpp) decide what the kernels flag as near ties.  The kernels'
    //      decisions before the first flagged step are the reference's (their margins exceed what the arithmetic can
    //      differ by); the flagged step is decided in the reference's own arithmetic given those.  If that confirms the
    //      kernel's choice, everything behind it stands as well and the next flagged step is looked at; if not, the
    //      choice is forced and the run repeated -- the steps behind it saw another state WHERE THEY SHARE A SNP with the
    //      overruled cell, or with a cell that does, and so on: those steps ("touched", tracked as a set of SNPs) are left
    //      to the repeated run, while an untouched step read none of the changed states, so its flag and the exact
    //      path's decision for it hold in the repeated run as well and are pinned in the same pass.  A pileup of many
    //      small droplets (sparse overlaps, many noise-level ties) therefore takes a few repeats, not one per overruled
    //      step; a dense one, where everything behind an overruled step is touched, one per overruled step as before.
    std::vector<int32_t> forced(npad, -1);
    std::vector<uint8_t> near(npad, 0);
    std::vector<int32_t> hsnp;     // host copy of entry_snp, fetched at the first overruled step
    std::vector<uint64_t> touched; // one bit per SNP
    int reruns = 0;
    int64_t n_near = 0, n_overruled = 0, n_redecided = 0;
    for (;;) {
      if (e == hipSuccess) e = hipMemcpyAsync(d_forced, forced.data(), sizeof(int32_t) * npad, hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) e = hipMemsetAsync(d_near, 0, npad, h->stream);
      if (e == hipSuccess) e = hipMemsetAsync(d_clust, 0xFF, sizeof(int32_t) * (size_t)C, h->stream);
      if (e == hipSuccess) e = hipMemsetAsync(d_diag, 0, sizeof(double) * (size_t)S * K * 4, h->stream);
      if (e != hipSuccess) break;
      const auto t_pass = std::chrono::steady_clock::now();
      if (use_batched) {
        (void)hipMemsetAsync(d_hist, 0, sizeof(int32_t) * (GB + 2), h->stream);
        (void)hipMemsetAsync(d_passw, 0, sizeof(unsigned long long) * (GB + 1) * GB, h->stream);
        (void)hipMemsetAsync(d_cflag, 0, sizeof(unsigned) * ((size_t)max_batch_chunks / 2 + 2), h->stream);
        (void)hipMemsetAsync(d_bar, 0, sizeof(unsigned) * GBAR_WORDS, h->stream);
        (void)hipMemsetAsync(d_guess, 0, sizeof(unsigned long long) * GB, h->stream);
        (void)hipMemsetAsync(d_ticks, 0, sizeof(uint64_t) * 24, h->stream);
        e = hipMemsetAsync(d_offd, 0, sizeof(double) * (size_t)S * K * 6, h->stream);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(greedy_batches_kernel, dim3((unsigned)wgs), dim3(BT), sizeof(greedy_lds), h->stream,
                           (const greedy_tabs*)d_tabs);
      } else {
        hipLaunchKernelGGL(fmx_greedy_kernel, dim3(1), dim3(GT), 0, h->stream, d_he0, d_hlen, d_hcell, (int64_t)n,
                           h->d_entry_snp, h->d_egls, h->d_af, (int)K, Kp, d_diag, d_offd, d_clust, d_near, d_forced, tie_eps, misdecide);
      }
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(clust_out, d_clust, sizeof(int32_t) * (size_t)C, hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(near.data(), d_near, npad, hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      if (use_batched) {
        unsigned bar[2] = {0, 0};
        if (e == hipSuccess) e = hipMemcpy(bar, d_bar, sizeof(bar), hipMemcpyDeviceToHost);
        if (e != hipSuccess || bar[1]) {
          // a launch failure, or a workgroup that gave up at a barrier (workgroups that are not co-resident after all: a CU
          // mask, another process on the device) or a batch without fixpoint: the serial kernel repeats the run
          if (tm.on)
            fprintf(stderr, "[muxgl] greedy_init: the batched kernel failed (%s), repeating with the serial one\n",
                    e != hipSuccess ? hipGetErrorString(e) : bar[1] == 2 ? "a batch did not reach its fixpoint" : "gave up at a grid barrier");
          (void)hipGetLastError();
          e = hipDeviceSynchronize();
          use_batched = false;
          if (e != hipSuccess) break;
          continue;
        }
      }
      if (e != hipSuccess) break;
      const double pass_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pass).count();
      tm.lap(use_batched ? "greedy_init: batches drained" : "greedy_init: serial kernel drained");
      // the flagged steps of this run that are not pinned yet, decided together (each against the run's decisions before it)
      std::vector<greedy_exact::step_req> rq;
      std::vector<size_t> rq_step;
      for (size_t i = 0; i < n; ++i)
        if (near[i] && forced[i] < 0) {
          rq.push_back(greedy_exact::step_req{he0[i], 0, hlen[i], (int32_t)i});
          rq_step.push_back(i);
        }
      std::vector<int32_t> win;
      std::vector<double> exact_scores;
      if (!rq.empty()) {
        if (!d_step) {  // step index of every cell and the SNP-major view in (SNP, step) order, once
          if (!h->d_snp_ptr && plan_build_snp_major(h)) { e = hipErrorUnknown; break; }
          std::vector<int32_t> step((size_t)C, 0x7fffffff);
          for (size_t k = 0; k < n; ++k) step[(size_t)todo[k]] = (int32_t)k;
          if (dev_alloc(h, &d_step, (size_t)C)) { e = hipErrorOutOfMemory; break; }
          e = hipMemcpy(d_step, step.data(), sizeof(int32_t) * (size_t)C, hipMemcpyHostToDevice);
          if (e != hipSuccess) break;
          if (greedy_exact::build_by_step(h, d_step, &d_bystep)) { e = hipErrorUnknown; break; }
        }
        if (greedy_exact::decide_many(h, rq, (int)K, d_bystep, d_clust, win, exact_scores, &d_xscr)) { e = hipErrorUnknown; break; }
      }
      tm.lap("greedy_init: near-tie steps in the reference's arithmetic");
      // The walk over the steps in order, REPAIRING the run as it goes: `touched` holds the SNPs of the cells whose decision
      // changed so far -- the only places where a later step can read a state other than the one this run showed it (a cell
      // that keeps its cluster merges the same likelihoods into the same states as before).  A step that reads none of them
      // stands: its kernel decision, or for a flagged step the exact decision of the batch above.  A step that does is decided
      // again in the reference's arithmetic against the corrected assignments (one small launch: the decisions are
      // sequential from here), flagged or not.  Every decision taken here is the reference's given the ones before it, so
      // when the walk reaches the end the clustering is final and no repeat of the pass is needed -- the case of a pileup
      // of many small droplets, whose overlaps are sparse.  Where almost every later step is touched (an overruled step among
      // large cells) the walk would decide thousands of steps one by one: past a time budget of twice the pass it stops,
      // what it decided stays pinned, and the pass is repeated.
      const auto t_rep = std::chrono::steady_clock::now();
      bool overruled = false, gave_up = false;
      size_t r = 0;  // next request
      // (touched steps are decided LOOK steps at a time -- the next ones that the changes so far touch -- and a result
      //  is used as long as no decision has changed since its launch; after a change the rest of the group is decided anew)
      constexpr size_t LOOK = 64;
      std::vector<greedy_exact::step_req> grp;
      std::vector<size_t> grp_step;
      std::vector<int32_t> gw;
      std::vector<double> gsc;
      size_t gnext = 0;        // next unused result of the group
      bool grp_valid = false;  // no decision has changed since the group was launched
      auto touches = [&](size_t i) {
        const int32_t* sn = hsnp.data() + he0[i];
        for (int32_t k = 0; k < hlen[i]; ++k)
          if ((touched[(size_t)sn[k] >> 6] >> (sn[k] & 63)) & 1u) return true;
        return false;
      };
      for (size_t i = 0; i < n; ++i) {
        const bool flagged = r < rq_step.size() && rq_step[r] == i;
        const size_t ri = r;
        if (flagged) ++r;
        const bool hit = overruled && touches(i);  // does the step read a state that a changed decision has changed?
        if (!flagged && !hit) continue;
        int w;
        const double* wsc;
        if (hit) {
          if (!(grp_valid && gnext < grp_step.size() && grp_step[gnext] == i)) {
            if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_rep).count() > 2.0 * pass_ms + 5.0) {
              gave_up = true;
              break;
            }
            grp.clear();
            grp_step.clear();
            for (size_t j = i; j < n && grp.size() < LOOK; ++j)
              if (j == i || touches(j)) {
                grp.push_back(greedy_exact::step_req{he0[j], 0, hlen[j], (int32_t)j});
                grp_step.push_back(j);
              }
            if (greedy_exact::decide_many(h, grp, (int)K, d_bystep, d_clust, gw, gsc, &d_xscr)) { e = hipErrorUnknown; break; }
            gnext = 0;
            grp_valid = true;
          }
          w = gw[gnext];
          wsc = gsc.data() + gnext * (size_t)K;
          ++gnext;
          ++n_redecided;
        } else {
          w = win[ri];
          wsc = exact_scores.data() + ri * (size_t)K;
        }
        if (flagged) {
          ++n_near;
          if (dump) {
            const int64_t st = (int64_t)i;
            fwrite(&st, sizeof(st), 1, dump);
            fwrite(wsc, sizeof(double), (size_t)K, dump);
          }
        }
        forced[i] = w;  // the reference's decision given the decisions before it: holds in a repeated pass as well
        if (w != clust_out[hcell[i]]) {
          if (flagged) ++n_overruled;
          if (!overruled) {
            overruled = true;
            if (hsnp.empty() && h->nnz) {
              hsnp.resize((size_t)h->nnz);
              e = hipMemcpy(hsnp.data(), h->d_entry_snp, sizeof(int32_t) * (size_t)h->nnz, hipMemcpyDeviceToHost);
              if (e != hipSuccess) break;
            }
            touched.assign((size_t)(S + 63) / 64, 0);
            // Is the walk worth starting?  The cells that cover a SNP of this one bound the steps it will have to decide
            // again; at the pileup's mean cell size and ~20 ns per (entry, cluster) of such a decision (kernel + two glibc
            // logs), a changed decision among large cells comes to many times the pass itself: then the pass is repeated
            // with this decision pinned, as before the walk existed.
            std::vector<int64_t> sp((size_t)S + 1);
            e = hipMemcpy(sp.data(), h->d_snp_ptr, sizeof(int64_t) * (size_t)(S + 1), hipMemcpyDeviceToHost);
            if (e != hipSuccess) break;
            const int32_t* sn0 = hsnp.data() + he0[i];
            double cover = 0;
            for (int32_t k = 0; k < hlen[i]; ++k) cover += (double)(sp[(size_t)sn0[k] + 1] - sp[(size_t)sn0[k]]);
            const double est_ms = std::min(cover, (double)(n - i)) * ((double)h->nnz / (double)std::max<int64_t>(C, 1)) * K * 2e-5;
            if (est_ms > 2.0 * pass_ms + 5.0) {
              gave_up = true;
              break;
            }
          }
          grp_valid = false;  // (results launched before this change may have read what it changes)
          clust_out[hcell[i]] = w;
          const int32_t w32 = w;
          e = hipMemcpy(d_clust + hcell[i], &w32, sizeof(int32_t), hipMemcpyHostToDevice);
          if (e != hipSuccess) break;
          const int32_t* sn = hsnp.data() + he0[i];
          for (int32_t k = 0; k < hlen[i]; ++k) touched[(size_t)sn[k] >> 6] |= (uint64_t)1 << (sn[k] & 63);
        }
      }
      tm.lap("greedy_init: the walk over the steps (decisions behind a changed one taken again)");
      if (e != hipSuccess || !gave_up) break;
      // (every rerun pins one more step for good, so the loop ends after at most n of them; beyond a number no real
      //  pileup has come near, give up loudly rather than take hours)
      if (++reruns > 64 + (int)std::min<size_t>(n, 4096)) {
        h->err = "muxgl_fmx_greedy_init: thousands of near-tie decisions overruled by the exact path";
        e = hipErrorUnknown;
        break;
      }
    }
    h->greedy_near_ties = n_near;
    h->greedy_overruled = n_overruled;
    if (tm.on)
      fprintf(stderr, "[muxgl] greedy_init: %lld near ties decided by the exact path, %lld of them against the kernel's choice; %lld steps behind a changed decision decided again; %d repeats of the pass\n",
              (long long)n_near, (long long)n_overruled, (long long)n_redecided, reruns);
    if (e != hipSuccess) {
      if (h->err.empty()) h->err = std::string("muxgl_fmx_greedy_init: ") + hipGetErrorString(e);
      break;
    }
    if (tm.on && d_hist) {
      int32_t hist[GB + 2];
      if (hipMemcpy(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost) == hipSuccess) {
        fprintf(stderr, "[muxgl] greedy_init: batches by passes of the fixpoint:");
        for (int q = 1; q < GB + 2; ++q)
          if (hist[q]) fprintf(stderr, " %d:%d", q, hist[q]);
        fprintf(stderr, "\n");
      }
      uint64_t tk[24];
      if (d_ticks && hipMemcpy(tk, d_ticks, sizeof(tk), hipMemcpyDeviceToHost) == hipSuccess)
        for (int w = 0; w < 2; ++w)
          fprintf(stderr,
                  "[muxgl] greedy_init: %s workgroup, ms: grid barrier %.1f | chunk products %.1f | wait for the cell's chunks %.1f | "
                  "start scores, first guesses %.1f | passes: ratios %.1f scores %.1f wait for the guesses %.1f + %.1f | merges %.1f | "
                  "fetch ahead %.1f\n",
                  w ? "an applying" : "the last deciding", tk[w * 12 + 6] * 1e-5, tk[w * 12 + 0] * 1e-5, tk[w * 12 + 1] * 1e-5,
                  tk[w * 12 + 8] * 1e-5, tk[w * 12 + 9] * 1e-5, tk[w * 12 + 2] * 1e-5, tk[w * 12 + 3] * 1e-5, tk[w * 12 + 7] * 1e-5,
                  tk[w * 12 + 4] * 1e-5, tk[w * 12 + 5] * 1e-5);
    }
    if (e != hipSuccess) {
      h->err = std::string("muxgl_fmx_greedy_init: ") + hipGetErrorString(e);
      break;
    }
    rc = 0;
  } while (0);
  if (rc && h->err.empty()) h->err = "muxgl_fmx_greedy_init: a device allocation, sort or launch failed";
  dev_free(&d_he0);
  dev_free(&d_hlen);
  dev_free(&d_hcell);
  if (dump) fclose(dump);
  dev_free(&d_near);
  dev_free(&d_forced);
  dev_free(&d_step);
  greedy_exact::release(&d_bystep);
  greedy_exact::release(&d_xscr);
  dev_free(&d_clust);
  dev_free(&d_diag);
  dev_free(&d_offd);
  dev_free(&d_chunk_first);
  dev_free(&d_chunk_p0);
  dev_free(&d_chunk_n);
  dev_free(&d_pm);
  dev_free(&d_px);
  dev_free(&d_pos_ptr);
  dev_free(&d_nhot);
  dev_free(&d_hot_ptr);
  dev_free(&d_hot_len);
  dev_free(&d_hinc_ptr);
  dev_free(&d_cinc_ptr);
  dev_free(&d_key);
  dev_free(&d_key2);
  dev_free(&d_val);
  dev_free(&d_val2);
  dev_free(&d_pos_cell);
  dev_free(&d_pos_snp);
  dev_free(&d_pos_w);
  dev_free(&d_pos_e);
  dev_free(&d_prev);
  dev_free(&d_pos_pl);
  dev_free(&d_srt_cell);
  dev_free(&d_hot_pos);
  dev_free(&d_inc_hp);
  dev_free(&d_inc_e);
  dev_free(&d_inc_meta);
  dev_free(&d_hist);
  dev_free(&d_passw);
  dev_free(&d_cflag);
  dev_free(&d_bar);
  dev_free(&d_guess);
  dev_free(&d_tabs);
  dev_free(&d_ticks);
  dev_free(&d_ic);
  dev_free(&d_rat);
  if (d_tmp) (void)hipFree(d_tmp);
  return rc;
}

extern "C" int muxgl_fmx_greedy_stats(const muxgl_handle* h, int64_t* near_ties, int64_t* overruled) {
  if (!h) return 1;
  if (near_ties) *near_ties = h->greedy_near_ties;
  if (overruled) *overruled = h->greedy_overruled;
  return 0;
}
