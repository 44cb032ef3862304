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
#include <vector>

#include "common.hpp"

namespace {

// workgroup barrier that orders LDS traffic only: outstanding global loads / stores stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr double kMinNormGL = 1e-6;  // sc_drop_seq.h:14
constexpr int GT = 1024;             // threads of the persistent workgroup
constexpr int GU = 6;                // gathers a thread has in flight in the distance phase
constexpr int ST = GT;               // entries staged in LDS per pass (one per thread)

__global__ void __launch_bounds__(GT)
    fmx_greedy_kernel(const int64_t* __restrict__ hdr_e0, const int32_t* __restrict__ hdr_len,
                      const int32_t* __restrict__ hdr_cell, int64_t n_order, const int32_t* __restrict__ entry_snp, const double* __restrict__ egls,
                      const double* __restrict__ af, int K, int Kp /* K rounded up to a power of two */,
                      double* diag, double* offd, int32_t* __restrict__ clust) {
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
      double bs = 0.0;
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
        const double sc = prodacc_log(a2, b2) - prodacc_log(a0, b0);
        if (best < 0 || sc > bs) {
          bs = sc;
          best = c;
        }
      }
      // the first maximum in cluster order == largest value, smallest index among equals
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double os = __shfl_xor(bs, off, 64);
        const int ob = __shfl_xor(best, off, 64);
        if (ob >= 0 && (best < 0 || os > bs || (os == bs && ob < best))) {
          bs = os;
          best = ob;
        }
      }
      if (t == 0) {
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
// Batched variant (K <= 64): the same sequential algorithm, with the bulk of a step taken off the critical path.
//
// Cell i's distance to cluster j is a product over its entries of terms that depend on the state (j, SNP).  Between the
// start of a batch of GB cells and cell i's turn only the states (winner(b), SNP) for SNPs of the earlier cells b of the
// batch have changed -- a few per cent of cell i's K x L terms.  So:
//   * greedy_dist_kernel (whole chip, one workgroup per 64 entries of a batch cell): every batch cell's products
//     against the snapshot `diag0` of the state at the start of the batch;
//   * greedy_batch_kernel (one workgroup, the cells of the batch in order): per cell, every entry looks up which
//     clusters changed at its SNP during this batch (modmask[SNP], one bit per cluster); for those the factor
//     term(current state) / term(snapshot) corrects the snapshot product: the entry's thread gathers the two states
//     and leaves the ratio in LDS (one round per changed cluster of the SNP, almost always one), thread (cluster,
//     stripe) multiplies the ratios of its cluster in a fixed order, and the partials are combined as in the serial
//     kernel (deterministic; no floating-point atomics).  Then argmax, merge into `diag` / `offd` and
//     modmask[SNP] |= winner bit, exactly as in the serial kernel;
//   * greedy_sync_kernel (whole chip): the snapshot is brought up to date for the touched (winner, SNP) states and the
//     masks are cleared.
// The step on the critical path shrinks from K x L gathers (0.5 MB at K = 16, 2 MB at K = 64) to L mask look-ups, the
// corrections and the merge.
constexpr int GB = 32;     // cells per batch
constexpr int GCH = 64;    // entries per workgroup of greedy_dist_kernel
constexpr int GA_T = 256;

__global__ void __launch_bounds__(GA_T)
    greedy_dist_kernel(int64_t chunk0, const int32_t* __restrict__ chunk_cell, const int64_t* __restrict__ chunk_first,
                       const int64_t* __restrict__ hdr_e0, const int32_t* __restrict__ hdr_len,
                       const int32_t* __restrict__ entry_snp, const double* __restrict__ egls,
                       const double* __restrict__ af, int K, int Kp, const double* __restrict__ diag0,
                       double2* __restrict__ pm, int2* __restrict__ px) {
  __shared__ int32_t s_snp[GCH];
  __shared__ __align__(16) double s_w[GCH][4];
  __shared__ double2 r_m[GA_T];
  __shared__ int2 r_x[GA_T];
  const int t = threadIdx.x;
  const int64_t g = chunk0 + blockIdx.x;
  const int oi = chunk_cell[g];
  const int64_t ec = hdr_e0[oi];
  const int64_t eb = ec + (g - chunk_first[oi]) * GCH, ee = ec + hdr_len[oi];
  const int n = (int)(ee - eb < GCH ? ee - eb : GCH);
  if (t < n) {
    const int64_t e = eb + t;
    const int32_t snp = entry_snp[e];
    const double a = af[snp];
    const double* gl = egls + (size_t)e * 9;
    s_snp[t] = snp;
    *reinterpret_cast<double4*>(s_w[t]) =
        make_double4(gl[0] * ((1.0 - a) * (1.0 - a)), gl[4] * (2.0 * a * (1.0 - a)), gl[8] * (a * a), a);
  }
  __syncthreads();
  const int j = t & (Kp - 1), stripe = t / Kp, nstripes = GA_T / Kp;
  double m2 = 1.0, m0 = 1.0;
  int32_t x2 = 0, x0 = 0;
  if (j < K) {
    int cnt = 0;
    for (int i = stripe; i < n; i += nstripes) {
      const double4 d = *reinterpret_cast<const double4*>(diag0 + ((size_t)s_snp[i] * K + j) * 4);
      if (d.w == 0.0) continue;  // no such (cluster, SNP) yet
      const double4 w = *reinterpret_cast<const double4*>(s_w[i]);
      m2 *= (w.x * d.x + w.y * d.y) + w.z * d.z;
      m0 *= ((w.x + w.y) + w.z) * d.w;
      if (++cnt == 4) {
        cnt = 0;
        prodacc_renorm(m2, x2);
        prodacc_renorm(m0, x0);
      }
    }
  }
  prodacc_renorm(m2, x2);
  prodacc_renorm(m0, x0);
  r_m[t] = make_double2(m2, m0);
  r_x[t] = make_int2(x2, x0);
  __syncthreads();
  if (t < Kp) {
    double a2 = 1.0, a0 = 1.0;
    int32_t b2 = 0, b0 = 0;
    for (int sidx = 0; sidx < nstripes; ++sidx) {
      const double2 m = r_m[sidx * Kp + t];
      const int2 x = r_x[sidx * Kp + t];
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
    pm[(size_t)blockIdx.x * Kp + t] = make_double2(a2, a0);
    px[(size_t)blockIdx.x * Kp + t] = make_int2(b2, b0);
  }
}

// snapshot <- current state for the (winner, SNP) states a batch touched, change masks cleared: one workgroup per 64
// entries of a batch cell, after greedy_batch_kernel
__global__ void __launch_bounds__(GCH)
    greedy_sync_kernel(int64_t chunk0, const int32_t* __restrict__ chunk_cell, const int64_t* __restrict__ chunk_first,
                       const int64_t* __restrict__ hdr_e0, const int32_t* __restrict__ hdr_len,
                       const int32_t* __restrict__ hdr_cell, const int32_t* __restrict__ entry_snp,
                       const int32_t* __restrict__ clust, int K, const double* __restrict__ diag,
                       double* __restrict__ diag0, unsigned long long* __restrict__ modmask) {
  const int64_t g = chunk0 + blockIdx.x;
  const int oi = chunk_cell[g];
  const int64_t ec = hdr_e0[oi];
  const int64_t e = ec + (g - chunk_first[oi]) * GCH + threadIdx.x;
  if (e >= ec + hdr_len[oi]) return;
  const int w = clust[hdr_cell[oi]];
  const int32_t snp = entry_snp[e];
  const size_t off = ((size_t)snp * K + w) * 4;
  *reinterpret_cast<double4*>(diag0 + off) = *reinterpret_cast<const double4*>(diag + off);
  modmask[snp] = 0;
}

constexpr int BT = 1024;      // threads of greedy_batch_kernel
constexpr int BE = ST / BT;   // entries per thread and staging pass

__global__ void __launch_bounds__(BT)
    greedy_batch_kernel(int64_t oi0, int nb, const int64_t* __restrict__ chunk_first,
                        const int64_t* __restrict__ hdr_e0, const int32_t* __restrict__ hdr_len,
                        const int32_t* __restrict__ hdr_cell, const int32_t* __restrict__ entry_snp,
                        const double* __restrict__ egls, const double* __restrict__ af, int K, int Kp,
                        const double* __restrict__ diag0, double* diag, double* offd, unsigned long long* modmask,
                        const double2* __restrict__ pm, const int2* __restrict__ px, int32_t* __restrict__ clust) {
  __shared__ double2 s_bm[GB * 64];  // snapshot products {m2, m0} of every batch cell and cluster
  __shared__ int2 s_bx[GB * 64];
  __shared__ int32_t s_snp[ST];
  __shared__ __align__(16) double s_w[ST][4];
  __shared__ unsigned long long s_mask[ST];
  __shared__ double2 s_rat[ST];  // this round's ratio {term2, term0}(current) / (snapshot) of entry i ...
  __shared__ int32_t s_rc[ST];   // ... for this cluster (-1: none)
  __shared__ double p_m2[BT / 64 * 64], p_m0[BT / 64 * 64];
  __shared__ int32_t p_x2[BT / 64 * 64], p_x0[BT / 64 * 64];
  __shared__ int s_maxb;
  __shared__ int winner;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int j = t & (Kp - 1);
  const int stripe = t / Kp, nstripes = BT / Kp;
  const int64_t cbase = chunk_first[oi0];
  // first pass of the first cell: requested before the prologue so that it is on chip when the loop starts
  int64_t e0_n = hdr_e0[oi0], e1_n = e0_n + hdr_len[oi0];
  int32_t pf_snp[BE];
  double pf_a[BE], pf_g0[BE], pf_g4[BE], pf_g8[BE];
#pragma unroll
  for (int u = 0; u < BE; ++u) {
    pf_snp[u] = 0;
    pf_a[u] = pf_g0[u] = pf_g4[u] = pf_g8[u] = 0.0;
    const int64_t e = e0_n + t + u * BT;
    if (e < e1_n) {
      pf_snp[u] = entry_snp[e];
      pf_a[u] = af[pf_snp[u]];
      const double* gl = egls + (size_t)e * 9;
      pf_g0[u] = gl[0];
      pf_g4[u] = gl[4];
      pf_g8[u] = gl[8];
    }
  }
  for (int idx = t; idx < nb * Kp; idx += BT) {  // snapshot products of the batch cells: chunk partials in entry order
    const int b = idx / Kp, jj = idx - b * Kp;
    const int64_t c0 = chunk_first[oi0 + b] - cbase, c1 = chunk_first[oi0 + b + 1] - cbase;
    double a2 = 1.0, a0 = 1.0;
    int32_t b2 = 0, b0 = 0;
    for (int64_t c = c0; c < c1; ++c) {
      const double2 m = pm[(size_t)c * Kp + jj];
      const int2 x = px[(size_t)c * Kp + jj];
      a2 *= m.x;
      a0 *= m.y;
      b2 += x.x;
      b0 += x.y;
      if (((c - c0) & 7) == 7) {
        prodacc_renorm(a2, b2);
        prodacc_renorm(a0, b0);
      }
    }
    prodacc_renorm(a2, b2);
    prodacc_renorm(a0, b0);
    s_bm[b * 64 + jj] = make_double2(a2, a0);
    s_bx[b * 64 + jj] = make_int2(b2, b0);
  }
  if (t == 0) s_maxb = 0;
  __syncthreads();
  for (int b = 0; b < nb; ++b) {
    const int32_t cell = hdr_cell[oi0 + b];
    const int64_t e0 = e0_n, e1 = e1_n;
    const bool have_next = b + 1 < nb;
    if (have_next) {
      e0_n = hdr_e0[oi0 + b + 1];
      e1_n = e0_n + hdr_len[oi0 + b + 1];
    }
    int32_t nx_snp[BE];
#pragma unroll
    for (int u = 0; u < BE; ++u) {
      const int64_t e = e0_n + t + u * BT;
      nx_snp[u] = (have_next && e < e1_n) ? entry_snp[e] : 0;
    }
    const bool staged = e1 - e0 <= ST;  // one pass: what is requested below is still valid at the merge
    double m2 = 1.0, m0 = 1.0;  // ratio products of cluster j over this thread's entry stripe
    int32_t x2 = 0, x0 = 0;
    for (int64_t cb = e0; cb < e1; cb += ST) {
      const int n = (int)((e1 - cb < ST) ? (e1 - cb) : ST);
      if (cb != e0) lds_barrier();  // the previous pass has been consumed
      unsigned long long mask[BE];
      int nbits[BE];
#pragma unroll
      for (int u = 0; u < BE; ++u) {
        const int i = t + u * BT;
        mask[u] = 0;
        if (i < n) {
          int32_t snp;
          double a, g0, g4, g8;
          if (cb == e0) {
            snp = pf_snp[u], a = pf_a[u], g0 = pf_g0[u], g4 = pf_g4[u], g8 = pf_g8[u];
          } else {
            const int64_t e = cb + i;
            snp = entry_snp[e];
            a = af[snp];
            const double* gl = egls + (size_t)e * 9;
            g0 = gl[0], g4 = gl[4], g8 = gl[8];
          }
          mask[u] = modmask[snp];  // clusters whose state at this SNP changed during the batch
          s_snp[i] = snp;
          s_mask[i] = mask[u];
          *reinterpret_cast<double4*>(s_w[i]) =
              make_double4(g0 * ((1.0 - a) * (1.0 - a)), g4 * (2.0 * a * (1.0 - a)), g8 * (a * a), a);
        }
        nbits[u] = __popcll(mask[u]);
      }
      {
        int nbm = 0;
#pragma unroll
        for (int u = 0; u < BE; ++u) nbm = nbits[u] > nbm ? nbits[u] : nbm;
        if (nbm) atomicMax(&s_maxb, nbm);
      }
      lds_barrier();
      const int rounds = s_maxb;
      for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < BE; ++u) {
          const int i = t + u * BT;
          int c = -1;
          if (nbits[u] > r) {
            unsigned long long mm = mask[u];
            for (int q = 0; q < r; ++q) mm &= mm - 1;
            c = __ffsll(mm) - 1;
            const size_t off = ((size_t)s_snp[i] * K + c) * 4;
            const double4 sv = *reinterpret_cast<const double4*>(diag0 + off);
            const double4 cv = *reinterpret_cast<const double4*>(diag + off);
            const double4 w = *reinterpret_cast<const double4*>(s_w[i]);
            const double A = (w.x + w.y) + w.z;
            const double n2 = (w.x * cv.x + w.y * cv.y) + w.z * cv.z, n0 = A * cv.w;
            const bool had = sv.w != 0.0;
            const double o2 = had ? (w.x * sv.x + w.y * sv.y) + w.z * sv.z : 1.0, o0 = had ? A * sv.w : 1.0;
            s_rat[i] = make_double2(n2 / o2, n0 / o0);
          }
          if (i < n) s_rc[i] = c;
        }
        lds_barrier();
        if (j < K) {
          int cnt = 0;
          for (int i = stripe; i < n; i += nstripes) {
            if (s_rc[i] != j) continue;
            const double2 rt = s_rat[i];
            m2 *= rt.x;
            m0 *= rt.y;
            if (++cnt == 4) {  // a ratio lies within 1e-30 .. 1e30
              cnt = 0;
              prodacc_renorm(m2, x2);
              prodacc_renorm(m0, x0);
            }
          }
          prodacc_renorm(m2, x2);
          prodacc_renorm(m0, x0);
        }
        lds_barrier();
      }
      if (t == 0) s_maxb = 0;
    }
#pragma unroll
    for (int u = 0; u < BE; ++u) {  // second half of the next cell's prefetch
      const int64_t e = e0_n + t + u * BT;
      if (have_next && e < e1_n) {
        pf_snp[u] = nx_snp[u];
        pf_a[u] = af[nx_snp[u]];
        const double* gl = egls + (size_t)e * 9;
        pf_g0[u] = gl[0];
        pf_g4[u] = gl[4];
        pf_g8[u] = gl[8];
      }
    }
    // stripes of one wave: lanes Kp apart (Kp <= 64 here); then one partial per wave and cluster
    for (int off = Kp; off < 64; off <<= 1) {
      m2 *= __shfl_xor(m2, off, 64);
      m0 *= __shfl_xor(m0, off, 64);
      x2 += __shfl_xor(x2, off, 64);
      x0 += __shfl_xor(x0, off, 64);
    }
    prodacc_renorm(m2, x2);
    prodacc_renorm(m0, x0);
    if (lane < Kp) {
      p_m2[wave * Kp + j] = m2;
      p_m0[wave * Kp + j] = m0;
      p_x2[wave * Kp + j] = x2;
      p_x0[wave * Kp + j] = x0;
    }
    lds_barrier();
    if (t < 64) {  // scores; first maximum in cluster order (:233-242)
      double bs = 0.0;
      int best = -1;
      if (t < K) {
        const double2 bm = s_bm[b * 64 + t];
        const int2 bx = s_bx[b * 64 + t];
        double a2 = bm.x, a0 = bm.y;
        int32_t b2 = bx.x, b0 = bx.y;
        for (int g = 0; g < BT / 64; ++g) {  // nine mantissas in [0.5,1): no underflow
          a2 *= p_m2[g * Kp + t];
          a0 *= p_m0[g * Kp + t];
          b2 += p_x2[g * Kp + t];
          b0 += p_x0[g * Kp + t];
        }
        bs = prodacc_log(a2, b2) - prodacc_log(a0, b0);
        best = t;
      }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double os = __shfl_xor(bs, off, 64);
        const int ob = __shfl_xor(best, off, 64);
        if (ob >= 0 && (best < 0 || os > bs || (os == bs && ob < best))) {
          bs = os;
          best = ob;
        }
      }
      if (t == 0) {
        winner = best;
        clust[cell] = best;
      }
    }
    lds_barrier();
    const int w = winner;
#pragma unroll
    for (int u = 0; u < BE; ++u) {
      for (int64_t e = e0 + t + u * BT; e < e1; e += ST) {  // merge, as in fmx_greedy_kernel
        const int32_t snp = staged ? s_snp[e - e0] : entry_snp[e];
        double2* dg = reinterpret_cast<double2*>(diag + ((size_t)snp * K + w) * 4);
        double2* od = reinterpret_cast<double2*>(offd + ((size_t)snp * K + w) * 6);
        const double a = staged ? s_w[e - e0][3] : af[snp];
        const double2 r0 = dg[0], r1 = dg[1], r2 = od[0], r3 = od[1], r4 = od[2];
        const double* o = egls + (size_t)e * 9;
        const bool present = r1.y != 0.0;
        double v[9] = {r0.x, r2.x, r2.y, r3.x, r0.y, r3.y, r4.x, r4.y, r1.x};
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
        modmask[snp] = (staged ? s_mask[e - e0] : modmask[snp]) | (1ull << w);  // SNPs are distinct inside a cell
      }
    }
    __syncthreads();  // states and masks are visible to the next cell's look-ups (one CU, one L1)
  }
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
  // sort: score descending, ties by id descending (sc_drop_seq.h:187-198); then the eligibility rules of :222-223
  std::vector<int32_t> order((size_t)C);
  for (int64_t i = 0; i < C; ++i) order[(size_t)i] = (int32_t)i;
  std::sort(order.begin(), order.end(), [&](int32_t lhs, int32_t rhs) {
    const double cmp = scores[lhs] - scores[rhs];
    if (cmp != 0) return cmp > 0;
    return lhs > rhs;
  });
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
  // batched path (K <= 64): snapshot table, change masks, chunk tables and chunk partials
  // the serial kernel's step grows with K (22 us per cell at K = 16, 58 us at K = 64), the batched one's does not
  // (~28 us): measured crossover near K = 24
  const bool batched = K <= 64 && !(h->flags & MUXGL_FLAG_FORCE_TILE_SWEEP) &&
                       (K > 24 || (h->flags & MUXGL_FLAG_FORCE_BATCHED_GREEDY));
  double* d_diag0 = nullptr;
  unsigned long long* d_mask = nullptr;
  int32_t* d_chunk_cell = nullptr;
  int64_t* d_chunk_first = nullptr;
  double2* d_pm = nullptr;
  int2* d_px = nullptr;
  std::vector<int64_t> chunk_first;
  std::vector<int32_t> chunk_cell;
  int64_t max_batch_chunks = 0;
  if (batched) {
    chunk_first.assign(npad + 1, 0);
    for (size_t i = 0; i < npad; ++i) chunk_first[i + 1] = chunk_first[i] + (i < n ? (hlen[i] + GCH - 1) / GCH : 0);
    chunk_cell.resize((size_t)chunk_first[n] + 1);
    for (size_t i = 0; i < n; ++i)
      for (int64_t c = chunk_first[i]; c < chunk_first[i + 1]; ++c) chunk_cell[(size_t)c] = (int32_t)i;
    for (size_t i = 0; i < n; i += GB) {
      const size_t e = std::min(n, i + GB);
      max_batch_chunks = std::max(max_batch_chunks, chunk_first[e] - chunk_first[i]);
    }
  }
  int rc = 1;
  do {
    if (dev_alloc(h, &d_he0, npad) || dev_alloc(h, &d_hlen, npad) || dev_alloc(h, &d_hcell, npad)) break;
    if (dev_alloc(h, &d_clust, (size_t)C)) break;
    if (dev_alloc(h, &d_diag, (size_t)S * K * 4) || dev_alloc(h, &d_offd, (size_t)S * K * 6)) break;
    hipError_t e = hipMemcpyAsync(d_he0, he0.data(), sizeof(int64_t) * npad, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_hlen, hlen.data(), sizeof(int32_t) * npad, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_hcell, hcell.data(), sizeof(int32_t) * npad, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_clust, 0xFF, sizeof(int32_t) * (size_t)C, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_diag, 0, sizeof(double) * (size_t)S * K * 4, h->stream);
    if (e == hipSuccess && !batched) {
      hipLaunchKernelGGL(fmx_greedy_kernel, dim3(1), dim3(GT), 0, h->stream, d_he0, d_hlen, d_hcell, (int64_t)n,
                         h->d_entry_snp, h->d_egls, h->d_af, (int)K, Kp, d_diag, d_offd, d_clust);
      e = hipGetLastError();
    }
    if (e == hipSuccess && batched) {
      if (dev_alloc(h, &d_diag0, (size_t)S * K * 4) || dev_alloc(h, &d_mask, (size_t)S) ||
          dev_alloc(h, &d_chunk_cell, chunk_cell.size()) || dev_alloc(h, &d_chunk_first, chunk_first.size()) ||
          dev_alloc(h, &d_pm, (size_t)max_batch_chunks * Kp) || dev_alloc(h, &d_px, (size_t)max_batch_chunks * Kp))
        break;
      e = hipMemsetAsync(d_diag0, 0, sizeof(double) * (size_t)S * K * 4, h->stream);
      if (e == hipSuccess) e = hipMemsetAsync(d_mask, 0, sizeof(unsigned long long) * (size_t)S, h->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(d_chunk_cell, chunk_cell.data(), sizeof(int32_t) * chunk_cell.size(), hipMemcpyHostToDevice,
                           h->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(d_chunk_first, chunk_first.data(), sizeof(int64_t) * chunk_first.size(), hipMemcpyHostToDevice,
                           h->stream);
      for (size_t i = 0; i < n && e == hipSuccess; i += GB) {
        const int nb = (int)std::min<size_t>(GB, n - i);
        const int64_t nch = chunk_first[i + nb] - chunk_first[i];
        if (nch > 0)
          hipLaunchKernelGGL(greedy_dist_kernel, dim3((unsigned)nch), dim3(GA_T), 0, h->stream, chunk_first[i],
                             d_chunk_cell, d_chunk_first, d_he0, d_hlen, h->d_entry_snp, h->d_egls, h->d_af, (int)K, Kp,
                             d_diag0, d_pm, d_px);
        hipLaunchKernelGGL(greedy_batch_kernel, dim3(1), dim3(BT), 0, h->stream, (int64_t)i, nb, d_chunk_first, d_he0,
                           d_hlen, d_hcell, h->d_entry_snp, h->d_egls, h->d_af, (int)K, Kp, d_diag0, d_diag, d_offd, d_mask,
                           d_pm, d_px, d_clust);
        if (nch > 0)
          hipLaunchKernelGGL(greedy_sync_kernel, dim3((unsigned)nch), dim3(GCH), 0, h->stream, chunk_first[i], d_chunk_cell,
                             d_chunk_first, d_he0, d_hlen, d_hcell, h->d_entry_snp, d_clust, (int)K, d_diag, d_diag0,
                             d_mask);
        e = hipGetLastError();
      }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(clust_out, d_clust, sizeof(int32_t) * (size_t)C, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
      h->err = std::string("muxgl_fmx_greedy_init: ") + hipGetErrorString(e);
      break;
    }
    rc = 0;
  } while (0);
  dev_free(&d_he0);
  dev_free(&d_hlen);
  dev_free(&d_hcell);
  dev_free(&d_clust);
  dev_free(&d_diag);
  dev_free(&d_offd);
  dev_free(&d_diag0);
  dev_free(&d_mask);
  dev_free(&d_chunk_cell);
  dev_free(&d_chunk_first);
  dev_free(&d_pm);
  dev_free(&d_px);
  return rc;
}
