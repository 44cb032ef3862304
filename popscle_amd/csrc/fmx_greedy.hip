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

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

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
// Batched variant (K <= 64): the same sequential algorithm with NOTHING of a cell's step left on a single compute unit's
// chain of memory round trips (the serial kernel above spends 22 us per cell at K = 16, 58 us at K = 64, in five
// dependent trips to L2 per step).
//
// Cells are taken in batches of GB.  Write w_i for the cluster cell i of a batch joins.  Cell i's distance to cluster c
// is a product over its entries of terms that depend on the state (c, SNP); relative to the state at the START of the
// batch only the states (w_b, SNP) for SNPs of EARLIER batch cells b < i have changed -- a few per cent of cell i's
// K x L terms.  The sequential rule  w_i = argmax_c D_i(c | w_0 .. w_{i-1})  is therefore solved as a FIXPOINT:
//   1. greedy_dist_kernel (whole chip): every batch cell's products against the state at the start of the batch;
//      greedy_argmax0_kernel: w := argmax of those (the guess that ignores the batch's own merges);
//   2. greedy_ratio_kernel (whole chip): for every entry of a batch cell whose SNP also occurs in an earlier cell of the
//      batch (a "hot" entry; its predecessors are a static chain, built once by a sort of (batch, SNP) keys), and every
//      cluster c some predecessor joined under the current guess: replay those predecessors' merges on top of the
//      start state of (c, SNP) -- in cell order, with the reference's merge() -- and leave term(replayed) / term(start);
//   3. greedy_decide_kernel (one workgroup): scores = start products x the ratios, per cell the first maximum; where a
//      guess changes, the ratios of the hot entries behind that cell are recomputed and the scores taken again, until
//      nothing changes.  By induction over i the fixpoint is the sequential result (cell 0 of the batch has no
//      predecessor, so it is final after the first pass; cell i is final once cells < i are) -- in exact arithmetic:
//      the scores here are start product x term(replayed)/term(start), not the product over the replayed state, so they
//      match the sequential rule up to rounding and a near tie within a few ulp could pick another cluster than the
//      reference's loop (strict '>' keeps the first maximum); tests/test_fmx_gpu.py::test_greedy_init_near_ties holds
//      12 000 low-margin cells against the serial kernel and the CPU restatement.  It is reached after one or two passes except
//      while the first clusters are being seeded;
//   4. greedy_apply_kernel (whole chip): the batch's merges into the (cluster, SNP) states, one thread per chain of
//      entries at the same SNP walking it in cell order (the order matters only inside a chain: merge() clamps).
// No state is written while a batch is being decided, so the "snapshot" is simply the table itself.
#ifndef MUXGL_GREEDY_GB
#define MUXGL_GREEDY_GB 32
#endif
constexpr int GB = MUXGL_GREEDY_GB;  // cells per batch
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

// ---- chain tables, built once per run -----------------------------------------------------------------------------------
// "position" p = index of an entry in processing order (cells in score order, a cell's entries in SNP order).

// key (batch, SNP) and payload p of every position; pos_cell[p] = step index of its cell.  One wave per cell.
__global__ void __launch_bounds__(256)
    greedy_keys_kernel(int64_t n, const int64_t* __restrict__ hdr_e0, const int64_t* __restrict__ pos_ptr,
                       const int32_t* __restrict__ entry_snp, uint64_t* __restrict__ key, uint32_t* __restrict__ val,
                       int32_t* __restrict__ pos_cell) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int64_t p0 = pos_ptr[i], p1 = pos_ptr[i + 1], e0 = hdr_e0[i];
  for (int64_t p = p0 + (threadIdx.x & 63); p < p1; p += 64) {
    key[p] = ((uint64_t)(i / GB) << 32) | (uint32_t)entry_snp[e0 + (p - p0)];
    val[p] = (uint32_t)p;
    pos_cell[p] = (int32_t)i;
  }
}

// after the stable sort, equal keys are the entries of one batch at one SNP, in cell order: link them
__global__ void __launch_bounds__(256)
    greedy_links_kernel(int64_t P, const uint64_t* __restrict__ skey, const uint32_t* __restrict__ sval,
                        int32_t* __restrict__ prev, int32_t* __restrict__ next) {
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < P; q += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t k = skey[q];
    const uint32_t p = sval[q];
    prev[p] = (q > 0 && skey[q - 1] == k) ? (int32_t)sval[q - 1] : -1;
    next[p] = (q + 1 < P && skey[q + 1] == k) ? (int32_t)sval[q + 1] : -1;
  }
}

// hot entries (positions with a predecessor in their batch) per cell, and the total length of their chains
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

// incidences: for hot entry h, its predecessors in cell order -- their position and the step index of their cell
__global__ void __launch_bounds__(256)
    greedy_inc_fill_kernel(int64_t H, const int32_t* __restrict__ hot_pos, const int32_t* __restrict__ prev,
                           const int64_t* __restrict__ hinc_ptr, const int32_t* __restrict__ pos_cell,
                           int32_t* __restrict__ inc_pos, int32_t* __restrict__ inc_cell, int32_t* __restrict__ inc_hot,
                           uchar4* __restrict__ inc_meta) {
  for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < H; h += (int64_t)gridDim.x * blockDim.x) {
    const int64_t xb = hinc_ptr[h], xe = hinc_ptr[h + 1];
    int32_t q = prev[hot_pos[h]];
    for (int64_t x = xe - 1; x >= xb; --x) {
      inc_pos[x] = q;
      inc_cell[x] = pos_cell[q];
      inc_hot[x] = (int32_t)h;
      // what the decide kernel keeps in LDS: cell of the predecessor inside its batch, place in the chain, chain length
      inc_meta[x] = make_uchar4((unsigned char)(pos_cell[q] % GB), (unsigned char)(x - xb), (unsigned char)(xe - xb), 0);
      q = prev[q];
    }
  }
}

// sc_drop_seq.h:77-101 on a nine-value state, divisions as reciprocal multiplies (as in the serial kernel)
__device__ __forceinline__ void greedy_merge9(double (&v)[9], bool present, const double* __restrict__ o) {
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

struct greedy_tabs {
  const int64_t* hdr_e0;
  const int64_t* pos_ptr;
  const int32_t* pos_cell;
  const int32_t* hot_pos;
  const int64_t* hinc_ptr;
  const int32_t* inc_pos;
  const int32_t* inc_cell;
  const int32_t* inc_hot;
  const uchar4* inc_meta;  // {cell of the predecessor inside the batch, place in its chain, chain length, -}
  const int32_t* entry_snp;
  const double* egls;
  const double* af;
  const double* diag;  // [S][K][4] = {g00, g11, g22, B}: the state at the start of the batch
  const double* offd;  // [S][K][6]
  int32_t* ic;         // per incidence: cluster this ratio belongs to (-1: none)
  double2* rat;        // {term2, term0}(replayed) / (start)
};

// Ratio of incidence x (predecessor x of its hot entry) under the guess w[]: only the LAST predecessor that joined a given
// cluster carries that cluster's ratio, and it replays every earlier predecessor of the same cluster before itself.
template <class W>
__device__ __forceinline__ void greedy_ratio(const greedy_tabs& T, int K, int64_t x, int64_t oi0, W w) {
  const int32_t h = T.inc_hot[x];
  const int64_t xb = T.hinc_ptr[h], xe = T.hinc_ptr[h + 1];
  const int c = w(T.inc_cell[x] - oi0);
  bool active = c >= 0;
  for (int64_t y = x + 1; y < xe && active; ++y) active = w(T.inc_cell[y] - oi0) != c;
  if (!active) {
    T.ic[x] = -1;
    return;
  }
  const int32_t p = T.hot_pos[h];
  const int32_t i = T.pos_cell[p];
  const int64_t e = T.hdr_e0[i] + (p - T.pos_ptr[i]);
  const int32_t snp = T.entry_snp[e];
  const double a = T.af[snp];
  const double h0 = (1.0 - a) * (1.0 - a), h1 = 2.0 * a * (1.0 - a), h2 = a * a;
  const double* gl = T.egls + (size_t)e * 9;
  const double wx = gl[0] * h0, wy = gl[4] * h1, wz = gl[8] * h2, A = (wx + wy) + wz;
  const double2* dg = reinterpret_cast<const double2*>(T.diag + ((size_t)snp * K + c) * 4);
  const double2* od = reinterpret_cast<const double2*>(T.offd + ((size_t)snp * K + c) * 6);
  const double2 r0 = dg[0], r1 = dg[1];
  bool present = r1.y != 0.0;
  const double o2 = present ? (wx * r0.x + wy * r0.y) + wz * r1.x : 1.0, o0 = present ? A * r1.y : 1.0;
  double v[9];
  if (present) {
    const double2 r2 = od[0], r3 = od[1], r4 = od[2];
    v[0] = r0.x, v[1] = r2.x, v[2] = r2.y, v[3] = r3.x, v[4] = r0.y, v[5] = r3.y, v[6] = r4.x, v[7] = r4.y, v[8] = r1.x;
  }
  for (int64_t y = xb; y <= x; ++y) {
    const int32_t iq = T.inc_cell[y];
    if (w(iq - oi0) != c) continue;
    const int32_t q = T.inc_pos[y];
    greedy_merge9(v, present, T.egls + (size_t)(T.hdr_e0[iq] + (q - T.pos_ptr[iq])) * 9);
    present = true;
  }
  const double B = (v[0] * h0 + v[4] * h1) + v[8] * h2;
  T.ic[x] = c;
  T.rat[x] = make_double2(((wx * v[0] + wy * v[4]) + wz * v[8]) / o2, (A * B) / o0);
}

// start products of the batch's cells per cluster (chunk partials in entry order) and the first guess: their argmax
__global__ void __launch_bounds__(1024)
    greedy_argmax0_kernel(int64_t oi0, int nb, const int64_t* __restrict__ chunk_first, int K, int Kp,
                          const double2* __restrict__ pm, const int2* __restrict__ px, double2* __restrict__ bm,
                          int2* __restrict__ bx, int32_t* __restrict__ wguess) {
  __shared__ double sc[GB * 64];
  const int t = threadIdx.x;
  const int64_t cbase = chunk_first[oi0];
  for (int idx = t; idx < nb * Kp; idx += blockDim.x) {
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
    bm[b * 64 + jj] = make_double2(a2, a0);
    bx[b * 64 + jj] = make_int2(b2, b0);
    sc[b * 64 + jj] = prodacc_log(a2, b2) - prodacc_log(a0, b0);
  }
  __syncthreads();
  if (t < nb) {  // first maximum in cluster order (:233-242)
    int best = 0;
    double bs = sc[t * 64];
    for (int c = 1; c < K; ++c)
      if (sc[t * 64 + c] > bs) {
        bs = sc[t * 64 + c];
        best = c;
      }
    wguess[t] = best;
  }
}

__global__ void __launch_bounds__(256)
    greedy_ratio_kernel(greedy_tabs T, int K, int64_t oi0, int nb, const int64_t* __restrict__ hot_ptr,
                        const int32_t* __restrict__ wguess) {
  const int64_t x0 = T.hinc_ptr[hot_ptr[oi0]], x1 = T.hinc_ptr[hot_ptr[oi0 + nb]];
  for (int64_t x = x0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < x1; x += (int64_t)gridDim.x * blockDim.x)
    greedy_ratio(T, K, x, oi0, [&](int b) { return wguess[b]; });
}

// Scores of one batch cell under the ratios in T.ic / T.rat: the calling wave's lane c holds cluster c.  The cell's
// incidences are read 256 at a time (lane = incidence, four coalesced requests in flight) and handed round with readlane
// in their fixed order (entry order, then chain order): lane c multiplies the ratios that belong to cluster c into its
// product -- one read of every ratio, not one per cluster.  Returns log lk2 - log lk0 of (cell, cluster c).
__device__ __forceinline__ double greedy_cell_score(const greedy_tabs& T, int K, int c, int b, int64_t oi0,
                                                    const int64_t* __restrict__ hot_ptr, const double2* __restrict__ bm,
                                                    const int2* __restrict__ bx) {
  double a2 = 1.0, a0 = 1.0;
  int32_t b2 = 0, b0 = 0;
  if (c < K) {
    const double2 m = bm[b * 64 + c];
    const int2 xx = bx[b * 64 + c];
    a2 = m.x, a0 = m.y, b2 = xx.x, b0 = xx.y;
  }
  const int64_t y0 = T.hinc_ptr[hot_ptr[oi0 + b]], y1 = T.hinc_ptr[hot_ptr[oi0 + b + 1]];
  int cnt = 0;
  for (int64_t yb = y0; yb < y1; yb += 256) {
    int32_t ci[4];
    double2 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t y = yb + u * 64 + c;
      ci[u] = (y < y1) ? T.ic[y] : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t y = yb + u * 64 + c;
      r[u] = (ci[u] >= 0) ? T.rat[y] : make_double2(1.0, 1.0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      uint64_t live = __ballot(ci[u] >= 0);
      while (live) {  // wave-uniform
        const int k = __builtin_ctzll(live);
        live &= live - 1;
        const int ck = __builtin_amdgcn_readlane(ci[u], k);
        const double rx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r[u].x), k),
                                           __builtin_amdgcn_readlane(__double2loint(r[u].x), k));
        const double ry = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r[u].y), k),
                                           __builtin_amdgcn_readlane(__double2loint(r[u].y), k));
        if (c == ck) {
          a2 *= rx;
          a0 *= ry;
          if (++cnt == 4) {  // a ratio lies within 1e-30 .. 1e30
            cnt = 0;
            prodacc_renorm(a2, b2);
            prodacc_renorm(a0, b0);
          }
        }
      }
    }
  }
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

// first pass of the fixpoint for every cell of the batch at once (one wave per cell, whole chip): the guess that takes
// the batch's own merges into account.  In all but a few batches it confirms the first guess and the decision is made.
__global__ void __launch_bounds__(64)
    greedy_score_kernel(greedy_tabs T, int K, int64_t oi0, const int64_t* __restrict__ hot_ptr,
                        const double2* __restrict__ bm, const int2* __restrict__ bx, const int32_t* __restrict__ wguess,
                        int32_t* __restrict__ wnew /*[GB] guesses, [GB .. 2 GB) changed flags, [2 GB] any*/) {
  const int b = blockIdx.x, c = threadIdx.x;
  const double sc = greedy_cell_score(T, K, c, b, oi0, hot_ptr, bm, bx);
  const int best = greedy_wave_argmax(sc, c, K);
  if (c == 0) {
    const int chg = best != wguess[b];
    wnew[b] = best;
    wnew[GB + b] = chg;
    if (chg) atomicOr(wnew + 2 * GB, 1);
  }
}

constexpr int BT = 1024;      // threads of greedy_decide_kernel
constexpr int DI_CAP = 8192;   // incidences of a batch whose chain bytes are kept in LDS (32 KB); denser batches read them from L2

__global__ void __launch_bounds__(BT)
    greedy_decide_kernel(greedy_tabs T, int K, int Kp, int64_t oi0, int nb, const int64_t* __restrict__ hot_ptr,
                         const int32_t* __restrict__ hdr_cell, const double2* __restrict__ bm,
                         const int2* __restrict__ bx, int32_t* wguess /* greedy_score_kernel's wnew */, int32_t* __restrict__ clust,
                         int32_t* __restrict__ pass_hist /* NULL or [GB + 2]: batches by number of passes (MUXGL_TIMING) */) {
  __shared__ double s_sc[GB * 64];
  __shared__ int32_t s_w[GB];
  __shared__ int32_t s_chg[GB];  // the guess of this cell changed in the last pass
  __shared__ int s_any;
  __shared__ uchar4 s_meta[DI_CAP];
  const int t = threadIdx.x;
  // the first pass was made by greedy_score_kernel: nothing changed -> the guesses are the decision
  const bool any0 = wguess[2 * GB] != 0;
  __syncthreads();
  if (t == 0) wguess[2 * GB] = 0;  // for the next batch's score kernel (stream order)
  if (!any0) {
    if (t < nb) clust[hdr_cell[oi0 + t]] = wguess[t];
    if (pass_hist && t == 0) atomicAdd(pass_hist + 1, 1);
    return;
  }
  if (t < nb) {
    s_w[t] = wguess[t];
    s_chg[t] = wguess[GB + t];
  }
  const int64_t x0 = T.hinc_ptr[hot_ptr[oi0]], x1 = T.hinc_ptr[hot_ptr[oi0 + nb]];
  const bool in_lds = x1 - x0 <= DI_CAP;
  if (in_lds)
    for (int64_t x = x0 + t; x < x1; x += BT) s_meta[x - x0] = T.inc_meta[x];
  __syncthreads();
  for (int pass = 1; pass <= nb; ++pass) {  // cell i of the batch is final after pass i at the latest
    // ratios of the hot entries that have a predecessor whose guess changed (every incidence of such an entry: which
    // predecessor carries a cluster's ratio depends on all of them)
    for (int64_t x = x0 + t; x < x1; x += BT) {
      bool redo = false;
      if (in_lds) {
        const uchar4 m = s_meta[x - x0];
        const int64_t yb = x - x0 - m.y;
        for (int k = 0; k < (int)m.z && !redo; ++k) redo = s_chg[s_meta[yb + k].x] != 0;
      } else {
        const int32_t h = T.inc_hot[x];
        for (int64_t y = T.hinc_ptr[h]; y < T.hinc_ptr[h + 1] && !redo; ++y) redo = s_chg[T.inc_cell[y] - oi0] != 0;
      }
      if (redo) greedy_ratio(T, K, x, oi0, [&](int b) { return s_w[b]; });
    }
    __syncthreads();  // (drains the stores: one workgroup, one L1 -- they are visible to the loads below)
    if (t == 0) s_any = 0;
    for (int b = t >> 6; b < nb; b += BT / 64) {  // a wave per cell, lane = cluster
      const int c = t & 63;
      const double sc = greedy_cell_score(T, K, c, b, oi0, hot_ptr, bm, bx);
      if (c < K) s_sc[b * 64 + c] = sc;
    }
    __syncthreads();
    if (t < nb) {
      int best = 0;
      double bs = s_sc[t * 64];
      for (int c = 1; c < K; ++c)
        if (s_sc[t * 64 + c] > bs) {
          bs = s_sc[t * 64 + c];
          best = c;
        }
      s_chg[t] = best != s_w[t];
      if (best != s_w[t]) {
        s_w[t] = best;
        s_any = 1;
      }
    }
    __syncthreads();
    if (!s_any) {
      if (pass_hist && t == 0) atomicAdd(pass_hist + pass + 1, 1);
      break;
    }
  }
  if (t < nb) clust[hdr_cell[oi0 + t]] = s_w[t];
}

// the batch's merges: one thread per chain (the entries of the batch at one SNP), in cell order
__global__ void __launch_bounds__(256)
    greedy_apply_kernel(greedy_tabs T, int K, int64_t oi0, int nb, const int32_t* __restrict__ prev,
                        const int32_t* __restrict__ next, const int32_t* __restrict__ hdr_cell,
                        const int32_t* __restrict__ clust, double* __restrict__ diag, double* __restrict__ offd) {
  const int64_t p0 = T.pos_ptr[oi0], p1 = T.pos_ptr[oi0 + nb];
  for (int64_t p = p0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < p1; p += (int64_t)gridDim.x * blockDim.x) {
    if (prev[p] >= 0) continue;  // not the head of its chain
    const int32_t i0 = T.pos_cell[p];
    const int32_t snp = T.entry_snp[T.hdr_e0[i0] + (p - T.pos_ptr[i0])];
    const double a = T.af[snp];
    for (int32_t q = (int32_t)p; q >= 0; q = next[q]) {
      const int32_t i = T.pos_cell[q];
      const int w = clust[hdr_cell[i]];
      double2* dg = reinterpret_cast<double2*>(diag + ((size_t)snp * K + w) * 4);
      double2* od = reinterpret_cast<double2*>(offd + ((size_t)snp * K + w) * 6);
      const double2 r0 = dg[0], r1 = dg[1];
      const bool present = r1.y != 0.0;
      double v[9];
      if (present) {
        const double2 r2 = od[0], r3 = od[1], r4 = od[2];
        v[0] = r0.x, v[1] = r2.x, v[2] = r2.y, v[3] = r3.x, v[4] = r0.y, v[5] = r3.y, v[6] = r4.x, v[7] = r4.y, v[8] = r1.x;
      }
      greedy_merge9(v, present, T.egls + (size_t)(T.hdr_e0[i] + (q - T.pos_ptr[i])) * 9);
      const double B = (v[0] * ((1.0 - a) * (1.0 - a)) + v[4] * (2.0 * a * (1.0 - a))) + v[8] * (a * a);
      dg[0] = make_double2(v[0], v[4]);
      dg[1] = make_double2(v[8], B);
      od[0] = make_double2(v[1], v[2]);
      od[1] = make_double2(v[3], v[5]);
      od[2] = make_double2(v[6], v[7]);
    }
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
  host_timer tm;
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
  // batched path (K <= 64): chunk tables and chunk partials of the distance kernel, chain tables of the fixpoint.
  // It takes the chain of memory round trips of a step off the critical path for every K it covers (1.x us per cell
  // against 22 us at K = 16 and 58 us at K = 64 for the serial kernel), so it is the default; the serial kernel remains
  // for K > 64, for jobs with 2^31 or more entries to cluster, and behind MUXGL_FLAG_FORCE_TILE_SWEEP.
  int64_t P = 0;
  std::vector<int64_t> pos_ptr(npad + 1, 0);
  for (size_t i = 0; i < npad; ++i) pos_ptr[i + 1] = pos_ptr[i] + (i < n ? hlen[i] : 0);
  P = pos_ptr[n];
  const bool batched = K <= 64 && P > 0 && P < ((int64_t)1 << 31) && !(h->flags & MUXGL_FLAG_FORCE_TILE_SWEEP);
  int32_t* d_chunk_cell = nullptr;
  int64_t* d_chunk_first = nullptr;
  double2* d_pm = nullptr;
  int2* d_px = nullptr;
  int64_t *d_pos_ptr = nullptr, *d_nhot = nullptr, *d_hot_ptr = nullptr, *d_hot_len = nullptr, *d_hinc_ptr = nullptr;
  uint64_t *d_key = nullptr, *d_key2 = nullptr;
  uint32_t *d_val = nullptr, *d_val2 = nullptr;
  int32_t *d_pos_cell = nullptr, *d_prev = nullptr, *d_next = nullptr, *d_hot_pos = nullptr;
  int32_t *d_inc_pos = nullptr, *d_inc_cell = nullptr, *d_inc_hot = nullptr, *d_ic = nullptr, *d_wguess = nullptr;
  uchar4* d_inc_meta = nullptr;
  int32_t* d_hist = nullptr;
  int32_t* d_wnew = nullptr;  // [GB] guesses after the first pass, [GB] changed flags, [1] any
  double2 *d_rat = nullptr, *d_bm = nullptr;
  int2* d_bx = nullptr;
  void* d_tmp = nullptr;
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
  h->err.clear();
  tm.lap("greedy_init: host sort + step headers");
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
      const size_t nP = (size_t)P;
      if (dev_alloc(h, &d_chunk_cell, chunk_cell.size()) || dev_alloc(h, &d_chunk_first, chunk_first.size()) ||
          dev_alloc(h, &d_pm, (size_t)max_batch_chunks * Kp) || dev_alloc(h, &d_px, (size_t)max_batch_chunks * Kp) ||
          dev_alloc(h, &d_pos_ptr, npad + 1) || dev_alloc(h, &d_key, nP) || dev_alloc(h, &d_key2, nP) ||
          dev_alloc(h, &d_val, nP) || dev_alloc(h, &d_val2, nP) || dev_alloc(h, &d_pos_cell, nP) ||
          dev_alloc(h, &d_prev, nP) || dev_alloc(h, &d_next, nP) || dev_alloc(h, &d_nhot, npad + 1) ||
          dev_alloc(h, &d_hot_ptr, npad + 1) || dev_alloc(h, &d_bm, (size_t)GB * 64) || dev_alloc(h, &d_bx, (size_t)GB * 64) ||
          dev_alloc(h, &d_wguess, (size_t)GB) || dev_alloc(h, &d_wnew, (size_t)2 * GB + 1) || dev_alloc(h, &d_hist, (size_t)GB + 2))
        break;
      (void)hipMemsetAsync(d_hist, 0, sizeof(int32_t) * (GB + 2), h->stream);
      (void)hipMemsetAsync(d_wnew, 0, sizeof(int32_t) * (2 * GB + 1), h->stream);
      e = hipMemsetAsync(d_offd, 0, sizeof(double) * (size_t)S * K * 6, h->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(d_chunk_cell, chunk_cell.data(), sizeof(int32_t) * chunk_cell.size(), hipMemcpyHostToDevice,
                           h->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(d_chunk_first, chunk_first.data(), sizeof(int64_t) * chunk_first.size(), hipMemcpyHostToDevice,
                           h->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(d_pos_ptr, pos_ptr.data(), sizeof(int64_t) * (npad + 1), hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) e = hipMemsetAsync(d_nhot, 0, sizeof(int64_t) * (npad + 1), h->stream);
      if (e != hipSuccess) break;
      // ---- chain tables: entries of a batch at the same SNP, linked in cell order (one stable sort of (batch, SNP))
      const unsigned cblocks = (unsigned)((n + 3) / 4);
      hipLaunchKernelGGL(greedy_keys_kernel, dim3(cblocks), dim3(256), 0, h->stream, (int64_t)n, d_he0, d_pos_ptr,
                         h->d_entry_snp, d_key, d_val, d_pos_cell);
      unsigned sbits = 1, bbits = 1;
      while (sbits < 31 && ((int64_t)1 << sbits) < S) ++sbits;
      while (bbits < 31 && ((int64_t)1 << bbits) < (int64_t)(n / GB + 1)) ++bbits;
      size_t tmp_bytes = 0;
      e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_key, d_key2, d_val, d_val2, nP, 0u, 32u + bbits, h->stream);
      if (e == hipSuccess) e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1);
      if (e == hipSuccess)
        e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_key, d_key2, d_val, d_val2, nP, 0u, 32u + bbits, h->stream);
      if (e != hipSuccess) break;
      (void)sbits;
      const unsigned pblocks = (unsigned)std::min<int64_t>((P + 255) / 256, 16384);
      hipLaunchKernelGGL(greedy_links_kernel, dim3(pblocks), dim3(256), 0, h->stream, P, d_key2, d_val2, d_prev, d_next);
      hipLaunchKernelGGL(greedy_hot_count_kernel, dim3(cblocks), dim3(256), 0, h->stream, (int64_t)n, d_pos_ptr, d_prev,
                         d_nhot);
      auto scan = [&](int64_t* in, int64_t* out, size_t cnt) -> hipError_t {
        size_t tb = 0;
        hipError_t er = rocprim::exclusive_scan(nullptr, tb, in, out, (int64_t)0, cnt, rocprim::plus<int64_t>(), h->stream);
        void* tmp = nullptr;
        if (er == hipSuccess) er = hipMalloc(&tmp, tb ? tb : 1);
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
      dev_free(&d_key);  // the unsorted keys and payloads are done with
      dev_free(&d_val);
      if (dev_alloc(h, &d_hot_pos, (size_t)H) || dev_alloc(h, &d_hot_len, (size_t)H + 1) ||
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
      if (dev_alloc(h, &d_inc_pos, (size_t)I) || dev_alloc(h, &d_inc_cell, (size_t)I) || dev_alloc(h, &d_inc_hot, (size_t)I) ||
          dev_alloc(h, &d_ic, (size_t)I) || dev_alloc(h, &d_rat, (size_t)I) || dev_alloc(h, &d_inc_meta, (size_t)I))
        break;
      if (H > 0)
        hipLaunchKernelGGL(greedy_inc_fill_kernel, dim3((unsigned)std::min<int64_t>((H + 255) / 256, 16384)), dim3(256), 0,
                           h->stream, H, d_hot_pos, d_prev, d_hinc_ptr, d_pos_cell, d_inc_pos, d_inc_cell, d_inc_hot, d_inc_meta);
      e = hipGetLastError();
      if (e != hipSuccess) break;
      if (tm.on) (void)hipStreamSynchronize(h->stream);
      tm.lap("greedy_init: chain tables (sort, links, hot lists)");
      greedy_tabs T;
      T.hdr_e0 = d_he0;
      T.pos_ptr = d_pos_ptr;
      T.pos_cell = d_pos_cell;
      T.hot_pos = d_hot_pos;
      T.hinc_ptr = d_hinc_ptr;
      T.inc_pos = d_inc_pos;
      T.inc_cell = d_inc_cell;
      T.inc_hot = d_inc_hot;
      T.inc_meta = d_inc_meta;
      T.entry_snp = h->d_entry_snp;
      T.egls = h->d_egls;
      T.af = h->d_af;
      T.diag = d_diag;
      T.offd = d_offd;
      T.ic = d_ic;
      T.rat = d_rat;
      for (size_t i = 0; i < n && e == hipSuccess; i += GB) {
        const int nb = (int)std::min<size_t>(GB, n - i);
        const int64_t nch = chunk_first[i + nb] - chunk_first[i];
        if (nch > 0)
          hipLaunchKernelGGL(greedy_dist_kernel, dim3((unsigned)nch), dim3(GA_T), 0, h->stream, chunk_first[i],
                             d_chunk_cell, d_chunk_first, d_he0, d_hlen, h->d_entry_snp, h->d_egls, h->d_af, (int)K, Kp,
                             d_diag, d_pm, d_px);
        hipLaunchKernelGGL(greedy_argmax0_kernel, dim3(1), dim3(1024), 0, h->stream, (int64_t)i, nb, d_chunk_first, (int)K,
                           Kp, d_pm, d_px, d_bm, d_bx, d_wguess);
        hipLaunchKernelGGL(greedy_ratio_kernel, dim3(64), dim3(256), 0, h->stream, T, (int)K, (int64_t)i, nb, d_hot_ptr,
                           d_wguess);
        hipLaunchKernelGGL(greedy_score_kernel, dim3((unsigned)nb), dim3(64), 0, h->stream, T, (int)K, (int64_t)i, d_hot_ptr,
                           d_bm, d_bx, d_wguess, d_wnew);
        hipLaunchKernelGGL(greedy_decide_kernel, dim3(1), dim3(BT), 0, h->stream, T, (int)K, Kp, (int64_t)i, nb, d_hot_ptr,
                           d_hcell, d_bm, d_bx, d_wnew, d_clust, tm.on ? d_hist : nullptr);
        hipLaunchKernelGGL(greedy_apply_kernel, dim3(128), dim3(256), 0, h->stream, T, (int)K, (int64_t)i, nb, d_prev, d_next,
                           d_hcell, d_clust, d_diag, d_offd);
        e = hipGetLastError();
      }
    }
    tm.lap("greedy_init: batches enqueued");
    if (e == hipSuccess) e = hipMemcpyAsync(clust_out, d_clust, sizeof(int32_t) * (size_t)C, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    tm.lap("greedy_init: batches drained");
    if (tm.on && d_hist) {
      int32_t hist[GB + 2];
      if (hipMemcpy(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost) == hipSuccess) {
        fprintf(stderr, "[muxgl] greedy_init: batches by passes of the fixpoint:");
        for (int q = 1; q < GB + 2; ++q)
          if (hist[q]) fprintf(stderr, " %d:%d", q, hist[q]);
        fprintf(stderr, "\n");
      }
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
  dev_free(&d_clust);
  dev_free(&d_diag);
  dev_free(&d_offd);
  dev_free(&d_chunk_cell);
  dev_free(&d_chunk_first);
  dev_free(&d_pm);
  dev_free(&d_px);
  dev_free(&d_pos_ptr);
  dev_free(&d_nhot);
  dev_free(&d_hot_ptr);
  dev_free(&d_hot_len);
  dev_free(&d_hinc_ptr);
  dev_free(&d_key);
  dev_free(&d_key2);
  dev_free(&d_val);
  dev_free(&d_val2);
  dev_free(&d_pos_cell);
  dev_free(&d_prev);
  dev_free(&d_next);
  dev_free(&d_hot_pos);
  dev_free(&d_inc_pos);
  dev_free(&d_inc_cell);
  dev_free(&d_inc_hot);
  dev_free(&d_inc_meta);
  dev_free(&d_hist);
  dev_free(&d_wnew);
  dev_free(&d_ic);
  dev_free(&d_wguess);
  dev_free(&d_rat);
  dev_free(&d_bm);
  dev_free(&d_bx);
  if (d_tmp) (void)hipFree(d_tmp);
  return rc;
}
