// fmx_old.hip -- what `popscle freemuxlet-old` (cmd_cram_freemuxlet.cpp) does differently from freemux2, on the device:
//
//   * the pairwise droplet distance matrix dropDs (cmd_cram_freemuxlet.cpp:176-221): for every SNP and every pair of
//     cells (a > b) covering it, llk2 += log(sum_g gl_a[g,g] gl_b[g,g] hwe[g]), llk0 += log(sum_{g,h} gl_a[g,g]
//     gl_b[h,h] hwe[g] hwe[h]), plus three counters.  O(sum_v n_v^2 / 2) terms, O(C^2) results -- the reference keeps
//     32 bytes per pair in host memory, which is what made the command infeasible for large runs; here the C x C
//     matrix of vote signs (1 byte per ordered pair) stays in HBM and the full records exist only on request.
//   * the first-pass voting (:245-291) and the refinement passes (:297-343): sequential over cells by construction
//     (every election changes the labels the next cell counts), parallel inside a step.
//
// pair kernel: workgroup = (cell a, block of PD_NB columns b < a), accumulators {llk0, llk2} of the block in LDS.  Each
// wave owns PD_SUB columns and walks ALL entries of cell a in ascending SNP order -- so every pair receives its terms
// in the reference's order (SNPs ascending, :187) and no two waves ever touch the same accumulator.  For an entry the
// wave needs the cells of that SNP that fall into its column range: the SNP-major view (cells ascending inside a SNP)
// makes that a contiguous run, found by two binary searches done 64 entries at a time (lane = entry).  The runs are
// short (a SNP's ~200 cells spread over 10 k columns), so the terms of the 64 entries are flattened: a prefix sum of
// the run lengths, then lane = term, 64 terms per step whatever entries they belong to -- one gather of the partner's
// diagonal likelihoods and two log's on full waves.  Only the LDS read-modify-write goes entry by entry (the same
// column can occur under two entries of a step, and the order of the additions is the reference's).
//
// vote kernel: ONE persistent 1024-thread workgroup.  A step counts, for every cluster k, the +1/-1 votes of the
// already labelled cells in a row of the sign matrix.  The reference adds them one by one to a double that starts at a
// random jitter < 0.001 (:257-259,304-306); the integer part of that sum is order-independent, so the fast path is an
// integer histogram: thread = (cluster k, row segment), four cells per v_dot4_i32_i8 on a byte-equality mask of the
// labels.  The jitter only matters between clusters whose integer votes tie for the maximum; for those the kernel
// reproduces the reference's sequence of roundings exactly (see vote_exact below) from four integers per cluster --
// sum, max and min prefix, sign of the first vote -- which combine associatively over row segments.
#include <climits>
#include <vector>

#include "common.hpp"

namespace {

constexpr int PD_T = 1024;  // 16 waves
// columns per workgroup: 6144 x 16 B = 96 KB of LDS accumulators (3072 x 28 B when the counters are wanted too), plus
// 3 KB of staging per wave for the entries of a batch
struct pd_stage {  // per wave and batch of 64 entries
  double4 ent[64];   // {gl[0,0], gl[1,1], gl[2,2], allele frequency} of the entry
  int64_t lo[64];    // start of the entry's run in the SNP-major arrays
  int32_t pre[64];   // exclusive prefix sum of the run lengths
  int32_t nr[64];    // reads of the entry (counters only)
};
template <bool FULL>
struct pd_geom {
  static constexpr int NB = FULL ? 3072 : 6144;
  static constexpr int SUB = NB / (PD_T / 64);
  static constexpr size_t ACC = (size_t)NB * (FULL ? 28 : 16);
  static constexpr size_t LDS = ACC + sizeof(pd_stage) * (PD_T / 64);
};

__device__ __forceinline__ int64_t lower_bound_i32(const int32_t* __restrict__ a, int64_t lo, int64_t hi, int32_t x) {
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

template <bool FULL>
__global__ void __launch_bounds__(PD_T)
    fmxold_pair_kernel(int32_t C, int64_t ld, const int64_t* __restrict__ cell_ptr, const int32_t* __restrict__ entry_snp,
                       const double* __restrict__ egls, const int32_t* __restrict__ ecnt, const double* __restrict__ af,
                       const int64_t* __restrict__ snp_ptr, const int32_t* __restrict__ snp_cell,
                       const double* __restrict__ segls, const int32_t* __restrict__ secnt, double bf_thres,
                       int8_t* __restrict__ sgn, muxgl_dropd* __restrict__ full) {
  constexpr int PD_NB = pd_geom<FULL>::NB, PD_SUB = pd_geom<FULL>::SUB;
  extern __shared__ double2 s_acc[];  // [PD_NB] {llk0, llk2}; FULL: followed by int32 [PD_NB][3]
  int32_t* s_cnt = reinterpret_cast<int32_t*>(s_acc + PD_NB);  // FULL only
  const int a = (int)blockIdx.x;
  const int base = (int)blockIdx.y * PD_NB;
  if (base >= a) return;
  const int ncol = min(PD_NB, a - base);
  const int tid = (int)threadIdx.x;
  for (int t = tid; t < ncol; t += PD_T) {
    s_acc[t] = make_double2(0.0, 0.0);
    if (FULL) s_cnt[3 * t] = s_cnt[3 * t + 1] = s_cnt[3 * t + 2] = 0;
  }
  __syncthreads();
  const int w = tid >> 6, lane = tid & 63;
  const int c_lo = base + w * PD_SUB;
  const int c_hi = min(c_lo + PD_SUB, a);
  pd_stage* st = reinterpret_cast<pd_stage*>(reinterpret_cast<char*>(s_acc) + pd_geom<FULL>::ACC) + w;
  if (c_lo < c_hi) {
    const int64_t e0 = cell_ptr[a], e1 = cell_ptr[a + 1];
    for (int64_t eb = e0; eb < e1; eb += 64) {
      // lane = entry: its run of partner cells inside this wave's column range, and what a term needs from the entry
      const int64_t e = eb + lane;
      int64_t lo = 0;
      int n = 0;
      double4 ent = make_double4(0, 0, 0, 0);
      int32_t nri = 0;
      if (e < e1) {
        const int32_t v = entry_snp[e];
        const int64_t p0 = snp_ptr[v], p1 = snp_ptr[v + 1];
        lo = lower_bound_i32(snp_cell, p0, p1, c_lo);
        n = (int)(lower_bound_i32(snp_cell, lo, p1, c_hi) - lo);
        if (n > 0) {
          ent = make_double4(egls[e * 9], egls[e * 9 + 4], egls[e * 9 + 8], af[v]);
          if (FULL) nri = ecnt[e * 3];
        }
      }
      int pre = n;  // inclusive scan over the lanes
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(pre, off, 64);
        if (lane >= off) pre += o;
      }
      const int total = __shfl(pre, 63, 64);
      st->ent[lane] = ent;
      st->lo[lane] = lo;
      st->pre[lane] = pre - n;
      if (FULL) st->nr[lane] = nri;
      // lane = term: 64 terms at a time, whatever entries they belong to (the wave's LDS traffic needs no barrier)
      for (int g0 = 0; g0 < total; g0 += 64) {
        const int term = g0 + lane;
        const bool valid = term < total;
        int k = 0;  // largest k with pre[k] <= term and a non-empty run
        if (valid) {
#pragma unroll
          for (int step = 32; step > 0; step >>= 1)
            if (st->pre[k + step] <= term) k += step;  // empty runs share their successor's prefix: the last one wins
        }
        double l0 = 0, l2 = 0;
        int col = 0;
        int32_t nrj = 0;
        if (valid) {
          const int64_t p = st->lo[k] + (term - st->pre[k]);
          col = snp_cell[p] - base;
          const double gj0 = segls[p * 9], gj1 = segls[p * 9 + 4], gj2 = segls[p * 9 + 8];
          if (FULL) nrj = secnt[p * 3];
          const double4 en = st->ent[k];
          const double gi0 = en.x, gi1 = en.y, gi2 = en.z, f = en.w;
          const double gp0 = (1.0 - f) * (1.0 - f), gp1 = 2.0 * f * (1.0 - f), gp2 = f * f;  // :199-201
          // :203-208, the reference's association: ((gl_a * gl_b) * hwe_g) * hwe_h, summed g-major
          double lk2 = (gi0 * gj0) * gp0;
          lk2 += (gi1 * gj1) * gp1;
          lk2 += (gi2 * gj2) * gp2;
          double lk0 = ((gi0 * gj0) * gp0) * gp0;
          lk0 += ((gi0 * gj1) * gp0) * gp1;
          lk0 += ((gi0 * gj2) * gp0) * gp2;
          lk0 += ((gi1 * gj0) * gp1) * gp0;
          lk0 += ((gi1 * gj1) * gp1) * gp1;
          lk0 += ((gi1 * gj2) * gp1) * gp2;
          lk0 += ((gi2 * gj0) * gp2) * gp0;
          lk0 += ((gi2 * gj1) * gp2) * gp1;
          lk0 += ((gi2 * gj2) * gp2) * gp2;
          l0 = log(lk0);
          l2 = log(lk2);
        }
        // accumulate entry by entry, in ascending SNP order (:187): the cells of one entry's run are distinct, the same
        // cell may appear again under a later entry of the group
        uint64_t pend = __ballot(valid);
        while (pend) {
          const int f = __ffsll((unsigned long long)pend) - 1;
          const int kf = __shfl(k, f, 64);
          const uint64_t sel = __ballot(valid && k == kf) & pend;
          if ((sel >> lane) & 1) {
            double2 acc = s_acc[col];
            acc.x += l0;
            acc.y += l2;
            s_acc[col] = acc;
            if (FULL) {
              s_cnt[3 * col] += 1;
              s_cnt[3 * col + 1] += st->nr[k];
              s_cnt[3 * col + 2] += nrj;
            }
          }
          pend &= ~sel;
        }
      }
    }
  }
  __syncthreads();
  for (int t = tid; t < ncol; t += PD_T) {
    const double2 acc = s_acc[t];
    const int b = base + t;
    // :273-278 / :312-313: -1 when llk0 - llk2 > thres, +1 when llk2 - llk0 > thres
    const int8_t s = (acc.x - acc.y > bf_thres) ? (int8_t)-1 : ((acc.y - acc.x > bf_thres) ? (int8_t)1 : (int8_t)0);
    sgn[(int64_t)a * ld + b] = s;
    sgn[(int64_t)b * ld + a] = s;
    if (FULL) {
      muxgl_dropd d;
      d.nsnps = s_cnt[3 * t];
      d.nread1 = s_cnt[3 * t + 1];
      d.nread2 = s_cnt[3 * t + 2];
      d._pad = 0;
      d.llk0 = acc.x;
      d.llk2 = acc.y;
      full[(int64_t)a * (a - 1) / 2 + b] = d;
    }
  }
}

// sign matrix in visiting order for the first pass: out[x][y] = sgn[order[x]][order[y]], x,y < n
__global__ void __launch_bounds__(256)
    fmxold_permute_kernel(int32_t n, int64_t ld, const int32_t* __restrict__ order, const int8_t* __restrict__ sgn,
                          int8_t* __restrict__ out) {
  const int x = (int)blockIdx.y;
  const int y = (int)(blockIdx.x * 256 + threadIdx.x);
  if (x < n && y < n) out[(int64_t)x * ld + y] = sgn[(int64_t)order[x] * ld + order[y]];
}

constexpr int VT = 1024;
constexpr int VSEG = 32;  // row segments per tied cluster on the exact path

// 0x01 in every byte of `lab` that equals the byte replicated in `kk`
__device__ __forceinline__ uint32_t byte_eq01(uint32_t lab, uint32_t kk) {
  const uint32_t x = lab ^ kk;
  uint32_t t = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
  t = ~(t | x | 0x7f7f7f7fu);  // 0x80 where the byte of x is zero
  return t >> 7;
}

// The reference's vote: a double that starts at the jitter f0 in [0, 0.001) and receives `+= 1.0` / `-= 1.0` one at a
// time.  After any prefix the value is n + f with n the integer vote sum and f the jitter rounded to the coarsest
// floating-point grid the running value has visited: a sum n + f with n >= 1 lies in binade floor(log2 n), one with
// n <= -1 in binade ceil(log2 |n|) - 1 (its magnitude is just below |n|), and n = 0 is exact.  |n| moves by one per
// vote, so the binades are visited in increasing order starting at 0 (first vote +1) or -1 (first vote -1), up to the
// binade of the largest excursion; rounding to a nested sequence of grids only depends on that sequence, and rounding f
// to the grid of binade e is (f + 2^e) - 2^e in IEEE arithmetic (ties-to-even agrees because n / ulp is even).
// Checked against the sequential sum on random walks in tests/test_fmxold.py.
__device__ __forceinline__ double vote_exact(double f, int n, int M, int m, int first) {
  if (first != 0) {
    int emax = first < 0 ? -1 : 0;
    if (M >= 1) emax = max(emax, 31 - __clz(M));
    if (m <= -2) emax = max(emax, 31 - __clz(-m - 1));
    for (int e = (first < 0 ? -1 : 0); e <= emax; ++e) {
      const double c = ldexp(1.0, e);
      f = (f + c) - c;
    }
  }
  return (double)n + f;
}

// mode 0: first pass (:245-291): step t scans positions [0,t) of row t of the permuted matrix, labels by position.
// mode 1: refinement (:297-343): step t scans the whole row of cell order[t], labels by cell.
// stats[0] = changed, stats[1..K] = ccounts.
__global__ void __launch_bounds__(VT)
    fmxold_vote_kernel(int mode, int32_t C, int64_t ld, int K, int32_t nsteps, const int8_t* __restrict__ mat,
                       const int32_t* __restrict__ order, const double* __restrict__ jitter, int keep_missing,
                       int32_t* __restrict__ clust, int32_t* __restrict__ stats) {
  extern __shared__ uint8_t s_lab[];  // [ld] cluster labels, 0xFF = none
  __shared__ int s_n[64];
  __shared__ int s_cc[64];
  __shared__ int s_ctl[4];
  __shared__ int s_part[64 * VSEG * 4];
  __shared__ double s_vote[64];
  const int tid = (int)threadIdx.x;
  for (int64_t j = tid; j < ld; j += VT) {
    int lab = 0xFF;
    if (mode == 1 && j < C && clust[j] >= 0) lab = clust[j];
    s_lab[j] = (uint8_t)lab;
  }
  if (mode == 0)
    for (int j = tid; j < C; j += VT) clust[j] = -1;
  if (tid < 64) {
    s_n[tid] = 0;
    s_cc[tid] = 0;
  }
  if (tid == 0) s_ctl[3] = 0;  // changed
  __syncthreads();

  const int nseg = VT / K;
  const int k = tid % K, seg = tid / K;
  const uint32_t kk = (uint32_t)k * 0x01010101u;
  int touch = 0;  // keeps the prefetch loads alive
  for (int t = 0; t < nsteps; ++t) {
    const int target = mode == 0 ? t : order[t];
    const int len = mode == 0 ? t : C;
    const int8_t* row = mat + (int64_t)target * ld;
    const int nchunks = (len + 15) >> 4;
    // the next step's row is known (the visiting order is input): one load per 128-byte line pulls it into L2 while
    // this step runs, so that the scan below does not start with a round trip to HBM
    int pf = 0;
    if (t + 1 < nsteps) {
      const int8_t* nrow = mat + (int64_t)(mode == 0 ? t + 1 : order[t + 1]) * ld;
      const int64_t off = (int64_t)tid * 128;
      if (off < (mode == 0 ? t + 1 : C)) pf = *reinterpret_cast<const int*>(nrow + off);
    }
    {  // integer votes of cluster k from this thread's row segment
      const int cps = (nchunks + nseg - 1) / nseg;
      const int c0 = seg * cps, c1 = min(c0 + cps, nchunks);
      int n = 0;
      if (seg < nseg)
        for (int c = c0; c < c1; ++c) {
          const uint4 g = *reinterpret_cast<const uint4*>(row + 16 * (int64_t)c);
          const uint4 L = *reinterpret_cast<const uint4*>(s_lab + 16 * c);
          n = __builtin_amdgcn_sdot4((int)g.x, (int)byte_eq01(L.x, kk), n, false);
          n = __builtin_amdgcn_sdot4((int)g.y, (int)byte_eq01(L.y, kk), n, false);
          n = __builtin_amdgcn_sdot4((int)g.z, (int)byte_eq01(L.z, kk), n, false);
          n = __builtin_amdgcn_sdot4((int)g.w, (int)byte_eq01(L.w, kk), n, false);
        }
      if (n != 0) atomicAdd(&s_n[k], n);
    }
    __syncthreads();
    if (tid < 64) {
      const int nv = tid < K ? s_n[tid] : INT32_MIN;
      const int mx = wave_max_i32(nv);
      const uint64_t tie = __ballot(tid < K && nv == mx);
      if (tid == 0) {
        s_ctl[0] = (__popcll(tie) == 1) ? (__ffsll((unsigned long long)tie) - 1) : -1;
        s_ctl[1] = (int)(uint32_t)tie;
        s_ctl[2] = (int)(uint32_t)(tie >> 32);
      }
    }
    __syncthreads();
    if (s_ctl[0] < 0) {  // several clusters share the largest integer vote: the jitter decides, exactly
      const uint64_t tie = (uint64_t)(uint32_t)s_ctl[1] | ((uint64_t)(uint32_t)s_ctl[2] << 32);
      const int nt = __popcll(tie);
      const int ti = tid % nt, sg = tid / nt;
      const int nsg = min(VSEG, VT / nt);  // row segments per tied cluster
      int kt = 0;
      {
        uint64_t mm = tie;
        for (int q = 0; q < ti; ++q) mm &= mm - 1;
        kt = __ffsll((unsigned long long)mm) - 1;
      }
      if (sg < nsg) {
        const int cps = (nchunks + nsg - 1) / nsg;
        const int c0 = sg * cps, c1 = min(c0 + cps, nchunks);
        int n = 0, M = 0, m = 0, first = 0;
        for (int c = c0; c < c1; ++c) {
          const uint4 g = *reinterpret_cast<const uint4*>(row + 16 * (int64_t)c);
          if ((g.x | g.y | g.z | g.w) == 0) continue;
          const uint4 L = *reinterpret_cast<const uint4*>(s_lab + 16 * c);
          const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, lw[4] = {L.x, L.y, L.z, L.w};
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int s = (int)(int8_t)(gw[q >> 2] >> (8 * (q & 3)));
            const int lab = (int)((lw[q >> 2] >> (8 * (q & 3))) & 0xFF);
            if (lab == kt && s != 0) {
              n += s;
              M = max(M, n);
              m = min(m, n);
              if (first == 0) first = s;
            }
          }
        }
        int* o = s_part + (ti * VSEG + sg) * 4;
        o[0] = n;
        o[1] = M;
        o[2] = m;
        o[3] = first;
      }
      __syncthreads();
      if (tid < nt) {
        int n = 0, M = 0, m = 0, first = 0;
        for (int q = 0; q < nsg; ++q) {
          const int* o = s_part + (tid * VSEG + q) * 4;
          M = max(M, n + o[1]);
          m = min(m, n + o[2]);
          n += o[0];
          if (first == 0) first = o[3];
        }
        s_vote[tid] = vote_exact(jitter[(int64_t)t * K + kt], n, M, m, first);
      }
      __syncthreads();
      if (tid == 0) {  // :280-287: first maximum in ascending cluster order (strict <)
        uint64_t mm = tie;
        int elected = __ffsll((unsigned long long)mm) - 1;
        double maxvote = s_vote[0];
        mm &= mm - 1;
        for (int q = 1; q < nt; ++q) {
          const int kq = __ffsll((unsigned long long)mm) - 1;
          mm &= mm - 1;
          if (maxvote < s_vote[q]) {
            elected = kq;
            maxvote = s_vote[q];
          }
        }
        s_ctl[0] = elected;
      }
      __syncthreads();
    }
    const int elected = s_ctl[0];
    if (tid == 0) {
      if (mode == 0) {
        s_lab[t] = (uint8_t)elected;
        clust[order[t]] = elected;
        s_cc[elected] += 1;
      } else {
        const int old = s_lab[target];
        if (old != 0xFF || !keep_missing) {  // :333-337
          if (old != elected) s_ctl[3] += 1;
          s_lab[target] = (uint8_t)elected;
          clust[target] = elected;
          s_cc[elected] += 1;
        }
      }
    }
    if (tid < 64) s_n[tid] = 0;
    touch ^= pf;
    __syncthreads();
  }
  if (touch == 0x5a5a5a5a && nsteps < 0) stats[0] = touch;  // never true: the sign bytes are -1, 0, 1
  if (tid == 0) stats[0] = s_ctl[3];
  if (tid < K) stats[1 + tid] = s_cc[tid];
}

}  // namespace

static int fmxold_check(muxgl_handle* h, const char* who, int32_t K) {
  if (!h->fmx_prepared) MUXGL_FAIL(h, "%s: call muxgl_fmx_prepare first", who);
  if (K != 0) {
    if (!h->d_sgn) MUXGL_FAIL(h, "%s: call muxgl_fmxold_pair_dist first", who);
    if (K < 1 || K > 64) MUXGL_FAIL(h, "%s: K=%d outside [1,64]", who, K);
    if (h->sgn_ld + 36 * 1024 > 160 * 1024)
      MUXGL_FAIL(h, "%s: %lld cells exceed the label table the vote kernel keeps in LDS", who, (long long)h->C);
  }
  return 0;
}

static int fmxold_vote_launch(muxgl_handle* h, int mode, int32_t K, int32_t nsteps, const int8_t* mat,
                              const int32_t* d_order, const double* d_jit, int keep, int32_t* d_stats) {
  const size_t lds = (size_t)h->sgn_ld;
  HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(fmxold_vote_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(fmxold_vote_kernel, dim3(1), dim3(VT), lds, h->stream, mode, (int32_t)h->C, h->sgn_ld, (int)K, nsteps,
                     mat, d_order, d_jit, keep, h->d_clust, d_stats);
  HIPCHK(h, hipGetLastError());
  return 0;
}

extern "C" {

int muxgl_fmxold_pair_dist(muxgl_handle* h, double bf_thres, muxgl_dropd* full) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmxold_pair_dist");
  if (h->col) MUXGL_FAIL(h, "muxgl_fmxold_pair_dist: needs the whole pileup on one handle (this one holds slabs)");
  HIPCHK(h, hipSetDevice(h->device));
  if (fmxold_check(h, "muxgl_fmxold_pair_dist", 0)) return 1;
  if (fmx_snp_major_full(h)) return 1;
  const int64_t C = h->C;
  if (C > INT32_MAX / 2) MUXGL_FAIL(h, "muxgl_fmxold_pair_dist: too many cells");
  const int64_t ld = (C + 15) / 16 * 16 + 16;
  h->sgn_ld = ld;
  if (dev_alloc(h, &h->d_sgn, (size_t)(C ? C : 1) * ld)) return 1;
  HIPCHK(h, hipMemsetAsync(h->d_sgn, 0, (size_t)(C ? C : 1) * ld, h->stream));
  muxgl_dropd* d_full = nullptr;
  const size_t npair = (size_t)(C * (C - 1) / 2);
  if (full && npair) {
    if (dev_alloc(h, &d_full, npair)) return 1;
    HIPCHK(h, hipMemsetAsync(d_full, 0, sizeof(muxgl_dropd) * npair, h->stream));
  }
  clear_timing(h);
  tic(h, MUXGL_T_FMXOLD_PAIR);
  if (C > 1) {
    if (d_full) {
      const dim3 grid((unsigned)C, (unsigned)((C + pd_geom<true>::NB - 1) / pd_geom<true>::NB));
      const size_t lds = pd_geom<true>::LDS;
      HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(fmxold_pair_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(fmxold_pair_kernel<true>, grid, dim3(PD_T), lds, h->stream, (int32_t)C, ld, h->d_cell_ptr,
                         h->d_entry_snp, h->d_egls, h->d_ecnt, h->d_af, h->d_snp_ptr, h->d_snp_cell, h->d_segls,
                         h->d_secnt, bf_thres, h->d_sgn, d_full);
    } else {
      const dim3 grid((unsigned)C, (unsigned)((C + pd_geom<false>::NB - 1) / pd_geom<false>::NB));
      const size_t lds = pd_geom<false>::LDS;
      HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(fmxold_pair_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(fmxold_pair_kernel<false>, grid, dim3(PD_T), lds, h->stream, (int32_t)C, ld, h->d_cell_ptr,
                         h->d_entry_snp, h->d_egls, h->d_ecnt, h->d_af, h->d_snp_ptr, h->d_snp_cell, h->d_segls,
                         h->d_secnt, bf_thres, h->d_sgn, d_full);
    }
  }
  toc(h, MUXGL_T_FMXOLD_PAIR);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess && d_full) e = hipMemcpy(full, d_full, sizeof(muxgl_dropd) * npair, hipMemcpyDeviceToHost);
  dev_free(&d_full);
  if (e != hipSuccess) MUXGL_FAIL(h, "muxgl_fmxold_pair_dist: %s", hipGetErrorString(e));
  collect_timing(h);
  return 0;
}

int muxgl_fmxold_get_signs(muxgl_handle* h, int8_t* out) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmxold_get_signs");
  if (h->col) MUXGL_FAIL(h, "muxgl_fmxold_get_signs: needs the whole pileup on one handle (this one holds slabs)");
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->d_sgn || !out) MUXGL_FAIL(h, "muxgl_fmxold_get_signs: no sign matrix (muxgl_fmxold_pair_dist) or NULL output");
  if (h->C)
    HIPCHK(h, hipMemcpy2D(out, (size_t)h->C, h->d_sgn, (size_t)h->sgn_ld, (size_t)h->C, (size_t)h->C,
                          hipMemcpyDeviceToHost));
  return 0;
}

int muxgl_fmxold_vote_init(muxgl_handle* h, int32_t K, const int32_t* order, const double* jitter,
                           double frac_init_clust, int32_t* clust_out, int32_t* ccounts) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmxold_vote_init");
  if (h->col) MUXGL_FAIL(h, "muxgl_fmxold_vote_init: needs the whole pileup on one handle (this one holds slabs)");
  HIPCHK(h, hipSetDevice(h->device));
  if (fmxold_check(h, "muxgl_fmxold_vote_init", K)) return 1;
  const int64_t C = h->C;
  if (C && (!order || !jitter || !clust_out)) MUXGL_FAIL(h, "muxgl_fmxold_vote_init: NULL array");
  int64_t nvis = 0;  // :248: `if ( i > nbcs * fracInitClust ) continue;` -- the visited cells are a prefix
  for (int64_t i = 0; i < C; ++i)
    if (!((double)i > (double)C * frac_init_clust)) ++nvis;
  int32_t *d_order = nullptr, *d_stats = nullptr;
  double* d_jit = nullptr;
  int8_t* d_perm = nullptr;
  auto cleanup = [&]() {
    dev_free(&d_order);
    dev_free(&d_stats);
    dev_free(&d_jit);
    dev_free(&d_perm);
  };
  std::vector<int32_t> stats((size_t)K + 1, 0);
  int rc = 0;
  if (dev_alloc(h, &d_order, (size_t)C) || dev_alloc(h, &d_stats, (size_t)K + 1) ||
      dev_alloc(h, &d_jit, (size_t)nvis * K) || dev_alloc(h, &d_perm, (size_t)(nvis ? nvis : 1) * h->sgn_ld))
    rc = 1;
  hipError_t e = hipSuccess;
  if (!rc && C) e = hipMemcpyAsync(d_order, order, sizeof(int32_t) * C, hipMemcpyHostToDevice, h->stream);
  if (!rc && e == hipSuccess && nvis)
    e = hipMemcpyAsync(d_jit, jitter, sizeof(double) * nvis * K, hipMemcpyHostToDevice, h->stream);
  if (!rc && e == hipSuccess) e = hipMemsetAsync(d_perm, 0, (size_t)(nvis ? nvis : 1) * h->sgn_ld, h->stream);
  clear_timing(h);
  if (!rc && e == hipSuccess) {
    tic(h, MUXGL_T_FMXOLD_VOTE);
    if (nvis)
      hipLaunchKernelGGL(fmxold_permute_kernel, dim3((unsigned)((nvis + 255) / 256), (unsigned)nvis), dim3(256), 0,
                         h->stream, (int32_t)nvis, h->sgn_ld, d_order, h->d_sgn, d_perm);
    rc = fmxold_vote_launch(h, 0, K, (int32_t)nvis, d_perm, d_order, d_jit, 0, d_stats);
    toc(h, MUXGL_T_FMXOLD_VOTE);
  }
  if (!rc && e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (!rc && e == hipSuccess) e = hipMemcpy(stats.data(), d_stats, sizeof(int32_t) * (K + 1), hipMemcpyDeviceToHost);
  if (!rc && e == hipSuccess && C) e = hipMemcpy(clust_out, h->d_clust, sizeof(int32_t) * C, hipMemcpyDeviceToHost);
  cleanup();
  if (rc) return 1;
  if (e != hipSuccess) MUXGL_FAIL(h, "muxgl_fmxold_vote_init: %s", hipGetErrorString(e));
  collect_timing(h);
  if (ccounts)
    for (int j = 0; j < K; ++j) ccounts[j] = stats[(size_t)j + 1];
  return 0;
}

int muxgl_fmxold_vote_refine(muxgl_handle* h, int32_t K, const int32_t* order, const double* jitter,
                             int32_t keep_init_missing, int32_t* clust_inout, int32_t* changed, int32_t* ccounts) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_fmxold_vote_refine");
  if (h->col) MUXGL_FAIL(h, "muxgl_fmxold_vote_refine: needs the whole pileup on one handle (this one holds slabs)");
  HIPCHK(h, hipSetDevice(h->device));
  if (fmxold_check(h, "muxgl_fmxold_vote_refine", K)) return 1;
  const int64_t C = h->C;
  if (C && (!order || !jitter || !clust_inout)) MUXGL_FAIL(h, "muxgl_fmxold_vote_refine: NULL array");
  for (int64_t i = 0; i < C; ++i) {
    if (clust_inout[i] >= K) MUXGL_FAIL(h, "muxgl_fmxold_vote_refine: cell %lld has cluster %d >= K", (long long)i, clust_inout[i]);
    if (order[i] < 0 || order[i] >= C) MUXGL_FAIL(h, "muxgl_fmxold_vote_refine: order[%lld] out of range", (long long)i);
  }
  int32_t *d_order = nullptr, *d_stats = nullptr;
  double* d_jit = nullptr;
  auto cleanup = [&]() {
    dev_free(&d_order);
    dev_free(&d_stats);
    dev_free(&d_jit);
  };
  std::vector<int32_t> stats((size_t)K + 1, 0);
  int rc = 0;
  if (dev_alloc(h, &d_order, (size_t)C) || dev_alloc(h, &d_stats, (size_t)K + 1) || dev_alloc(h, &d_jit, (size_t)C * K))
    rc = 1;
  hipError_t e = hipSuccess;
  if (!rc && C) {
    e = hipMemcpyAsync(d_order, order, sizeof(int32_t) * C, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_jit, jitter, sizeof(double) * C * K, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(h->d_clust, clust_inout, sizeof(int32_t) * C, hipMemcpyHostToDevice, h->stream);
  }
  clear_timing(h);
  if (!rc && e == hipSuccess) {
    tic(h, MUXGL_T_FMXOLD_VOTE);
    rc = fmxold_vote_launch(h, 1, K, (int32_t)C, h->d_sgn, d_order, d_jit, keep_init_missing ? 1 : 0, d_stats);
    toc(h, MUXGL_T_FMXOLD_VOTE);
  }
  if (!rc && e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (!rc && e == hipSuccess) e = hipMemcpy(stats.data(), d_stats, sizeof(int32_t) * (K + 1), hipMemcpyDeviceToHost);
  if (!rc && e == hipSuccess && C) e = hipMemcpy(clust_inout, h->d_clust, sizeof(int32_t) * C, hipMemcpyDeviceToHost);
  cleanup();
  if (rc) return 1;
  if (e != hipSuccess) MUXGL_FAIL(h, "muxgl_fmxold_vote_refine: %s", hipGetErrorString(e));
  collect_timing(h);
  if (changed) *changed = stats[0];
  if (ccounts)
    for (int j = 0; j < K; ++j) ccounts[j] = stats[(size_t)j + 1];
  return 0;
}

}  // extern "C"
