// fmx_oct.hip -- freemuxlet E-step (cmd_cram_freemux2.cpp:383-456) for K <= 16 clusters with the eight-lanes-per-entry
// tiling of demux_oct.hip (oct_tiling.hpp): lane p of an entry owns clusters p and p + 8, eight entries per wave step.
//
// Per entry the reference evaluates, for every cluster pair k < j,  lk = sum_{g1,g2} glis[g1][g2] gp_j[g1] gp_k[g2]
// (:440-446) and for every cluster  lk = sum_g glis[g][g] gp_j[g]  (:448-452), and adds log(lk) to llks (:454-455).
// glis (calculate_snp_droplet_pileup, alpha = 0.5) is symmetric, so unordered pairs suffice.  Here:
//   u[m] = sum_l gp_c[l] * glis[l][m]  for the lane's two clusters, then every pair costs 3 FMA + 1 multiply into a
//   product accumulator.  Products leave as (mantissa, exponent) per chunk; fmx_oct_reduce_kernel takes one log per
//   (cell, pair).
// Entries whose likelihoods are linear in the genotypes (fmx_entry_kernel: glis[g1][g2] = c0 + c1 (g1 + g2)) are swept by
// a loop of their own from the clusters' moments E = g1 + 2 g2 (fmx_ceo_kernel): pair (c0 + c1 E_j) + c1 E_k, singlet
// c0 + 2 c1 E_j -- an FMA and the product update per hypothesis, a row of 128 bytes (ONE line per entry).
// Both loops are software pipelines without LDS and without barriers: the entries of a unit (eight chunks that are
// neighbours in the launch order) are laid out step-major at muxgl_fmx_prepare time (fo_repack_*), so a step of the wave
// reads eight consecutive records; records three steps ahead, rows two ahead.  (The four-lanes-per-entry kernel this
// replaces, fmx_estep_quad_kernel of rounds 1-2, took 1.52 ms per 47.9 M entries at configs[3]: its row gathers and its
// arithmetic did not overlap at two waves per SIMD -- demux_oct.hip has the measurements.)
#include <vector>

#include "oct_tiling.hpp"

using namespace oct;

namespace {

constexpr int FO_LPAD = 3, FO_GPAD = 3;  // neutral steps behind a unit's longest list (the loops read ahead)

// E = g1 + 2 g2 of the cluster posteriors, (E_p, E_p+8) adjacent: [S + 1][8][2].  Clusters >= K and the neutral row S: 0.
__global__ void __launch_bounds__(256) fmx_ceo_kernel(int64_t S, int K, const double* __restrict__ cgp, double* __restrict__ ceo) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (S + 1) * 16) return;
  const int64_t s = tid >> 4;
  const int w = (int)(tid & 15), j = (w >> 1) + 8 * (w & 1);
  double E = 0.0;
  if (s < S && j < K) {
    const double* g = cgp + ((size_t)s * K + j) * 3;
    E = fma(2.0, g[2], g[1]);
  }
  ceo[tid] = E;
}

// cluster-GP rows [S][K][3] -> [S + 1][3][8][2]: lane p's six doubles (cluster p: l = 0, 1, 2; cluster p + 8: l = 0, 1, 2) as
// three 16-byte pieces, piece t of the eight lanes contiguous (a whole line).  Clusters >= K and the neutral row S are
// (1, 0, 0): their factors are exactly 1.
__global__ void __launch_bounds__(256) fmx_cgpo_kernel(int64_t S, int K, const double* __restrict__ cgp, double* __restrict__ cgpo) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (S + 1) * 48) return;
  const int64_t s = tid / 48;
  const int w = (int)(tid - s * 48);  // ((t * 8 + p) * 2 + half)
  const int half = w & 1, p = (w >> 1) & 7, t = w >> 4;
  const int d = 2 * t + half, j = p + 8 * (d / 3), l = d % 3;
  cgpo[tid] = (s < S && j < K) ? cgp[((size_t)s * K + j) * 3 + l] : (l == 0 ? 1.0 : 0.0);
}

// number of linear entries per chunk (flin == NULL: that form is off, every entry takes the nine-term loop)
__global__ void __launch_bounds__(64)
    fo_count_kernel(int n_chunks, const row_chunk* __restrict__ chunks, const uint32_t* __restrict__ flin, int32_t* __restrict__ nlin) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_chunks) return;
  int w = 0;
  if (flin)
    for (int64_t e = chunks[q].e0, e1 = e + chunks[q].len; e < e1; ++e) w += (flin[e >> 5] >> (e & 31)) & 1u;
  nlin[q] = w;
}

// The entries of every unit, step-major: loff / lc[lptr[u] + i * 8 + slot] the i-th linear entry of the unit's slot-th
// chunk (row offset; (c0, c1)), goff / ggl[gptr[u] + i * 8 + slot] its i-th other entry (row offset; six likelihoods
// {00,11,22,01,02,12}); neutral records (the dummy row S, likelihoods of 1 resp. c0 = 1, c1 = 0: every factor exactly 1)
// behind the end of a list.
__global__ void __launch_bounds__(256)
    fo_repack_kernel(int n_units, int n_chunks, const row_chunk* __restrict__ chunks, const int32_t* __restrict__ order,
                     const uint32_t* __restrict__ flin, const int32_t* __restrict__ entry_snp, const double* __restrict__ egls,
                     const int32_t* __restrict__ lsteps, const int64_t* __restrict__ lptr, const int32_t* __restrict__ gsteps,
                     const int64_t* __restrict__ gptr, uint32_t S, uint32_t* __restrict__ loff, double2* __restrict__ lc,
                     uint32_t* __restrict__ goff, double* __restrict__ ggl) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int u = t / SLOTS, slot = t % SLOTS;
  if (u >= n_units) return;
  const int w = u * SLOTS + slot;
  int64_t e0 = 0;
  int len = 0;
  if (w < n_chunks) {
    const int q = order ? order[w] : w;
    e0 = chunks[q].e0;
    len = chunks[q].len;
  }
  uint32_t* lo = loff + lptr[u] + slot;
  double2* ld = lc + lptr[u] + slot;
  uint32_t* od = goff + gptr[u] + slot;
  double* gd = ggl + (gptr[u] + slot) * 6;
  int il = 0, ig = 0;
  for (int i = 0; i < len; ++i) {
    const int64_t e = e0 + i;
    const double* g9 = egls + (size_t)e * 9;  // the six distinct values {00,11,22,01,02,12} of the symmetric matrix
    const double g[6] = {g9[0], g9[4], g9[8], g9[1], g9[2], g9[5]};
    if (flin && ((flin[e >> 5] >> (e & 31)) & 1u)) {
      lo[(size_t)il * SLOTS] = (uint32_t)entry_snp[e] * 128u;
      ld[(size_t)il * SLOTS] = double2{g[0], g[3] - g[0]};
      ++il;
    } else {
      od[(size_t)ig * SLOTS] = (uint32_t)entry_snp[e] * 384u;
#pragma unroll
      for (int k = 0; k < 6; ++k) gd[(size_t)ig * SLOTS * 6 + k] = g[k];
      ++ig;
    }
  }
  for (const int n = lsteps[u]; il < n; ++il) {
    lo[(size_t)il * SLOTS] = S * 128u;
    ld[(size_t)il * SLOTS] = double2{1.0, 0.0};
  }
  for (const int n = gsteps[u]; ig < n; ++ig) {
    od[(size_t)ig * SLOTS] = S * 384u;
#pragma unroll
    for (int k = 0; k < 6; ++k) gd[(size_t)ig * SLOTS * 6 + k] = 1.0;
  }
}

#ifndef FO_WAVES
#define FO_WAVES 4  // waves per SIMD the register allocation aims at (128 VGPRs; 3: 136 VGPRs, sweep +10 % at configs[3])
#endif
__global__ void __launch_bounds__(64, FO_WAVES)
    fmx_estep_oct_kernel(int n_chunks, const int32_t* __restrict__ order, const int32_t* __restrict__ lsteps,
                         const int64_t* __restrict__ lptr, const uint32_t* __restrict__ loff,
                         const double2* __restrict__ lc,
                         const int32_t* __restrict__ gsteps, const int64_t* __restrict__ gptr,
                         const uint32_t* __restrict__ goff, const double* __restrict__ ggl, const double* __restrict__ cgpo,
                         const double* __restrict__ ceo, double* __restrict__ part_m, int32_t* __restrict__ part_e) {
  const int lane = threadIdx.x;
  const int p = (lane >> 1) & 7;                    // position: clusters p and p + 8
  const int slot = ((lane >> 4) << 1) | (lane & 1);  // 8 entry streams per wave
  const uint32_t p16 = (uint32_t)p * 16u;
  const int unit = xcd_swizzle(blockIdx.x, gridDim.x >> 3);
  const int wq = unit * SLOTS + slot;
  const int q = wq < n_chunks ? (order ? order[wq] : wq) : n_chunks;

  double acc[N_ACC];
  int32_t ex[N_ACC];
#pragma unroll
  for (int a = 0; a < N_ACC; ++a) {
    acc[a] = 1.0;
    ex[a] = 0;
  }
  auto renorm = [&]() {
#pragma unroll
    for (int a = 0; a < N_ACC; ++a) prodacc_renorm(acc[a], ex[a]);
  };

  // ---- the linear entries: glis[g1][g2] = c0 + c1 (g1 + g2), taken at s = 1 (posteriors normalised in FP64, see
  //      fmx_wave.hip): singlet c0 + 2 c1 E_j, pair (c0 + c1 E_j) + c1 E_k with E = g1 + 2 g2
  const int nL = lsteps[unit];  // a multiple of 3 plus the read-ahead
  if (nL > 0) {
    struct rowl_t {
      double Ea, Eb;
    };
    auto load_rowl = [&](rowl_t& R, uint32_t row_off) {
      const double2 v = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(ceo) + (size_t)(row_off + p16));
      R.Ea = v.x;
      R.Eb = v.y;
    };
    const uint32_t* lo = loff + lptr[unit] + slot;
    const double2* lcs = lc + lptr[unit] + slot;
    auto sweepL = [&](const rowl_t& R, const double2& rc) {
      const double c0 = rc.x, c1 = rc.y, c2 = c1 + c1;
      acc[acc_single(0)] *= fma(c2, R.Ea, c0);  // singlet (:448-452)
      acc[acc_single(1)] *= fma(c2, R.Eb, c0);
      const double Xa = fma(c1, R.Ea, c0), Xb = fma(c1, R.Eb, c0);
      acc[ACC_AB] *= fma(c1, R.Eb, Xa);  // :440-446
#define FO_ROTL(T, CTRL)                                                    \
  {                                                                         \
    const double Pa = dpp_rot<CTRL>(R.Ea), Pb = dpp_rot<CTRL>(R.Eb);        \
    acc[acc_rot(T, 0, 0)] *= fma(c1, Pa, Xa);                               \
    acc[acc_rot(T, 0, 1)] *= fma(c1, Pb, Xa);                               \
    acc[acc_rot(T, 1, 0)] *= fma(c1, Pa, Xb);                               \
    acc[acc_rot(T, 1, 1)] *= fma(c1, Pb, Xb);                               \
  }
      FO_ROTL(1, ROR2)
      FO_ROTL(2, ROR4)
      FO_ROTL(3, ROR6)
#undef FO_ROTL
      {  // the lane facing this one: (a, b') here and (a', b) over there; (a, a') and (b, b') on both sides
        const double Pa = dpp_rot<ROR8>(R.Ea), Pb = dpp_rot<ROR8>(R.Eb);
        acc[ACC_F_AB] *= fma(c1, Pb, Xa);
        acc[ACC_F_AA] *= fma(c1, Pa, Xa);
        acc[ACC_F_BB] *= fma(c1, Pb, Xb);
      }
    };
    // entry i: row Rc, (c0, c1) in cc; o0 held its row offset (free), o2 holds that of i + 2; cn receives i + 1's (c0, c1)
    auto step = [&](int i, const rowl_t& Rc, rowl_t& Rnn, const double2& cc, double2& cn, uint32_t& o0, uint32_t o2) {
      // Issue order = completion order (vmcnt): the row offset, which the NEXT step's first load needs, goes first and
      // the gathered row, which nobody reads for two sweeps, last -- so that waiting for the offset (vmcnt(2)) and for
      // the next entry's (c0, c1) (vmcnt(4)) leaves the gather in flight.  (Round 5: the gather used to be issued first
      // and the offset last; the next step's wait for the offset was vmcnt(0) and gave the gather one sweep instead of two.)
      o0 = lo[(size_t)(i + 3) * SLOTS];
      cn = lcs[(size_t)(i + 1) * SLOTS];
      load_rowl(Rnn, o2);
      __builtin_amdgcn_sched_barrier(0);  // the loads are issued in front of the sweep they hide behind
      sweepL(Rc, cc);
      __builtin_amdgcn_sched_barrier(0);
    };
    rowl_t L0, L1, L2;
    double2 c0v = lcs[0], c1v;
    uint32_t oa = lo[0], ob = lo[SLOTS], oc = lo[2 * SLOTS];
    load_rowl(L0, oa);
    load_rowl(L1, ob);
    int since = 0;
    for (int i = 0; i + FO_LPAD < nL; i += 6) {  // (rings of three row sets and two (c0, c1) sets: six steps written out)
      step(i, L0, L2, c0v, c1v, oa, oc);
      step(i + 1, L1, L0, c1v, c0v, ob, oa);
      step(i + 2, L2, L1, c0v, c1v, oc, ob);
      step(i + 3, L0, L2, c1v, c0v, oa, oc);
      step(i + 4, L1, L0, c0v, c1v, ob, oa);
      step(i + 5, L2, L1, c1v, c0v, oc, ob);
      // a factor of a linear entry is > 2^-27 (both ends of the line within 1e-7 of each other, fmx_entry_kernel):
      // 24 of them between two renormalisations cannot underflow
      if (++since == 4) {
        since = 0;
        renorm();
      }
    }
    renorm();
  }

  // ---- the other entries: nine-term form.  Row offsets three steps ahead, rows two ahead, likelihoods one ahead ----
  const int nG = gsteps[unit];
  if (nG > 0) {
    struct row_t {
      double a[3], b[3];  // posterior triples of clusters p and p + 8
    };
    auto load_row = [&](row_t& R, uint32_t off) {
      const double2* pc = reinterpret_cast<const double2*>(reinterpret_cast<const char*>(cgpo) + (size_t)(off + p16));
      const double2 v0 = pc[0], v1 = pc[8], v2 = pc[16];
      R.a[0] = v0.x;
      R.a[1] = v0.y;
      R.a[2] = v1.x;
      R.b[0] = v1.y;
      R.b[1] = v2.x;
      R.b[2] = v2.y;
    };
    struct gl_t {
      double v[6];  // glis {00, 11, 22, 01, 02, 12}
    };
    const uint32_t* go = goff + gptr[unit] + slot;
    const double* gg = ggl + (gptr[unit] + slot) * 6;
    auto fetch_gl = [&](gl_t& g, int i) {
      const double2* src = reinterpret_cast<const double2*>(gg + (size_t)i * SLOTS * 6);
      const double2 x0 = src[0], x1 = src[1], x2 = src[2];
      g.v[0] = x0.x, g.v[1] = x0.y, g.v[2] = x1.x, g.v[3] = x1.y, g.v[4] = x2.x, g.v[5] = x2.y;
    };
    auto sweep = [&](const row_t& R, const gl_t& g) {
      const double p0 = g.v[0], p4 = g.v[1], p8 = g.v[2], p1 = g.v[3], p2 = g.v[4], p5 = g.v[5];  // glis[g1][g2] == glis[g2][g1]
      acc[acc_single(0)] *= fma(R.a[2], p8, fma(R.a[1], p4, R.a[0] * p0));  // singlet (:448-452)
      acc[acc_single(1)] *= fma(R.b[2], p8, fma(R.b[1], p4, R.b[0] * p0));
      double ua[3], ub[3];
      ua[0] = fma(R.a[2], p2, fma(R.a[1], p1, R.a[0] * p0));
      ua[1] = fma(R.a[2], p5, fma(R.a[1], p4, R.a[0] * p1));
      ua[2] = fma(R.a[2], p8, fma(R.a[1], p5, R.a[0] * p2));
      ub[0] = fma(R.b[2], p2, fma(R.b[1], p1, R.b[0] * p0));
      ub[1] = fma(R.b[2], p5, fma(R.b[1], p4, R.b[0] * p1));
      ub[2] = fma(R.b[2], p8, fma(R.b[1], p5, R.b[0] * p2));
      auto dot = [](const double* gq, const double* u) { return fma(gq[2], u[2], fma(gq[1], u[1], gq[0] * u[0])); };  // :440-446
      acc[ACC_AB] *= dot(R.b, ua);
#define FO_ROT(T, CTRL)                                                                             \
  {                                                                                                 \
    double Pa[3], Pb[3];                                                                            \
    Pa[0] = dpp_rot<CTRL>(R.a[0]);                                                                  \
    Pa[1] = dpp_rot<CTRL>(R.a[1]);                                                                  \
    Pa[2] = dpp_rot<CTRL>(R.a[2]);                                                                  \
    Pb[0] = dpp_rot<CTRL>(R.b[0]);                                                                  \
    Pb[1] = dpp_rot<CTRL>(R.b[1]);                                                                  \
    Pb[2] = dpp_rot<CTRL>(R.b[2]);                                                                  \
    acc[acc_rot(T, 0, 0)] *= dot(Pa, ua);                                                           \
    acc[acc_rot(T, 0, 1)] *= dot(Pb, ua);                                                           \
    acc[acc_rot(T, 1, 0)] *= dot(Pa, ub);                                                           \
    acc[acc_rot(T, 1, 1)] *= dot(Pb, ub);                                                           \
  }
      FO_ROT(1, ROR2)
      FO_ROT(2, ROR4)
      FO_ROT(3, ROR6)
#undef FO_ROT
      {
        double Pa[3], Pb[3];
        Pa[0] = dpp_rot<ROR8>(R.a[0]);
        Pa[1] = dpp_rot<ROR8>(R.a[1]);
        Pa[2] = dpp_rot<ROR8>(R.a[2]);
        Pb[0] = dpp_rot<ROR8>(R.b[0]);
        Pb[1] = dpp_rot<ROR8>(R.b[1]);
        Pb[2] = dpp_rot<ROR8>(R.b[2]);
        acc[ACC_F_AB] *= dot(Pb, ua);
        acc[ACC_F_AA] *= dot(Pa, ua);
        acc[ACC_F_BB] *= dot(Pb, ub);
      }
    };
    // entry i: row Rc, likelihoods gc; o0 held its row offset (free), o2 holds that of i + 2; gn receives i + 1's likelihoods
    auto step = [&](int i, const row_t& Rc, row_t& Rnn, const gl_t& gc, gl_t& gn, uint32_t& o0, uint32_t o2) {
      o0 = go[(size_t)(i + 3) * SLOTS];  // (offset first, the gathered rows last: see the linear loop)
      fetch_gl(gn, i + 1);
      load_row(Rnn, o2);
      __builtin_amdgcn_sched_barrier(0);
      sweep(Rc, gc);
      __builtin_amdgcn_sched_barrier(0);
    };
    row_t R0, R1, R2;
    gl_t g0, g1;
    uint32_t oa = go[0], ob = go[SLOTS], oc = go[2 * SLOTS];
    load_row(R0, oa);
    load_row(R1, ob);
    fetch_gl(g0, 0);
    int since = 0;
    for (int i = 0; i + FO_GPAD < nG; i += 6) {  // (rings of three row sets and two likelihood sets: six steps written out)
      step(i, R0, R2, g0, g1, oa, oc);
      step(i + 1, R1, R0, g1, g0, ob, oa);
      step(i + 2, R2, R1, g0, g1, oc, ob);
      step(i + 3, R0, R2, g1, g0, oa, oc);
      step(i + 4, R1, R0, g0, g1, ob, oa);
      step(i + 5, R2, R1, g1, g0, oc, ob);
      // a likelihood is >= 1e-6 / 9 after the clamp (sc_drop_seq.cpp:498-506) and the posteriors sum to 1: a factor is
      // > 2^-24; 18 of them between two renormalisations
      if (++since == 3) {
        since = 0;
        renorm();
      }
    }
    renorm();
  }

  if (q < n_chunks) {
#pragma unroll
    for (int a = 0; a < N_ACC; ++a) {
      part_m[((size_t)q * N_ACC + a) * 8 + p] = acc[a];
      part_e[((size_t)q * N_ACC + a) * 8 + p] = ex[a];
    }
  }
}

__global__ void __launch_bounds__(192)
    fmx_oct_reduce_kernel(const int64_t* __restrict__ cell_chunk_ptr, const int32_t* __restrict__ cell_chunks,
                          const double* __restrict__ part_m, const int32_t* __restrict__ part_e,
                          const int32_t* __restrict__ pmap, int K, int64_t c_off, double* __restrict__ fll) {
  const int64_t c = c_off + blockIdx.x;
  const int64_t c0 = cell_chunk_ptr[c], c1 = cell_chunk_ptr[c + 1];
  const int npairs = K * (K + 1) / 2;
  const int idx = threadIdx.x;
  if (idx >= N_HYP) return;
  int j, k;
  if (!hypothesis_of(idx, pmap, j, k)) return;
  if (k < 0) k = j;  // a singlet sits on the diagonal of the packed triangle
  if (j >= K || k >= K) return;
  double m = 1.0;
  int64_t e = 0;
  int cnt = 0;
  for (int64_t ci = c0; ci < c1; ++ci) {
    const size_t o = (size_t)cell_chunks[ci] * N_HYP + idx;
    m *= part_m[o];
    e += part_e[o];
    if (++cnt == 512) {
      cnt = 0;
      int ee;
      m = frexp(m, &ee);
      e += ee;
    }
  }
  const int hi = j > k ? j : k, lo = j > k ? k : j;
  fll[(size_t)c * npairs + hi * (hi + 1) / 2 + lo] = (c0 == c1) ? 0.0 : pos_log(m, (double)e);
}

// prefix sums of 8 x steps (host: a few thousand units, once per muxgl_fmx_prepare)
int fo_unit_ptr(muxgl_handle* h, const int32_t* d_steps, int n_units, int64_t** d_ptr, int64_t* total) {
  std::vector<int32_t> steps((size_t)n_units);
  std::vector<int64_t> ptr((size_t)n_units + 1, 0);
  HIPCHK(h, hipMemcpyAsync(steps.data(), d_steps, sizeof(int32_t) * (size_t)n_units, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int u = 0; u < n_units; ++u) ptr[(size_t)u + 1] = ptr[(size_t)u] + (int64_t)steps[(size_t)u] * SLOTS;
  if (dev_alloc(h, d_ptr, ptr.size())) return 1;
  HIPCHK(h, hipMemcpy(*d_ptr, ptr.data(), sizeof(int64_t) * ptr.size(), hipMemcpyHostToDevice));
  *total = ptr.back();
  return 0;
}

}  // namespace

// oct E-step for the cell shard [c0, c0+nc) described by the chunk tables st; -1 if not applicable
int fmx_oct_estep_launch(muxgl_handle* h, muxgl_row_state* st, int64_t c0, int64_t nc) {
  if (h->K > 16 || !st || (h->flags & (MUXGL_FLAG_FORCE_TILE_SWEEP | MUXGL_FLAG_FORCE_ROW_KERNEL))) return -1;
  if (h->S + 1 >= ((int64_t)1 << 23)) return -1;  // (32-bit byte offsets of the rows)
  if (!st->d_tmap) {
    if (dev_alloc(h, &st->d_tmap, 32)) return 1;
    hipLaunchKernelGGL(pmap_kernel, dim3(1), dim3(64), 0, h->stream, st->d_tmap);
    HIPCHK(h, hipGetLastError());
  }
  const size_t need = (size_t)st->n_chunks * N_HYP;
  if (need > st->part_cap) {
    if (dev_alloc(h, &st->d_part, need)) return 1;
    st->part_cap = need;
  }
  if (need > st->part_e_cap) {
    if (dev_alloc(h, &st->d_part_e, need)) return 1;
    st->part_e_cap = need;
  }
  const unsigned blocks = (unsigned)((((st->n_chunks + SLOTS - 1) / SLOTS) + 7) / 8 * 8);  // units (multiple of 8 for xcd_swizzle)
  const bool use_lin = h->d_flin && h->nnz > 0 && !(h->flags & MUXGL_FLAG_NO_LINEAR_ENTRIES);
  if (!st->d_fq_nlin && st->n_chunks) {  // once per muxgl_fmx_prepare and chunk table: the units' entries, step-major
    const uint32_t* flin = use_lin ? h->d_flin : nullptr;
    if (dev_alloc(h, &st->d_fq_nlin, (size_t)st->n_chunks) || dev_alloc(h, &st->d_fo_lsteps, (size_t)blocks) ||
        dev_alloc(h, &st->d_fo_gsteps, (size_t)blocks))
      return 1;
    hipLaunchKernelGGL(fo_count_kernel, dim3((unsigned)((st->n_chunks + 63) / 64)), dim3(64), 0, h->stream, (int)st->n_chunks,
                       st->d_chunks, flin, st->d_fq_nlin);
    HIPCHK(h, hipGetLastError());
    if (quad_launch_order(h, st->d_chunks, st->d_fq_nlin, st->n_chunks, &st->d_fq_order)) return 1;
    const unsigned ub = (blocks + 255) / 256;
    hipLaunchKernelGGL(unit_steps_kernel, dim3(ub), dim3(256), 0, h->stream, (int)blocks, (int)st->n_chunks, st->d_fq_order,
                       st->d_fq_nlin, st->d_chunks, 0, 6, FO_LPAD, st->d_fo_lsteps);
    hipLaunchKernelGGL(unit_steps_kernel, dim3(ub), dim3(256), 0, h->stream, (int)blocks, (int)st->n_chunks, st->d_fq_order,
                       st->d_fq_nlin, st->d_chunks, 1, 6, FO_GPAD, st->d_fo_gsteps);
    HIPCHK(h, hipGetLastError());
    int64_t nl_total = 0, ng_total = 0;
    if (fo_unit_ptr(h, st->d_fo_lsteps, (int)blocks, &st->d_fo_lptr, &nl_total) ||
        fo_unit_ptr(h, st->d_fo_gsteps, (int)blocks, &st->d_fo_gptr, &ng_total))
      return 1;
    if (dev_alloc(h, &st->d_fo_loff, (size_t)nl_total + 1) || dev_alloc(h, &st->d_fo_lc, (size_t)nl_total + 1) ||
        dev_alloc(h, &st->d_fo_goff, (size_t)ng_total + 1) || dev_alloc(h, &st->d_fo_ggl, ((size_t)ng_total + 1) * 6))
      return 1;
    hipLaunchKernelGGL(fo_repack_kernel, dim3((unsigned)(((size_t)blocks * SLOTS + 255) / 256)), dim3(256), 0, h->stream,
                       (int)blocks, (int)st->n_chunks, st->d_chunks, st->d_fq_order, flin, h->d_entry_snp, h->d_egls,
                       st->d_fo_lsteps, st->d_fo_lptr, st->d_fo_gsteps, st->d_fo_gptr, (uint32_t)h->S,
                       st->d_fo_loff, st->d_fo_lc, st->d_fo_goff, st->d_fo_ggl);
    HIPCHK(h, hipGetLastError());
  }
  // this iteration's cluster posteriors in the two row formats
  const size_t nq = ((size_t)h->S + 1) * 48, ne = ((size_t)h->S + 1) * 16;
  if (nq > h->cgpq_cap) {
    if (dev_alloc(h, &h->d_cgpq, nq)) return 1;
    h->cgpq_cap = nq;
  }
  if (ne > h->ceq_cap) {
    if (dev_alloc(h, &h->d_ceq, ne)) return 1;
    h->ceq_cap = ne;
  }
  hipLaunchKernelGGL(fmx_cgpo_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, h->stream, h->S, h->K, h->d_cgp, h->d_cgpq);
  hipLaunchKernelGGL(fmx_ceo_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, h->stream, h->S, h->K, h->d_cgp, h->d_ceq);
  tic(h, MUXGL_T_FMX_ESTEP_SWEEP);
  if (blocks)
    hipLaunchKernelGGL(fmx_estep_oct_kernel, dim3(blocks), dim3(64), 0, h->stream, (int)st->n_chunks, st->d_fq_order,
                       st->d_fo_lsteps, st->d_fo_lptr, st->d_fo_loff, st->d_fo_lc, st->d_fo_gsteps,
                       st->d_fo_gptr, st->d_fo_goff, st->d_fo_ggl, h->d_cgpq, h->d_ceq, st->d_part, st->d_part_e);
  toc(h, MUXGL_T_FMX_ESTEP_SWEEP);
  if (nc > 0)
    hipLaunchKernelGGL(fmx_oct_reduce_kernel, dim3((unsigned)nc), dim3(192), 0, h->stream, st->d_cell_chunk_ptr,
                       st->d_cell_chunks, st->d_part, st->d_part_e, st->d_tmap, h->K, c0, h->d_fll);
  HIPCHK(h, hipGetLastError());
  return 0;
}
