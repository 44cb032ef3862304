// fmx_greedy.hip -- freemux2's greedy initial clustering (cmd_cram_freemux2.cpp:217-261 with
// calculate_droplet_clust_distance, sc_drop_seq.cpp:544-578, and merge(), sc_drop_seq.h:77-101) on the device.
//
// The algorithm is sequential across cells by construction: cell i (in score order) joins the cluster that maximises
// sum_snp [log lk2 - log lk0] against the pileups built from cells 0..i-1, and is merged into it before cell i+1 is
// looked at.  What is parallel is the inside of one step -- L entries x K clusters independent likelihood terms, then L
// independent merges -- so ONE persistent 1024-thread workgroup walks the cell list (a grid-wide barrier per cell
// would cost more than the step itself):
//   * distance: thread = (cluster j, entry stripe); the K diagonal triples of a SNP sit in one 32*K-byte row of a
//     SNP-major table, so an entry costs K/4 cache lines instead of K gathers into [K][S][9]; per-thread products
//     (mantissa, exponent) instead of two log's per term, a tree over the stripes in LDS, two log's per cluster;
//   * argmax: strict `>` from cluster 0 (:235-242);
//   * merge: thread = entry, the full nine-value state of (SNP, winner) is updated in the reference's operation order
//     (multiply, normalise, clamp at 1e-6, normalise) and its diagonal copied to the distance table.
// The per-(cluster, SNP) states built here are discarded afterwards, exactly as in the reference, which rebuilds the
// cluster pileups from the assignment in ascending cell order (:277-288 -> muxgl_fmx_set_clusters).
//
// Measured on MI355X: 10 k cells x 16 clusters (9.5 M entries) ...  see DESIGN.md section 4.2.
#include <algorithm>
#include <vector>

#include "common.hpp"

namespace {

constexpr double kMinNormGL = 1e-6;  // sc_drop_seq.h:14
constexpr int GT = 1024;             // threads of the persistent workgroup

// diag[snp][j] = {gl00, gl11, gl22, present}; full[snp][j][9]
__global__ void __launch_bounds__(GT)
    fmx_greedy_kernel(const int32_t* __restrict__ order, int64_t n_order, const int64_t* __restrict__ cell_ptr,
                      const int32_t* __restrict__ entry_snp, const double* __restrict__ egls,
                      const double* __restrict__ af, int K, int Kp /* K rounded up to a power of two */,
                      double* diag, double* full, int32_t* __restrict__ clust) {
  __shared__ double sm2[GT], sm0[GT];
  __shared__ int32_t se2[GT], se0[GT];
  __shared__ double score[256];
  __shared__ int winner;
  const int t = threadIdx.x;
  const int j = t & (Kp - 1);
  const int stripe = t / Kp, nstripes = GT / Kp;
  for (int64_t oi = 0; oi < n_order; ++oi) {
    const int32_t cell = order[oi];
    const int64_t e0 = cell_ptr[cell], e1 = cell_ptr[cell + 1];
    // ---- distance to every cluster
    double m2 = 1.0, m0 = 1.0;
    int32_t x2 = 0, x0 = 0;
    if (j < K) {
      int cnt = 0;
      for (int64_t e = e0 + stripe; e < e1; e += nstripes) {
        const int32_t snp = entry_snp[e];
        const double4 d = *reinterpret_cast<const double4*>(diag + ((size_t)snp * K + j) * 4);
        if (d.w == 0.0) continue;  // the (cluster, SNP) key does not exist yet (sc_drop_seq.cpp:549-550)
        const double a = af[snp];
        const double gps[3] = {(1.0 - a) * (1.0 - a), 2.0 * a * (1.0 - a), a * a};
        const double* gl = egls + (size_t)e * 9;
        const double gi[3] = {gl[0], gl[4], gl[8]};
        const double gj[3] = {d.x, d.y, d.z};
        double lk0 = 0, lk2 = 0;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          lk2 += (gi[p] * gj[p] * gps[p]);
#pragma unroll
          for (int q = 0; q < 3; ++q) lk0 += (gi[p] * gj[q] * gps[p] * gps[q]);
        }
        m2 *= lk2;
        m0 *= lk0;
        if (++cnt == 4) {  // a term is >= ~1e-30 (clamped likelihoods x HWE priors): four cannot underflow
          cnt = 0;
          prodacc_renorm(m2, x2);
          prodacc_renorm(m0, x0);
        }
      }
      prodacc_renorm(m2, x2);
      prodacc_renorm(m0, x0);
    }
    sm2[t] = m2;
    sm0[t] = m0;
    se2[t] = x2;
    se0[t] = x0;
    __syncthreads();
    for (int s = nstripes >> 1; s > 0; s >>= 1) {  // stripes of one cluster are Kp threads apart
      if (stripe < s) {
        const int o = t + s * Kp;
        double a2 = sm2[t] * sm2[o], a0 = sm0[t] * sm0[o];
        int32_t b2 = se2[t] + se2[o], b0 = se0[t] + se0[o];
        prodacc_renorm(a2, b2);
        prodacc_renorm(a0, b0);
        sm2[t] = a2;
        sm0[t] = a0;
        se2[t] = b2;
        se0[t] = b0;
      }
      __syncthreads();
    }
    if (t < K) score[t] = prodacc_log(sm2[t], se2[t]) - prodacc_log(sm0[t], se0[t]);
    __syncthreads();
    if (t == 0) {  // :233-242
      int best = 0;
      double bs = score[0];
      for (int c = 1; c < K; ++c)
        if (score[c] > bs) {
          best = c;
          bs = score[c];
        }
      winner = best;
      clust[cell] = best;
    }
    __syncthreads();
    // ---- merge the cell into the winner (:248-251)
    const int w = winner;
    for (int64_t e = e0 + t; e < e1; e += GT) {
      const int32_t snp = entry_snp[e];
      double* dg = diag + ((size_t)snp * K + w) * 4;
      double* g = full + ((size_t)snp * K + w) * 9;
      const double* o = egls + (size_t)e * 9;
      const bool present = dg[3] != 0.0;
      double v[9];
      double tmp = 0;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        v[q] = (present ? g[q] : 1.0) * o[q];
        tmp += v[q];
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) v[q] /= tmp;
      tmp = 0;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        if (v[q] < kMinNormGL) v[q] = kMinNormGL;
        tmp += v[q];
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        v[q] /= tmp;
        g[q] = v[q];
      }
      *reinterpret_cast<double4*>(dg) = make_double4(v[0], v[4], v[8], 1.0);
    }
    __syncthreads();  // the workgroup's stores are visible to its own later loads (one CU, one L1)
  }
}

}  // namespace

extern "C" int muxgl_fmx_greedy_init(muxgl_handle* h, int32_t K, const double* scores, double frac_init_clust,
                                     double singlet_score_thres, int32_t* clust_out) {
  if (!h) return 1;
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
  int32_t *d_order = nullptr, *d_clust = nullptr;
  double *d_diag = nullptr, *d_full = nullptr;
  int rc = 1;
  do {
    if (dev_alloc(h, &d_order, todo.size())) break;
    if (dev_alloc(h, &d_clust, (size_t)C)) break;
    if (dev_alloc(h, &d_diag, (size_t)S * K * 4)) break;
    if (dev_alloc(h, &d_full, (size_t)S * K * 9)) break;
    hipError_t e = hipMemcpyAsync(d_order, todo.data(), sizeof(int32_t) * todo.size(), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_clust, 0xFF, sizeof(int32_t) * (size_t)C, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_diag, 0, sizeof(double) * (size_t)S * K * 4, h->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(fmx_greedy_kernel, dim3(1), dim3(GT), 0, h->stream, d_order, (int64_t)todo.size(),
                         h->d_cell_ptr, h->d_entry_snp, h->d_egls, h->d_af, (int)K, Kp, d_diag, d_full, d_clust);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(clust_out, d_clust, sizeof(int32_t) * (size_t)C, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
      h->err = std::string("muxgl_fmx_greedy_init: ") + hipGetErrorString(e);
      break;
    }
    rc = 0;
  } while (0);
  dev_free(&d_order);
  dev_free(&d_clust);
  dev_free(&d_diag);
  dev_free(&d_full);
  return rc;
}
