// demux_call16.hip -- evidence sums, best/next scans and the SNG/DBL/AMB call (cmd_cram_demuxlet.cpp:788-991) for
// V <= 16, sixteen lanes per cell instead of one.
//
// The reference walks the V*V*A hypotheses of a cell sequentially; one lane per cell doing the same leaves a 10 k-cell
// batch with 157 waves of ~270 serial logAdd's each (0.1 ms, as long as a fifth of the whole sweep; that kernel was
// retired in round 6).  Here lane j of a 16-lane group owns row j of llksAB:
//   * scans: the reference's update rule (strict <, first-seen wins; :827-837, :883-906) leaves the two largest
//     hypotheses under the total order (value descending, scan position ascending); that order is associative, so each
//     lane scans its row and a 4-step butterfly merges the sixteen top-2 lists -- the result is the reference's, tie
//     for tie.
//   * evidence (:804-821): the reference chains logAdd over all terms, log(e^a + e^b) one term at a time.  The same
//     sum is taken here in max-shifted form, M + log(sum_i exp(x_i - M)) with M the largest term of the cell: every
//     lane sums its row's exp's (independent, pipelined), a butterfly adds the row sums, one log closes it, and the
//     reference's start value -1e-300 (:791, sic) joins as a last logAdd term.  The value is the same to ~1e-16
//     relative (the bar is 1e-5); the serial chain of ~25 dependent exp/log pairs per lane -- the whole run time of
//     this kernel -- is gone.
#include "demux_call_body.hpp"

namespace {

using namespace muxgl_call;

// V <= 16: a workgroup of four waves takes four cells -- a wave per cell for the scans (sixteen rows x four ranges of
// columns), the four decisions side by side in wave 0 (demux_call_decide's note) -- exactly as the oct path's finish
// kernel does on its tile in LDS: the two paths give the same records bit for bit.
__global__ void __launch_bounds__(256)
    demux_call16_kernel(int64_t C, const int64_t* __restrict__ cell_ptr, int nv, int nAlpha, call_alpha al,
                        const double* __restrict__ ll, muxgl_demux_cell* __restrict__ out) {
  __shared__ call_partial parts[4];
  const int tid = threadIdx.x, lc = tid >> 6;
  const int64_t cbase = (int64_t)blockIdx.x * 4, i = cbase + lc;
  const bool cell_ok = i < C;
  const call_partial cp = demux_call_scan<16, 4>(tid & 63, cell_ok, nv, nAlpha, al, ll + (size_t)(cell_ok ? i : 0) * nv * nv * nAlpha);
  if ((tid & 63) == 0) parts[lc] = cp;
  __syncthreads();
  if (tid < 4 && cbase + tid < C)
    demux_call_decide(parts[tid], (int32_t)(cell_ptr[cbase + tid + 1] - cell_ptr[cbase + tid]), nv, nAlpha, al, out + cbase + tid);
}

// 16 < V: a wave per cell, lane = row
__global__ void __launch_bounds__(64)
    demux_callg_kernel(int64_t C, const int64_t* __restrict__ cell_ptr, int nv, int nAlpha, call_alpha al,
                        double doublet_prior, const double* __restrict__ ll, muxgl_demux_cell* __restrict__ out) {
  const int lane = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x;
  const bool cell_ok = i < C;
  const int64_t ic = cell_ok ? i : 0;
  demux_call_group<64>(lane, cell_ok, cell_ok ? (int32_t)(cell_ptr[i + 1] - cell_ptr[i]) : 0, nv, nAlpha, al,
                       doublet_prior, ll + (size_t)ic * nv * nv * nAlpha, out + ic);
}

// The call on the wave layout llw[c][n][step t][lane j] (demux_wave.hip): lane j owns row j; at step t it faces sample
// k = j - t - 1 mod 64 (re-derived with the rotation the sweep used), so every read of the wave is one contiguous 512-byte
// line run instead of 64 lines 3 KB apart.  Rows are scanned in step order, not in k order: the top-2 lists are kept
// with the position-aware insertion, the evidence sums do not depend on the order.
__global__ void __launch_bounds__(64)
    demux_call_wave_kernel(int64_t C, const int64_t* __restrict__ cell_ptr, int nv, int nAlpha, call_alpha al,
                           double doublet_prior, const double* __restrict__ llw, muxgl_demux_cell* __restrict__ out) {
  const int64_t i = blockIdx.x;
  const int j = threadIdx.x;
  const bool live = j < nv;
  const double* in = llw + (size_t)i * nAlpha * 4096;
  const double log_single_prior = log((1.0 - doublet_prior) / nv);
  const double log_doublet_prior1 = log(doublet_prior / nv / (nv - 1.) / (nAlpha - 1.));
  const double log_doublet_prior2 = log(doublet_prior / nv / (nv - 1.) / (nAlpha - 1.) * 2);
  const int32_t nsnps = (int32_t)(cell_ptr[i + 1] - cell_ptr[i]);
  top2 sng = {-1e300, -1e300, -1, -1, -1e300}, dbl = {-1e300, -1e300, -1, -1, -1e300};
  const double NEG_INF = -__builtin_huge_val();
  double sterm = NEG_INF, rowmax = NEG_INF, racc = 0.0;
  if (nsnps > 0) {  // (an empty cell leaves no LL behind; its record is all zeros, :653)
    if (live) {
      const double s = in[j];  // llksAB[j][0][0]
      top2_push(sng, s, j);
      sterm = s + log_single_prior;
      rowmax = sterm;
    }
    // ONE pass over the row's hypotheses (round 5; two passes -- the maximum, then the terms relative to it -- read the
    // 196 KB of a 64-sample, six-alpha cell twice, and the slab is 19.6 GB at configs[2]): scans, and the evidence as a
    // running (maximum, sum relative to it) that is rescaled when a larger term turns up, which after the first few
    // elements it rarely does
    if (rowmax > NEG_INF) racc = 1.0;  // the singlet term itself
    for (int n = 1; n < nAlpha; ++n) {
      const bool sym = al.a[n] == 0.5;
      const double prior = sym ? log_doublet_prior2 : log_doublet_prior1;
      int k = j;
      for (int t0 = 0; t0 < 63; t0 += 9) {  // nine loads in flight (every slot of the slab exists: unconditional)
        double vv[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) vv[u] = in[((size_t)n * 64 + t0 + u) * 64 + j];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
          k = __builtin_amdgcn_mov_dpp(k, 0x13C, 0xF, 0xF, false);
          if (!live || k >= nv) continue;
          const double v = vv[u];
          if (!(sym && k < j)) top2_insert(dbl, v, (j * nv + k) * nAlpha + n);  // an alpha = 0.5 pair is listed once, as (lo, hi)
          if (sym && k > j) continue;  // :812-815: an alpha = 0.5 pair counts once
          const double term = v + prior;
          if (term > rowmax) {
            racc = racc * exp_nonpos(rowmax - term) + 1.0;  // (rowmax = -inf: racc is 0 and exp_nonpos gives 0)
            rowmax = term;
          } else {
            racc += exp_nonpos(term - rowmax);
          }
        }
      }
    }
  }
  demux_call_finish<64>(j, true, nsnps, nv, nAlpha, al, doublet_prior, sng, dbl, sterm, rowmax, racc, out + i);
}

}  // namespace

int demux_call_wave_launch(muxgl_handle* h, const muxgl_demux_params* p) {
  const call_alpha al = make_call_alpha(p, h->V);
  hipLaunchKernelGGL(demux_call_wave_kernel, dim3((unsigned)h->C), dim3(64), 0, h->stream, h->C, h->d_cell_ptr, h->V,
                     p->n_alpha, al, p->doublet_prior, h->d_llw, h->d_dcells);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int demux_call16_launch(muxgl_handle* h, const muxgl_demux_params* p) {
  const call_alpha al = make_call_alpha(p, h->V);
  const unsigned blocks = (unsigned)h->C;
  if (h->V <= 16) {
    hipLaunchKernelGGL(demux_call16_kernel, dim3((blocks + 3) / 4 ? (blocks + 3) / 4 : 1), dim3(256), 0, h->stream, h->C,
                       h->d_cell_ptr, h->V, p->n_alpha, al, h->d_ll, h->d_dcells);
  } else {
    hipLaunchKernelGGL(demux_callg_kernel, dim3(blocks ? blocks : 1), dim3(64), 0, h->stream, h->C, h->d_cell_ptr,
                       h->V, p->n_alpha, al, p->doublet_prior, h->d_ll, h->d_dcells);
  }
  HIPCHK(h, hipGetLastError());
  return 0;
}
