// oct_tiling.hpp -- the eight-lanes-per-entry tiling shared by demux_oct.hip (V <= 16 samples) and fmx_oct.hip (K <= 16
// clusters): lane = 16 g + 2 p + h is position p (items p and p + 8) of entry h of DPP row g, so that row_ror:2t rotates
// the eight positions of both entries by t.  18 accumulators per lane: 2 singlets, the in-lane pair (a, b), four pairs
// with the partner at each of the rotations t = 1, 2, 3, and at t = 4, where lane p faces lane p + 4, the pair (a, b')
// plus (a, a') and (b, b'), which both lanes hold and one publishes.  See demux_oct.hip for why eight.
#pragma once
#include "common.hpp"

namespace oct {

constexpr int N_ACC = 18;          // accumulators per lane
constexpr int SLOTS = 8;           // entry streams (chunks) per wave
constexpr int N_HYP = N_ACC * 8;   // accumulators of a chunk: 144 slots for the 136 hypotheses
constexpr int ROR2 = 0x122, ROR4 = 0x124, ROR6 = 0x126, ROR8 = 0x128;  // row_ror:2t = the ring of eight positions by t

template <int CTRL>
__device__ __forceinline__ double dpp_rot(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// accumulator index layout of a lane (own items a = p, b = p + 8; partner at rotation t: a' = p_t, b' = p_t + 8)
__host__ __device__ constexpr int acc_single(int c) { return c; }  // c = 0: a, 1: b
constexpr int ACC_AB = 2;                                          // (a, b)
__host__ __device__ constexpr int acc_rot(int t, int c, int d) { return 3 + (t - 1) * 4 + c * 2 + d; }  // t = 1..3
constexpr int ACC_F_AB = 15, ACC_F_AA = 16, ACC_F_BB = 17;         // t = 4: (a, b'), (a, a'), (b, b')

// position (0..7) whose items a lane sees after row_ror:2t, t = 1..4, measured rather than assumed: pmap[4][8]
static __global__ void pmap_kernel(int32_t* pmap) {
  const int lane = threadIdx.x;
  const int p = (lane >> 1) & 7;
  const int p1 = __builtin_amdgcn_mov_dpp(p, ROR2, 0xF, 0xF, false);
  const int p2 = __builtin_amdgcn_mov_dpp(p, ROR4, 0xF, 0xF, false);
  const int p3 = __builtin_amdgcn_mov_dpp(p, ROR6, 0xF, 0xF, false);
  const int p4 = __builtin_amdgcn_mov_dpp(p, ROR8, 0xF, 0xF, false);
  if (lane < 16 && (lane & 1) == 0) {
    pmap[p] = p1;
    pmap[8 + p] = p2;
    pmap[16 + p] = p3;
    pmap[24 + p] = p4;
  }
}

// accumulator idx = a * 8 + p of a chunk -> the items (j, k) of its hypothesis (singlets: k = -1); false for the copy of
// a pair that the facing lane publishes
__device__ __forceinline__ bool hypothesis_of(int idx, const int32_t* __restrict__ pmap, int& j, int& k) {
  const int a = idx >> 3, p = idx & 7;
  if (a < 2) {
    j = p + 8 * a;
    k = -1;
    return true;
  }
  if (a == ACC_AB) {
    j = p + 8;
    k = p;
    return true;
  }
  if (a < 15) {
    const int t = (a - 3) >> 2, c = ((a - 3) >> 1) & 1, d = (a - 3) & 1;
    j = p + 8 * c;
    k = pmap[t * 8 + p] + 8 * d;
    return true;
  }
  const int pf = pmap[24 + p];
  if (a == ACC_F_AB) {
    j = p;
    k = pf + 8;
    return true;
  }
  j = p + (a == ACC_F_AA ? 0 : 8);
  k = pf + (a == ACC_F_AA ? 0 : 8);
  return p < pf;
}

// Steps of a unit's loop over one kind of entries: the longest list among its eight chunks, rounded to the loop's
// unrolling, plus the read-ahead (0 for a unit without such entries).  counts[q] = entries of that kind in chunk q.
static __global__ void __launch_bounds__(256)
    unit_steps_kernel(int n_units, int n_chunks, const int32_t* __restrict__ order, const int32_t* __restrict__ counts,
                      const row_chunk* __restrict__ chunks, int complement, int unroll, int pad, int32_t* __restrict__ steps) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_units) return;
  int m = 0;
  for (int k = 0; k < SLOTS; ++k) {
    const int w = u * SLOTS + k;
    if (w < n_chunks) {
      const int q = order ? order[w] : w;
      m = max(m, complement ? chunks[q].len - counts[q] : counts[q]);
    }
  }
  steps[u] = m > 0 ? (m + unroll - 1) / unroll * unroll + pad : 0;
}

}  // namespace oct
