// row2.hpp -- what the row kernels for 17..32 samples / clusters (demux_row2.hip, fmx_row2.hip) share: rotations of a
// 16-lane DPP row by a constant, the lane map that goes with them, and the accumulator layout.
#pragma once
#include "common.hpp"

namespace {

// the value lane (j + T) mod 16 of the same 16-lane row holds (DPP row_ror:T): every rotation reads the original
// operand, so the eight steps of a sweep do not form a chain
template <int T>
__device__ __forceinline__ int row2_ror_i32(int x) {
  return __builtin_amdgcn_mov_dpp(x, 0x120 + T, 0xF, 0xF, false);
}
template <int T>
__device__ __forceinline__ double row2_ror(double x) {
  return __hiloint2double(row2_ror_i32<T>(__double2hiint(x)), row2_ror_i32<T>(__double2loint(x)));
}

// which lane's operands lane j sees after row_ror:t (measured with the same instruction, so the kernels never assume a
// rotation direction): kmap[t][j], t = 0..8
__global__ void row2_kmap_kernel(int32_t* kmap) {
  const int lane = threadIdx.x, v = lane & 15;
  int r[9] = {v, row2_ror_i32<1>(v), row2_ror_i32<2>(v), row2_ror_i32<3>(v), row2_ror_i32<4>(v), row2_ror_i32<5>(v),
              row2_ror_i32<6>(v), row2_ror_i32<7>(v), row2_ror_i32<8>(v)};
  if (lane < 16)
    for (int t = 0; t < 9; ++t) kmap[t * 16 + lane] = r[t];
}

// accumulators of lane j (a = unit j, b = unit j + 16 -- samples or clusters; after t rotations the partner lane's units
// are ka = kmap[t][j] and kb = ka + 16):
//   0 singlet a, 1 singlet b, 2 pair (a,b), 3 + 4 (t-1) + {0 (a,ka), 1 (a,kb), 2 (b,ka), 3 (b,kb)} for t = 1..8
constexpr int ROW2_NACC = 35;

// the pair (x, y) accumulator `a` of lane j stands for; false where another lane is the writer (rotation 8 visits every
// unordered pair of lanes twice) -- singlets come back as x == y
__device__ __forceinline__ bool row2_pair_of(int a, int j, const int32_t* __restrict__ kmap, int& x, int& y) {
  if (a < 2) {
    x = y = j + 16 * a;
  } else if (a == 2) {
    x = j, y = j + 16;
  } else {
    const int t = 1 + ((a - 3) >> 2), combo = (a - 3) & 3;
    const int ka = kmap[t * 16 + j];
    if (t == 8 && j < ka) return false;
    x = j + 16 * (combo >> 1), y = ka + 16 * (combo & 1);
  }
  return true;
}

// per-chunk partials beyond this many bytes: the caller falls back to the wave kernels
constexpr double ROW2_PART_LIMIT = 64e9;

}  // namespace
