// fmx_mstep.hip -- freemuxlet's ordered, clamped M-step as a stream (K <= 64).
//
// Reference being replaced (statgen/popscle): the rebuild of the cluster pileups at the end of an EM iteration,
// cmd_cram_freemux2.cpp:586-597 (and the first build, :277-288): for every cell called a singlet, in ascending cell id,
// every entry of the cell is merged into the state of (its cluster, its SNP) by snp_droplet_pileup::merge
// (sc_drop_seq.h:77-101) -- multiply, normalise, clamp at MIN_NORM_GL = 1e-6, normalise again.  The clamp after every
// merge makes a (cluster, SNP) chain order-dependent, so the chains are evaluated exactly in the reference's order and
// the parallelism is over the S*K independent chains.
//
// fmx_mstep_snp_kernel (fmx_kernels.hip) gives a lane a SNP and keeps that SNP's K states in LDS: every element of the
// SNP's list is one step of ONE dependent chain per lane (an LDS round trip, ~25 dependent FP64 operations, the LDS
// write), and LDS caps the lanes in flight: 1.13 ms per 47.9 M entries, a quarter of what the 48-byte rows cost to
// stream.  Here a lane is a CHAIN -- lane = (SNP, cluster), state in registers -- and the G lanes of a SNP share the
// SNP's list as a stream: per batch of BS elements the group loads cell ids, assignments and 48-byte rows coalesced
// (one element per lane and load), parks the rows in LDS and leaves one bit per element in the mask of the cluster it
// belongs to (ds_or); then every lane pops the set bits of its own mask in ascending position (= ascending cell id)
// and merges those rows into its state.  The merges of a round are K independent chains on full wave instructions;
// a batch costs as many rounds as its most frequent cluster has elements.  The arithmetic of a merge is the one of
// fmx_mstep_snp_kernel, operation for operation, so the two kernels agree bit for bit (tests/test_fmx_gpu.py).
#include <cstdlib>

#include "common.hpp"

namespace {

constexpr double kMinNormGL = 1e-6;  // sc_drop_seq.h:14

__device__ __forceinline__ double mstep_rcp(double x) {  // as fast_rcp of fmx_kernels.hip
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  return fma(r, fma(-x, r, 1.0), r);
}

template <int BS>
struct mask_of {
  using type = uint32_t;
};
template <>
struct mask_of<64> {
  using type = unsigned long long;
};

// assignments as bytes (K <= 64 here; 255 = no cluster): a quarter of the bytes behind the gather below
__global__ void __launch_bounds__(256) mstep_clust8_kernel(int64_t C, const int32_t* __restrict__ clust, uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) {
    const int32_t k = clust[i];
    out[i] = (k >= 0 && k < 255) ? (uint8_t)k : (uint8_t)255;
  }
}

// LDS accesses of one wave are served in order; the wave's staging area is its own.  All that is needed between a wave's
// writes and its reads of other lanes' words is that the compiler keeps the order: wait for the LDS counter, clobber
// memory.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// G lanes per SNP (lane = cluster), NB elements per lane and batch; BS = G * NB <= 64 elements per batch and SNP.
// Workgroup = W waves, each with markers of its own.  TAB: the assignment of every cell as a byte table in LDS, copied
// once per workgroup (C bytes of dynamic LDS) -- the gather clust[cell] of an element is then an LDS byte read instead of
// a 64-line vector-memory gather per wave instruction (47.9 M of them per iteration at configs[3]: 0.2 ms of the 0.68 the
// kernel took with it).
template <int G, int NB, int W, bool TAB>
__global__ void __launch_bounds__(64 * W)
    fmx_mstep_stream_kernel(int64_t S, int64_t s0, int64_t s1, int K, int64_t C, const int64_t* __restrict__ snp_ptr,
                            const int32_t* __restrict__ snp_cell, const uint8_t* __restrict__ clust8,
                            const double* __restrict__ segls6, double* __restrict__ cgls,
                            const uint16_t* __restrict__ scode, const double* __restrict__ mtab) {
  // scode / mtab (round 5): three quarters of the elements are entries with at most one usable read, whose six
  // likelihoods are a function of that read's byte alone -- one of 256 rows of mtab (fmx_mstep_codes: made by the entry
  // kernel itself, so the bits are those of segls6).  Such an element's row is read from the 12 KB table (cache hits)
  // instead of from its 48 bytes of the stream: 5 + 0.26 x 48 bytes per element instead of 52.  The loads stay
  // unconditional: the code only selects the address.
  constexpr int BS = G * NB, NG = 64 / G;
  using mask_t = typename mask_of<BS>::type;
  // rows of the batch, one array per matrix element (a lane's write and the reads of a round are then 8 bytes apart
  // between neighbouring positions: no bank conflicts on the way in, few on the way out), and the per-cluster masks
  // (slot G collects the elements without a cluster -- cells not called singlets, positions behind the end of the list --
  // and is never read)
  __shared__ double rows_all[W][NG][6][BS];
  __shared__ mask_t masks_all[W][NG][G + 1];
  extern __shared__ __align__(16) uint8_t tab[];  // filled and read through uint32_t*
  static_assert((sizeof(rows_all) + sizeof(masks_all)) % 16 == 0, "the byte table behind the static LDS must stay 16-byte aligned");
  if (TAB) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(clust8);  // (the byte array is padded to a multiple of 16)
    uint32_t* dst = reinterpret_cast<uint32_t*>(tab);
    for (int64_t i = threadIdx.x; i < (C + 3) / 4; i += 64 * W) dst[i] = src[i];
    __syncthreads();
  }
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane / G, k = lane % G;
  double (*rows)[BS] = rows_all[w][g];
  mask_t* masks = masks_all[w][g];
  const int64_t s = s0 + ((int64_t)blockIdx.x * W + w) * NG + g;
  int64_t p = 0, p1 = 0;
  if (s < s1) {
    p = snp_ptr[s];
    p1 = snp_ptr[s + 1];
  }
  const int64_t plast = (p1 > p) ? p1 - 1 : 0;
  masks[k] = 0;
  double st[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) st[i] = 1.0;

  // pipeline: cell ids two batches ahead, assignments and rows one batch ahead, ONE set of row registers (the loads of
  // the next batch are issued as soon as the current one is parked in LDS, and land while its rounds run); all loads
  // are unconditional from clamped positions (a load under a lane mask makes the compiler wait for it on the spot)
  auto load_ids = [&](int32_t (&c)[NB], uint32_t (&cd)[NB], int64_t base) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int64_t q = base + j * G + k;
      c[j] = snp_cell[q < p1 ? q : plast];
      cd[j] = scode ? (uint32_t)scode[q < p1 ? q : plast] : 0x100u;
    }
  };
  int32_t kk[NB];
  double2 r[NB][3];
  auto load_data = [&](const int32_t (&c)[NB], const uint32_t (&cd)[NB], int64_t base) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int64_t q = base + j * G + k;
      const uint32_t x = TAB ? tab[c[j]] : clust8[c[j]];
      kk[j] = (q < p1 && x < (uint32_t)G) ? (int32_t)x : G;
      const double2* o = reinterpret_cast<const double2*>(cd[j] < 0x100u ? mtab + (size_t)cd[j] * 6
                                                                         : segls6 + (size_t)(q < p1 ? q : plast) * 6);
#pragma unroll
      for (int i = 0; i < 3; ++i) r[j][i] = o[i];
    }
  };
  int32_t cn[NB], cnn[NB];
  uint32_t dn[NB], dnn[NB];
  load_ids(cn, dn, p);
  load_data(cn, dn, p);
  load_ids(cn, dn, p + BS);
  while (__any(p < p1)) {
    // park the batch: rows by position, one bit per element in its cluster's mask
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int q = j * G + k;
      rows[0][q] = r[j][0].x, rows[1][q] = r[j][0].y, rows[2][q] = r[j][1].x;
      rows[3][q] = r[j][1].y, rows[4][q] = r[j][2].x, rows[5][q] = r[j][2].y;
      atomicOr(&masks[kk[j]], (mask_t)1 << q);
    }
    wave_lds_sync();
    mask_t m = masks[k];
    masks[k] = 0;
    load_ids(cnn, dnn, p + 2 * BS);
    load_data(cn, dn, p + BS);
#pragma unroll
    for (int j = 0; j < NB; ++j) cn[j] = cnn[j], dn[j] = dnn[j];
    while (m != 0) {  // (a divergent loop: lanes leave as their masks run out, the state is updated under the exec mask)
      const int j = (sizeof(mask_t) == 8) ? __builtin_ctzll((unsigned long long)m) : __builtin_ctz((uint32_t)m);
      m &= m - 1;
      double v[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = st[i] * rows[i][j];
      double inv = mstep_rcp(((v[0] + v[1]) + v[2]) + 2.0 * ((v[3] + v[4]) + v[5]));
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        v[i] *= inv;
        if (v[i] < kMinNormGL) v[i] = kMinNormGL;
      }
      inv = mstep_rcp(((v[0] + v[1]) + v[2]) + 2.0 * ((v[3] + v[4]) + v[5]));
#pragma unroll
      for (int i = 0; i < 6; ++i) st[i] = v[i] * inv;
    }
    wave_lds_sync();  // the rows are overwritten by the next batch
    p += BS;
  }
  if (s < s1 && k < K) {
    double* og = cgls + ((size_t)k * S + s) * 9;
    og[0] = st[0], og[1] = st[3], og[2] = st[4];
    og[3] = st[3], og[4] = st[1], og[5] = st[5];
    og[6] = st[4], og[7] = st[5], og[8] = st[2];
  }
}

#ifndef MS_NB16
#define MS_NB16 2
#endif
#ifndef MS_W
#define MS_W 16  // waves per workgroup with the table in LDS
#endif
#ifndef MS_WG
#define MS_WG 4  // ... without it (7.3 ms against 7.8 ms with one at configs[4])
#endif

template <int G, int NB>
int mstep_go(muxgl_handle* h, int64_t ns) {
  constexpr int NG = 64 / G;
  const int64_t C = h->C;
  // the table next to the waves' staging areas within the 160 KB of a CU
  constexpr size_t stat = (size_t)MS_W * NG * (6 * G * NB * sizeof(double) + (G + 1) * sizeof(typename mask_of<G * NB>::type));
  const size_t dyn = (size_t)((C + 15) / 16 * 16);
  static const bool no_tab = getenv("MUXGL_MSTEP_NO_TABLE") != nullptr;  // (tests: the variant for many cells on few)
  // (measured: with the table 0.54 vs 0.59 ms at configs[3], K = 16; at K = 64 and 50 k cells 2.87 vs 2.55 ms -- one
  // 16-wave workgroup per CU loses more than the gather costs -- so the table is for the 16-lane groups only)
  if (G == 16 && stat + dyn <= 160 * 1024 && !no_tab) {
    auto kern = fmx_mstep_stream_kernel<G, NB, MS_W, true>;
    HIPCHK(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    const int64_t per = (int64_t)NG * MS_W;
    hipLaunchKernelGGL(kern, dim3((unsigned)((ns + per - 1) / per)), dim3(64 * MS_W), dyn, h->stream, h->S, h->fs0, h->fs1,
                       h->K, C, h->d_snp_ptr, h->d_snp_cell, h->d_clust8, h->d_segls6, h->d_cgls, h->d_scode, h->d_mtab);
  } else {
    auto kern = fmx_mstep_stream_kernel<G, NB, MS_WG, false>;  // MS_WG independent waves per workgroup
    const int64_t per = (int64_t)NG * MS_WG;
    hipLaunchKernelGGL(kern, dim3((unsigned)((ns + per - 1) / per)), dim3(64 * MS_WG), 0, h->stream, h->S, h->fs0, h->fs1, h->K,
                       C, h->d_snp_ptr, h->d_snp_cell, h->d_clust8, h->d_segls6, h->d_cgls, h->d_scode, h->d_mtab);
  }
  HIPCHK(h, hipGetLastError());
  return 0;
}

}  // namespace

// -1: not applicable (K > 64 or an empty pileup: fmx_mstep_snp_kernel)
int fmx_mstep_stream_launch(muxgl_handle* h) {
  const int64_t ns = h->fs1 - h->fs0;
  const int K = h->K;
  if (K > 64 || ns <= 0 || h->nnz <= 0 || h->C <= 0) return -1;
  const int64_t C = h->C;  // cells d_clust spans (a column slab: the cells of the whole job)
  if (h->clust8_n != C || !h->d_clust8) {
    if (dev_alloc(h, &h->d_clust8, (size_t)((C + 15) / 16 * 16))) return 1;
    h->clust8_n = C;
  }
  hipLaunchKernelGGL(mstep_clust8_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, h->stream, C, h->d_clust, h->d_clust8);
  if (K <= 16) return mstep_go<16, MS_NB16>(h, ns);
  if (K <= 32) return mstep_go<32, 1>(h, ns);
  return mstep_go<64, 1>(h, ns);
}
