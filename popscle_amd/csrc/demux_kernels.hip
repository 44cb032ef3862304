// demux_kernels.hip -- demuxlet hot path on gfx950: per-entry doublet-genotype likelihoods (a4,a5), the
// sample-pair x alpha sweep (a6) and the per-cell evidence/scan/call step (a7-a9).
//
// Reference being replaced: cmd_cram_demuxlet.cpp:636-991 (statgen/popscle).  FP64 throughout.
//
// Data-parallel decomposition (new design, the reference is one thread):
//   * one workgroup per (cell, pair-tile); lanes own (j,k) sample pairs and keep one product accumulator per alpha
//     in registers for the whole cell, so a cell's  sum_e log(sumP_e)  becomes  log(prod_e sumP_e)  with the
//     mantissa/exponent split of prodacc (one log per hypothesis instead of one per entry).
//   * the cell's entries stream through the workgroup in chunks: phase 1, lane <-> entry, turns the entry's reads
//     into pG[nAlpha][3][3] (LDS); the SNP's GP row gp[snp][V][3] is staged coalesced into LDS; phase 2, every lane
//     reads its g_j, g_k (LDS) and the chunk's pG (LDS broadcast) and multiplies its accumulators.
//   * only hypotheses the reference ever reads are computed: (j,0,0) singlets and (j,k!=j,n>=1) doublets; for
//     alpha==0.5 the likelihood is symmetric in (j,k), so k<j is computed once and mirrored.
#include "common.hpp"
#include "demux_entry.hpp"

namespace {

constexpr int kMaxChunk = 64;

__device__ __forceinline__ double lut_err(const double* lut, uint32_t bq) { return lut[bq]; }
__device__ __forceinline__ double lut_mat(const double* lut, uint32_t bq) { return lut[128 + bq]; }

// cmd_cram_demuxlet.cpp:655-725 for one entry.  pG[n*9+l*3+m]; reads in the reference's iteration order.
// Division by the running max is a multiplication by its reciprocal (<=1 ulp per element, tolerance 1e-5 on LLs).
template <int NA>
__device__ __forceinline__ void entry_pg(const uint8_t* __restrict__ reads, int64_t r0, int64_t r1, int nAlpha,
                                         const double* __restrict__ alpha, const double* lut, double (&pG)[NA * 9]) {
#pragma unroll
  for (int i = 0; i < NA * 9; ++i) pG[i] = 1.0;
  for (int64_t r = r0; r < r1; ++r) {
    uint32_t b = reads[r];
    if (b == MUXGL_READ_OTHER) continue;  // :664
    uint32_t al = b >> 7, bq = b & 0x7f;
    double e3 = lut_err(lut, bq) / 3.0, mt = lut_mat(lut, bq);
    double pR = (al == 0) ? mt : e3;  // :666
    double pA = (al == 1) ? mt : e3;  // :667
    double mx = 0.0;
#pragma unroll
    for (int n = 0; n < NA; ++n) {
      if (n < nAlpha) {
        double a = alpha[n];
#pragma unroll
        for (int l = 0; l < 3; ++l) {
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            double p = 0.5 * l + (m - l) * 0.5 * a;  // :673
            double v = pG[n * 9 + l * 3 + m] * (pR * (1.0 - p) + pA * p);
            pG[n * 9 + l * 3 + m] = v;
            mx = fmax(mx, v);
          }
        }
      }
    }
    double inv = 1.0 / mx;
#pragma unroll
    for (int i = 0; i < NA * 9; ++i) pG[i] *= inv;
  }
  double mx = 0.0;
#pragma unroll
  for (int n = 0; n < NA; ++n) {
    if (n < nAlpha) {
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        pG[n * 9 + i] += 1e-10;  // :711
        mx = fmax(mx, pG[n * 9 + i]);
      }
    }
  }
  double inv = 1.0 / mx;
#pragma unroll
  for (int i = 0; i < NA * 9; ++i) pG[i] *= inv;
}

struct alpha_args {
  double a[MUXGL_MAX_ALPHA];
};

// The table of per-entry likelihoods pG[entry][alpha][9] (a4/a5) for the wave / tile kernels and muxgl_demux_get_entry_pg.
// rec == NULL: every entry, lane <-> entry.  rec != NULL (the wave kernels at V <= 64): lane <-> record of the stream of the
// NON-linear entries (plan_build_bit_streams) -- nobody reads the likelihoods of an entry with one usable read, the ring
// kernel takes its (A, Bl, Bm) from a table by allele and quality (demux_ring.hip) -- so a quarter of the entries are
// computed; their rows (still indexed by entry) are written by the lane itself, and the rows of markers without
// genotypes are all ones (neutral: demux_wave.hip).
template <int NA>
__global__ void __launch_bounds__(256) demux_entry_pg_kernel(int64_t nrows, const int64_t* __restrict__ entry_rptr,
                                                              const uint8_t* __restrict__ reads,
                                                              const double* __restrict__ lut_g, int nAlpha,
                                                              alpha_args al, double* __restrict__ pg,
                                                              const fmx_grec* __restrict__ rec,
                                                              const uint8_t* __restrict__ has_gp, int by_record) {
  // by_record (with rec): row r of the table belongs to record r of the stream -- a quarter of the memory of the
  // entry-indexed table, written in whole lines, and what the ring kernel's loader reads in the order it walks
  __shared__ double lut[384];
  __shared__ double stage[4][64 * 9 + 1];  // one alpha of a wave's 64 entries at a time (+1: odd stride, no bank conflicts)
  for (int i = threadIdx.x; i < 384; i += 256) lut[i] = lut_g[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int W = nAlpha * 9;  // doubles per row of the table
  // lane <-> row for the arithmetic; the table rows of a wave's 64 entries are contiguous, so they are written by
  // consecutive lanes through LDS instead of 64 streams W doubles apart
  for (int64_t eb = ((int64_t)blockIdx.x * 4 + w) * 64; eb < nrows; eb += (int64_t)gridDim.x * 256) {
    const int64_t r = eb + lane;
    double pG[NA * 9];
    if (r < nrows) {  // the row kernel's formulation (demux_entry.hpp); slots beyond nAlpha repeat alpha[0]
      const int64_t e = rec ? rec[r].e : r;
      const int64_t r0 = entry_rptr[e], r1 = entry_rptr[e + 1];
      uint32_t first4 = 0;
      for (int64_t k = 0; k < 4 && r0 + k < r1; ++k) first4 |= (uint32_t)reads[r0 + k] << (8 * (int)k);
      row_entry_pg<NA>(reads, r0, r1, first4, al.a, lut, pG);
      if (rec && !has_gp[rec[r].snp]) {
#pragma unroll
        for (int i = 0; i < NA * 9; ++i) pG[i] = 1.0;
      }
    }
    const int ne = (int)((nrows - eb < 64) ? (nrows - eb) : 64);
    if (rec && !by_record) {  // rows of a wave's records are far apart: every lane writes its own 72 * nAlpha bytes
      if (r < nrows) {
        double* o = pg + (size_t)rec[r].e * W;
        if ((W & 1) == 0) {  // rows are 16-byte aligned
#pragma unroll
          for (int x = 0; x < NA * 9 / 2; ++x)
            if (2 * x < W) reinterpret_cast<double2*>(o)[x] = double2{pG[2 * x], pG[2 * x + 1]};
        } else {
#pragma unroll
          for (int x = 0; x < NA * 9; ++x)
            if (x < W) o[x] = pG[x];
        }
      }
      continue;
    }
#pragma unroll
    for (int n = 0; n < NA; ++n) {
      if (n >= nAlpha) break;
#pragma unroll
      for (int i = 0; i < 9; ++i) stage[w][lane * 9 + i] = pG[n * 9 + i];
      // (a wave's LDS traffic is in order: no barrier between the phases)
      for (int x = lane; x < ne * 9; x += 64) {
        const int le = x / 9, i = x - le * 9;
        pg[(size_t)(eb + le) * W + n * 9 + i] = stage[w][x];
      }
    }
  }
}

// fused entry + pair sweep.  grid = (C, n_tiles); block = T threads; thread t of tile y owns pairs[y*T + t].
// dynamic LDS: lut[256] | rows[EC][V*3] | pgs[EC][nAlpha*9] | snp_ok[EC] (int32: snp id or -1)
template <int NA>
__global__ void __launch_bounds__(256)
    demux_sweep_kernel(const int64_t* __restrict__ cell_ptr, const int32_t* __restrict__ entry_snp,
                       const int64_t* __restrict__ entry_rptr, const uint8_t* __restrict__ reads,
                       const double* __restrict__ gp, const uint8_t* __restrict__ has_gp,
                       const double* __restrict__ lut_g, const uint32_t* __restrict__ pairs, int n_pairs, int V,
                       int nAlpha, uint32_t symmask, alpha_args al, int EC, double* __restrict__ ll) {
  extern __shared__ double smem[];
  double* lut = smem;
  double* rows = lut + 256;
  double* pgs = rows + (size_t)EC * V * 3;
  int32_t* snp_ok = (int32_t*)(pgs + (size_t)EC * nAlpha * 9);

  const int T = blockDim.x;
  const int t = threadIdx.x;
  const int64_t c = blockIdx.x;
  const int64_t e0 = cell_ptr[c], e1 = cell_ptr[c + 1];
  if (e0 == e1) return;

  for (int i = t; i < 256; i += T) lut[i] = lut_g[i];

  const int pi = blockIdx.y * T + t;
  uint32_t pk = (pi < n_pairs) ? pairs[pi] : 0u;
  const int j = pk & 0xff, k = (pk >> 8) & 0xff;
  const uint32_t nmask = pk >> 16;

  double acc[NA];
  int32_t ex[NA];
#pragma unroll
  for (int n = 0; n < NA; ++n) {
    acc[n] = 1.0;
    ex[n] = 0;
  }
  const int V3 = V * 3;
  const int PG = nAlpha * 9;
  int since = 0;  // factors multiplied into the accumulators since the last renormalisation
  __syncthreads();

  for (int64_t eb = e0; eb < e1; eb += EC) {
    const int nc = (int)((e1 - eb) < EC ? (e1 - eb) : EC);
    // phase 1: lane <-> entry
    if (t < nc) {
      const int64_t e = eb + t;
      double pG[NA * 9];
      entry_pg<NA>(reads, entry_rptr[e], entry_rptr[e + 1], nAlpha, al.a, lut, pG);
#pragma unroll
      for (int n = 0; n < NA; ++n)
        if (n < nAlpha) {
#pragma unroll
          for (int i = 0; i < 9; ++i) pgs[t * PG + n * 9 + i] = pG[n * 9 + i];
        }
      const int32_t s = entry_snp[e];
      snp_ok[t] = has_gp[s] ? s : -1;
    }
    __syncthreads();
    // stage the GP rows of the chunk, coalesced along the row
    for (int idx = t; idx < nc * V3; idx += T) {
      const int r = idx / V3, i = idx - r * V3;
      const int32_t s = snp_ok[r];
      if (s >= 0) rows[r * V3 + i] = gp[(size_t)s * V3 + i];
    }
    __syncthreads();
    // phase 2: lane <-> pair
    if (nmask) {
      for (int r = 0; r < nc; ++r) {
        if (snp_ok[r] >= 0) {  // :733
          const double* g = rows + r * V3;
          const double gj0 = g[j * 3], gj1 = g[j * 3 + 1], gj2 = g[j * 3 + 2];
          const double gk0 = g[k * 3], gk1 = g[k * 3 + 1], gk2 = g[k * 3 + 2];
          const double p00 = gj0 * gk0, p01 = gj0 * gk1, p02 = gj0 * gk2;  // :740
          const double p10 = gj1 * gk0, p11 = gj1 * gk1, p12 = gj1 * gk2;
          const double p20 = gj2 * gk0, p21 = gj2 * gk1, p22 = gj2 * gk2;
          const double* q = pgs + r * PG;
#pragma unroll
          for (int n = 0; n < NA; ++n) {
            if (n < nAlpha && ((nmask >> n) & 1u)) {
              const double* qn = q + n * 9;
              double s = p00 * qn[0];
              s = fma(p01, qn[1], s);
              s = fma(p02, qn[2], s);
              s = fma(p10, qn[3], s);
              s = fma(p11, qn[4], s);
              s = fma(p12, qn[5], s);
              s = fma(p20, qn[6], s);
              s = fma(p21, qn[7], s);
              s = fma(p22, qn[8], s);
              acc[n] *= s;  // :746 as a product
            }
          }
        }
        if (++since == 16) {  // every factor is >= ~1e-11: sixteen of them cannot underflow.  Counted across chunks:
          since = 0;           // for V > 96 a chunk holds fewer than 16 entries
#pragma unroll
          for (int n = 0; n < NA; ++n) prodacc_renorm(acc[n], ex[n]);
        }
      }
    }
    __syncthreads();
  }

  if (nmask) {
    double* out = ll + (size_t)c * V * V * nAlpha;
#pragma unroll
    for (int n = 0; n < NA; ++n) {
      if (n < nAlpha && ((nmask >> n) & 1u)) {
        const double v = prodacc_log(acc[n], ex[n]);
        out[((size_t)j * V + k) * nAlpha + n] = v;
        if ((symmask >> n) & 1u) out[((size_t)k * V + j) * nAlpha + n] = v;
      }
    }
  }
}

template <int NA>
int launch_sweep(muxgl_handle* h, const muxgl_demux_params* p, uint32_t symmask, const alpha_args& al) {
  const int V = h->V, A = p->n_alpha;
  int T = ((h->n_pairs + 63) / 64) * 64;
  if (T > 256) T = 256;
  if (T < 64) T = 64;
  const int tiles = (h->n_pairs + T - 1) / T;
  int EC = (36 * 1024) / (V * 24);
  if (EC > kMaxChunk) EC = kMaxChunk;
  if (EC > T) EC = T;
  if (EC < 1) EC = 1;
  size_t lds = sizeof(double) * (256 + (size_t)EC * V * 3 + (size_t)EC * A * 9) + sizeof(int32_t) * EC;
  if (lds > 160 * 1024) MUXGL_FAIL(h, "demux sweep needs %zu B of LDS for V=%d (max 163840)", lds, V);
  HIPCHK(h, hipFuncSetAttribute((const void*)demux_sweep_kernel<NA>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
  dim3 grid((unsigned)h->C, (unsigned)tiles);
  hipLaunchKernelGGL(demux_sweep_kernel<NA>, grid, dim3(T), lds, h->stream, h->d_cell_ptr, h->d_entry_snp,
                     h->d_entry_rptr, h->d_reads, h->d_gp, h->d_has_gp, h->d_lut, h->d_pairs, h->n_pairs, V, A,
                     symmask, al, EC, h->d_ll);
  HIPCHK(h, hipGetLastError());
  return 0;
}

template <int NA>
int launch_entry_pg(muxgl_handle* h, const muxgl_demux_params* p, const alpha_args& al, double* d_pg, bool gen_stream,
                    bool by_record) {
  const int64_t nrows = gen_stream ? h->nnz - h->n_lin_rec : h->nnz;
  int64_t blocks = (nrows + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(demux_entry_pg_kernel<NA>, dim3((unsigned)blocks), dim3(256), 0, h->stream, nrows, h->d_entry_rptr,
                     h->d_reads, h->d_lut, p->n_alpha, al, d_pg, gen_stream ? h->d_gen_rec : (const fmx_grec*)nullptr,
                     h->d_has_gp, gen_stream && by_record ? 1 : 0);
  HIPCHK(h, hipGetLastError());
  return 0;
}

}  // namespace

#define DISPATCH_NA(A, CALL)                    \
  do {                                          \
    if ((A) <= 2) return CALL(2);               \
    if ((A) <= 3) return CALL(3);               \
    if ((A) <= 4) return CALL(4);               \
    if ((A) <= 6) return CALL(6);               \
    if ((A) <= 8) return CALL(8);               \
    if ((A) <= 12) return CALL(12);             \
    return CALL(16);                            \
  } while (0)

// builds the (j,k,nmask) work list: (j,0) pairs first so that the singlet slot n=0 lives in the first wave(s)
static int build_pairs(muxgl_handle* h, const muxgl_demux_params* p, uint32_t* symmask_out) {
  const int V = h->V, A = p->n_alpha;
  uint32_t symmask = 0;
  for (int n = 1; n < A; ++n)
    if (p->alpha[n] == 0.5) symmask |= 1u << n;
  std::string key;
  uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)V * V + 4);
  int np = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int j = 0; j < V; ++j) {
      for (int k = 0; k < V; ++k) {
        if ((pass == 0) != (k == 0)) continue;
        uint32_t nmask = 0;
        if (k == 0) nmask |= 1u;  // singlet slot llksAB[j][0][0] (:806,828)
        if (j != k) {
          for (int n = 1; n < A; ++n) {
            if ((symmask >> n) & 1u) {
              if (k < j) nmask |= 1u << n;  // mirrored on store
            } else
              nmask |= 1u << n;
          }
        }
        if (nmask) tmp[np++] = (uint32_t)j | ((uint32_t)k << 8) | (nmask << 16);
      }
    }
  }
  if (np > h->pairs_cap) {
    if (dev_alloc(h, &h->d_pairs, (size_t)np)) {
      free(tmp);
      return 1;
    }
    h->pairs_cap = np;
  }
  hipError_t e = hipMemcpyAsync(h->d_pairs, tmp, sizeof(uint32_t) * np, hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  free(tmp);
  if (e != hipSuccess) MUXGL_FAIL(h, "pair list upload failed: %s", hipGetErrorString(e));
  h->n_pairs = np;
  *symmask_out = symmask;
  return 0;
}

static bool same_params(const muxgl_demux_params& a, const muxgl_demux_params& b) {
  if (a.n_alpha != b.n_alpha) return false;
  for (int i = 0; i < a.n_alpha; ++i)
    if (a.alpha[i] != b.alpha[i]) return false;
  return true;
}

// LL tensor [C][V][V][A]; slots the sweep never writes must read 0
int demux_ensure_ll(muxgl_handle* h, const muxgl_demux_params* p) {
  const size_t need = (size_t)h->C * h->V * h->V * p->n_alpha;
  if (need > h->ll_cap) {
    if (dev_alloc(h, &h->d_ll, need)) return 1;
    h->ll_cap = need;
    h->ll_zeroed = false;
  }
  if (!h->ll_zeroed) {
    HIPCHK(h, hipMemsetAsync(h->d_ll, 0, sizeof(double) * (need ? need : 1), h->stream));
    h->ll_zeroed = true;
  }
  return 0;
}

int demux_launch(muxgl_handle* h, const muxgl_demux_params* p) {
  const int A = p->n_alpha;
  alpha_args al;
  for (int i = 0; i < MUXGL_MAX_ALPHA; ++i) al.a[i] = (i < A) ? p->alpha[i] : 0.0;
  uint32_t symmask = 0;
  for (int n = 1; n < A; ++n)
    if (p->alpha[n] == 0.5) symmask |= 1u << n;
  if (!h->have_dp || !same_params(h->last_dp, *p)) {
    h->last_dp = *p;
    h->have_dp = true;
    h->pairs_valid = false;
    h->ll_zeroed = false;
  }
  h->records_on_host = false;
  h->ll_wave = false;
  int rc = -1;
  if (h->V <= 16) {
    if (demux_ensure_ll(h, p)) return 1;
    rc = demux_oct_launch(h, p);               // the reference's default grid {0, 0.5}: oct kernel
    if (rc == 0 && h->records_on_host) return 0;  // reduce and call were fused into the oct path's finish kernel
    if (rc < 0) rc = demux_row_launch(h, p);    // other grids: row kernel
  }
  if (h->V > 16 && h->V <= 32 && !(h->flags & (MUXGL_FLAG_FORCE_TILE_SWEEP | MUXGL_FLAG_FORCE_WAVE_KERNEL))) {
    if (h->want_full_ll && demux_ensure_ll(h, p)) return 1;  // (without the tensor the call is made in LDS)
    rc = demux_oct_launch(h, p);     // the default grid {0, 0.5}: the oct tiling with sixteen lanes per entry
    if (rc == 0 && h->records_on_host) return 0;
    if (rc < 0) rc = demux_row2_launch(h, p);  // {a0, 0.5} (or MUXGL_FLAG_FORCE_ROW_KERNEL): row kernel, two samples per lane
    if (rc == 0 && h->records_on_host) return 0;  // reduce and call were fused into its finish kernel
  }
  if (rc < 0) rc = demux_wave_launch(h, p);  // one wave per cell and 64 x 64 block of the pair matrix, lane = sample
  if (rc > 0) return rc;
  if (rc < 0) {                     // general tile sweep
    if (demux_ensure_ll(h, p)) return 1;
    if (!h->pairs_valid) {
      if (build_pairs(h, p, &symmask)) return 1;
      h->pairs_valid = true;
    }
    tic(h, MUXGL_T_DEMUX_SWEEP);
#define CALL_SWEEP(N) launch_sweep<N>(h, p, symmask, al)
    rc = [&]() -> int { DISPATCH_NA(A, CALL_SWEEP); }();
#undef CALL_SWEEP
    if (rc) return rc;
    toc(h, MUXGL_T_DEMUX_SWEEP);
  }

  tic(h, MUXGL_T_DEMUX_CALL);
  if (h->ll_wave) {
    if (demux_call_wave_launch(h, p)) return 1;
  } else {
    if (demux_call16_launch(h, p)) return 1;  // 16 or 64 lanes per cell (a lane takes every 64th row beyond 64 samples)
  }
  toc(h, MUXGL_T_DEMUX_CALL);
  return 0;
}

// gen_stream: one row per NON-linear entry, in the order of h->d_gen_rec (needs plan_build_bit_streams)
// by_record (with gen_stream): the table is indexed by the record's position in the stream instead of by entry
int demux_entry_pg_launch(muxgl_handle* h, const muxgl_demux_params* p, double* d_pg, bool gen_stream, bool by_record) {
  const int A = p->n_alpha;
  alpha_args al;
  for (int i = 0; i < MUXGL_MAX_ALPHA; ++i) al.a[i] = (i < A) ? p->alpha[i] : p->alpha[0];
#define CALL_PG(N) launch_entry_pg<N>(h, p, al, d_pg, gen_stream, by_record)
  DISPATCH_NA(A, CALL_PG);
#undef CALL_PG
}
