// common.hpp -- handle layout, error plumbing and small device helpers shared by the libmuxgl translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <unordered_map>
#include <unordered_set>

#include "../../include/muxgl.h"

constexpr int MUXGL_ROW_CH = 128;   // entries per chunk of the row kernels (16-lane slots)
constexpr int MUXGL_QUAD_CH = 192;  // entries per chunk of the freemuxlet oct E-step (configs[3], E-step in ms: 128: 1.63,
                                    // 160: 1.58, 192: 1.50, 256: 1.49)
constexpr int MUXGL_OCT_CH = 192;   // entries per chunk of the demuxlet oct kernel (8-lane slots).  Measured at configs[1]
                                    // (sweep + finish, ms): 96: 0.270 + 0.107, 128: 0.258 + 0.083, 160: 0.243 + 0.074,
                                    // 192: 0.252 + 0.068, 224: 0.250 + 0.066, 256: 0.253 + 0.067, 320: 0.281 + 0.066 --
                                    // longer chunks are fewer partial products for the finish kernel, shorter ones more
                                    // work units to balance.  A constant: a cell's cut must not depend on its neighbours.

// per-entry record of the quad kernel: everything phase 1 needs for an entry with <= 4 reads in ONE 16-byte load
struct quad_entry {
  int32_t snp;      // SNP id
  uint32_t nreads;  // reads of the entry
  uint32_t first4;  // its first four packed read bytes (byte k = read k)
  uint32_t r0;      // offset of read 0 in the reads array (used only beyond four reads)
};

// Record of a linear entry (at most one usable read) of the quad kernel's sweep: its marker and the read byte that
// counts.  code 0xFF (MUXGL_READ_OTHER never counts) = no usable read, or a marker without genotypes.
struct quad_lrec {
  int32_t snp;
  uint32_t code;
};

// one work unit of the row kernels (demux_row.hip, fmx_kernels.hip): <= 128 consecutive entries of one cell
struct row_chunk {
  int64_t e0;
  int32_t len;
  int32_t cell;
};

// chunk tables of the row kernels, built by demux_row_plan() at muxgl_set_pileup time
struct muxgl_row_state {
  row_chunk* d_chunks = nullptr;        // launch order: non-increasing length, then ascending first SNP
  int64_t* d_cell_chunk_ptr = nullptr;  // [C+1]
  int32_t* d_cell_chunks = nullptr;     // chunk positions of each cell in entry order
  int32_t* d_kmap = nullptr;            // [16][16]: sample held by lane j after t DPP row rotations
  int32_t* d_tmap = nullptr;            // [P/2][P]: position seen by a lane of the oct tiling after each rotation (P = tmap_p)
  int tmap_p = 0;
  double* d_part = nullptr;             // per-chunk partial log-likelihoods (row kernels) / mantissas (quad kernel)
  quad_entry* d_qent_lin = nullptr;     // quad kernel: the entry records with every chunk's linear entries first ...
  int32_t* d_chunk_nlin = nullptr;      // ... and how many they are, per chunk (demux_oct.hip, built on first use)
  uint2* d_orec = nullptr;              // ... and the linear ones as {row offset, table offset} records, step-major per
  int64_t* d_unit_ptr = nullptr;        //     unit of eight chunks: unit u starts at d_orec[d_unit_ptr[u]] (demux_oct.hip)
  int32_t* d_quad_order = nullptr;      // ... and the launch order of the chunks (sorted by trip count within buckets)
  // freemuxlet quad E-step (fmx_quad.hip, built on first use after muxgl_fmx_prepare): per chunk, its linear entries
  // as {c0, c1, snp} records in front, then the SNP ids and six likelihoods of the others
  int32_t* d_fq_nlin = nullptr;
  int32_t *d_fo_lsteps = nullptr, *d_fo_gsteps = nullptr;  // steps of every unit's two loops (fmx_oct.hip)
  int64_t *d_fo_lptr = nullptr, *d_fo_gptr = nullptr;      // first record of every unit in the step-major tables below
  uint32_t *d_fo_loff = nullptr, *d_fo_goff = nullptr;     // row offsets of the linear / other entries
  double2* d_fo_lc = nullptr;                              // (c0, c1) of the linear entries
  double* d_fo_ggl = nullptr;                              // six likelihoods of the other entries
  int32_t* d_fq_order = nullptr;        // launch order of the chunks, as d_quad_order
  int32_t* d_part_e = nullptr;          // per-chunk partial exponents (quad kernel)
  int32_t* d_chunk_pos = nullptr;       // position of every chunk in its cell's list: where the oct kernel leaves its partials
  size_t part_cap = 0, part_e_cap = 0;
  int64_t n_chunks = 0;
};

// what a handle's pileup is (muxgl_fmx_set_column_slab / device groups): the whole pileup, the cell-major row slab of a
// rank (its cells, every SNP: E-step, scans, re-assignment), or the column slab (every cell, its SNPs: ordered M-step)
enum { MUXGL_ROLE_FULL = 0, MUXGL_ROLE_ROWS = 1, MUXGL_ROLE_COLS = 2 };

struct muxgl_group;

// records of the wave E-step's entry streams (fmx_wave.hip)
struct fmx_lrec {  // a linear entry: glis[g1][g2] = c0 + c1 (g1 + g2)
  double c0, c1;
  int32_t snp, pad;
};
struct fmx_grec {  // any other entry: its index and SNP
  int64_t e;
  int32_t snp, pad;
};
// exact scan results of a settled near-tie cell (fmx_exact.hip), read back by fmx_call_kernel while the cell's inputs last
struct fmx_xc {
  int32_t sBest, sNext, dBest1, dBest2, dNext1, dNext2;
  double sngBestLLK, sngNextLLK, dblBestLLK, dblNextLLK;
};

struct muxgl_handle {
  std::multimap<size_t, void*> pool;   // cached device blocks by size (dev_alloc / dev_free below)
  size_t pool_bytes = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = true;
  std::string err;
  int role = MUXGL_ROLE_FULL;
  muxgl_handle* col = nullptr;         // column slab of a slabbed handle (owned; same device, same stream)
  int64_t C_total = 0, cell_base = 0;  // slabbed handle: cells of the whole job, global id of local cell 0
  int64_t slab_s0 = 0, slab_s1 = 0;    // column slab: its SNP range
  muxgl_group* group = nullptr;        // device group behind this handle (muxgl_group.hip); the fields below are unused then
  int32_t* h_fstat = nullptr;          // pinned: counters of the last E-step (asynchronous phases)
  hipEvent_t ev_stat = nullptr;

  // packed pileup (device)
  int64_t C = 0, S = 0, nnz = 0, R = 0;
  int64_t* d_cell_ptr = nullptr;
  int32_t* d_entry_snp = nullptr;
  int64_t* d_entry_rptr = nullptr;
  uint8_t* d_reads = nullptr;
  int32_t* d_entry_cell = nullptr;  // cell id of each entry (for SNP-major views)
  quad_entry* d_qent = nullptr;     // [nnz] packed records of the quad kernel (built when R < 2^32)
  // One bit per entry: the entry has at most one usable read, so its likelihoods are LINEAR in the two genotypes
  // (demuxlet: pG[l][m] = A + Bl*l + Bm*m, cmd_cram_demuxlet.cpp:673-685 for a single factor) and a pair hypothesis is a
  // two-term form in the samples' moments (sum g, sum l*g) instead of a three-term one -- see demux_wave.hip.
  uint32_t* d_lin = nullptr;        // [ceil(nnz/32)] demuxlet
  int64_t* d_lin_rank = nullptr;    // demuxlet wave kernels, built on first use: plan_build_bit_streams of d_lin
  fmx_grec* d_lin_rec = nullptr;    // [n_lin_rec] {entry, snp} of the linear entries, in entry order
  fmx_grec* d_gen_rec = nullptr;    // [nnz - n_lin_rec] the others
  int64_t n_lin_rec = -1;
  uint2* d_ring_rec = nullptr;      // demux_ring.hip: {snp, table row} of the linear entries, in stream order
  double* d_ring_lut = nullptr;     // ... and the table of (A, Bl, Bm) rows of the current launch
  int64_t ring_rec_n = -1;
  uint32_t* d_flin = nullptr;       // [ceil(nnz/32)] freemuxlet: additionally, no clamp fired (checked on the values)
  // The wave E-step's two streams (fmx_wave.hip, built on first use): a cell's linear entries as 24-byte records
  // {c0, c1, snp} and its other entries as {entry, snp}, both in entry order; d_flin_rank[w] = linear entries before
  // entry 32 w (so a cut of a long cell finds its place in both streams); d_cE[S][K] = g1 + 2 g2 of the posteriors.
  int64_t* d_flin_rank = nullptr;   // [ceil(nnz/32) + 1]
  fmx_lrec* d_lrec = nullptr;
  fmx_grec* d_grec = nullptr;
  double* d_cE = nullptr;
  int64_t n_lrec = -1, cE_n = 0;
  int64_t max_cell_entries = 0;

  // Phred LUT: [0..127] = phred2Err, [128..255] = phred2Mat (bq is 7 bits in the packed read byte)
  double* d_lut = nullptr;

  // demuxlet
  int32_t V = 0;
  double* d_gp = nullptr;
  uint8_t* d_has_gp = nullptr;
  double* d_gpq = nullptr;   // V <= 16: GP tensor re-laid for the quad kernel, [S][6][4][2] (demux_oct.hip)
  double gp_min_sum = 1.0;    // V > 32: smallest sum of a genotype triple (muxgl_demux_set_gp): below 0.35 the split sweep
  bool gp_unit_sums = false;  // V <= 16: every triple sums to 1 within 4 ulp; d_gpq / d_gmq then carry no sums (demux_oct.hip)
  double* d_gp0s = nullptr;  // V <= 16: per-SNP sum of sample 0's triple (the factor every singlet carries, :806)
  double* d_gmq = nullptr;   // V <= 16: moments (s, rho) of every triple in the quad layout, [S + 1][4][4][2] (demux_oct.hip)
  double* d_ll = nullptr;  // [C][V][V][A]
  double* d_llw = nullptr; // wave path: [C][A][64 rotation steps][64 lanes], see demux_wave.hip
  size_t llw_cap = 0;
  bool ll_wave = false;    // the last sweep left its result in d_llw
  size_t ll_cap = 0;
  bool ll_zeroed = false;
  muxgl_demux_cell* d_dcells = nullptr;
  muxgl_demux_cell* h_dcells = nullptr;  // pinned, device-visible
  bool want_full_ll = false;     // this run must leave the LL tensor in d_ll
  bool records_on_host = false;  // the launch wrote the records into h_dcells itself (no D2H copy needed)
  int64_t dcells_cap = 0;
  uint32_t* d_pairs = nullptr;  // packed (j | k<<8 | nmask<<16) work list of the sweep
  int32_t n_pairs = 0;
  int32_t pairs_cap = 0;
  muxgl_demux_params last_dp{};
  bool have_dp = false;
  bool pairs_valid = false;
  struct muxgl_row_state* row = nullptr;  // chunk tables of the V<=16 row kernel (demux_row.hip)
  struct muxgl_row_state* qrow = nullptr; // chunk tables of the default-grid quad kernel (demux_oct.hip)
  struct muxgl_wave_state* wave = nullptr; // cell order + pG table of the 16 < V <= 64 wave kernel (demux_wave.hip)
  int32_t flags = 0;

  // freemuxlet
  double* d_af = nullptr;
  int64_t greedy_near_ties = 0, greedy_overruled = 0;  // of the last muxgl_fmx_greedy_init (fmx_greedy.hip)
  int64_t fmx_exact_scores = 0;  // of the last muxgl_fmx_prepare (score_exact.hpp)
  double* d_egls = nullptr;       // [nnz][9] entry pileup likelihoods (b1)
  int32_t* d_ecnt = nullptr;      // [nnz][3] nreads,nref,nalt
  int32_t K = 0;
  double* d_cgls = nullptr;       // [K][S][9] cluster pileup likelihoods
  int32_t* d_ccnt = nullptr;      // [K][S][3]
  double* d_cgp = nullptr;        // [S][K][3] cluster genotype posteriors for the E-step
  int32_t* d_clust = nullptr;     // [C] current singlet cluster or -1
  uint8_t* d_clust8 = nullptr;    // the same as bytes (255: none), rewritten by every M-step (fmx_mstep.hip)
  int64_t clust8_n = -1;
  muxgl_fmx_cell* d_fcells = nullptr;
  muxgl_fmx_cell* h_fcells = nullptr;  // pinned
  double* d_fll = nullptr;        // [C][K(K+1)/2]
  int32_t* d_fstat = nullptr;     // nsingle, namb, nchanged, cells listed for the exact-call path
  // near-tie calls (fmx_exact.hip): what fmx_call_kernel keeps aside for the exact path
  int32_t* d_prev_clust = nullptr;  // the assignments the cluster pileups of the running iteration were built from: a copy
                                    // of d_clust taken before the call kernel runs, on the handle that runs the M-step
  int32_t* d_prev_state = nullptr;  // [C] (type, jBest, kBest) before the running iteration, a byte each
  int32_t* d_flagged = nullptr;     // [C] cells whose call is within rounding reach, d_fstat[3] of them
  int32_t* d_xc_epoch = nullptr;    // [C] epoch a cell's entry of d_xc was written in (0: none)
  fmx_xc* d_xc = nullptr;           // [C] exact scan results of settled cells
  int32_t xs_epoch = 1;             // advances whenever the E-step's inputs may have changed
  bool xs_keep = false, xs_have_p = false;
  muxgl_fmx_params xs_p{};
  int64_t fmx_exact_cells = 0, fmx_exact_changed = 0, fmx_exact_unresolved = 0;  // since muxgl_fmx_set_clusters
  struct fmx_exact_state* xs = nullptr;  // between steps A and C of the exact path (fmx_exact.hip)
  int32_t fmx_listed = 0;                // cells the last fetch of a sharded phase found listed and left open
  std::vector<int32_t> xs_snps;          // muxgl_fmx_exact_snps: the list between the size query and the copy
  bool xs_snps_valid = false;
  // SNP-major (CSC) view of the entries, cells ascending inside each SNP: M-step walks it
  int64_t* d_snp_ptr = nullptr;   // [S+1]
  int64_t* d_snp_entry = nullptr; // [nnz] entry index
  int32_t* d_snp_cell = nullptr;  // [nnz] cell id of the same SNP-major element
  double* d_segls = nullptr;      // [nnz][9] entry likelihoods in SNP-major order (freemuxlet-old's kernels)
  double* d_segls6 = nullptr;     // [nnz][6] their six distinct values {00,11,22,01,02,12}: the ordered M-step streams them
  uint16_t* d_scode = nullptr;    // [nnz] SNP-major: read byte of an entry with at most one usable read, 0x100 otherwise (fmx_scode_kernel)
  double* d_mtab = nullptr;       // [256][6] the six values of a one-read entry by its read byte (fmx_mstep.hip)
  double* d_egls6 = nullptr;      // (unused since round 4: the oct E-step's streams are repacked from d_egls)
  int32_t* d_secnt = nullptr;     // [nnz][3] entry counts in SNP-major order
  bool fmx_prepared = false;
  int64_t fc0 = 0, fc1 = 0, fs0 = 0, fs1 = 0;  // active cell / SNP shard of the EM phases (default: everything)
  muxgl_row_state* frow = nullptr;             // chunk tables restricted to the cell shard
  muxgl_row_state* fqrow = nullptr;            // same for the quad E-step
  double* d_cgpq = nullptr;                    // [S][6][4][2] cluster-GP rows re-laid per quad (fmx_quad.hip)
  double* d_ceq = nullptr;                     // [S + 1][2][4][2] their moments E = g1 + 2 g2 (fmx_quad.hip)
  size_t ceq_cap = 0;
  size_t cgpq_cap = 0;

  // freemuxlet-old (fmx_old.hip)
  int8_t* d_sgn = nullptr;  // [C][sgn_ld] vote sign of every ordered cell pair: +1 / -1 / 0 against --bf-thres
  int64_t sgn_ld = 0;       // row stride, a multiple of 16 with at least 16 zero bytes of padding

  hipEvent_t ev[2 * MUXGL_T_COUNT] = {};
  bool ev_used[MUXGL_T_COUNT] = {};
  int ev_start[MUXGL_T_COUNT] = {};  // index into ev[] of a slot's start event (toc_tic: the previous slot's stop event)
  float ms[MUXGL_T_COUNT] = {};
  double ms_sum[MUXGL_T_COUNT] = {};  // the same summed over the calls since muxgl_get_timing_sum(.., reset = 1)
  int64_t ms_calls = 0;
};

extern thread_local std::string g_muxgl_create_error;

// MUXGL_TIMING=1: host-side stage times of the hand-over calls on stderr
struct host_timer {
  bool on;
  double t0;
  static double now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
  }
  host_timer() : on(getenv("MUXGL_TIMING") != nullptr), t0(now()) {}
  void lap(const char* what) {
    if (!on) return;
    const double t = now();
    fprintf(stderr, "[muxgl] %-44s %8.1f ms\n", what, (t - t0) * 1e3);
    t0 = t;
  }
};

#define MUXGL_FAIL(h, ...)                                  \
  do {                                                      \
    char _buf[512];                                         \
    snprintf(_buf, sizeof(_buf), __VA_ARGS__);              \
    (h)->err = _buf;                                        \
    return 1;                                               \
  } while (0)

#define HIPCHK(h, call)                                                                         \
  do {                                                                                          \
    hipError_t _e = (call);                                                                     \
    if (_e != hipSuccess) MUXGL_FAIL(h, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e),  \
                                     __FILE__, __LINE__);                                       \
  } while (0)

// ---- device memory: dev_alloc / dev_free with a per-handle cache of large blocks -----------------------------------------
// The phases of a run allocate and free tens of GB each (entry likelihoods, the greedy start's tables, cluster pileups);
// hipMalloc of memory that was in use before costs ~30 ms per GB on these boxes (the driver clears it) and hipFree drains
// the device.  Blocks of DEV_POOL_MIN bytes or more therefore go back to a cache of the handle that allocated them and
// are handed out again to a request they fit (best fit, at most a quarter + 16 MB larger than asked for); the stream of
// the handle is drained before a cached block is reused, which is what hipFree would have done when it was released.
// A failed hipMalloc empties every cache of the process and tries again; muxgl_destroy empties the handle's.
// MUXGL_NO_POOL=1 turns the cache off.
constexpr size_t DEV_POOL_MIN = (size_t)8 << 20;
struct dev_block_info {
  size_t bytes;
  muxgl_handle* owner;  // NULL once its handle is gone: the block is then freed for good
};
struct dev_registry {
  std::mutex mu;
  std::unordered_map<void*, dev_block_info> blocks;   // live and cached blocks of at least DEV_POOL_MIN bytes
  std::unordered_set<muxgl_handle*> handles;          // handles with a cache
};
inline dev_registry& dev_reg() {
  static dev_registry r;
  return r;
}
inline bool dev_pool_on() {
  static const bool on = getenv("MUXGL_NO_POOL") == nullptr;
  return on;
}
void dev_pool_release(muxgl_handle* h, bool forget_owner);  // muxgl_api.hip
void dev_pool_release_all();
// hipMalloc for buffers that do not go through the cache (rocprim temporaries): on failure the cached blocks of every handle
// go back to the driver and the allocation is tried once more, as dev_alloc does
hipError_t dev_malloc_retry(void** p, size_t bytes);

int dev_alloc_bytes(muxgl_handle* h, void** p, size_t bytes);  // muxgl_api.hip
void dev_free_bytes(void* p);

template <typename T>
static inline void dev_free(T** p) {
  if (*p) dev_free_bytes((void*)*p);
  *p = nullptr;
}

template <typename T>
static inline int dev_alloc(muxgl_handle* h, T** p, size_t n) {
  dev_free(p);
  if (n == 0) n = 1;
  return dev_alloc_bytes(h, (void**)p, n * sizeof(T));
}

static inline bool timing_off() {  // MUXGL_NO_EVENTS=1: no hipEvent records around the kernels (launch-gap experiments)
  static const bool off = getenv("MUXGL_NO_EVENTS") != nullptr;
  return off;
}
static inline void tic(muxgl_handle* h, int id) {
  if (timing_off()) return;
  (void)hipEventRecord(h->ev[2 * id], h->stream);
  h->ev_start[id] = 2 * id;
  h->ev_used[id] = true;
}
// stop of slot `id` and start of slot `next` as ONE event record (a record costs ~2.6 us of host time per call)
static inline void toc_tic(muxgl_handle* h, int id, int next) {
  if (timing_off()) return;
  (void)hipEventRecord(h->ev[2 * id + 1], h->stream);
  h->ev_start[next] = 2 * id + 1;
  h->ev_used[next] = true;
}
static inline void toc(muxgl_handle* h, int id) {
  if (timing_off()) return;
  (void)hipEventRecord(h->ev[2 * id + 1], h->stream);
}
static inline void clear_timing(muxgl_handle* h) {
  for (int i = 0; i < MUXGL_T_COUNT; ++i) {
    h->ev_used[i] = false;
    h->ms[i] = 0.f;
  }
}
// Reads the brackets recorded since the last call: a slot is summed ONCE per bracket (its `used` mark is cleared here,
// so the phased API, which collects after every phase, does not add the earlier phases' times again) and only when
// the elapsed-time query succeeds; ms[] keeps the last value of every slot until clear_timing.  ms_calls counts the
// collecting calls (include/muxgl.h says which entry points those are).
static inline void collect_timing(muxgl_handle* h) {
  for (int i = 0; i < MUXGL_T_COUNT; ++i) {
    if (h->ev_used[i]) {
      h->ev_used[i] = false;
      float t = 0.f;
      if (hipEventElapsedTime(&t, h->ev[h->ev_start[i]], h->ev[2 * i + 1]) == hipSuccess) {
        h->ms[i] = t;
        h->ms_sum[i] += (double)t;
      }
    }
  }
  ++h->ms_calls;
}

// ---- device helpers ---------------------------------------------------------------------------------------------

// product accumulator: value = m * 2^e, renormalised to m in [0.5,1) every few factors so that a cell's
// sum of log(sumP) becomes one log() at the end:  sum log x_i = log(prod mantissas) + ln2 * sum exponents
struct prodacc {
  double m;
  int32_t e;
};
__device__ __forceinline__ void prodacc_renorm(double& m, int32_t& e) {
  int ex;
  m = frexp(m, &ex);
  e += ex;
}
// log of a positive finite double for the end of a product accumulator (round 5).  The library log costs 98 vector
// instructions for special cases a product of positive likelihoods does not have; at 64 samples and six alphas a cell has
// 18 208 hypotheses, one log each: 3 % of the sweep's instructions.  Here: x = f 2^k with f in [sqrt(1/2), sqrt(2)),
// log f = 2 atanh(z), z = (f - 1) / (f + 1), |z| <= 0.1716, as the odd series up to z^21 (truncation 2e-17 relative to z);
// ~30 instructions, absolute error <= 2e-16 + 1 ulp of the result.  x = 0 gives -inf (the series' own result would be
// meaningless there), as log does.
__device__ __forceinline__ double pos_log(double x, double k0) {
  int ee;
  double f = frexp(x, &ee);  // [0.5, 1)
  const bool lo = f < 0.70710678118654752440;
  f = lo ? f + f : f;
  const double k = (double)(ee - (lo ? 1 : 0)) + k0;
  const double z = (f - 1.0) / (f + 1.0);
  const double w = z * z;
  double p = 1.0 / 21.0;
  p = fma(p, w, 1.0 / 19.0);
  p = fma(p, w, 1.0 / 17.0);
  p = fma(p, w, 1.0 / 15.0);
  p = fma(p, w, 1.0 / 13.0);
  p = fma(p, w, 1.0 / 11.0);
  p = fma(p, w, 1.0 / 9.0);
  p = fma(p, w, 1.0 / 7.0);
  p = fma(p, w, 1.0 / 5.0);
  p = fma(p, w, 1.0 / 3.0);
  const double zz = z + z;
  const double r = fma(zz * w, p, zz);  // 2 atanh(z)
  // k ln 2 in two pieces (the high one exact for |k| < 2^21), smallest terms first
  const double v = fma(k, 6.93147180369123816490e-01, fma(k, 1.90821492927058770002e-10, r));
  // log's own conventions outside the domain: 0 -> -inf; a negative or NaN product (a corrupt genotype tensor) stays NaN
  // instead of turning into a -inf score that later sums would absorb
  return x > 0.0 ? v : (x == 0.0 ? -__builtin_huge_val() : __builtin_nan(""));
}
__device__ __forceinline__ double prodacc_log(double m, int32_t e) { return pos_log(m, (double)e); }

// ---- the ring of partner values of the wave kernels (demux_wave.hip, fmx_wave.hip) ----
// One 8-byte read per lane from the ring in LDS, at an immediate offset from the lane's slot.  As an opaque instruction
// because two such reads with one base are otherwise merged into a ds_read2_b64, which takes 8 LDS cycles instead of
// 2 + 2 (MI355X_MICROARCH.md, LDS table).  The caller waits (s_waitcnt lgkmcnt) before using the value; "memory" keeps
// the next entry's ring stores behind it.
template <int OFF>
__device__ __forceinline__ double wave_ring_rd(uint32_t a) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
  return v;
}
typedef double dbl2 __attribute__((ext_vector_type(2)));
// ... and 16 bytes (two doubles at a 16-byte aligned address): 4 LDS cycles, and one DS instruction where two 8-byte reads
// are two
template <int OFF>
__device__ __forceinline__ dbl2 wave_ring_rd128(uint32_t a) {
  dbl2 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
  return v;
}
template <int I, int N, class F>
__device__ __forceinline__ void wave_for(F&& f) {  // f(integral_constant<int, I>) ... f(integral_constant<int, N - 1>)
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    wave_for<I + 1, N>(f);
  }
}

// which entries of a cell a wave-per-cell sweep walks: all of them, or those whose bit in the linear-entry set is
// set / clear (plan_build_lin, fmx_entry_kernel)
enum { EM_ALL = 0, EM_LINEAR = 1, EM_GENERAL = 2 };
// first entry >= e in [e, e1) of the kind the launch sweeps (wave-uniform: scalar loads and bit scans)
template <int EM>
__device__ __forceinline__ int64_t wave_next_entry(const uint32_t* __restrict__ lin, int64_t e, int64_t e1) {
  if (EM == EM_ALL) return e;
  while (e < e1) {
    uint32_t w = lin[e >> 5];
    if (EM == EM_GENERAL) w = ~w;
    w >>= (uint32_t)(e & 31);
    if (w) {
      e += __builtin_ctz(w);
      return e < e1 ? e : e1;
    }
    e = (e | 31) + 1;
  }
  return e1;
}

// where the entries [e0, e1) of a work unit lie in the stream of the set (EM_LINEAR) or clear (EM_GENERAL) bits of `bits`
// (plan_build_bit_streams): set bits before e = rank[e / 32] + set bits below e in its word
template <int EM>
__device__ __forceinline__ void wave_stream_range(const uint32_t* __restrict__ bits, const int64_t* __restrict__ rank, int64_t e0,
                                                  int64_t e1, int64_t& i0, int64_t& i1) {
  auto before = [&](int64_t e) {
    int64_t r = rank[e >> 5];
    if (e & 31) r += __popc(bits[e >> 5] & ((1u << (e & 31)) - 1u));
    return r;
  };
  const int64_t l0 = before(e0), l1 = before(e1);
  i0 = EM == EM_LINEAR ? l0 : e0 - l0;
  i1 = EM == EM_LINEAR ? l1 : e1 - l1;
}

// Workgroup -> work-unit index with XCD affinity.  The dispatcher is observed to place workgroup b on XCD b % 8
// (MI355X_MICROARCH.md; used for speed only, nothing depends on it): the work list is cut into 8 contiguous blocks,
// one per XCD, so that workgroups resident on one XCD walk neighbouring work units -- here neighbouring SNP windows of
// the GP tensor, which then fit that XCD's 4 MiB L2.  n8 = ceil(n/8); the result may be >= n (caller checks).
__device__ __forceinline__ int xcd_swizzle(int b, int n8) { return (b & 7) * n8 + (b >> 3); }

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// sc_drop_seq.cpp:5-8
__device__ __forceinline__ double dev_logadd(double la, double lb) {
  if (la > lb) return la + log(1.0 + exp(lb - la));
  return lb + log(1.0 + exp(la - lb));
}

// Work unit of the wave kernels (demux_wave.hip, fmx_wave.hip): a run of entries of one cell.  A cell is walked by one
// wave, so a launch cannot end before its longest cell has been walked; with few cells that walk IS the launch (10 k
// cells: 10 ms for any shape), with many it hides behind throughput-bound work.  Cells longer than 2048 entries are
// therefore cut into equal parts, each with a result slab of its own (part 0: the cell's slab, the others: overflow
// slabs behind the C cell slabs), and the parts' log-likelihoods are added up afterwards.  The cut depends on the cell
// alone, so a cell's result does not depend on which other cells share the handle (shards reproduce the whole run bit
// for bit).
struct wave_item {
  int64_t e0, e1;  // entries
  int64_t slab;    // result slab (in units of one cell's slabs)
  int64_t cell;
};
struct wave_cut {
  int64_t cell, first, count;  // overflow slabs [first, first + count)
};
struct ring_sel {          // one launch of demux_ring.hip
  int32_t n[4];            // alpha indices of the non-symmetric slots
  int32_t nsym;            // ... of the symmetric slot (0: none)
  int32_t with_singlet;    // the launch also fills llw[c][block][0][0][lane]
  int32_t jbase, blk, nblk2;  // first sample of the diagonal block, its slab, slabs per cell
};

// kernel launchers implemented in the kernel TUs
int demux_launch(muxgl_handle* h, const muxgl_demux_params* p);
int demux_entry_pg_launch(muxgl_handle* h, const muxgl_demux_params* p, double* d_pg, bool gen_stream = false,
                          bool by_record = false);
int demux_ring_lin_launch(muxgl_handle* h, const muxgl_demux_params* p, const wave_item* items, int64_t n_items,
                          const double* gm, int na, const ring_sel& sel, double* llw, const double* pgt = nullptr,
                          bool pg_by_record = false);
void demux_ring_release(muxgl_handle* h);
int demux_row_plan(muxgl_handle* h);
int demux_row_launch(muxgl_handle* h, const muxgl_demux_params* p);  // -1: not applicable
void demux_row_free(muxgl_handle* h);
int demux_row_build(muxgl_handle* h, muxgl_row_state** st, int64_t c0, int64_t c1, int ch);
int demux_oct_launch(muxgl_handle* h, const muxgl_demux_params* p);  // -1: not applicable
int demux_row2_launch(muxgl_handle* h, const muxgl_demux_params* p);  // -1: not applicable (demux_row2.hip)
int demux_call16_launch(muxgl_handle* h, const muxgl_demux_params* p);
int fmx_oct_estep_launch(muxgl_handle* h, muxgl_row_state* st, int64_t c0, int64_t nc);  // -1: not applicable
int fmx_row2_estep_launch(muxgl_handle* h, muxgl_row_state* st, int64_t c0, int64_t nc);  // -1: not applicable
int demux_wave_plan(muxgl_handle* h, const int64_t* cell_ptr);
int demux_wave_launch(muxgl_handle* h, const muxgl_demux_params* p);  // -1: not applicable
const int32_t* demux_wave_order(const muxgl_handle* h);  // cells, longest first (device)
int demux_wave_items(const muxgl_handle* h, const wave_item** items, int64_t* n_items, const wave_cut** cuts,
                     int64_t* n_cuts, int64_t* n_over);  // work units of the wave kernels (device)
int fmx_wave_estep_launch(muxgl_handle* h, int64_t c0, int64_t nc);  // 16 < K <= 255; -1: not applicable
int64_t fmx_wave_fll_rows(const muxgl_handle* h);  // rows of d_fll: C + extra parts of long cells
void fmx_wave_streams_release(muxgl_handle* h);  // the linear/general entry streams and their rank table
int demux_ensure_ll(muxgl_handle* h, const muxgl_demux_params* p);  // standard LL tensor allocated and zeroed
int demux_call_wave_launch(muxgl_handle* h, const muxgl_demux_params* p);
void demux_wave_free(muxgl_handle* h);
int demux_gp_neutral_rows(muxgl_handle* h, int V);  // d_gp rows of markers without genotypes := (1,0,0) (demux_wave.hip)
void demux_row_release(muxgl_row_state** st);
int plan_build_chunks(muxgl_handle* h, muxgl_row_state* st, int64_t cb, int64_t ce, int ch);  // chunk tables (plan_kernels.hip)
int plan_build_qent(muxgl_handle* h);       // packed entry records of the quad kernel, on the device (plan_kernels.hip)
int quad_launch_order(muxgl_handle* h, const row_chunk* d_chunks, const int32_t* d_nlin, int64_t n, int32_t** order);
int plan_build_lin(muxgl_handle* h);        // d_lin (plan_kernels.hip)
int plan_build_bit_streams(muxgl_handle* h, const uint32_t* bits, int64_t** rank, fmx_grec** rec_set, fmx_grec** rec_clr,
                           int64_t* n_set);
void plan_lin_streams_release(muxgl_handle* h);
int plan_build_snp_major(muxgl_handle* h);  // d_entry_cell, d_snp_ptr, d_snp_entry, d_snp_cell (plan_kernels.hip)

// handle plumbing shared by muxgl_api.hip and muxgl_group.hip
int muxgl_handle_create(int device, int32_t flags, hipStream_t shared_stream, muxgl_handle** out, std::string* err);
int muxgl_validate_pileup(muxgl_handle* h, int64_t C, int64_t S, int64_t nnz, int64_t R, const int64_t* cell_ptr,
                          const int32_t* entry_snp, const int64_t* entry_rptr, int64_t* maxlen_out);
int muxgl_set_pileup_role(muxgl_handle* h, int role, int64_t C, int64_t S, int64_t nnz, int64_t R, const int64_t* cell_ptr,
                          const int32_t* entry_snp, const int64_t* entry_rptr, const uint8_t* reads, bool trusted = false);
int fmx_attach_column_slab(muxgl_handle* h, int64_t C_total, int64_t c0, int64_t s0, int64_t s1, int64_t nnz_s, int64_t R_s,
                           const int64_t* cell_ptr_s, const int32_t* entry_snp_s, const int64_t* entry_rptr_s,
                           const uint8_t* reads_s, bool trusted);
int fmx_cluster_counts_device(muxgl_handle* m);  // m->d_ccnt := read counts of the cluster pileups (stream-ordered)
int fmx_snp_major_full(muxgl_handle* h);  // d_segls / d_secnt on first use (freemuxlet-old)
int fmx_phase_gp(muxgl_handle* h, const muxgl_fmx_params* p);
int fmx_phase_estep(muxgl_handle* h, const muxgl_fmx_params* p);
int fmx_phase_mstep(muxgl_handle* h);
int fmx_mstep_stream_launch(muxgl_handle* h);  // K <= 64 (fmx_mstep.hip); -1: not applicable
// fmx_exact.hip: near-tie calls settled in the reference's arithmetic (steps A, B, C; all three on one handle)
int fmx_exact_snps(muxgl_handle* h, std::vector<int32_t>* snps);
int fmx_exact_rows(muxgl_handle* h, const muxgl_fmx_params* p, const int32_t* snps, int64_t n, double* rows, uint8_t* owned);
int fmx_exact_finish(muxgl_handle* h, const muxgl_fmx_params* p, const int32_t* snps, int64_t n, const double* rows,
                     int64_t* deltas, int32_t* reassigned);
int fmx_exact_resolve(muxgl_handle* h, const muxgl_fmx_params* p, bool* reassigned);
void fmx_exact_release(muxgl_handle* h);

// device groups (muxgl_group.hip): every entry point of the C-ABI forwards here when h->group is set
int group_create(const muxgl_config* cfg, muxgl_handle** out, std::string* err);
void group_destroy(muxgl_handle* h);
int group_set_pileup(muxgl_handle* h, int64_t C, int64_t S, int64_t nnz, int64_t R, const int64_t* cell_ptr,
                     const int32_t* entry_snp, const int64_t* entry_rptr, const uint8_t* reads);
int group_demux_set_gp(muxgl_handle* h, int32_t V, const double* gp, const uint8_t* has_gp);
int group_demux_run(muxgl_handle* h, const muxgl_demux_params* p, muxgl_demux_cell* out, double* full_ll);
const muxgl_demux_cell* group_demux_results(const muxgl_handle* h);
int group_demux_get_entry_pg(muxgl_handle* h, double* pg);
int group_fmx_prepare(muxgl_handle* h, const double* af, double* cell_llk0, double* cell_llk2, int32_t* cell_nsnps,
                      int32_t* cell_nreads);
int group_fmx_get_entry_gls(muxgl_handle* h, double* gls, int32_t* counts);
int group_fmx_set_clusters(muxgl_handle* h, int32_t K, const int32_t* clust);
int group_fmx_iterate(muxgl_handle* h, const muxgl_fmx_params* p, muxgl_fmx_cell* out, int32_t* nsingle, int32_t* namb,
                      int32_t* nchanged, double* full_ll);
int group_fmx_get_cluster_pileup(muxgl_handle* h, double* gls, int32_t* counts);
void group_fmx_exact_stats(const muxgl_handle* h, int64_t* cells, int64_t* changed, int64_t* unresolved);
void group_peer_stats(const muxgl_handle* h, int32_t* out);
int group_get_timing(const muxgl_handle* h, float* ms);
#define MUXGL_NOT_FOR_GROUPS(h, who) \
  if ((h)->group) MUXGL_FAIL(h, who ": not available on a device group (use a one-device handle)")
