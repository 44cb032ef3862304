/*
 * muxgl.h -- C-ABI of libmuxgl, the MI355X (gfx950) genotype-likelihood engine behind popscle's
 *            demuxlet / freemuxlet hot path.
 *
 * popscle has no plugin/FFI surface: the hot loops are inline in cmdCramDemuxlet / cmdCramFreemux2.  The drop-in
 * boundary is therefore the data hand-over between `scl.load_from_plp(...)` returning
 * (cmd_cram_demuxlet.cpp:122, cmd_cram_freemux2.cpp:87) and the `hprintf` row writers
 * (cmd_cram_demuxlet.cpp:993-1013, cmd_cram_freemux2.cpp:608-665).  Each entry point below names the reference
 * lines it replaces.  INTEGRATION.md shows the glue a popscle maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C, no exceptions across the ABI; every call returns 0 on success, nonzero on failure, and
 *     muxgl_last_error(h) (or muxgl_last_error(NULL) after a failed muxgl_create) describes the failure.  This
 *     replaces the reference's error() -> throw pexception -> abort (Error.cpp:29-43).
 *   - the library needs a HIP device; there is NO CPU fallback.  muxgl_create fails if no gfx950 device is usable.
 *   - all pointer arguments are HOST pointers borrowed for the duration of the call unless the name ends in _dev.
 *   - every call is synchronous: results are complete when it returns (the EM phases of a handle created with
 *     MUXGL_FLAG_ASYNC_PHASES excepted).  The reference is single-threaded and so is each handle: use one handle per
 *     thread / process, on one device or on a device group (muxgl_config).
 *
 * Packed pileup (what sc_dropseq_lib_t holds after load_from_plp, sc_drop_seq.h:130-184, flattened):
 *   cell_ptr   int64[C+1]   entries of cell c = [cell_ptr[c], cell_ptr[c+1])   <- cell_umis[c] (std::map order =
 *                           ascending SNP id, sc_drop_seq.h:165)
 *   entry_snp  int32[nnz]   SNP id of the entry                                <- cell_umis[c] keys
 *   entry_rptr int64[nnz+1] reads of entry e = [entry_rptr[e], entry_rptr[e+1]) <- sc_snp_droplet_t iteration order
 *   reads      uint8[R]     bit7 allele (0 ref, 1 alt), bits0-6 capped BQ; MUXGL_READ_OTHER = allele "2"
 *                           <- (allele<<24 | bq<<16 | count) words of sc_drop_seq.h:23-27
 */
#ifndef MUXGL_H
#define MUXGL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUXGL_VERSION 3
#define MUXGL_READ_OTHER 0xFF
#define MUXGL_MAX_ALPHA 16
#define MUXGL_MAX_DEVICES 16

enum { MUXGL_SNG = 0, MUXGL_DBL = 1, MUXGL_AMB = 2 };

/* muxgl_config.flags */
#define MUXGL_FLAG_FORCE_TILE_SWEEP 1 /* never take the V<=16 row/oct kernels (lets tests cover the general tile sweep) */
#define MUXGL_FLAG_FORCE_ROW_KERNEL 2  /* never take the default-grid oct kernels (V, K <= 16: lets tests cover the row
                                          kernels behind them) */
#define MUXGL_FLAG_FORCE_WAVE_KERNEL 4 /* demuxlet: take the wave kernels for V <= 16 too, and the ring of 32 instead of the
                                          two-per-lane row kernel at 17..32 samples; freemuxlet: never take the
                                          two-per-lane row kernel at 17..32 clusters (test coverage) */
#define MUXGL_FLAG_FORCE_BATCHED_GREEDY 8 /* (no effect since the batched greedy-init kernels became the default for K <= 64;
                                             MUXGL_FLAG_FORCE_TILE_SWEEP selects the serial kernel) */
#define MUXGL_FLAG_DEMUX_ONLY 16   /* device group: the caller will only run demuxlet, so muxgl_set_pileup need not cut
                                      the SNP-major column slabs freemuxlet's ordered M-step works on */
#define MUXGL_FLAG_ASYNC_PHASES 32 /* muxgl_fmx_iter_gp / _estep / _mstep return once their kernels are enqueued on the
                                      handle's stream (muxgl_stream); muxgl_fmx_iter_fetch and every other call still
                                      return with the stream drained.  For callers that order their own collectives
                                      against that stream (popscle_amd/freemuxlet.py) */
#define MUXGL_FLAG_NO_LINEAR_ENTRIES 64 /* sweep every entry through the general three-term form, also those whose
                                          likelihoods are linear in the genotypes (one usable read) and would take the
                                          one-moment form of the oct (V, K <= 16) and wave (V, K > 32) kernels
                                          (lets tests compare the two) */
#define MUXGL_FLAG_NO_PIVOT_SUMS 256 /* freemuxlet E-step beyond 32 clusters: the pair sums of the non-linear entries as the
                                        three-term sums instead of around the lane's smallest term (lets tests compare the
                                        two forms) */
#define MUXGL_FLAG_GROUP_PROBE_SELF 1024 /* device group: muxgl_create also walks the pairs of members that share a device when
                                           it enables peer access (tests: the refusal path on a one-GPU box) */
/* (128 was MUXGL_FLAG_MSTEP_LDS_STATES until round 6: the M-step variant with the cluster states in LDS is retired) */
#define MUXGL_FLAG_SPLIT_GENERAL_SWEEP 512 /* demuxlet beyond 32 samples: sweep the entries with more than one usable read in
                                             launches of their own on top of the linear entries' slab (round 3's scheme)
                                             instead of in the same launch with the same accumulators (lets tests
                                             compare the two) */

typedef struct muxgl_handle muxgl_handle;

/* One handle drives one device (n_devices <= 1) or a device group (n_devices > 1): the same entry points, with the
 * cells of the pileup cut into n_devices contiguous ranges (demuxlet: no exchange; freemuxlet: E-step by cells, ordered
 * M-step by SNPs, two all-gathers per EM iteration as peer-to-peer copies over xGMI).  The reference's own way to use
 * several workers is the same cut at file level (--group-list, README.md:168; sc_drop_seq.cpp:93-101,164-170).  A
 * device may be named more than once (virtual ranks on one GPU: tests). */
typedef struct {
  int32_t struct_size; /* sizeof(muxgl_config) of the header the caller was built against (MUXGL_CONFIG_INIT sets it):
                          muxgl_create rejects any other value instead of reading past a shorter struct */
  int32_t device_id; /* HIP device ordinal this handle runs on (n_devices == 0) */
  int32_t flags;     /* MUXGL_FLAG_* bits, normally 0 */
  int32_t n_devices; /* 0: device_id; >= 1: device_ids[0 .. n_devices) */
  int32_t device_ids[MUXGL_MAX_DEVICES];
} muxgl_config;
#define MUXGL_CONFIG_INIT {(int32_t)sizeof(muxgl_config), 0, 0, 0, {0}}

/* demuxlet parameters: --alpha grid and --doublet-prior (cmd_cram_demuxlet.cpp:32,62-63,85-89) */
typedef struct {
  int32_t n_alpha;
  int32_t _pad;
  double alpha[MUXGL_MAX_ALPHA];
  double doublet_prior;
} muxgl_demux_params;

#define MUXGL_CELL_DEEP_SNG 2
#define MUXGL_CELL_DEEP_DBL 4

/* per-cell demuxlet result: every quantity cmd_cram_demuxlet.cpp:788-991 derives and :993-1013 prints.
 * Sample indices are positions in the GP tensor's V axis; -1 = none.  valid=0: the cell has no entries and the
 * reference prints no row (:653). */
typedef struct {
  int32_t valid; /* bit 0: the cell has entries (0: the reference prints no row, :653).  muxgl_demux_run also sets
                    MUXGL_CELL_DEEP_SNG / MUXGL_CELL_DEEP_DBL where the THIRD-largest log-likelihood of the singlet /
                    doublet scan is within rounding reach of the runner-up (three or more hypotheses contend): read and
                    cleared by muxgl_demux_exact_calls, which then recomputes every hypothesis of that scan */
  int32_t nsnps;
  int32_t type, next_type;
  int32_t sBest, sNext;
  int32_t dBest1, dBest2, dBestA;
  int32_t dNext1, dNext2, dNextA;
  int32_t jBest, kBest, aBest;
  int32_t jNext, kNext, aNext;
  double sngBestLLK, sngNextLLK, dblBestLLK, dblNextLLK;
  double sumLLK, sngLLK;
  double bestLLK, nextLLK;
  double bestPP, sngPP, sngOnlyPP;
} muxgl_demux_cell;

/* freemuxlet parameters: --doublet-prior, --geno-error (cmd_cram_freemux2.cpp:19-20) */
typedef struct {
  double doublet_prior;
  double geno_error;
} muxgl_fmx_params;

/* per-cell freemuxlet state/result (cmd_cram_freemux2.cpp:345-367, printed at :660-665) */
typedef struct {
  int32_t type;  /* 0 SNG, 1 DBL, 2 AMB, -1 never classified */
  int32_t clust; /* clusts[i]: jBest for SNG after an iteration, else -1 */
  int32_t jBest, kBest, jNext, kNext;
  int32_t sBest, sNext, dBest1, dBest2, dNext1, dNext2;
  double bestLLK, nextLLK;
  double sngBestLLK, sngNextLLK, dblBestLLK, dblNextLLK;
  double bestPP, sngPP, sngOnlyPP, sumLLK;
  double sngThirdLLK, dblThirdLLK; /* third-largest log-likelihood of each scan (-1e300: none), see muxgl_demux_cell */
} muxgl_fmx_cell;

/* kernel timings of the most recent *_run / *_iterate call, milliseconds, measured with hipEvents recorded on the
 * stream the kernels were launched on */
enum {
  MUXGL_T_DEMUX_REDUCE = 0, /* row path only: chunk partial log-likelihoods summed per cell */
  MUXGL_T_DEMUX_SWEEP = 1,  /* per-entry likelihoods (a4,a5) fused with the sample-pair x alpha sweep (a6) */
  MUXGL_T_DEMUX_CALL = 2,  /* evidence sums, best/next scans, call (a7-a9) */
  MUXGL_T_DEMUX_D2H = 3,   /* per-cell records to host */
  MUXGL_T_FMX_ENTRY = 4,   /* entry 9-GL pileup + cell scores (b1,b2) */
  MUXGL_T_FMX_GP = 5,      /* cluster genotype posterior tensor */
  MUXGL_T_FMX_ESTEP = 6,   /* E-step pair sweep (b6) */
  MUXGL_T_FMX_CALL = 7,    /* scans + re-assignment (b7,b8 classification) */
  MUXGL_T_FMX_MSTEP = 8,   /* ordered clamped merge (b5,b8) */
  MUXGL_T_FMXOLD_PAIR = 9, /* freemuxlet-old: pairwise droplet distance matrix (c1) */
  MUXGL_T_FMXOLD_VOTE = 10, /* freemuxlet-old: one voting pass (c1) */
  MUXGL_T_FMX_ESTEP_SWEEP = 11, /* the pair-sweep kernel(s) alone, inside MUXGL_T_FMX_ESTEP (which also brackets the
                                   relayout of the cluster posteriors and the reduction of the chunk partials) */
  MUXGL_T_COUNT = 16
};

/* ---- lifetime ------------------------------------------------------------------------------------------------ */
int muxgl_create(const muxgl_config* cfg, muxgl_handle** out);
void muxgl_destroy(muxgl_handle* h);
const char* muxgl_last_error(const muxgl_handle* h);
int muxgl_version(void);

/* device group: what muxgl_create found when it enabled direct peer copies between its members' devices: out[3] = ordered
 * pairs of members on distinct devices (with MUXGL_FLAG_GROUP_PROBE_SELF: all ordered pairs), pairs with peer access
 * enabled, pairs left on the runtime's staged copy (no peer access between the two devices, or MUXGL_GROUP_NO_PEER=1 in
 * the environment).  Either way the copies are hipMemcpyPeerAsync and the results the same. */
int muxgl_group_peer_stats(const muxgl_handle* h, int32_t* out);

/* ---- pileup hand-over: replaces the in-memory result of sc_dropseq_lib_t::load_from_plp
 *      (sc_drop_seq.cpp:103-384; containers sc_drop_seq.h:130-184).  Copies to device memory. --------------------- */
int muxgl_set_pileup(muxgl_handle* h, int64_t C, int64_t S, int64_t nnz, int64_t R, const int64_t* cell_ptr,
                     const int32_t* entry_snp, const int64_t* entry_rptr, const uint8_t* reads);

/* ---- demuxlet ------------------------------------------------------------------------------------------------ */
/* genotype-probability tensor gp[S][V][3] (sc_snp_t::gps, sc_drop_seq.h:29-37, built at sc_drop_seq.cpp:287-315) and
 * has_gp[S] (0 <=> gps == NULL, sc_drop_seq.cpp:258-282) */
int muxgl_demux_set_gp(muxgl_handle* h, int32_t V, const double* gp, const uint8_t* has_gp);

/* replaces the per-cell loop cmd_cram_demuxlet.cpp:636-991.  out: NULL or [C].  full_ll: NULL or [C][V][V][n_alpha]
 * receiving llksAB for the entries the reference ever reads: (j,0,0) and (j,k!=j,n>=1); other slots are 0. */
int muxgl_demux_run(muxgl_handle* h, const muxgl_demux_params* p, muxgl_demux_cell* out, double* full_ll);

/* The calls rounding noise could decide, settled in the reference's own arithmetic -- HOST pass, no device work, no
 * handle (popscle_amd/host/exact_calls.hpp).  The kernels' log-likelihoods equal the reference's to ~1e-12, not to the last
 * bit, and every decision of cmd_cram_demuxlet.cpp:827-837,883-906,925-988 compares two of them.  For the cells of
 * cells[C] (records of muxgl_demux_run) where a comparison's margin is within 1e-9 x max(1, |LL|) this call recomputes the
 * contested hypotheses on the host as the reference does (IEEE doubles in its operation order, glibc log) and rewrites the
 * record from those numbers:
 *   - the best (or next) doublet is an alpha = 0.5 pair: the reference evaluates (j,k) and (k,j) with transposed
 *     summation orders (:738-746) and reports whichever came out larger as DBL.BEST.GUESS, the other as runner-up;
 *     muxgl_demux_run names the pair (lo, hi).  Every such cell is looked at (two hypotheses);
 *   - best and next of a scan, or a +2 threshold, within reach: the named hypotheses are recomputed and compared;
 *   - next and third of a scan within reach (MUXGL_CELL_DEEP_* in `valid`): every hypothesis of that scan is recomputed.
 * Integer fields and the log-likelihoods of recomputed hypotheses then equal the reference's exactly; sumLLK / sngLLK
 * stay as the device summed them.  The pileup and gp / has_gp are the arrays handed to muxgl_set_pileup /
 * muxgl_demux_set_gp.  nthreads host threads (cells are independent).
 * stats: NULL or int64[6] = cells looked at, mirrored pairs put into (hi, lo) order, mirrored pairs whose two orders tied
 * exactly, cells with a near tie other than the mirror, cells that needed every hypothesis of a scan, cells whose call
 * changed beyond the order of a mirrored pair. */
int muxgl_demux_exact_calls(int64_t C, int32_t V, const int64_t* cell_ptr, const int32_t* entry_snp,
                            const int64_t* entry_rptr, const uint8_t* reads, const double* gp, const uint8_t* has_gp,
                            const muxgl_demux_params* p, muxgl_demux_cell* cells, int32_t nthreads, int64_t* stats);

/* pinned host view of the last run's [C] records (valid until the next run or destroy) */
const muxgl_demux_cell* muxgl_demux_results(const muxgl_handle* h);

/* per-entry pGs[nnz][n_alpha*9] of the last run (cmd_cram_demuxlet.cpp:655-725), for parity tests */
int muxgl_demux_get_entry_pg(muxgl_handle* h, double* pg);

/* ---- freemuxlet ---------------------------------------------------------------------------------------------- */
/* b1+b2: calculate_snp_droplet_pileup for every entry (sc_drop_seq.cpp:452-509) and the per-cell singlet scores
 * (cmd_cram_freemux2.cpp:117-160).  af[S] from the .var.gz AF column.  Outputs [C], any may be NULL.
 * The reference sorts the cells by llk2 - llk0 (ties by index, sc_drop_seq.h:190-198) and the greedy start walks them in
 * that order, so where two cells' scores are within 1e-9 x max(1, |llk|) of each other -- droplets of single-read entries
 * score 0 +- rounding noise -- the last bits decide.  When both cell_llk0 and cell_llk2 are asked for on a handle that
 * holds the whole pileup, the sums of every such cell are therefore recomputed exactly as the reference forms them (IEEE
 * operations in its order on the device, glibc log on the host, score_exact.hpp); all other cells keep the device's sums
 * (equal to ~1e-13); a device group does the same, each member for its cells.  A slabbed handle (one rank of a sharded run:
 * it sees only its own cells) returns the device's sums for all of them. */
int muxgl_fmx_prepare(muxgl_handle* h, const double* af, double* cell_llk0, double* cell_llk2, int32_t* cell_nsnps,
                      int32_t* cell_nreads);

/* of the last muxgl_fmx_prepare: the number of cells whose sums were recomputed in the reference's arithmetic */
int muxgl_fmx_score_stats(const muxgl_handle* h, int64_t* exact_scores);

/* entry pileups (for parity checks and callers that want them): gls[nnz][9],
 * counts[nnz][3] = nreads,nref,nalt.  Either may be NULL. */
int muxgl_fmx_get_entry_gls(muxgl_handle* h, double* gls, int32_t* counts);

/* b3+b4: sort cells by singlet score (descending, ties by id descending: cmd_cram_freemux2.cpp:184-189 with the
 * comparator sc_drop_seq.h:187-198) and run the greedy initial clustering (:217-261, distance =
 * sc_dropseq_lib_t::calculate_droplet_clust_distance, sc_drop_seq.cpp:544-578).  scores[C] = cell_scores (llk2-llk0,
 * possibly shuffled by --randomize-singlet-score on the caller side).  The procedure is sequential over cells by
 * construction (each assignment changes the cluster pileups the next cell is scored against): the cells are sorted on
 * the host; the device then decides them in batches of 32, each batch as a fixpoint of the sequential rule (guess from
 * the state at the start of the batch, replay of the predecessors' merges at shared SNPs, iterate until no guess
 * changes), or, beyond 64 clusters or where the batched kernel cannot be resident, with one persistent workgroup walking
 * the cells in order (fmx_greedy.hip).  The kernels form their scores in another association than the reference's
 * (products instead of sums of logs), equal to ~1e-13 relative: a decision whose margin over the runner-up is within
 * 1e-9 x max(1, |score|) is therefore not taken by them but by greedy_exact.hpp, which recomputes that step's K distances
 * in the reference's own arithmetic (IEEE operations in the reference's order on the device, glibc log on the host) given
 * the earlier decisions -- all such steps of a pass in one launch.  Where it overrules the kernel, the steps behind it that
 * cover a SNP of a cell whose decision changed are decided again the same way against the corrected assignments, in
 * order; every other step stands (it read none of the states that changed).  Only if that walk outlasts twice the pass
 * itself is the pass repeated with the decisions so far pinned.  A score of exactly 0 counts as the
 * structural tie of clusters sharing no SNP with the cell only when both of its products are empty.  The result is the
 * reference's clustering, not an approximation of it (muxgl_fmx_greedy_stats reports how often the exact path was taken).
 * One device, whole pileup: not available on a device group or a slabbed handle.
 * clust_out[C] receives the cluster id, or -1 for cells skipped by frac_init_clust / singlet_score_thres. */
int muxgl_fmx_greedy_init(muxgl_handle* h, int32_t K, const double* scores, double frac_init_clust,
                          double singlet_score_thres, int32_t* clust_out);

/* of the last muxgl_fmx_greedy_init: steps whose margin was a near tie and went through the exact path, and how many of
 * those it decided differently from the kernel (each such step repeats the run once).  Either pointer may be NULL. */
int muxgl_fmx_greedy_stats(const muxgl_handle* h, int64_t* near_ties, int64_t* overruled);

/* initial clusters (after --init-cluster or greedy init): builds the cluster pileups in ascending cell id
 * (cmd_cram_freemux2.cpp:277-288) and resets types/jBest/kBest (:263-265,349-350).  clust[C], -1 = unassigned. */
int muxgl_fmx_set_clusters(muxgl_handle* h, int32_t K, const int32_t* clust);

/* one EM iteration, cmd_cram_freemux2.cpp:375-597: E-step, scans, re-assignment, ordered M-step.
 * out: NULL or [C].  full_ll: NULL or [C][K(K+1)/2]. */
int muxgl_fmx_iterate(muxgl_handle* h, const muxgl_fmx_params* p, muxgl_fmx_cell* out, int32_t* nsingle, int32_t* namb,
                      int32_t* nchanged, double* full_ll);

/* Near-tie calls of the EM iterations since muxgl_fmx_set_clusters.  The kernels' log-likelihoods equal the reference's to
 * ~1e-12, not to the last bit; a cell where a comparison of cmd_cram_freemux2.cpp:469-497,521-584 has a margin within
 * 1e-9 x max(1, |LL|) is therefore not decided by them but listed, and its contested hypotheses are recomputed in the
 * reference's own arithmetic (cluster states as ordered clamped chains from the read bytes, IEEE operations in the
 * reference's order on the device, glibc log on the host: fmx_exact.hip); record, assignment and counters are patched
 * and the M-step runs again when an assignment changed.  muxgl_fmx_iterate does all of that itself, on one device and on a
 * device group.  near_tie_cells = cells that went through that path, calls_changed = how many of them it decided
 * differently from the kernels, unresolved = listed cells a caller of the sharded phases left open (see below).  Any
 * pointer may be NULL. */
int muxgl_fmx_exact_stats(const muxgl_handle* h, int64_t* near_tie_cells, int64_t* calls_changed, int64_t* unresolved);

/* The same for the sharded phases (muxgl_fmx_iter_*): a listed cell's SNPs are spread over the ranks' M-step ranges, so the
 * caller carries two small exchanges in the iterations where the job-wide count of listed cells (the fourth counter of
 * MUXGL_BUF_STAT after the all-reduce, or the sum of muxgl_fmx_exact_pending over the ranks) is not zero -- after
 * muxgl_fmx_iter_estep + muxgl_fmx_iter_fetch and BEFORE anything overwrites the assignments of the previous iteration
 * the chains are built from (the library keeps its own copy, taken when the E-step started):
 *   1. muxgl_fmx_exact_snps on every rank: the SNPs its listed cells cover (call with out = NULL for the count, then
 *      with room); the caller unites the lists (ascending, unique);
 *   2. muxgl_fmx_exact_rows on every rank with the united list: rows[n][K][3] and owned[n] are filled for the SNPs of the
 *      rank's own M-step range; the caller gathers every row from its owner;
 *   3. muxgl_fmx_exact_finish on every rank with the complete rows: settles the rank's listed cells, patches records,
 *      assignments (also in MUXGL_BUF_CLUST) and the rank's counters; deltas[3] = what it added to (nsingle, namb,
 *      nchanged), *reassigned != 0: an assignment changed.
 * Then the assignments are exchanged again and, if any rank reports a reassignment, muxgl_fmx_iter_mstep is repeated on
 * all.  popscle_amd/freemuxlet.py run_em is the reference driver. */
int muxgl_fmx_exact_pending(const muxgl_handle* h, int64_t* cells);
/* A settled cell is not listed again while its inputs last: the library keeps the exact scan results per cell and
 * fmx_call_kernel takes them from there as long as no assignment has changed anywhere since (a converged job pays nothing
 * for its near ties).  muxgl_fmx_iterate knows that from its own counters; a caller of the phases says so after every
 * iteration with the JOB-WIDE number of changed cells (after the exact path) -- without this call every E-step starts
 * afresh, which is always correct. */
int muxgl_fmx_exact_hint(muxgl_handle* h, int32_t nchanged_jobwide);
int muxgl_fmx_exact_snps(muxgl_handle* h, int32_t* out, int64_t cap, int64_t* n);
int muxgl_fmx_exact_rows(muxgl_handle* h, const muxgl_fmx_params* p, const int32_t* snps, int64_t n, double* rows,
                         uint8_t* owned);
int muxgl_fmx_exact_finish(muxgl_handle* h, const muxgl_fmx_params* p, const int32_t* snps, int64_t n, const double* rows,
                           int64_t* deltas, int32_t* reassigned);

/* cluster pileups for the .clust1.vcf.gz writer (cmd_cram_freemux2.cpp:608-658): gls[K][S][9], counts[K][S][3] */
int muxgl_fmx_get_cluster_pileup(muxgl_handle* h, double* gls, int32_t* counts);

/* ---- freemuxlet-old (`popscle freemuxlet-old`, cmd_cram_freemuxlet.cpp): the parts that differ from freemux2.  The
 *      entry pileups, scores and the EM loop are the calls above (geno_error = 0 except in the tenth and last
 *      iteration, no early stop: cmd_cram_freemuxlet.cpp:457,485,500); what is particular to the old command is its
 *      initial clustering from a pairwise droplet distance matrix.  rand() and std::random_shuffle stay with the
 *      caller, which passes the visiting orders and the vote jitters in the reference's drawing order. ------------- */

/* dropD (sc_drop_seq.h:45-52) of the cell pair a > b, stored at a(a-1)/2 + b */
typedef struct {
  int32_t nsnps, nread1, nread2, _pad;
  double llk0, llk2;
} muxgl_dropd;

/* pairwise distance matrix, cmd_cram_freemuxlet.cpp:176-221, after muxgl_fmx_prepare.  The device keeps one vote sign
 * per ordered pair (+1: llk2 - llk0 > bf_thres, -1: llk0 - llk2 > bf_thres, :273-278,312-313; C^2 bytes).
 * full: NULL or [C(C-1)/2] records (what --aux-files prints to .ldist.gz, :264, and what parity tests compare). */
int muxgl_fmxold_pair_dist(muxgl_handle* h, double bf_thres, muxgl_dropd* full);

/* the sign matrix as [C][C] bytes (tests) */
int muxgl_fmxold_get_signs(muxgl_handle* h, int8_t* out);

/* first-pass voting, cmd_cram_freemuxlet.cpp:245-291.  order[C] = drops_srted (muxgl_fmx_greedy_init's sort order:
 * score descending, ties by id descending); jitter[n_visited][K] = the values `rand()/(RAND_MAX+1.)/1000.` drawn for
 * the visited cells (:257-259).  clust_out[C]: cluster, or -1 for cells beyond frac_init_clust; ccounts: NULL or [K].
 * K <= 64. */
int muxgl_fmxold_vote_init(muxgl_handle* h, int32_t K, const int32_t* order, const double* jitter,
                           double frac_init_clust, int32_t* clust_out, int32_t* ccounts);

/* one refinement pass, cmd_cram_freemuxlet.cpp:297-343 (one value of `iter`).  order[C] = orand after
 * std::random_shuffle (:301-302), jitter[C][K] in visiting order.  clust_inout[C] is updated in place; changed and
 * ccounts[K] as the reference reports them (:346-349). */
int muxgl_fmxold_vote_refine(muxgl_handle* h, int32_t K, const int32_t* order, const double* jitter,
                             int32_t keep_init_missing, int32_t* clust_inout, int32_t* changed, int32_t* ccounts);

/* ---- sharded EM: one handle per rank (one process per GPU, popscle_amd/freemuxlet.py; a device group does the same
 *      inside one process, see muxgl_config).  Rank r owns a cell range [c0,c1) for the E-step / scans / re-assignment
 *      and a SNP range [s0,s1) for the cluster GP rows and the ordered M-step (a chain per (cluster, SNP) in ascending
 *      cell id with a clamp after every merge, sc_drop_seq.h:77-101: not an associative reduction, so exactly one rank
 *      evaluates it).  One iteration =
 *          iter_gp -> all-gather MUXGL_BUF_CGP slices [s0*K*3, s1*K*3) -> iter_estep -> all-gather MUXGL_BUF_CLUST
 *          slices [c0,c1) (+ sum of the three counters) -> iter_mstep.
 *      The collectives are the caller's (RCCL over xGMI via torch.distributed in this repo); muxgl_fmx_buffer exposes
 *      the device buffers they run on, muxgl_stream the stream the phases are enqueued on.  Both buffers have
 *      MUXGL_XCHG_PAD units of slack behind them, so equal slices of ceil(n / ranks) units can be gathered in place.
 *      With the full ranges the phases reproduce muxgl_fmx_iterate bit for bit.
 *
 *      What a rank holds, two ways:
 *        a) slabs -- muxgl_set_pileup(row slab: the rank's cells [c0,c1) with every SNP, cells renumbered from 0) then
 *           muxgl_fmx_set_column_slab(column slab: all C_total cells, only the entries with s0 <= SNP < s1, SNP ids
 *           unchanged), then muxgl_fmx_prepare / muxgl_fmx_set_clusters(clust[C_total]) / the phases.  Device memory
 *           and H2D traffic are 2/ranks of the pileup.  Per-cell outputs (prepare's scores, iter_fetch's records) cover
 *           the rank's own cells [C = c1-c0]; MUXGL_BUF_CLUST is job-wide [C_total], MUXGL_BUF_CGP is [S][K][3];
 *           muxgl_fmx_get_cluster_pileup returns the rows of [s0,s1) and zeros elsewhere.
 *        b) the whole pileup on every rank + muxgl_fmx_set_shard(c0,c1,s0,s1) after muxgl_fmx_prepare (simple, but every
 *           rank uploads and keeps everything; kept for callers that hold the whole pileup anyway). ------------------- */
enum { MUXGL_BUF_CGP = 0 /* f64[S][K][3] */, MUXGL_BUF_CLUST = 1 /* i32[C_total] */, MUXGL_BUF_CELLS = 2 /* muxgl_fmx_cell[C] */,
       MUXGL_BUF_STAT = 3 /* i32[4]: nsingle, namb, nchanged of the local cell range */ };
#define MUXGL_XCHG_PAD 64
int muxgl_fmx_set_column_slab(muxgl_handle* h, int64_t C_total, int64_t c0, int64_t s0, int64_t s1, int64_t nnz_s,
                              int64_t R_s, const int64_t* cell_ptr_s /*[C_total+1]*/, const int32_t* entry_snp_s,
                              const int64_t* entry_rptr_s, const uint8_t* reads_s);
int muxgl_fmx_set_shard(muxgl_handle* h, int64_t c0, int64_t c1, int64_t s0, int64_t s1);
int muxgl_fmx_iter_gp(muxgl_handle* h, const muxgl_fmx_params* p);
int muxgl_fmx_iter_estep(muxgl_handle* h, const muxgl_fmx_params* p);
int muxgl_fmx_iter_mstep(muxgl_handle* h);
/* counters of the last E-step and, when out / full_ll are given, the records / E-step tensor of the own cells.  With
 * out == NULL and full_ll == NULL it waits for the counters only (work enqueued behind the E-step keeps running). */
int muxgl_fmx_iter_fetch(muxgl_handle* h, muxgl_fmx_cell* out, int32_t* nsingle, int32_t* namb, int32_t* nchanged,
                         double* full_ll);
int muxgl_fmx_buffer(muxgl_handle* h, int32_t which, void** dev_ptr, int64_t* n_elems);
int muxgl_memcpy_dev(muxgl_handle* h, void* dst_dev, const void* src_dev, int64_t bytes);
/* the hipStream_t every kernel of this handle is enqueued on (NULL for a device group) */
void* muxgl_stream(const muxgl_handle* h);

/* ---- measurement --------------------------------------------------------------------------------------------- */
/* ms[MUXGL_T_COUNT]: hipEvent durations of the kernels of the most recent run/iterate call (0 where not run) */
int muxgl_get_timing(const muxgl_handle* h, float* ms);
/* ms_sum[MUXGL_T_COUNT]: the same durations summed since the last reset -- every event bracket is added exactly once,
 * whichever entry point collects it -- and `calls`, the number of collecting calls: one per muxgl_demux_run /
 * muxgl_fmx_iterate, and with the phased API one per phase that reads its brackets back (iter_gp, iter_estep,
 * iter_mstep; under MUXGL_FLAG_ASYNC_PHASES only an iter_fetch that drains the stream).  Either pointer may be NULL;
 * reset != 0 clears both afterwards.  For loops that time many calls without a fetch per call.  One-device handles. */
int muxgl_get_timing_sum(muxgl_handle* h, double* ms_sum, int64_t* calls, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif
