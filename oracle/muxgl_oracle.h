/*
 * muxgl_oracle.h -- CPU restatement of popscle's demuxlet / freemuxlet genotype-likelihood path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under popscle_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY STATUS (round 5): PINNED to the reference's own code for every arithmetic row of the path; "parity unpinned"
 * remains for the file parsers and text writers only.
 *   - The reference (statgen/popscle) ships no tests, fixtures or golden vectors, and its command translation units
 *     include htslib headers (absent from this image), so the binaries cannot be built here and no stand-in header,
 *     type or function body is written for them.
 *   - What does compile from the reference's own text, unmodified, where it lies under /root/reference
 *     (oracle/Makefile; nothing is copied into the repository):
 *       _ref/libphred_ref.so   PhredHelper.cpp                                   -> oracle_phred_tables
 *       _ref/libmerge_ref.so   sc_drop_seq.h (merge(), sc_drop_comp_t)           -> oracle_plp_merge*, oracle_fmx_sort
 *       _ref/libscdrop_ref.so  sc_drop_seq.cpp:1-92,386-578 (logAdd, add_snp/add_cell/add_read with the real
 *                              std::map<std::string> containers, calculate_snp_droplet_pileup,
 *                              calculate_droplet_clust_distance) and, as VERBATIM LINE RANGES inside wrapper functions
 *                              that only declare the locals those lines name (oracle/ref_hot.cpp.in),
 *                              cmd_cram_demuxlet.cpp:428-440,590-622,634-991 (the whole droplet loop up to the hprintf)
 *                              and cmd_cram_freemux2.cpp:108-109,114-159,184-189,192-262,277-288,350-370,373-605
 *                              (entry pileups + singlet scores, sort, given / greedy initial clusters, cluster
 *                              pileups, the EM loop with its early stop).
 *     tests/test_oracle_ref.py holds every function below (except the freemuxlet-old block) to those libraries BIT FOR
 *     BIT -- all record fields, the full LL tensors, cluster pileups after every M-step, on shallow and deep entries,
 *     allele "2", Q < 2, missing genotypes, empty cells, the nAlpha == 1 and nv == 1 quirks, --init-cluster starts,
 *     frac-init / score-threshold skips -- and tests/golden/*.npz (demux_*, fmx_k4*) are outputs of those libraries.
 *   - Still unpinned (need htslib / tsv_reader): load_from_plp's file parsing, parse_posteriors (VCF -> GP), the
 *     hprintf row formats; freemuxlet-old's pairwise init (out of scope).
 *   - Everything below is a literal, operation-order-preserving restatement, each function citing the
 *     reference file:line it follows (paths relative to the reference root).
 *
 * Packed pileup (same layout the C-ABI in include/muxgl.h takes):
 *   cell_ptr   int64[C+1]   entries of cell c are [cell_ptr[c], cell_ptr[c+1]), ascending SNP id
 *   entry_snp  int32[nnz]   SNP id of each (cell,SNP) entry
 *   entry_rptr int64[nnz+1] reads of entry e are [entry_rptr[e], entry_rptr[e+1]), in the reference's iteration
 *                           order (std::map<std::string UMI> order, sc_drop_seq.h:26)
 *   reads      uint8[R]     bit7 = allele (0 ref / 1 alt), bits0-6 = capped base quality; 0xFF = allele "2" (other)
 */
#ifndef MUXGL_ORACLE_H
#define MUXGL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_READ_OTHER 0xFF

/* droplet types; the reference prints the strings, freemux2 stores 0/1/2 (cmd_cram_freemux2.cpp:528,540,568) */
enum { ORACLE_SNG = 0, ORACLE_DBL = 1, ORACLE_AMB = 2 };

/* per-cell result of demuxlet, one field per quantity of cmd_cram_demuxlet.cpp:788-1013 */
typedef struct {
  int32_t valid;                 /* 0: cell has no entries -> the reference emits no row (:653) */
  int32_t nsnps;                 /* scl.cell_umis[i].size() (:996) */
  int32_t type, next_type;       /* bestType / nextType */
  int32_t sBest, sNext;          /* :827-837 */
  int32_t dBest1, dBest2, dBestA;/* :883-906 */
  int32_t dNext1, dNext2, dNextA;
  int32_t jBest, kBest, aBest;   /* :921-988 */
  int32_t jNext, kNext, aNext;
  double sngBestLLK, sngNextLLK, dblBestLLK, dblNextLLK;
  double sumLLK, sngLLK;
  double bestLLK, nextLLK;
  double bestPP, sngPP, sngOnlyPP;
} oracle_demux_cell;

/* per-(cell,SNP) or per-(cluster,SNP) pileup, sc_drop_seq.h:65-75 minus logdenom (never read outside the struct) */
typedef struct {
  int32_t nreads, nref, nalt, _pad;
  double gls[9];
} oracle_plp;

/* per-cell result of one freemuxlet EM iteration / of the final table (cmd_cram_freemux2.cpp:458-584,660-665) */
typedef struct {
  int32_t type;                  /* 0 SNG, 1 DBL, 2 AMB, -1 never classified */
  int32_t clust;                 /* clusts[i] after the iteration: jBest for SNG, else -1 */
  int32_t jBest, kBest, jNext, kNext;
  int32_t sBest, sNext, dBest1, dBest2, dNext1, dNext2;
  double bestLLK, nextLLK;
  double sngBestLLK, sngNextLLK, dblBestLLK, dblNextLLK;
  double bestPP, sngPP, sngOnlyPP, sumLLK;
} oracle_fmx_cell;

/* PhredHelper.cpp:24-41 */
void oracle_phred_tables(double* err256, double* mat256);

/* sc_drop_seq.cpp:5-8 */
double oracle_logadd(double la, double lb);

/* demuxlet, cmd_cram_demuxlet.cpp:636-991.  gp = [S][V][3] doubles, has_gp = [S] (0 => gps==NULL, :733).
 * full_ll: NULL or [C][V][V][nAlpha] receiving llksAB of every cell.  nthreads<=1: serial, else OpenMP over cells
 * (cells are independent; results do not depend on nthreads).  Returns 0. */
int oracle_demux(int64_t C, int64_t S, int32_t V,
                 const int64_t* cell_ptr, const int32_t* entry_snp, const int64_t* entry_rptr, const uint8_t* reads,
                 const double* gp, const uint8_t* has_gp,
                 int32_t nAlpha, const double* alphas, double doublet_prior,
                 oracle_demux_cell* out, double* full_ll, int32_t nthreads);

/* per-entry normalised doublet-genotype likelihoods pGs[nAlpha*9] of demuxlet, cmd_cram_demuxlet.cpp:655-725 */
void oracle_demux_entry_pg(const uint8_t* reads, int64_t nreads, int32_t nAlpha, const double* alphas, double* pGs);

/* freemuxlet b1: calculate_snp_droplet_pileup(alpha=0.5), sc_drop_seq.cpp:452-509, for every entry */
void oracle_fmx_entry_pileup(int64_t nnz, const int64_t* entry_rptr, const uint8_t* reads, oracle_plp* out);

/* snp_droplet_pileup::merge, sc_drop_seq.h:77-101 */
void oracle_plp_merge(oracle_plp* dst, const oracle_plp* src);

/* chains of merges from default-constructed pileups; elements ptr[i] .. ptr[i+1]-1 of chain i in order */
void oracle_plp_merge_chains(int64_t nchains, const int64_t* ptr, const oracle_plp* elems, oracle_plp* out);

/* freemuxlet b2: per-cell llk0/llk2, nSNPs, nReads, cmd_cram_freemux2.cpp:117-160 */
void oracle_fmx_cell_scores(int64_t C, const int64_t* cell_ptr, const int32_t* entry_snp, const oracle_plp* eplp,
                            const double* af, double* llk0, double* llk2, int32_t* nsnps, int32_t* nreads);

/* freemuxlet b3: order = cells sorted by score descending, ties by id descending
 * (cmd_cram_freemux2.cpp:184-189, comparator sc_drop_seq.h:187-198) */
void oracle_fmx_sort(int64_t C, const double* scores, int32_t* order);

/* freemuxlet b4: the droplet-to-cluster distance of the greedy loop, sc_drop_seq.cpp:544-578.  d[n], c[n], present[n],
 * af[n] aligned to the droplet's entries (ascending marker id); out = {llk0, llk2}, counts = {nsnps, nread1, nread2} */
void oracle_fmx_clust_distance(int64_t n, const oracle_plp* d, const oracle_plp* c, const uint8_t* present,
                               const double* af, double* out, int32_t* counts);

/* freemuxlet b4: greedy initial clustering, cmd_cram_freemux2.cpp:217-261 + sc_drop_seq.cpp:544-578.
 * clust[C] receives the cluster id (or -1 if skipped by frac_init / score threshold). */
void oracle_fmx_greedy_init(int64_t C, int64_t S, int32_t K, const int64_t* cell_ptr, const int32_t* entry_snp,
                            const oracle_plp* eplp, const double* af, const double* scores, const int32_t* order,
                            double frac_init_clust, double singlet_score_thres, int32_t* clust);

/* the same, also returning the K distances llk2 - llk0 (:235-240) of every visited cell in visiting order:
 * step_scores[visit][K] */
void oracle_fmx_greedy_init_scores(int64_t C, int64_t S, int32_t K, const int64_t* cell_ptr, const int32_t* entry_snp,
                                   const oracle_plp* eplp, const double* af, const double* scores, const int32_t* order,
                                   double frac_init_clust, double singlet_score_thres, int32_t* clust,
                                   double* step_scores);

/* cluster pileup from assignments, ascending cell id, cmd_cram_freemux2.cpp:277-288.
 * cplp = [K][S], default-constructed (gls all 1, counts 0) where nothing merged. */
void oracle_fmx_build_cluster_pileup(int64_t C, int64_t S, int32_t K, const int64_t* cell_ptr, const int32_t* entry_snp,
                                     const oracle_plp* eplp, const int32_t* clust, oracle_plp* cplp);

/* one EM iteration, cmd_cram_freemux2.cpp:375-597: E-step + scans for all cells from cplp, then re-assignment and the
 * sequential M-step that rebuilds cplp in place.  cells[C] carries types/jBest/kBest from the previous iteration
 * (initialise with oracle_fmx_init_cells).  Returns nchanged; writes nsingle/namb. */
int32_t oracle_fmx_iterate(int64_t C, int64_t S, int32_t K, const int64_t* cell_ptr, const int32_t* entry_snp,
                           const oracle_plp* eplp, const double* af, double doublet_prior, double geno_error,
                           oracle_plp* cplp, oracle_fmx_cell* cells, int32_t* nsingle, int32_t* namb,
                           double* full_ll /* NULL or [C][K(K+1)/2] */, int32_t nthreads);

/* state before the first iteration: types[i] = 0 if clust[i]>=0 else -1 (:194,213,244), jBest=kBest=-1 (:349-350) */
void oracle_fmx_init_cells(int64_t C, const int32_t* clust, oracle_fmx_cell* cells);

/* ---- freemuxlet-old (cmd_cram_freemuxlet.cpp), the rows that differ from freemux2 ------------------------------- */

/* dropD of cmd_cram_freemuxlet.cpp:176-221 (struct in sc_drop_seq.h): one record per cell pair a > b, stored at
 * a(a-1)/2 + b */
typedef struct {
  int32_t nsnps, nread1, nread2, _pad;
  double llk0, llk2;
} oracle_dropd;

/* pairwise distance matrix, cmd_cram_freemuxlet.cpp:187-221: SNPs ascending, then cell pairs (a, b < a) of the SNP's
 * std::map order.  out = [C(C-1)/2], zero-initialised here. */
void oracle_fmxold_pair_dist(int64_t C, int64_t S, const int64_t* cell_ptr, const int32_t* entry_snp,
                             const oracle_plp* eplp, const double* af, oracle_dropd* out);

/* first-pass voting, cmd_cram_freemuxlet.cpp:245-291.  order[C] = drops_srted; jitter[n_visited][K] holds the values
 * `rand()/(RAND_MAX+1.)/1000.` the reference draws for the visited cells, in drawing order (the RNG stays with the
 * caller).  clust[C] is set to -1 and then filled for the visited cells; ccounts[K] receives the cluster sizes. */
void oracle_fmxold_vote_init(int64_t C, int32_t K, const oracle_dropd* dd, const int32_t* order, const double* jitter,
                             double bf_thres, double frac_init_clust, int32_t* clust, int32_t* ccounts);

/* one refinement pass, cmd_cram_freemuxlet.cpp:297-343.  order[C] = orand (after std::random_shuffle),
 * jitter[C][K] in visiting order.  Returns the number of changed cells; ccounts[K] as the reference counts them. */
int32_t oracle_fmxold_vote_refine(int64_t C, int32_t K, const oracle_dropd* dd, const int32_t* order,
                                  const double* jitter, double bf_thres, int32_t keep_init_missing, int32_t* clust,
                                  int32_t* ccounts);

#ifdef __cplusplus
}
#endif
#endif
