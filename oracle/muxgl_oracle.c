/*
 * muxgl_oracle.c -- CPU restatement of popscle's demuxlet / freemuxlet genotype-likelihood path.
 *
 * TEST INFRASTRUCTURE ONLY (see muxgl_oracle.h).  PARITY STATUS: pinned bit for bit to the reference's own code
 * compiled from /root/reference (oracle/_ref/libphred_ref.so, libmerge_ref.so, libscdrop_ref.so;
 * tests/test_oracle.py, tests/test_oracle_ref.py) for every function except the freemuxlet-old block; the file
 * parsers and text writers of the reference (htslib) stay "parity unpinned" -- see muxgl_oracle.h.
 *
 * Rules followed here: IEEE doubles, the reference's operation order and association, glibc log/exp/pow,
 * no FMA contraction (-ffp-contract=off), no reassociation.  Every function cites the reference lines it restates.
 */
#include "muxgl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MIN_NORM_GL 1e-6 /* sc_drop_seq.h:14 */

/* ------------------------------------------------------------------------------------------------ Phred LUT */

/* PhredHelper.cpp:24-41: phred2Err[i] = (i > 1) ? pow(0.1, i*0.1) : 0.75; phred2Mat[i] = 1.-phred2Err[i] */
void oracle_phred_tables(double* err256, double* mat256) {
  for (int i = 0; i <= 255; i++) {
    err256[i] = (i > 1) ? pow(0.1, i * 0.1) : 0.75;
    mat256[i] = 1. - err256[i];
  }
}

static double g_err[256], g_mat[256];
static int g_lut_ready = 0;
static void lut_init(void) {
  if (!g_lut_ready) {
    oracle_phred_tables(g_err, g_mat);
    g_lut_ready = 1;
  }
}

/* sc_drop_seq.cpp:5-8 */
double oracle_logadd(double la, double lb) {
  if (la > lb) {
    return la + log(1.0 + exp(lb - la));
  } else {
    return lb + log(1.0 + exp(la - lb));
  }
}

/* ------------------------------------------------------------------------------------------------ demuxlet */

/* cmd_cram_demuxlet.cpp:655-725: per-read update, divide by running max, +1e-10 floor, divide by max */
void oracle_demux_entry_pg(const uint8_t* reads, int64_t nreads, int32_t nAlpha, const double* alphas, double* pGs) {
  lut_init();
  for (int32_t i = 0; i < nAlpha * 9; ++i) pGs[i] = 1.0; /* :657 */
  for (int64_t r = 0; r < nreads; ++r) {                  /* :660 */
    uint8_t b = reads[r];
    if (b == ORACLE_READ_OTHER) continue;                 /* :664  if ( al == 2 ) continue; */
    uint8_t al = b >> 7;
    uint8_t bq = b & 0x7f;
    double pR = (al == 0) ? g_mat[bq] : g_err[bq] / 3.0;  /* :666 */
    double pA = (al == 1) ? g_mat[bq] : g_err[bq] / 3.0;  /* :667 */
    double maxpG = 0;
    for (int32_t k = 0; k < nAlpha; ++k) {
      for (int32_t l = 0; l < 3; ++l) {
        for (int32_t m = 0; m < 3; ++m) {
          double p = 0.5 * l + (m - l) * 0.5 * alphas[k];  /* :673 */
          double* pG = &pGs[k * 9 + l * 3 + m];
          *pG *= (pR * (1.0 - p) + pA * p);                 /* :685 */
          if (maxpG < *pG) maxpG = *pG;
        }
      }
    }
    for (int32_t i = 0; i < nAlpha * 9; ++i) pGs[i] /= maxpG; /* :692-699 */
  }
  double maxpG = 0;                                        /* :705 */
  for (int32_t i = 0; i < nAlpha * 9; ++i) {
    pGs[i] += 1e-10;                                       /* :711 */
    if (maxpG < pGs[i]) maxpG = pGs[i];
  }
  for (int32_t i = 0; i < nAlpha * 9; ++i) pGs[i] /= maxpG; /* :719-725 */
}

/* cmd_cram_demuxlet.cpp:788-991 for one cell; llksAB = [nv][nv][nAlpha] */
static void demux_call_cell(int32_t nv, int32_t nAlpha, const double* gridAlpha, double doublet_prior,
                            const double* llksAB, oracle_demux_cell* o) {
  int32_t j, k, n;
  int32_t sBest = -1, sNext = -1, dBest1 = -1, dBest2 = -1, dNext1 = -1, dNext2 = -1, dblBestAlpha = -1,
          dblNextAlpha = -1; /* :788 */
  double sngBestLLK = -1e300, sngNextLLK = -1e300;
  double dblBestLLK = -1e300, dblNextLLK = -1e300;
  double sumLLK = -1e-300, sngLLK = -1e-300; /* :791 (sic: -1e-300) */
  double bestPP = -1e300, sngPP = -1e300, sngOnlyPP = -1e300;
  double log_single_prior = log((1.0 - doublet_prior) / nv);                            /* :793 */
  double log_doublet_prior1 = log(doublet_prior / nv / (nv - 1.) / (nAlpha - 1.));      /* :794 */
  double log_doublet_prior2 = log(doublet_prior / nv / (nv - 1.) / (nAlpha - 1.) * 2);  /* :795 */

  for (j = 0; j < nv; ++j) { /* :804-821 */
    sumLLK = oracle_logadd(sumLLK, llksAB[j * nv * nAlpha] + log_single_prior);
    sngLLK = oracle_logadd(sngLLK, llksAB[j * nv * nAlpha] + log_single_prior);
    for (k = 0; k < nv; ++k) {
      if (j == k) continue;
      for (n = 1; n < nAlpha; ++n) {
        if (gridAlpha[n] == 0.5) {
          if (k > j) continue;
          sumLLK = oracle_logadd(sumLLK, llksAB[j * nv * nAlpha + k * nAlpha + n] + log_doublet_prior2);
        } else
          sumLLK = oracle_logadd(sumLLK, llksAB[j * nv * nAlpha + k * nAlpha + n] + log_doublet_prior1);
      }
    }
  }

  for (j = 0; j < nv; ++j) { /* :827-837 */
    if (sngBestLLK < llksAB[j * nv * nAlpha]) {
      sngNextLLK = sngBestLLK;
      sNext = sBest;
      sBest = j;
      sngBestLLK = llksAB[j * nv * nAlpha];
    } else if (sngNextLLK < llksAB[j * nv * nAlpha]) {
      sNext = j;
      sngNextLLK = llksAB[j * nv * nAlpha];
    }
  }

  for (j = 0; j < nv; ++j) { /* :883-906 */
    for (k = 0; k < nv; ++k) {
      if (j == k) continue;
      for (n = 1; n < nAlpha; ++n) {
        double v = llksAB[j * nv * nAlpha + k * nAlpha + n];
        if (dblBestLLK < v) {
          dNext1 = dBest1;
          dNext2 = dBest2;
          dblNextAlpha = dblBestAlpha;
          dblNextLLK = dblBestLLK;
          dBest1 = j;
          dBest2 = k;
          dblBestAlpha = n;
          dblBestLLK = v;
        } else if (dblNextLLK < v) {
          dNext1 = j;
          dNext2 = k;
          dblNextAlpha = n;
          dblNextLLK = v;
        }
      }
    }
  }

  int32_t bestType, nextType;
  int32_t jBest = -1, kBest = -1, jNext = -1, kNext = -1, alphaBest = -1, alphaNext = -1;
  double bestLLK = -1e300, nextLLK = -1e300;

  if (dblBestLLK > sngBestLLK + 2) { /* :925-946 */
    bestType = ORACLE_DBL;
    bestPP = exp(dblBestLLK + ((gridAlpha[dblBestAlpha] == 0.5) ? log_doublet_prior2 : log_doublet_prior1) - sumLLK);
    jBest = dBest1;
    kBest = dBest2;
    bestLLK = dblBestLLK;
    alphaBest = dblBestAlpha;
    if (dblNextLLK > sngBestLLK + 2) {
      nextType = ORACLE_DBL;
      jNext = dNext1;
      kNext = dNext2;
      nextLLK = dblNextLLK;
      alphaNext = dblNextAlpha;
    } else {
      nextType = ORACLE_SNG;
      jNext = kNext = sBest;
      nextLLK = sngBestLLK;
      alphaNext = 0;
    }
  } else if (sngBestLLK > sngNextLLK + 2) { /* :947-967 */
    bestType = ORACLE_SNG;
    bestPP = sngBestLLK + log_single_prior - sumLLK; /* not exponentiated in the reference (:949) */
    jBest = kBest = sBest;
    bestLLK = sngBestLLK;
    alphaBest = 0;
    if (dblBestLLK > sngNextLLK + 2) {
      nextType = ORACLE_DBL;
      jNext = dBest1;
      kNext = dBest2;
      nextLLK = dblBestLLK;
      alphaNext = dblBestAlpha;
    } else {
      nextType = ORACLE_SNG;
      jNext = kNext = sNext;
      nextLLK = sngNextLLK;
      alphaNext = 0;
    }
  } else { /* :968-988 */
    bestType = ORACLE_AMB;
    bestPP = sngBestLLK + log_single_prior - sumLLK;
    jBest = kBest = sBest;
    bestLLK = sngBestLLK;
    alphaBest = 0;
    if (dblBestLLK > sngNextLLK + 2) {
      nextType = ORACLE_DBL;
      jNext = dBest1;
      kNext = dBest2;
      nextLLK = dblBestLLK;
      alphaNext = dblBestAlpha;
    } else {
      nextType = ORACLE_SNG;
      jNext = kNext = sNext;
      nextLLK = sngNextLLK;
      alphaNext = 0;
    }
  }
  sngPP = exp(sngLLK - sumLLK);                           /* :990 */
  sngOnlyPP = exp(sngBestLLK + log_single_prior - sngLLK); /* :991 */

  o->type = bestType;
  o->next_type = nextType;
  o->sBest = sBest;
  o->sNext = sNext;
  o->dBest1 = dBest1;
  o->dBest2 = dBest2;
  o->dBestA = dblBestAlpha;
  o->dNext1 = dNext1;
  o->dNext2 = dNext2;
  o->dNextA = dblNextAlpha;
  o->jBest = jBest;
  o->kBest = kBest;
  o->aBest = alphaBest;
  o->jNext = jNext;
  o->kNext = kNext;
  o->aNext = alphaNext;
  o->sngBestLLK = sngBestLLK;
  o->sngNextLLK = sngNextLLK;
  o->dblBestLLK = dblBestLLK;
  o->dblNextLLK = dblNextLLK;
  o->sumLLK = sumLLK;
  o->sngLLK = sngLLK;
  o->bestLLK = bestLLK;
  o->nextLLK = nextLLK;
  o->bestPP = bestPP;
  o->sngPP = sngPP;
  o->sngOnlyPP = sngOnlyPP;
}

int oracle_demux(int64_t C, int64_t S, int32_t V, const int64_t* cell_ptr, const int32_t* entry_snp,
                 const int64_t* entry_rptr, const uint8_t* reads, const double* gp, const uint8_t* has_gp,
                 int32_t nAlpha, const double* alphas, double doublet_prior, oracle_demux_cell* out, double* full_ll,
                 int32_t nthreads) {
  (void)S;
  lut_init();
  const int32_t nv = V;
  const size_t nll = (size_t)nv * nv * nAlpha;
  if (nthreads < 1) nthreads = 1;

#pragma omp parallel num_threads(nthreads)
  {
    double* llksAB = (double*)malloc(sizeof(double) * (nll ? nll : 1));
    double* pGs = (double*)malloc(sizeof(double) * nAlpha * 9);
    double* sumPs = (double*)malloc(sizeof(double) * nAlpha);
#pragma omp for schedule(dynamic, 4)
    for (int64_t i = 0; i < C; ++i) {
      oracle_demux_cell* o = &out[i];
      memset(o, 0, sizeof(*o));
      memset(llksAB, 0, sizeof(double) * nll); /* :643 */
      int64_t e0 = cell_ptr[i], e1 = cell_ptr[i + 1];
      o->nsnps = (int32_t)(e1 - e0);
      if (e1 == e0) { /* :653  if ( snps.empty() ) continue; */
        o->valid = 0;
        if (full_ll) memset(full_ll + (size_t)i * nll, 0, sizeof(double) * nll);
        continue;
      }
      o->valid = 1;
      for (int64_t e = e0; e < e1; ++e) { /* :656 */
        oracle_demux_entry_pg(reads + entry_rptr[e], entry_rptr[e + 1] - entry_rptr[e], nAlpha, alphas, pGs);
        int32_t isnp = entry_snp[e];
        if (has_gp[isnp]) { /* :733 */
          const double* g = gp + (size_t)isnp * nv * 3;
          for (int32_t j = 0; j < nv; ++j) {
            for (int32_t k = 0; k < nv; ++k) {
              for (int32_t n = 0; n < nAlpha; ++n) sumPs[n] = 0; /* :737 */
              for (int32_t l = 0; l < 3; ++l) {
                for (int32_t m = 0; m < 3; ++m) {
                  double p = g[j * 3 + l] * g[k * 3 + m];                         /* :740 */
                  for (int32_t n = 0; n < nAlpha; ++n) sumPs[n] += (p * pGs[n * 9 + l * 3 + m]); /* :742 */
                }
              }
              for (int32_t n = 0; n < nAlpha; ++n) llksAB[(size_t)j * nv * nAlpha + k * nAlpha + n] += log(sumPs[n]); /* :746 */
            }
          }
        }
      }
      demux_call_cell(nv, nAlpha, alphas, doublet_prior, llksAB, o);
      if (full_ll) memcpy(full_ll + (size_t)i * nll, llksAB, sizeof(double) * nll);
    }
    free(llksAB);
    free(pGs);
    free(sumPs);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------ freemuxlet */

static void plp_default(oracle_plp* p) { /* sc_drop_seq.h:72-75 */
  p->nreads = p->nref = p->nalt = 0;
  p->_pad = 0;
  for (int i = 0; i < 9; ++i) p->gls[i] = 1.0;
}

/* sc_drop_seq.h:77-101 (logdenom bookkeeping dropped: never read) */
void oracle_plp_merge(oracle_plp* d, const oracle_plp* o) {
  d->nreads += o->nreads;
  d->nref += o->nref;
  d->nalt += o->nalt;
  for (int i = 0; i < 9; ++i) d->gls[i] *= o->gls[i];
  double tmp = 0;
  for (int i = 0; i < 9; ++i) tmp += d->gls[i];
  for (int i = 0; i < 9; ++i) d->gls[i] /= tmp;
  for (int i = 0; i < 9; ++i) {
    if (d->gls[i] < MIN_NORM_GL) d->gls[i] = MIN_NORM_GL;
  }
  tmp = 0;
  for (int i = 0; i < 9; ++i) tmp += d->gls[i];
  for (int i = 0; i < 9; ++i) d->gls[i] /= tmp;
}

/* nchains chains of merges, each from a default-constructed pileup (std::map::operator[] at
 * cmd_cram_freemux2.cpp:277-288), elements ptr[i] .. ptr[i+1]-1 in that order; used to pin oracle_plp_merge to the
 * reference's own header-inline merge() (oracle/merge_ref.cpp) */
void oracle_plp_merge_chains(int64_t nchains, const int64_t* ptr, const oracle_plp* elems, oracle_plp* out) {
  for (int64_t i = 0; i < nchains; ++i) {
    oracle_plp d;
    plp_default(&d);
    for (int64_t e = ptr[i]; e < ptr[i + 1]; ++e) oracle_plp_merge(&d, &elems[e]);
    out[i] = d;
  }
}

/* sc_drop_seq.cpp:452-509 with alpha = 0.5 (the only value freemux2 passes, cmd_cram_freemux2.cpp:135) */
void oracle_fmx_entry_pileup(int64_t nnz, const int64_t* entry_rptr, const uint8_t* reads, oracle_plp* out) {
  lut_init();
  const double alpha = 0.5;
  for (int64_t e = 0; e < nnz; ++e) {
    oracle_plp* sdp = &out[e];
    plp_default(sdp);
    double* gls = sdp->gls;
    double tmp;
    for (int64_t r = entry_rptr[e]; r < entry_rptr[e + 1]; ++r) {
      uint8_t b = reads[r];
      ++(sdp->nreads);                     /* :466 */
      if (b == ORACLE_READ_OTHER) continue; /* :468  if ( al > 1 ) continue; */
      uint8_t al = b >> 7;
      uint8_t bq = b & 0x7f;
      if (al == 0) ++(sdp->nref);
      else ++(sdp->nalt);
      double mat = g_mat[bq], err = g_err[bq];
      gls[0] *= (mat * (al == 0 ? 1.0 : 0.0) + err / 4.);                             /* :483 */
      gls[1] *= (mat * (al == 0 ? 1. - alpha / 2. : alpha / 2.) + err / 4.);           /* :484 */
      gls[2] *= (mat * (al == 0 ? 1.0 - alpha : alpha) + err / 4.);                    /* :485 */
      gls[3] *= (mat * (al == 0 ? (1. + alpha) / 2. : (1. - alpha) / 2.) + err / 4.);  /* :486 */
      gls[4] *= (mat * (al == 0 ? .5 : .5) + err / 4.);                                /* :487 */
      gls[5] *= (mat * (al == 0 ? (1. - alpha) / 2. : (1. + alpha) / 2.) + err / 4.);  /* :488 */
      gls[6] *= (mat * (al == 0 ? alpha : 1. - alpha) + err / 4.);                     /* :489 */
      gls[7] *= (mat * (al == 0 ? alpha / 2. : 1. - alpha / 2.) + err / 4.);           /* :490 */
      gls[8] *= (mat * (al == 0 ? 0.0 : 1.0) + err / 4.);                              /* :491 */
      tmp = 0;
      for (int32_t i = 0; i < 9; ++i) tmp += gls[i]; /* :493 */
      for (int32_t i = 0; i < 9; ++i) gls[i] /= tmp; /* :494 */
    }
    for (int32_t i = 0; i < 9; ++i) { /* :498-501 */
      if (gls[i] < MIN_NORM_GL) gls[i] = MIN_NORM_GL;
    }
    tmp = 0;
    for (int32_t i = 0; i < 9; ++i) tmp += gls[i];
    for (int32_t i = 0; i < 9; ++i) gls[i] /= tmp;
  }
}

/* cmd_cram_freemux2.cpp:117-160 */
void oracle_fmx_cell_scores(int64_t C, const int64_t* cell_ptr, const int32_t* entry_snp, const oracle_plp* eplp,
                            const double* afs, double* llk0_out, double* llk2_out, int32_t* nsnps, int32_t* nreads) {
  for (int64_t i = 0; i < C; ++i) {
    double llk0 = 0, llk2 = 0;
    int32_t ns = 0, nr = 0;
    for (int64_t e = cell_ptr[i]; e < cell_ptr[i + 1]; ++e) {
      double af = afs[entry_snp[e]];
      const double* gls = eplp[e].gls;
      double lk0 = 0, lk2 = 0;
      double gps[3];
      gps[0] = (1.0 - af) * (1.0 - af);
      gps[1] = 2.0 * af * (1.0 - af);
      gps[2] = af * af;
      for (int32_t gi = 0; gi < 3; ++gi) {
        lk2 += (gls[gi * 3 + gi] * gps[gi]);
        for (int32_t gj = 0; gj < 3; ++gj) {
          lk0 += (gls[gi * 3 + gj] * gps[gi] * gps[gj]);
        }
      }
      nr += eplp[e].nreads; /* :150  it->second->size(): one UMI per kept base */
      ++ns;
      llk0 += log(lk0);
      llk2 += log(lk2);
    }
    llk0_out[i] = llk0;
    llk2_out[i] = llk2;
    nsnps[i] = ns;
    nreads[i] = nr;
  }
}

/* comparator sc_drop_seq.h:187-198: score descending, ties by id descending */
static const double* g_sort_scores;
static int fmx_cmp(const void* a, const void* b) {
  int32_t lhs = *(const int32_t*)a, rhs = *(const int32_t*)b;
  double cmp = g_sort_scores[lhs] - g_sort_scores[rhs];
  if (cmp != 0) return (cmp > 0) ? -1 : 1;
  return (lhs > rhs) ? -1 : (lhs < rhs ? 1 : 0);
}
void oracle_fmx_sort(int64_t C, const double* scores, int32_t* order) {
  for (int64_t i = 0; i < C; ++i) order[i] = (int32_t)i;
  g_sort_scores = scores;
  qsort(order, (size_t)C, sizeof(int32_t), fmx_cmp); /* total order => same permutation as std::sort */
}

/* one marker of calculate_droplet_clust_distance, sc_drop_seq.cpp:552-568 */
static void clust_dist_term(const double* glis, const double* gljs, double af, double* plk0, double* plk2) {
  double lk0 = 0, lk2 = 0;
  double gps[3];
  gps[0] = (1.0 - af) * (1.0 - af);
  gps[1] = 2.0 * af * (1.0 - af);
  gps[2] = af * af;
  for (int32_t gi = 0; gi < 3; ++gi) {
    lk2 += (glis[gi * 3 + gi] * gljs[gi * 3 + gi] * gps[gi]);
    for (int32_t gj = 0; gj < 3; ++gj) {
      lk0 += (glis[gi * 3 + gi] * gljs[gj * 3 + gj] * gps[gi] * gps[gj]);
    }
  }
  *plk0 = lk0;
  *plk2 = lk2;
}

/* sc_dropseq_lib_t::calculate_droplet_clust_distance, sc_drop_seq.cpp:544-578, on a droplet of n entries in ascending
 * marker order: d[i] its pileup at marker i, c[i] the cluster's state there and present[i] != 0 iff the cluster's
 * std::map holds that marker (jt != end, :550).  out = {llk0, llk2}; counts = {nsnps, nread1, nread2}. */
void oracle_fmx_clust_distance(int64_t n, const oracle_plp* d, const oracle_plp* c, const uint8_t* present,
                               const double* af, double* out, int32_t* counts) {
  double llk0 = 0, llk2 = 0;
  counts[0] = counts[1] = counts[2] = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (!present[i]) continue;
    double lk0, lk2;
    clust_dist_term(d[i].gls, c[i].gls, af[i], &lk0, &lk2);
    ++counts[0];
    counts[1] += d[i].nreads;
    counts[2] += c[i].nreads;
    llk2 += log(lk2);
    llk0 += log(lk0);
  }
  out[0] = llk0;
  out[1] = llk2;
}

/* cmd_cram_freemux2.cpp:217-261; distance = sc_drop_seq.cpp:544-578 */
static void greedy_init_impl(int64_t C, int64_t S, int32_t K, const int64_t* cell_ptr, const int32_t* entry_snp,
                             const oracle_plp* eplp, const double* afs, const double* scores, const int32_t* order,
                             double frac_init_clust, double singlet_score_thres, int32_t* clust, double* step_scores) {
  int64_t step = 0; /* visited cells so far */
  oracle_plp* cp = (oracle_plp*)malloc(sizeof(oracle_plp) * (size_t)K * S);
  uint8_t* present = (uint8_t*)calloc((size_t)K * S, 1); /* key exists in the std::map (jt != end, :550) */
  for (size_t i = 0; i < (size_t)K * S; ++i) plp_default(&cp[i]);
  for (int64_t i = 0; i < C; ++i) clust[i] = -1;
  double* d2 = (double*)malloc(sizeof(double) * K);
  double* d0 = (double*)malloc(sizeof(double) * K);
  for (int64_t i = 0; i < C; ++i) {
    int32_t si = order[i];
    if (i > C * frac_init_clust) continue;          /* :222 */
    if (scores[si] < singlet_score_thres) continue; /* :223 */
    for (int32_t j = 0; j < K; ++j) {               /* :227-230 */
      double llk0 = 0, llk2 = 0;
      for (int64_t e = cell_ptr[si]; e < cell_ptr[si + 1]; ++e) {
        int32_t snp = entry_snp[e];
        size_t ci = (size_t)j * S + snp;
        if (!present[ci]) continue;
        double lk0, lk2;
        clust_dist_term(eplp[e].gls, cp[ci].gls, afs[snp], &lk0, &lk2);
        llk2 += log(lk2);
        llk0 += log(lk0);
      }
      d2[j] = llk2;
      d0[j] = llk0;
    }
    if (step_scores)
      for (int32_t j = 0; j < K; ++j) step_scores[step * K + j] = d2[j] - d0[j];
    ++step;
    int32_t maxClust = 0; /* :233-242 */
    double maxScore = d2[0] - d0[0];
    for (int32_t j = 1; j < K; ++j) {
      if (d2[j] - d0[j] > maxScore) {
        maxClust = j;
        maxScore = d2[j] - d0[j];
      }
    }
    clust[si] = maxClust;
    for (int64_t e = cell_ptr[si]; e < cell_ptr[si + 1]; ++e) { /* :248-251 */
      size_t ci = (size_t)maxClust * S + entry_snp[e];
      oracle_plp_merge(&cp[ci], &eplp[e]);
      present[ci] = 1;
    }
  }
  free(cp);
  free(present);
  free(d2);
  free(d0);
}

void oracle_fmx_greedy_init(int64_t C, int64_t S, int32_t K, const int64_t* cell_ptr, const int32_t* entry_snp,
                            const oracle_plp* eplp, const double* afs, const double* scores, const int32_t* order,
                            double frac_init_clust, double singlet_score_thres, int32_t* clust) {
  greedy_init_impl(C, S, K, cell_ptr, entry_snp, eplp, afs, scores, order, frac_init_clust, singlet_score_thres, clust, NULL);
}

/* the same, also returning dropDs[j].llk2 - dropDs[j].llk0 (:235-240) of every visited cell: step_scores[visit][K] */
void oracle_fmx_greedy_init_scores(int64_t C, int64_t S, int32_t K, const int64_t* cell_ptr, const int32_t* entry_snp,
                                   const oracle_plp* eplp, const double* afs, const double* scores, const int32_t* order,
                                   double frac_init_clust, double singlet_score_thres, int32_t* clust,
                                   double* step_scores) {
  greedy_init_impl(C, S, K, cell_ptr, entry_snp, eplp, afs, scores, order, frac_init_clust, singlet_score_thres, clust,
                   step_scores);
}

/* cmd_cram_freemux2.cpp:277-288 */
void oracle_fmx_build_cluster_pileup(int64_t C, int64_t S, int32_t K, const int64_t* cell_ptr, const int32_t* entry_snp,
                                     const oracle_plp* eplp, const int32_t* clust, oracle_plp* cplp) {
  for (size_t i = 0; i < (size_t)K * S; ++i) plp_default(&cplp[i]);
  for (int64_t i = 0; i < C; ++i) {
    if (clust[i] < 0) continue;
    for (int64_t e = cell_ptr[i]; e < cell_ptr[i + 1]; ++e) {
      oracle_plp_merge(&cplp[(size_t)clust[i] * S + entry_snp[e]], &eplp[e]);
    }
  }
}

void oracle_fmx_init_cells(int64_t C, const int32_t* clust, oracle_fmx_cell* cells) {
  for (int64_t i = 0; i < C; ++i) {
    oracle_fmx_cell* c = &cells[i];
    memset(c, 0, sizeof(*c));
    c->type = (clust[i] >= 0) ? 0 : -1;
    c->clust = clust[i];
    c->jBest = c->kBest = c->jNext = c->kNext = -1; /* :349-352 */
    c->sBest = c->sNext = c->dBest1 = c->dBest2 = c->dNext1 = c->dNext2 = -1;
    c->bestLLK = c->nextLLK = c->sngBestLLK = c->sngNextLLK = c->dblBestLLK = c->dblNextLLK = -1e300;
    c->bestPP = c->sngPP = c->sngOnlyPP = c->sumLLK = -1e300;
  }
}

/* cmd_cram_freemux2.cpp:375-597 */
int32_t oracle_fmx_iterate(int64_t C, int64_t S, int32_t K, const int64_t* cell_ptr, const int32_t* entry_snp,
                           const oracle_plp* eplp, const double* afs, double doublet_prior, double geno_error,
                           oracle_plp* cplp, oracle_fmx_cell* cells, int32_t* nsingle_out, int32_t* namb_out,
                           double* full_ll, int32_t nthreads) {
  const int32_t nSamples = K;
  const int32_t npairs = nSamples * (nSamples + 1) / 2;
  const double log_single_prior = log((1.0 - doublet_prior) / nSamples);                  /* :379 */
  const double log_double_prior = log(doublet_prior / nSamples / (nSamples - 1) * 2.0);   /* :380 */
  if (nthreads < 1) nthreads = 1;

#pragma omp parallel num_threads(nthreads)
  {
    double* llks = (double*)malloc(sizeof(double) * npairs);
    double* lks = (double*)malloc(sizeof(double) * npairs);
    double* gpc = (double*)malloc(sizeof(double) * 3 * nSamples);
#pragma omp for schedule(dynamic, 4)
    for (int64_t i = 0; i < C; ++i) { /* :383 */
      for (int32_t p = 0; p < npairs; ++p) llks[p] = 0;
      for (int64_t e = cell_ptr[i]; e < cell_ptr[i + 1]; ++e) { /* :386 */
        int32_t snp = entry_snp[e];
        double af = afs[snp];
        double gp0s[3];
        gp0s[0] = (1.0 - af) * (1.0 - af);
        gp0s[1] = 2 * af * (1.0 - af);
        gp0s[2] = af * af;
        const double* glis = eplp[e].gls;
        /* gp1s / gp2s of :402-415 and :425-438 are the same expression of (af, cluster pileup): evaluate once per
         * cluster -- bit-identical to re-evaluating it for every pair as the reference does */
        for (int32_t j = 0; j < nSamples; ++j) {
          const oracle_plp* sdp = &cplp[(size_t)j * S + snp];
          double* g = &gpc[3 * j];
          g[0] = (1.0 - af) * (1.0 - af) * sdp->gls[0];
          g[1] = 2 * af * (1.0 - af) * sdp->gls[4];
          g[2] = af * af * sdp->gls[8];
          double sum1 = g[0] + g[1] + g[2];
          g[0] /= sum1;
          g[1] /= sum1;
          g[2] /= sum1;
          if (geno_error > 0) {
            g[0] = (1 - geno_error) * g[0] + geno_error * gp0s[0];
            g[1] = (1 - geno_error) * g[1] + geno_error * gp0s[1];
            g[2] = (1 - geno_error) * g[2] + geno_error * gp0s[2];
          }
        }
        for (int32_t j = 0; j < nSamples; ++j) {
          const double* gp1s = &gpc[3 * j];
          double lk;
          for (int32_t k = 0; k < j; ++k) {
            const double* gp2s = &gpc[3 * k];
            lk = 0;
            for (int32_t g1 = 0; g1 < 3; ++g1) {
              for (int32_t g2 = 0; g2 < 3; ++g2) {
                lk += (glis[g1 * 3 + g2] * gp1s[g1] * gp2s[g2]); /* :443 */
              }
            }
            lks[j * (j + 1) / 2 + k] = lk;
          }
          lk = 0;
          for (int32_t g1 = 0; g1 < 3; ++g1) {
            lk += (glis[g1 * 3 + g1] * gp1s[g1]); /* :450 */
          }
          lks[j * (j + 1) / 2 + j] = lk;
        }
        for (int32_t p = 0; p < npairs; ++p) llks[p] += log(lks[p]); /* :454-455 */
      }
      if (full_ll) memcpy(full_ll + (size_t)i * npairs, llks, sizeof(double) * npairs);

      int32_t sBest = -1, sNext = -1, dBest1 = -1, dBest2 = -1, dNext1 = -1, dNext2 = -1; /* :459 */
      double sngBestLLK = -1e300, sngNextLLK = -1e300, dblBestLLK = -1e300, dblNextLLK = -1e300;
      double sumLLK = -1e300, sngLLK = -1e300;
      double tmpLLK;
      for (int32_t j = 0; j < nSamples; ++j) { /* :469-497 */
        for (int32_t k = 0; k < j; ++k) {
          tmpLLK = llks[j * (j + 1) / 2 + k];
          if (tmpLLK > dblBestLLK) {
            dNext1 = dBest1;
            dNext2 = dBest2;
            dblNextLLK = dblBestLLK;
            dBest1 = j;
            dBest2 = k;
            dblBestLLK = tmpLLK;
          } else if (tmpLLK > dblNextLLK) {
            dNext1 = j;
            dNext2 = k;
            dblNextLLK = tmpLLK;
          }
          sumLLK = oracle_logadd(sumLLK, tmpLLK + log_double_prior);
        }
        tmpLLK = llks[j * (j + 1) / 2 + j];
        if (tmpLLK > sngBestLLK) {
          sNext = sBest;
          sngNextLLK = sngBestLLK;
          sBest = j;
          sngBestLLK = tmpLLK;
        } else if (tmpLLK > sngNextLLK) {
          sNext = j;
          sngNextLLK = tmpLLK;
        }
        sumLLK = oracle_logadd(sumLLK, tmpLLK + log_single_prior);
        sngLLK = oracle_logadd(sngLLK, tmpLLK + log_single_prior);
      }
      oracle_fmx_cell* c = &cells[i]; /* :499-511 */
      c->sBest = sBest;
      c->sngBestLLK = sngBestLLK;
      c->sNext = sNext;
      c->sngNextLLK = sngNextLLK;
      c->dBest1 = dBest1;
      c->dBest2 = dBest2;
      c->dblBestLLK = dblBestLLK;
      c->dNext1 = dNext1;
      c->dNext2 = dNext2;
      c->dblNextLLK = dblNextLLK;
      c->sngPP = exp(sngLLK - sumLLK);
      c->sngOnlyPP = exp(sngBestLLK + log_single_prior - sngLLK);
      c->sumLLK = sumLLK;
    }
    free(llks);
    free(lks);
    free(gpc);
  }

  /* re-assign sample identities + sequential M-step, :515-597 */
  for (size_t i = 0; i < (size_t)K * S; ++i) plp_default(&cplp[i]); /* :516-517 */
  int32_t nsingle = 0, namb = 0, nchanged = 0;
  for (int64_t i = 0; i < C; ++i) {
    oracle_fmx_cell* c = &cells[i];
    c->clust = -1;                               /* :520 */
    if (c->dblBestLLK > c->sngBestLLK + 2) {     /* :521 */
      if (c->type != 1) ++nchanged;
      c->type = 1;
      c->bestPP = (c->dblBestLLK + log_double_prior - c->sumLLK);
      c->jBest = c->dBest1;
      c->kBest = c->dBest2;
      c->bestLLK = c->dblBestLLK;
      if (c->dblNextLLK > c->sngBestLLK + 2) {
        c->jNext = c->dNext1;
        c->kNext = c->dNext2;
        c->nextLLK = c->dblNextLLK;
      } else {
        c->jNext = c->kNext = c->sBest;
        c->nextLLK = c->sngBestLLK;
      }
    } else if (c->sngBestLLK > c->sngNextLLK + 2) { /* :542 */
      if ((c->type != 0) || (c->jBest != c->sBest) || (c->kBest != c->sBest)) ++nchanged;
      c->type = 0;
      ++nsingle;
      c->bestPP = (c->sngBestLLK + log_single_prior - c->sumLLK);
      c->jBest = c->kBest = c->sBest;
      c->bestLLK = c->sngBestLLK;
      c->clust = c->jBest;
      if (c->dblBestLLK > c->sngNextLLK + 2) {
        c->jNext = c->dBest1;
        c->kNext = c->dBest2;
        c->nextLLK = c->dblBestLLK;
      } else {
        c->jNext = c->kNext = c->sNext;
        c->nextLLK = c->sngNextLLK;
      }
    } else { /* :565 */
      if (c->type != 2) ++nchanged;
      c->type = 2;
      ++namb;
      c->bestPP = (c->sngBestLLK + log_single_prior - c->sumLLK);
      c->jBest = c->kBest = c->sBest;
      c->bestLLK = c->sngBestLLK;
      if (c->dblBestLLK > c->sngNextLLK + 2) {
        c->jNext = c->dBest1;
        c->kNext = c->dBest2;
        c->nextLLK = c->dblNextLLK; /* sic, :577 */
      } else {
        c->jNext = c->kNext = c->sNext;
        c->nextLLK = c->sngNextLLK;
      }
    }
    if ((c->jBest == c->kBest) && (c->type == 0)) { /* :590-596 */
      for (int64_t e = cell_ptr[i]; e < cell_ptr[i + 1]; ++e) {
        oracle_plp_merge(&cplp[(size_t)c->jBest * S + entry_snp[e]], &eplp[e]);
      }
    }
  }
  *nsingle_out = nsingle;
  *namb_out = namb;
  return nchanged;
}

/* ---------------------------------------------------------------------------------------- freemuxlet-old deltas */

/* cmd_cram_freemuxlet.cpp:187-221 */
void oracle_fmxold_pair_dist(int64_t C, int64_t S, const int64_t* cell_ptr, const int32_t* entry_snp,
                             const oracle_plp* eplp, const double* afs, oracle_dropd* out) {
  memset(out, 0, sizeof(oracle_dropd) * (size_t)(C * (C - 1) / 2));
  /* snp_cell_plps[v] : std::map<cell, plp*> (:113,135) == the entries of SNP v in ascending cell id */
  int64_t nnz = cell_ptr[C];
  int64_t* sp = (int64_t*)calloc((size_t)S + 1, sizeof(int64_t));
  for (int64_t e = 0; e < nnz; ++e) ++sp[entry_snp[e] + 1];
  for (int64_t v = 0; v < S; ++v) sp[v + 1] += sp[v];
  int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)S);
  memcpy(fill, sp, sizeof(int64_t) * (size_t)S);
  int64_t* sent = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nnz ? nnz : 1));
  int32_t* scell = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
  for (int64_t i = 0; i < C; ++i)
    for (int64_t e = cell_ptr[i]; e < cell_ptr[i + 1]; ++e) {
      int64_t p = fill[entry_snp[e]]++;
      sent[p] = e;
      scell[p] = (int32_t)i;
    }
  for (int64_t v = 0; v < S; ++v) {
    for (int64_t p = sp[v]; p < sp[v + 1]; ++p) {
      const oracle_plp* pi = &eplp[sent[p]];
      const double* glis = pi->gls;
      for (int64_t q = sp[v]; q < p; ++q) {
        const oracle_plp* pj = &eplp[sent[q]];
        const double* gljs = pj->gls;
        double af = afs[v];
        double lk0 = 0, lk2 = 0;
        double gps[3];
        gps[0] = (1.0 - af) * (1.0 - af);
        gps[1] = 2.0 * af * (1.0 - af);
        gps[2] = af * af;
        for (int32_t gi = 0; gi < 3; ++gi) { /* :203-208 */
          lk2 += (glis[gi * 3 + gi] * gljs[gi * 3 + gi] * gps[gi]);
          for (int32_t gj = 0; gj < 3; ++gj) {
            lk0 += (glis[gi * 3 + gi] * gljs[gj * 3 + gj] * gps[gi] * gps[gj]);
          }
        }
        int64_t a = scell[p], b = scell[q];
        oracle_dropd* d = &out[a * (a - 1) / 2 + b]; /* dropDs[it->first][jt->first] (:210) */
        ++d->nsnps;
        d->nread1 += pi->nreads;
        d->nread2 += pj->nreads;
        d->llk2 += log(lk2);
        d->llk0 += log(lk0);
      }
    }
  }
  free(sp);
  free(fill);
  free(sent);
  free(scell);
}

static const oracle_dropd* dd_at(const oracle_dropd* dd, int64_t a, int64_t b) {
  return (a > b) ? &dd[a * (a - 1) / 2 + b] : &dd[b * (b - 1) / 2 + a];
}

/* cmd_cram_freemuxlet.cpp:245-291 */
void oracle_fmxold_vote_init(int64_t C, int32_t K, const oracle_dropd* dd, const int32_t* order, const double* jitter,
                             double bfThres, double fracInitClust, int32_t* clusts, int32_t* ccounts) {
  double* votes = (double*)malloc(sizeof(double) * (size_t)K);
  for (int64_t i = 0; i < C; ++i) clusts[i] = -1;
  for (int32_t j = 0; j < K; ++j) ccounts[j] = 0;
  int64_t t = 0;
  for (int64_t i = 0; i < C; ++i) {
    int32_t si = order[i];
    if (i > C * fracInitClust) continue; /* :248 */
    for (int32_t j = 0; j < K; ++j) votes[j] = jitter[t * K + j]; /* :257-259 */
    ++t;
    for (int64_t j = 0; j < i; ++j) {
      int32_t sj = order[j];
      const oracle_dropd* d = dd_at(dd, si, sj);
      if (d->llk0 - d->llk2 > bfThres) votes[clusts[sj]] -= 1.0;      /* :273-275 */
      else if (d->llk2 - d->llk0 > bfThres) votes[clusts[sj]] += 1.0; /* :276-278 */
    }
    int32_t elected = 0;
    double maxvote = votes[0];
    for (int32_t j = 1; j < K; ++j) {
      if (maxvote < votes[j]) {
        elected = j;
        maxvote = votes[j];
      }
    }
    clusts[si] = elected;
    ++ccounts[elected];
  }
  free(votes);
}

/* cmd_cram_freemuxlet.cpp:297-343, one value of iter */
int32_t oracle_fmxold_vote_refine(int64_t C, int32_t K, const oracle_dropd* dd, const int32_t* order,
                                  const double* jitter, double bfThres, int32_t keepInitMissing, int32_t* clusts,
                                  int32_t* ccounts) {
  double* votes = (double*)malloc(sizeof(double) * (size_t)K);
  int32_t changed = 0;
  for (int32_t j = 0; j < K; ++j) ccounts[j] = 0;
  for (int64_t i = 0; i < C; ++i) {
    int32_t si = order[i];
    for (int32_t j = 0; j < K; ++j) votes[j] = jitter[i * K + j];
    for (int64_t j = 0; j < C; ++j) {
      if (si != j) {
        const oracle_dropd* d = dd_at(dd, si, j);
        double bf = d->llk2 - d->llk0;
        if (clusts[j] >= 0) {
          if (bf > bfThres) ++votes[clusts[j]];
          else if (bf < 0 - bfThres) --votes[clusts[j]];
        }
      }
    }
    int32_t elected = 0;
    double maxvote = votes[0];
    for (int32_t j = 1; j < K; ++j) {
      if (maxvote < votes[j]) {
        elected = j;
        maxvote = votes[j];
      }
    }
    if ((clusts[si] >= 0) || (keepInitMissing == 0)) { /* :333-337 */
      if (clusts[si] != elected) ++changed;
      clusts[si] = elected;
      ++ccounts[elected];
    }
  }
  free(votes);
  return changed;
}
