// C-ABI export of the host-side pass that settles the demuxlet calls rounding noise could decide -- the order of a
// mirrored alpha = 0.5 pair and every near tie of the scans and thresholds -- in the reference's own arithmetic
// (popscle_amd/host/exact_calls.hpp; cmd_cram_demuxlet.cpp:738-746,827-837,883-906,925-988).  Host code only: no kernel,
// no handle.
#include "../host/exact_calls.hpp"

extern "C" int muxgl_demux_exact_calls(int64_t C, int32_t V, const int64_t* cell_ptr, const int32_t* entry_snp,
                                       const int64_t* entry_rptr, const uint8_t* reads, const double* gp,
                                       const uint8_t* has_gp, const muxgl_demux_params* p, muxgl_demux_cell* cells,
                                       int32_t nthreads, int64_t* stats) {
  if (C < 0 || V < 1 || !cell_ptr || !entry_snp || !entry_rptr || !reads || !gp || !has_gp || !p || !cells) return 1;
  if (p->n_alpha < 1 || p->n_alpha > MUXGL_MAX_ALPHA) return 1;
  try {
    exact_calls::exact_calls(C, V, cell_ptr, entry_snp, entry_rptr, reads, gp, has_gp, p->n_alpha, p->alpha,
                             p->doublet_prior, cells, nthreads, stats);
  } catch (...) {
    return 2;
  }
  return 0;
}
