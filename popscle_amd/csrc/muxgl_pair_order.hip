// C-ABI export of the host-side pass that orders a mirrored alpha = 0.5 doublet pair as the reference's scan does
// (popscle_amd/host/pair_order.hpp; cmd_cram_demuxlet.cpp:738-746,883-906).  Host code only: no kernel, no handle.
#include "../host/pair_order.hpp"

extern "C" int muxgl_demux_reference_pair_order(int64_t C, int32_t V, const int64_t* cell_ptr, const int32_t* entry_snp,
                                                const int64_t* entry_rptr, const uint8_t* reads, const double* gp,
                                                const uint8_t* has_gp, const muxgl_demux_params* p,
                                                muxgl_demux_cell* cells, int32_t nthreads, int64_t* stats) {
  if (C < 0 || V < 1 || !cell_ptr || !entry_snp || !entry_rptr || !reads || !gp || !has_gp || !p || !cells) return 1;
  if (p->n_alpha < 1 || p->n_alpha > MUXGL_MAX_ALPHA) return 1;
  try {
    pair_order::reference_pair_order(C, V, cell_ptr, entry_snp, entry_rptr, reads, gp, has_gp, p->n_alpha, p->alpha,
                                     cells, nthreads, stats);
  } catch (...) {
    return 2;
  }
  return 0;
}
