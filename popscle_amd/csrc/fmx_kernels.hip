// fmx_kernels.hip -- freemuxlet hot path on gfx950 (cmd_cram_freemux2.cpp:117-160,277-288,375-597).
#include "common.hpp"

extern "C" {

int muxgl_fmx_prepare(muxgl_handle* h, const double*, double*, double*, int32_t*, int32_t*) {
  if (!h) return 1;
  MUXGL_FAIL(h, "muxgl_fmx_prepare: not implemented yet");
}
int muxgl_fmx_get_entry_gls(muxgl_handle* h, double*, int32_t*) {
  if (!h) return 1;
  MUXGL_FAIL(h, "muxgl_fmx_get_entry_gls: not implemented yet");
}
int muxgl_fmx_set_clusters(muxgl_handle* h, int32_t, const int32_t*) {
  if (!h) return 1;
  MUXGL_FAIL(h, "muxgl_fmx_set_clusters: not implemented yet");
}
int muxgl_fmx_iterate(muxgl_handle* h, const muxgl_fmx_params*, muxgl_fmx_cell*, int32_t*, int32_t*, int32_t*, double*) {
  if (!h) return 1;
  MUXGL_FAIL(h, "muxgl_fmx_iterate: not implemented yet");
}
int muxgl_fmx_get_cluster_pileup(muxgl_handle* h, double*, int32_t*) {
  if (!h) return 1;
  MUXGL_FAIL(h, "muxgl_fmx_get_cluster_pileup: not implemented yet");
}

}  // extern "C"
