// Test infrastructure (never linked into the product): extern "C" wrappers around two header-inline pieces of the
// reference, compiled from the reference's own, unmodified sc_drop_seq.h where it lies under /root/reference:
//
//   snp_droplet_pileup::merge   sc_drop_seq.h:77-101   (the order-dependent clamped merge of freemuxlet's M-step)
//   sc_drop_comp_t              sc_drop_seq.h:187-198  (the comparator behind std::sort, cmd_cram_freemux2.cpp:184-189)
//
// The header's only htslib-dependent include is "bcf_filtered_reader.h"; the build predefines that header's include
// guard (-D__BCF_FILTERED_READER_H, oracle/Makefile), so it is skipped and nothing stands in for it: the one name
// the header needs from it is a pointer type, forward-declared below.  No reference source is copied: this file holds
// no line of the reference, only calls into it.
#include <algorithm>
#include <cstring>
#include <set>
#include <vector>

class BCFFilteredReader;
#include "sc_drop_seq.h"

extern "C" {

// layout handed over by the tests: {int32 nreads, nref, nalt, (pad), double gls[9]} == oracle_plp (muxgl_oracle.h)
struct merge_ref_plp {
  int32_t nreads, nref, nalt, pad;
  double gls[9];
};

static void to_ref(const merge_ref_plp& a, snp_droplet_pileup& b) {
  b.nreads = a.nreads;
  b.nref = a.nref;
  b.nalt = a.nalt;
  std::memcpy(b.gls, a.gls, sizeof(b.gls));
}

static void from_ref(const snp_droplet_pileup& b, merge_ref_plp& a) {
  a.nreads = b.nreads;
  a.nref = b.nref;
  a.nalt = b.nalt;
  std::memcpy(a.gls, b.gls, sizeof(b.gls));
}

int merge_ref_sizeof_plp(void) { return (int)sizeof(merge_ref_plp); }

// dst.merge(src), once
void merge_ref_merge(merge_ref_plp* dst, const merge_ref_plp* src) {
  snp_droplet_pileup d, s;
  to_ref(*dst, d);
  to_ref(*src, s);
  d.merge(s);
  from_ref(d, *dst);
}

// nchains chains: chain i starts from a default-constructed pileup (as cmd_cram_freemux2.cpp:277-288 does through
// std::map::operator[]) and merges elems[ptr[i]] .. elems[ptr[i+1]-1] in that order; out[i] = the final state
void merge_ref_chains(int64_t nchains, const int64_t* ptr, const merge_ref_plp* elems, merge_ref_plp* out) {
  for (int64_t i = 0; i < nchains; ++i) {
    snp_droplet_pileup d, s;
    for (int64_t e = ptr[i]; e < ptr[i + 1]; ++e) {
      to_ref(elems[e], s);
      d.merge(s);
    }
    from_ref(d, out[i]);
  }
}

// order = 0..C-1 sorted with std::sort under sc_drop_comp_t over scl.cell_scores, as cmd_cram_freemux2.cpp:184-189
void merge_ref_sort(int64_t C, const double* scores, int32_t* order) {
  sc_dropseq_lib_t scl;
  scl.cell_scores.assign(scores, scores + C);
  std::vector<int32_t> v((size_t)C);
  for (int64_t i = 0; i < C; ++i) v[(size_t)i] = (int32_t)i;
  sc_drop_comp_t cmp(&scl);
  std::sort(v.begin(), v.end(), cmp);
  for (int64_t i = 0; i < C; ++i) order[i] = v[(size_t)i];
}

// the comparator itself on one pair
int merge_ref_comp(int64_t C, const double* scores, int32_t lhs, int32_t rhs) {
  sc_dropseq_lib_t scl;
  scl.cell_scores.assign(scores, scores + C);
  sc_drop_comp_t cmp(&scl);
  return cmp(lhs, rhs) ? 1 : 0;
}

}  // extern "C"
