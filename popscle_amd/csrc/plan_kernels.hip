// plan_kernels.hip -- derived, device-resident views of the packed pileup, built on the device at hand-over time:
//   * the packed 16-byte entry records of the quad kernel (common.hpp: quad_entry);
//   * the chunk tables of the row / quad kernels (common.hpp: muxgl_row_state): a cell is cut into the fewest chunks of
//     <= CH entries, of equal length rounded up to four; launch order = ascending first SNP id, ties in (cell, entry)
//     order (stable radix sort of the chunks' first SNP ids); per cell the positions of its chunks in entry order;
//   * the SNP-major (CSC) view of the entries -- what the reference holds as snp_cell_plps / walks per SNP in the
//     ordered merge (cmd_cram_freemux2.cpp:277-288,590-596): a STABLE sort of the cell-major entries by SNP id, so
//     that the cells of a SNP come out in ascending id (rocPRIM LSD radix sort over the log2(S) key bits), row
//     pointers by binary search in the sorted keys.
// Both used to be host loops over all entries (0.4 s and 0.6 s per 48 M entries); on the device they take milliseconds.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.hpp"

namespace {

__global__ void __launch_bounds__(256)
    qent_kernel(int64_t nnz, const int32_t* __restrict__ entry_snp, const int64_t* __restrict__ entry_rptr,
                const uint8_t* __restrict__ reads, quad_entry* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r0 = entry_rptr[e], n = entry_rptr[e + 1] - r0;
    uint32_t f4 = 0;
    for (int64_t k = 0; k < n && k < 4; ++k) f4 |= (uint32_t)reads[r0 + k] << (8 * k);
    quad_entry q;
    q.snp = entry_snp[e];
    q.nreads = (uint32_t)(n > 0xffffffffLL ? 0xffffffffLL : n);
    q.first4 = f4;
    q.r0 = (uint32_t)r0;
    out[e] = q;
  }
}

// bit e of lin: entry e has at most one usable read (alleles other than 0/1 are skipped, cmd_cram_demuxlet.cpp:664), and
// that read's base quality is at most 60: the moment forms of the sweeps reach the small end of the linear likelihood
// through cancellation, at a relative error of ~1e-16 * max / min = 3e-16 * 10^(Q/10) -- 3e-10 at Q60; anything above
// (no sequencer in use writes it) stays on the nine-term path.
// 64 consecutive entries per wave, the two words of their ballot written by lanes 0 and 32.
__global__ void __launch_bounds__(256)
    lin_kernel(int64_t nnz, const int64_t* __restrict__ entry_rptr, const uint8_t* __restrict__ reads,
               uint32_t* __restrict__ lin) {
  const int lane = threadIdx.x & 63;
  for (int64_t eb = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; eb < nnz; eb += (int64_t)gridDim.x * 256) {
    const int64_t e = eb + lane;
    bool one = false;
    if (e < nnz) {
      int usable = 0;
      bool lowq = true;
      for (int64_t r = entry_rptr[e]; r < entry_rptr[e + 1] && usable < 2; ++r) {
        const uint8_t bb = reads[r];
        if (bb == MUXGL_READ_OTHER) continue;
        ++usable;
        lowq = lowq && (bb & 0x7f) <= 60;
      }
      one = usable <= 1 && lowq;
    }
    const uint64_t m = __ballot(one);
    if (lane == 0) lin[eb >> 5] = (uint32_t)m;
    if (lane == 32 && eb + 32 < nnz) lin[(eb >> 5) + 1] = (uint32_t)(m >> 32);
  }
}

// one wave per cell: entry_cell[e] = c, and the identity permutation that the sort carries along
__global__ void __launch_bounds__(256)
    entry_cell_kernel(int64_t C, const int64_t* __restrict__ cell_ptr, int32_t* __restrict__ entry_cell,
                      int64_t* __restrict__ iota) {
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  for (int64_t e = cell_ptr[c] + (threadIdx.x & 63); e < cell_ptr[c + 1]; e += 64) {
    entry_cell[e] = (int32_t)c;
    iota[e] = e;
  }
}

__global__ void __launch_bounds__(256)
    snp_ptr_kernel(int64_t S, int64_t nnz, const int32_t* __restrict__ keys, int64_t* __restrict__ snp_ptr) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v > S) return;
  int64_t lo = 0, hi = nnz;  // first position with key >= v
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < (int32_t)v) lo = mid + 1;
    else hi = mid;
  }
  snp_ptr[v] = lo;
}

__global__ void __launch_bounds__(256)
    snp_cell_kernel(int64_t nnz, const int64_t* __restrict__ snp_entry, const int32_t* __restrict__ entry_cell,
                    int32_t* __restrict__ snp_cell) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x)
    snp_cell[p] = entry_cell[snp_entry[p]];
}

// chunk geometry of a cell with n entries: nch chunks of `step` entries (the last one shorter)
__device__ __forceinline__ void chunk_geom(int64_t n, int ch, int64_t& nch, int64_t& step) {
  nch = (n + ch - 1) / ch;
  step = nch ? ((n + nch - 1) / nch + 3) / 4 * 4 : 0;
  if (step > ch) step = ch;
}

__global__ void __launch_bounds__(256)
    chunk_count_kernel(int64_t C, int64_t cb, int64_t ce, int ch, const int64_t* __restrict__ cell_ptr,
                       int64_t* __restrict__ cnt) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > C) return;
  int64_t nch = 0, step = 0;
  if (c >= cb && c < ce) chunk_geom(cell_ptr[c + 1] - cell_ptr[c], ch, nch, step);
  cnt[c] = nch;  // cnt[C] = 0: the exclusive scan then ends with the total
}

__global__ void __launch_bounds__(256)
    chunk_fill_kernel(int64_t cb, int64_t ce, int ch, const int64_t* __restrict__ cell_ptr,
                      const int32_t* __restrict__ entry_snp, const int64_t* __restrict__ chunk_ptr,
                      row_chunk* __restrict__ nat, int32_t* __restrict__ key, int32_t* __restrict__ iota) {
  const int64_t c = cb + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ce) return;
  const int64_t e0 = cell_ptr[c], e1 = cell_ptr[c + 1];
  int64_t nch, step;
  chunk_geom(e1 - e0, ch, nch, step);
  int64_t o = chunk_ptr[c];
  for (int64_t e = e0; e < e1; e += step, ++o) {
    row_chunk r;
    r.e0 = e;
    r.len = (int32_t)(e1 - e < step ? e1 - e : step);
    r.cell = (int32_t)c;
    nat[o] = r;
    key[o] = entry_snp[e];
    iota[o] = (int32_t)o;
  }
}

__global__ void __launch_bounds__(256)
    chunk_place_kernel(int64_t n, const int32_t* __restrict__ ord, const row_chunk* __restrict__ nat,
                       row_chunk* __restrict__ chunks, int32_t* __restrict__ cell_chunks) {
  const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n) return;
  const int32_t id = ord[pos];
  chunks[pos] = nat[id];
  cell_chunks[id] = (int32_t)pos;  // natural ids are (cell, entry) ordered: a cell's chunks in entry order
}

__global__ void __launch_bounds__(256)
    bits_popc_kernel(int64_t nwords, int64_t nnz, const uint32_t* __restrict__ bits, int64_t* __restrict__ cnt) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w > nwords) return;
  uint32_t v = 0;
  if (w < nwords) {
    v = bits[w];
    const int64_t left = nnz - w * 32;
    if (left < 32) v &= (1u << left) - 1u;
  }
  cnt[w] = __popc(v);
}

__global__ void __launch_bounds__(256)
    bits_stream_kernel(int64_t nnz, const uint32_t* __restrict__ bits, const int64_t* __restrict__ rank,
                       const int32_t* __restrict__ entry_snp, fmx_grec* __restrict__ rec_set, fmx_grec* __restrict__ rec_clr) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnz) return;
  const uint32_t w = bits[e >> 5];
  const int64_t r = rank[e >> 5] + __popc(w & ((1u << (e & 31)) - 1u));
  if ((w >> (e & 31)) & 1u) rec_set[r] = fmx_grec{e, entry_snp[e], 0};
  else rec_clr[e - r] = fmx_grec{e, entry_snp[e], 0};
}

unsigned grid_for(int64_t n, int64_t cap = 16384) {
  int64_t b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

int plan_build_qent(muxgl_handle* h) {
  dev_free(&h->d_qent);
  if (h->R >= ((int64_t)1 << 32) || h->nnz == 0) return 0;  // read offsets must fit the record's 32 bits
  if (dev_alloc(h, &h->d_qent, (size_t)h->nnz)) return 1;
  hipLaunchKernelGGL(qent_kernel, dim3(grid_for(h->nnz)), dim3(256), 0, h->stream, h->nnz, h->d_entry_snp,
                     h->d_entry_rptr, h->d_reads, h->d_qent);
  HIPCHK(h, hipGetLastError());
  return 0;
}

// The entries of a bit set and of its complement as two streams of {entry, snp} records in entry order, and the table
// rank[w] = set bits before entry 32 w that places any entry in both (wave kernels: a cell's entries of one kind are
// then contiguous records instead of a scan of the bits).
int plan_build_bit_streams(muxgl_handle* h, const uint32_t* bits, int64_t** rank, fmx_grec** rec_set, fmx_grec** rec_clr,
                           int64_t* n_set) {
  const int64_t nnz = h->nnz, nwords = (nnz + 31) / 32;
  if (dev_alloc(h, rank, (size_t)nwords + 1)) return 1;
  hipLaunchKernelGGL(bits_popc_kernel, dim3((unsigned)((nwords + 256) / 256)), dim3(256), 0, h->stream, nwords, nnz, bits, *rank);
  size_t tb = 0;
  void* tmp = nullptr;
  HIPCHK(h, rocprim::exclusive_scan(nullptr, tb, *rank, *rank, (int64_t)0, (size_t)nwords + 1, rocprim::plus<int64_t>(), h->stream));
  HIPCHK(h, dev_malloc_retry((void**)&tmp, tb ? tb : 1));
  hipError_t e = rocprim::exclusive_scan(tmp, tb, *rank, *rank, (int64_t)0, (size_t)nwords + 1, rocprim::plus<int64_t>(), h->stream);
  int64_t n = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&n, *rank + nwords, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(tmp);
  if (e != hipSuccess) MUXGL_FAIL(h, "plan_build_bit_streams: %s", hipGetErrorString(e));
  if (dev_alloc(h, rec_set, (size_t)n) || dev_alloc(h, rec_clr, (size_t)(nnz - n))) return 1;
  if (nnz)
    hipLaunchKernelGGL(bits_stream_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, h->stream, nnz, bits, *rank,
                       h->d_entry_snp, *rec_set, *rec_clr);
  HIPCHK(h, hipGetLastError());
  *n_set = n;
  return 0;
}

void plan_lin_streams_release(muxgl_handle* h) {
  dev_free(&h->d_lin_rank);
  dev_free(&h->d_lin_rec);
  dev_free(&h->d_gen_rec);
  h->n_lin_rec = -1;
  demux_ring_release(h);
}

int plan_build_lin(muxgl_handle* h) {
  dev_free(&h->d_lin);
  dev_free(&h->d_flin);
  fmx_wave_streams_release(h);
  plan_lin_streams_release(h);
  if (h->qrow) {  // the quad kernel's records with the linear entries first: made from the bits, on first use
    dev_free(&h->qrow->d_qent_lin);
    dev_free(&h->qrow->d_chunk_nlin);
    dev_free(&h->qrow->d_orec);
    dev_free(&h->qrow->d_unit_ptr);
    dev_free(&h->qrow->d_quad_order);
  }
  if (h->nnz == 0) return 0;
  if (dev_alloc(h, &h->d_lin, (size_t)((h->nnz + 31) / 32))) return 1;
  hipLaunchKernelGGL(lin_kernel, dim3(grid_for(h->nnz, 4096)), dim3(256), 0, h->stream, h->nnz, h->d_entry_rptr,
                     h->d_reads, h->d_lin);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int plan_build_snp_major(muxgl_handle* h) {
  const int64_t C = h->C, S = h->S, nnz = h->nnz;
  if (dev_alloc(h, &h->d_snp_ptr, (size_t)S + 1)) return 1;
  if (dev_alloc(h, &h->d_snp_entry, (size_t)nnz)) return 1;
  if (dev_alloc(h, &h->d_entry_cell, (size_t)nnz)) return 1;
  if (dev_alloc(h, &h->d_snp_cell, (size_t)nnz)) return 1;
  if (nnz == 0) {
    HIPCHK(h, hipMemsetAsync(h->d_snp_ptr, 0, sizeof(int64_t) * (S + 1), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
  }
  int32_t* d_keys = nullptr;
  int64_t* d_iota = nullptr;
  void* d_tmp = nullptr;
  auto cleanup = [&]() {
    dev_free(&d_keys);
    dev_free(&d_iota);
    if (d_tmp) (void)hipFree(d_tmp);
    d_tmp = nullptr;
  };
  if (dev_alloc(h, &d_keys, (size_t)nnz) || dev_alloc(h, &d_iota, (size_t)nnz)) {
    cleanup();
    return 1;
  }
  hipLaunchKernelGGL(entry_cell_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, h->stream, C, h->d_cell_ptr,
                     h->d_entry_cell, d_iota);
  unsigned bits = 1;
  while (bits < 31 && ((int64_t)1 << bits) < S) ++bits;
  size_t tmp_bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, h->d_entry_snp, d_keys, d_iota, h->d_snp_entry,
                                           (size_t)nnz, 0u, bits, h->stream);
  if (e == hipSuccess) e = dev_malloc_retry((void**)&d_tmp, tmp_bytes ? tmp_bytes : 1);
  if (e == hipSuccess)
    e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, h->d_entry_snp, d_keys, d_iota, h->d_snp_entry, (size_t)nnz, 0u,
                                  bits, h->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(snp_ptr_kernel, dim3((unsigned)((S + 1 + 255) / 256)), dim3(256), 0, h->stream, S, nnz, d_keys,
                       h->d_snp_ptr);
    hipLaunchKernelGGL(snp_cell_kernel, dim3(grid_for(nnz)), dim3(256), 0, h->stream, nnz, h->d_snp_entry,
                       h->d_entry_cell, h->d_snp_cell);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  cleanup();
  if (e != hipSuccess) MUXGL_FAIL(h, "SNP-major view: %s", hipGetErrorString(e));
  return 0;
}

// chunk tables of the cells [cb, ce) for chunks of <= ch entries; see the header comment.  n_chunks comes back to the
// host (one 8-byte copy); everything else stays on the device.
int plan_build_chunks(muxgl_handle* h, muxgl_row_state* st, int64_t cb, int64_t ce, int ch) {
  const int64_t C = h->C;
  if (dev_alloc(h, &st->d_cell_chunk_ptr, (size_t)C + 1)) return 1;
  int64_t* d_cnt = nullptr;
  row_chunk* d_nat = nullptr;
  int32_t *d_key = nullptr, *d_key2 = nullptr, *d_iota = nullptr, *d_ord = nullptr;
  void* d_tmp = nullptr;
  auto cleanup = [&]() {
    dev_free(&d_cnt);
    dev_free(&d_nat);
    dev_free(&d_key);
    dev_free(&d_key2);
    dev_free(&d_iota);
    dev_free(&d_ord);
    if (d_tmp) (void)hipFree(d_tmp);
    d_tmp = nullptr;
  };
  auto fail = [&](hipError_t e) {
    cleanup();
    MUXGL_FAIL(h, "chunk tables: %s", hipGetErrorString(e));
  };
  if (dev_alloc(h, &d_cnt, (size_t)C + 1)) return 1;
  hipLaunchKernelGGL(chunk_count_kernel, dim3((unsigned)((C + 1 + 255) / 256)), dim3(256), 0, h->stream, C, cb, ce, ch,
                     h->d_cell_ptr, d_cnt);
  size_t tmp_bytes = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, tmp_bytes, d_cnt, st->d_cell_chunk_ptr, (int64_t)0, (size_t)C + 1,
                                         rocprim::plus<int64_t>(), h->stream);
  if (e == hipSuccess) e = dev_malloc_retry((void**)&d_tmp, tmp_bytes ? tmp_bytes : 1);
  if (e == hipSuccess)
    e = rocprim::exclusive_scan(d_tmp, tmp_bytes, d_cnt, st->d_cell_chunk_ptr, (int64_t)0, (size_t)C + 1,
                                rocprim::plus<int64_t>(), h->stream);
  int64_t n = 0;
  if (e == hipSuccess)
    e = hipMemcpyAsync(&n, st->d_cell_chunk_ptr + C, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return fail(e);
  (void)hipFree(d_tmp);
  d_tmp = nullptr;
  if (n > INT32_MAX) {
    cleanup();
    MUXGL_FAIL(h, "chunk tables: %lld chunks exceed int32", (long long)n);
  }
  st->n_chunks = n;
  if (dev_alloc(h, &st->d_chunks, (size_t)n) || dev_alloc(h, &st->d_cell_chunks, (size_t)n)) {
    cleanup();
    return 1;
  }
  if (n > 0) {
    if (dev_alloc(h, &d_nat, (size_t)n) || dev_alloc(h, &d_key, (size_t)n) || dev_alloc(h, &d_key2, (size_t)n) ||
        dev_alloc(h, &d_iota, (size_t)n) || dev_alloc(h, &d_ord, (size_t)n)) {
      cleanup();
      return 1;
    }
    hipLaunchKernelGGL(chunk_fill_kernel, dim3((unsigned)((ce - cb + 255) / 256)), dim3(256), 0, h->stream, cb, ce, ch,
                       h->d_cell_ptr, h->d_entry_snp, st->d_cell_chunk_ptr, d_nat, d_key, d_iota);
    unsigned bits = 1;
    while (bits < 31 && ((int64_t)1 << bits) < h->S) ++bits;
    tmp_bytes = 0;
    e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_key, d_key2, d_iota, d_ord, (size_t)n, 0u, bits, h->stream);
    if (e == hipSuccess) e = dev_malloc_retry((void**)&d_tmp, tmp_bytes ? tmp_bytes : 1);
    if (e == hipSuccess)
      e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_key, d_key2, d_iota, d_ord, (size_t)n, 0u, bits, h->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(chunk_place_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, n, d_ord, d_nat,
                         st->d_chunks, st->d_cell_chunks);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(e);
  }
  cleanup();
  return 0;
}
