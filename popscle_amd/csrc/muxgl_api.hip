// muxgl_api.hip -- the C-ABI of include/muxgl.h: handle lifetime, hand-over of the packed pileup and GP tensor to
// device memory, and the synchronous run/iterate entry points.  No CPU fallback: every compute call needs a HIP device.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "common.hpp"

thread_local std::string g_muxgl_create_error;

// ---- device memory cache (common.hpp) -----------------------------------------------------------------------------------
int dev_alloc_bytes(muxgl_handle* h, void** p, size_t bytes) {
  *p = nullptr;
  dev_registry& R = dev_reg();
  const bool pooled = dev_pool_on() && h && bytes >= DEV_POOL_MIN;
  if (pooled) {
    void* hit = nullptr;
    {
      std::lock_guard<std::mutex> g(R.mu);
      auto it = h->pool.lower_bound(bytes);
      if (it != h->pool.end() && it->first <= bytes + bytes / 4 + ((size_t)16 << 20)) {
        hit = it->second;
        h->pool_bytes -= it->first;
        h->pool.erase(it);
      }
    }
    if (hit) {
      // whatever still reads or writes the block was enqueued on this handle's stream (or has completed) -- a handle's
      // blocks are only ever touched by work on its own stream; exchanges between the members of a device group are
      // ordered against it by events before the phase returns (muxgl_group.hip) -- so draining that stream is what the
      // hipFree this replaces did for the block
      const hipError_t e = h->stream ? hipStreamSynchronize(h->stream) : hipSuccess;
      if (e != hipSuccess) {  // the block is still registered to this handle: give it back to the driver, do not lose it
        {
          std::lock_guard<std::mutex> g(R.mu);
          R.blocks.erase(hit);
        }
        (void)hipFree(hit);
        MUXGL_FAIL(h, "hipStreamSynchronize before reusing a cached block: %s", hipGetErrorString(e));
      }
      *p = hit;
      return 0;
    }
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {  // give the cached blocks back to the driver and try once more
    (void)hipGetLastError();
    dev_pool_release_all();
    e = hipMalloc(p, bytes);
  }
  if (e != hipSuccess) {
    *p = nullptr;
    if (h) MUXGL_FAIL(h, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return 1;
  }
  if (pooled) {
    std::lock_guard<std::mutex> g(R.mu);
    R.blocks[*p] = dev_block_info{bytes, h};
    R.handles.insert(h);
  }
  return 0;
}

hipError_t dev_malloc_retry(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    dev_pool_release_all();
    e = hipMalloc(p, bytes);
  }
  return e;
}

// what a handle's cache may hold: a third of the device's memory (a phase that parks more than that gives the rest back to
// the driver at once, so that allocations outside the cache keep finding room)
static size_t dev_pool_cap() {
  static const size_t cap = [] {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || tot == 0) return (size_t)64 << 30;
    return tot / 3;
  }();
  return cap;
}

void dev_free_bytes(void* p) {
  if (!p) return;
  dev_registry& R = dev_reg();
  {
    std::lock_guard<std::mutex> g(R.mu);
    auto it = R.blocks.find(p);
    if (it != R.blocks.end()) {
      muxgl_handle* o = it->second.owner;
      if (o && dev_pool_on() && o->pool_bytes + it->second.bytes <= dev_pool_cap()) {
        o->pool.emplace(it->second.bytes, p);
        o->pool_bytes += it->second.bytes;
        return;
      }
      R.blocks.erase(it);
    }
  }
  (void)hipFree(p);
}

// frees the cached blocks of h; forget_owner: h is going away -- its blocks that are still in use lose their owner and are
// freed for good when they are released
void dev_pool_release(muxgl_handle* h, bool forget_owner) {
  dev_registry& R = dev_reg();
  std::vector<void*> out;
  {
    std::lock_guard<std::mutex> g(R.mu);
    for (auto& kv : h->pool) {
      out.push_back(kv.second);
      R.blocks.erase(kv.second);
    }
    h->pool.clear();
    h->pool_bytes = 0;
    if (forget_owner) {
      for (auto& kv : R.blocks)
        if (kv.second.owner == h) kv.second.owner = nullptr;
      R.handles.erase(h);
    }
  }
  for (void* q : out) (void)hipFree(q);
}

void dev_pool_release_all() {
  // (one critical section for the walk over ALL handles: a handle that another thread destroys meanwhile leaves the
  //  registry under the same lock, so no pointer of the set is followed after its handle is gone)
  dev_registry& R = dev_reg();
  std::vector<void*> out;
  {
    std::lock_guard<std::mutex> g(R.mu);
    for (muxgl_handle* h : R.handles) {
      for (auto& kv : h->pool) {
        out.push_back(kv.second);
        R.blocks.erase(kv.second);
      }
      h->pool.clear();
      h->pool_bytes = 0;
    }
  }
  for (void* q : out) (void)hipFree(q);
}


// One handle on one device.  shared_stream != nullptr: the handle launches on that stream and does not own it (the
// column slab of a slabbed handle shares its parent's stream, so the phases of an EM iteration stay in stream order).
int muxgl_handle_create(int dev, int32_t flags, hipStream_t shared_stream, muxgl_handle** out, std::string* errp) {
  std::string& err = *errp;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    err = std::string("muxgl_create: no HIP device available (") +
          (e != hipSuccess ? hipGetErrorString(e) : "device count 0") + "); libmuxgl has no CPU fallback";
    return 1;
  }
  if (dev < 0 || dev >= ndev) {
    err = "muxgl_create: device_id out of range";
    return 1;
  }
  muxgl_handle* h = new muxgl_handle();
  h->device = dev;
  h->flags = flags;
  auto fail = [&](const char* what, hipError_t er) {
    err = std::string("muxgl_create: ") + what + ": " + hipGetErrorString(er);
    muxgl_destroy(h);
    return 1;
  };
  if ((e = hipSetDevice(dev)) != hipSuccess) return fail("hipSetDevice", e);
  if (shared_stream) {
    h->stream = shared_stream;
    h->owns_stream = false;
  } else if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) {
    h->stream = nullptr;
    return fail("hipStreamCreate", e);
  }
  for (int i = 0; i < 2 * MUXGL_T_COUNT; ++i)
    if ((e = hipEventCreate(&h->ev[i])) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventCreateWithFlags(&h->ev_stat, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipHostMalloc((void**)&h->h_fstat, 4 * sizeof(int32_t))) != hipSuccess) return fail("hipHostMalloc(stat)", e);

  // Phred tables, PhredHelper.cpp:24-41: phred2Err[i] = (i > 1) ? pow(0.1, i*0.1) : 0.75; phred2Mat = 1 - Err.
  // The packed read byte carries 7 bits of quality, so 128 entries of each suffice.
  // [256..383] holds Err/3.0 (cmd_cram_demuxlet.cpp:666-667), divided here once, IEEE-exactly, instead of per read.
  double lut[384];
  for (int i = 0; i < 128; ++i) {
    lut[i] = (i > 1) ? pow(0.1, i * 0.1) : 0.75;
    lut[128 + i] = 1. - lut[i];
    lut[256 + i] = lut[i] / 3.0;
  }
  if ((e = hipMalloc((void**)&h->d_lut, sizeof(lut))) != hipSuccess) return fail("hipMalloc(lut)", e);
  if ((e = hipMemcpy(h->d_lut, lut, sizeof(lut), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(lut)", e);
  if ((e = hipMalloc((void**)&h->d_fstat, 4 * sizeof(int32_t))) != hipSuccess) return fail("hipMalloc(stat)", e);
  *out = h;
  return 0;
}

// structural validation of a packed pileup (the reference's containers make these states unrepresentable), cells cut
// into slices checked by host threads: 480 M entries took 0.32 s in one loop, the largest stage of that hand-over
int muxgl_validate_pileup(muxgl_handle* h, int64_t C, int64_t S, int64_t nnz, int64_t R, const int64_t* cell_ptr,
                           const int32_t* entry_snp, const int64_t* entry_rptr, int64_t* maxlen_out) {
  if (cell_ptr[0] != 0 || cell_ptr[C] != nnz) MUXGL_FAIL(h, "muxgl_set_pileup: cell_ptr must span [0,nnz]");
  if (entry_rptr[0] != 0 || entry_rptr[nnz] != R) MUXGL_FAIL(h, "muxgl_set_pileup: entry_rptr must span [0,R]");
  unsigned hw = std::thread::hardware_concurrency();
  int T = (int)(hw ? (hw > 16 ? 16 : hw) : 1);
  if (nnz < (1 << 20)) T = 1;
  struct result {
    int64_t maxlen = 0;
    std::string err;
  };
  std::vector<result> res((size_t)T);
  auto work = [&](int t) {
    result& r = res[(size_t)t];
    char buf[256];
    const int64_t cb = C * t / T, ce = C * (t + 1) / T;
    for (int64_t c = cb; c < ce; ++c) {
      const int64_t b = cell_ptr[c], e1 = cell_ptr[c + 1], len = e1 - b;
      if (len < 0 || b < 0 || e1 > nnz) {
        snprintf(buf, sizeof(buf), "muxgl_set_pileup: cell_ptr not monotone at cell %lld", (long long)c);
        r.err = buf;
        return;
      }
      if (len > r.maxlen) r.maxlen = len;
      for (int64_t e = b; e < e1; ++e) {
        if (entry_snp[e] < 0 || entry_snp[e] >= S) {
          snprintf(buf, sizeof(buf), "muxgl_set_pileup: entry %lld has SNP id %d outside [0,%lld)", (long long)e,
                   entry_snp[e], (long long)S);
          r.err = buf;
          return;
        }
        if (e > b && entry_snp[e] <= entry_snp[e - 1]) {
          snprintf(buf, sizeof(buf), "muxgl_set_pileup: SNP ids of cell %lld are not strictly ascending", (long long)c);
          r.err = buf;
          return;
        }
        if (entry_rptr[e + 1] < entry_rptr[e]) {
          snprintf(buf, sizeof(buf), "muxgl_set_pileup: entry_rptr not monotone at %lld", (long long)e);
          r.err = buf;
          return;
        }
      }
    }
  };
  if (T == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  int64_t maxlen = 0;
  for (auto& r : res) {
    if (!r.err.empty()) {
      h->err = r.err;
      return 1;
    }
    if (r.maxlen > maxlen) maxlen = r.maxlen;
  }
  *maxlen_out = maxlen;
  return 0;
}

// Hand-over of a packed pileup in one of three roles: the whole pileup (every derived table), a rank's row slab (the same,
// its cells only), or a column slab (all cells, the entries of a SNP range; it only ever feeds the SNP-major view of the
// ordered M-step, so the chunk plans, the wave plan and the packed quad records are skipped).
int muxgl_set_pileup_role(muxgl_handle* h, int role, int64_t C, int64_t S, int64_t nnz, int64_t R, const int64_t* cell_ptr,
                          const int32_t* entry_snp, const int64_t* entry_rptr, const uint8_t* reads, bool trusted) {
  HIPCHK(h, hipSetDevice(h->device));
  if (C < 0 || S < 0 || nnz < 0 || R < 0) MUXGL_FAIL(h, "muxgl_set_pileup: negative size");
  if (!cell_ptr || !entry_rptr || (nnz > 0 && !entry_snp) || (R > 0 && !reads))
    MUXGL_FAIL(h, "muxgl_set_pileup: NULL array");
  if (S > INT32_MAX) MUXGL_FAIL(h, "muxgl_set_pileup: S exceeds int32");
  if (C > INT32_MAX) MUXGL_FAIL(h, "muxgl_set_pileup: C exceeds int32");
  host_timer tm;
  int64_t maxlen = 0;
  if (trusted) {  // a slab cut by the library from arrays it has validated: only the longest cell is needed
    for (int64_t c = 0; c < C; ++c) maxlen = std::max(maxlen, cell_ptr[c + 1] - cell_ptr[c]);
  } else if (muxgl_validate_pileup(h, C, S, nnz, R, cell_ptr, entry_snp, entry_rptr, &maxlen)) {
    return 1;
  }
  tm.lap("set_pileup: validation");
  if (h->col) {  // a new pileup: the column slab of the previous one goes
    HIPCHK(h, hipStreamSynchronize(h->stream));
    muxgl_destroy(h->col);
    h->col = nullptr;
  }
  h->role = role;
  h->C_total = C;
  h->cell_base = 0;
  h->C = C;
  h->S = S;
  h->nnz = nnz;
  h->R = R;
  h->max_cell_entries = maxlen;
  if (dev_alloc(h, &h->d_cell_ptr, (size_t)C + 1)) return 1;
  if (dev_alloc(h, &h->d_entry_snp, (size_t)nnz)) return 1;
  if (dev_alloc(h, &h->d_entry_rptr, (size_t)nnz + 1)) return 1;
  if (dev_alloc(h, &h->d_reads, (size_t)R)) return 1;
  HIPCHK(h, hipMemcpyAsync(h->d_cell_ptr, cell_ptr, sizeof(int64_t) * (C + 1), hipMemcpyHostToDevice, h->stream));
  if (nnz) HIPCHK(h, hipMemcpyAsync(h->d_entry_snp, entry_snp, sizeof(int32_t) * nnz, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_entry_rptr, entry_rptr, sizeof(int64_t) * (nnz + 1), hipMemcpyHostToDevice, h->stream));
  if (R) HIPCHK(h, hipMemcpyAsync(h->d_reads, reads, (size_t)R, hipMemcpyHostToDevice, h->stream));

  // per-cell result buffers
  if (C > h->dcells_cap && role != MUXGL_ROLE_COLS) {
    if (dev_alloc(h, &h->d_dcells, (size_t)C)) return 1;
    if (dev_alloc(h, &h->d_fcells, (size_t)C)) return 1;
    if (dev_alloc(h, &h->d_clust, (size_t)(C + MUXGL_XCHG_PAD))) return 1;
    if (h->h_dcells) (void)hipHostFree(h->h_dcells);
    if (h->h_fcells) (void)hipHostFree(h->h_fcells);
    h->h_dcells = nullptr;
    h->h_fcells = nullptr;
    HIPCHK(h, hipHostMalloc((void**)&h->h_dcells, sizeof(muxgl_demux_cell) * (size_t)(C ? C : 1)));
    HIPCHK(h, hipHostMalloc((void**)&h->h_fcells, sizeof(muxgl_fmx_cell) * (size_t)(C ? C : 1)));
    h->dcells_cap = C;
  }
  if (role == MUXGL_ROLE_COLS && dev_alloc(h, &h->d_clust, (size_t)(C + MUXGL_XCHG_PAD))) return 1;
  h->ll_zeroed = false;  // the LL tensor must be re-zeroed for the new cell set
  tm.lap("set_pileup: alloc + H2D enqueue");
  if (role != MUXGL_ROLE_COLS) {
    if (demux_row_plan(h)) return 1;
    tm.lap("set_pileup: chunk plans (row, quad)");
    if (demux_wave_plan(h, cell_ptr)) return 1;
    tm.lap("set_pileup: wave plan");
    if (plan_build_qent(h)) return 1;  // packed per-entry records of the quad kernel
    if (plan_build_lin(h)) return 1;   // which entries are linear in the genotypes (one usable read)
    tm.lap("set_pileup: quad entry records, linear-entry bits");
  }
  h->fmx_prepared = false;
  h->K = 0;
  dev_free(&h->d_sgn);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  tm.lap("set_pileup: final sync");
  return 0;
}

extern "C" {

int muxgl_group_peer_stats(const muxgl_handle* h, int32_t* out) {
  if (!h || !out) return 1;
  out[0] = out[1] = out[2] = 0;
  if (h->group) group_peer_stats(h, out);
  return 0;
}

int muxgl_version(void) { return MUXGL_VERSION; }

const char* muxgl_last_error(const muxgl_handle* h) { return h ? h->err.c_str() : g_muxgl_create_error.c_str(); }

int muxgl_create(const muxgl_config* cfg, muxgl_handle** out) {
  if (!out) {
    g_muxgl_create_error = "muxgl_create: out is NULL";
    return 1;
  }
  *out = nullptr;
  if (cfg && cfg->struct_size != (int32_t)sizeof(muxgl_config)) {
    g_muxgl_create_error = "muxgl_create: muxgl_config.struct_size is not sizeof(muxgl_config) of this library (MUXGL_VERSION " +
                           std::to_string(MUXGL_VERSION) + "): initialise the struct with MUXGL_CONFIG_INIT";
    return 1;
  }
  if (cfg && (cfg->n_devices < 0 || cfg->n_devices > MUXGL_MAX_DEVICES)) {
    g_muxgl_create_error = "muxgl_create: n_devices outside [0, MUXGL_MAX_DEVICES]";
    return 1;
  }
  if (cfg && cfg->n_devices > 1) return group_create(cfg, out, &g_muxgl_create_error);
  const int dev = cfg ? (cfg->n_devices == 1 ? cfg->device_ids[0] : cfg->device_id) : 0;
  return muxgl_handle_create(dev, cfg ? cfg->flags : 0, nullptr, out, &g_muxgl_create_error);
}

void muxgl_destroy(muxgl_handle* h) {
  if (!h) return;
  if (h->group) {
    group_destroy(h);
    delete h;
    return;
  }
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->col) muxgl_destroy(h->col);
  h->col = nullptr;
  dev_free(&h->d_cell_ptr);
  dev_free(&h->d_entry_snp);
  dev_free(&h->d_entry_rptr);
  dev_free(&h->d_reads);
  dev_free(&h->d_entry_cell);
  dev_free(&h->d_qent);
  dev_free(&h->d_lin);
  dev_free(&h->d_flin);
  fmx_wave_streams_release(h);
  plan_lin_streams_release(h);
  dev_free(&h->d_lut);
  dev_free(&h->d_gp);
  dev_free(&h->d_has_gp);
  dev_free(&h->d_gpq);
  dev_free(&h->d_gp0s);
  dev_free(&h->d_gmq);
  dev_free(&h->d_ll);
  dev_free(&h->d_dcells);
  dev_free(&h->d_llw);
  dev_free(&h->d_pairs);
  dev_free(&h->d_af);
  dev_free(&h->d_egls);
  dev_free(&h->d_ecnt);
  dev_free(&h->d_cgls);
  dev_free(&h->d_ccnt);
  dev_free(&h->d_cgp);
  dev_free(&h->d_clust);
  dev_free(&h->d_clust8);
  h->clust8_n = -1;
  dev_free(&h->d_fcells);
  dev_free(&h->d_fll);
  dev_free(&h->d_fstat);
  dev_free(&h->d_prev_clust);
  dev_free(&h->d_prev_state);
  dev_free(&h->d_flagged);
  dev_free(&h->d_xc_epoch);
  dev_free(&h->d_xc);
  fmx_exact_release(h);
  dev_free(&h->d_snp_ptr);
  dev_free(&h->d_snp_entry);
  dev_free(&h->d_snp_cell);
  dev_free(&h->d_segls);
  dev_free(&h->d_segls6);
  dev_free(&h->d_scode);
  dev_free(&h->d_mtab);
  dev_free(&h->d_egls6);
  dev_free(&h->d_cgpq);
  dev_free(&h->d_ceq);
  dev_free(&h->d_secnt);
  dev_free(&h->d_sgn);
  demux_row_free(h);
  demux_wave_free(h);
  if (h->h_dcells) (void)hipHostFree(h->h_dcells);
  if (h->h_fcells) (void)hipHostFree(h->h_fcells);
  if (h->h_fstat) (void)hipHostFree(h->h_fstat);
  if (h->ev_stat) (void)hipEventDestroy(h->ev_stat);
  for (int i = 0; i < 2 * MUXGL_T_COUNT; ++i)
    if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  dev_pool_release(h, true);  // the cached blocks go back to the driver; blocks still out lose their owner
  if (h->stream && h->owns_stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int muxgl_set_pileup(muxgl_handle* h, int64_t C, int64_t S, int64_t nnz, int64_t R, const int64_t* cell_ptr,
                     const int32_t* entry_snp, const int64_t* entry_rptr, const uint8_t* reads) {
  if (!h) return 1;
  if (h->group) return group_set_pileup(h, C, S, nnz, R, cell_ptr, entry_snp, entry_rptr, reads);
  return muxgl_set_pileup_role(h, MUXGL_ROLE_FULL, C, S, nnz, R, cell_ptr, entry_snp, entry_rptr, reads);
}

int muxgl_demux_set_gp(muxgl_handle* h, int32_t V, const double* gp, const uint8_t* has_gp) {
  if (!h) return 1;
  if (h->group) return group_demux_set_gp(h, V, gp, has_gp);
  HIPCHK(h, hipSetDevice(h->device));
  if (V < 1 || V > 255) MUXGL_FAIL(h, "muxgl_demux_set_gp: V=%d outside [1,255]", V);
  if (!gp || !has_gp) MUXGL_FAIL(h, "muxgl_demux_set_gp: NULL array");
  if (h->S <= 0 && h->nnz > 0) MUXGL_FAIL(h, "muxgl_demux_set_gp: call muxgl_set_pileup first");
  const size_t n = (size_t)h->S * V * 3;
  if (dev_alloc(h, &h->d_gp, n)) return 1;
  if (dev_alloc(h, &h->d_has_gp, (size_t)h->S)) return 1;
  if (n) HIPCHK(h, hipMemcpyAsync(h->d_gp, gp, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  if (h->S) HIPCHK(h, hipMemcpyAsync(h->d_has_gp, has_gp, (size_t)h->S, hipMemcpyHostToDevice, h->stream));
  dev_free(&h->d_gpq);
  dev_free(&h->d_gp0s);
  dev_free(&h->d_gmq);
  demux_ring_release(h);  // (its records name the neutral table row for markers without genotypes)
  if (h->qrow) {  // the oct kernel's partitioned records carry has_gp (quad_lrec::code): rebuilt on the next run
    dev_free(&h->qrow->d_qent_lin);
    dev_free(&h->qrow->d_chunk_nlin);
    dev_free(&h->qrow->d_orec);
    dev_free(&h->qrow->d_unit_ptr);
    dev_free(&h->qrow->d_quad_order);
  }
  h->gp_min_sum = 1.0;
  if (V > 32 && h->S > 0) {
    // smallest triple sum: the one-kernel sweep of demux_ring.hip multiplies g_j x pG x g_k into its products directly and
    // renormalises them on a budget of 37 bits per entry, which holds while every factor is >= 1.1e-11, i.e. for sums
    // >= ~0.35; posteriors scaled further down than that (nothing the loaders produce) take the split sweep instead
    double mn = 1.0;
    for (int64_t s = 0; s < h->S; ++s) {
      if (!has_gp[s]) continue;
      const double* row = gp + (size_t)s * V * 3;
      for (int j = 0; j < V; ++j) mn = std::fmin(mn, (row[3 * j] + row[3 * j + 1]) + row[3 * j + 2]);
    }
    h->gp_min_sum = mn;
  }
  if (V <= 32 && h->S > 0) {
    const int P = V <= 16 ? 8 : 16;  // lanes per entry of the oct tiling (demux_oct.hip): lane p owns samples p and p + P
    // Do all triples sum to 1 within 4 ulp?  (Hard calls through the reference's error mixing do, sc_drop_seq.cpp:287-315;
    // posteriors normalised in float do not.)  The rows of the oct kernel then carry no sums -- see demux_oct.hip.
    bool unit = !(h->flags & MUXGL_FLAG_NO_LINEAR_ENTRIES);  // (the flag keeps the general formats: tests compare the two)
    for (int64_t s = 0; s < h->S && unit; ++s) {
      if (!has_gp[s]) continue;
      const double* row = gp + (size_t)s * V * 3;
      for (int j = 0; j < V; ++j) {
        const double sm = (row[3 * j] + row[3 * j + 1]) + row[3 * j + 2];
        if (!(std::fabs(sm - 1.0) <= 0x1p-50)) {
          unit = false;
          break;
        }
      }
    }
    h->gp_unit_sums = unit;
    // Layout for the oct kernel (demux_oct.hip): lane p of an entry's eight lanes owns samples p and p + 8.  General
    // format: 6 doubles d = 3c + l, read as three 16-byte pieces; piece t of the eight lanes is stored contiguously
    // ([S][3][8][2]) so that one load instruction of an entry covers one whole 128-byte line.
    // Samples >= V are padded with (1,0,0), which makes their factors exactly 1.
    // Row S is a dummy marker, (1,0,0) for every sample and sum 1: padding entries and markers without genotypes are
    // pointed at it, so the kernel loads rows unconditionally.
    std::vector<double> q((size_t)(h->S + 1) * 6 * P), g0((size_t)h->S + 1);
    for (int64_t s = 0; s <= h->S; ++s) {
      const double* row = s < h->S ? gp + (size_t)s * V * 3 : nullptr;
      const bool have = row && has_gp[s];
      for (int pp = 0; pp < P; ++pp)
        for (int c = 0; c < 2; ++c)
          for (int l = 0; l < 3; ++l) {
            const int j = pp + P * c, d = 3 * c + l;
            const double v = (j < V && have) ? row[j * 3 + l] : (l == 0 ? 1.0 : 0.0);  // no genotypes: neutral row
            q[(size_t)s * 6 * P + ((size_t)(d / 2) * P + pp) * 2 + (d & 1)] = v;
          }
      // a SNP without genotypes (gps == NULL, cmd_cram_demuxlet.cpp:733) is marked by a negative sum
      g0[(size_t)s] = s == h->S ? 1.0 : (have ? (row[0] + row[1]) + row[2] : -1.0);
    }
    // The same rows as moments (s, rho = (g1 + 2 g2) / s) for the entries with one usable read: 16 bytes per sample in
    // sample order ([S + 1][16][2]; lane p reads samples p and p + 8, each a whole line per entry); with unit sums rho
    // alone, (rho_p, rho_p+8) adjacent ([S + 1][8][2]: one line per entry).
    std::vector<double> gm((size_t)(h->S + 1) * (unit ? 2 * P : 4 * P));  // (P = 16: [S + 1][32][2] / [S + 1][16][2])
    for (int64_t s = 0; s <= h->S; ++s)
      for (int j = 0; j < 2 * P; ++j) {
        double sm = 1.0, rho = 0.0;
        if (s < h->S && j < V && has_gp[s]) {
          const double* t = gp + ((size_t)s * V + j) * 3;
          sm = (t[0] + t[1]) + t[2];
          rho = sm > 0.0 ? std::fma(2.0, t[2], t[1]) / sm : 0.0;
        }
        if (unit) {
          gm[(size_t)s * 2 * P + (size_t)(j % P) * 2 + (j / P)] = rho;
        } else {
          gm[(size_t)s * 4 * P + (size_t)j * 2] = sm;
          gm[(size_t)s * 4 * P + (size_t)j * 2 + 1] = rho;
        }
      }
    if (dev_alloc(h, &h->d_gmq, gm.size())) return 1;
    HIPCHK(h, hipMemcpy(h->d_gmq, gm.data(), sizeof(double) * gm.size(), hipMemcpyHostToDevice));
    if (dev_alloc(h, &h->d_gpq, q.size())) return 1;
    if (dev_alloc(h, &h->d_gp0s, g0.size())) return 1;
    HIPCHK(h, hipMemcpy(h->d_gpq, q.data(), sizeof(double) * q.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_gp0s, g0.data(), sizeof(double) * g0.size(), hipMemcpyHostToDevice));
  }
  if (demux_gp_neutral_rows(h, V)) return 1;  // rows of markers without genotypes: (1,0,0) in the device copy
  h->V = V;
  h->have_dp = false;
  h->pairs_valid = false;
  h->ll_zeroed = false;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

static int check_demux_params(muxgl_handle* h, const muxgl_demux_params* p) {
  if (!p) MUXGL_FAIL(h, "demux params NULL");
  if (p->n_alpha < 1 || p->n_alpha > MUXGL_MAX_ALPHA)
    MUXGL_FAIL(h, "n_alpha=%d outside [1,%d]", p->n_alpha, MUXGL_MAX_ALPHA);
  if (!h->d_cell_ptr) MUXGL_FAIL(h, "no pileup set (muxgl_set_pileup)");
  if (!h->d_gp) MUXGL_FAIL(h, "no GP tensor set (muxgl_demux_set_gp)");
  return 0;
}

int muxgl_demux_run(muxgl_handle* h, const muxgl_demux_params* p, muxgl_demux_cell* out, double* full_ll) {
  if (!h) return 1;
  if (h->group) return group_demux_run(h, p, out, full_ll);
  HIPCHK(h, hipSetDevice(h->device));
  if (check_demux_params(h, p)) return 1;
  clear_timing(h);
  if (h->C > 0) {
    h->want_full_ll = full_ll != nullptr;
    if (demux_launch(h, p)) return 1;
    if (!h->records_on_host) {
      tic(h, MUXGL_T_DEMUX_D2H);
      HIPCHK(h, hipMemcpyAsync(h->h_dcells, h->d_dcells, sizeof(muxgl_demux_cell) * (size_t)h->C,
                               hipMemcpyDeviceToHost, h->stream));
      toc(h, MUXGL_T_DEMUX_D2H);
    }
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  collect_timing(h);
  if (out && h->C) memcpy(out, h->h_dcells, sizeof(muxgl_demux_cell) * (size_t)h->C);
  if (full_ll && h->C)
    HIPCHK(h, hipMemcpy(full_ll, h->d_ll, sizeof(double) * (size_t)h->C * h->V * h->V * p->n_alpha,
                        hipMemcpyDeviceToHost));
  return 0;
}

const muxgl_demux_cell* muxgl_demux_results(const muxgl_handle* h) {
  if (h && h->group) return group_demux_results(h);
  return h ? h->h_dcells : nullptr;
}

int muxgl_demux_get_entry_pg(muxgl_handle* h, double* pg) {
  if (!h) return 1;
  if (h->group) return group_demux_get_entry_pg(h, pg);
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->have_dp) MUXGL_FAIL(h, "muxgl_demux_get_entry_pg: no previous muxgl_demux_run");
  if (!pg) MUXGL_FAIL(h, "muxgl_demux_get_entry_pg: NULL output");
  const size_t n = (size_t)h->nnz * h->last_dp.n_alpha * 9;
  double* d_pg = nullptr;
  if (dev_alloc(h, &d_pg, n)) return 1;
  int rc = demux_entry_pg_launch(h, &h->last_dp, d_pg);
  if (!rc) {
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && n) e = hipMemcpy(pg, d_pg, sizeof(double) * n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      h->err = std::string("muxgl_demux_get_entry_pg: ") + hipGetErrorString(e);
      rc = 1;
    }
  }
  dev_free(&d_pg);
  return rc;
}

void* muxgl_stream(const muxgl_handle* h) { return (h && !h->group) ? (void*)h->stream : nullptr; }

int muxgl_get_timing(const muxgl_handle* h, float* ms) {
  if (!h || !ms) return 1;
  if (h->group) return group_get_timing(h, ms);
  memcpy(ms, h->ms, sizeof(float) * MUXGL_T_COUNT);
  return 0;
}

int muxgl_get_timing_sum(muxgl_handle* h, double* ms_sum, int64_t* calls, int32_t reset) {
  if (!h) return 1;
  MUXGL_NOT_FOR_GROUPS(h, "muxgl_get_timing_sum");
  if (ms_sum) memcpy(ms_sum, h->ms_sum, sizeof(double) * MUXGL_T_COUNT);
  if (calls) *calls = h->ms_calls;
  if (reset) {
    memset(h->ms_sum, 0, sizeof(h->ms_sum));
    h->ms_calls = 0;
  }
  return 0;
}

}  // extern "C"
