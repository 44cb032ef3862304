// muxgl_group.hip -- a device group behind one muxgl handle (muxgl_config.n_devices > 1).
//
// The reference parallelises by cutting the barcodes into groups and running one process per group (--group-list,
// README.md:168; sc_drop_seq.cpp:93-101,164-170).  A device group is the same cut inside the library:
//   * demuxlet (cmd_cram_demuxlet.cpp:636-1013 carries no cross-cell state): member r holds the cells [cb[r], cb[r+1])
//     (contiguous, balanced by entries) and a replica of the GP tensor; no exchange; the records are concatenated.
//   * freemuxlet: member r holds its ROW slab (its cells, every SNP: E-step, scans, re-assignment) and its COLUMN slab
//     (every cell, the SNPs [sb[r], sb[r+1]): cluster pileups, posteriors and the ordered clamped merge, which is a
//     chain per (cluster, SNP) in ascending cell id -- sc_drop_seq.h:77-101 -- and therefore owned by exactly one
//     member).  Per EM iteration two all-gathers: the cluster-GP rows f64[S][K][3] after the posterior phase and the
//     assignments i32[C] after the E-step.  xGMI is point to point, so each all-gather is N-1 direct peer copies per
//     member (hipMemcpyPeerAsync, pulled on the destination's stream behind an event of the source's stream); nothing is
//     staged through the host and no host thread waits inside an iteration: one thread enqueues the whole iteration on
//     all devices, then drains them.
// The hand-over calls (slicing, validation, H2D) run one host thread per member.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "common.hpp"
#include "score_exact.hpp"

struct muxgl_group {
  int n = 0;
  std::vector<muxgl_handle*> m;
  std::vector<int64_t> cb, eb, sb;  // [n+1] cell, entry and SNP cuts
  int64_t C = 0, S = 0, nnz = 0, R = 0;
  int32_t V = 0, K = 0, n_alpha = 0;
  bool have_pileup = false, have_cols = false, prepared = false;
  muxgl_demux_cell* h_dcells = nullptr;  // [C] records of the last demuxlet run, in cell order
  std::vector<hipEvent_t> ev_gp, ev_es;  // per member: posterior rows ready / assignments ready
  float ms[MUXGL_T_COUNT] = {};
  int32_t peer_pairs = 0, peer_enabled = 0, peer_refused = 0;  // of group_create's walk over the member pairs
};

namespace {

// f(r) for every member on its own host thread; the first failure's message becomes the group's
template <class F>
int for_members(muxgl_handle* h, F f) {
  muxgl_group* g = h->group;
  std::vector<int> rc((size_t)g->n, 0);
  std::vector<std::thread> th;
  for (int r = 0; r < g->n; ++r) th.emplace_back([&, r]() { rc[(size_t)r] = f(r); });
  for (auto& t : th) t.join();
  for (int r = 0; r < g->n; ++r)
    if (rc[(size_t)r]) {
      char buf[64];
      snprintf(buf, sizeof(buf), "device group member %d (device %d): ", r, g->m[(size_t)r]->device);
      h->err = std::string(buf) + g->m[(size_t)r]->err;
      return 1;
    }
  return 0;
}

void max_timing(muxgl_group* g) {
  for (int i = 0; i < MUXGL_T_COUNT; ++i) {
    g->ms[i] = 0.f;
    for (auto* m : g->m) g->ms[i] = std::max(g->ms[i], m->ms[i]);
  }
}

// n contiguous ranges over `items` units with near-equal weight: cut i at the first prefix sum >= total * i / n
std::vector<int64_t> cuts_by_prefix(const int64_t* prefix /*[items+1]*/, int64_t items, int n) {
  std::vector<int64_t> cut((size_t)n + 1, 0);
  const int64_t total = prefix[items];
  for (int i = 1; i < n; ++i) {
    const int64_t target = (int64_t)((__int128)total * i / n);
    int64_t c = std::lower_bound(prefix, prefix + items + 1, target) - prefix;
    cut[(size_t)i] = std::min(std::max(c, cut[(size_t)i - 1]), items);
  }
  cut[(size_t)n] = items;
  return cut;
}

}  // namespace

int group_create(const muxgl_config* cfg, muxgl_handle** out, std::string* err) {
  muxgl_handle* h = new muxgl_handle();
  muxgl_group* g = new muxgl_group();
  h->group = g;
  h->flags = cfg->flags;
  h->device = cfg->device_ids[0];
  g->n = cfg->n_devices;
  for (int r = 0; r < g->n; ++r) {
    muxgl_handle* m = nullptr;
    if (muxgl_handle_create(cfg->device_ids[r], cfg->flags & ~(MUXGL_FLAG_DEMUX_ONLY | MUXGL_FLAG_ASYNC_PHASES), nullptr,
                            &m, err)) {
      muxgl_destroy(h);
      return 1;
    }
    g->m.push_back(m);
  }
  // Direct peer copies between distinct devices.  Whatever goes wrong here -- no peer access between two devices, access
  // already enabled by another handle -- leaves hipMemcpyPeerAsync on its staged path, which is still correct, and must
  // not leave a sticky error behind for the next HIP call.  MUXGL_GROUP_NO_PEER=1 skips the enabling altogether (the
  // staged path on a box that does have peer access); MUXGL_FLAG_GROUP_PROBE_SELF also walks the pairs of members that
  // share a device (virtual ranks on one GPU: hipDeviceEnablePeerAccess(self) fails -- the error path, on any box).
  g->peer_pairs = g->peer_enabled = g->peer_refused = 0;
  const bool no_peer = getenv("MUXGL_GROUP_NO_PEER") != nullptr;
  const bool probe_self = (cfg->flags & MUXGL_FLAG_GROUP_PROBE_SELF) != 0;
  for (int a = 0; a < g->n; ++a)
    for (int b = 0; b < g->n; ++b) {
      const int da = g->m[(size_t)a]->device, db = g->m[(size_t)b]->device;
      if (a == b || (da == db && !probe_self)) continue;
      ++g->peer_pairs;
      int can = 0;
      hipError_t e = hipDeviceCanAccessPeer(&can, da, db);
      if (e != hipSuccess) (void)hipGetLastError();
      if (e == hipSuccess && da == db) can = 1;  // (probe: the runtime answers 0 for a device and itself; ask it to enable anyway)
      if (e == hipSuccess && can && !no_peer && hipSetDevice(da) == hipSuccess) {
        e = hipDeviceEnablePeerAccess(db, 0);
        if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ++g->peer_enabled;
        else ++g->peer_refused;
        if (e != hipSuccess) (void)hipGetLastError();
      } else {
        ++g->peer_refused;
      }
    }
  g->ev_gp.assign((size_t)g->n, nullptr);
  g->ev_es.assign((size_t)g->n, nullptr);
  for (int r = 0; r < g->n; ++r) {
    hipError_t e = hipSetDevice(g->m[(size_t)r]->device);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_gp[(size_t)r], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_es[(size_t)r], hipEventDisableTiming);
    if (e != hipSuccess) {
      *err = std::string("muxgl_create: device group events: ") + hipGetErrorString(e);
      muxgl_destroy(h);
      return 1;
    }
  }
  *out = h;
  return 0;
}

void group_destroy(muxgl_handle* h) {
  muxgl_group* g = h->group;
  if (!g) return;
  for (size_t r = 0; r < g->m.size(); ++r) {
    (void)hipSetDevice(g->m[r]->device);
    if (g->m[r]->stream) (void)hipStreamSynchronize(g->m[r]->stream);
  }
  for (size_t r = 0; r < g->ev_gp.size(); ++r) {
    if (g->ev_gp[r]) (void)hipEventDestroy(g->ev_gp[r]);
    if (g->ev_es[r]) (void)hipEventDestroy(g->ev_es[r]);
  }
  for (auto* m : g->m) muxgl_destroy(m);
  free(g->h_dcells);
  delete g;
  h->group = nullptr;
}

int group_set_pileup(muxgl_handle* h, int64_t C, int64_t S, int64_t nnz, int64_t R, const int64_t* cell_ptr,
                     const int32_t* entry_snp, const int64_t* entry_rptr, const uint8_t* reads) {
  muxgl_group* g = h->group;
  if (C < 0 || S < 0 || nnz < 0 || R < 0) MUXGL_FAIL(h, "muxgl_set_pileup: negative size");
  if (!cell_ptr || !entry_rptr || (nnz > 0 && !entry_snp) || (R > 0 && !reads))
    MUXGL_FAIL(h, "muxgl_set_pileup: NULL array");
  if (S > INT32_MAX || C > INT32_MAX) MUXGL_FAIL(h, "muxgl_set_pileup: C or S exceeds int32");
  host_timer tm;
  int64_t maxlen = 0;
  if (muxgl_validate_pileup(h, C, S, nnz, R, cell_ptr, entry_snp, entry_rptr, &maxlen)) return 1;
  tm.lap("group set_pileup: validation");
  g->have_pileup = g->have_cols = g->prepared = false;
  g->C = C;
  g->S = S;
  g->nnz = nnz;
  g->R = R;
  g->K = 0;
  // cells: contiguous ranges balanced by entries (the work unit of every sweep)
  g->cb = cuts_by_prefix(cell_ptr, C, g->n);
  g->eb.assign((size_t)g->n + 1, 0);
  for (int r = 0; r <= g->n; ++r) g->eb[(size_t)r] = cell_ptr[g->cb[(size_t)r]];
  // SNPs: contiguous ranges balanced by entries per SNP (the length of the merge chains), +1 so that bare SNPs spread
  const bool cols = !(h->flags & MUXGL_FLAG_DEMUX_ONLY);
  g->sb.assign((size_t)g->n + 1, 0);
  g->sb[(size_t)g->n] = S;
  if (cols) {
    unsigned hw = std::thread::hardware_concurrency();
    int T = (int)(hw ? (hw > 16 ? 16 : hw) : 1);
    if (nnz < (1 << 20)) T = 1;
    std::vector<std::vector<int32_t>> part((size_t)T, std::vector<int32_t>((size_t)S, 0));
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        int32_t* cnt = part[(size_t)t].data();
        for (int64_t e = nnz * t / T; e < nnz * (t + 1) / T; ++e) ++cnt[entry_snp[e]];
      });
    for (auto& x : th) x.join();
    std::vector<int64_t> pre((size_t)S + 1, 0);
    for (int64_t s = 0; s < S; ++s) {
      int64_t c = 1;
      for (int t = 0; t < T; ++t) c += part[(size_t)t][(size_t)s];
      pre[(size_t)s + 1] = pre[(size_t)s] + c;
    }
    g->sb = cuts_by_prefix(pre.data(), S, g->n);
    tm.lap("group set_pileup: SNP coverage + cuts");
  }
  const int rc = for_members(h, [&](int r) -> int {
    muxgl_handle* m = g->m[(size_t)r];
    const int64_t c0 = g->cb[(size_t)r], c1 = g->cb[(size_t)r + 1], nc = c1 - c0;
    const int64_t e0 = g->eb[(size_t)r], e1 = g->eb[(size_t)r + 1], ne = e1 - e0;
    const int64_t r0 = entry_rptr[e0], r1 = entry_rptr[e1];
    {  // row slab: the member's cells, every SNP; pointers rebased, payload arrays handed over in place
      std::vector<int64_t> cp((size_t)nc + 1), rp((size_t)ne + 1);
      for (int64_t i = 0; i <= nc; ++i) cp[(size_t)i] = cell_ptr[c0 + i] - e0;
      for (int64_t i = 0; i <= ne; ++i) rp[(size_t)i] = entry_rptr[e0 + i] - r0;
      if (muxgl_set_pileup_role(m, MUXGL_ROLE_FULL, nc, S, ne, r1 - r0, cp.data(), entry_snp + e0, rp.data(),
                                reads + r0, true))
        return 1;
    }
    if (!cols) return 0;
    // column slab: every cell, the entries with s0 <= SNP < s1 -- a contiguous run of each cell's ascending SNP ids
    const int32_t s0 = (int32_t)g->sb[(size_t)r], s1 = (int32_t)g->sb[(size_t)r + 1];
    std::vector<int64_t> cps((size_t)C + 1, 0), lo((size_t)C);
    int64_t nr = 0;
    for (int64_t c = 0; c < C; ++c) {
      const int32_t* b = entry_snp + cell_ptr[c];
      const int32_t* e = entry_snp + cell_ptr[c + 1];
      const int32_t* a = std::lower_bound(b, e, s0);
      const int32_t* z = std::lower_bound(a, e, s1);
      lo[(size_t)c] = a - entry_snp;
      cps[(size_t)c + 1] = cps[(size_t)c] + (z - a);
      nr += entry_rptr[z - entry_snp] - entry_rptr[a - entry_snp];
    }
    const int64_t ns = cps[(size_t)C];
    std::vector<int32_t> es((size_t)ns);
    std::vector<int64_t> rps((size_t)ns + 1);
    std::vector<uint8_t> rd((size_t)nr);
    int64_t rpos = 0;
    for (int64_t c = 0; c < C; ++c) {
      const int64_t a = lo[(size_t)c], k = cps[(size_t)c + 1] - cps[(size_t)c], o = cps[(size_t)c];
      if (!k) continue;
      memcpy(es.data() + o, entry_snp + a, sizeof(int32_t) * (size_t)k);
      const int64_t ra = entry_rptr[a];
      for (int64_t i = 0; i < k; ++i) rps[(size_t)(o + i)] = rpos + (entry_rptr[a + i] - ra);
      const int64_t nb = entry_rptr[a + k] - ra;
      if (nb) memcpy(rd.data() + rpos, reads + ra, (size_t)nb);
      rpos += nb;
    }
    rps[(size_t)ns] = rpos;
    return fmx_attach_column_slab(m, C, c0, s0, s1, ns, nr, cps.data(), es.data(), rps.data(), rd.data(), true);
  });
  if (rc) return 1;
  tm.lap("group set_pileup: slabs cut and handed over");
  free(g->h_dcells);
  g->h_dcells = (muxgl_demux_cell*)calloc((size_t)(C ? C : 1), sizeof(muxgl_demux_cell));
  if (!g->h_dcells) MUXGL_FAIL(h, "muxgl_set_pileup: out of host memory");
  g->have_pileup = true;
  g->have_cols = cols;
  h->C = C;
  h->S = S;
  h->nnz = nnz;
  h->R = R;
  return 0;
}

int group_demux_set_gp(muxgl_handle* h, int32_t V, const double* gp, const uint8_t* has_gp) {
  muxgl_group* g = h->group;
  if (!g->have_pileup) MUXGL_FAIL(h, "muxgl_demux_set_gp: call muxgl_set_pileup first");
  if (for_members(h, [&](int r) { return muxgl_demux_set_gp(g->m[(size_t)r], V, gp, has_gp); })) return 1;
  g->V = V;
  h->V = V;
  return 0;
}

int group_demux_run(muxgl_handle* h, const muxgl_demux_params* p, muxgl_demux_cell* out, double* full_ll) {
  muxgl_group* g = h->group;
  if (!g->have_pileup || g->V < 1) MUXGL_FAIL(h, "muxgl_demux_run: no pileup / GP tensor set");
  if (!p) MUXGL_FAIL(h, "demux params NULL");
  if (p->n_alpha < 1 || p->n_alpha > MUXGL_MAX_ALPHA) MUXGL_FAIL(h, "n_alpha=%d outside [1,%d]", p->n_alpha, MUXGL_MAX_ALPHA);
  const size_t per_cell = (size_t)g->V * g->V * p->n_alpha;
  if (for_members(h, [&](int r) {
        muxgl_handle* m = g->m[(size_t)r];
        const int64_t c0 = g->cb[(size_t)r];
        if (muxgl_demux_run(m, p, g->h_dcells + c0, full_ll ? full_ll + (size_t)c0 * per_cell : nullptr)) return 1;
        return 0;
      }))
    return 1;
  g->n_alpha = p->n_alpha;
  if (out && g->C) memcpy(out, g->h_dcells, sizeof(muxgl_demux_cell) * (size_t)g->C);
  max_timing(g);
  return 0;
}

const muxgl_demux_cell* group_demux_results(const muxgl_handle* h) { return h->group->h_dcells; }

int group_demux_get_entry_pg(muxgl_handle* h, double* pg) {
  muxgl_group* g = h->group;
  if (g->n_alpha < 1) MUXGL_FAIL(h, "muxgl_demux_get_entry_pg: no previous muxgl_demux_run");
  if (!pg) MUXGL_FAIL(h, "muxgl_demux_get_entry_pg: NULL output");
  return for_members(h, [&](int r) {
    if (g->eb[(size_t)r + 1] == g->eb[(size_t)r]) return 0;  // a member without entries has nothing to report
    return muxgl_demux_get_entry_pg(g->m[(size_t)r], pg + (size_t)g->eb[(size_t)r] * g->n_alpha * 9);
  });
}

int group_fmx_prepare(muxgl_handle* h, const double* af, double* cell_llk0, double* cell_llk2, int32_t* cell_nsnps,
                      int32_t* cell_nreads) {
  muxgl_group* g = h->group;
  if (!g->have_pileup) MUXGL_FAIL(h, "muxgl_fmx_prepare: no pileup set (muxgl_set_pileup)");
  if (!g->have_cols)
    MUXGL_FAIL(h, "muxgl_fmx_prepare: the group was created with MUXGL_FLAG_DEMUX_ONLY (no column slabs were cut)");
  if (for_members(h, [&](int r) {
        const int64_t c0 = g->cb[(size_t)r];
        return muxgl_fmx_prepare(g->m[(size_t)r], af, cell_llk0 ? cell_llk0 + c0 : nullptr,
                                 cell_llk2 ? cell_llk2 + c0 : nullptr, cell_nsnps ? cell_nsnps + c0 : nullptr,
                                 cell_nreads ? cell_nreads + c0 : nullptr);
      }))
    return 1;
  // cells whose scores come within rounding reach of each other (in the whole job's order): the reference's own sums, each
  // from the member that holds the cell (score_exact.hpp) -- the group returns what a one-device handle returns
  h->fmx_exact_scores = 0;
  if (cell_llk0 && cell_llk2) {
    std::vector<std::vector<int64_t>> cps((size_t)g->n);
    auto exact_sums = [&](const std::vector<int32_t>& cells) -> int {
      size_t a = 0;
      for (int r = 0; r < g->n && a < cells.size(); ++r) {
        const int64_t c0 = g->cb[(size_t)r], c1 = g->cb[(size_t)r + 1];
        size_t b = a;
        while (b < cells.size() && cells[b] < c1) ++b;
        if (b > a) {
          muxgl_handle* m = g->m[(size_t)r];
          std::vector<int32_t> local(cells.begin() + (long)a, cells.begin() + (long)b);
          for (int32_t& c : local) c -= (int32_t)c0;
          if (hipSetDevice(m->device) != hipSuccess || (cps[(size_t)r].empty() && score_exact::fetch_cell_ptr(m, &cps[(size_t)r])) ||
              score_exact::compute(m, local, cps[(size_t)r], cell_llk0 + c0, cell_llk2 + c0)) {
            h->err = "device group member: " + m->err;
            return 1;
          }
        }
        a = b;
      }
      return 0;
    };
    if (score_exact::settle_with(g->C, cell_llk0, cell_llk2, &h->fmx_exact_scores, &h->err, exact_sums)) return 1;
  }
  g->prepared = true;
  g->K = 0;
  max_timing(g);
  return 0;
}

int group_fmx_get_entry_gls(muxgl_handle* h, double* gls, int32_t* counts) {
  muxgl_group* g = h->group;
  if (!g->prepared) MUXGL_FAIL(h, "muxgl_fmx_get_entry_gls: call muxgl_fmx_prepare first");
  return for_members(h, [&](int r) {
    const size_t e0 = (size_t)g->eb[(size_t)r];
    return muxgl_fmx_get_entry_gls(g->m[(size_t)r], gls ? gls + e0 * 9 : nullptr, counts ? counts + e0 * 3 : nullptr);
  });
}

void group_peer_stats(const muxgl_handle* h, int32_t* out) {
  out[0] = h->group->peer_pairs, out[1] = h->group->peer_enabled, out[2] = h->group->peer_refused;
}

void group_fmx_exact_stats(const muxgl_handle* h, int64_t* cells, int64_t* changed, int64_t* unresolved) {
  *cells = *changed = *unresolved = 0;
  for (muxgl_handle* m : h->group->m) {
    *cells += m->fmx_exact_cells;
    *changed += m->fmx_exact_changed;
    *unresolved += m->fmx_exact_unresolved;
  }
}

int group_fmx_set_clusters(muxgl_handle* h, int32_t K, const int32_t* clust) {
  muxgl_group* g = h->group;
  if (!g->prepared) MUXGL_FAIL(h, "muxgl_fmx_set_clusters: call muxgl_fmx_prepare first");
  if (for_members(h, [&](int r) { return muxgl_fmx_set_clusters(g->m[(size_t)r], K, clust); })) return 1;
  g->K = K;
  h->K = K;
  max_timing(g);
  return 0;
}

// One EM iteration (cmd_cram_freemux2.cpp:375-597) on the group.  Everything is enqueued from this thread; the devices
// wait for each other through events only.  Buffer hazards: a member's posterior rows are rewritten by its next
// posterior phase, which sits behind its M-step, which sits behind its assignment pulls, which wait for every member's
// E-step -- and a member's E-step sits behind its pulls of those rows; the assignment slices likewise.
int group_fmx_iterate(muxgl_handle* h, const muxgl_fmx_params* p, muxgl_fmx_cell* out, int32_t* nsingle, int32_t* namb,
                      int32_t* nchanged, double* full_ll) {
  muxgl_group* g = h->group;
  if (!p) MUXGL_FAIL(h, "muxgl_fmx_iterate: params NULL");
  if (!g->prepared || g->K < 1) MUXGL_FAIL(h, "muxgl_fmx_iterate: call muxgl_fmx_prepare and muxgl_fmx_set_clusters first");
  const int n = g->n, K = g->K;
  const size_t row = (size_t)K * 3;
#define GCHK(m, call)                                                                                     \
  do {                                                                                                    \
    hipError_t _e = (call);                                                                               \
    if (_e != hipSuccess)                                                                                 \
      MUXGL_FAIL(h, "device group, device %d: %s failed: %s", (m)->device, #call, hipGetErrorString(_e)); \
  } while (0)
  auto phase_fail = [&](muxgl_handle* m) {
    h->err = m->err;
    return 1;
  };
  for (int r = 0; r < n; ++r) {  // cluster genotype posteriors of the own SNP range
    muxgl_handle* m = g->m[(size_t)r];
    GCHK(m, hipSetDevice(m->device));
    clear_timing(m);
    if (fmx_phase_gp(m, p)) return phase_fail(m);
    GCHK(m, hipEventRecord(g->ev_gp[(size_t)r], m->stream));
  }
  for (int r = 0; r < n; ++r) {  // all-gather of the posterior rows, then the E-step of the own cells
    muxgl_handle* m = g->m[(size_t)r];
    GCHK(m, hipSetDevice(m->device));
    for (int o = 0; o < n; ++o) {
      if (o == r) continue;
      muxgl_handle* src = g->m[(size_t)o];
      const int64_t s0 = g->sb[(size_t)o], s1 = g->sb[(size_t)o + 1];
      if (s1 <= s0) continue;
      GCHK(m, hipStreamWaitEvent(m->stream, g->ev_gp[(size_t)o], 0));
      GCHK(m, hipMemcpyPeerAsync(m->d_cgp + (size_t)s0 * row, m->device, src->d_cgp + (size_t)s0 * row, src->device,
                                 sizeof(double) * (size_t)(s1 - s0) * row, m->stream));
    }
    if (fmx_phase_estep(m, p)) return phase_fail(m);
    GCHK(m, hipEventRecord(g->ev_es[(size_t)r], m->stream));
  }
  for (int r = 0; r < n; ++r) {  // all-gather of the assignments, then the ordered merge of the own SNP range
    muxgl_handle* m = g->m[(size_t)r];
    GCHK(m, hipSetDevice(m->device));
    for (int o = 0; o < n; ++o) {
      if (o == r) continue;
      muxgl_handle* src = g->m[(size_t)o];
      const int64_t c0 = g->cb[(size_t)o], c1 = g->cb[(size_t)o + 1];
      if (c1 <= c0) continue;
      GCHK(m, hipStreamWaitEvent(m->stream, g->ev_es[(size_t)o], 0));
      GCHK(m, hipMemcpyPeerAsync(m->col->d_clust + c0, m->device, src->col->d_clust + c0, src->device,
                                 sizeof(int32_t) * (size_t)(c1 - c0), m->stream));
    }
    if (fmx_phase_mstep(m)) return phase_fail(m);
  }
  // Near-tie calls (fmx_exact.hip): the members' listed cells are settled in the reference's arithmetic -- the SNP lists
  // united, every row computed by the member whose M-step range holds the SNP, the rows handed to all -- and, when an
  // assignment changed, the assignments are pulled and the ordered merge is run once more.
  int64_t listed = 0;
  for (int r = 0; r < n; ++r) {
    muxgl_handle* m = g->m[(size_t)r];
    GCHK(m, hipSetDevice(m->device));
    GCHK(m, hipStreamSynchronize(m->stream));
    listed += m->h_fstat[3];
  }
  if (listed > 0) {
    std::vector<int32_t> uni;
    for (int r = 0; r < n; ++r) {
      muxgl_handle* m = g->m[(size_t)r];
      GCHK(m, hipSetDevice(m->device));
      std::vector<int32_t> sn;
      if (fmx_exact_snps(m, &sn)) return phase_fail(m);
      uni.insert(uni.end(), sn.begin(), sn.end());
    }
    std::sort(uni.begin(), uni.end());
    uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
    const int64_t nu = (int64_t)uni.size();
    std::vector<double> rows((size_t)nu * row);
    std::vector<uint8_t> owned((size_t)nu, 0);
    for (int r = 0; r < n; ++r) {
      muxgl_handle* m = g->m[(size_t)r];
      GCHK(m, hipSetDevice(m->device));
      if (fmx_exact_rows(m, p, uni.data(), nu, rows.data(), owned.data())) return phase_fail(m);
    }
    for (int64_t i = 0; i < nu; ++i)
      if (!owned[(size_t)i]) MUXGL_FAIL(h, "device group: no member's M-step range holds SNP %d", uni[(size_t)i]);
    bool reassigned = false;
    for (int r = 0; r < n; ++r) {
      muxgl_handle* m = g->m[(size_t)r];
      GCHK(m, hipSetDevice(m->device));
      int64_t dl[3];
      int32_t re = 0;
      if (fmx_exact_finish(m, p, uni.data(), nu, rows.data(), dl, &re)) return phase_fail(m);
      for (int i = 0; i < 3; ++i) m->h_fstat[i] += (int32_t)dl[i];
      m->h_fstat[3] = 0;
      reassigned = reassigned || re != 0;
    }
    if (reassigned) {
      for (int r = 0; r < n; ++r) {  // (finish drained every member's stream: no events needed for these pulls)
        muxgl_handle* m = g->m[(size_t)r];
        GCHK(m, hipSetDevice(m->device));
        for (int o = 0; o < n; ++o) {
          if (o == r) continue;
          muxgl_handle* src = g->m[(size_t)o];
          const int64_t c0 = g->cb[(size_t)o], c1 = g->cb[(size_t)o + 1];
          if (c1 <= c0) continue;
          GCHK(m, hipMemcpyPeerAsync(m->col->d_clust + c0, m->device, src->col->d_clust + c0, src->device,
                                     sizeof(int32_t) * (size_t)(c1 - c0), m->stream));
        }
        if (fmx_phase_mstep(m)) return phase_fail(m);
      }
    }
  }
  const size_t npairs = (size_t)K * (K + 1) / 2;
  int32_t st[3] = {0, 0, 0};
  for (int r = 0; r < n; ++r) {
    muxgl_handle* m = g->m[(size_t)r];
    const int64_t c0 = g->cb[(size_t)r], nc = g->cb[(size_t)r + 1] - c0;
    GCHK(m, hipSetDevice(m->device));
    if (out && nc)
      GCHK(m, hipMemcpyAsync(m->h_fcells, m->d_fcells, sizeof(muxgl_fmx_cell) * (size_t)nc, hipMemcpyDeviceToHost, m->stream));
    GCHK(m, hipStreamSynchronize(m->stream));
    collect_timing(m);
    if (out && nc) memcpy(out + c0, m->h_fcells, sizeof(muxgl_fmx_cell) * (size_t)nc);
    for (int i = 0; i < 3; ++i) st[i] += m->h_fstat[i];
    if (full_ll && nc)
      GCHK(m, hipMemcpy(full_ll + (size_t)c0 * npairs, m->d_fll, sizeof(double) * (size_t)nc * npairs, hipMemcpyDeviceToHost));
  }
#undef GCHK
  for (muxgl_handle* m : g->m) m->xs_keep = st[2] == 0;  // (fmx_exact.hip: the settled cells' table stays valid while nothing moves)
  if (nsingle) *nsingle = st[0];
  if (namb) *namb = st[1];
  if (nchanged) *nchanged = st[2];
  max_timing(g);
  return 0;
}

// gls[K][S][9], counts[K][S][3]: every member contributes the rows of its SNP range
int group_fmx_get_cluster_pileup(muxgl_handle* h, double* gls, int32_t* counts) {
  muxgl_group* g = h->group;
  if (g->K < 1) MUXGL_FAIL(h, "muxgl_fmx_get_cluster_pileup: no clusters set");
  for (int r = 0; r < g->n; ++r) {
    muxgl_handle* m = g->m[(size_t)r];
    muxgl_handle* c = m->col;
    const int64_t s0 = g->sb[(size_t)r], s1 = g->sb[(size_t)r + 1];
    HIPCHK(h, hipSetDevice(m->device));
    HIPCHK(h, hipStreamSynchronize(m->stream));
    if (s1 <= s0) continue;
    if (gls)
      HIPCHK(h, hipMemcpy2D(gls + (size_t)s0 * 9, sizeof(double) * 9 * (size_t)g->S, c->d_cgls + (size_t)s0 * 9,
                            sizeof(double) * 9 * (size_t)g->S, sizeof(double) * 9 * (size_t)(s1 - s0), (size_t)g->K,
                            hipMemcpyDeviceToHost));
    if (counts) {
      if (fmx_cluster_counts_device(c)) {
        h->err = c->err;
        return 1;
      }
      HIPCHK(h, hipStreamSynchronize(m->stream));
      HIPCHK(h, hipMemcpy2D(counts + (size_t)s0 * 3, sizeof(int32_t) * 3 * (size_t)g->S, c->d_ccnt + (size_t)s0 * 3,
                            sizeof(int32_t) * 3 * (size_t)g->S, sizeof(int32_t) * 3 * (size_t)(s1 - s0), (size_t)g->K,
                            hipMemcpyDeviceToHost));
    }
  }
  return 0;
}

int group_get_timing(const muxgl_handle* h, float* ms) {
  memcpy(ms, h->group->ms, sizeof(float) * MUXGL_T_COUNT);
  return 0;
}
