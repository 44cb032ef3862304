"""ctypes binding of libmuxgl.so (include/muxgl.h).

This is plumbing: it loads the in-tree shared library, mirrors the C structs as numpy dtypes and checks return
codes.  There is no Python or CPU implementation of the hot path behind it -- if the library is missing, or no HIP
device is usable, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MUXGL_LIB") or os.path.join(HERE, "lib", "libmuxgl.so")  # (MUXGL_LIB: kernel experiments)

MAX_ALPHA = 16
READ_OTHER = 0xFF
SNG, DBL, AMB = 0, 1, 2
TYPE_NAMES = {SNG: "SNG", DBL: "DBL", AMB: "AMB"}

# timing slots of muxgl_get_timing
T_DEMUX_REDUCE, T_DEMUX_SWEEP, T_DEMUX_CALL, T_DEMUX_D2H = 0, 1, 2, 3
FLAG_FORCE_TILE_SWEEP = 1
FLAG_FORCE_ROW_KERNEL = 2
FLAG_FORCE_WAVE_KERNEL = 4
FLAG_FORCE_BATCHED_GREEDY = 8
FLAG_DEMUX_ONLY = 16
FLAG_ASYNC_PHASES = 32
FLAG_NO_LINEAR_ENTRIES = 64
FLAG_NO_PIVOT_SUMS = 256
FLAG_SPLIT_GENERAL_SWEEP = 512
FLAG_GROUP_PROBE_SELF = 1024
MAX_DEVICES = 16
XCHG_PAD = 64
T_FMX_ENTRY, T_FMX_GP, T_FMX_ESTEP, T_FMX_CALL, T_FMX_MSTEP = 4, 5, 6, 7, 8
T_FMXOLD_PAIR, T_FMXOLD_VOTE = 9, 10
T_FMX_ESTEP_SWEEP = 11
T_COUNT = 16
BUF_CGP, BUF_CLUST, BUF_CELLS, BUF_STAT = 0, 1, 2, 3

DEMUX_CELL = np.dtype(
    [(n, np.int32) for n in ("valid", "nsnps", "type", "next_type", "sBest", "sNext", "dBest1", "dBest2", "dBestA",
                             "dNext1", "dNext2", "dNextA", "jBest", "kBest", "aBest", "jNext", "kNext", "aNext")]
    + [(n, np.float64) for n in ("sngBestLLK", "sngNextLLK", "dblBestLLK", "dblNextLLK", "sumLLK", "sngLLK",
                                 "bestLLK", "nextLLK", "bestPP", "sngPP", "sngOnlyPP")],
    align=True,
)
CELL_DEEP_SNG, CELL_DEEP_DBL = 2, 4   # bits of DEMUX_CELL["valid"] next to bit 0 (include/muxgl.h)
FMX_CELL = np.dtype(
    [(n, np.int32) for n in ("type", "clust", "jBest", "kBest", "jNext", "kNext", "sBest", "sNext", "dBest1",
                             "dBest2", "dNext1", "dNext2")]
    + [(n, np.float64) for n in ("bestLLK", "nextLLK", "sngBestLLK", "sngNextLLK", "dblBestLLK", "dblNextLLK",
                                 "bestPP", "sngPP", "sngOnlyPP", "sumLLK", "sngThirdLLK", "dblThirdLLK")],
    align=True,
)
DROPD = np.dtype([("nsnps", np.int32), ("nread1", np.int32), ("nread2", np.int32), ("_pad", np.int32),
                  ("llk0", np.float64), ("llk2", np.float64)], align=True)
assert DROPD.itemsize == 32
assert DEMUX_CELL.itemsize == 18 * 4 + 11 * 8
assert FMX_CELL.itemsize == 12 * 4 + 12 * 8


class _Config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device_id", C.c_int32), ("flags", C.c_int32), ("n_devices", C.c_int32),
                ("device_ids", C.c_int32 * MAX_DEVICES)]


class _DemuxParams(C.Structure):
    _fields_ = [("n_alpha", C.c_int32), ("_pad", C.c_int32), ("alpha", C.c_double * MAX_ALPHA),
                ("doublet_prior", C.c_double)]


class _FmxParams(C.Structure):
    _fields_ = [("doublet_prior", C.c_double), ("geno_error", C.c_double)]


# every symbol include/muxgl.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "muxgl_version": (C.c_int, []),
    "muxgl_group_peer_stats": (C.c_int, [_VP, _VP]),
    "muxgl_create": (C.c_int, [C.POINTER(_Config), C.POINTER(_VP)]),
    "muxgl_destroy": (None, [_VP]),
    "muxgl_last_error": (C.c_char_p, [_VP]),
    "muxgl_set_pileup": (C.c_int, [_VP, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _VP, _VP, _VP, _VP]),
    "muxgl_demux_set_gp": (C.c_int, [_VP, C.c_int32, _VP, _VP]),
    "muxgl_demux_run": (C.c_int, [_VP, C.POINTER(_DemuxParams), _VP, _VP]),
    "muxgl_demux_results": (_VP, [_VP]),
    "muxgl_demux_exact_calls": (C.c_int, [C.c_int64, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP,
                                          C.POINTER(_DemuxParams), _VP, C.c_int32, _VP]),
    "muxgl_demux_get_entry_pg": (C.c_int, [_VP, _VP]),
    "muxgl_fmx_prepare": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "muxgl_fmx_get_entry_gls": (C.c_int, [_VP, _VP, _VP]),
    "muxgl_fmx_greedy_init": (C.c_int, [_VP, C.c_int32, _VP, C.c_double, C.c_double, _VP]),
    "muxgl_fmx_greedy_stats": (C.c_int, [_VP, _VP, _VP]),
    "muxgl_fmx_score_stats": (C.c_int, [_VP, _VP]),
    "muxgl_fmx_set_clusters": (C.c_int, [_VP, C.c_int32, _VP]),
    "muxgl_fmx_iterate": (C.c_int, [_VP, C.POINTER(_FmxParams), _VP, _VP, _VP, _VP, _VP]),
    "muxgl_fmx_get_cluster_pileup": (C.c_int, [_VP, _VP, _VP]),
    "muxgl_fmx_exact_stats": (C.c_int, [_VP, _VP, _VP, _VP]),
    "muxgl_fmx_exact_pending": (C.c_int, [_VP, _VP]),
    "muxgl_fmx_exact_hint": (C.c_int, [_VP, C.c_int32]),
    "muxgl_fmx_exact_snps": (C.c_int, [_VP, _VP, C.c_int64, _VP]),
    "muxgl_fmx_exact_rows": (C.c_int, [_VP, C.POINTER(_FmxParams), _VP, C.c_int64, _VP, _VP]),
    "muxgl_fmx_exact_finish": (C.c_int, [_VP, C.POINTER(_FmxParams), _VP, C.c_int64, _VP, _VP, _VP]),
    "muxgl_fmxold_pair_dist": (C.c_int, [_VP, C.c_double, _VP]),
    "muxgl_fmxold_get_signs": (C.c_int, [_VP, _VP]),
    "muxgl_fmxold_vote_init": (C.c_int, [_VP, C.c_int32, _VP, _VP, C.c_double, _VP, _VP]),
    "muxgl_fmxold_vote_refine": (C.c_int, [_VP, C.c_int32, _VP, _VP, C.c_int32, _VP, _VP, _VP]),
    "muxgl_fmx_set_column_slab": (C.c_int, [_VP, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _VP, _VP,
                                            _VP, _VP]),
    "muxgl_fmx_set_shard": (C.c_int, [_VP, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "muxgl_fmx_iter_gp": (C.c_int, [_VP, C.POINTER(_FmxParams)]),
    "muxgl_fmx_iter_estep": (C.c_int, [_VP, C.POINTER(_FmxParams)]),
    "muxgl_fmx_iter_mstep": (C.c_int, [_VP]),
    "muxgl_fmx_iter_fetch": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "muxgl_fmx_buffer": (C.c_int, [_VP, C.c_int32, C.POINTER(_VP), C.POINTER(C.c_int64)]),
    "muxgl_memcpy_dev": (C.c_int, [_VP, _VP, _VP, C.c_int64]),
    "muxgl_stream": (_VP, [_VP]),
    "muxgl_get_timing": (C.c_int, [_VP, _VP]),
    "muxgl_get_timing_sum": (C.c_int, [_VP, _VP, _VP, C.c_int32]),
}

_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen libmuxgl.so and attach prototypes.  Raises if the in-tree build is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build it with `python -m popscle_amd.build` (hipcc, gfx950); "
                           "there is no fallback implementation")
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


class MuxglError(RuntimeError):
    pass


EXACT_STATS = ("cells", "mirror_turned", "mirror_exact_ties", "near_ties", "deep", "changed")


def demux_exact_calls(p, alphas, cells, doublet_prior=0.5, nthreads=0):
    """muxgl_demux_exact_calls: the host pass (no device) that settles, in the reference's own arithmetic, every call of
    `cells` (records of demux_run over pileup p, modified in place) that rounding noise could decide: the order of a
    mirrored alpha = 0.5 pair (cmd_cram_demuxlet.cpp:738-746,883-906) and the near ties of the scans and thresholds
    (:827-837,925-988).  Returns a dict of the six counters (EXACT_STATS)."""
    lib = load_library()
    dp = _DemuxParams()
    dp.n_alpha = len(alphas)
    for i, a in enumerate(alphas):
        dp.alpha[i] = float(a)
    dp.doublet_prior = float(doublet_prior)
    cell_ptr = _arr(p.cell_ptr, np.int64, "cell_ptr")
    entry_snp = _arr(p.entry_snp, np.int32, "entry_snp")
    entry_rptr = _arr(p.entry_rptr, np.int64, "entry_rptr")
    reads = _arr(p.reads, np.uint8, "reads")
    gp = _arr(p.gp, np.float64, "gp")
    has_gp = _arr(p.has_gp, np.uint8, "has_gp")
    if cells.dtype != DEMUX_CELL or not cells.flags.c_contiguous or cells.shape != (cell_ptr.size - 1,):
        raise ValueError("cells must be the contiguous [C] record array demux_run returned")
    stats = np.zeros(len(EXACT_STATS), dtype=np.int64)
    rc = lib.muxgl_demux_exact_calls(cell_ptr.size - 1, gp.shape[1], _ptr(cell_ptr), _ptr(entry_snp), _ptr(entry_rptr),
                                     _ptr(reads), _ptr(gp), _ptr(has_gp), C.byref(dp), _ptr(cells),
                                     int(nthreads) if nthreads else (os.cpu_count() or 1), _ptr(stats))
    if rc != 0:
        raise MuxglError(f"muxgl_demux_exact_calls failed ({rc})")
    return dict(zip(EXACT_STATS, (int(x) for x in stats)))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_VP)


def _arr(a, dtype, name):
    a = np.ascontiguousarray(a, dtype=dtype)
    if a.dtype != np.dtype(dtype):
        raise TypeError(name)
    return a


class Engine:
    """One muxgl handle: one GPU (device_id an int) or a device group (a list of device ordinals; the same ordinal may
    appear more than once = virtual ranks on one GPU).  Methods mirror the C-ABI one to one."""

    def __init__(self, device_id=0, flags: int = 0):
        self.lib = load_library()
        self.h = _VP()
        cfg = _Config()
        cfg.struct_size = C.sizeof(_Config)
        cfg.flags = flags
        if isinstance(device_id, (list, tuple)):
            if not 1 <= len(device_id) <= MAX_DEVICES:
                raise ValueError("1..MAX_DEVICES device ordinals")
            cfg.n_devices = len(device_id)
            for i, d in enumerate(device_id):
                cfg.device_ids[i] = int(d)
            cfg.device_id = int(device_id[0])
        else:
            cfg.device_id = int(device_id)
        if self.lib.muxgl_create(C.byref(cfg), C.byref(self.h)) != 0:
            raise MuxglError(self.lib.muxgl_last_error(None).decode())
        self.C = self.S = self.nnz = self.R = 0
        self.C_total = 0   # slabbed handle: cells of the whole job
        self.cell_base = 0
        self.V = 0
        self.K = 0
        self.n_alpha = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.muxgl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise MuxglError(self.lib.muxgl_last_error(self.h).decode())

    # ---- pileup
    def set_pileup(self, S, cell_ptr, entry_snp, entry_rptr, reads):
        cell_ptr = _arr(cell_ptr, np.int64, "cell_ptr")
        entry_snp = _arr(entry_snp, np.int32, "entry_snp")
        entry_rptr = _arr(entry_rptr, np.int64, "entry_rptr")
        reads = _arr(reads, np.uint8, "reads")
        C_ = cell_ptr.size - 1
        nnz = entry_snp.size
        R = reads.size
        if entry_rptr.size != nnz + 1:
            raise ValueError("entry_rptr must have nnz+1 elements")
        self._check(self.lib.muxgl_set_pileup(self.h, C_, int(S), nnz, R, _ptr(cell_ptr), _ptr(entry_snp),
                                               _ptr(entry_rptr), _ptr(reads)))
        self.C, self.S, self.nnz, self.R = C_, int(S), nnz, R
        self.C_total, self.cell_base = C_, 0

    def fmx_set_column_slab(self, C_total, c0, s0, s1, cell_ptr, entry_snp, entry_rptr, reads):
        """attach the column slab (all C_total cells, entries with s0 <= SNP < s1) to a handle holding a row slab"""
        cell_ptr = _arr(cell_ptr, np.int64, "cell_ptr")
        entry_snp = _arr(entry_snp, np.int32, "entry_snp")
        entry_rptr = _arr(entry_rptr, np.int64, "entry_rptr")
        reads = _arr(reads, np.uint8, "reads")
        if cell_ptr.size != int(C_total) + 1 or entry_rptr.size != entry_snp.size + 1:
            raise ValueError("column slab: cell_ptr must have C_total+1 and entry_rptr nnz+1 elements")
        self._check(self.lib.muxgl_fmx_set_column_slab(self.h, int(C_total), int(c0), int(s0), int(s1), entry_snp.size,
                                                        reads.size, _ptr(cell_ptr), _ptr(entry_snp), _ptr(entry_rptr),
                                                        _ptr(reads)))
        self.C_total, self.cell_base = int(C_total), int(c0)

    def stream(self):
        """the hipStream_t (as an integer) the handle's kernels are enqueued on"""
        return self.lib.muxgl_stream(self.h)

    # ---- demuxlet
    def demux_set_gp(self, gp, has_gp):
        gp = _arr(gp, np.float64, "gp")
        has_gp = _arr(has_gp, np.uint8, "has_gp")
        if gp.ndim != 3 or gp.shape[0] != self.S or gp.shape[2] != 3 or has_gp.shape != (self.S,):
            raise ValueError("gp must be [S][V][3] and has_gp [S]")
        self._check(self.lib.muxgl_demux_set_gp(self.h, gp.shape[1], _ptr(gp), _ptr(has_gp)))
        self.V = gp.shape[1]

    def demux_run(self, alphas=(0.0, 0.5), doublet_prior=0.5, want_cells=True, want_full_ll=False):
        key = (tuple(alphas), float(doublet_prior))
        if getattr(self, "_dp_key", None) != key:  # (the struct of the last call is kept: a tight loop re-uses it)
            if len(alphas) > MAX_ALPHA:
                raise ValueError("too many alphas")
            p = _DemuxParams()
            p.n_alpha = len(alphas)
            for i, a in enumerate(alphas):
                p.alpha[i] = float(a)
            p.doublet_prior = float(doublet_prior)
            self._dp, self._dp_key = p, key
        p = self._dp
        out = np.zeros(self.C, dtype=DEMUX_CELL) if want_cells else None
        full = np.zeros((self.C, self.V, self.V, len(alphas)), dtype=np.float64) if want_full_ll else None
        self._check(self.lib.muxgl_demux_run(self.h, C.byref(p), _ptr(out), _ptr(full)))
        self.n_alpha = len(alphas)
        if want_full_ll:
            return out, full
        return out

    def demux_results_view(self):
        """zero-copy view of the pinned [C] record buffer of the last run"""
        p = self.lib.muxgl_demux_results(self.h)
        buf = (C.c_char * (self.C * DEMUX_CELL.itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=DEMUX_CELL, count=self.C)

    def demux_entry_pg(self):
        pg = np.zeros((self.nnz, self.n_alpha, 3, 3), dtype=np.float64)
        self._check(self.lib.muxgl_demux_get_entry_pg(self.h, _ptr(pg)))
        return pg

    # ---- freemuxlet
    def fmx_prepare(self, af):
        af = _arr(af, np.float64, "af")
        if af.shape != (self.S,):
            raise ValueError("af must be [S]")
        llk0 = np.zeros(self.C)
        llk2 = np.zeros(self.C)
        nsnps = np.zeros(self.C, dtype=np.int32)
        nreads = np.zeros(self.C, dtype=np.int32)
        self._check(self.lib.muxgl_fmx_prepare(self.h, _ptr(af), _ptr(llk0), _ptr(llk2), _ptr(nsnps), _ptr(nreads)))
        return llk0, llk2, nsnps, nreads

    def fmx_entry_gls(self):
        gls = np.zeros((self.nnz, 9))
        cnt = np.zeros((self.nnz, 3), dtype=np.int32)
        self._check(self.lib.muxgl_fmx_get_entry_gls(self.h, _ptr(gls), _ptr(cnt)))
        return gls, cnt

    def fmx_greedy_init(self, K, scores, frac_init_clust=1.0, singlet_score_thres=-1e300):
        scores = _arr(scores, np.float64, "scores")
        if scores.shape != (self.C,):
            raise ValueError("scores must be [C]")
        clust = np.zeros(self.C, dtype=np.int32)
        self._check(self.lib.muxgl_fmx_greedy_init(self.h, int(K), _ptr(scores), float(frac_init_clust),
                                                    float(singlet_score_thres), _ptr(clust)))
        return clust

    def fmx_score_stats(self):
        """cells whose llk0 / llk2 the last fmx_prepare recomputed in the reference's arithmetic (near-tied scores)"""
        a = C.c_int64()
        self._check(self.lib.muxgl_fmx_score_stats(self.h, C.byref(a)))
        return a.value

    def fmx_greedy_stats(self):
        """(near ties decided by the exact path, of those against the kernel's choice) of the last fmx_greedy_init"""
        a, b = C.c_int64(), C.c_int64()
        self._check(self.lib.muxgl_fmx_greedy_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def fmx_set_clusters(self, K, clust):
        clust = _arr(clust, np.int32, "clust")
        if clust.shape != (self.C_total,):
            raise ValueError("clust must be [C] (the whole job's cells for a slabbed handle)")
        self._check(self.lib.muxgl_fmx_set_clusters(self.h, int(K), _ptr(clust)))
        self.K = int(K)

    def fmx_iterate(self, doublet_prior=0.5, geno_error=0.1, want_cells=True, want_full_ll=False):
        p = _FmxParams(float(doublet_prior), float(geno_error))
        out = np.zeros(self.C, dtype=FMX_CELL) if want_cells else None
        full = np.zeros((self.C, self.K * (self.K + 1) // 2)) if want_full_ll else None
        ns, na, nc = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.muxgl_fmx_iterate(self.h, C.byref(p), _ptr(out), C.byref(ns), C.byref(na), C.byref(nc),
                                                _ptr(full)))
        stats = (ns.value, na.value, nc.value)
        if want_full_ll:
            return out, stats, full
        return out, stats

    # ---- freemuxlet-old: pairwise distance matrix and voting passes (cmd_cram_freemuxlet.cpp:176-343)
    def fmxold_pair_dist(self, bf_thres=5.41, want_full=False):
        full = np.zeros(self.C * (self.C - 1) // 2, dtype=DROPD) if want_full else None
        self._check(self.lib.muxgl_fmxold_pair_dist(self.h, float(bf_thres), _ptr(full)))
        return full

    def fmxold_signs(self):
        out = np.zeros((self.C, self.C), dtype=np.int8)
        self._check(self.lib.muxgl_fmxold_get_signs(self.h, _ptr(out)))
        return out

    def fmxold_vote_init(self, K, order, jitter, frac_init_clust=1.0):
        order = _arr(order, np.int32, "order")
        jitter = _arr(jitter, np.float64, "jitter")
        clust = np.zeros(self.C, dtype=np.int32)
        cc = np.zeros(int(K), dtype=np.int32)
        self._check(self.lib.muxgl_fmxold_vote_init(self.h, int(K), _ptr(order), _ptr(jitter), float(frac_init_clust),
                                                     _ptr(clust), _ptr(cc)))
        return clust, cc

    def fmxold_vote_refine(self, K, order, jitter, clust, keep_init_missing=False):
        order = _arr(order, np.int32, "order")
        jitter = _arr(jitter, np.float64, "jitter")
        clust = np.array(clust, dtype=np.int32)
        cc = np.zeros(int(K), dtype=np.int32)
        ch = C.c_int32()
        self._check(self.lib.muxgl_fmxold_vote_refine(self.h, int(K), _ptr(order), _ptr(jitter),
                                                       int(bool(keep_init_missing)), _ptr(clust), C.byref(ch), _ptr(cc)))
        return clust, ch.value, cc

    # ---- sharded EM phases (multi-GPU); the collectives between them belong to the caller (popscle_amd/freemuxlet.py)
    def fmx_set_shard(self, c0, c1, s0, s1):
        self._check(self.lib.muxgl_fmx_set_shard(self.h, int(c0), int(c1), int(s0), int(s1)))

    def fmx_iter_gp(self, doublet_prior=0.5, geno_error=0.1):
        p = _FmxParams(float(doublet_prior), float(geno_error))
        self._check(self.lib.muxgl_fmx_iter_gp(self.h, C.byref(p)))

    def fmx_iter_estep(self, doublet_prior=0.5, geno_error=0.1):
        p = _FmxParams(float(doublet_prior), float(geno_error))
        self._check(self.lib.muxgl_fmx_iter_estep(self.h, C.byref(p)))

    def fmx_iter_mstep(self):
        self._check(self.lib.muxgl_fmx_iter_mstep(self.h))

    def fmx_iter_fetch(self, want_full_ll=False, want_cells=True):
        out = np.empty(self.C, dtype=FMX_CELL) if want_cells else None
        full = np.zeros((self.C, self.K * (self.K + 1) // 2)) if want_full_ll else None
        ns, na, nc = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.muxgl_fmx_iter_fetch(self.h, _ptr(out), C.byref(ns), C.byref(na), C.byref(nc), _ptr(full)))
        stats = (ns.value, na.value, nc.value)
        return (out, stats, full) if want_full_ll else (out, stats)

    def fmx_buffer(self, which):
        """(device pointer, element count) of an internal exchange buffer: BUF_CGP f64, BUF_CLUST i32, BUF_CELLS
        records, BUF_STAT i32[4]"""
        ptr = _VP()
        n = C.c_int64()
        self._check(self.lib.muxgl_fmx_buffer(self.h, int(which), C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def memcpy_dev(self, dst_ptr, src_ptr, nbytes):
        self._check(self.lib.muxgl_memcpy_dev(self.h, _VP(dst_ptr), _VP(src_ptr), int(nbytes)))

    def group_peer_stats(self):
        """(member pairs walked, pairs with direct peer access, pairs left on the staged copy) of a device group"""
        out = np.zeros(3, dtype=np.int32)
        self._check(self.lib.muxgl_group_peer_stats(self.h, _ptr(out)))
        return tuple(int(x) for x in out)

    def fmx_exact_stats(self):
        """(near-tie cells settled by the exact path, calls it changed, near-tie cells left unresolved) since set_clusters"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.muxgl_fmx_exact_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- the exact path for the sharded phases (include/muxgl.h; freemuxlet.settle_near_ties drives it)
    def fmx_exact_pending(self):
        n = C.c_int64()
        self._check(self.lib.muxgl_fmx_exact_pending(self.h, C.byref(n)))
        return n.value

    def fmx_exact_hint(self, nchanged_jobwide):
        self._check(self.lib.muxgl_fmx_exact_hint(self.h, int(nchanged_jobwide)))

    def fmx_exact_snps(self):
        n = C.c_int64()
        self._check(self.lib.muxgl_fmx_exact_snps(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.int32)
        self._check(self.lib.muxgl_fmx_exact_snps(self.h, _ptr(out), n.value, C.byref(n)))
        return out

    def fmx_exact_rows(self, snps, doublet_prior=0.5, geno_error=0.1):
        snps = _arr(snps, np.int32, "snps")
        p = _FmxParams(float(doublet_prior), float(geno_error))
        rows = np.zeros((snps.size, self.K, 3))
        owned = np.zeros(snps.size, dtype=np.uint8)
        self._check(self.lib.muxgl_fmx_exact_rows(self.h, C.byref(p), _ptr(snps), snps.size, _ptr(rows), _ptr(owned)))
        return rows, owned.astype(bool)

    def fmx_exact_finish(self, snps, rows, doublet_prior=0.5, geno_error=0.1):
        snps = _arr(snps, np.int32, "snps")
        rows = _arr(rows, np.float64, "rows")
        p = _FmxParams(float(doublet_prior), float(geno_error))
        deltas = np.zeros(3, dtype=np.int64)
        re = C.c_int32()
        self._check(self.lib.muxgl_fmx_exact_finish(self.h, C.byref(p), _ptr(snps), snps.size, _ptr(rows), _ptr(deltas),
                                                    C.byref(re)))
        return deltas, bool(re.value)

    def fmx_cluster_pileup(self):
        gls = np.zeros((self.K, self.S, 9))
        cnt = np.zeros((self.K, self.S, 3), dtype=np.int32)
        self._check(self.lib.muxgl_fmx_get_cluster_pileup(self.h, _ptr(gls), _ptr(cnt)))
        return gls, cnt

    # ---- measurement
    def timing_sum(self, reset=False):
        """(kernel times summed over the run / iterate calls since the last reset, ms; number of those calls)"""
        ms = np.zeros(T_COUNT, dtype=np.float64)
        n = C.c_int64(0)
        self._check(self.lib.muxgl_get_timing_sum(self.h, _ptr(ms), C.byref(n), 1 if reset else 0))
        return ms, int(n.value)

    def timing(self, out=None):
        """kernel times of the last run / iterate call, ms (out: a float32[T_COUNT] array to fill instead of a new one)"""
        ms = np.zeros(T_COUNT, dtype=np.float32) if out is None else out
        self._check(self.lib.muxgl_get_timing(self.h, _ptr(ms)))
        return ms
