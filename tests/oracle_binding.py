"""ctypes binding of the CPU oracle (oracle/libmuxgl_oracle.so).  TEST INFRASTRUCTURE: importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only -- never from popscle_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libmuxgl_oracle.so")
REF_PHRED_SO = os.path.join(ORACLE_DIR, "_ref", "libphred_ref.so")
REF_MERGE_SO = os.path.join(ORACLE_DIR, "_ref", "libmerge_ref.so")

DEMUX_CELL = np.dtype(
    [(n, np.int32) for n in ("valid", "nsnps", "type", "next_type", "sBest", "sNext", "dBest1", "dBest2", "dBestA",
                             "dNext1", "dNext2", "dNextA", "jBest", "kBest", "aBest", "jNext", "kNext", "aNext")]
    + [(n, np.float64) for n in ("sngBestLLK", "sngNextLLK", "dblBestLLK", "dblNextLLK", "sumLLK", "sngLLK",
                                 "bestLLK", "nextLLK", "bestPP", "sngPP", "sngOnlyPP")],
    align=True,
)
FMX_CELL = np.dtype(
    [(n, np.int32) for n in ("type", "clust", "jBest", "kBest", "jNext", "kNext", "sBest", "sNext", "dBest1",
                             "dBest2", "dNext1", "dNext2")]
    + [(n, np.float64) for n in ("bestLLK", "nextLLK", "sngBestLLK", "sngNextLLK", "dblBestLLK", "dblNextLLK",
                                 "bestPP", "sngPP", "sngOnlyPP", "sumLLK")],
    align=True,
)
PLP = np.dtype([("nreads", np.int32), ("nref", np.int32), ("nalt", np.int32), ("_pad", np.int32),
                ("gls", np.float64, (9,))], align=True)
assert PLP.itemsize == 16 + 72

_VP = C.c_void_p
_lib = None


def build():
    """(re)build the oracle with gcc; also builds oracle/_ref when /root/reference is present"""
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        _lib = C.CDLL(ORACLE_SO)
        _lib.oracle_logadd.restype = C.c_double
        _lib.oracle_logadd.argtypes = [C.c_double, C.c_double]
        _lib.oracle_fmx_iterate.restype = C.c_int32
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_VP)


def phred_tables():
    err = np.zeros(256)
    mat = np.zeros(256)
    lib().oracle_phred_tables(_p(err), _p(mat))
    return err, mat


def ref_phred_tables():
    """phred2Err / phred2Mat of the reference's own PhredHelper.cpp (global object `phredConv`,
    members in declaration order PhredHelper.h:27-33: Err, Prob, Mat, Mat3, LogMat, LogMat3, HalfLogMat3)"""
    ref = C.CDLL(REF_PHRED_SO)
    arr = (C.c_double * (256 * 7)).in_dll(ref, "phredConv")
    a = np.frombuffer(arr, dtype=np.float64).copy()
    return a[0:256], a[512:768]


def logadd(a, b):
    return lib().oracle_logadd(float(a), float(b))


def demux_entry_pg(reads, alphas):
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    al = np.ascontiguousarray(alphas, dtype=np.float64)
    out = np.zeros(al.size * 9)
    lib().oracle_demux_entry_pg(_p(reads), C.c_int64(reads.size), C.c_int32(al.size), _p(al), _p(out))
    return out.reshape(al.size, 3, 3)


def demux(plp, alphas=(0.0, 0.5), doublet_prior=0.5, full_ll=False, nthreads=1):
    """oracle demuxlet over a popscle_amd.synth.Pileup-like object (needs gp/has_gp)"""
    al = np.ascontiguousarray(alphas, dtype=np.float64)
    V = plp.gp.shape[1]
    out = np.zeros(plp.C, dtype=DEMUX_CELL)
    full = np.zeros((plp.C, V, V, al.size)) if full_ll else None
    gp = np.ascontiguousarray(plp.gp, dtype=np.float64)
    hg = np.ascontiguousarray(plp.has_gp, dtype=np.uint8)
    rc = lib().oracle_demux(C.c_int64(plp.C), C.c_int64(plp.S), C.c_int32(V), _p(plp.cell_ptr), _p(plp.entry_snp),
                            _p(plp.entry_rptr), _p(plp.reads), _p(gp), _p(hg), C.c_int32(al.size), _p(al),
                            C.c_double(doublet_prior), _p(out), _p(full), C.c_int32(nthreads))
    assert rc == 0
    return (out, full) if full_ll else out


def fmx_entry_pileup(plp):
    out = np.zeros(plp.nnz, dtype=PLP)
    lib().oracle_fmx_entry_pileup(C.c_int64(plp.nnz), _p(plp.entry_rptr), _p(plp.reads), _p(out))
    return out


def plp_merge(dst, src):
    """dst, src: PLP scalars (arrays of shape (1,))"""
    lib().oracle_plp_merge(_p(dst), _p(src))


def plp_merge_chains(ptr, elems):
    """final states of chains of merges from default-constructed pileups (elements ptr[i]..ptr[i+1]-1 in order)"""
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    elems = np.ascontiguousarray(elems, dtype=PLP)
    out = np.zeros(ptr.size - 1, dtype=PLP)
    lib().oracle_plp_merge_chains(C.c_int64(ptr.size - 1), _p(ptr), _p(elems), _p(out))
    return out


_ref_merge = None


def ref_merge_lib():
    """oracle/_ref/libmerge_ref.so: the reference's own header-inline snp_droplet_pileup::merge (sc_drop_seq.h:77-101)
    and sc_drop_comp_t (:187-198) behind extern "C" wrappers (oracle/merge_ref.cpp)"""
    global _ref_merge
    if _ref_merge is None:
        _ref_merge = C.CDLL(REF_MERGE_SO)
        _ref_merge.merge_ref_sizeof_plp.restype = C.c_int
        _ref_merge.merge_ref_comp.restype = C.c_int
        assert _ref_merge.merge_ref_sizeof_plp() == PLP.itemsize
    return _ref_merge


def ref_plp_merge(dst, src):
    ref_merge_lib().merge_ref_merge(_p(dst), _p(src))


def ref_plp_merge_chains(ptr, elems):
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    elems = np.ascontiguousarray(elems, dtype=PLP)
    out = np.zeros(ptr.size - 1, dtype=PLP)
    ref_merge_lib().merge_ref_chains(C.c_int64(ptr.size - 1), _p(ptr), _p(elems), _p(out))
    return out


def ref_fmx_sort(scores):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    order = np.zeros(scores.size, dtype=np.int32)
    ref_merge_lib().merge_ref_sort(C.c_int64(scores.size), _p(scores), _p(order))
    return order


def ref_comp(scores, lhs, rhs):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    return bool(ref_merge_lib().merge_ref_comp(C.c_int64(scores.size), _p(scores), C.c_int32(lhs), C.c_int32(rhs)))


def fmx_cell_scores(plp, eplp):
    llk0 = np.zeros(plp.C)
    llk2 = np.zeros(plp.C)
    ns = np.zeros(plp.C, dtype=np.int32)
    nr = np.zeros(plp.C, dtype=np.int32)
    af = np.ascontiguousarray(plp.af, dtype=np.float64)
    lib().oracle_fmx_cell_scores(C.c_int64(plp.C), _p(plp.cell_ptr), _p(plp.entry_snp), _p(eplp), _p(af), _p(llk0),
                                 _p(llk2), _p(ns), _p(nr))
    return llk0, llk2, ns, nr


def fmx_sort(scores):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    order = np.zeros(scores.size, dtype=np.int32)
    lib().oracle_fmx_sort(C.c_int64(scores.size), _p(scores), _p(order))
    return order


def fmx_clust_distance(d, c, present, af):
    """oracle_fmx_clust_distance: (llk0, llk2, counts[3]) of a droplet's entries d against the aligned states c"""
    d = np.ascontiguousarray(d, dtype=PLP)
    c = np.ascontiguousarray(c, dtype=PLP)
    present = np.ascontiguousarray(present, dtype=np.uint8)
    af = np.ascontiguousarray(af, dtype=np.float64)
    out = np.zeros(2)
    cnt = np.zeros(3, dtype=np.int32)
    lib().oracle_fmx_clust_distance(C.c_int64(d.size), _p(d), _p(c), _p(present), _p(af), _p(out), _p(cnt))
    return out[0], out[1], cnt


def fmx_greedy_init(plp, eplp, K, scores, order, frac_init_clust=1.0, singlet_score_thres=-1e300):
    clust = np.zeros(plp.C, dtype=np.int32)
    af = np.ascontiguousarray(plp.af, dtype=np.float64)
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    lib().oracle_fmx_greedy_init(C.c_int64(plp.C), C.c_int64(plp.S), C.c_int32(K), _p(plp.cell_ptr),
                                 _p(plp.entry_snp), _p(eplp), _p(af), _p(scores), _p(order),
                                 C.c_double(frac_init_clust), C.c_double(singlet_score_thres), _p(clust))
    return clust


def fmx_greedy_init_scores(plp, eplp, K, scores, order, frac_init_clust=1.0, singlet_score_thres=-1e300):
    """(clust, step_scores[C][K]): the greedy loop and the K distances of every visited cell in visiting order (rows
    beyond the number of visited cells stay 0)"""
    clust = np.zeros(plp.C, dtype=np.int32)
    ss = np.zeros((plp.C, K))
    af = np.ascontiguousarray(plp.af, dtype=np.float64)
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    lib().oracle_fmx_greedy_init_scores(C.c_int64(plp.C), C.c_int64(plp.S), C.c_int32(K), _p(plp.cell_ptr),
                                        _p(plp.entry_snp), _p(eplp), _p(af), _p(scores), _p(order),
                                        C.c_double(frac_init_clust), C.c_double(singlet_score_thres), _p(clust), _p(ss))
    return clust, ss


def fmx_build_cluster_pileup(plp, eplp, K, clust):
    cplp = np.zeros((K, plp.S), dtype=PLP)
    clust = np.ascontiguousarray(clust, dtype=np.int32)
    lib().oracle_fmx_build_cluster_pileup(C.c_int64(plp.C), C.c_int64(plp.S), C.c_int32(K), _p(plp.cell_ptr),
                                          _p(plp.entry_snp), _p(eplp), _p(clust), _p(cplp))
    return cplp


def fmx_init_cells(clust):
    clust = np.ascontiguousarray(clust, dtype=np.int32)
    cells = np.zeros(clust.size, dtype=FMX_CELL)
    lib().oracle_fmx_init_cells(C.c_int64(clust.size), _p(clust), _p(cells))
    return cells


def fmx_iterate(plp, eplp, K, cplp, cells, doublet_prior=0.5, geno_error=0.1, full_ll=False, nthreads=1):
    """one EM iteration in place on cplp/cells; returns (nsingle, namb, nchanged[, full_ll])"""
    af = np.ascontiguousarray(plp.af, dtype=np.float64)
    ns, na = C.c_int32(), C.c_int32()
    full = np.zeros((plp.C, K * (K + 1) // 2)) if full_ll else None
    nch = lib().oracle_fmx_iterate(C.c_int64(plp.C), C.c_int64(plp.S), C.c_int32(K), _p(plp.cell_ptr),
                                   _p(plp.entry_snp), _p(eplp), _p(af), C.c_double(doublet_prior),
                                   C.c_double(geno_error), _p(cplp), _p(cells), C.byref(ns), C.byref(na), _p(full),
                                   C.c_int32(nthreads))
    if full_ll:
        return ns.value, na.value, nch, full
    return ns.value, na.value, nch


# ---- freemuxlet-old (cmd_cram_freemuxlet.cpp:176-343)
DROPD = np.dtype([("nsnps", np.int32), ("nread1", np.int32), ("nread2", np.int32), ("_pad", np.int32),
                  ("llk0", np.float64), ("llk2", np.float64)], align=True)
assert DROPD.itemsize == 32


def fmxold_pair_dist(plp, eplp):
    out = np.zeros(plp.C * (plp.C - 1) // 2, dtype=DROPD)
    af = np.ascontiguousarray(plp.af, dtype=np.float64)
    lib().oracle_fmxold_pair_dist(C.c_int64(plp.C), C.c_int64(plp.S), _p(plp.cell_ptr), _p(plp.entry_snp), _p(eplp),
                                  _p(af), _p(out))
    return out


def fmxold_vote_init(ncells, K, dd, order, jitter, bf_thres=5.41, frac_init_clust=1.0):
    order = np.ascontiguousarray(order, dtype=np.int32)
    jitter = np.ascontiguousarray(jitter, dtype=np.float64)
    clust = np.zeros(ncells, dtype=np.int32)
    cc = np.zeros(K, dtype=np.int32)
    lib().oracle_fmxold_vote_init(C.c_int64(ncells), C.c_int32(K), _p(dd), _p(order), _p(jitter), C.c_double(bf_thres),
                                  C.c_double(frac_init_clust), _p(clust), _p(cc))
    return clust, cc


def fmxold_vote_refine(ncells, K, dd, order, jitter, clust, bf_thres=5.41, keep_init_missing=False):
    order = np.ascontiguousarray(order, dtype=np.int32)
    jitter = np.ascontiguousarray(jitter, dtype=np.float64)
    clust = np.array(clust, dtype=np.int32)
    cc = np.zeros(K, dtype=np.int32)
    f = lib().oracle_fmxold_vote_refine
    f.restype = C.c_int32
    changed = f(C.c_int64(ncells), C.c_int32(K), _p(dd), _p(order), _p(jitter), C.c_double(bf_thres),
                C.c_int32(int(bool(keep_init_missing))), _p(clust), _p(cc))
    return clust, changed, cc
