"""ctypes binding of oracle/_ref/libscdrop_ref.so: the REFERENCE'S OWN CODE compiled in the build container
(oracle/Makefile target `scdrop`): sc_drop_seq.cpp's containers and per-entry arithmetic, and the hot loops of
cmdCramDemuxlet / cmdCramFreemux2 as verbatim line ranges (oracle/ref_hot.cpp.in).  TEST INFRASTRUCTURE: the checker
of the oracle (tests/test_oracle_ref.py), never imported from popscle_amd/.

The library exists only where /root/reference was present at build time; it travels to the GPU box as a built .so
(oracle/_ref/ is git-ignored, not gpurun-ignored), its sources do not.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

import oracle_binding as ob
from popscle_amd import synth

REF_SCDROP_SO = os.path.join(ob.ORACLE_DIR, "_ref", "libscdrop_ref.so")
_VP = C.c_void_p
_lib = None


def available():
    return os.path.exists(REF_SCDROP_SO)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(REF_SCDROP_SO)
        _lib.scref_new.restype = _VP
        _lib.scref_logadd.restype = C.c_double
        _lib.scref_logadd.argtypes = [C.c_double, C.c_double]
        _lib.scref_add_bases.restype = C.c_int64
        _lib.scref_add_snps.restype = C.c_int32
        _lib.scref_add_cells.restype = C.c_int32
        _lib.scref_demux.restype = C.c_int32
        _lib.scref_freemux2.restype = C.c_int32
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_VP)


def logadd(a, b):
    return lib().scref_logadd(float(a), float(b))


def file_order_bases(p, raw_bq=None):
    """The base stream of the PLP file plpio.write_plp(p, raw_bq=...) writes: rows sorted by SNP then droplet, the
    bases of a row in the order p.reads lists them.  Returns (snp, cell, allele digit, raw quality) per base."""
    nreads = np.diff(p.entry_rptr)
    cell_of = np.repeat(np.arange(p.C), np.diff(p.cell_ptr))
    order = np.lexsort((cell_of, p.entry_snp))
    ridx = synth._ranges(p.entry_rptr[order], nreads[order])
    ent = np.repeat(order, nreads[order])
    rd = p.reads[ridx]
    al = np.where(rd == synth.READ_OTHER, 2, rd >> 7).astype(np.uint8)
    if raw_bq is None:
        bq = np.where(rd == synth.READ_OTHER, 20, rd & 0x7F).astype(np.int8)
    else:
        bq = np.asarray(raw_bq)[ridx].astype(np.int8)
    return (np.ascontiguousarray(p.entry_snp[ent], dtype=np.int32), np.ascontiguousarray(cell_of[ent], dtype=np.int32),
            al, bq)


class RefScl:
    """A sc_dropseq_lib_t of the reference, filled through its own add_snp / add_cell / add_read."""

    def __init__(self, C_, S, af, gp=None, has_gp=None, min_bq=13, cap_bq=20, names=None):
        self.h = _VP(lib().scref_new(C.c_int32(min_bq), C.c_int32(cap_bq)))
        self.C, self.S = int(C_), int(S)
        self.af = np.ascontiguousarray(af, dtype=np.float64)
        self.gp = None if gp is None else np.ascontiguousarray(gp, dtype=np.float64)  # kept alive: add_snp borrows rows
        self.has_gp = None if has_gp is None else np.ascontiguousarray(has_gp, dtype=np.uint8)
        self.V = 0 if self.gp is None else int(self.gp.shape[1])
        assert lib().scref_add_snps(self.h, C.c_int64(S), C.c_int32(self.V), _p(self.af), _p(self.gp),
                                    _p(self.has_gp)) == S - 1
        arr = None
        if names is not None:
            arr = (C.c_char_p * self.C)(*[n.encode() for n in names])
        assert lib().scref_add_cells(self.h, C.c_int64(self.C), arr) == self.C - 1
        self.numi = 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().scref_free(self.h)
            self.h = None

    def add_bases(self, snp, cell, allele_digit, rawbq, fixed_width=False):
        snp = np.ascontiguousarray(snp, dtype=np.int32)
        cell = np.ascontiguousarray(cell, dtype=np.int32)
        al = np.ascontiguousarray(allele_digit, dtype=np.uint8)
        bq = np.ascontiguousarray(rawbq, dtype=np.int8)
        n = lib().scref_add_bases(self.h, C.c_int64(snp.size), _p(snp), _p(cell), _p(al), _p(bq), C.c_int64(self.numi),
                                  C.c_int32(int(fixed_width)))
        assert n >= 0, "the reference's add_read raised"
        self.numi = int(n)
        return self.numi

    @classmethod
    def from_pileup(cls, p, raw_bq=None, min_bq=13, cap_bq=20, names=None):
        r = cls(p.C, p.S, p.af, p.gp, p.has_gp, min_bq, cap_bq, names)
        r.add_bases(*file_order_bases(p, raw_bq))
        return r

    @classmethod
    def from_packed(cls, p, names=None):
        """p's entries with their reads in exactly p's order (UMIs named at fixed width), no quality filter or cap: the
        reference's containers then iterate as the packed arrays do"""
        r = cls(p.C, p.S, p.af, p.gp, p.has_gp, min_bq=0, cap_bq=127, names=names)
        nreads = np.diff(p.entry_rptr)
        ent = np.repeat(np.arange(p.nnz), nreads)
        cell_of = np.repeat(np.arange(p.C), np.diff(p.cell_ptr))
        rd = p.reads
        al = np.where(rd == synth.READ_OTHER, 2, rd >> 7).astype(np.uint8)
        bq = np.where(rd == synth.READ_OTHER, 20, rd & 0x7F).astype(np.int8)
        r.add_bases(p.entry_snp[ent], cell_of[ent], al, bq, fixed_width=True)
        return r

    def export(self):
        """the reference's containers in the reference's iteration order, packed (a synth.Pileup) + the read counters"""
        c_, s_, nnz, r_ = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        lib().scref_sizes(self.h, C.byref(c_), C.byref(s_), C.byref(nnz), C.byref(r_))
        assert c_.value == self.C and s_.value == self.S
        cell_ptr = np.zeros(self.C + 1, dtype=np.int64)
        entry_snp = np.zeros(nnz.value, dtype=np.int32)
        entry_rptr = np.zeros(nnz.value + 1, dtype=np.int64)
        reads = np.zeros(r_.value, dtype=np.uint8)
        uniq = np.zeros(self.C, dtype=np.int32)
        totl = np.zeros(self.C, dtype=np.int32)
        lib().scref_export(self.h, _p(cell_ptr), _p(entry_snp), _p(entry_rptr), _p(reads), _p(uniq), _p(totl))
        p = synth.Pileup(self.C, self.S, cell_ptr, entry_snp, entry_rptr, reads, self.af, self.gp, self.has_gp, {})
        return p, uniq, totl

    def entry_pileup(self, nnz, alpha=0.5):
        out = np.zeros(nnz, dtype=ob.PLP)
        lib().scref_entry_pileup(self.h, C.c_double(alpha), _p(out))
        return out

    def clust_distance(self, snp, d, csnp, c):
        snp = np.ascontiguousarray(snp, dtype=np.int32)
        csnp = np.ascontiguousarray(csnp, dtype=np.int32)
        d = np.ascontiguousarray(d, dtype=ob.PLP)
        c = np.ascontiguousarray(c, dtype=ob.PLP)
        out = np.zeros(2)
        cnt = np.zeros(3, dtype=np.int32)
        lib().scref_clust_distance(self.h, C.c_int64(snp.size), _p(snp), _p(d), C.c_int64(csnp.size), _p(csnp), _p(c),
                                   _p(out), _p(cnt))
        return out[0], out[1], cnt

    def demux(self, alphas=(0.0, 0.5), doublet_prior=0.5, full_ll=False, min_total=0, min_umi=0, min_snp=0):
        al = np.ascontiguousarray(alphas, dtype=np.float64)
        out = np.zeros(self.C, dtype=ob.DEMUX_CELL)
        int_id = np.full(self.C, -1, dtype=np.int32)
        full = np.zeros((self.C, self.V, self.V, al.size)) if full_ll else None
        rc = lib().scref_demux(self.h, C.c_int32(self.V), C.c_int32(al.size), _p(al), C.c_double(doublet_prior),
                               C.c_int32(min_total), C.c_int32(min_umi), C.c_int32(min_snp), _p(out), _p(int_id),
                               _p(full))
        assert rc == 0
        return out, int_id, full

    def freemux2(self, K, doublet_prior=0.5, geno_error=0.1, frac_init_clust=1.0, singlet_score_thres=-1e300,
                 init_clust=None, full_ll=False, cluster_pileups=False):
        Cn, S = self.C, self.S
        res = dict(llk0=np.zeros(Cn), llk2=np.zeros(Cn), nsnps=np.zeros(Cn, dtype=np.int32),
                   nreads=np.zeros(Cn, dtype=np.int32), order=np.zeros(Cn, dtype=np.int32),
                   clust0=np.zeros(Cn, dtype=np.int32), cells=np.zeros((10, Cn), dtype=ob.FMX_CELL),
                   counters=np.zeros((10, 3), dtype=np.int32),
                   full_ll=np.zeros((10, Cn, K * (K + 1) // 2)) if full_ll else None,
                   cplp=np.zeros((10, K, S), dtype=ob.PLP) if cluster_pileups else None)
        ic = None if init_clust is None else np.ascontiguousarray(init_clust, dtype=np.int32)
        n_iter = C.c_int32()
        rc = lib().scref_freemux2(self.h, C.c_int32(K), C.c_double(doublet_prior), C.c_double(geno_error),
                                  C.c_double(frac_init_clust), C.c_double(singlet_score_thres), _p(ic),
                                  _p(res["llk0"]), _p(res["llk2"]), _p(res["nsnps"]), _p(res["nreads"]),
                                  _p(res["order"]), _p(res["clust0"]), _p(res["cells"]), _p(res["counters"]),
                                  _p(res["full_ll"]), _p(res["cplp"]), C.byref(n_iter))
        assert rc == 0
        res["n_iter"] = n_iter.value
        return res


def _freemuxlet_old(self, K, init_clust, doublet_prior=0.5, geno_error=0.0, full_ll=False, cluster_pileups=False):
    """cmdCramFreemuxlet's EM (cmd_cram_freemuxlet.cpp:107-161,359-370,432-653 compiled as verbatim ranges) from given
    initial clusters; always ten iterations (no early stop)."""
    Cn, S = self.C, self.S
    res = dict(llk0=np.zeros(Cn), llk2=np.zeros(Cn), nsnps=np.zeros(Cn, dtype=np.int32),
               nreads=np.zeros(Cn, dtype=np.int32), cplp0=np.zeros((K, S), dtype=ob.PLP),
               cells=np.zeros((10, Cn), dtype=ob.FMX_CELL), counters=np.zeros((10, 2), dtype=np.int32),
               full_ll=np.zeros((10, Cn, K * (K + 1) // 2)) if full_ll else None,
               cplp=np.zeros((10, K, S), dtype=ob.PLP) if cluster_pileups else None)
    ic = np.ascontiguousarray(init_clust, dtype=np.int32)
    assert ic.size == Cn
    n_iter = C.c_int32()
    f = lib().scref_freemuxlet_old
    f.restype = C.c_int32
    rc = f(self.h, C.c_int32(K), C.c_double(doublet_prior), C.c_double(geno_error), _p(ic), _p(res["llk0"]),
           _p(res["llk2"]), _p(res["nsnps"]), _p(res["nreads"]), _p(res["cplp0"]), _p(res["cells"]),
           _p(res["counters"]), _p(res["full_ll"]), _p(res["cplp"]), C.byref(n_iter))
    assert rc == 0
    res["n_iter"] = n_iter.value
    return res


RefScl.freemuxlet_old = _freemuxlet_old


# ---- the VCF -> GP arithmetic of bcf_filtered_reader.cpp / sc_drop_seq.cpp:287-315 (oracle/ref_vcf.cpp.in) -----------
PL_MISSING = -2**31  # bcf_int32_missing


def _sel(nsamples, sel_cols):
    sel = np.arange(nsamples, dtype=np.int32) if sel_cols is None else np.ascontiguousarray(sel_cols, dtype=np.int32)
    return sel


def vcf_pl(pls, nalleles=2, sel_cols=None, ploidies=None):
    """parse_likelihoods' EM.  pls [nsamples][ngenos] int32 (all VCF columns).  Returns (float32 gps of the SELECTED
    samples [nsel][ngenos], acs[nalleles], an)."""
    pls = np.ascontiguousarray(pls, dtype=np.int32)
    ns, ng = pls.shape
    sel = _sel(ns, sel_cols)
    pl8 = np.full(sel.size, 2, dtype=np.int8) if ploidies is None else np.ascontiguousarray(ploidies, dtype=np.int8)
    gps = np.zeros((ns, ng), dtype=np.float32)
    acs = np.zeros(nalleles)
    an = C.c_int32()
    f = lib().scref_vcf_pl
    f.restype = C.c_int32
    assert f(C.c_int32(nalleles), C.c_int32(ns), C.c_int32(sel.size), _p(sel), _p(pl8), _p(pls), _p(gps), _p(acs),
             C.byref(an)) == 0
    return gps[sel], acs, an.value


def vcf_gp(vals, nalleles=2, sel_cols=None, gt_error=0.0):
    """parse_posteriors' GP branch.  vals [nsamples][ngenos] float32 as bcf_get_format_float leaves them."""
    g = np.array(vals, dtype=np.float32)
    ns, ng = g.shape
    sel = _sel(ns, sel_cols)
    f = lib().scref_vcf_gp
    f.restype = C.c_int32
    assert f(C.c_int32(nalleles), C.c_int32(ns), C.c_int32(sel.size), _p(sel), C.c_double(gt_error), _p(g)) == 0
    return g[sel]


def vcf_gt(gidx, acs, an, nalleles=2, nsamples=None, sel_cols=None, ploidies=None, gt_error=0.0):
    """parse_posteriors' GT branch given get_genotype_at(i) of every selected sample and parse_genotypes' acs / an.
    Returns the whole [nsamples][ngenos] buffer (columns no selected sample wrote stay 0)."""
    gidx = np.ascontiguousarray(gidx, dtype=np.int32)
    ns = gidx.size if nsamples is None else nsamples
    sel = _sel(ns, sel_cols)
    assert sel.size == gidx.size
    ng = nalleles * (nalleles + 1) // 2
    pl8 = np.full(sel.size, 2, dtype=np.int8) if ploidies is None else np.ascontiguousarray(ploidies, dtype=np.int8)
    acs = np.ascontiguousarray(acs, dtype=np.float64)
    out = np.zeros((ns, ng), dtype=np.float32)
    f = lib().scref_vcf_gt
    f.restype = C.c_int32
    assert f(C.c_int32(nalleles), C.c_int32(ns), C.c_int32(sel.size), _p(sel), _p(pl8), _p(gidx), _p(acs),
             C.c_int32(an), C.c_double(gt_error), _p(out)) == 0
    return out


def gp_row(float_gp, geno_error_offset=0.1, geno_error_coeff=0.0, r2=0.0):
    """load_from_plp's row construction, sc_drop_seq.cpp:287-315"""
    g = np.ascontiguousarray(float_gp, dtype=np.float32).reshape(-1)
    nv = g.size // 3
    out = np.zeros(nv * 3)
    f = lib().scref_gp_row
    f.restype = C.c_int32
    assert f(C.c_int32(nv), _p(g), C.c_double(geno_error_offset), C.c_double(geno_error_coeff), C.c_float(r2), _p(out)) == 0
    return out
