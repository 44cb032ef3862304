"""CPU tests that PIN THE ORACLE TO THE REFERENCE'S OWN CODE, compiled in the build container.

oracle/_ref/libscdrop_ref.so (oracle/Makefile target `scdrop`) holds, compiled from /root/reference where the files
lie and with the reference's flags (CMakeLists.txt:4-5: C++14 -O3 -pthread):
  * sc_drop_seq.cpp:1-92,386-578 unmodified (logAdd, add_snp / add_cell / add_read with the real
    std::map<std::string UMI> containers, calculate_snp_droplet_pileup, calculate_droplet_clust_distance),
    PhredHelper.cpp, Error.cpp;
  * the hot loops of cmd_cram_demuxlet.cpp (:428-440, :590-622, :634-991), cmd_cram_freemux2.cpp (:108-109, :114-159,
    :184-189, :192-262, :277-288, :350-370, :373-605) and cmd_cram_freemuxlet.cpp (:107-108, :113-161, :359-370, :432-453,
    :456-653) as verbatim line ranges inside wrapper functions that only declare the locals those lines name and copy
    their variables out (oracle/ref_hot.cpp.in);
  * the htslib-free arithmetic of the VCF -> genotype-posterior path: bcf_filtered_reader.cpp :262-324 (PL EM),
    :422-457 (GP branch), :376-409 (GT branch given genotype indices and allele counts) and sc_drop_seq.cpp:287-315 (the
    double row handed to add_snp), the same way (oracle/ref_vcf.cpp.in) -- what tests/pyplp.py and the product's loader
    (popscle_amd/host/vcf.hpp, through `popscle-amd dump-plp`) are held to.
Every comparison below is BIT FOR BIT (np.array_equal on doubles / raw bytes of the records): the oracle
(oracle/muxgl_oracle.c) restates the same operations in the same order, and -ffp-contract=off / no FMA on x86-64 makes
both sides plain IEEE double arithmetic with glibc's log / exp.

Without the library (a checkout where /root/reference was absent at build time) the tests skip; the committed
tests/golden/*.npz -- generated from this library by tests/golden/make_golden.py -- carry the pin to such machines.
"""
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import ref_binding as rb
from popscle_amd import plpio, synth

pytestmark = pytest.mark.skipif(not rb.available(), reason="oracle/_ref/libscdrop_ref.so not built (needs /root/reference)")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "popscle_amd", "bin", "popscle-amd")


def same_records(a, b):
    """two structured arrays, every field bit for bit (NaN == NaN); returns the names of differing fields"""
    bad = []
    for n in a.dtype.names:
        if n.startswith("_"):
            continue
        x, y = a[n], b[n]
        if x.dtype.kind == "f":
            ok = np.array_equal(x.view(np.int64), y.view(np.int64)) or np.array_equal(x, y, equal_nan=True)
        else:
            ok = np.array_equal(x, y)
        if not ok:
            bad.append(n)
    return bad


def deep_pileup(C, S, V, seed, with_gp=True, **kw):
    """a pileup with shallow AND deep entries (up to hundreds of reads: the hex UMI order matters, clamps fire),
    allele "2" bases and raw qualities 0 ... 60 (below min-BQ, Q < 2, above the cap)"""
    kw.setdefault("mean_entries", 60)
    kw.setdefault("min_entries", 3)
    p = synth.make_pileup(C, S, V, seed=seed, reads_lambda=1.2, other=0.03, with_gp=with_gp, **kw)
    rng = np.random.default_rng([seed, 9])
    # make every 50th entry deep by repeating its reads
    nreads = np.diff(p.entry_rptr).copy()
    deep = np.arange(p.nnz) % 50 == 7
    nreads[deep] = rng.integers(17, 300, deep.sum())
    rptr = np.zeros(p.nnz + 1, dtype=np.int64)
    np.cumsum(nreads, out=rptr[1:])
    R = int(rptr[-1])
    ent = np.repeat(np.arange(p.nnz), nreads)
    # reads of a deep entry: mostly one allele (concordant: clamps fire), a few of the other
    major = rng.integers(0, 2, p.nnz)[ent]
    al = np.where(rng.random(R) < 0.9, major, 1 - major).astype(np.uint8)
    reads = ((al << 7) | 20).astype(np.uint8)
    reads[rng.random(R) < 0.03] = synth.READ_OTHER
    q = synth.Pileup(p.C, p.S, p.cell_ptr, p.entry_snp, rptr, reads, p.af, p.gp, p.has_gp, {})
    raw = rng.integers(0, 61, R).astype(np.uint8)
    return q, raw


# ------------------------------------------------------------------------------------------------ logAdd
def test_logadd_is_the_references():
    rng = np.random.default_rng(11)
    a = np.concatenate([rng.uniform(-800, 0, 20000), [-1e-300, -1e300, 0.0, -745.2, -3.0]])
    b = np.concatenate([rng.uniform(-800, 0, 20000), [-108.5, -5.0, -745.0, -1e-300, -3.0]])
    for x, y in zip(a, b):
        assert ob.logadd(x, y) == rb.logadd(x, y)
        assert ob.logadd(y, x) == rb.logadd(y, x)


# --------------------------------------------------------------------------------- a2: containers, read order
def test_read_order_of_the_loader_rule_vs_reference_containers():
    """load_from_plp's rule (filter, cap, UMI = "%x" of a global counter, add_read) run through the reference's
    containers; the packed form must equal what tests/pyplp.py's restatement of the rule yields -- i.e. the order the
    product's loader is held to in tests/test_host_loader.py."""
    p, raw = deep_pileup(150, 900, 3, seed=21)
    r = rb.RefScl.from_pileup(p, raw_bq=raw, min_bq=13, cap_bq=20)
    q, uniq, totl = r.export()
    assert q.R >= 10_000 and q.nnz > 3000
    # independent packing of the same rule in Python (dict of hex strings, sorted)
    snp, cell, al, bq = rb.file_order_bases(p, raw)
    per = {}
    numi = 0
    for s, c, a, b in zip(snp.tolist(), cell.tolist(), al.tolist(), bq.tolist()):
        if b >= 13:
            b = min(b, 20)
            per.setdefault((c, s), {})["%x" % numi] = b if a == 0 else (0x80 | b) if a == 1 else 0xFF
            numi += 1
    keys = sorted(per)
    want_reads = [per[k][u] for k in keys for u in sorted(per[k])]
    assert np.array_equal(q.entry_snp, np.array([k[1] for k in keys], dtype=np.int32))
    assert np.array_equal(q.reads, np.array(want_reads, dtype=np.uint8))
    assert np.array_equal(np.diff(q.entry_rptr), np.array([len(per[k]) for k in keys]))
    assert np.array_equal(uniq, np.bincount([k[0] for k in keys for _ in per[k]], minlength=p.C))
    # some entry must actually be reordered by the string order ("10" < "9")
    assert any(sorted(per[k]) != sorted(per[k], key=lambda u: int(u, 16)) for k in keys)


def test_product_loader_packs_in_the_reference_containers_order(tmp_path):
    """the C++ loader of the front end (popscle-amd dump-plp, no GPU) on files of the real format vs the reference's
    own containers filled from the same rows"""
    if not os.path.exists(BIN):
        from popscle_amd.build import build_lib

        build_lib()
        subprocess.run(["make", "-C", os.path.join(ROOT, "popscle_amd", "host")], check=True)
    p, raw = deep_pileup(40, 300, 3, seed=23)
    prefix = str(tmp_path / "plp")
    bcs = plpio.write_plp(prefix, p, raw_bq=raw, seed=23)
    out = str(tmp_path / "d.bin")
    run = subprocess.run([BIN, "dump-plp", "--plp", prefix, "--out", out], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    got = plpio.read_dump(out)
    r = rb.RefScl.from_pileup(p, raw_bq=raw, min_bq=13, cap_bq=20, names=bcs)
    q, uniq, totl = r.export()
    for k, want in (("cell_ptr", q.cell_ptr), ("entry_snp", q.entry_snp), ("entry_rptr", q.entry_rptr),
                    ("reads", q.reads), ("cell_uniq_reads", uniq)):
        assert np.array_equal(got[k], want), k


def test_from_packed_keeps_the_packed_order():
    p, _ = deep_pileup(60, 400, 3, seed=25)
    q, _, _ = rb.RefScl.from_packed(p).export()
    keep = np.diff(p.entry_rptr) > 0   # an entry without reads does not exist in the reference's containers
    assert np.array_equal(q.entry_snp, p.entry_snp[keep])
    assert np.array_equal(q.reads, p.reads)


# ------------------------------------------------------------------------------------ b1: entry pileups
@pytest.mark.parametrize("min_bq,cap_bq", [(13, 20), (0, 60), (2, 40)])
def test_entry_pileup_is_the_references(min_bq, cap_bq):
    p, raw = deep_pileup(600, 4000, 3, seed=31, with_gp=False, mean_entries=200)
    r = rb.RefScl.from_pileup(p, raw_bq=raw, min_bq=min_bq, cap_bq=cap_bq)
    q, _, _ = r.export()
    assert q.nnz >= 100_000
    want = r.entry_pileup(q.nnz)
    got = ob.fmx_entry_pileup(q)
    assert got.tobytes() == want.tobytes()
    assert (np.diff(q.entry_rptr) > 100).sum() > 500                 # deep entries
    assert (want["gls"] < 1.0000001e-6).sum() > 1000                 # the clamp fired (1e-6, then renormalised)
    if min_bq == 0:
        assert ((q.reads != 0xFF) & ((q.reads & 0x7F) < 2)).sum() > 1000  # Q < 2: phred2Err = 0.75
    assert (q.reads == 0xFF).sum() > 1000                            # allele "2": counted, not multiplied in


# ------------------------------------------------------------------- b4: droplet-to-cluster distance
def test_cluster_distance_is_the_references():
    rng = np.random.default_rng(41)
    p, raw = deep_pileup(300, 500, 3, seed=41, with_gp=False, mean_entries=120)
    r = rb.RefScl.from_pileup(p, raw_bq=raw)
    q, _, _ = r.export()
    e = ob.fmx_entry_pileup(q)
    # cluster states: clamped merge chains of random entries per marker, present at ~70 % of the markers
    present_snp = rng.random(q.S) < 0.7
    by_snp = [[] for _ in range(q.S)]
    for i in rng.permutation(q.nnz)[: q.nnz // 2]:
        by_snp[q.entry_snp[i]].append(i)
    ptr = np.zeros(q.S + 1, dtype=np.int64)
    np.cumsum([len(x) for x in by_snp], out=ptr[1:])
    elems = e[np.array([i for x in by_snp for i in x], dtype=np.int64)]
    state = ob.plp_merge_chains(ptr, elems)
    csnp = np.nonzero(present_snp)[0].astype(np.int32)
    ndiff = 0
    for c in range(q.C):
        e0, e1 = q.cell_ptr[c], q.cell_ptr[c + 1]
        snps = q.entry_snp[e0:e1]
        l0, l2, cnt = r.clust_distance(snps, e[e0:e1], csnp, state[csnp])
        o0, o2, ocnt = ob.fmx_clust_distance(e[e0:e1], state[snps], present_snp[snps], q.af[snps])
        assert (l0, l2) == (o0, o2) and np.array_equal(cnt, ocnt)
        ndiff += int(cnt[0] > 0)
    assert ndiff > 250


# -------------------------------------------------------------------------- demuxlet: the whole droplet loop
DEMUX_CASES = [
    # C, S, V, alphas, kw
    (200, 1500, 4, (0.0, 0.5), dict(missing_gp_frac=0.05)),
    (120, 1500, 16, (0.0, 0.5), dict()),
    (120, 1200, 5, (0.0, 0.1, 0.2, 0.3, 0.4, 0.5), dict(missing_gp_frac=0.02)),
    (80, 800, 2, (0.0, 0.25), dict()),
    (60, 800, 3, (0.0, 0.5, 0.25, 0.5), dict()),          # a repeated 0.5 and an unsorted grid
    (40, 600, 1, (0.0, 0.5), dict()),                      # one sample: nv - 1 = 0 in the priors
    (40, 600, 3, (0.0,), dict()),                          # nAlpha = 1: division by nAlpha - 1 = 0
    (30, 500, 33, (0.0, 0.5), dict()),
]


@pytest.mark.parametrize("C,S,V,alphas,kw", DEMUX_CASES)
def test_demuxlet_loop_is_the_references(C, S, V, alphas, kw):
    p, raw = deep_pileup(C, S, V, seed=50 + V, **kw)
    # two cells without any kept base (the reference emits no row, :653) -- all their qualities below min-BQ
    for c in (3, C - 1):
        raw[p.entry_rptr[p.cell_ptr[c]]:p.entry_rptr[p.cell_ptr[c + 1]]] = 5
    names = plpio.barcodes(C, seed=V)    # shuffled barcodes: INT_ID is the rank in std::map<std::string> order
    r = rb.RefScl.from_pileup(p, raw_bq=raw, names=names)
    q, uniq, _ = r.export()
    want, int_id, want_ll = r.demux(alphas, doublet_prior=0.5, full_ll=True)
    got, got_ll = ob.demux(q, alphas, doublet_prior=0.5, full_ll=True)
    assert want["valid"].sum() == C - 2 and not want["valid"][3]
    assert same_records(got, want) == []
    assert np.array_equal(got_ll, want_ll, equal_nan=True)
    # INT_ID (cmd_cram_demuxlet.cpp:636-641,994): position in barcode-sorted order, counted over skipped cells too
    rank = np.argsort(np.argsort(np.array(names)))
    assert np.array_equal(int_id[want["valid"] == 1], rank[want["valid"] == 1])
    if len(alphas) > 1 and V > 1:
        assert len(set(want["type"][want["valid"] == 1].tolist())) >= 2


def test_demuxlet_loop_filters_and_prior():
    p, raw = deep_pileup(100, 800, 4, seed=61)
    r = rb.RefScl.from_pileup(p, raw_bq=raw)
    q, uniq, totl = r.export()
    nsnp = np.diff(q.cell_ptr)
    want, _, _ = r.demux((0.0, 0.5), doublet_prior=0.2, min_total=int(np.median(totl)), min_umi=3,
                         min_snp=int(np.median(nsnp)))
    got = ob.demux(q, (0.0, 0.5), doublet_prior=0.2)
    keep = want["valid"] == 1
    # the reference's filter (cmd_cram_demuxlet.cpp:641) is the caller's in the build: compare the surviving cells
    assert 10 < keep.sum() < 90
    assert np.array_equal(keep, (totl >= int(np.median(totl))) & (uniq >= 3) & (nsnp >= int(np.median(nsnp))) & (nsnp > 0))
    assert same_records(got[keep], want[keep]) == []


# ------------------------------------------------------------------ freemuxlet: scores, sort, init, EM loop
def oracle_freemux2(q, K, doublet_prior=0.5, geno_error=0.1, frac=1.0, thres=-1e300, init_clust=None):
    """the oracle driven the way cmdCramFreemux2 runs: ten iterations at most, early stop on nchanged == 0"""
    e = ob.fmx_entry_pileup(q)
    llk0, llk2, ns, nr = ob.fmx_cell_scores(q, e)
    order = ob.fmx_sort(llk2 - llk0)
    if init_clust is None:
        clust0 = ob.fmx_greedy_init(q, e, K, llk2 - llk0, order, frac, thres)
    else:
        clust0 = np.where(np.asarray(init_clust) >= 0, init_clust, -1).astype(np.int32)
    cplp = ob.fmx_build_cluster_pileup(q, e, K, clust0)
    cells = ob.fmx_init_cells(clust0)
    iters = []
    for _ in range(10):
        nsng, namb, nch, full = ob.fmx_iterate(q, e, K, cplp, cells, doublet_prior, geno_error, full_ll=True)
        iters.append((cells.copy(), (nsng, namb, nch), full, cplp.copy()))
        if nch == 0:
            break
    return dict(e=e, llk0=llk0, llk2=llk2, nsnps=ns, nreads=nr, order=order, clust0=clust0, iters=iters)


FMX_CASES = [
    # C, S, K, kw of freemux2
    (300, 1500, 4, dict()),
    (200, 1500, 8, dict(geno_error=0.0)),
    (250, 1200, 3, dict(doublet_prior=0.1, frac=0.6)),
    (150, 1000, 16, dict()),
    (150, 1000, 2, dict(thres=-40.0)),
    (120, 900, 5, dict(init=True)),
]


@pytest.mark.parametrize("C,S,K,kw", FMX_CASES)
def test_freemux2_is_the_references(C, S, K, kw):
    kw = dict(kw)
    p, raw = deep_pileup(C, S, K, seed=70 + K, with_gp=False, mean_entries=90)
    r = rb.RefScl.from_pileup(p, raw_bq=raw)
    q, _, _ = r.export()
    init = None
    if kw.pop("init", False):
        rng = np.random.default_rng(K)
        init = rng.integers(-1, K, q.C).astype(np.int32)   # -1: droplets the --init-cluster table does not list
    want = r.freemux2(K, doublet_prior=kw.get("doublet_prior", 0.5), geno_error=kw.get("geno_error", 0.1),
                      frac_init_clust=kw.get("frac", 1.0), singlet_score_thres=kw.get("thres", -1e300),
                      init_clust=init, full_ll=True, cluster_pileups=True)
    got = oracle_freemux2(q, K, kw.get("doublet_prior", 0.5), kw.get("geno_error", 0.1), kw.get("frac", 1.0),
                          kw.get("thres", -1e300), init)
    assert got["e"].tobytes() == r.entry_pileup(q.nnz).tobytes()
    for k in ("llk0", "llk2", "nsnps", "nreads", "order", "clust0"):
        assert np.array_equal(got[k], want[k]), k
    if "frac" in kw or "thres" in kw:
        assert (want["clust0"] < 0).sum() > 0      # the skip rules (:222-223) left droplets unassigned
    assert len(got["iters"]) == want["n_iter"]
    for it, (cells, counters, full, cplp) in enumerate(got["iters"]):
        assert counters == tuple(want["counters"][it]), it
        assert same_records(cells, want["cells"][it]) == [], it
        assert np.array_equal(full, want["full_ll"][it]), it
        assert cplp.tobytes() == want["cplp"][it].tobytes(), it
    assert want["n_iter"] >= 2


def test_freemux2_runs_ten_iterations_without_convergence():
    """droplets too shallow to settle: the reference's loop ends at max_iter = 10 (cmd_cram_freemux2.cpp:373)"""
    p, raw = deep_pileup(400, 3000, 6, seed=91, with_gp=False, mean_entries=12, min_entries=2)
    r = rb.RefScl.from_pileup(p, raw_bq=raw)
    q, _, _ = r.export()
    want = r.freemux2(6, full_ll=False, cluster_pileups=False)
    got = oracle_freemux2(q, 6)
    assert len(got["iters"]) == want["n_iter"]
    for it, (cells, counters, _, _) in enumerate(got["iters"]):
        assert counters == tuple(want["counters"][it])
        assert same_records(cells, want["cells"][it]) == []
    types = set(want["cells"][want["n_iter"] - 1]["type"].tolist())
    assert {0, 2} <= types or {0, 1} <= types


# ------------------------------------------------------------------ freemuxlet-old: the EM loop of the file north_star names
def oracle_freemuxlet_old(q, K, init_clust, doublet_prior=0.5, geno_error=0.0):
    """the oracle driven the way cmdCramFreemuxlet runs its EM (and the way popscle-amd freemuxlet-old drives the C-ABI,
    popscle_amd/host/main.cpp): always ten iterations, geno_error only in the last one (:485,500), no early stop"""
    e = ob.fmx_entry_pileup(q)
    llk0, llk2, ns, nr = ob.fmx_cell_scores(q, e)
    clust0 = np.ascontiguousarray(init_clust, dtype=np.int32)
    cplp = ob.fmx_build_cluster_pileup(q, e, K, clust0)
    cplp0 = cplp.copy()
    cells = ob.fmx_init_cells(clust0)
    iters = []
    for it in range(10):
        nsng, namb, _, full = ob.fmx_iterate(q, e, K, cplp, cells, doublet_prior, geno_error if it == 9 else 0.0,
                                             full_ll=True)
        iters.append((cells.copy(), (nsng, namb), full, cplp.copy()))
    return dict(e=e, llk0=llk0, llk2=llk2, nsnps=ns, nreads=nr, cplp0=cplp0, iters=iters)


FMXOLD_FIELDS_NOT_COMPARED = {"clust"}   # cmdCramFreemuxlet never updates clusts in its loop; freemux2 does (:545,568)

FMXOLD_CASES = [
    # C, S, K, doublet_prior, geno_error, fraction of droplets without an initial cluster
    (300, 1500, 4, 0.5, 0.0, 0.0),
    (200, 1500, 8, 0.5, 0.1, 0.0),
    (250, 1200, 3, 0.1, 0.05, 0.3),
    (150, 1000, 16, 0.5, 0.1, 0.1),
    (150, 1000, 2, 0.5, 0.0, 0.0),
]


@pytest.mark.parametrize("C,S,K,dp,ge,unassigned", FMXOLD_CASES)
def test_freemuxlet_old_em_is_the_references(C, S, K, dp, ge, unassigned):
    """cmd_cram_freemuxlet.cpp:107-161,359-370,432-653 compiled as verbatim ranges (scref_freemuxlet_old) against the
    oracle's iteration driven with freemuxlet-old's loop control: bit for bit, every iteration."""
    p, raw = deep_pileup(C, S, K, seed=170 + K, with_gp=False, mean_entries=90)
    r = rb.RefScl.from_pileup(p, raw_bq=raw, min_bq=1, cap_bq=60)   # sc_drop_seq.h:181: freemuxlet-old's loader defaults
    q, _, _ = r.export()
    rng = np.random.default_rng(K)
    # a plausible start: the greedy clusters of freemux2 with some droplets moved and some left out
    e = ob.fmx_entry_pileup(q)
    llk0, llk2, _, _ = ob.fmx_cell_scores(q, e)
    init = ob.fmx_greedy_init(q, e, K, llk2 - llk0, ob.fmx_sort(llk2 - llk0))
    move = rng.random(q.C) < 0.15
    init[move] = rng.integers(0, K, int(move.sum()))
    init[rng.random(q.C) < unassigned] = -1
    want = r.freemuxlet_old(K, init, dp, ge, full_ll=True, cluster_pileups=True)
    got = oracle_freemuxlet_old(q, K, init, dp, ge)
    assert want["n_iter"] == 10
    assert got["e"].tobytes() == r.entry_pileup(q.nnz).tobytes()
    for k in ("llk0", "llk2", "nsnps", "nreads"):
        assert np.array_equal(got[k], want[k]), k
    assert got["cplp0"].tobytes() == want["cplp0"].tobytes()
    for it, (cells, counters, full, cplp) in enumerate(got["iters"]):
        assert counters == tuple(want["counters"][it]), it
        bad = [f for f in same_records(cells, want["cells"][it]) if f not in FMXOLD_FIELDS_NOT_COMPARED]
        assert bad == [], (it, bad)
        assert np.array_equal(want["cells"][it]["clust"], init)
        assert np.array_equal(full, want["full_ll"][it]), it
        assert cplp.tobytes() == want["cplp"][it].tobytes(), it
    if ge > 0:
        # the genotype error acts in iteration 10 only: iteration 9 -> 10 changes every LL although the clusters settled
        assert not np.array_equal(want["full_ll"][9], want["full_ll"][8])
    types = np.bincount(want["cells"][9]["type"], minlength=3)
    assert types[0] > 0 and types[1] + types[2] > 0


def test_freemuxlet_old_differs_from_freemux2_where_the_survey_says():
    """row c1: with geno_error > 0 the two commands' first iteration already differs (freemux2 mixes in every iteration,
    cmd_cram_freemux2.cpp:410-415; the old loop only when iter + 1 == max_iter), and the old loop runs on after
    nchanged == 0"""
    p, raw = deep_pileup(200, 1200, 4, seed=33, with_gp=False, mean_entries=90)
    r = rb.RefScl.from_pileup(p, raw_bq=raw)
    q, _, _ = r.export()
    e = ob.fmx_entry_pileup(q)
    llk0, llk2, _, _ = ob.fmx_cell_scores(q, e)
    init = ob.fmx_greedy_init(q, e, 4, llk2 - llk0, ob.fmx_sort(llk2 - llk0))
    new = r.freemux2(4, geno_error=0.1, init_clust=init, full_ll=True)
    old = r.freemuxlet_old(4, init, 0.5, 0.1, full_ll=True)
    assert new["n_iter"] < 10 and old["n_iter"] == 10
    assert not np.array_equal(new["full_ll"][0], old["full_ll"][0])
    new0 = r.freemux2(4, geno_error=0.0, init_clust=init, full_ll=True)
    old0 = r.freemuxlet_old(4, init, 0.5, 0.0, full_ll=True)
    assert np.array_equal(new0["full_ll"][0], old0["full_ll"][0])   # same arithmetic without the mixing


# ------------------------------------------------------------------ VCF -> genotype posteriors
def _rand_pls(rng, ns, nal, missing=0.1):
    ng = nal * (nal + 1) // 2
    pls = rng.integers(0, 120, (ns, ng)).astype(np.int32)
    pls[np.arange(ns), rng.integers(0, ng, ns)] = 0            # a called genotype has PL 0
    pls[rng.random(ns) < 0.1] = rng.integers(200, 400, ng)     # beyond the table's 255 (toProb clamps)
    miss = rng.random(ns) < missing
    pls[miss] = rb.PL_MISSING                                  # "." -> bcf_int32_missing -> toProb(255)
    return pls


@pytest.mark.parametrize("nal", [2, 3])
def test_pl_em_is_the_references(nal):
    """bcf_filtered_reader.cpp:262-324 (the 10-iteration EM of parse_likelihoods) vs tests/pyplp.pl_em: float outputs and
    allele counts bit for bit; all samples, a subset in arbitrary column order, haploid columns, missing PLs"""
    import pyplp
    rng = np.random.default_rng(nal)
    for trial in range(60):
        ns = int(rng.integers(1, 12))
        pls = _rand_pls(rng, ns, nal)
        sel = None
        if trial % 3 == 1:
            sel = rng.permutation(ns)[:max(1, ns // 2)].astype(np.int32)
        nsel = ns if sel is None else sel.size
        ploidy = None if trial % 4 else rng.integers(1, 3, nsel).astype(np.int8)
        want, wacs, wan = rb.vcf_pl(pls, nal, sel, ploidy)
        rows = pls if sel is None else pls[sel]
        got, gacs, gan = pyplp.pl_em(rows.tolist(), nal, None if ploidy is None else ploidy.tolist())
        assert got.dtype == np.float32 and want.dtype == np.float32
        assert np.array_equal(got.view(np.int32), want.view(np.int32)), trial
        assert gan == wan and np.array_equal(np.array(gacs), wacs), trial


@pytest.mark.parametrize("nal", [2, 3])
def test_gp_branch_is_the_references(nal):
    """bcf_filtered_reader.cpp:422-457 vs tests/pyplp.gp_normalise, incl. a missing value (NaN) in one sample, which the
    reference's `gt_error*gpSums[j]` term spreads to every sample of the record even at gt_error == 0"""
    import pyplp
    rng = np.random.default_rng(10 + nal)
    ng = nal * (nal + 1) // 2
    for trial in range(80):
        ns = int(rng.integers(1, 12))
        vals = rng.dirichlet([0.4] * ng, ns).astype(np.float32)
        if trial % 5 == 0:
            vals = np.round(vals, 3).astype(np.float32)        # as printed with three decimals: sums != 1
        if trial % 7 == 3:
            vals[rng.integers(0, ns)] = np.float32(np.nan)
        sel = None if trial % 3 else rng.permutation(ns)[:max(1, ns // 2)].astype(np.int32)
        gt_error = 0.0 if trial % 4 else 0.01
        want = rb.vcf_gp(vals, nal, sel, gt_error)
        got = pyplp.gp_normalise((vals if sel is None else vals[sel]).tolist(), nal, gt_error)
        assert np.array_equal(got.view(np.int32), want.view(np.int32)), trial
        if trial % 7 == 3 and (sel is None or np.isnan(vals[sel]).any()):
            assert np.isnan(want).all()


def test_gt_branch_is_the_references():
    """bcf_filtered_reader.cpp:385-409 given the genotype indices and allele counts vs tests/pyplp.gt_posteriors"""
    import pyplp
    rng = np.random.default_rng(5)
    for trial in range(80):
        nal = 2 if trial % 3 else 3
        ng = nal * (nal + 1) // 2
        ns = int(rng.integers(1, 12))
        gidx = rng.integers(-1, ng, ns).astype(np.int32)
        an = int(rng.integers(0, 2 * ns + 1))
        acs = rng.multinomial(an, [1.0 / nal] * nal).astype(np.float64)
        gt_error = 0.0 if trial % 4 else 0.02
        ploidy = None if trial % 5 else rng.integers(1, 3, ns).astype(np.int8)
        want = rb.vcf_gt(gidx, acs, an, nal, ploidies=ploidy, gt_error=gt_error)
        got = pyplp.gt_posteriors(gidx.tolist(), acs.tolist(), an, nal, gt_error, None if ploidy is None else ploidy.tolist())
        assert np.array_equal(got.view(np.int32), want.view(np.int32)), trial


def test_gp_row_mixing_is_the_references():
    """sc_drop_seq.cpp:287-315 (float posteriors -> the double row of add_snp: avgGP with 1e-10 pseudo-counts, error =
    offset + (1-offset)(1-R2)coeff clipped to [0, 0.999]) vs tests/pyplp.gp_row"""
    import pyplp
    rng = np.random.default_rng(6)
    for trial in range(100):
        nv = int(rng.integers(1, 20))
        g = rng.dirichlet([0.3, 0.3, 0.3], nv).astype(np.float32)
        if trial % 3 == 0:
            g = np.eye(3, dtype=np.float32)[rng.integers(0, 3, nv)]
        off = [0.1, 0.0, 0.05, 1.5, -0.2][trial % 5]
        coeff = [0.0, 0.5, 2.0][trial % 3]
        r2 = np.float32(rng.random())
        want = rb.gp_row(g, off, coeff, float(r2))
        got = pyplp.gp_row(g, off, coeff, float(r2))
        assert np.array_equal(got, want), trial


def _vcf_columns(path):
    """(sample ids, [(pos, FORMAT keys, per-sample field strings)]) of a small VCF, parsed independently of pyplp"""
    import gzip
    ids, recs = None, []
    with gzip.open(path, "rt") as f:
        for line in f:
            t = line.rstrip("\n").split("\t")
            if line.startswith("#CHROM"):
                ids = t[9:]
            elif not line.startswith("#"):
                recs.append((int(t[1]), t[8].split(":"), [c.split(":") for c in t[9:]], t[7]))
    return ids, recs


@pytest.mark.parametrize("field,subset", [("PL", False), ("PL", True), ("GP", False), ("GP", True), ("GT", False)])
def test_product_loader_vcf_fields_vs_reference_arithmetic(tmp_path, field, subset):
    """`popscle-amd dump-plp --vcf V --field PL|GP|GT` (popscle_amd/host/vcf.hpp) against the reference's own arithmetic:
    the text of each record is split here, the numbers go through scref_vcf_pl / scref_vcf_gp / scref_vcf_gt and
    scref_gp_row, and the loader's GP tensor must equal the result bit for bit.  (What stays unpinned: text -> number
    conversion and allele counting, which live in htslib.)"""
    if not os.path.exists(BIN):
        pytest.skip("popscle-amd not built")
    V = 6
    p = synth.make_pileup(8, 70, V, seed=21, mean_entries=20, min_entries=4)
    G = p.truth["G"].astype(np.int64)
    rng = np.random.default_rng(3)
    gp = rng.dirichlet([0.3, 0.3, 0.3], size=G.shape)
    pl = _rand_pls(rng, G.size, 2, missing=0.0).reshape(G.shape + (3,))
    pl = np.where(pl > 255, 255, pl)
    prefix = str(tmp_path / "plp")
    plpio.write_plp(prefix, p, seed=21)
    vcf = str(tmp_path / "g.vcf.gz")
    plpio.write_vcf(vcf, p, G, field=field, missing_frac=0.08, gp=gp, pl=pl)
    poisoned = None
    if field == "GP":
        # one sample of one record without a value: htslib hands out NaN bit patterns, and the reference's
        # `gt_error*gpSums[j]` term turns the whole record NaN (bcf_filtered_reader.cpp:444,455)
        import gzip
        lines = gzip.open(vcf, "rt").read().split("\n")
        poisoned = set()
        for k in [i for i, ln in enumerate(lines) if ln and not ln.startswith("#")][5::9]:
            t = lines[k].split("\t")
            poisoned.add(int(t[1]))
            t[9 + 4] = t[9 + 4].split(":")[0] + ":."
            lines[k] = "\t".join(t)
        with gzip.open(vcf, "wt") as f:
            f.write("\n".join(lines))
    extra = []
    if subset:
        extra = ["--sm", "S4", "--sm", "S1", "--sm", "S5"]       # numbered in sorted-ID order: S1, S4, S5
    out = str(tmp_path / "d.bin")
    r = subprocess.run([BIN, "dump-plp", "--plp", prefix, "--out", out, "--vcf", vcf, "--field", field,
                        "--geno-error-offset", "0.07", "--geno-error-coeff", "0.3", *extra], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = plpio.read_dump(out)
    ids, recs = _vcf_columns(vcf)
    sel = np.array([1, 4, 5] if subset else range(V), dtype=np.int32)
    assert got["sample_ids"] == [ids[i] for i in sel]
    by_pos = {rec[0]: rec for rec in recs}
    checked = seen_poisoned = 0
    for s in range(p.S):
        if not got["has_gp"][s]:
            continue
        _, keys, cols, info = by_pos[1000 + 10 * s]
        if field == "PL":
            fi = keys.index("PL")
            pls = np.array([[int(x) for x in c[fi].split(",")] for c in cols], dtype=np.int32)
            fgp, _, _ = rb.vcf_pl(pls, 2, sel)
        elif field == "GP":
            fi = keys.index("GP")
            vals = np.array([([np.float32(float(x)) for x in c[fi].split(",")] if c[fi] != "." else [np.nan] * 3)
                             for c in cols], dtype=np.float32)
            fgp = rb.vcf_gp(vals, 2, sel, 0.0)
        else:
            al = [[(-1 if a == "." else int(a)) for a in c[0].replace("|", "/").split("/")] for c in cols]
            acs, an = np.zeros(2), 0
            for i in sel:                              # parse_genotypes counts over the selected samples (:225-247)
                for a in al[i]:
                    if a >= 0:
                        acs[a] += 1
                        an += 1
            gidx = [(-1 if min(al[i]) < 0 else max(al[i]) * (max(al[i]) + 1) // 2 + min(al[i])) for i in sel]
            fgp = rb.vcf_gt(gidx, acs, an, 2, nsamples=V, sel_cols=sel)[sel]
        r2 = np.float32(float(info.split("R2=")[1]))
        want = rb.gp_row(fgp, 0.07, 0.3, float(r2)).reshape(sel.size, 3)
        assert np.array_equal(got["gp"][s], want, equal_nan=True), (field, s)
        if poisoned and 1000 + 10 * s in poisoned:
            assert np.isnan(want).all()
            seen_poisoned += 1
        checked += 1
    assert poisoned is None or seen_poisoned >= 2
    assert checked > 30


# ------------------------------------------------- the synthetic generator's packed pileups, as the bench feeds them
def test_packed_synthetic_pileup_through_the_reference():
    """a slice of BASELINE configs[1] (10 k x 16 x 50 k shape) handed to the reference in packed order"""
    p = synth.make_config(1, scale=0.004)    # 40 cells, ~38 k entries
    r = rb.RefScl.from_packed(p)
    want, _, want_ll = r.demux((0.0, 0.5), full_ll=True)
    got, got_ll = ob.demux(p, (0.0, 0.5), full_ll=True)
    assert same_records(got, want) == []
    assert np.array_equal(got_ll, want_ll)


# ------------------------------------------------------------------ the oracle on the GPU suite's unfriendly generator
@pytest.mark.parametrize("seed", range(30))
def test_oracle_is_the_reference_on_unfriendly_cases(seed):
    """tests/test_fuzz_gpu.py's generator (droplets of a few entries, identical samples, markers without genotypes, entries
    of ~60 reads or with reads of another allele only, qualities up to 127, hard calls, partial and handed-in starts): the
    oracle -- the checker of most GPU tests -- must be the reference bit for bit there too"""
    import test_fuzz_gpu as fz

    info, p = fz.demux_case(seed)
    if info["V"] <= 40:
        want, _, wll = rb.RefScl.from_packed(p).demux(info["alphas"], doublet_prior=info["dp"], full_ll=True)
        got, gll = ob.demux(p, info["alphas"], doublet_prior=info["dp"], full_ll=True)
        assert same_records(got, want) == [], info
        m = parity_mask(info["V"], info["alphas"])
        assert np.array_equal(gll[:, m], wll[:, m], equal_nan=True), info
    info, p = fz.fmx_case(seed)
    if info["K"] <= 33:
        K = info["K"]
        want = rb.RefScl.from_packed(p).freemux2(K, doublet_prior=info["dp"], geno_error=info["ge"],
                                                frac_init_clust=info["frac"], singlet_score_thres=info["thres"],
                                                init_clust=info["init"], full_ll=True, cluster_pileups=True)
        got = oracle_freemux2(p, K, info["dp"], info["ge"], info["frac"], info["thres"], info["init"])
        for k in ("llk0", "llk2", "nsnps", "nreads", "clust0"):
            assert np.array_equal(got[k], want[k]), (k, info)
        if info["init"] is None:
            assert np.array_equal(got["order"], want["order"]), info
        assert len(got["iters"]) == want["n_iter"], info
        for it, (cells, counters, full, cplp) in enumerate(got["iters"]):
            assert counters == tuple(want["counters"][it]), (it, info)
            assert same_records(cells, want["cells"][it]) == [], (it, info)
            assert np.array_equal(full, want["full_ll"][it], equal_nan=True), (it, info)
            assert cplp.tobytes() == want["cplp"][it].tobytes(), (it, info)


def parity_mask(V, alphas):
    """the llksAB slots the reference ever writes: (j, 0, 0) and (j, k != j, n >= 1)"""
    A = len(alphas)
    m = np.zeros((V, V, A), dtype=bool)
    m[:, 0, 0] = True
    off = ~np.eye(V, dtype=bool)
    for n in range(1, A):
        m[:, :, n] = off
    return m
