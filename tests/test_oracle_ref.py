"""CPU tests that PIN THE ORACLE TO THE REFERENCE'S OWN CODE, compiled in the build container.

oracle/_ref/libscdrop_ref.so (oracle/Makefile target `scdrop`) holds, compiled from /root/reference where the files
lie and with the reference's flags (CMakeLists.txt:4-5: C++14 -O3 -pthread):
  * sc_drop_seq.cpp:1-92,386-578 unmodified (logAdd, add_snp / add_cell / add_read with the real
    std::map<std::string UMI> containers, calculate_snp_droplet_pileup, calculate_droplet_clust_distance),
    PhredHelper.cpp, Error.cpp;
  * the hot loops of cmd_cram_demuxlet.cpp (:428-440, :590-622, :634-991) and cmd_cram_freemux2.cpp (:108-109, :114-159,
    :184-189, :192-262, :277-288, :350-370, :373-605) as verbatim line ranges inside wrapper functions that only declare
    the locals those lines name and copy their variables out (oracle/ref_hot.cpp.in).
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


# ------------------------------------------------- the synthetic generator's packed pileups, as the bench feeds them
def test_packed_synthetic_pileup_through_the_reference():
    """a slice of BASELINE configs[1] (10 k x 16 x 50 k shape) handed to the reference in packed order"""
    p = synth.make_config(1, scale=0.004)    # 40 cells, ~38 k entries
    r = rb.RefScl.from_packed(p)
    want, _, want_ll = r.demux((0.0, 0.5), full_ll=True)
    got, got_ll = ob.demux(p, (0.0, 0.5), full_ll=True)
    assert same_records(got, want) == []
    assert np.array_equal(got_ll, want_ll)
