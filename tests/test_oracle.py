"""CPU tests of the oracle (oracle/muxgl_oracle.c) against an independent pure-Python restatement (tests/pyref.py),
hand-derivable cases and the committed golden vectors.

Where the oracle is pinned to the REFERENCE ITSELF: the Phred tables, merge() and the sort comparator below
(oracle/_ref/libphred_ref.so, libmerge_ref.so), and -- round 5 -- everything else on the arithmetic path in
tests/test_oracle_ref.py (oracle/_ref/libscdrop_ref.so: the reference's sc_drop_seq.cpp and the hot loops of
cmd_cram_demuxlet.cpp / cmd_cram_freemux2.cpp compiled from /root/reference).  The golden vectors demux_*.npz and
fmx_k4*.npz under tests/golden are outputs of that library (tests/golden/make_golden.py), so
test_oracle_reproduces_golden_* holds the oracle to the reference's numbers on machines without /root/reference too.
"""
import math
import os

import numpy as np
import pytest

import oracle_binding as ob
import pyref
from popscle_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_phred_tables_match_reference_build():
    if not os.path.exists(ob.REF_PHRED_SO):
        pytest.skip("oracle/_ref/libphred_ref.so not built (needs /root/reference at build time)")
    err, mat = ob.phred_tables()
    rerr, rmat = ob.ref_phred_tables()
    assert np.array_equal(err, rerr)  # bit-for-bit
    assert np.array_equal(mat, rmat)
    assert err[0] == 0.75 and err[1] == 0.75 and err[20] == pytest.approx(0.01, rel=1e-15)


def test_phred_tables_match_pyref():
    err, mat = ob.phred_tables()
    assert np.array_equal(err, np.array(pyref.ERR))
    assert np.array_equal(mat, np.array(pyref.MAT))


def test_logadd():
    for a, b in [(-1e-300, -108.5), (-3.0, -3.0), (-1e300, -5.0), (0.0, -745.0), (-10.0, -2.0)]:
        assert ob.logadd(a, b) == pyref.logadd(a, b)
    assert ob.logadd(-2.0, -2.0) == pytest.approx(-2.0 + math.log(2.0), abs=1e-15)


@pytest.mark.parametrize("alphas", [(0.0, 0.5), (0.0, 0.1, 0.2, 0.3, 0.4, 0.5), (0.0,), (0.0, 0.25)])
def test_demux_entry_pg_vs_pyref(alphas):
    rng = np.random.default_rng(7)
    for nreads in [0, 1, 2, 5, 40]:
        reads = ((rng.integers(0, 2, nreads) << 7) | rng.integers(2, 41, nreads)).astype(np.uint8)
        if nreads > 2:
            reads[1] = 0xFF  # one "other" allele
        got = ob.demux_entry_pg(reads, alphas).ravel()
        want = np.array(pyref.demux_entry_pg(list(map(int, reads)), list(alphas)))
        assert np.array_equal(got, want)
        assert got.max() == 1.0  # after the final division the largest element is exactly 1


def test_demux_entry_pg_no_reads_is_uniform():
    pg = ob.demux_entry_pg(np.zeros(0, dtype=np.uint8), (0.0, 0.5))
    assert np.all(pg == 1.0)
    pg = ob.demux_entry_pg(np.array([0xFF, 0xFF], dtype=np.uint8), (0.0, 0.5))
    assert np.all(pg == 1.0)


def test_demux_entry_pg_alpha0_independent_of_m():
    """with alpha[0]==0 the mixing proportion does not depend on the second genotype (singlet slot)"""
    reads = np.array([(1 << 7) | 20, 20, (1 << 7) | 13], dtype=np.uint8)
    pg = ob.demux_entry_pg(reads, (0.0, 0.5))
    assert np.array_equal(pg[0, :, 0], pg[0, :, 1]) and np.array_equal(pg[0, :, 0], pg[0, :, 2])
    # alpha = 0.5 is symmetric in (l, m)
    assert np.array_equal(pg[1], pg[1].T)


def _cells_as_python(p, c):
    ents = []
    for e in range(p.cell_ptr[c], p.cell_ptr[c + 1]):
        ents.append((int(p.entry_snp[e]), [int(x) for x in p.reads[p.entry_rptr[e]:p.entry_rptr[e + 1]]]))
    return ents


@pytest.mark.parametrize("V,alphas", [(3, (0.0, 0.5)), (4, (0.0, 0.1, 0.2, 0.3, 0.4, 0.5)), (2, (0.0, 0.3)), (1, (0.0, 0.5)),
                                      (3, (0.0,))])
def test_demux_vs_pyref_bit_exact(V, alphas):
    p = synth.make_pileup(6, 300, V, seed=11 + V, mean_entries=40, min_entries=5, missing_gp_frac=0.1)
    out, full = ob.demux(p, alphas=alphas, doublet_prior=0.5, full_ll=True)
    gp = [list(map(float, p.gp[s].ravel())) for s in range(p.S)]
    for c in range(p.C):
        ll = pyref.demux_cell_ll(_cells_as_python(p, c), gp, p.has_gp, V, list(alphas))
        assert np.array_equal(full[c].ravel(), np.array(ll))
        rec = pyref.demux_call(ll, V, list(alphas), 0.5)
        for k, v in rec.items():
            got = out[k][c]
            assert (got == v) or (isinstance(v, float) and math.isnan(v) and math.isnan(got)), (c, k, got, v)


def test_demux_threads_do_not_change_results():
    p = synth.make_pileup(40, 500, 4, seed=3, mean_entries=60, min_entries=5)
    a = ob.demux(p, nthreads=1)
    b = ob.demux(p, nthreads=4)
    assert a.tobytes() == b.tobytes()


def test_demux_empty_cell_and_missing_gp():
    p = synth.make_pileup(5, 200, 3, seed=5, mean_entries=30, min_entries=5)
    # make cell 2 empty
    keep = np.ones(p.nnz, dtype=bool)
    keep[p.cell_ptr[2]:p.cell_ptr[3]] = False
    q = p.subset_cells([0, 1, 3, 4])
    cell_ptr = np.concatenate((q.cell_ptr[:3], q.cell_ptr[2:]))  # duplicate boundary -> empty cell at index 2
    q2 = synth.Pileup(5, q.S, cell_ptr, q.entry_snp, q.entry_rptr, q.reads, q.af, q.gp, q.has_gp)
    out = ob.demux(q2)
    assert out["valid"].tolist() == [1, 1, 0, 1, 1]
    ref = ob.demux(q)
    assert out[[0, 1, 3, 4]].tobytes() == ref.tobytes()
    # all SNPs without GP: every LL is 0, first sample wins the strict-< scans
    q3 = synth.Pileup(q.C, q.S, q.cell_ptr, q.entry_snp, q.entry_rptr, q.reads, q.af, q.gp, np.zeros(q.S, np.uint8))
    o3 = ob.demux(q3)
    assert np.all(o3["sngBestLLK"] == 0.0) and np.all(o3["sBest"] == 0) and np.all(o3["sNext"] == 1)


def test_demux_recovers_truth():
    p = synth.make_pileup(150, 3000, 8, seed=21, mean_entries=300)
    out = ob.demux(p)
    t = p.truth
    sng = ~t["is_doublet"]
    called_sng = out["type"] == 0
    assert (called_sng & sng).sum() >= 0.95 * sng.sum()
    ok = called_sng & sng
    assert np.all(out["sBest"][ok] == t["s1"][ok])
    dbl = out["type"] == 1
    assert (dbl & t["is_doublet"]).sum() >= 0.8 * t["is_doublet"].sum()
    for c in np.nonzero(dbl & t["is_doublet"])[0]:
        assert {int(out["dBest1"][c]), int(out["dBest2"][c])} == {int(t["s1"][c]), int(t["s2"][c])}


# ---------------------------------------------------------------------------------------------- freemuxlet

def test_fmx_entry_pileup_vs_pyref():
    p = synth.make_pileup(4, 100, 2, seed=2, mean_entries=30, min_entries=5, reads_lambda=2.0, other=0.05)
    e = ob.fmx_entry_pileup(p)
    for i in range(p.nnz):
        reads = [int(x) for x in p.reads[p.entry_rptr[i]:p.entry_rptr[i + 1]]]
        nr, nref, nalt, gls = pyref.fmx_entry_pileup(reads)
        assert (e["nreads"][i], e["nref"][i], e["nalt"][i]) == (nr, nref, nalt)
        assert np.array_equal(e["gls"][i], np.array(gls))
        assert abs(e["gls"][i].sum() - 1.0) < 1e-15 and e["gls"][i].min() >= 1e-6 / 1.00001


def test_plp_merge_vs_pyref_and_order_dependence():
    p = synth.make_pileup(3, 50, 2, seed=9, mean_entries=30, min_entries=10, reads_lambda=3.0)
    e = ob.fmx_entry_pileup(p)
    dst = np.zeros(1, dtype=ob.PLP)
    dst["gls"] = 1.0
    py = [0, 0, 0, [1.0] * 9]
    for i in range(12):
        ob.plp_merge(dst, e[i:i + 1])
        pyref.plp_merge(py, [int(e["nreads"][i]), int(e["nref"][i]), int(e["nalt"][i]), list(map(float, e["gls"][i]))])
        assert np.array_equal(dst["gls"][0], np.array(py[3]))
        assert (dst["nreads"][0], dst["nref"][0], dst["nalt"][0]) == tuple(py[:3])
    # the clamp makes the merge order-dependent (SURVEY hard part 4): reversing the order changes the bits
    rev = np.zeros(1, dtype=ob.PLP)
    rev["gls"] = 1.0
    for i in reversed(range(12)):
        ob.plp_merge(rev, e[i:i + 1])
    assert rev["nreads"][0] == dst["nreads"][0] and not np.array_equal(rev["gls"], dst["gls"])


needs_ref_merge = pytest.mark.skipif(not os.path.exists(ob.REF_MERGE_SO),
                                     reason="oracle/_ref/libmerge_ref.so not built (needs /root/reference at build time)")


def _random_chain_elements(rng, n, deep_frac=0.3):
    """entry pileups as calculate_snp_droplet_pileup leaves them (normalised, clamped at 1e-6), from random reads:
    shallow ones (1 + Poisson(0.3) reads, BQ 13..20 after the cap) and deep / high-quality ones whose merges make
    the clamp fire for most genotype pairs"""
    nreads = 1 + rng.poisson(0.3, n)
    deep = rng.random(n) < deep_frac
    nreads[deep] = rng.integers(2, 9, int(deep.sum()))
    rptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(nreads, out=rptr[1:])
    R = int(rptr[-1])
    allele = (rng.random(R) < 0.5).astype(np.uint8)
    # chains that agree (one allele throughout) push the opposite homozygote below the clamp quickly
    bq = rng.integers(13, 21, R).astype(np.uint8)
    hq = np.repeat(deep, nreads)
    bq[hq] = rng.integers(20, 41, int(hq.sum())).astype(np.uint8)
    reads = ((allele << 7) | bq).astype(np.uint8)
    reads[rng.random(R) < 0.005] = 0xFF
    out = np.zeros(n, dtype=ob.PLP)
    ob.lib().oracle_fmx_entry_pileup(ob.C.c_int64(n), ob._p(rptr), ob._p(reads), ob._p(out))
    return out


@needs_ref_merge
def test_plp_merge_matches_reference_header_bit_for_bit():
    """oracle_plp_merge against the reference's own snp_droplet_pileup::merge (sc_drop_seq.h:77-101, compiled from the
    unmodified header): 120 000 random chains, bit for bit, with the clamp firing in most of them"""
    rng = np.random.default_rng(20240904)
    nch = 120_000
    length = np.minimum(1 + rng.geometric(0.25, nch), 40)
    length[:2000] = rng.integers(30, 200, 2000)       # long chains: clamp + renormalise many times over
    ptr = np.zeros(nch + 1, dtype=np.int64)
    np.cumsum(length, out=ptr[1:])
    elems = _random_chain_elements(rng, int(ptr[-1]))
    # a third of the chains concordant: every element carries the same allele pattern as the chain's first
    conc = np.flatnonzero(rng.random(nch) < 0.33)
    for c in conc[:20000]:
        elems[ptr[c]:ptr[c + 1]] = elems[ptr[c]]
    got = ob.plp_merge_chains(ptr, elems)
    want = ob.ref_plp_merge_chains(ptr, elems)
    assert np.array_equal(got["gls"].view(np.uint64), want["gls"].view(np.uint64))
    for f in ("nreads", "nref", "nalt"):
        assert np.array_equal(got[f], want[f])
    # the clamp did fire: final states sitting exactly on the renormalised floor
    floor_hits = (want["gls"] < 1.0000001e-6).any(axis=1).mean()
    assert floor_hits > 0.3, floor_hits
    # and the single-merge entry point, from arbitrary (not default) states
    d0 = want[:5000].copy()
    d1 = d0.copy()
    for i in range(5000):
        ob.plp_merge(d0[i:i + 1], elems[i:i + 1])
        ob.ref_plp_merge(d1[i:i + 1], elems[i:i + 1])
    assert np.array_equal(d0["gls"].view(np.uint64), d1["gls"].view(np.uint64))


@needs_ref_merge
def test_fmx_sort_matches_std_sort_with_reference_comparator():
    """oracle_fmx_sort against std::sort under the reference's sc_drop_comp_t (sc_drop_seq.h:187-198,
    cmd_cram_freemux2.cpp:184-189) on tie-heavy scores; the comparator is a strict total order on distinct ids, so the
    result does not depend on the sort algorithm"""
    rng = np.random.default_rng(5)
    for n, levels in [(1, 1), (2, 1), (17, 3), (1000, 7), (50_000, 40), (200_000, 1000), (30_000, 10**9)]:
        scores = rng.integers(0, levels, n).astype(np.float64) * 0.25 - 3.0
        if n > 100:
            scores[rng.integers(0, n, 20)] = 0.0
            scores[rng.integers(0, n, 5)] = -0.0
        assert np.array_equal(ob.fmx_sort(scores), ob.ref_fmx_sort(scores))
    s = np.array([1.0, 3.0, 3.0, -2.0, 1.0, 0.0, -0.0])
    assert ob.ref_fmx_sort(s).tolist() == ob.fmx_sort(s).tolist() == [2, 1, 4, 0, 6, 5, 3]
    assert ob.ref_comp(s, 2, 1) and not ob.ref_comp(s, 1, 2) and ob.ref_comp(s, 1, 0) and not ob.ref_comp(s, 3, 0)
    assert ob.ref_comp(s, 6, 5) and not ob.ref_comp(s, 5, 6)      # 0.0 - (-0.0) == 0: tie, id descending
    # two infinite scores of one sign: cmp = inf - inf = NaN is "!= 0" and "not > 0" either way round, i.e. the
    # comparator stops being a strict weak order and the result would depend on the sort algorithm.  Scores are
    # differences of sums of logs of likelihoods >= 1e-6 (sc_drop_seq.cpp:498-506), so this cannot occur on the path.
    t = np.array([-np.inf, -np.inf])
    assert not ob.ref_comp(t, 0, 1) and not ob.ref_comp(t, 1, 0)


def test_fmx_sort_comparator():
    scores = np.array([1.0, 3.0, 3.0, -2.0, 1.0])
    assert ob.fmx_sort(scores).tolist() == [2, 1, 4, 0, 3]  # score desc, ties id desc


def test_fmx_estep_vs_pyref_and_em_runs():
    K = 3
    p = synth.make_pileup(60, 300, K, seed=13, mean_entries=150, min_entries=50, with_gp=False)
    e = ob.fmx_entry_pileup(p)
    llk0, llk2, ns, nr = ob.fmx_cell_scores(p, e)
    assert np.array_equal(ns, np.diff(p.cell_ptr).astype(np.int32))
    order = ob.fmx_sort(llk2 - llk0)
    clust = ob.fmx_greedy_init(p, e, K, llk2 - llk0, order)
    assert clust.min() >= 0 and clust.max() < K
    cplp = ob.fmx_build_cluster_pileup(p, e, K, clust)
    cells = ob.fmx_init_cells(clust)
    cplp0 = cplp.copy()
    ns_, na_, nch, full = ob.fmx_iterate(p, e, K, cplp, cells, full_ll=True)
    assert nch == p.C  # first iteration: every cell counts as changed (jBest starts at -1)
    # E-step of the first iteration vs the pair-by-pair Python restatement
    cp = [[list(map(float, cplp0["gls"][k, s])) for s in range(p.S)] for k in range(K)]
    for c in range(0, p.C, 13):
        ents = [(int(p.entry_snp[i]), list(map(float, e["gls"][i]))) for i in range(p.cell_ptr[c], p.cell_ptr[c + 1])]
        ll = pyref.fmx_estep_cell(ents, p.af, cp, K, 0.1)
        assert np.array_equal(full[c], np.array(ll))
    # iterate to convergence; singlet cells of one true donor end up in one cluster
    for _ in range(9):
        ns_, na_, nch = ob.fmx_iterate(p, e, K, cplp, cells)
        if nch == 0:
            break
    assert nch == 0
    t = p.truth
    sng = (cells["type"] == 0) & ~t["is_doublet"]
    assert sng.sum() >= 0.8 * (~t["is_doublet"]).sum()
    for d in range(K):
        cl = cells["clust"][sng & (t["s1"] == d)]
        if cl.size:
            assert np.all(cl == cl[0])


# ---------------------------------------------------------------------------------------------- golden vectors

@pytest.mark.parametrize("name", ["demux_v4_a2", "demux_v4_a6", "demux_v16_a2", "demux_v8_a3_deep", "demux_v64_a6"])
def test_oracle_reproduces_golden_demux(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = synth.Pileup(int(z["C"]), int(z["S"]), z["cell_ptr"], z["entry_snp"], z["entry_rptr"], z["reads"], z["af"],
                     z["gp"], z["has_gp"])
    out = ob.demux(p, alphas=tuple(z["alphas"]), doublet_prior=float(z["doublet_prior"]))
    assert out.tobytes() == z["cells"].tobytes()


@pytest.mark.parametrize("name", ["fmx_k4", "fmx_k4_mixed"])
def test_oracle_reproduces_golden_fmx(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = synth.Pileup(int(z["C"]), int(z["S"]), z["cell_ptr"], z["entry_snp"], z["entry_rptr"], z["reads"], z["af"])
    K = int(z["K"])
    e = ob.fmx_entry_pileup(p)
    assert np.array_equal(e["gls"], z["entry_gls"])
    llk0, llk2, ns, nr = ob.fmx_cell_scores(p, e)
    assert np.array_equal(llk0, z["llk0"]) and np.array_equal(llk2, z["llk2"])
    clust = ob.fmx_greedy_init(p, e, K, llk2 - llk0, ob.fmx_sort(llk2 - llk0))
    assert np.array_equal(clust, z["clust0"])
    cplp = ob.fmx_build_cluster_pileup(p, e, K, clust)
    cells = ob.fmx_init_cells(clust)
    stats = []
    for _ in range(int(z["n_iter"])):
        stats.append(ob.fmx_iterate(p, e, K, cplp, cells))
    assert np.array_equal(np.array(stats), z["stats"])
    assert cells.tobytes() == z["cells"].tobytes()
    assert np.array_equal(cplp["gls"], z["cluster_gls"])
