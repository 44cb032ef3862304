"""GPU parity tests of the freemuxlet-old deltas: pairwise droplet distance matrix and the voting
passes of cmd_cram_freemuxlet.cpp:176-343 through the C-ABI, and the `freemuxlet-old` command end to end, against the
CPU oracle.  Bar: counters and cluster labels exact, log-likelihoods within 1e-5 (observed ~1e-12)."""
import ctypes
import gzip
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import parity
import pyplp
from popscle_amd import muxgl, plpio, synth
from test_cli_gpu import BIN, TYPES, as_pileup, assert_rows_match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = muxgl.Engine(0)
    yield e
    e.close()


def prepared(eng, p):
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    eng.fmx_prepare(p.af)
    return ob.fmx_entry_pileup(p)


def oracle_signs(C, dd, thres):
    s = np.zeros((C, C), dtype=np.int8)
    bf = dd["llk2"] - dd["llk0"]
    a, b = np.tril_indices(C, -1)
    v = np.where(-bf > thres, -1, np.where(bf > thres, 1, 0)).astype(np.int8)
    s[a, b] = v[a * (a - 1) // 2 + b]
    s[b, a] = s[a, b]
    return s, bf


@pytest.mark.parametrize("C,S,me", [(1, 50, 20), (2, 50, 30), (37, 200, 60), (300, 1500, 150), (700, 900, 120)])
def test_pair_dist_vs_oracle(eng, C, S, me):
    p = synth.make_pileup(C, S, 3, seed=40 + C, mean_entries=me, min_entries=5, with_gp=False)
    e = prepared(eng, p)
    want = ob.fmxold_pair_dist(p, e)
    got = eng.fmxold_pair_dist(5.41, want_full=True)
    for f in ("nsnps", "nread1", "nread2"):
        assert np.array_equal(got[f], want[f]), f
    if C > 1:
        assert max(np.abs(got["llk0"] - want["llk0"]).max(), np.abs(got["llk2"] - want["llk2"]).max()) < 1e-9
    ws, bf = oracle_signs(C, want, 5.41)
    gs = eng.fmxold_signs()
    sure = np.ones((C, C), dtype=bool)
    if C > 1:
        a, b = np.tril_indices(C, -1)
        close = np.abs(np.abs(bf) - 5.41) < 1e-8
        sure[a[close], b[close]] = False
        sure[b[close], a[close]] = False
    assert np.array_equal(gs[sure], ws[sure])
    # the sign-only kernel (no records) yields the same matrix
    eng.fmxold_pair_dist(5.41, want_full=False)
    assert np.array_equal(eng.fmxold_signs(), gs)
    if C >= 300:
        assert (gs != 0).mean() > 0.02


def test_pair_dist_wide_columns(eng):
    """more cells than one column block of the record-collecting kernel (4096) and of a wave's column range"""
    C = 4300
    p = synth.make_pileup(C, 400, 3, seed=77, mean_entries=12, min_entries=3, with_gp=False)
    e = prepared(eng, p)
    want = ob.fmxold_pair_dist(p, e)
    got = eng.fmxold_pair_dist(2.0, want_full=True)
    assert np.array_equal(got["nsnps"], want["nsnps"]) and np.array_equal(got["nread2"], want["nread2"])
    assert np.abs(got["llk2"] - want["llk2"]).max() < 1e-9 and np.abs(got["llk0"] - want["llk0"]).max() < 1e-9


def test_pair_dist_sign_only_several_column_blocks(eng):
    """more cells than one column block of the sign-only kernel (6144): signs against the oracle's records"""
    C = 6300
    p = synth.make_pileup(C, 300, 3, seed=78, mean_entries=8, min_entries=2, with_gp=False)
    e = prepared(eng, p)
    want = ob.fmxold_pair_dist(p, e)
    eng.fmxold_pair_dist(1.0, want_full=False)
    gs = eng.fmxold_signs()
    bf = want["llk2"] - want["llk0"]
    del want
    a, b = np.tril_indices(C, -1)
    gv = gs[a, b]
    assert np.array_equal(gs, gs.T) and not gs.diagonal().any()
    del gs
    wv = np.where(-bf > 1.0, -1, np.where(bf > 1.0, 1, 0)).astype(np.int8)
    sure = np.abs(np.abs(bf) - 1.0) > 1e-8
    assert np.array_equal(gv[sure], wv[sure])
    assert (wv != 0).mean() > 0.001


def test_golden(eng):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fmxold_k4.npz"))
    p = synth.Pileup(int(g["C"]), int(g["S"]), g["cell_ptr"], g["entry_snp"], g["entry_rptr"], g["reads"], g["af"])
    K, thres, frac = int(g["K"]), float(g["bf_thres"]), float(g["frac_init_clust"])
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    eng.fmx_prepare(p.af)
    got = eng.fmxold_pair_dist(thres, want_full=True)
    want = g["dropd"]
    for f in ("nsnps", "nread1", "nread2"):
        assert np.array_equal(got[f], want[f])
    assert np.abs(got["llk0"] - want["llk0"]).max() < 1e-9 and np.abs(got["llk2"] - want["llk2"]).max() < 1e-9
    cl, cc = eng.fmxold_vote_init(K, g["order"], g["jitter0"], frac)
    assert np.array_equal(cl, g["clust0"]) and np.array_equal(cc, g["ccounts0"])
    for it in range(3):
        cl, ch, _ = eng.fmxold_vote_refine(K, g["orands"][it], g["jitters"][it], cl, it == 0)
        assert np.array_equal(cl, g["clusts"][it]) and ch == g["changed"][it]
    # the EM loop from those clusters against THE REFERENCE'S OWN run of cmd_cram_freemuxlet.cpp:456-653 (em_* arrays,
    # tests/golden/make_golden.py): driven the way popscle-amd freemuxlet-old drives the C-ABI (host/main.cpp): ten
    # iterations, geno_error only in the last, no early stop
    eng.fmx_set_clusters(K, cl)
    ge, dp = float(g["em_geno_error"]), float(g["em_doublet_prior"])
    worst = 0.0
    for it in range(10):
        cells, stats, full = eng.fmx_iterate(dp, ge if it == 9 else 0.0, want_full_ll=True)
        assert stats[:2] == tuple(g["em_counters"][it]), it
        want = g["em_cells"][it].copy()
        want["clust"] = np.where(want["type"] == 0, want["jBest"], -1)   # the old loop has no clusts update (row c1)
        rep = parity.compare_fmx(cells, want)
        assert rep["cells_needing_an_excuse"] == 0, (it, rep)
        worst = max(worst, rep["max_abs_ll_diff"])
        if it == 0:
            assert np.abs(full - g["em_full_ll_first"]).max() < 1e-7
    assert np.abs(full - g["em_full_ll_last"]).max() < 1e-7 and worst < 1e-7
    gls, cnt = eng.fmx_cluster_pileup()
    assert np.array_equal(cnt, g["em_cluster_cnt"])
    assert np.allclose(gls, g["em_cluster_gls"], rtol=1e-11, atol=1e-300)


def jitters(rng, n, K, mode):
    if mode == "rand":  # what the reference draws
        return rng.integers(0, 2**31, (n, K)) / (2.0**31) / 1000.0
    if mode == "zero":
        return np.zeros((n, K))
    if mode == "equal":  # all clusters share the jitter: the first maximum must win
        return np.repeat(rng.integers(0, 2**31, (n, 1)) / (2.0**31) / 1000.0, K, axis=1)
    # "near": jitters that differ far below the ulp of 1.0, so the order of the roundings decides the election
    base = rng.integers(0, 2**31, (n, 1)) / (2.0**31) / 1000.0
    return base + rng.permuted(np.tile(np.arange(K), (n, 1)), axis=1) * 2.0**-62


@pytest.mark.parametrize("K", [1, 2, 3, 7, 16, 33, 64])
@pytest.mark.parametrize("mode", ["rand", "zero", "equal", "near"])
def test_votes_vs_oracle(eng, K, mode):
    C = 230 if K < 33 else 150
    p = synth.make_pileup(C, 700, min(K, 6), seed=5 + K, mean_entries=160, min_entries=30, with_gp=False)
    e = prepared(eng, p)
    dd = ob.fmxold_pair_dist(p, e)
    thres = 1.5  # many informative pairs, many near-empty ones
    eng.fmxold_pair_dist(thres)
    rng = np.random.default_rng(K * 7 + len(mode))
    llk0, llk2, _, _ = ob.fmx_cell_scores(p, e)
    order = ob.fmx_sort(llk2 - llk0)
    for frac in (1.0, 0.6):
        nvis = sum(1 for i in range(C) if not i > C * frac)
        jit = jitters(rng, nvis, K, mode)
        want, wcc = ob.fmxold_vote_init(C, K, dd, order, jit, thres, frac)
        got, gcc = eng.fmxold_vote_init(K, order, jit, frac)
        assert np.array_equal(got, want), (K, mode, frac, np.flatnonzero(got != want)[:10])
        assert np.array_equal(gcc, wcc)
        clust = want
        for it in range(3):
            orand = rng.permutation(C).astype(np.int32)
            jit = jitters(rng, C, K, mode)
            keep = bool(it == 1)
            w, wch, wcc = ob.fmxold_vote_refine(C, K, dd, orand, jit, clust, thres, keep)
            g, gch, gcc = eng.fmxold_vote_refine(K, orand, jit, clust, keep)
            assert np.array_equal(g, w), (K, mode, frac, it, np.flatnonzero(g != w)[:10])
            assert gch == wch and np.array_equal(gcc, wcc)
            clust = w


def test_votes_edge_cases(eng):
    """empty droplets among the cells, a threshold nobody passes (every election is decided by the jitters alone), a
    threshold of zero (almost every pair votes), a single cell"""
    C, K = 120, 6
    p = synth.make_pileup(C, 400, 4, seed=9, mean_entries=60, min_entries=0, with_gp=False)
    p = p.subset_cells(np.arange(C))
    keep = np.ones(C, dtype=bool)
    keep[[3, 50, 119]] = False  # three droplets without any entry
    lens = np.where(keep, np.diff(p.cell_ptr), 0)
    q = p.subset_cells(np.flatnonzero(keep))
    cp = np.zeros(C + 1, dtype=np.int64)
    np.cumsum(lens, out=cp[1:])
    p = synth.Pileup(C, p.S, cp, q.entry_snp, q.entry_rptr, q.reads, p.af)
    assert (np.diff(p.cell_ptr) == 0).sum() == 3
    e = prepared(eng, p)
    dd = ob.fmxold_pair_dist(p, e)
    rng = np.random.default_rng(17)
    order = rng.permutation(C).astype(np.int32)
    for thres in (1e9, 0.0):
        got_full = eng.fmxold_pair_dist(thres, want_full=True)
        assert np.array_equal(got_full["nsnps"], dd["nsnps"])
        jit = jitters(rng, C, K, "rand")
        w, wcc = ob.fmxold_vote_init(C, K, dd, order, jit, thres, 1.0)
        g, gcc = eng.fmxold_vote_init(K, order, jit, 1.0)
        assert np.array_equal(g, w) and np.array_equal(gcc, wcc)
        if thres > 1:
            assert np.array_equal(g[order], np.argmax(jit, axis=1))  # no votes at all: the largest jitter wins
        jit = jitters(rng, C, K, "rand")
        orand = rng.permutation(C).astype(np.int32)
        w2, wch, _ = ob.fmxold_vote_refine(C, K, dd, orand, jit, w, thres, False)
        g2, gch, _ = eng.fmxold_vote_refine(K, orand, jit, w, False)
        assert np.array_equal(g2, w2) and gch == wch
    # frac_init_clust = 0: only the first cell of the order is visited (i > C * 0 is false for i = 0 only)
    jit = jitters(rng, 1, K, "rand")
    w, _ = ob.fmxold_vote_init(C, K, dd, order, jit, 2.0, 0.0)
    g, _ = eng.fmxold_vote_init(K, order, jit, 0.0)
    assert np.array_equal(g, w) and (g >= 0).sum() == 1
    # one cell
    q = synth.make_pileup(1, 50, 2, seed=3, mean_entries=20, with_gp=False)
    prepared(eng, q)
    eng.fmxold_pair_dist(5.41)
    jit = jitters(rng, 1, 3, "rand")
    g, cc = eng.fmxold_vote_init(3, np.zeros(1, dtype=np.int32), jit, 1.0)
    assert g[0] == int(np.argmax(jit[0])) and cc.sum() == 1
    g2, ch, _ = eng.fmxold_vote_refine(3, np.zeros(1, dtype=np.int32), jit, g, False)
    assert g2[0] == g[0] and ch == 0


def test_votes_rounding_sensitive(eng):
    """a hand-made sign matrix is not available through the ABI, so build pileups whose pair matrix is dense in +-1 and
    give the clusters jitters 2^-62 apart: the elected cluster then depends on which binades each cluster's running vote
    visited (vote_exact in fmx_old.hip).  The oracle adds the votes one by one like the reference."""
    C, K = 400, 5
    p = synth.make_pileup(C, 300, 3, seed=91, mean_entries=120, min_entries=60, with_gp=False)
    e = prepared(eng, p)
    dd = ob.fmxold_pair_dist(p, e)
    eng.fmxold_pair_dist(0.3)
    rng = np.random.default_rng(3)
    clust = rng.integers(0, K, C).astype(np.int32)
    ndiff = 0
    for it in range(6):
        orand = rng.permutation(C).astype(np.int32)
        jit = jitters(rng, C, K, "near")
        w, wch, _ = ob.fmxold_vote_refine(C, K, dd, orand, jit, clust, 0.3, False)
        g, gch, _ = eng.fmxold_vote_refine(K, orand, jit, clust, False)
        assert np.array_equal(g, w) and gch == wch
        # the same pass with the jitters taken as exact reals (no rounding path) must differ somewhere, otherwise this
        # test does not exercise what it claims to
        w2, _, _ = ob.fmxold_vote_refine(C, K, dd, orand, np.round(jit, 12), clust, 0.3, False)
        ndiff += int((w2 != w).sum())
        clust = rng.integers(0, K, C).astype(np.int32)
    assert ndiff > 0


def glibc_rand_stream():
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)  # the C library's state when no srand() was called
    libc.rand.restype = ctypes.c_int
    return libc.rand


@pytest.mark.parametrize("geno_error,frac", [(0.0, 1.0), (0.05, 0.7)])
def test_freemuxlet_old_cli(tmp_path, geno_error, frac):
    K = 4
    p = synth.make_pileup(160, 1200, K, seed=18, mean_entries=220, min_entries=40, with_gp=False, cap_bq=60, min_bq=2)
    prefix = str(tmp_path / "plp")
    plpio.write_plp(prefix, p, seed=18)
    out = str(tmp_path / "out")
    cmd = [BIN, "freemuxlet-old", "--plp", prefix, "--nsample", str(K), "--out", out, "--aux-files", "--geno-error",
           str(geno_error), "--frac-init-clust", str(frac), "--cap-BQ", "20", "--min-BQ", "30"]  # parsed, never applied
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = pyplp.load(prefix, min_bq=1, cap_bq=60)  # sc_drop_seq.h:181: the loader's own defaults
    q = as_pileup(d)
    C = q.C
    e = ob.fmx_entry_pileup(q)
    llk0, llk2, ns, nr = ob.fmx_cell_scores(q, e)
    want = ["INT_ID\tBARCODE\tNSNPs\tNREADs\tDBL.LLK\tSNG.LLK\tLOG.BF\tBFpSNP\n"]
    for i in range(C):
        want.append("%d\t%s\t%d\t%d\t%.2f\t%.2f\t%.2f\t%.4f\n" % (i, d["bcs"][i], ns[i], nr[i], llk0[i], llk2[i],
                                                                 llk0[i] - llk2[i], (llk0[i] - llk2[i]) / ns[i]))
    assert_rows_match(open(out + ".lmix").readlines(), want)

    # the reference's random stream: K draws per visited cell, then per pass C-1 draws of random_shuffle and C*K jitters
    rand = glibc_rand_stream()
    RAND_MAX = 2147483647
    order = ob.fmx_sort(llk2 - llk0)
    dd = ob.fmxold_pair_dist(q, e)
    nvis = sum(1 for i in range(C) if not i > C * frac)
    jit = np.array([[rand() / (RAND_MAX + 1.0) / 1000.0 for _ in range(K)] for _ in range(nvis)])
    clust, _ = ob.fmxold_vote_init(C, K, dd, order, jit, 5.41, frac)
    with gzip.open(out + ".ldist.gz", "rt") as f:
        rows = f.readlines()
    assert len(rows) == 1 + nvis * (nvis - 1) // 2
    t = rows[1 + 5 * 4 // 2 + 2].split("\t")  # i = 5, j = 2 of the visiting order
    si, sj = int(order[5]), int(order[2])
    rec = dd[max(si, sj) * (max(si, sj) - 1) // 2 + min(si, sj)]
    assert (int(t[0]), int(t[1]), int(t[2]), int(t[3]), int(t[4])) == (si, sj, rec["nsnps"], rec["nread1"], rec["nread2"])
    for it in range(10):
        orand = list(range(C))
        for i in range(1, C):
            j = rand() % (i + 1)
            if i != j:
                orand[i], orand[j] = orand[j], orand[i]
        jit = np.array([[rand() / (RAND_MAX + 1.0) / 1000.0 for _ in range(K)] for _ in range(C)])
        clust, _, _ = ob.fmxold_vote_refine(C, K, dd, np.array(orand, dtype=np.int32), jit, clust, 5.41, False)
    with gzip.open(out + ".clust0.samples.gz", "rt") as f:
        got0 = np.array([int(x.split("\t")[2]) for x in f.readlines()[1:]])
    assert np.array_equal(got0, clust)
    assert len(np.unique(clust)) > 1

    cplp = ob.fmx_build_cluster_pileup(q, e, K, clust)
    cells = ob.fmx_init_cells(clust)
    for it in range(10):
        ob.fmx_iterate(q, e, K, cplp, cells, 0.5, geno_error if it == 9 else 0.0)
    want = ["INT_ID\tBARCODE\tNUM.SNPS\tNUM.READS\tDROPLET.TYPE\tBEST.GUESS\tBEST.LLK\tNEXT.GUESS\tNEXT.LLK\t"
            "DIFF.LLK.BEST.NEXT\tBEST.POSTERIOR\tSNG.POSTERIOR\tSNG.BEST.GUESS\tSNG.BEST.LLK\tSNG.NEXT.GUESS\t"
            "SNG.NEXT.LLK\tSNG.ONLY.POSTERIOR\tDBL.BEST.GUESS\tDBL.BEST.LLK\tDIFF.LLK.SNG.DBL\n"]
    for i in range(C):
        c = cells[i]
        want.append("%d\t%s\t%d\t%d\t%s\t%d,%d\t%.2f\t%d,%d\t%.2f\t%.2f\t%.5f\t%.2g\t%d\t%.2f\t%d\t%.2f\t%.5f\t%d,%d\t%.2f\t"
                    "%.2f\n" % (i, d["bcs"][i], ns[i], nr[i], TYPES[int(c["type"])], c["jBest"], c["kBest"], c["bestLLK"],
                                c["jNext"], c["kNext"], c["nextLLK"], c["bestLLK"] - c["nextLLK"], c["bestPP"],
                                c["sngPP"], c["sBest"], c["sngBestLLK"], c["sNext"], c["sngNextLLK"], c["sngOnlyPP"],
                                c["dBest1"], c["dBest2"], c["dblBestLLK"], c["sngBestLLK"] - c["dblBestLLK"]))
    with gzip.open(out + ".clust1.samples.gz", "rt") as f:
        assert_rows_match(f.readlines(), want)
    assert os.path.exists(out + ".clust0.vcf.gz") and os.path.exists(out + ".clust1.vcf.gz")
    # most singlets end up with their source sample's cluster mates
    truth = p.truth["s1"]
    sng = np.array([int(c["type"]) == 0 for c in cells]) & ~p.truth["is_doublet"]
    lab = np.array([int(c["jBest"]) for c in cells])
    agree = 0
    for k in np.unique(lab[sng]):
        agree += np.bincount(truth[sng & (lab == k)]).max()
    assert agree > 0.9 * sng.sum()
