"""GPU parity at the shapes of BASELINE.json's configs.  configs[2] (demuxlet, 100 k cells x 64 samples x 200 k SNPs, six
alphas) and configs[4] (freemuxlet, 500 k cells x 500 k SNPs, K = 64, eight cell-sharded ranks) run at FULL size with
size-independent properties and oracle samples; configs[2] and configs[3] (freemuxlet, 16 clusters, 100 k SNPs) also run
with a reduced number of cells, where the CPU oracle checks the whole trajectory (cells are independent in demuxlet and
enter freemuxlet's E-step independently, so the per-cell arithmetic is the full-size one: same V / K / A / SNP axis /
entry density)."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import parity
from popscle_amd import muxgl, synth

pytestmark = pytest.mark.gpu
NT = max(1, min(64, os.cpu_count() or 1))


def test_config2_shape_demuxlet():
    cfg = synth.CONFIGS[2]
    alphas = cfg["alphas"]
    p = synth.make_pileup(1500, cfg["S"], cfg["V"], seed=synth.BASE_SEED + 2)
    with muxgl.Engine(0) as eng:
        eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        eng.demux_set_gp(p.gp, p.has_gp)
        cells = eng.demux_run(alphas, 0.5)
        # oracle on a sample of the cells
        pick = np.sort(np.random.default_rng(0).choice(p.C, 24, replace=False))
        sub = p.subset_cells(pick)
        want, wfull = ob.demux(sub, alphas=alphas, full_ll=True, nthreads=NT)
        rep = parity.compare_demux(cells[pick], want, alphas, sub)
        assert rep["max_abs_ll_diff"] < 1e-6
        # the same cells alone: bit-identical records, and the full hypothesis tensor against the oracle's
        eng.set_pileup(sub.S, sub.cell_ptr, sub.entry_snp, sub.entry_rptr, sub.reads)
        got, gfull = eng.demux_run(alphas, 0.5, want_full_ll=True)
        assert got.tobytes() == cells[pick].tobytes()
        assert parity.compare_full_ll(gfull, wfull, cfg["V"], alphas) < 1e-6
        n5 = alphas.index(0.5)
        assert np.array_equal(gfull[..., n5], gfull[..., n5].transpose(0, 2, 1))
    # calls agree with the simulated truth
    t = p.truth
    sng = (cells["type"] == 0) & ~t["is_doublet"]
    assert sng.sum() > 0.9 * (~t["is_doublet"]).sum() and np.all(cells["sBest"][sng] == t["s1"][sng])


def test_config2_full_size_demuxlet():
    """BASELINE configs[2] at FULL size: 100 k cells x 64 samples x 200 k SNPs, six alphas (95 M entries; the 41 GB pG
    table and the 19.6 GB result slabs of the wave path at size).  Size-independent properties: a sample of the cells run
    alone gives bit-identical records (cells are independent and the work cut of a cell depends on the cell alone), the
    alpha = 0.5 slice of their hypothesis tensor is symmetric, the oracle agrees on them, and the calls agree with the
    simulated truth."""
    cfg = synth.CONFIGS[2]
    alphas = cfg["alphas"]
    p = synth.make_pileup(cfg["C"], cfg["S"], cfg["V"], seed=synth.BASE_SEED + 2)
    assert p.C == 100_000 and p.nnz > 90_000_000
    with muxgl.Engine(0) as eng:
        eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        eng.demux_set_gp(p.gp, p.has_gp)
        cells = eng.demux_run(alphas, 0.5)
        assert cells["valid"].all()
        again = eng.demux_run(alphas, 0.5)
        assert again.tobytes() == cells.tobytes()  # run to run
        pick = np.sort(np.random.default_rng(0).choice(p.C, 24, replace=False))
        longest = np.argsort(np.diff(p.cell_ptr))[-2:]  # and the two longest cells (walked in parts)
        pick = np.unique(np.concatenate([pick, longest]))
        sub = p.subset_cells(pick)
        eng.set_pileup(sub.S, sub.cell_ptr, sub.entry_snp, sub.entry_rptr, sub.reads)
        got, gfull = eng.demux_run(alphas, 0.5, want_full_ll=True)
    assert got.tobytes() == cells[pick].tobytes()
    n5 = alphas.index(0.5)
    assert np.array_equal(gfull[..., n5], gfull[..., n5].transpose(0, 2, 1))
    want, wfull = ob.demux(sub, alphas=alphas, full_ll=True, nthreads=NT)
    rep = parity.compare_demux(cells[pick], want, alphas, sub)   # every integer field equal, no relaxation
    assert rep["max_abs_ll_diff"] < 1e-6
    print("configs[2] full size, oracle sample:", rep["cells"], "cells,", rep["exact_pass"])
    # the exact-call pass over ALL 100 k cells: how many it looks at, how many near ties, how long
    import time
    allc = cells.copy()
    t0 = time.perf_counter()
    st = muxgl.demux_exact_calls(p, alphas, allc, 0.5)
    dt = time.perf_counter() - t0
    print("configs[2] full size, exact-call pass over all cells:", st, f"{dt:.2f} s on {os.cpu_count()} threads")
    assert st["deep"] < 50 and st["near_ties"] < 0.01 * p.C, st
    assert allc[pick].tobytes() == np.ascontiguousarray(parity.exact(cells[pick], alphas, sub)).tobytes()
    assert parity.compare_full_ll(gfull, wfull, cfg["V"], alphas) < 1e-6
    t = p.truth
    sng = (cells["type"] == 0) & ~t["is_doublet"]
    assert sng.sum() > 0.9 * (~t["is_doublet"]).sum() and np.all(cells["sBest"][sng] == t["s1"][sng])
    dbl = (cells["type"] == 1) & t["is_doublet"]
    assert dbl.sum() > 0.9 * t["is_doublet"].sum()


def test_config4_full_size_freemuxlet_eight_ranks():
    """BASELINE configs[4] at FULL size -- freemuxlet --nsample 64, 500 k cells x 500 k SNPs (478 M entries), cell-sharded
    over 8 ranks -- on one GPU: a device group of eight virtual ranks (device 0 named eight times), each with the slabs,
    shard shapes and exchanges of the 8-GPU run (62 500 cells x 500 k SNPs row slab, 500 k cells x 62 500 SNPs column slab).
    From a seeded --init-cluster start, two EM iterations must equal the one-device, whole-pileup run of the same job bit
    for bit (records of all 500 k cells and counters); the oracle checks a sample: the cluster pileups of sampled SNPs
    (ordered merge over ALL cells of those SNPs) and the E-step / scans / re-assignment of sampled cells."""
    cfg = synth.CONFIGS[4]
    K, S, C = cfg["V"], cfg["S"], cfg["C"]
    p = synth.make_pileup(C, S, K, seed=synth.BASE_SEED + 4, with_gp=False)
    assert p.C == 500_000 and p.nnz > 450_000_000
    rng = np.random.default_rng(4)
    clust0 = np.where(rng.random(C) < 0.9, p.truth["s1"], -1).astype(np.int32)
    wrong = rng.random(C) < 0.02  # a few cells start in a random cluster
    clust0[wrong] = rng.integers(0, K, int(wrong.sum()))
    with muxgl.Engine([0] * 8) as g:
        g.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        llk0, llk2, ns, nr = g.fmx_prepare(p.af)
        g.fmx_set_clusters(K, clust0)
        gls0, cnt0 = g.fmx_cluster_pileup()  # cluster pileups of the start, [K][S][9]
        its = [g.fmx_iterate(0.5, 0.1) for _ in range(2)]
    assert np.array_equal(ns, np.diff(p.cell_ptr).astype(np.int32))
    # oracle, cluster pileups: the ordered clamped merge over all cells, for a sample of the SNPs
    snps = np.sort(rng.choice(S, 300, replace=False))
    keep = np.isin(p.entry_snp, snps)
    cp, es, er, rd = _masked(p, keep)
    q = synth.Pileup(C, S, cp, es, er, rd, p.af)
    qe = ob.fmx_entry_pileup(q)
    oc = ob.fmx_build_cluster_pileup(q, qe, K, clust0)
    assert np.array_equal(cnt0[:, snps], np.stack([oc["nreads"], oc["nref"], oc["nalt"]], axis=-1)[:, snps])
    assert np.allclose(gls0[:, snps], oc["gls"][:, snps], rtol=1e-10, atol=1e-300)
    # oracle, E-step + scans + re-assignment of a sample of the cells against the device's own cluster pileups
    cells_s = np.sort(rng.choice(C, 40, replace=False))
    sub = p.subset_cells(cells_s)
    se = ob.fmx_entry_pileup(sub)
    o0, o2, _, _ = ob.fmx_cell_scores(sub, se)
    assert np.max(np.abs(llk0[cells_s] - o0)) < 1e-7 and np.max(np.abs(llk2[cells_s] - o2)) < 1e-7
    cplp = np.zeros((K, S), dtype=ob.PLP)
    cplp["gls"] = gls0
    cplp["nreads"], cplp["nref"], cplp["nalt"] = cnt0[..., 0], cnt0[..., 1], cnt0[..., 2]
    del gls0, cnt0
    ocells = ob.fmx_init_cells(np.ascontiguousarray(clust0[cells_s]))
    ob.fmx_iterate(sub, se, K, cplp, ocells, 0.5, 0.1, nthreads=NT)
    rep = parity.compare_fmx(its[0][0][cells_s], ocells, resolved=False)   # a group counts near ties, and the checker starts from the device's pileups
    assert rep["max_abs_ll_diff"] < 1e-6
    del cplp
    # the one-device, whole-pileup run of the same job: bit-identical records and counters
    with muxgl.Engine(0) as e:
        e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        w0, w2, _, _ = e.fmx_prepare(p.af)
        assert np.array_equal(w0, llk0) and np.array_equal(w2, llk2)
        e.fmx_set_clusters(K, clust0)
        for it in range(2):
            cells, stats = e.fmx_iterate(0.5, 0.1)
            assert tuple(stats) == tuple(its[it][1]), (it, stats, its[it][1])
            assert cells.tobytes() == its[it][0].tobytes()
    ok = (cells["type"] == 0) & ~p.truth["is_doublet"]
    assert ok.sum() > 0.9 * (~p.truth["is_doublet"]).sum() and (cells["clust"][ok] == p.truth["s1"][ok]).mean() > 0.99


def test_config3_full_size_freemuxlet_oracle_sample():
    """BASELINE configs[3] at FULL size (50 k cells x 100 k SNPs, K = 16, 47.6 M entries; the oct E-step and the stream
    M-step): three EM iterations from a seeded start; after each, the oracle redoes the E-step, scans and re-assignment of a
    sample of the cells against the device's own cluster pileups of that iteration, and the ordered clamped merge of a
    sample of the SNPs over ALL cells."""
    cfg = synth.CONFIGS[3]
    K, S, C = cfg["V"], cfg["S"], cfg["C"]
    p = synth.make_pileup(C, S, K, seed=synth.BASE_SEED + 3, with_gp=False)
    assert p.C == 50_000 and p.nnz > 45_000_000
    rng = np.random.default_rng(3)
    clust = np.where(rng.random(C) < 0.9, p.truth["s1"], -1).astype(np.int32)
    pick = np.sort(rng.choice(C, 32, replace=False))
    sub = p.subset_cells(pick)
    se = ob.fmx_entry_pileup(sub)
    snps = np.sort(rng.choice(S, 200, replace=False))
    cp, es, er, rd = _masked(p, np.isin(p.entry_snp, snps))
    q = synth.Pileup(C, S, cp, es, er, rd, p.af)
    qe = ob.fmx_entry_pileup(q)
    with muxgl.Engine(0) as e:
        e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        e.fmx_prepare(p.af)
        e.fmx_set_clusters(K, clust)
        for it in range(3):
            gls, cnt = e.fmx_cluster_pileup()
            # the ordered merge over all cells of the sampled SNPs (cmd_cram_freemux2.cpp:277-288 / :590-596)
            oc = ob.fmx_build_cluster_pileup(q, qe, K, clust)
            assert np.array_equal(cnt[:, snps], np.stack([oc["nreads"], oc["nref"], oc["nalt"]], axis=-1)[:, snps])
            assert np.allclose(gls[:, snps], oc["gls"][:, snps], rtol=1e-10, atol=1e-300)
            cplp = np.zeros((K, S), dtype=ob.PLP)
            cplp["gls"] = gls
            cplp["nreads"], cplp["nref"], cplp["nalt"] = cnt[..., 0], cnt[..., 1], cnt[..., 2]
            cells, stats = e.fmx_iterate(0.5, 0.1)
            ocells = ob.fmx_init_cells(np.ascontiguousarray(clust[pick]))
            ob.fmx_iterate(sub, se, K, cplp, ocells, 0.5, 0.1, nthreads=NT)
            rep = parity.compare_fmx(cells[pick], ocells, resolved=False)   # (the checker starts from the device's own pileups)
            assert rep["max_abs_ll_diff"] < 1e-6, (it, rep)
            clust = np.where(cells["type"] == 0, cells["clust"], -1).astype(np.int32)  # only singlets merge (:590-596)
    ok = (cells["type"] == 0) & ~p.truth["is_doublet"]
    assert ok.sum() > 0.9 * (~p.truth["is_doublet"]).sum()


def _masked(p, keep):
    """packed arrays of the entries selected by the boolean mask (all cells)"""
    from popscle_amd.synth import _ranges

    cell_of = np.repeat(np.arange(p.C, dtype=np.int64), np.diff(p.cell_ptr))
    cp = np.zeros(p.C + 1, dtype=np.int64)
    np.cumsum(np.bincount(cell_of[keep], minlength=p.C), out=cp[1:])
    eidx = np.flatnonzero(keep)
    rl = p.entry_rptr[eidx + 1] - p.entry_rptr[eidx]
    er = np.zeros(eidx.size + 1, dtype=np.int64)
    np.cumsum(rl, out=er[1:])
    return cp, np.ascontiguousarray(p.entry_snp[eidx]), er, np.ascontiguousarray(p.reads[_ranges(p.entry_rptr[eidx], rl)])


def test_config3_full_size_greedy_init_vs_oracle():
    """BASELINE configs[3] at FULL size (50 k cells x 100 k SNPs, K = 16): the greedy initial clustering of the device
    (one launch of the batched kernel, near ties through the exact path) must be the oracle's sequential loop
    (cmd_cram_freemux2.cpp:217-261; ~40 s on one core) cell for cell -- about 3 000 cells per cluster, clamps saturated,
    which is where start product x ratio-of-replayed-terms has the most to get wrong."""
    cfg = synth.CONFIGS[3]
    K, S, C = cfg["V"], cfg["S"], cfg["C"]
    p = synth.make_pileup(C, S, K, seed=synth.BASE_SEED + 3, with_gp=False)
    e = ob.fmx_entry_pileup(p)
    o0, o2, _, _ = ob.fmx_cell_scores(p, e)
    scores = o2 - o0
    want = ob.fmx_greedy_init(p, e, K, scores, ob.fmx_sort(scores))
    with muxgl.Engine(0) as en:
        en.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        en.fmx_prepare(p.af)
        got = en.fmx_greedy_init(K, scores)
        near, over = en.fmx_greedy_stats()
    diff = np.flatnonzero(got != want)
    assert diff.size == 0, f"{diff.size} of {C} cells differ, first {diff[:5]}"
    print(f"configs[3] greedy init: {near} near ties through the exact path, {over} overruled")
    assert np.bincount(want, minlength=K).min() > 1000


def test_greedy_init_k64_20k_cells_vs_oracle():
    """K = 64 at 20 000 cells x 100 k SNPs (configs[4]'s cluster count; ~450 entries per cell, which keeps the oracle's
    sequential loop at half a minute)"""
    K, S, C = 64, 100_000, 20_000
    p = synth.make_pileup(C, S, K, seed=synth.BASE_SEED + 44, with_gp=False, mean_entries=400.0)
    e = ob.fmx_entry_pileup(p)
    o0, o2, _, _ = ob.fmx_cell_scores(p, e)
    scores = o2 - o0
    want = ob.fmx_greedy_init(p, e, K, scores, ob.fmx_sort(scores))
    with muxgl.Engine(0) as en:
        en.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        en.fmx_prepare(p.af)
        got = en.fmx_greedy_init(K, scores)
        near, over = en.fmx_greedy_stats()
    diff = np.flatnonzero(got != want)
    assert diff.size == 0, f"{diff.size} of {C} cells differ, first {diff[:5]}"
    print(f"K = 64 greedy init: {near} near ties through the exact path, {over} overruled")


def test_config3_shape_freemuxlet():
    cfg = synth.CONFIGS[3]
    K = cfg["V"]
    p = synth.make_pileup(2500, cfg["S"], K, seed=synth.BASE_SEED + 3, with_gp=False)
    rng = np.random.default_rng(1)
    clust0 = np.where(rng.random(p.C) < 0.85, p.truth["s1"], rng.integers(0, K, p.C)).astype(np.int32)
    clust0[rng.random(p.C) < 0.05] = -1
    e = ob.fmx_entry_pileup(p)
    cplp = ob.fmx_build_cluster_pileup(p, e, K, clust0)
    ocells = ob.fmx_init_cells(clust0)
    with muxgl.Engine(0) as eng:
        eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        llk0, llk2, ns, nr = eng.fmx_prepare(p.af)
        o0, o2, ons, onr = ob.fmx_cell_scores(p, e)
        assert np.array_equal(ns, ons) and np.array_equal(nr, onr)
        assert np.max(np.abs(llk0 - o0)) < 1e-7 and np.max(np.abs(llk2 - o2)) < 1e-7
        eng.fmx_set_clusters(K, clust0)
        for it in range(3):
            ostats = ob.fmx_iterate(p, e, K, cplp, ocells, 0.5, 0.1, nthreads=NT)
            gcells, gstats = eng.fmx_iterate(0.5, 0.1)
            assert tuple(gstats) == tuple(ostats), (it, gstats, ostats)
            rep = parity.compare_fmx(gcells, ocells)
            assert rep["max_abs_ll_diff"] < 1e-6
        g, c = eng.fmx_cluster_pileup()
        assert np.array_equal(c, np.stack([cplp["nreads"], cplp["nref"], cplp["nalt"]], axis=-1))
        assert np.allclose(g, cplp["gls"], rtol=1e-10, atol=1e-300)
    ok = (gcells["type"] == 0) & ~p.truth["is_doublet"]
    assert ok.sum() > 0.5 * (~p.truth["is_doublet"]).sum()  # sanity only: parity with the oracle is the test above


def test_config1_shape_freemuxlet_old_pair_matrix():
    """freemuxlet-old's pair matrix at the cell and SNP counts of configs[1] (909 M pair terms): size-independent
    properties.  A pair's record depends on its two cells only and its terms are accumulated in SNP order, so the signs
    among a subset of the cells must be reproduced bit for bit by a run on that subset alone; the oracle checks a
    sample of the pairs."""
    cfg = synth.CONFIGS[1]
    p = synth.make_pileup(cfg["C"], cfg["S"], cfg["V"], seed=synth.BASE_SEED + 1, with_gp=False, cap_bq=60)
    pick = np.sort(np.random.default_rng(2).choice(p.C, 400, replace=False))
    sub = p.subset_cells(pick)
    with muxgl.Engine(0) as eng:
        eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        eng.fmx_prepare(p.af)
        eng.fmxold_pair_dist(5.41)
        s = eng.fmxold_signs()
        assert np.array_equal(s, s.T) and not s.diagonal().any()
        assert 0.05 < (s != 0).mean() < 1.0
        eng.set_pileup(sub.S, sub.cell_ptr, sub.entry_snp, sub.entry_rptr, sub.reads)
        eng.fmx_prepare(sub.af)
        got = eng.fmxold_pair_dist(5.41, want_full=True)
        ssub = eng.fmxold_signs()
    assert np.array_equal(ssub, s[np.ix_(pick, pick)])
    want = ob.fmxold_pair_dist(sub, ob.fmx_entry_pileup(sub))
    assert np.array_equal(got["nsnps"], want["nsnps"]) and np.array_equal(got["nread1"], want["nread1"])
    assert np.abs(got["llk0"] - want["llk0"]).max() < 1e-8 and np.abs(got["llk2"] - want["llk2"]).max() < 1e-8
    # same-donor singlet pairs look alike (+1), different donors differ (-1), wherever the evidence passes the threshold
    t = p.truth
    sng = ~t["is_doublet"][pick]
    same = t["s1"][pick][:, None] == t["s1"][pick][None, :]
    m = sng[:, None] & sng[None, :] & ~np.eye(len(pick), dtype=bool)
    assert (ssub[m & same] >= 0).mean() > 0.99 and (ssub[m & ~same] <= 0).mean() > 0.99


def test_linear_entry_forms_agree_with_the_general_ones():
    """Entries with one usable read take a two-term form of the pair sum (demux_wave.hip EM_LINEAR, fmx_wave.hip LIN);
    MUXGL_FLAG_NO_LINEAR_ENTRIES sends every entry through the general three-term form.  Same calls, log-likelihoods
    within rounding of each other, and a mix of both kinds of entries in every cell (reads_lambda = 0.6)."""
    V = 40
    alphas = (0.0, 0.2, 0.35, 0.45, 0.5)
    p = synth.make_pileup(120, 3000, V, seed=77, mean_entries=300, min_entries=40, reads_lambda=0.6, other=0.03)
    res = []
    for flags in (0, muxgl.FLAG_NO_LINEAR_ENTRIES):
        with muxgl.Engine(0, flags) as eng:
            eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
            eng.demux_set_gp(p.gp, p.has_gp)
            res.append(eng.demux_run(alphas, 0.5, want_full_ll=True))
    (c1, f1), (c2, f2) = res
    assert np.array_equal(c1["type"], c2["type"]) and np.array_equal(c1["sBest"], c2["sBest"])
    assert np.abs(f1 - f2).max() < 1e-9 and (f1 != f2).any()  # two different association orders were really run
    want, wfull = ob.demux(p, alphas=alphas, full_ll=True, nthreads=NT)
    assert parity.compare_full_ll(f1, wfull, V, alphas) < 1e-7 and parity.compare_full_ll(f2, wfull, V, alphas) < 1e-7
    K = 40
    q = synth.make_pileup(100, 3000, 8, seed=78, mean_entries=300, min_entries=40, reads_lambda=0.6, other=0.03, with_gp=False,
                          cap_bq=60)  # quality 60: the clamp fires on some single-read entries, which must stay general
    clust0 = (np.arange(q.C) % K).astype(np.int32)
    out = []
    for flags in (0, muxgl.FLAG_NO_LINEAR_ENTRIES):
        with muxgl.Engine(0, flags) as eng:
            eng.set_pileup(q.S, q.cell_ptr, q.entry_snp, q.entry_rptr, q.reads)
            eng.fmx_prepare(q.af)
            eng.fmx_set_clusters(K, clust0)
            out.append(eng.fmx_iterate(0.5, 0.1, want_full_ll=True))
    (a, sa, fa), (b, sb, fb) = out
    assert tuple(sa) == tuple(sb) and np.array_equal(a["type"], b["type"]) and np.array_equal(a["clust"], b["clust"])
    assert np.abs(fa - fb).max() < 1e-9 and (fa != fb).any()
    e = ob.fmx_entry_pileup(q)
    cplp = ob.fmx_build_cluster_pileup(q, e, K, clust0)
    ocells = ob.fmx_init_cells(clust0)
    ob.fmx_iterate(q, e, K, cplp, ocells, 0.5, 0.1, nthreads=NT)
    assert parity.compare_fmx(a, ocells)["max_abs_ll_diff"] < 1e-7


def test_pivoted_pair_sums_agree_with_the_three_term_sums():
    """freemuxlet E-step beyond 32 clusters: the pair sums of the non-linear entries taken around the lane's smallest
    term (fmx_wave.hip, PIV) against the three-term sums (MUXGL_FLAG_NO_PIVOT_SUMS) and the oracle: same counters and
    assignments, LL tensors within 1e-9 -- on shallow cells and on deep entries with reads of both alleles, where a pivot
    fixed per entry would cancel."""
    for K, C, S, me, lam in [(40, 60, 4000, 600, 0.3), (64, 40, 3000, 500, 30.0), (70, 24, 4000, 900, 4.0)]:
        p = synth.make_pileup(C, S, 8, seed=1234 + K, mean_entries=me, min_entries=30, reads_lambda=lam, max_bq=40, cap_bq=40,
                              with_gp=False)
        e = ob.fmx_entry_pileup(p)
        llk0, llk2, _, _ = ob.fmx_cell_scores(p, e)
        clust = ob.fmx_greedy_init(p, e, K, llk2 - llk0, ob.fmx_sort(llk2 - llk0))
        cplp = ob.fmx_build_cluster_pileup(p, e, K, clust)
        cells = ob.fmx_init_cells(clust)
        want = ob.fmx_iterate(p, e, K, cplp, cells, 0.5, 0.1, full_ll=True)
        got = []
        for flags in (0, muxgl.FLAG_NO_PIVOT_SUMS):
            with muxgl.Engine(0, flags) as eng:
                eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
                eng.fmx_prepare(p.af)
                eng.fmx_set_clusters(K, clust)
                gc, gs, gf = eng.fmx_iterate(0.5, 0.1, want_full_ll=True)
                assert tuple(gs) == tuple(want[:3])
                d = np.abs(gf - want[3])
                assert np.max(d[np.isfinite(d)], initial=0.0) < 1e-9
                got.append((gc, gf))
        assert np.array_equal(got[0][0]["clust"], got[1][0]["clust"]) and np.array_equal(got[0][0]["type"], got[1][0]["type"])
        d = np.abs(got[0][1] - got[1][1])
        assert np.max(d[np.isfinite(d)], initial=0.0) < 1e-9


def test_linear_entry_loops_of_the_oct_kernels_agree_with_the_general_ones():
    """The oct kernels (V, K <= 16) sweep a chunk's entries with one usable read in a loop of their own (moments of the
    triples, demux_oct.hip / fmx_oct.hip); MUXGL_FLAG_NO_LINEAR_ENTRIES keeps every entry in the nine-term loop.  Mixed
    chunks (reads_lambda = 0.6), alleles other than 0/1 among the reads, several chunks per cell, markers without
    genotypes."""
    V, alphas = 16, (0.0, 0.5)
    p = synth.make_pileup(150, 3000, V, seed=79, mean_entries=400, min_entries=40, reads_lambda=0.6, other=0.05,
                          missing_gp_frac=0.04)
    res = []
    for flags in (0, muxgl.FLAG_NO_LINEAR_ENTRIES):
        with muxgl.Engine(0, flags) as eng:
            eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
            eng.demux_set_gp(p.gp, p.has_gp)
            full = eng.demux_run(alphas, 0.5, want_full_ll=True)
            cells = eng.demux_run(alphas, 0.5)  # the fused finish kernel
            assert np.array_equal(cells["sBest"], full[0]["sBest"]) and np.array_equal(cells["type"], full[0]["type"])
            res.append(full)
    (c1, f1), (c2, f2) = res
    assert np.array_equal(c1["type"], c2["type"]) and np.array_equal(c1["sBest"], c2["sBest"])
    assert np.abs(f1 - f2).max() < 1e-9 and (f1 != f2).any()
    want, wfull = ob.demux(p, alphas=alphas, full_ll=True, nthreads=NT)
    assert parity.compare_full_ll(f1, wfull, V, alphas) < 1e-7 and parity.compare_full_ll(f2, wfull, V, alphas) < 1e-7
    K = 13
    q = synth.make_pileup(120, 3000, 8, seed=80, mean_entries=400, min_entries=40, reads_lambda=0.6, other=0.03, with_gp=False,
                          cap_bq=60)
    clust0 = (np.arange(q.C) % K).astype(np.int32)
    out = []
    for flags in (0, muxgl.FLAG_NO_LINEAR_ENTRIES):
        with muxgl.Engine(0, flags) as eng:
            eng.set_pileup(q.S, q.cell_ptr, q.entry_snp, q.entry_rptr, q.reads)
            eng.fmx_prepare(q.af)
            eng.fmx_set_clusters(K, clust0)
            out.append(eng.fmx_iterate(0.5, 0.1, want_full_ll=True))
    (a, sa, fa), (b, sb, fb) = out
    assert tuple(sa) == tuple(sb) and np.array_equal(a["type"], b["type"]) and np.array_equal(a["clust"], b["clust"])
    assert np.abs(fa - fb).max() < 1e-9 and (fa != fb).any()
    e = ob.fmx_entry_pileup(q)
    cplp = ob.fmx_build_cluster_pileup(q, e, K, clust0)
    ocells = ob.fmx_init_cells(clust0)
    ob.fmx_iterate(q, e, K, cplp, ocells, 0.5, 0.1, nthreads=NT)
    assert parity.compare_fmx(a, ocells)["max_abs_ll_diff"] < 1e-7


@pytest.mark.parametrize("V,alphas", [(16, (0.0, 0.5)), (40, (0.0, 0.3, 0.5))])
def test_high_base_qualities_stay_on_the_nine_term_path(V, alphas):
    """The moment forms of the linear-entry class lose relative accuracy in proportion to the dynamic range of an entry's
    likelihoods, so reads above Q60 (lin_kernel) keep their entries on the nine-term path: with qualities up to 93 the
    log-likelihoods still match the oracle to the usual bar."""
    p = synth.make_pileup(60, 2500, V, seed=81 + V, mean_entries=300, min_entries=40, reads_lambda=0.4, min_bq=30, max_bq=93, cap_bq=93)
    assert (p.reads[p.reads != 0xFF] & 0x7F).max() > 60
    with muxgl.Engine(0) as eng:
        eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        eng.demux_set_gp(p.gp, p.has_gp)
        got, full = eng.demux_run(alphas, 0.5, want_full_ll=True)
    want, wfull = ob.demux(p, alphas=alphas, full_ll=True, nthreads=NT)
    assert parity.compare_full_ll(full, wfull, V, alphas) < 1e-7
    assert parity.compare_demux(got, want, alphas, p)["max_abs_ll_diff"] < 1e-7
