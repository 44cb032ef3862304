"""GPU parity at the shapes of BASELINE.json configs[2] (demuxlet, 64 samples, six alphas, 200 k SNPs) and configs[3]
(freemuxlet, 16 clusters, 100 k SNPs) with a reduced number of cells, so that the CPU oracle can still check a sample /
the whole trajectory in seconds.  Cells are independent in demuxlet and enter freemuxlet's E-step independently, so the
per-cell arithmetic exercised here is the full-size one (same V / K / A / SNP axis / entry density)."""
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
        rep = parity.compare_demux(cells[pick], want, alphas, want_full=wfull)
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
