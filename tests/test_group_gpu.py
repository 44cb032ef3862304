"""GPU tests of the two multi-device layers of the C-ABI, on ONE device (a device may be named more than once):

  * slabs -- a handle holding a rank's row slab (muxgl_set_pileup) and column slab (muxgl_fmx_set_column_slab): several
    such handles act as ranks, the exchanges RCCL performs between GPUs are device copies between their buffers;
  * device groups -- muxgl_config.n_devices > 1: one handle, the same entry points, the library cuts the slabs and
    moves the slices itself (hipMemcpyPeerAsync behind events).

Both must reproduce the one-device, whole-pileup run bit for bit (records, counters, E-step tensor, cluster pileups)."""
import numpy as np
import pytest

import oracle_binding as ob
from popscle_amd import freemuxlet, muxgl, shard, synth

pytestmark = pytest.mark.gpu


def single_run(p, K, clust0, iters=3):
    with muxgl.Engine(0) as e:
        e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        scores = e.fmx_prepare(p.af)
        e.fmx_set_clusters(K, clust0)
        its = [e.fmx_iterate(0.5, 0.1, want_full_ll=True) for _ in range(iters)]
        gls, cnt = e.fmx_cluster_pileup()
    return scores, its, gls, cnt


def start_clusters(p, K):
    rng = np.random.default_rng(K)
    c = np.where(rng.random(p.C) < 0.8, p.truth["s1"] % K, rng.integers(0, K, p.C)).astype(np.int32)
    c[rng.random(p.C) < 0.05] = -1
    return c


def slab_allgather(engs, which, ranges, row_bytes):
    for owner, (b, e) in enumerate(ranges):
        if e <= b:
            continue
        src, _ = engs[owner].fmx_buffer(which)
        for r, other in enumerate(engs):
            if r != owner:
                dst, _ = other.fmx_buffer(which)
                other.memcpy_dev(dst + b * row_bytes, src + b * row_bytes, (e - b) * row_bytes)


@pytest.mark.parametrize("K,world,C", [(4, 2, 240), (16, 3, 240), (20, 2, 200), (64, 2, 60), (3, 5, 3)])
def test_slab_ranks_match_single_handle(K, world, C):
    p = synth.make_pileup(C, 2000 if C > 3 else 40, min(K, 8), seed=160 + K, mean_entries=250 if C > 3 else 12,
                          min_entries=30 if C > 3 else 5, with_gp=False)
    clust0 = start_clusters(p, K)
    scores, ref, ref_gls, ref_cnt = single_run(p, K, clust0)
    (c_ranges, _), (s_ranges, _) = freemuxlet.plan_ranges(p.C, p.S, world)
    engs = []
    for r in range(world):
        e = muxgl.Engine(0)
        got = freemuxlet.load_rank(e, p, c_ranges[r], s_ranges[r])
        b, en = c_ranges[r]
        for g, w in zip(got, scores):  # prepare's per-cell outputs cover the rank's own cells
            assert g.shape == (en - b,) and np.array_equal(g, w[b:en])
        e.fmx_set_clusters(K, clust0)
        engs.append(e)
    with pytest.raises(muxgl.MuxglError, match="sharded"):
        engs[0].fmx_iterate(0.5, 0.1)
    with pytest.raises(muxgl.MuxglError, match="whole pileup"):
        engs[0].fmx_greedy_init(K, np.zeros(engs[0].C))
    for it in range(3):
        for e in engs:
            e.fmx_iter_gp(0.5, 0.1)
        slab_allgather(engs, muxgl.BUF_CGP, s_ranges, K * 3 * 8)
        for e in engs:
            e.fmx_iter_estep(0.5, 0.1)
        fetched = [e.fmx_iter_fetch(want_full_ll=True) for e in engs]
        if sum(e.fmx_exact_pending() for e in engs) > 0:   # near-tie calls: the exact path across the ranks
            freemuxlet.settle_near_ties(engs, lambda obj: [obj], 0.5, 0.1)
            fetched = [e.fmx_iter_fetch(want_full_ll=True) for e in engs]
        slab_allgather(engs, muxgl.BUF_CLUST, c_ranges, 4)
        for e in engs:
            e.fmx_iter_mstep()
        cells = np.concatenate([f[0] for f in fetched])
        full = np.concatenate([f[2] for f in fetched])
        stats = np.sum([f[1] for f in fetched], axis=0)
        assert cells.tobytes() == ref[it][0].tobytes(), f"iteration {it}: records differ from the single-handle run"
        assert tuple(stats) == tuple(ref[it][1])
        assert np.array_equal(full, ref[it][2])
    for r, e in enumerate(engs):
        g, c = e.fmx_cluster_pileup()
        b, en = s_ranges[r]
        assert np.array_equal(g[:, b:en], ref_gls[:, b:en]) and np.array_equal(c[:, b:en], ref_cnt[:, b:en])
        assert not g[:, :b].any() and not g[:, en:].any() and not c[:, :b].any() and not c[:, en:].any()
        e.close()


def test_column_slab_is_validated():
    p = synth.make_pileup(30, 300, 3, seed=9, mean_entries=40, min_entries=5, with_gp=False)
    with muxgl.Engine(0) as e:
        sub = shard.take_cells(p, 10, 20)
        e.set_pileup(p.S, sub.cell_ptr, sub.entry_snp, sub.entry_rptr, sub.reads)
        cp, es, er, rd = shard.take_snps(p, 100, 200)
        with pytest.raises(muxgl.MuxglError, match="do not fit"):
            e.fmx_set_column_slab(p.C, 25, 100, 200, cp, es, er, rd)
        with pytest.raises(muxgl.MuxglError, match="outside the slab"):
            e.fmx_set_column_slab(p.C, 10, 100, 150, cp, es, er, rd)
        e.fmx_set_column_slab(p.C, 10, 100, 200, cp, es, er, rd)
        e.fmx_prepare(p.af)
        with pytest.raises(muxgl.MuxglError, match="slabs"):
            e.fmx_set_shard(0, 5, 0, 10)
        # a new pileup drops the slabs again
        e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        e.fmx_prepare(p.af)
        e.fmx_set_clusters(3, np.zeros(p.C, dtype=np.int32))
        e.fmx_iterate(0.5, 0.1)


@pytest.mark.parametrize("V,alphas,devs", [(16, (0.0, 0.5), [0, 0]), (5, (0.0, 0.1, 0.3, 0.5), [0, 0, 0]),
                                           (40, (0.0, 0.25, 0.5), [0, 0]), (3, (0.0, 0.5), [0] * 7)])
def test_device_group_demuxlet(V, alphas, devs):
    p = synth.make_pileup(5 if len(devs) > 5 else 150, 1500, V, seed=70 + V, mean_entries=150, min_entries=10,
                          missing_gp_frac=0.05)
    with muxgl.Engine(0) as e:
        e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        e.demux_set_gp(p.gp, p.has_gp)
        want, wfull = e.demux_run(alphas, 0.5, want_full_ll=True)
        wpg = e.demux_entry_pg()
    with muxgl.Engine(devs, muxgl.FLAG_DEMUX_ONLY) as g:
        g.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        g.demux_set_gp(p.gp, p.has_gp)
        got, gfull = g.demux_run(alphas, 0.5, want_full_ll=True)
        assert got.tobytes() == want.tobytes() and np.array_equal(gfull, wfull)
        assert g.demux_results_view().tobytes() == want.tobytes()
        assert np.array_equal(g.demux_entry_pg(), wpg)
        assert g.timing()[muxgl.T_DEMUX_SWEEP] > 0
        with pytest.raises(muxgl.MuxglError, match="DEMUX_ONLY"):
            g.fmx_prepare(p.af)


@pytest.mark.parametrize("K,devs,C", [(4, [0, 0], 200), (16, [0, 0, 0], 240), (24, [0, 0], 120), (64, [0, 0, 0, 0], 80),
                                      (3, [0] * 6, 4)])
def test_device_group_freemuxlet(K, devs, C):
    p = synth.make_pileup(C, 2500 if C > 4 else 50, min(K, 8), seed=260 + K, mean_entries=220 if C > 4 else 12,
                          min_entries=30 if C > 4 else 5, with_gp=False)
    clust0 = start_clusters(p, K)
    scores, ref, ref_gls, ref_cnt = single_run(p, K, clust0)
    with muxgl.Engine(0) as e:
        e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        e.fmx_prepare(p.af)
        wgls, wcnt = e.fmx_entry_gls()
    with muxgl.Engine(devs) as g:
        g.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        got = g.fmx_prepare(p.af)
        for a, b in zip(got, scores):
            assert np.array_equal(a, b)
        ggls, gcnt = g.fmx_entry_gls()
        assert np.array_equal(ggls, wgls) and np.array_equal(gcnt, wcnt)
        with pytest.raises(muxgl.MuxglError, match="device group"):
            g.fmx_greedy_init(K, got[1] - got[0])
        with pytest.raises(muxgl.MuxglError, match="device group"):
            g.fmx_set_shard(0, 1, 0, 1)
        g.fmx_set_clusters(K, clust0)
        for it in range(3):
            cells, stats, full = g.fmx_iterate(0.5, 0.1, want_full_ll=True)
            assert cells.tobytes() == ref[it][0].tobytes(), f"iteration {it}: records differ from the one-device run"
            assert tuple(stats) == tuple(ref[it][1]) and np.array_equal(full, ref[it][2])
        gls, cnt = g.fmx_cluster_pileup()
        assert np.array_equal(gls, ref_gls) and np.array_equal(cnt, ref_cnt)
        tm = g.timing()
        assert tm[muxgl.T_FMX_ESTEP] > 0 and tm[muxgl.T_FMX_MSTEP] > 0
        # the same job again on the same group: hand-over is repeatable
        g.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        g.fmx_prepare(p.af)
        g.fmx_set_clusters(K, clust0)
        cells, stats = g.fmx_iterate(0.5, 0.1)
        assert cells.tobytes() == ref[0][0].tobytes()


def test_device_group_against_oracle():
    """the group is also checked against the oracle directly (not only against the one-device run)"""
    K = 6
    p = synth.make_pileup(90, 1200, K, seed=11, mean_entries=150, min_entries=20, with_gp=False)
    e = ob.fmx_entry_pileup(p)
    o0, o2, _, _ = ob.fmx_cell_scores(p, e)
    clust0 = ob.fmx_greedy_init(p, e, K, o2 - o0, ob.fmx_sort(o2 - o0))
    cplp = ob.fmx_build_cluster_pileup(p, e, K, clust0)
    ocells = ob.fmx_init_cells(clust0)
    import parity

    with muxgl.Engine([0, 0, 0]) as g:
        g.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        g.fmx_prepare(p.af)
        g.fmx_set_clusters(K, clust0)
        for it in range(3):
            ostats = ob.fmx_iterate(p, e, K, cplp, ocells, 0.5, 0.1)
            gcells, gstats = g.fmx_iterate(0.5, 0.1)
            assert tuple(gstats) == tuple(ostats)
            assert parity.compare_fmx(gcells, ocells)["max_abs_ll_diff"] < 1e-7   # (a group settles its near ties like one device)


def test_device_group_settles_near_tie_calls():
    """clusters without cells and very shallow droplets (exact and near ties in the reference): the group's members unite
    their lists, compute the rows of their own SNP ranges and settle their own cells -- the same records, counters and
    cluster pileups as the oracle and as one device, over whole trajectories"""
    import parity

    K, used = 12, 4
    p = synth.make_pileup(350, 500, used, seed=21, mean_entries=7, min_entries=1, with_gp=False)
    rng = np.random.default_rng(2)
    clust0 = rng.integers(0, used, p.C).astype(np.int32)
    clust0[rng.random(p.C) < 0.1] = -1
    e = ob.fmx_entry_pileup(p)
    cplp = ob.fmx_build_cluster_pileup(p, e, K, clust0)
    ocells = ob.fmx_init_cells(clust0)
    with muxgl.Engine([0, 0, 0, 0]) as g, muxgl.Engine(0) as one:
        for en in (g, one):
            en.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
            en.fmx_prepare(p.af)
            en.fmx_set_clusters(K, clust0)
        for it in range(3):
            ostats = ob.fmx_iterate(p, e, K, cplp, ocells, 0.5, 0.1)
            gcells, gstats = g.fmx_iterate(0.5, 0.1)
            ocell1, ostat1 = one.fmx_iterate(0.5, 0.1)
            assert tuple(gstats) == tuple(ostats) == tuple(ostat1), it
            parity.compare_fmx(gcells, ocells)
            assert gcells.tobytes() == ocell1.tobytes()
        near, changed, unresolved = g.fmx_exact_stats()
        assert near > 0 and unresolved == 0 and (near, changed) == one.fmx_exact_stats()[:2]
        gg, gc = g.fmx_cluster_pileup()
        og, oc = one.fmx_cluster_pileup()
        assert np.array_equal(gg, og) and np.array_equal(gc, oc)


def test_group_peer_access_walk(monkeypatch):
    """muxgl_create's walk over the member pairs (muxgl_group.hip): on a one-GPU box the members share the device and the
    walk has nothing to do; MUXGL_FLAG_GROUP_PROBE_SELF makes it ask for every pair, which the runtime refuses -- the
    refusal must leave no sticky error behind and the group must work (its copies then go the runtime's staged way);
    MUXGL_GROUP_NO_PEER=1 skips the enabling on any box.  On a multi-GPU box the same test also sees pairs enabled."""
    import torch

    ndev = torch.cuda.device_count()
    K = 5
    p = synth.make_pileup(80, 900, K, seed=31, mean_entries=100, min_entries=10, with_gp=False)
    clust0 = (np.arange(p.C) % K).astype(np.int32)
    with muxgl.Engine(0) as one:
        one.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        one.fmx_prepare(p.af)
        one.fmx_set_clusters(K, clust0)
        want = [one.fmx_iterate(0.5, 0.1) for _ in range(2)]
    devs = [i % ndev for i in range(3)]
    for env, flags in ((None, 0), (None, muxgl.FLAG_GROUP_PROBE_SELF), ("1", muxgl.FLAG_GROUP_PROBE_SELF)):
        if env:
            monkeypatch.setenv("MUXGL_GROUP_NO_PEER", env)
        with muxgl.Engine(devs, flags) as g:
            pairs, enabled, refused = g.group_peer_stats()
            distinct = sum(1 for a in range(3) for b in range(3) if a != b and devs[a] != devs[b])
            assert pairs == (6 if flags else distinct) and enabled + refused == pairs
            if env:
                assert enabled == 0
            elif ndev == 1:
                assert refused == pairs
            g.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
            g.fmx_prepare(p.af)
            g.fmx_set_clusters(K, clust0)
            for it in range(2):
                cells, st = g.fmx_iterate(0.5, 0.1)
                assert cells.tobytes() == want[it][0].tobytes() and tuple(st) == tuple(want[it][1])


def test_group_create_errors():
    with pytest.raises(muxgl.MuxglError, match="out of range"):
        muxgl.Engine([0, 99])
    with pytest.raises(ValueError):
        muxgl.Engine([0] * 17)


def test_slab_ranks_loaded_from_files_match_single_handle(tmp_path):
    """The ranks' slabs cut by the C++ loader while it parses the .plp.gz (popscle-amd dump-plp --rank r --world N,
    freemuxlet.load_rank_from_files): three EM iterations equal the one-handle run on the pileup loaded whole."""
    import os

    from popscle_amd import plpio

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "popscle_amd", "bin", "popscle-amd")
    if not os.path.exists(exe):
        pytest.skip("popscle-amd not built")
    K, world = 5, 3
    src = synth.make_pileup(150, 1500, K, seed=77, mean_entries=200, min_entries=20, reads_lambda=1.0, with_gp=False)
    prefix = str(tmp_path / "plp")
    plpio.write_plp(prefix, src, seed=5)
    whole = plpio.read_dump(_dump(exe, prefix, str(tmp_path / "whole.bin")))
    p = synth.Pileup(whole["C"], whole["S"], whole["cell_ptr"], whole["entry_snp"], whole["entry_rptr"], whole["reads"],
                     whole["af"], truth=src.truth)
    clust0 = start_clusters(p, K)
    scores, ref, ref_gls, ref_cnt = single_run(p, K, clust0)
    (c_ranges, _), (s_ranges, _) = freemuxlet.plan_ranges(p.C, p.S, world)
    engs = []
    for r in range(world):
        e = muxgl.Engine(0)
        got, d = freemuxlet.load_rank_from_files(e, exe, prefix, r, world, str(tmp_path / f"slab{r}.bin"))
        b, en = c_ranges[r]
        assert (d["c0"], d["c1"]) == (b, en)
        for g, w in zip(got, scores):
            assert np.array_equal(g, w[b:en])
        e.fmx_set_clusters(K, clust0)
        engs.append(e)
    for it in range(3):
        for e in engs:
            e.fmx_iter_gp(0.5, 0.1)
        slab_allgather(engs, muxgl.BUF_CGP, s_ranges, K * 3 * 8)
        for e in engs:
            e.fmx_iter_estep(0.5, 0.1)
        fetched = [e.fmx_iter_fetch() for e in engs]
        if sum(e.fmx_exact_pending() for e in engs) > 0:   # near-tie calls: the exact path across the ranks
            freemuxlet.settle_near_ties(engs, lambda obj: [obj], 0.5, 0.1)
            fetched = [e.fmx_iter_fetch() for e in engs]
        slab_allgather(engs, muxgl.BUF_CLUST, c_ranges, 4)
        for e in engs:
            e.fmx_iter_mstep()
        cells = np.concatenate([f[0] for f in fetched])
        assert cells.tobytes() == ref[it][0].tobytes(), f"iteration {it}"
        assert tuple(np.sum([f[1] for f in fetched], axis=0)) == tuple(ref[it][1])
    for e in engs:
        e.close()


def _dump(exe, prefix, out):
    import subprocess

    r = subprocess.run([exe, "dump-plp", "--plp", prefix, "--out", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out
