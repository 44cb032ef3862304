"""CPU tests of the N>1 path: world_size-2 runs over torch.distributed/gloo of the sharded demuxlet and freemuxlet
drivers (popscle_amd/demuxlet.py, popscle_amd/freemuxlet.py) and of the shard planner.

There is no GPU here, so the engines are oracle-backed stand-ins that implement exactly the phase interface of
muxgl.Engine (range-restricted E-step, SNP-sharded ordered M-step, exchange buffers).  What is under test is the
orchestration: shard planning, which slices travel, the order of phases and collectives, counter reduction, and the final
gather -- the sharded run must reproduce the single-process oracle run bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
from popscle_amd import demuxlet, freemuxlet, shard, synth


def test_split_by_weight_properties():
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 8):
        w = rng.integers(0, 100, size=57)
        r = shard.split_by_weight(w, n)
        assert len(r) == n and r[0][0] == 0 and r[-1][1] == w.size
        assert all(r[i][1] == r[i + 1][0] for i in range(n - 1))
        sums = [w[b:e].sum() for b, e in r]
        assert max(sums) <= w.sum() / n + w.max()
    assert shard.split_by_weight([], 3) == [(0, 0), (0, 0), (0, 0)]
    assert shard.split_by_weight([5, 5], 4)[-1][1] == 2


def test_equal_ranges_and_slabs():
    for n, w in ((10, 3), (3, 5), (0, 2), (64, 8), (65, 8)):
        r, per = shard.equal_ranges(n, w)
        assert len(r) == w and r[0][0] == 0 and r[-1][1] == n and per * w >= n and per * w < n + w
        assert all(r[i][1] == r[i + 1][0] for i in range(w - 1)) and all(e - b <= per for b, e in r)
    p = synth.make_pileup(40, 300, 3, seed=5, mean_entries=50, min_entries=5, with_gp=False)
    total = 0
    for s0, s1 in shard.equal_ranges(p.S, 4)[0]:
        cp, es, er, rd = shard.take_snps(p, s0, s1)
        assert cp.size == p.C + 1 and cp[-1] == es.size and er[-1] == rd.size
        assert ((es >= s0) & (es < s1)).all()
        for c in (0, 7, 39):  # the slab's run of a cell is that cell's entries in the range, reads attached
            m = (p.entry_snp[p.cell_ptr[c]:p.cell_ptr[c + 1]] >= s0) & (p.entry_snp[p.cell_ptr[c]:p.cell_ptr[c + 1]] < s1)
            idx = np.arange(p.cell_ptr[c], p.cell_ptr[c + 1])[m]
            assert np.array_equal(es[cp[c]:cp[c + 1]], p.entry_snp[idx])
            want = np.concatenate([p.reads[p.entry_rptr[i]:p.entry_rptr[i + 1]] for i in idx]) if len(idx) else np.zeros(0, np.uint8)
            assert np.array_equal(rd[er[cp[c]]:er[cp[c + 1]]], want)
        total += es.size
    assert total == p.nnz


def test_cell_and_snp_shards_cover_everything():
    p = synth.make_pileup(50, 400, 3, seed=2, mean_entries=60, min_entries=5, with_gp=False)
    cr = shard.cell_shards(p.cell_ptr, 4)
    sr = shard.snp_shards(p.entry_snp, p.S, 4)
    assert cr[0][0] == 0 and cr[-1][1] == p.C and sr[0][0] == 0 and sr[-1][1] == p.S
    ent = [p.cell_ptr[e] - p.cell_ptr[b] for b, e in cr]
    assert max(ent) < 1.6 * p.nnz / 4


class FakeDemuxEngine:
    def set_pileup(self, S, cell_ptr, entry_snp, entry_rptr, reads):
        self.args = (S, cell_ptr, entry_snp, entry_rptr, reads)

    def demux_set_gp(self, gp, has_gp):
        self.gp, self.has_gp = gp, has_gp

    def demux_run(self, alphas, doublet_prior):
        S, cp, es, er, rd = self.args
        q = synth.Pileup(cp.size - 1, S, cp, es, er, rd, np.zeros(S), self.gp, self.has_gp)
        return ob.demux(q, alphas=alphas, doublet_prior=doublet_prior)


class FakeFmxEngine:
    """oracle-backed stand-in with the interface of a muxgl.Engine holding a rank's slabs (set_pileup + set_column_slab):
    E-step on the row slab (own cells, every SNP), ordered merge on the column slab (every cell, own SNPs)"""

    def __init__(self, p, c_range, s_range, near_ties=False):
        self.p = p
        self.near_ties = near_ties
        self.C_total, self.S = p.C, p.S
        self.c0, self.c1 = c_range
        self.s0, self.s1 = s_range
        self.cell_base = self.c0
        self.rows = shard.take_cells(p, self.c0, self.c1)
        self.e_rows = ob.fmx_entry_pileup(self.rows)
        cp, es, er, rd = shard.take_snps(p, self.s0, self.s1)
        self.cols = synth.Pileup(p.C, p.S, cp, es, er, rd, p.af)
        self.e_cols = ob.fmx_entry_pileup(self.cols)

    def _mstep(self):
        full = ob.fmx_build_cluster_pileup(self.cols, self.e_cols, self.K, self.clust[:self.C_total])
        self.cplp[:, self.s0:self.s1] = full[:, self.s0:self.s1]

    def fmx_set_clusters(self, K, clust):
        assert clust.shape == (self.C_total,)
        self.K = K
        self.clust = np.concatenate([np.ascontiguousarray(clust, dtype=np.int32), np.full(64, -7, np.int32)])  # + slack
        self.cells = ob.fmx_init_cells(self.clust[self.c0:self.c1].copy())
        self.cplp = np.zeros((K, self.S), dtype=ob.PLP)
        self.cplp["gls"] = np.nan  # rows of foreign SNP ranges must arrive through the exchange
        self.xg = np.full((self.S + 64, K * 9), np.nan)
        self.stat = np.zeros(4, dtype=np.int32)  # the exchange aliases it, as it aliases the library's device buffer
        self._mstep()

    def fmx_iter_gp(self, dp, ge):
        self.xg[self.s0:self.s1] = self.cplp["gls"][:, self.s0:self.s1].transpose(1, 0, 2).reshape(-1, self.K * 9)

    def fmx_iter_estep(self, dp, ge):
        assert not np.isnan(self.xg[:self.S]).any(), "a cluster-GP slice was not exchanged"
        cp = np.zeros((self.K, self.S), dtype=ob.PLP)
        cp["gls"] = self.xg[:self.S].reshape(self.S, self.K, 9).transpose(1, 0, 2)
        ns, na, nch = ob.fmx_iterate(self.rows, self.e_rows, self.K, cp, self.cells, dp, ge)
        self.clust[:] = -1000  # poison: every slice must come back through the exchange
        self.clust[self.c0:self.c1] = self.cells["clust"]
        self.stat[:] = (ns, na, nch, 0)
        if self.near_ties:
            self._list_and_corrupt()
            self.stat_local = self.stat.copy()   # (the exchange all-reduces self.stat in place)

    # ---- the exact path for near-tie calls (include/muxgl.h muxgl_fmx_exact_*; freemuxlet.settle_near_ties).  The
    #      stand-in LISTS some of its cells in every iteration and leaves them with a WRONG call (assignment and counters),
    #      as if rounding noise had decided them; the protocol must bring the reference's back -- which it can only do if
    #      run_em notices the job-wide count, unites the lists, fetches every row from the rank that owns its SNP, and
    #      exchanges the assignments and the counters once more before the ordered merge.
    def _list_and_corrupt(self):
        self.iteration = getattr(self, "iteration", 0) + 1
        ids = np.arange(self.c0, self.c1)
        self.listed = np.flatnonzero((ids + self.iteration) % 5 == 0)
        self.true_cells, self.true_stat = self.cells.copy(), self.stat.copy()
        for i in self.listed:
            self.cells["clust"][i] = -1 if self.cells["clust"][i] >= 0 else 0
            self.cells["type"][i] = 2 if self.cells["clust"][i] < 0 else 0
        self.clust[self.c0:self.c1] = self.cells["clust"]
        self.stat[:] = (self.stat[0] + 3, self.stat[1] + 1, self.stat[2] + 2, len(self.listed))

    def fmx_exact_snps(self):
        r = self.rows
        ent = np.concatenate([np.arange(r.cell_ptr[i], r.cell_ptr[i + 1]) for i in self.listed] + [np.zeros(0, np.int64)])
        return np.unique(r.entry_snp[ent.astype(np.int64)]).astype(np.int32)

    def _row_of(self, gls):  # [K][9] -> [K][3]: stands for the posterior row of a SNP
        return gls[:, [0, 4, 8]]

    def fmx_exact_rows(self, snps, dp, ge):
        rows = np.zeros((len(snps), self.K, 3))
        owned = (snps >= self.s0) & (snps < self.s1)
        for i in np.flatnonzero(owned):
            rows[i] = self._row_of(self.cplp["gls"][:, snps[i]])
        return rows, owned

    def fmx_exact_finish(self, snps, rows, dp, ge):
        assert np.array_equal(snps, np.unique(snps))
        mine = self.fmx_exact_snps()
        assert np.isin(mine, snps).all(), "the united list lacks a SNP of this rank's listed cells"
        xg = self.xg[:self.S].reshape(self.S, self.K, 9)
        for s_ in mine:   # every row must be the one its owner holds (= the one the all-gathered tensor carries)
            assert np.array_equal(rows[np.searchsorted(snps, s_)], self._row_of(xg[s_])), "a row did not come from its owner"
        deltas = self.true_stat[:3].astype(np.int64) - self.stat_local[:3]
        self.cells = self.true_cells
        self.clust[self.c0:self.c1] = self.cells["clust"]
        self.stat[:] = (*self.true_stat[:3], 0)
        return deltas, len(self.listed) > 0

    def fmx_iter_fetch(self, want_cells=True):
        return (self.cells.copy() if want_cells else None), tuple(int(x) for x in self.stat[:3])

    def fmx_iter_mstep(self):
        assert (self.clust[:self.C_total] > -1000).all(), "an assignment slice was not exchanged"
        self.cplp["gls"] = np.nan
        self._mstep()
        self.xg[:] = np.nan


def fake_exchange_tensor(eng, which):
    if which == freemuxlet.UNIT_CGP:
        return torch.from_numpy(eng.xg)
    if which == freemuxlet.UNIT_STAT:
        return torch.from_numpy(eng.stat)
    return torch.from_numpy(eng.clust).view(-1, 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, kind, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = freemuxlet.TorchExchange(dist, rank, world)
    if kind == "demux":
        p = synth.make_pileup(41, 600, 4, seed=33, mean_entries=80, min_entries=5)
        out = demuxlet.run_sharded(FakeDemuxEngine, p, (0.0, 0.5), 0.5, exchange=ex)
        np.save(os.path.join(outdir, f"demux_{rank}.npy"), out)
    else:
        K = 3
        p = synth.make_pileup(60, 500, K, seed=44, mean_entries=120, min_entries=20, with_gp=False)
        (c_ranges, per_c), (s_ranges, per_s) = freemuxlet.plan_ranges(p.C, p.S, world)
        eng = FakeFmxEngine(p, c_ranges[rank], s_ranges[rank], near_ties=(kind == "fmx_ties"))
        e = ob.fmx_entry_pileup(p)
        llk0, llk2, _, _ = ob.fmx_cell_scores(p, e)
        clust0 = ob.fmx_greedy_init(p, e, K, llk2 - llk0, ob.fmx_sort(llk2 - llk0))
        cells, hist = freemuxlet.run_em(eng, K, clust0, 0.5, 0.1, max_iter=6, exchange=ex,
                                        exchange_tensor=fake_exchange_tensor, per=(per_c, per_s))
        np.save(os.path.join(outdir, f"fmx_{rank}.npy"), cells)
        np.save(os.path.join(outdir, f"fmxhist_{rank}.npy"), np.array(hist))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_demuxlet_gloo(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), "demux", str(tmp_path)), nprocs=world, join=True)
    p = synth.make_pileup(41, 600, 4, seed=33, mean_entries=80, min_entries=5)
    want = ob.demux(p)
    for r in range(world):
        got = np.load(tmp_path / f"demux_{r}.npy")
        assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("world,kind", [(2, "fmx"), (3, "fmx"), (2, "fmx_ties"), (3, "fmx_ties")])
def test_sharded_freemuxlet_gloo(tmp_path, world, kind):
    """kind "fmx_ties": every rank lists cells as near ties in every iteration and leaves them wrong; the exact-path
    protocol of run_em (two extra exchanges, assignments and counters again, then the M-step) must restore the run"""
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path)), nprocs=world, join=True)
    K = 3
    p = synth.make_pileup(60, 500, K, seed=44, mean_entries=120, min_entries=20, with_gp=False)
    e = ob.fmx_entry_pileup(p)
    llk0, llk2, _, _ = ob.fmx_cell_scores(p, e)
    clust0 = ob.fmx_greedy_init(p, e, K, llk2 - llk0, ob.fmx_sort(llk2 - llk0))
    cplp = ob.fmx_build_cluster_pileup(p, e, K, clust0)
    cells = ob.fmx_init_cells(clust0)
    hist = []
    for _ in range(6):
        st = ob.fmx_iterate(p, e, K, cplp, cells)
        hist.append(st)
        if st[2] == 0:
            break
    for r in range(world):
        got = np.load(tmp_path / f"fmx_{r}.npy")
        assert got.tobytes() == cells.tobytes()
        assert np.array_equal(np.load(tmp_path / f"fmxhist_{r}.npy"), np.array(hist))
