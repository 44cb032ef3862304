"""GPU tests of the sharded EM phases (multi-GPU path of freemuxlet) on one device: two muxgl handles act as two ranks,
each owning a cell range and a SNP range; the exchanges that RCCL performs between GPUs are done here by device copies
between the two handles' buffers.  The sharded run must reproduce the single-handle muxgl_fmx_iterate bit for bit."""
import numpy as np
import pytest

import oracle_binding as ob
from popscle_amd import freemuxlet, muxgl, shard, synth

pytestmark = pytest.mark.gpu


def prepare(p):
    e = muxgl.Engine(0)
    e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    e.fmx_prepare(p.af)
    return e


def local_allgather(engs, which, ranges, row_bytes):
    for owner, (b, e) in enumerate(ranges):
        if e <= b:
            continue
        src, _ = engs[owner].fmx_buffer(which)
        for r, other in enumerate(engs):
            if r == owner:
                continue
            dst, _ = other.fmx_buffer(which)
            other.memcpy_dev(dst + b * row_bytes, src + b * row_bytes, (e - b) * row_bytes)


@pytest.mark.parametrize("K,world,C", [(4, 2, 240), (16, 3, 240), (20, 2, 240), (64, 2, 60),
                                       (3, 5, 3)])  # more ranks than cells: empty cell shards
def test_virtual_ranks_match_single_handle(K, world, C):
    p = synth.make_pileup(C, 2000 if C > 3 else 40, min(K, 8), seed=60 + K, mean_entries=250 if C > 3 else 12,
                          min_entries=30 if C > 3 else 5, with_gp=False)
    oe = ob.fmx_entry_pileup(p)
    o0, o2, _, _ = ob.fmx_cell_scores(p, oe)
    clust0 = ob.fmx_greedy_init(p, oe, K, o2 - o0, ob.fmx_sort(o2 - o0))

    single = prepare(p)
    single.fmx_set_clusters(K, clust0)
    ref = [single.fmx_iterate(0.5, 0.1) for _ in range(3)]
    ref_gls, ref_cnt = single.fmx_cluster_pileup()
    ref_near = single.fmx_exact_stats()[0]
    single.close()

    engs = [prepare(p) for _ in range(world)]
    c_ranges = shard.cell_shards(p.cell_ptr, world)
    s_ranges = shard.snp_shards(p.entry_snp, p.S, world)
    for r, e in enumerate(engs):
        e.fmx_set_shard(*c_ranges[r], *s_ranges[r])
        e.fmx_set_clusters(K, clust0)
    settled_sharded = 0
    for it in range(3):
        for e in engs:
            e.fmx_iter_gp(0.5, 0.1)
        local_allgather(engs, muxgl.BUF_CGP, s_ranges, K * 3 * 8)
        for e in engs:
            e.fmx_iter_estep(0.5, 0.1)
        fetched = [e.fmx_iter_fetch() for e in engs]
        if sum(e.fmx_exact_pending() for e in engs) > 0:
            # calls within rounding reach of the kernels' numbers (clusters without cells tie exactly in the reference): the
            # exact path across the ranks -- lists united, rows from their owners, every rank settles its own cells
            freemuxlet.settle_near_ties(engs, lambda obj: [obj], 0.5, 0.1)
            fetched = [e.fmx_iter_fetch() for e in engs]
            settled_sharded += 1
        local_allgather(engs, muxgl.BUF_CLUST, c_ranges, 4)
        for e in engs:
            e.fmx_iter_mstep()
        cells = np.zeros(p.C, dtype=muxgl.FMX_CELL)
        stats = np.zeros(3, dtype=np.int64)
        for r, (cs, st) in enumerate(fetched):
            b, en = c_ranges[r]
            cells[b:en] = cs[b:en]
            stats += np.array(st)
        assert cells.tobytes() == ref[it][0].tobytes(), f"iteration {it}: records differ from the single-handle run"
        assert tuple(stats) == tuple(ref[it][1])
    for r, e in enumerate(engs):
        g, c = e.fmx_cluster_pileup()
        b, en = s_ranges[r]
        assert np.array_equal(g[:, b:en], ref_gls[:, b:en]) and np.array_equal(c[:, b:en], ref_cnt[:, b:en])
        assert e.fmx_exact_stats()[2] == 0   # nothing left open
        e.close()
    assert (ref_near > 0) == (settled_sharded > 0)   # the exact path ran on both sides or on neither ...
    if K == 20:
        assert ref_near > 0                           # ... and this shape does have such cells


def test_exchange_tensor_aliases_library_memory():
    """the zero-copy torch view used for the RCCL collectives really is the library's buffer"""
    import torch

    K = 3
    p = synth.make_pileup(50, 400, K, seed=3, mean_entries=80, min_entries=10, with_gp=False)
    e = prepare(p)
    clust0 = (np.arange(p.C) % K).astype(np.int32)
    e.fmx_set_clusters(K, clust0)
    t = freemuxlet.engine_exchange_tensor(e, freemuxlet.UNIT_CLUST)
    assert t.shape == (p.C + muxgl.XCHG_PAD, 1) and np.array_equal(t[:p.C].cpu().numpy().ravel(), clust0)
    e.fmx_iter_gp(0.5, 0.1)
    g = freemuxlet.engine_exchange_tensor(e, freemuxlet.UNIT_CGP)
    torch.cuda.synchronize()
    assert g.shape == (p.S + muxgl.XCHG_PAD, K * 3)
    rows = g[:p.S].cpu().numpy().reshape(p.S, K, 3)
    assert np.allclose(rows.sum(axis=2), 1.0, atol=1e-12)
    # run_em on one rank goes through the same phases and equals iterate()
    cells, hist = freemuxlet.run_em(e, K, clust0, max_iter=3)
    e2 = prepare(p)
    e2.fmx_set_clusters(K, clust0)
    want = None
    for i in range(len(hist)):
        want, st = e2.fmx_iterate(0.5, 0.1)
        assert tuple(st) == tuple(hist[i])
    assert cells.tobytes() == want.tobytes()
    e.close()
    e2.close()
