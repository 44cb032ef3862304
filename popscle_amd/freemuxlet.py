"""Host-side driver of the freemuxlet EM loop -- the sequential control flow of cmdCramFreemux2
(cmd_cram_freemux2.cpp:373-605) around the libmuxgl phases, for one GPU or for one process per GPU.

Multi-GPU scheme (exact): every rank holds the whole packed pileup and entry likelihoods;
rank r owns the cell range c_ranges[r] (E-step, scans, re-assignment) and the SNP range s_ranges[r] (cluster-GP rows,
ordered M-step).  Per iteration:

    iter_gp     -> all-gather of the cluster-GP rows   f64[S][K][3]   (38 MB at config 3, 768 MB at config 4)
    iter_estep  -> all-gather of the assignments       i32[C]         + all-reduce of (nsingle, namb, nchanged)
    iter_mstep

The collectives run over RCCL (torch.distributed backend "nccl") directly on the library's device buffers; xGMI is
point-to-point, so each rank's slice is sent as one broadcast per owner (N large messages, no small-bucket traffic).
An all-reduce of sufficient statistics would NOT reproduce the reference: merge() clamps after every cell
(sc_drop_seq.h:92-100), so the chain per (cluster, SNP) is evaluated by exactly one rank, in ascending cell id.
"""
from __future__ import annotations

import numpy as np

from . import shard

UNIT_CGP, UNIT_CLUST = 0, 1


class TorchExchange:
    """Collectives over torch.distributed on tensors that alias the engine's exchange buffers."""

    def __init__(self, dist_module, rank: int, world: int):
        self.dist = dist_module
        self.rank = rank
        self.world = world

    def allgather_rows(self, tensor, ranges):
        """tensor: [n_units, row]; ranges[r] = (b, e) unit range owned by rank r.  One broadcast per owner.
        Returns when the data has landed: the engine launches its kernels on a stream of its own, which does not
        wait for torch's streams (a collective that has merely been enqueued would be raced by the next phase)."""
        for r, (b, e) in enumerate(ranges):
            if e > b:
                self.dist.broadcast(tensor[b:e], src=r)
        self._landed(tensor)

    def allreduce_sum(self, tensor):
        self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM)
        self._landed(tensor)

    @staticmethod
    def _landed(tensor):
        if getattr(tensor, "is_cuda", False):
            import torch

            torch.cuda.current_stream(tensor.device).synchronize()

    def gather_objects(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


class NoExchange:
    rank, world = 0, 1

    def allgather_rows(self, tensor, ranges):
        pass

    def allreduce_sum(self, tensor):
        pass

    def gather_objects(self, obj):
        return [obj]


def engine_exchange_tensor(eng, which, device_index=0):
    """torch tensor aliasing a libmuxgl device buffer (zero-copy, __cuda_array_interface__): CGP -> f64[S, K*3],
    CLUST -> i32[C, 1]"""
    import torch

    from . import muxgl

    buf = muxgl.BUF_CGP if which == UNIT_CGP else muxgl.BUF_CLUST
    ptr, n = eng.fmx_buffer(buf)
    typestr, dtype = ("<f8", torch.float64) if which == UNIT_CGP else ("<i4", torch.int32)

    class _Wrap:
        __cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}

    t = torch.as_tensor(_Wrap(), device=torch.device("cuda", device_index))
    assert t.dtype == dtype and t.data_ptr() == ptr
    return t.view(eng.S, eng.K * 3) if which == UNIT_CGP else t.view(eng.C, 1)


def run_em(eng, K, clust0, cell_ptr, entry_snp, doublet_prior=0.5, geno_error=0.1, max_iter=10, early_stop=True,
           exchange=None, exchange_tensor=engine_exchange_tensor, log=None, ranges=None, timings=None, sync=None):
    """EM loop of cmd_cram_freemux2.cpp:373-605 after muxgl_fmx_prepare.  Returns (cells[C] (complete on every rank),
    per-iteration stats).  `eng` needs the fmx_* phase methods of muxgl.Engine; `exchange` a TorchExchange/NoExchange."""
    import time

    ex = exchange or NoExchange()
    C, S = eng.C, eng.S
    t_start = time.perf_counter()
    # ranges = (cell ranges, SNP ranges) planned earlier (shard.cell_shards / snp_shards walk all entries)
    c_ranges, s_ranges = ranges if ranges is not None else (shard.cell_shards(cell_ptr, ex.world),
                                                            shard.snp_shards(entry_snp, S, ex.world))
    c0, c1 = c_ranges[ex.rank]
    s0, s1 = s_ranges[ex.rank]
    eng.fmx_set_shard(c0, c1, s0, s1)
    eng.fmx_set_clusters(K, np.ascontiguousarray(clust0, dtype=np.int32))  # :277-288, own SNP shard
    t_cgp = exchange_tensor(eng, UNIT_CGP) if ex.world > 1 else None
    t_clust = exchange_tensor(eng, UNIT_CLUST) if ex.world > 1 else None
    history = []
    if sync:
        sync()
    t_loop = time.perf_counter()
    for it in range(max_iter):
        eng.fmx_iter_gp(doublet_prior, geno_error)
        if ex.world > 1:
            ex.allgather_rows(t_cgp, s_ranges)
        eng.fmx_iter_estep(doublet_prior, geno_error)
        _, stats = eng.fmx_iter_fetch(want_cells=False)  # three counters; the records are fetched once, below
        if ex.world > 1:
            ex.allgather_rows(t_clust, c_ranges)
            import torch

            st = torch.tensor(list(stats), dtype=torch.int64, device=t_clust.device)
            ex.allreduce_sum(st)
            stats = tuple(int(x) for x in st.tolist())
        eng.fmx_iter_mstep()  # :516-517 + :590-596 for the own SNP shard
        history.append(stats)
        if log:
            log(f"iter {it + 1}: {stats[0]} singlets, {C - stats[0] - stats[1]} doublets, {stats[1]} ambiguous, "
                f"{stats[2]} changed")
        if stats[2] == 0 and early_stop:  # :601-604
            break
    if sync:
        sync()
    t_end = time.perf_counter()
    if timings is not None:  # `sync` (a barrier) makes these comparable across ranks
        timings.update(setup_s=t_loop - t_start, loop_s=t_end - t_loop, iterations=len(history))
    cells, _ = eng.fmx_iter_fetch()
    parts = ex.gather_objects((c0, c1, cells[c0:c1].tobytes()))
    out = np.zeros(C, dtype=cells.dtype)
    for b, e, raw in parts:
        out[b:e] = np.frombuffer(raw, dtype=cells.dtype)
    return out, history
