"""Host-side driver of a sharded demuxlet run: one process per GPU, contiguous cell ranges balanced by entries, GP tensor
replicated, no data-path collective (cmd_cram_demuxlet.cpp:636-1013 has no cross-cell state); the per-cell records are
gathered once at the end.  The reference's own way to parallelise is the same cut at file level (`--group-list`,
README.md:168)."""
from __future__ import annotations

import numpy as np

from . import shard
from .freemuxlet import NoExchange


def run_sharded(engine_factory, p, alphas=(0.0, 0.5), doublet_prior=0.5, exchange=None):
    """engine_factory() -> object with set_pileup / demux_set_gp / demux_run (muxgl.Engine).  Returns the [C] records of
    the whole pileup on every rank, in the original cell order."""
    ex = exchange or NoExchange()
    ranges = shard.cell_shards(p.cell_ptr, ex.world)
    c0, c1 = ranges[ex.rank]
    sub = shard.take_cells(p, c0, c1)
    eng = engine_factory()
    eng.set_pileup(sub.S, sub.cell_ptr, sub.entry_snp, sub.entry_rptr, sub.reads)
    eng.demux_set_gp(p.gp, p.has_gp)
    cells = eng.demux_run(alphas, doublet_prior)
    parts = ex.gather_objects((c0, c1, cells.tobytes()))
    out = np.zeros(p.C, dtype=cells.dtype)
    for b, e, raw in parts:
        out[b:e] = np.frombuffer(raw, dtype=cells.dtype)
    return out
