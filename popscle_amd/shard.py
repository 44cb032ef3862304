"""Shard planning for the multi-GPU runs.

demuxlet: cells are independent given the GP tensor (cmd_cram_demuxlet.cpp:636-1013 carries no cross-cell state), so
the cell axis is cut into contiguous ranges balanced by ENTRY count (the work unit of the sweep), the GP tensor is
replicated, and there is no data-path collective.
freemuxlet: the E-step/scans are sharded by cells the same way; the ordered M-step and the cluster-GP rows are sharded
by SNPs, balanced by the number of entries per SNP (the length of the merge chains).
"""
from __future__ import annotations

import numpy as np


def split_by_weight(weights, n: int):
    """n contiguous ranges [b_i, e_i) over len(weights) items with near-equal weight sums (prefix-sum cuts)."""
    w = np.asarray(weights, dtype=np.float64)
    m = w.size
    if n <= 1 or m == 0:
        return [(0, m)] + [(m, m)] * (max(n, 1) - 1)
    cum = np.concatenate(([0.0], np.cumsum(w)))
    total = cum[-1]
    cuts = [0]
    for i in range(1, n):
        target = total * i / n
        c = int(np.searchsorted(cum, target, side="left"))
        c = min(max(c, cuts[-1]), m)
        cuts.append(c)
    cuts.append(m)
    return [(cuts[i], cuts[i + 1]) for i in range(n)]


def cell_shards(cell_ptr, n: int):
    """contiguous cell ranges balanced by entries per cell"""
    return split_by_weight(np.diff(np.asarray(cell_ptr, dtype=np.int64)), n)


def snp_shards(entry_snp, S: int, n: int):
    """contiguous SNP ranges balanced by entries per SNP (+1 so that uncovered SNPs still spread)"""
    cov = np.bincount(np.asarray(entry_snp, dtype=np.int64), minlength=S).astype(np.float64) + 1.0
    return split_by_weight(cov, n)


def take_cells(p, c0: int, c1: int):
    """the pileup of cells [c0, c1) (same SNP axis / GP tensor): what a demuxlet rank uploads"""
    return p.subset_cells(np.arange(c0, c1, dtype=np.int64))
