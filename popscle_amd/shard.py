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


def equal_ranges(n: int, world: int):
    """world contiguous ranges of ceil(n / world) units (the last ones shorter or empty): slices of equal size can be
    all-gathered in place, as ONE collective, on a buffer with a little slack behind it (muxgl.XCHG_PAD units).  Cells
    arrive in barcode order and SNPs in genome order, neither sorted by depth, so equal counts are balanced in entries
    to within a percent at the sizes where balance matters (relative spread ~ 0.7 / sqrt(units per rank))."""
    per = -(-n // world) if world > 0 and n > 0 else 0
    return [(min(r * per, n), min((r + 1) * per, n)) for r in range(world)], per


def take_cells(p, c0: int, c1: int):
    """the pileup of cells [c0, c1) (same SNP axis / GP tensor): a rank's row slab -- what a demuxlet rank uploads, and
    what a freemuxlet rank runs its E-step on"""
    return p.subset_cells(np.arange(c0, c1, dtype=np.int64))


def take_snps(p, s0: int, s1: int):
    """(cell_ptr[C+1], entry_snp, entry_rptr, reads) of the entries with s0 <= SNP < s1, all cells, SNP ids unchanged: a
    rank's column slab (muxgl_fmx_set_column_slab).  A cell's entries ascend by SNP, so the slab is a run per cell."""
    from .synth import _ranges

    keep = (p.entry_snp >= s0) & (p.entry_snp < s1)
    cell_of = np.repeat(np.arange(p.C, dtype=np.int64), np.diff(p.cell_ptr))
    cell_ptr = np.zeros(p.C + 1, dtype=np.int64)
    np.cumsum(np.bincount(cell_of[keep], minlength=p.C), out=cell_ptr[1:])
    eidx = np.flatnonzero(keep)
    rl = p.entry_rptr[eidx + 1] - p.entry_rptr[eidx]
    entry_rptr = np.zeros(eidx.size + 1, dtype=np.int64)
    np.cumsum(rl, out=entry_rptr[1:])
    reads = p.reads[_ranges(p.entry_rptr[eidx], rl)]
    return cell_ptr, np.ascontiguousarray(p.entry_snp[eidx]), entry_rptr, np.ascontiguousarray(reads)
