"""Synthetic packed pileups of the shapes BASELINE.json names.

Generator family: AF_s ~ U(0.05,0.95); genotypes G[s,v] ~ Binomial(2,AF_s); entries per cell
L_c = clip(round(LogNormal(ln 800, 0.6)), 50, 8000) SNPs (duplicates within a cell removed); reads per entry =
1 + Poisson(0.3); raw base quality ~ UniformInt[13,40], capped at cap_bq (20, the reference default
cmd_cram_demuxlet.cpp:17); a fraction of droplets are doublets mixing two samples at alpha = 0.5; the observed allele
is Bernoulli(g/2) of the source sample's genotype with 1 % flips and 0.5 % "other" alleles.  The GP tensor is what
load_from_plp builds from hard GT calls: one-hot (through float) then (1-err)*gp + err*avgGP with err = 0.1
(sc_drop_seq.cpp:287-315).

Everything is numpy and seeded; the same call yields the same bytes on every machine.  make_pileup_device() is the
same generator family written in torch for a GPU (bench.py's large configs: 10^8 entries take minutes in numpy and a
second on the device); it is seeded too, but its bytes are torch's, not numpy's.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

READ_OTHER = 0xFF
BASE_SEED = 20240901


@dataclass
class Pileup:
    C: int
    S: int
    cell_ptr: np.ndarray    # int64[C+1]
    entry_snp: np.ndarray   # int32[nnz]
    entry_rptr: np.ndarray  # int64[nnz+1]
    reads: np.ndarray       # uint8[R]
    af: np.ndarray          # float64[S]
    gp: np.ndarray | None = None       # float64[S][V][3]
    has_gp: np.ndarray | None = None   # uint8[S]
    truth: dict = field(default_factory=dict)

    @property
    def nnz(self) -> int:
        return int(self.entry_snp.size)

    @property
    def R(self) -> int:
        return int(self.reads.size)

    def subset_cells(self, cells) -> "Pileup":
        """Pileup restricted to the given cells (in the given order); SNP axis and GP tensor unchanged."""
        cells = np.asarray(cells, dtype=np.int64)
        lens = self.cell_ptr[cells + 1] - self.cell_ptr[cells]
        cell_ptr = np.zeros(cells.size + 1, dtype=np.int64)
        np.cumsum(lens, out=cell_ptr[1:])
        eidx = _ranges(self.cell_ptr[cells], lens)
        entry_snp = self.entry_snp[eidx]
        rl = self.entry_rptr[eidx + 1] - self.entry_rptr[eidx]
        entry_rptr = np.zeros(eidx.size + 1, dtype=np.int64)
        np.cumsum(rl, out=entry_rptr[1:])
        ridx = _ranges(self.entry_rptr[eidx], rl)
        truth = {k: (v[cells] if isinstance(v, np.ndarray) and v.shape[:1] == (self.C,) else v)
                 for k, v in self.truth.items()}
        return Pileup(int(cells.size), self.S, cell_ptr, entry_snp, entry_rptr, self.reads[ridx], self.af, self.gp,
                      self.has_gp, truth)


def _ranges(starts, lens):
    """concatenate arange(s, s+l) for every (s, l)"""
    lens = np.asarray(lens, dtype=np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    offs = np.repeat(np.asarray(starts, dtype=np.int64) - np.concatenate(([0], np.cumsum(lens)[:-1])), lens)
    return offs + np.arange(total, dtype=np.int64)


def gt_to_gp(G: np.ndarray, err: float = 0.1) -> np.ndarray:
    """GP rows as sc_drop_seq.cpp:287-315 builds them from hard calls (--field GT): gp float one-hot -> double,
    avgGP[g] = (1e-10 + sum_v gp[v,g]) / sum, gp = (1-err)*gp + err*avgGP[g].  G is int[S][V] in {0,1,2}."""
    S, V = G.shape
    gp = np.zeros((S, V, 3), dtype=np.float64)
    np.put_along_axis(gp, G[:, :, None].astype(np.int64), np.float64(np.float32(1.0)), axis=2)
    avg = np.full((S, 3), 1e-10)
    for v in range(V):  # sequential accumulation order of the reference loop
        avg += gp[:, v, :]
    avg /= (avg[:, 0] + avg[:, 1] + avg[:, 2])[:, None]
    if err > 0:
        gp = (1 - err) * gp + err * avg[:, None, :]
    return gp


def make_pileup(C: int, S: int, V: int, seed: int = BASE_SEED, mean_entries: float = 800.0, sigma: float = 0.6,
                min_entries: int = 50, max_entries: int = 8000, reads_lambda: float = 0.3, doublet_frac: float = 0.10,
                flip: float = 0.01, other: float = 0.005, min_bq: int = 13, max_bq: int = 40, cap_bq: int = 20,
                geno_err: float = 0.1, missing_gp_frac: float = 0.0, with_gp: bool = True,
                donor_seed: int | None = None) -> Pileup:
    """Seeded synthetic pileup.  V = number of true donors (samples / clusters).  donor_seed (default: seed) fixes the
    SNP panel and donor genotypes independently of the cells, so that several cell shards can share one GP tensor."""
    rng_d = np.random.default_rng(seed if donor_seed is None else donor_seed)
    af = rng_d.uniform(0.05, 0.95, size=S)
    G = rng_d.binomial(2, af[:, None], size=(S, V)).astype(np.int8)
    rng = np.random.default_rng([seed, 1])

    max_entries = min(max_entries, S)
    min_entries = min(min_entries, max_entries)
    L = np.clip(np.rint(rng.lognormal(np.log(mean_entries), sigma, size=C)), min_entries, max_entries).astype(np.int64)
    # distinct SNPs per cell: draw with replacement, drop duplicates (keeps the lengths within ~L^2/2S of L)
    cell_of = np.repeat(np.arange(C, dtype=np.int64), L)
    snp = rng.integers(0, S, size=cell_of.size, dtype=np.int64)
    key = np.unique(cell_of * S + snp)  # sorted by (cell, snp)
    cell_of = key // S
    entry_snp = (key % S).astype(np.int32)
    counts = np.bincount(cell_of, minlength=C).astype(np.int64)
    cell_ptr = np.zeros(C + 1, dtype=np.int64)
    np.cumsum(counts, out=cell_ptr[1:])
    nnz = entry_snp.size

    nreads = 1 + rng.poisson(reads_lambda, size=nnz).astype(np.int64)
    entry_rptr = np.zeros(nnz + 1, dtype=np.int64)
    np.cumsum(nreads, out=entry_rptr[1:])
    R = int(entry_rptr[-1])
    read_entry = np.repeat(np.arange(nnz, dtype=np.int64), nreads)
    read_cell = cell_of[read_entry]
    read_snp = entry_snp[read_entry]

    is_dbl = rng.random(C) < doublet_frac
    s1 = rng.integers(0, V, size=C)
    s2 = (s1 + 1 + rng.integers(0, max(V - 1, 1), size=C)) % V if V > 1 else s1.copy()
    pick2 = is_dbl[read_cell] & (rng.random(R) < 0.5)
    src = np.where(pick2, s2[read_cell], s1[read_cell])
    g = G[read_snp, src].astype(np.float64)
    allele = (rng.random(R) < g / 2.0).astype(np.uint8)
    allele ^= (rng.random(R) < flip).astype(np.uint8)
    bq = np.minimum(rng.integers(min_bq, max_bq + 1, size=R), cap_bq).astype(np.uint8)
    reads = (allele << 7) | bq
    reads[rng.random(R) < other] = READ_OTHER

    gp = has_gp = None
    if with_gp:
        gp = gt_to_gp(G.astype(np.int64), geno_err)
        has_gp = np.ones(S, dtype=np.uint8)
        if missing_gp_frac > 0:
            has_gp[rng_d.random(S) < missing_gp_frac] = 0
    truth = {"is_doublet": is_dbl, "s1": s1.astype(np.int32), "s2": s2.astype(np.int32), "G": G}
    return Pileup(C, S, cell_ptr, entry_snp, entry_rptr, reads.astype(np.uint8), af, gp, has_gp, truth)


# BASELINE.json configs (index = position in "configs")
CONFIGS = {
    1: dict(C=10_000, S=50_000, V=16, alphas=(0.0, 0.5)),
    2: dict(C=100_000, S=200_000, V=64, alphas=(0.0, 0.1, 0.2, 0.3, 0.4, 0.5)),
    3: dict(C=50_000, S=100_000, V=16),
    4: dict(C=500_000, S=500_000, V=64),
}


def make_config(index: int, scale: float = 1.0, **kw) -> Pileup:
    """Pileup of BASELINE.json configs[index]; scale<1 shrinks the cell count only (same S, V, density)."""
    cfg = CONFIGS[index]
    C = max(1, int(round(cfg["C"] * scale)))
    return make_pileup(C, cfg["S"], cfg["V"], seed=BASE_SEED + index, **kw)


# ---- the same generator family on a GPU (torch): bench inputs of the large configs --------------------------------
class DevicePileup:
    """A synthetic packed pileup held as torch tensors on a device (same fields as Pileup).  host() copies it to numpy;
    take_cells / take_snps cut a rank's row / column slab on the device, so a rank of a sharded run moves only its
    2/N of the job to the host (shard.take_cells / shard.take_snps are the numpy counterparts)."""

    def __init__(self, C, S, cell_ptr, entry_snp, entry_rptr, reads, af, gp, has_gp, truth):
        self.C, self.S = int(C), int(S)
        self.cell_ptr, self.entry_snp, self.entry_rptr, self.reads = cell_ptr, entry_snp, entry_rptr, reads
        self.af, self.gp, self.has_gp, self.truth = af, gp, has_gp, truth

    @property
    def nnz(self):
        return int(self.entry_snp.numel())

    @property
    def R(self):
        return int(self.reads.numel())

    @staticmethod
    def _np(t):
        return None if t is None else t.cpu().numpy()

    def host(self) -> Pileup:
        n = self._np
        return Pileup(self.C, self.S, n(self.cell_ptr), n(self.entry_snp), n(self.entry_rptr), n(self.reads), n(self.af),
                      n(self.gp), n(self.has_gp), {k: n(v) for k, v in self.truth.items()})

    def _cut(self, keep_entries, cell_counts):
        """(cell_ptr, entry_snp, entry_rptr, reads) of the entries picked by the boolean mask, as numpy"""
        import torch

        eidx = torch.nonzero(keep_entries, as_tuple=False).flatten()
        cell_ptr = torch.zeros(cell_counts.numel() + 1, dtype=torch.int64, device=eidx.device)
        torch.cumsum(cell_counts, 0, out=cell_ptr[1:])
        r0 = self.entry_rptr[eidx]
        rl = self.entry_rptr[eidx + 1] - r0
        entry_rptr = torch.zeros(eidx.numel() + 1, dtype=torch.int64, device=eidx.device)
        torch.cumsum(rl, 0, out=entry_rptr[1:])
        ridx = torch.repeat_interleave(r0 - entry_rptr[:-1], rl) + torch.arange(int(entry_rptr[-1]), device=eidx.device)
        n = self._np
        return n(cell_ptr), n(self.entry_snp[eidx]), n(entry_rptr), n(self.reads[ridx])

    def take_cells(self, c0, c1) -> Pileup:
        """row slab: cells [c0, c1) with every SNP, cells renumbered from 0"""
        import torch

        e0, e1 = int(self.cell_ptr[c0]), int(self.cell_ptr[c1])
        keep = torch.zeros(self.nnz, dtype=torch.bool, device=self.entry_snp.device)
        keep[e0:e1] = True
        cp, es, er, rd = self._cut(keep, self.cell_ptr[c0 + 1:c1 + 1] - self.cell_ptr[c0:c1])
        return Pileup(c1 - c0, self.S, cp, es, er, rd, self._np(self.af), self._np(self.gp), self._np(self.has_gp), {})

    def take_snps(self, s0, s1):
        """column slab: every cell, the entries with s0 <= SNP < s1 (arguments of muxgl_fmx_set_column_slab)"""
        import torch

        keep = (self.entry_snp >= s0) & (self.entry_snp < s1)
        csum = torch.zeros(self.nnz + 1, dtype=torch.int64, device=keep.device)
        torch.cumsum(keep, 0, out=csum[1:])
        counts = csum[self.cell_ptr[1:]] - csum[self.cell_ptr[:-1]]
        return self._cut(keep, counts)


def make_pileup_device(C: int, S: int, V: int, seed: int = BASE_SEED, device="cuda", mean_entries: float = 800.0,
                       sigma: float = 0.6, min_entries: int = 50, max_entries: int = 8000, reads_lambda: float = 0.3,
                       doublet_frac: float = 0.10, flip: float = 0.01, other: float = 0.005, min_bq: int = 13,
                       max_bq: int = 40, cap_bq: int = 20, geno_err: float = 0.1, with_gp: bool = True,
                       donor_seed: int | None = None) -> DevicePileup:
    """make_pileup() on a torch device: same distributions and the same GP rule, torch's random streams."""
    import torch

    dev = torch.device(device)
    gd = torch.Generator(device=dev)
    gd.manual_seed(int(seed if donor_seed is None else donor_seed))
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed) * 7919 + 1)
    f64, i64 = torch.float64, torch.int64

    af = torch.rand(S, generator=gd, device=dev, dtype=f64) * 0.9 + 0.05
    G = ((torch.rand(S, V, generator=gd, device=dev, dtype=f64) < af[:, None]).to(torch.int8)
         + (torch.rand(S, V, generator=gd, device=dev, dtype=f64) < af[:, None]).to(torch.int8))  # Binomial(2, af)

    max_entries = min(max_entries, S)
    min_entries = min(min_entries, max_entries)
    L = torch.exp(torch.randn(C, generator=g, device=dev, dtype=f64) * sigma + float(np.log(mean_entries)))
    L = torch.clamp(torch.round(L), min_entries, max_entries).to(i64)
    cell_of = torch.repeat_interleave(torch.arange(C, device=dev, dtype=i64), L)
    key = cell_of * S + torch.randint(0, S, (cell_of.numel(),), generator=g, device=dev, dtype=i64)
    del cell_of
    key = torch.unique(key)  # sorted by (cell, snp); duplicates within a cell dropped
    cell_of = torch.div(key, S, rounding_mode="floor")
    entry_snp = (key - cell_of * S).to(torch.int32)
    del key
    cell_ptr = torch.zeros(C + 1, dtype=i64, device=dev)
    torch.cumsum(torch.bincount(cell_of, minlength=C), 0, out=cell_ptr[1:])
    nnz = entry_snp.numel()

    nreads = 1 + torch.poisson(torch.full((nnz,), reads_lambda, device=dev, dtype=torch.float32), generator=g).to(i64)
    entry_rptr = torch.zeros(nnz + 1, dtype=i64, device=dev)
    torch.cumsum(nreads, 0, out=entry_rptr[1:])
    R = int(entry_rptr[-1])
    read_entry = torch.repeat_interleave(torch.arange(nnz, device=dev, dtype=i64), nreads)
    del nreads
    read_cell = cell_of[read_entry]
    read_snp = entry_snp[read_entry].to(i64)
    del read_entry, cell_of

    is_dbl = torch.rand(C, generator=g, device=dev) < doublet_frac
    s1 = torch.randint(0, V, (C,), generator=g, device=dev, dtype=i64)
    s2 = (s1 + 1 + torch.randint(0, max(V - 1, 1), (C,), generator=g, device=dev, dtype=i64)) % V if V > 1 else s1.clone()
    pick2 = is_dbl[read_cell] & (torch.rand(R, generator=g, device=dev) < 0.5)
    src = torch.where(pick2, s2[read_cell], s1[read_cell])
    del pick2, read_cell
    gg = G.flatten()[read_snp * V + src].to(torch.float32)
    del read_snp, src
    allele = (torch.rand(R, generator=g, device=dev) < gg * 0.5).to(torch.uint8)
    del gg
    allele ^= (torch.rand(R, generator=g, device=dev) < flip).to(torch.uint8)
    bq = torch.clamp(torch.randint(min_bq, max_bq + 1, (R,), generator=g, device=dev, dtype=torch.int32), max=cap_bq).to(torch.uint8)
    reads = (allele << 7) | bq
    del allele, bq
    reads[torch.rand(R, generator=g, device=dev) < other] = READ_OTHER

    gp = has_gp = None
    if with_gp:  # gt_to_gp(): one-hot through float, avgGP accumulated over the samples in order, (1-err) gp + err avg
        gp = torch.zeros(S, V, 3, dtype=f64, device=dev)
        gp.scatter_(2, G.to(i64)[:, :, None], 1.0)
        avg = torch.full((S, 3), 1e-10, dtype=f64, device=dev)
        for v in range(V):
            avg += gp[:, v, :]
        avg /= (avg[:, 0] + avg[:, 1] + avg[:, 2])[:, None]
        if geno_err > 0:
            gp = (1 - geno_err) * gp + geno_err * avg[:, None, :]
        has_gp = torch.ones(S, dtype=torch.uint8, device=dev)
    truth = {"is_doublet": is_dbl, "s1": s1.to(torch.int32), "s2": s2.to(torch.int32)}
    return DevicePileup(C, S, cell_ptr, entry_snp, entry_rptr, reads, af, gp, has_gp, truth)
