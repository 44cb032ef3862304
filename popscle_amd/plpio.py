"""Writers of popscle's digital-pileup text formats (CEL/VAR/PLP, written by `dsc-pileup`,
cmd_cram_dsc_pileup.cpp:438-523) and of a matching genotype VCF, from a synthetic `synth.Pileup`.

Used to exercise the C++ loader and the popscle-amd CLI on files of the real format; the benchmark hands the packed
arrays to the library directly.
"""
from __future__ import annotations

import gzip

import numpy as np

READ_OTHER = 0xFF


def barcodes(C: int, seed: int = 0):
    """C distinct 16-mers + '-1' in a shuffled (non-sorted) order, like a 10x barcode list"""
    rng = np.random.default_rng([seed, 77])
    out = set()
    alphabet = np.array(list("ACGT"))
    while len(out) < C:
        out.add("".join(alphabet[rng.integers(0, 4, 16)]) + "-1")
    out = sorted(out)
    rng.shuffle(out)
    return out


def write_plp(prefix: str, p, bcs=None, raw_bq=None, chrom="1", seed: int = 0, extra_cells=0):
    """Write prefix.cel.gz / .var.gz / .plp.gz for pileup p.

    raw_bq: optional uint8[R] of raw (uncapped) base qualities to write instead of the capped ones in p.reads (lets a
    test exercise --min-BQ / --cap-BQ).  extra_cells: number of additional droplets with NUM.SNP = 0 appended to the CEL
    file (they produce no PLP rows), to exercise --min-snp filtering and DROPLET_ID bookkeeping.
    Returns the barcode list (droplet id order).
    """
    C, S = p.C, p.S
    bcs = list(bcs) if bcs is not None else barcodes(C + extra_cells, seed)
    al = np.where(p.reads == READ_OTHER, 2, p.reads >> 7).astype(np.uint8)
    if raw_bq is None:
        bq = np.where(p.reads == READ_OTHER, 20, p.reads & 0x7F).astype(np.uint8)  # 'other' bases carry no quality
    else:
        bq = np.asarray(raw_bq, dtype=np.uint8)
    nreads_e = np.diff(p.entry_rptr)
    with gzip.open(prefix + ".cel.gz", "wt") as f:
        f.write("#DROPLET_ID\tBARCODE\tNUM.READ\tNUM.UMI\tNUM.UMIwSNP\tNUM.SNP\n")
        for c in range(C):
            e0, e1 = p.cell_ptr[c], p.cell_ptr[c + 1]
            nr = int(nreads_e[e0:e1].sum())
            f.write(f"{c}\t{bcs[c]}\t{nr + 100}\t{nr + 10}\t{nr}\t{e1 - e0}\n")
        for c in range(C, C + extra_cells):
            f.write(f"{c}\t{bcs[c]}\t5\t5\t0\t0\n")
    with gzip.open(prefix + ".var.gz", "wt") as f:
        f.write("#SNP_ID\tCHROM\tPOS\tREF\tALT\tAF\n")
        for s in range(S):
            f.write(f"{s}\t{chrom}\t{1000 + 10 * s}\tA\tG\t{p.af[s]:.5f}\n")
    # rows sorted by SNP then droplet (cmd_cram_dsc_pileup.cpp:497-518)
    cell_of = np.repeat(np.arange(C), np.diff(p.cell_ptr))
    order = np.lexsort((cell_of, p.entry_snp))
    with gzip.open(prefix + ".plp.gz", "wt") as f:
        f.write("#DROPLET_ID\tSNP_ID\tALLELES\tBASEQS\n")
        for e in order:
            r0, r1 = p.entry_rptr[e], p.entry_rptr[e + 1]
            if r1 == r0:
                continue
            a = "".join(str(int(x)) for x in al[r0:r1])
            q = "".join(chr(int(x) + 33) for x in bq[r0:r1])
            f.write(f"{cell_of[e]}\t{p.entry_snp[e]}\t{a}\t{q}\n")
    return bcs


def write_vcf(path: str, p, G, field="GT", chrom="1", sample_prefix="S", missing_frac=0.0, drop_snps=(), seed=0,
              gp=None, pl=None):
    """Write a VCF (plain or .gz) with the donors' genotypes G[S][V] at the VAR positions of write_plp.

    field: "GT" only, or "GP"/"PL" to add that FORMAT key (values from gp[S][V][3] / pl[S][V][3]).
    drop_snps: SNP ids left out of the VCF (the loader must then mark them has_gp = 0).
    """
    S, V = G.shape
    rng = np.random.default_rng([seed, 5])
    opener = gzip.open if path.endswith(".gz") else open
    drop = set(int(x) for x in drop_snps)
    with opener(path, "wt") as f:
        f.write("##fileformat=VCFv4.2\n")
        f.write(f"##contig=<ID={chrom}>\n")
        f.write('##INFO=<ID=R2,Number=1,Type=Float,Description="imputation r2">\n')
        f.write('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n')
        f.write('##FORMAT=<ID=GP,Number=G,Type=Float,Description="Genotype posterior">\n')
        f.write('##FORMAT=<ID=PL,Number=G,Type=Integer,Description="Phred likelihood">\n')
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" +
                "\t".join(f"{sample_prefix}{v}" for v in range(V)) + "\n")
        gts = ["0/0", "0/1", "1/1"]
        for s in range(S):
            if s in drop:
                continue
            cols = []
            for v in range(V):
                g = gts[int(G[s, v])] if rng.random() >= missing_frac else "./."
                if field == "GP":
                    g += ":" + ",".join(f"{x:.3f}" for x in gp[s, v])
                elif field == "PL":
                    g += ":" + ",".join(str(int(x)) for x in pl[s, v])
                cols.append(g)
            fmt = "GT" if field == "GT" else "GT:" + field
            f.write(f"{chrom}\t{1000 + 10 * s}\t.\tA\tG\t.\tPASS\tR2={0.5 + 0.5 * p.af[s]:.3f}\t{fmt}\t" + "\t".join(cols) + "\n")


def read_dump(path: str):
    """Parse the binary file `popscle-amd dump-plp` writes (popscle_amd/host/main.cpp)."""
    buf = open(path, "rb").read()
    assert buf[:8] == b"MUXGLPLP"
    off = 8
    C, S, nnz, R, nv = np.frombuffer(buf, dtype=np.int64, count=5, offset=off)
    off += 40

    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(buf, dtype=dtype, count=int(n), offset=off).copy()
        off += a.nbytes
        return a

    out = dict(C=int(C), S=int(S), nnz=int(nnz), R=int(R), nv=int(nv))
    out["cell_ptr"] = take(np.int64, C + 1)
    out["entry_snp"] = take(np.int32, nnz)
    out["entry_rptr"] = take(np.int64, nnz + 1)
    out["reads"] = take(np.uint8, R)
    out["af"] = take(np.float64, S)
    out["has_gp"] = take(np.uint8, S)
    out["gp"] = take(np.float64, S * nv * 3).reshape(int(S), int(nv), 3) if nv else np.zeros((int(S), 0, 3))
    out["cell_totl_reads"] = take(np.int32, C)
    out["cell_uniq_reads"] = take(np.int32, C)
    strs = buf[off:].split(b"\0")
    out["bcs"] = [s.decode() for s in strs[:int(C)]]
    out["sample_ids"] = [s.decode() for s in strs[int(C):int(C) + int(nv)]]
    return out


def read_slab_dump(path: str):
    """Parse the file `popscle-amd dump-plp --rank r --world N` writes: one rank's row slab (its cells, every marker,
    cells renumbered from 0) and column slab (every cell, its markers) of a sharded run, cut by the loader itself."""
    buf = open(path, "rb").read()
    assert buf[:8] == b"MUXGLSLB"
    off = 8
    C, S, c0, c1, s0, s1, nnz_r, R_r, nnz_c, R_c = (int(x) for x in np.frombuffer(buf, dtype=np.int64, count=10, offset=off))
    off += 80

    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(buf, dtype=dtype, count=int(n), offset=off).copy()
        off += a.nbytes
        return a

    out = dict(C=C, S=S, c0=c0, c1=c1, s0=s0, s1=s1)
    out["rows"] = (take(np.int64, c1 - c0 + 1), take(np.int32, nnz_r), take(np.int64, nnz_r + 1), take(np.uint8, R_r))
    out["cols"] = (take(np.int64, C + 1), take(np.int32, nnz_c), take(np.int64, nnz_c + 1), take(np.uint8, R_c))
    out["af"] = take(np.float64, S)
    out["cell_totl_reads"] = take(np.int32, C)
    out["cell_uniq_reads"] = take(np.int32, C)
    assert off == len(buf)
    return out
