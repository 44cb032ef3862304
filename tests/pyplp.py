"""Independent pure-Python restatement of sc_dropseq_lib_t::load_from_plp (sc_drop_seq.cpp:103-384) and of
BCFFilteredReader::parse_posteriors for GT / GP / PL (bcf_filtered_reader.cpp:250-327,367-461), used to check the C++
loader (popscle_amd/host/plp.hpp, vcf.hpp) on files of the real format.  Test infrastructure; small inputs only."""
from __future__ import annotations

import gzip
import math

import numpy as np


def _lines(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        for line in f:
            t = line.split()
            if not t:
                return  # a blank line ends the reference's read loops
            yield t


PL_MISSING = -2**31  # bcf_int32_missing: toProb() takes it as uint32 > 255 -> phred2Prob[255] (PhredHelper.h:40)


def _toprob(pl):
    """phredConv.toProb (PhredHelper.h:40, table PhredHelper.cpp:31)"""
    pl = pl & 0xFFFFFFFF
    return math.pow(0.1, (255 if pl > 255 else pl) * 0.1)


def pl_em(pls, nal, ploidies=None):
    """BCFFilteredReader::parse_likelihoods, bcf_filtered_reader.cpp:261-324: pls[n][ngenos] of the selected samples ->
    (float32 gps[n][ngenos], acs[nal] (allele counts), an)"""
    n = len(pls)
    ng = nal * (nal + 1) // 2
    ploidies = [2] * n if ploidies is None else list(ploidies)
    gps = np.zeros((n, ng), dtype=np.float32)
    acs = [1.0 / nal] * nal
    gp = [0.0] * ng
    an = 0
    for it in range(10):
        newacs = [0.0] * nal
        an = 0
        for v in range(n):
            if ploidies[v] == 2:
                sumgp = 0.0
                l = 0
                for j in range(nal):
                    for k in range(j + 1):
                        gp[l] = (1 if j == k else 2) * acs[j] * acs[k] * _toprob(int(pls[v][l]))
                        sumgp += gp[l]
                        l += 1
                l = 0
                for j in range(nal):
                    for k in range(j + 1):
                        gp[l] /= sumgp
                        newacs[j] += gp[l]
                        newacs[k] += gp[l]
                        l += 1
                an += 2
            elif ploidies[v] == 1:
                gp = [0.0] * ng
                sumgp = 0.0
                for j in range(nal):
                    l = (j + 1) * (j + 2) // 2 - 1
                    gp[l] = acs[j] * _toprob(int(pls[v][l]))
                    sumgp += gp[l]
                for j in range(nal):
                    l = (j + 1) * (j + 2) // 2 - 1
                    gp[l] /= sumgp
                    newacs[j] += gp[l]
                an += 1
            if it == 9:
                for l in range(ng):
                    gps[v, l] = np.float32(gp[l])
        acs = [x / an for x in newacs]
    return gps, [a * an for a in acs], an


def gp_normalise(vals, nal, gt_error=0.0):
    """parse_posteriors, GP branch, bcf_filtered_reader.cpp:422-457: float arithmetic; each sample divided by its own
    float sum; then (1-gt_error)*gp + gt_error*gpSums in double, stored as float.  With gt_error == 0 (the only value
    load_from_plp passes) the second step changes nothing -- unless gpSums is NaN (a sample with a missing value), which
    0*NaN spreads to every sample of the record."""
    f32 = np.float32
    n = len(vals)
    ng = nal * (nal + 1) // 2
    g = np.array(vals, dtype=np.float32).reshape(n, ng)
    sums = np.zeros(ng, dtype=np.float32)
    for i in range(nal):
        for j in range(i + 1):
            sums[(i + 1) * i // 2 + j] = f32((1.0 if i == j else 2.0) / float(f32(nal * nal)))
    with np.errstate(all="ignore"):
        for v in range(n):
            s = f32(0)
            for j in range(ng):
                s = f32(s + g[v, j])
            for j in range(ng):
                g[v, j] = f32(g[v, j] / s)
                sums[j] = f32(sums[j] + g[v, j])
        for j in range(ng):
            sums[j] = f32(sums[j] / f32(int(n + 1.0)))
        for v in range(n):
            for j in range(ng):
                g[v, j] = f32((1.0 - gt_error) * float(g[v, j]) + gt_error * float(sums[j]))
    return g


def gt_posteriors(gidx, acs, an, nal, gt_error=0.0, ploidies=None):
    """parse_posteriors, GT branch, bcf_filtered_reader.cpp:385-409 (samples in consecutive columns): one-hot of the
    genotype index, a missing genotype -> HWE from the allele counts with pseudo-counts, all through float"""
    f32 = np.float32
    n = len(gidx)
    ng = nal * (nal + 1) // 2
    ploidies = [2] * n if ploidies is None else list(ploidies)
    out = np.zeros((n + 1) * ng, dtype=np.float32)
    for v in range(n):
        g = gidx[v]
        o = v * ng
        if g < 0:
            if ploidies[v] == 2:
                l = 0
                for j in range(nal):
                    for k in range(j + 1):
                        out[o + l] = f32((1.0 if j == k else 2.0) * (acs[j] + 1.0 / nal) / (an + 1.0) * (acs[k] + 1.0 / nal) / (an + 1.0))
                        l += 1
            elif ploidies[v] == 1:
                out[o:o + 2 * ng] = 0  # :401 clears ngenos * sizeof(double) bytes of a float array: the next column too
                for j in range(nal):
                    out[o + (j + 1) * (j + 2) // 2 - 1] = f32((acs[j] + 1.0 / nal) / (an + 1.0))
        else:
            for j in range(ng):
                out[o + j] = f32(1.0 - gt_error) if g == j else f32(gt_error / (ng - 1.0))
    return out[:n * ng].reshape(n, ng)


def gp_row(float_gp, geno_error_offset=0.1, geno_error_coeff=0.0, r2=None):
    """load_from_plp, sc_drop_seq.cpp:287-315: the double row handed to add_snp from the reader's float posteriors"""
    g = np.asarray(float_gp, dtype=np.float32).astype(np.float64).reshape(-1)
    avg = [1e-10, 1e-10, 1e-10]
    for i in range(g.size):
        avg[i % 3] += g[i]
    s = avg[0] + avg[1] + avg[2]
    avg = [a / s for a in avg]
    err = geno_error_offset
    if geno_error_coeff > 0:
        err += (1 - geno_error_offset) * float(np.float32(1) - np.float32(r2)) * geno_error_coeff  # `1-r2flts[0]` is a float expression
    if err > 0.999:
        err = 0.999
    if err < 0:
        err = 0
    if err > 0:
        g = np.array([(1 - err) * g[i] + err * avg[i % 3] for i in range(g.size)])
    return g


def vcf_records(path, field, min_mac=1, min_callrate=0.5, max_alleles=2):
    """yield (rid, pos, ref0, alt0, float32 gps[nv*3], info dict) for records passing the variant filter"""
    contigs = {}
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        nv = 0
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("##contig=<"):
                name = line.split("ID=")[1].split(",")[0].split(">")[0]
                contigs.setdefault(name, len(contigs))
                continue
            if line.startswith("#CHROM"):
                nv = len(line.split("\t")) - 9
                continue
            if line.startswith("#") or not line:
                continue
            t = line.split("\t")
            alleles = [t[3]] + ([] if t[4] == "." else t[4].split(","))
            rid = contigs.setdefault(t[0], len(contigs))
            pos = int(t[1])
            if len(alleles) > max_alleles:
                continue
            keys = t[8].split(":")
            gi = keys.index("GT")
            gts = []
            acs = [0.0] * len(alleles)
            an = 0
            for v in range(nv):
                g = t[9 + v].split(":")[gi].replace("|", "/").split("/")
                a = [(-1 if x in (".", "") else int(x)) for x in g[:2]] + [-1] * (2 - len(g[:2]))
                gts.append(a)
                for x in a:
                    if x >= 0:
                        an += 1
                        acs[x] += 1
            if min_callrate > an / (2.0 * nv):
                continue
            ac = an - int(acs[0])
            if ac < min_mac or an - ac < min_mac:
                continue
            nal = len(alleles)
            if field == "GT":
                gidx = [(-1 if (a1 < 0 or a2 < 0) else max(a1, a2) * (max(a1, a2) + 1) // 2 + min(a1, a2)) for a1, a2 in gts]
                gps = gt_posteriors(gidx, acs, an, nal).reshape(-1)
            elif field == "PL":
                fi = keys.index("PL")
                pls = [[(PL_MISSING if x == "." else int(x)) for x in t[9 + v].split(":")[fi].split(",")] for v in range(nv)]
                gps = pl_em(pls, nal)[0].reshape(-1)
            else:
                fi = keys.index(field)
                vals = [[(np.float32(np.nan) if x == "." else np.float32(float(x))) for x in t[9 + v].split(":")[fi].split(",")]
                        for v in range(nv)]
                gps = gp_normalise(vals, nal).reshape(-1)
            info = dict(kv.split("=") for kv in t[7].split(";") if "=" in kv)
            yield rid, pos, alleles[0][0], (alleles[1][0] if len(alleles) > 1 else "."), gps, info


def load(prefix, vcf=None, field="GP", min_bq=13, cap_bq=20, geno_error_offset=0.1, geno_error_coeff=0.0, r2="R2",
         min_total=0, min_umi=0, min_snp=0, group_list=None, min_mac=1, min_callrate=0.5):
    valid = set(group_list) if group_list else set()
    index_bcs, bcs, tmp = [], [], []
    nskip = 0
    it = _lines(prefix + ".cel.gz")
    assert next(it) == ["#DROPLET_ID", "BARCODE", "NUM.READ", "NUM.UMI", "NUM.UMIwSNP", "NUM.SNP"]
    for t in it:
        if valid and t[1] not in valid:
            nskip += 1
            index_bcs.append(-1)
            continue
        if int(t[2]) < min_total or int(t[3]) < min_umi or int(t[5]) < min_snp:
            nskip += 1
            index_bcs.append(-1)
            continue
        index_bcs.append(len(bcs))
        assert len(bcs) + nskip == int(t[0])
        bcs.append(t[1])
        tmp.append((int(t[2]), int(t[4]), int(t[5])))
    C = len(bcs)

    recs = vcf_records(vcf, field, min_mac, min_callrate) if vcf else None
    cur = next(recs, None) if recs else None
    nv = (cur[4].size // 3) if cur else 0
    af, gp, has_gp, chr2rid = [], [], [], {}
    it = _lines(prefix + ".var.gz")
    assert next(it) == ["#SNP_ID", "CHROM", "POS", "REF", "ALT", "AF"]
    for t in it:
        rid = chr2rid.setdefault(t[1], len(chr2rid))
        pos, ref, alt = int(t[2]), t[3][0], t[4][0]
        af.append(float(t[5]))
        if recs is None:
            continue
        row = None
        while True:
            if cur is None or cur[0] > rid:
                break
            if cur[0] == rid:
                if cur[1] > pos:
                    break
                if cur[1] == pos:
                    if cur[2] != ref or cur[3] != alt:
                        break
                    g = gp_row(cur[4], geno_error_offset, geno_error_coeff,
                               float(cur[5][r2]) if geno_error_coeff > 0 else None)
                    row = g
                    break
            cur = next(recs, None)
        has_gp.append(0 if row is None else 1)
        gp.append(np.zeros(nv * 3) if row is None else row)
    S = len(af)

    per_cell = [dict() for _ in range(C)]
    uniq = [0] * C
    numi = 0
    it = _lines(prefix + ".plp.gz")
    assert next(it) == ["#DROPLET_ID", "SNP_ID", "ALLELES", "BASEQS"]
    for t in it:
        ibc = index_bcs[int(t[0])]
        if ibc < 0:
            continue
        snp = int(t[1])
        for a, q in zip(t[2], t[3]):
            bq = ord(q) - 33
            if bq >= min_bq:
                bq = min(bq, cap_bq)
                umi = "%x" % numi
                numi += 1
                al = ord(a) - ord("0")
                byte = bq if al == 0 else (0x80 | bq) if al == 1 else 0xFF
                per_cell[ibc].setdefault(snp, {})[umi] = byte
                uniq[ibc] += 1
    cell_ptr, entry_snp, entry_rptr, reads = [0], [], [0], []
    for c in range(C):
        for snp in sorted(per_cell[c]):
            entry_snp.append(snp)
            for umi in sorted(per_cell[c][snp]):  # std::map<std::string> order
                reads.append(per_cell[c][snp][umi])
            entry_rptr.append(len(reads))
        cell_ptr.append(len(entry_snp))
    totl = []
    for c in range(C):
        nent = cell_ptr[c + 1] - cell_ptr[c]
        totl.append(tmp[c][0] if (uniq[c] == tmp[c][1] and tmp[c][2] == nent) else uniq[c])
    return dict(C=C, S=S, nv=nv, bcs=bcs, cell_ptr=np.array(cell_ptr, dtype=np.int64),
                entry_snp=np.array(entry_snp, dtype=np.int32), entry_rptr=np.array(entry_rptr, dtype=np.int64),
                reads=np.array(reads, dtype=np.uint8), af=np.array(af),
                has_gp=(np.array(has_gp, dtype=np.uint8) if recs is not None else np.zeros(S, dtype=np.uint8)),
                gp=(np.array(gp).reshape(S, nv, 3) if nv else np.zeros((S, 0, 3))),
                cell_totl_reads=np.array(totl, dtype=np.int32), cell_uniq_reads=np.array(uniq, dtype=np.int32))
