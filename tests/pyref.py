"""Second, independent restatement of the reference arithmetic in plain Python floats (IEEE doubles, math.log/exp =
glibc), written from the reference text separately from oracle/muxgl_oracle.c.  Pure-Python loops: tiny cases only.
It exists to catch transcription slips in the C oracle (the two must agree bit-for-bit); like the oracle it is test
infrastructure and "parity unpinned" with respect to the reference binary.
"""
from __future__ import annotations

import math

READ_OTHER = 0xFF


def phred():
    # PhredHelper.cpp:31-33
    err = [(math.pow(0.1, i * 0.1) if i > 1 else 0.75) for i in range(256)]
    mat = [1.0 - e for e in err]
    return err, mat


ERR, MAT = phred()


def logadd(la, lb):
    # sc_drop_seq.cpp:5-8
    if la > lb:
        return la + math.log(1.0 + math.exp(lb - la))
    return lb + math.log(1.0 + math.exp(la - lb))


def demux_entry_pg(reads, alphas):
    # cmd_cram_demuxlet.cpp:655-725
    nA = len(alphas)
    pGs = [1.0] * (nA * 9)
    for b in reads:
        if b == READ_OTHER:
            continue
        al, bq = b >> 7, b & 0x7F
        pR = MAT[bq] if al == 0 else ERR[bq] / 3.0
        pA = MAT[bq] if al == 1 else ERR[bq] / 3.0
        maxpG = 0.0
        for k in range(nA):
            for l in range(3):
                for m in range(3):
                    p = 0.5 * l + (m - l) * 0.5 * alphas[k]
                    pGs[k * 9 + l * 3 + m] *= (pR * (1.0 - p) + pA * p)
                    if maxpG < pGs[k * 9 + l * 3 + m]:
                        maxpG = pGs[k * 9 + l * 3 + m]
        for i in range(nA * 9):
            pGs[i] /= maxpG
    maxpG = 0.0
    for i in range(nA * 9):
        pGs[i] += 1e-10
        if maxpG < pGs[i]:
            maxpG = pGs[i]
    for i in range(nA * 9):
        pGs[i] /= maxpG
    return pGs


def demux_cell_ll(entries, gp, has_gp, nv, alphas):
    """entries: list of (snp, [read bytes]); gp[snp] = flat list nv*3.  Returns llksAB flat [nv][nv][nA]."""
    nA = len(alphas)
    ll = [0.0] * (nv * nv * nA)
    for snp, reads in entries:
        pGs = demux_entry_pg(reads, alphas)
        if not has_gp[snp]:
            continue
        g = gp[snp]
        for j in range(nv):
            for k in range(nv):
                sumPs = [0.0] * nA
                for l in range(3):
                    for m in range(3):
                        p = g[j * 3 + l] * g[k * 3 + m]
                        for n in range(nA):
                            sumPs[n] += (p * pGs[n * 9 + l * 3 + m])
                for n in range(nA):
                    ll[j * nv * nA + k * nA + n] += math.log(sumPs[n])
    return ll


def demux_call(ll, nv, alphas, doublet_prior):
    """cmd_cram_demuxlet.cpp:788-991 -> dict of the record fields"""
    nA = len(alphas)
    sBest = sNext = dBest1 = dBest2 = dNext1 = dNext2 = dBA = dNA = -1
    sngBest = sngNext = dblBest = dblNext = -1e300
    sumLLK = sngLLK = -1e-300
    lsp = math.log((1.0 - doublet_prior) / nv)

    def safe_log(x):
        return math.log(x) if x > 0 and x != math.inf else (math.inf if x > 0 else -math.inf)

    try:
        d1 = doublet_prior / nv / (nv - 1.0) / (nA - 1.0)
    except ZeroDivisionError:
        d1 = math.inf
    ldp1 = safe_log(d1)
    ldp2 = safe_log(d1 * 2)
    for j in range(nv):
        sumLLK = logadd(sumLLK, ll[j * nv * nA] + lsp)
        sngLLK = logadd(sngLLK, ll[j * nv * nA] + lsp)
        for k in range(nv):
            if j == k:
                continue
            for n in range(1, nA):
                if alphas[n] == 0.5:
                    if k > j:
                        continue
                    sumLLK = logadd(sumLLK, ll[j * nv * nA + k * nA + n] + ldp2)
                else:
                    sumLLK = logadd(sumLLK, ll[j * nv * nA + k * nA + n] + ldp1)
    for j in range(nv):
        v = ll[j * nv * nA]
        if sngBest < v:
            sngNext, sNext, sBest, sngBest = sngBest, sBest, j, v
        elif sngNext < v:
            sNext, sngNext = j, v
    for j in range(nv):
        for k in range(nv):
            if j == k:
                continue
            for n in range(1, nA):
                v = ll[j * nv * nA + k * nA + n]
                if dblBest < v:
                    dNext1, dNext2, dNA, dblNext = dBest1, dBest2, dBA, dblBest
                    dBest1, dBest2, dBA, dblBest = j, k, n, v
                elif dblNext < v:
                    dNext1, dNext2, dNA, dblNext = j, k, n, v
    if dblBest > sngBest + 2:
        typ = 1
        bestPP = math.exp(dblBest + (ldp2 if alphas[dBA] == 0.5 else ldp1) - sumLLK)
        jB, kB, bestLLK, aB = dBest1, dBest2, dblBest, dBA
        if dblNext > sngBest + 2:
            ntyp, jN, kN, nextLLK, aN = 1, dNext1, dNext2, dblNext, dNA
        else:
            ntyp, jN, kN, nextLLK, aN = 0, sBest, sBest, sngBest, 0
    else:
        typ = 0 if sngBest > sngNext + 2 else 2
        bestPP = sngBest + lsp - sumLLK
        jB, kB, bestLLK, aB = sBest, sBest, sngBest, 0
        if dblBest > sngNext + 2:
            ntyp, jN, kN, nextLLK, aN = 1, dBest1, dBest2, dblBest, dBA
        else:
            ntyp, jN, kN, nextLLK, aN = 0, sNext, sNext, sngNext, 0
    return dict(type=typ, next_type=ntyp, sBest=sBest, sNext=sNext, dBest1=dBest1, dBest2=dBest2, dBestA=dBA,
                dNext1=dNext1, dNext2=dNext2, dNextA=dNA, jBest=jB, kBest=kB, aBest=aB, jNext=jN, kNext=kN, aNext=aN,
                sngBestLLK=sngBest, sngNextLLK=sngNext, dblBestLLK=dblBest, dblNextLLK=dblNext, sumLLK=sumLLK,
                sngLLK=sngLLK, bestLLK=bestLLK, nextLLK=nextLLK, bestPP=bestPP, sngPP=math.exp(sngLLK - sumLLK),
                sngOnlyPP=math.exp(sngBest + lsp - sngLLK))


# ---------------------------------------------------------------------------------------------- freemuxlet

MIN_NORM_GL = 1e-6


def fmx_entry_pileup(reads):
    # sc_drop_seq.cpp:452-509, alpha = 0.5
    a = 0.5
    gls = [1.0] * 9
    nreads = nref = nalt = 0
    for b in reads:
        nreads += 1
        if b == READ_OTHER:
            continue
        al, bq = b >> 7, b & 0x7F
        if al == 0:
            nref += 1
        else:
            nalt += 1
        ref = al == 0
        fr = [1.0 if ref else 0.0, (1. - a / 2.) if ref else a / 2., (1.0 - a) if ref else a,
              ((1. + a) / 2.) if ref else (1. - a) / 2., .5, ((1. - a) / 2.) if ref else (1. + a) / 2.,
              a if ref else 1. - a, (a / 2.) if ref else 1. - a / 2., 0.0 if ref else 1.0]
        for i in range(9):
            gls[i] *= (MAT[bq] * fr[i] + ERR[bq] / 4.)
        tmp = 0.0
        for i in range(9):
            tmp += gls[i]
        for i in range(9):
            gls[i] /= tmp
    for i in range(9):
        if gls[i] < MIN_NORM_GL:
            gls[i] = MIN_NORM_GL
    tmp = 0.0
    for i in range(9):
        tmp += gls[i]
    for i in range(9):
        gls[i] /= tmp
    return nreads, nref, nalt, gls


def plp_merge(dst, src):
    """dst/src = [nreads, nref, nalt, gls(list of 9)]; sc_drop_seq.h:77-101"""
    dst[0] += src[0]
    dst[1] += src[1]
    dst[2] += src[2]
    g = dst[3]
    for i in range(9):
        g[i] *= src[3][i]
    tmp = 0.0
    for i in range(9):
        tmp += g[i]
    for i in range(9):
        g[i] /= tmp
    for i in range(9):
        if g[i] < MIN_NORM_GL:
            g[i] = MIN_NORM_GL
    tmp = 0.0
    for i in range(9):
        tmp += g[i]
    for i in range(9):
        g[i] /= tmp


def fmx_estep_cell(entries, af, cplp, K, geno_error):
    """entries: list of (snp, gls9); cplp[k][snp] -> gls list (default all ones).  cmd_cram_freemux2.cpp:386-456.
    Written pair-by-pair exactly as the reference (gp2s re-evaluated inside the k loop)."""
    npairs = K * (K + 1) // 2
    llks = [0.0] * npairs
    for snp, glis in entries:
        a = af[snp]
        gp0s = [(1.0 - a) * (1.0 - a), 2 * a * (1.0 - a), a * a]
        lks = [0.0] * npairs
        for j in range(K):
            s1 = cplp[j][snp]
            gp1s = [(1.0 - a) * (1.0 - a) * s1[0], 2 * a * (1.0 - a) * s1[4], a * a * s1[8]]
            sum1 = gp1s[0] + gp1s[1] + gp1s[2]
            gp1s = [x / sum1 for x in gp1s]
            if geno_error > 0:
                gp1s = [(1 - geno_error) * gp1s[i] + geno_error * gp0s[i] for i in range(3)]
            for k in range(j):
                s2 = cplp[k][snp]
                gp2s = [(1.0 - a) * (1.0 - a) * s2[0], 2 * a * (1.0 - a) * s2[4], a * a * s2[8]]
                sum2 = gp2s[0] + gp2s[1] + gp2s[2]
                gp2s = [x / sum2 for x in gp2s]
                if geno_error > 0:
                    gp2s = [(1 - geno_error) * gp2s[i] + geno_error * gp0s[i] for i in range(3)]
                lk = 0.0
                for g1 in range(3):
                    for g2 in range(3):
                        lk += (glis[g1 * 3 + g2] * gp1s[g1] * gp2s[g2])
                lks[j * (j + 1) // 2 + k] = lk
            lk = 0.0
            for g1 in range(3):
                lk += (glis[g1 * 3 + g1] * gp1s[g1])
            lks[j * (j + 1) // 2 + j] = lk
        for i in range(npairs):
            llks[i] += math.log(lks[i])
    return llks
