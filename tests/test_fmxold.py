"""CPU tests of the freemuxlet-old restatement (oracle/muxgl_oracle.c, cmd_cram_freemuxlet.cpp:176-343) and of the
arithmetic fact the device vote kernel relies on."""
import math

import numpy as np

import oracle_binding as ob
from popscle_amd import synth


def small(seed=3, C=60, S=300, K=3, mean_entries=80):
    return synth.make_pileup(C, S, K, seed=seed, mean_entries=mean_entries, with_gp=False)


def pair_dist_py(p, e):
    """independent restatement: SNP-major dict-of-dicts walk like the reference's snp_cell_plps (:113,187-221)"""
    snp_cells = {}
    for i in range(p.C):
        for x in range(p.cell_ptr[i], p.cell_ptr[i + 1]):
            snp_cells.setdefault(int(p.entry_snp[x]), []).append((i, x))
    out = np.zeros(p.C * (p.C - 1) // 2, dtype=ob.DROPD)
    for v in sorted(snp_cells):
        af = float(p.af[v])
        gps = [(1.0 - af) * (1.0 - af), 2.0 * af * (1.0 - af), af * af]
        lst = snp_cells[v]
        for ai in range(len(lst)):
            a, xa = lst[ai]
            ga = e["gls"][xa]
            for bi in range(ai):
                b, xb = lst[bi]
                gb = e["gls"][xb]
                lk0 = 0.0
                lk2 = 0.0
                for gi in range(3):
                    lk2 += ga[gi * 3 + gi] * gb[gi * 3 + gi] * gps[gi]
                    for gj in range(3):
                        lk0 += ga[gi * 3 + gi] * gb[gj * 3 + gj] * gps[gi] * gps[gj]
                d = out[a * (a - 1) // 2 + b]
                d["nsnps"] += 1
                d["nread1"] += e["nreads"][xa]
                d["nread2"] += e["nreads"][xb]
                d["llk2"] += math.log(lk2)
                d["llk0"] += math.log(lk0)
    return out


def test_pair_dist_vs_python():
    p = small()
    e = ob.fmx_entry_pileup(p)
    got = ob.fmxold_pair_dist(p, e)
    want = pair_dist_py(p, e)
    for f in ("nsnps", "nread1", "nread2", "llk0", "llk2"):
        assert np.array_equal(got[f], want[f]), f
    assert got["nsnps"].max() > 3


def dd_from_bf(C, bf):
    """dropD table whose llk2 - llk0 equals the symmetric matrix bf"""
    dd = np.zeros(C * (C - 1) // 2, dtype=ob.DROPD)
    for a in range(C):
        for b in range(a):
            dd[a * (a - 1) // 2 + b]["llk2"] = bf[a, b]
    return dd


def vote_init_py(C, K, bf, order, jitter, thres, frac):
    clust = np.full(C, -1, dtype=np.int32)
    t = 0
    for i in range(C):
        si = order[i]
        if i > C * frac:
            continue
        votes = [np.float64(x) for x in jitter[t]]
        t += 1
        for j in range(i):
            sj = order[j]
            d = bf[max(si, sj), min(si, sj)]
            if -d > thres:
                votes[clust[sj]] -= 1.0
            elif d > thres:
                votes[clust[sj]] += 1.0
        clust[si] = int(np.argmax(votes))  # first maximum == strict '<' scan
    return clust


def test_vote_init_and_refine_vs_python():
    rng = np.random.default_rng(5)
    C, K = 70, 4
    truth = rng.integers(0, K, C)
    bf = np.where(truth[:, None] == truth[None, :], 9.0, -9.0) * (rng.random((C, C)) < 0.6) + rng.normal(0, 2, (C, C))
    bf = np.tril(bf, -1)
    bf = bf + bf.T
    dd = dd_from_bf(C, bf)
    order = rng.permutation(C).astype(np.int32)
    jit = rng.integers(0, 2**31, (C, K)) / (2.0**31) / 1000.0
    got, cc = ob.fmxold_vote_init(C, K, dd, order, jit, 5.41, 0.8)
    want = vote_init_py(C, K, bf, order, jit, 5.41, 0.8)
    assert np.array_equal(got, want)
    assert (got == -1).sum() == C - sum(1 for i in range(C) if not i > C * 0.8)
    assert np.array_equal(cc, np.bincount(got[got >= 0], minlength=K))
    # refinement: python restatement of :297-343
    clust = got.copy()
    order2 = rng.permutation(C).astype(np.int32)
    jit2 = rng.integers(0, 2**31, (C, K)) / (2.0**31) / 1000.0
    for keep in (False, True):
        g, ch, cc = ob.fmxold_vote_refine(C, K, dd, order2, jit2, clust, 5.41, keep)
        w = clust.copy()
        changed = 0
        for i in range(C):
            si = order2[i]
            votes = [np.float64(x) for x in jit2[i]]
            for j in range(C):
                if si != j and w[j] >= 0:
                    d = bf[max(si, j), min(si, j)]
                    if d > 5.41:
                        votes[w[j]] += 1.0
                    elif d < -5.41:
                        votes[w[j]] -= 1.0
            el = int(np.argmax(votes))
            if w[si] >= 0 or not keep:
                changed += int(w[si] != el)
                w[si] = el
        assert np.array_equal(g, w) and ch == changed
        assert (g == -1).any() == keep


def vote_exact_py(f0, steps):
    """the closed form the device vote kernel uses (popscle_amd/csrc/fmx_old.hip: vote_exact)"""
    n = M = m = first = 0
    for s in steps:
        n += s
        if first == 0:
            first = s
        M = max(M, n)
        m = min(m, n)
    f = np.float64(f0)
    if first != 0:
        emax = -1 if first < 0 else 0
        if M >= 1:
            emax = max(emax, M.bit_length() - 1)
        if m <= -2:
            emax = max(emax, (-m - 1).bit_length() - 1)
        for e in range(-1 if first < 0 else 0, emax + 1):
            c = np.float64(2.0**e)
            f = (f + c) - c
    return np.float64(n) + f


def test_vote_rounding_path_is_a_function_of_four_integers():
    """sequential `votes[k] += 1.0 / -= 1.0` on a double that starts at a jitter < 0.001
    (cmd_cram_freemuxlet.cpp:257-278) == integer sum + jitter rounded through the binades the walk visited"""
    rng = np.random.default_rng(11)
    for trial in range(20000):
        L = int(rng.integers(0, 80)) if trial % 50 else int(rng.integers(500, 5000))
        pr = rng.random()
        steps = rng.choice([-1, 1], size=L, p=[pr, 1 - pr]).tolist()
        mode = trial % 4
        if mode == 0:
            f0 = float(rng.integers(0, 2**31)) / (2.0**31) / 1000.0
        elif mode == 1:
            f0 = 0.0
        elif mode == 2:
            f0 = float(np.ldexp(float(rng.integers(1, 2**20)), -int(rng.integers(40, 75))))
        else:
            f0 = float(np.ldexp(float(2 * rng.integers(1, 2**10) + 1), -int(rng.integers(50, 60))))  # ties
        x = np.float64(f0)
        for s in steps:
            x = x + np.float64(s)
        assert x == vote_exact_py(f0, steps), (f0, steps[:20])


def test_private_random_r_state_equals_rand():
    """the front end draws from a private glibc state (popscle_amd/host/main.cpp: RefRand) because the HIP runtime
    disturbs the process-wide rand(); both must yield the same values for the same seed"""
    import ctypes

    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 77):
        rd = ctypes.create_string_buffer(64)  # struct random_data (48 bytes on x86-64), zeroed
        st = ctypes.create_string_buffer(128)
        assert libc.initstate_r(ctypes.c_uint(seed), st, ctypes.c_size_t(128), rd) == 0
        libc.srand(seed)
        r = ctypes.c_int32()
        for _ in range(2000):
            libc.random_r(rd, ctypes.byref(r))
            assert r.value == libc.rand()


def test_oracle_reproduces_golden_fmxold():
    """tests/golden/fmxold_k4.npz: initial clustering = regression vectors of the restatement; EM = the reference's own run
    (tests/golden/make_golden.py)"""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fmxold_k4.npz"))
    p = synth.Pileup(int(g["C"]), int(g["S"]), g["cell_ptr"], g["entry_snp"], g["entry_rptr"], g["reads"], g["af"])
    e = ob.fmx_entry_pileup(p)
    dd = ob.fmxold_pair_dist(p, e)
    assert dd.tobytes() == g["dropd"].tobytes()
    K, thres, frac = int(g["K"]), float(g["bf_thres"]), float(g["frac_init_clust"])
    cl, cc = ob.fmxold_vote_init(p.C, K, dd, g["order"], g["jitter0"], thres, frac)
    assert np.array_equal(cl, g["clust0"]) and np.array_equal(cc, g["ccounts0"])
    for it in range(3):
        cl, ch, _ = ob.fmxold_vote_refine(p.C, K, dd, g["orands"][it], g["jitters"][it], cl, thres, it == 0)
        assert np.array_equal(cl, g["clusts"][it]) and ch == g["changed"][it]
    # the EM part of the fixture is the REFERENCE's run (cmd_cram_freemuxlet.cpp:456-653, see make_golden.py)
    cplp = ob.fmx_build_cluster_pileup(p, e, K, cl)
    cells = ob.fmx_init_cells(cl)
    ge, dp = float(g["em_geno_error"]), float(g["em_doublet_prior"])
    for it in range(10):
        nsng, namb, _, full = ob.fmx_iterate(p, e, K, cplp, cells, dp, ge if it == 9 else 0.0, full_ll=True)
        assert (nsng, namb) == tuple(g["em_counters"][it])
        for f in cells.dtype.names:
            if f not in ("clust", "_pad"):
                assert cells[f].tobytes() == g["em_cells"][it][f].tobytes(), (it, f)
        if it == 0:
            assert np.array_equal(full, g["em_full_ll_first"])
    assert np.array_equal(full, g["em_full_ll_last"])
    assert np.array_equal(cplp["gls"], g["em_cluster_gls"])
    assert np.array_equal(np.stack([cplp["nreads"], cplp["nref"], cplp["nalt"]], axis=-1), g["em_cluster_cnt"])
