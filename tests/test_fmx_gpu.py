"""GPU parity tests of the freemuxlet path: libmuxgl (HIP, through the C-ABI) vs the CPU oracle and the golden vectors.

Bar: cluster / type calls exact, log-likelihoods within 1e-5 absolute, over whole EM trajectories (the M-step feeds the
next E-step, so a deviation in the ordered clamped merge would show up one iteration later).
"""
import os

import numpy as np
import pytest

import oracle_binding as ob
import parity
from popscle_amd import muxgl, synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module", params=["default", "row", "wave", "pair"])
def eng(request):
    """'default' = normal dispatch (oct E-step for K <= 16, two clusters per lane up to 32, wave kernels above), 'row' = the
    oct kernel disabled (row E-step, two clusters per lane),
    'wave' = no two-per-lane kernel for 16 < K <= 32 (those shapes then take the pair kernel), 'pair' = the general pair kernel and the (SNP, cluster)-
    parallel M-step forced for every K"""
    flags = {"default": 0, "row": muxgl.FLAG_FORCE_ROW_KERNEL, "wave": muxgl.FLAG_FORCE_WAVE_KERNEL,
             "pair": muxgl.FLAG_FORCE_TILE_SWEEP}[request.param]
    e = muxgl.Engine(0, flags)
    yield e
    e.close()


@pytest.fixture(scope="module", params=["auto", "batched", "serial"])
def geng(request):
    """greedy initial clustering: 'auto' = the batched fixpoint kernels up to K = 64 and the serial persistent-workgroup
    kernel above, 'serial' = the latter for every K ('batched' is what 'auto' does since round 2; the flag is kept)"""
    flags = {"auto": 0, "batched": muxgl.FLAG_FORCE_BATCHED_GREEDY, "serial": muxgl.FLAG_FORCE_TILE_SWEEP}[request.param]
    e = muxgl.Engine(0, flags)
    yield e
    e.close()


def oracle_init(p, K, clust=None):
    e = ob.fmx_entry_pileup(p)
    llk0, llk2, ns, nr = ob.fmx_cell_scores(p, e)
    if clust is None:
        clust = ob.fmx_greedy_init(p, e, K, llk2 - llk0, ob.fmx_sort(llk2 - llk0))
    cplp = ob.fmx_build_cluster_pileup(p, e, K, clust)
    cells = ob.fmx_init_cells(clust)
    return e, (llk0, llk2, ns, nr), clust, cplp, cells


def check_prepare(eng, p, e, scores):
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    llk0, llk2, ns, nr = eng.fmx_prepare(p.af)
    assert np.array_equal(ns, scores[2]) and np.array_equal(nr, scores[3])
    assert np.max(np.abs(llk0 - scores[0]), initial=0.0) < 1e-8 and np.max(np.abs(llk2 - scores[1]), initial=0.0) < 1e-8
    gls, cnt = eng.fmx_entry_gls()
    assert np.array_equal(cnt, np.stack([e["nreads"], e["nref"], e["nalt"]], axis=-1))
    assert np.allclose(gls, e["gls"], rtol=1e-13, atol=1e-300)


def run_em(eng, p, K, n_iter, clust=None, doublet_prior=0.5, geno_error=0.1):
    e, scores, clust, cplp, cells = oracle_init(p, K, clust)
    check_prepare(eng, p, e, scores)
    eng.fmx_set_clusters(K, clust)
    g0, c0 = eng.fmx_cluster_pileup()
    assert np.array_equal(c0, np.stack([cplp["nreads"], cplp["nref"], cplp["nalt"]], axis=-1))
    assert np.allclose(g0, cplp["gls"], rtol=1e-12, atol=1e-300)
    worst = 0.0
    for it in range(n_iter):
        ostats = ob.fmx_iterate(p, e, K, cplp, cells, doublet_prior, geno_error, full_ll=True)
        gcells, gstats, gfull = eng.fmx_iterate(doublet_prior, geno_error, want_full_ll=True)
        d = np.abs(gfull - ostats[3])
        assert np.max(d[np.isfinite(d)], initial=0.0) < 1e-7, f"iteration {it}: E-step LL tensor off by {d.max()}"
        rep = parity.compare_fmx(gcells, cells)
        worst = max(worst, rep["max_abs_ll_diff"])
        assert tuple(gstats) == tuple(ostats[:3]), f"iteration {it}: (nsingle, namb, nchanged) {gstats} vs {ostats[:3]}"
        g, c = eng.fmx_cluster_pileup()
        assert np.array_equal(c, np.stack([cplp["nreads"], cplp["nref"], cplp["nalt"]], axis=-1))
        assert np.allclose(g, cplp["gls"], rtol=1e-11, atol=1e-300), f"iteration {it}: cluster pileup differs"
    return worst


@pytest.mark.parametrize("name", ["fmx_k4", "fmx_k4_mixed"])
def test_golden(eng, name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = synth.Pileup(int(z["C"]), int(z["S"]), z["cell_ptr"], z["entry_snp"], z["entry_rptr"], z["reads"], z["af"])
    K = int(z["K"])
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    llk0, llk2, ns, nr = eng.fmx_prepare(p.af)
    assert np.max(np.abs(llk0 - z["llk0"])) < 1e-8 and np.max(np.abs(llk2 - z["llk2"])) < 1e-8
    assert np.array_equal(ns, z["nsnps"]) and np.array_equal(nr, z["nreads"])
    gls, cnt = eng.fmx_entry_gls()
    assert np.array_equal(cnt, z["entry_cnt"]) and np.allclose(gls, z["entry_gls"], rtol=1e-13, atol=1e-300)
    eng.fmx_set_clusters(K, z["clust0"])
    stats = []
    for it in range(int(z["n_iter"])):
        cells, st, full = eng.fmx_iterate(0.5, 0.1, want_full_ll=True)
        if it == 0:
            assert np.max(np.abs(full - z["full_ll_iter1"])) < 1e-7
        stats.append(st)
    assert np.array_equal(np.array(stats), z["stats"])
    parity.compare_fmx(cells, z["cells"])
    g, c = eng.fmx_cluster_pileup()
    assert np.array_equal(c, z["cluster_cnt"]) and np.allclose(g, z["cluster_gls"], rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize("K,C,S,ment,iters", [
    (2, 80, 800, 150, 3),
    (4, 200, 2000, 250, 4),
    (16, 300, 3000, 400, 3),
    (5, 120, 1000, 200, 3),
    (17, 60, 2000, 300, 2),    # 16 < K <= 24: row E-step + broadcast clusters ('row': two clusters per lane, 'wave':
    (18, 40, 2000, 300, 2),    #   ring of 32, 'pair': pair kernel); even / odd rings among the extra clusters
    (20, 100, 2500, 400, 2),
    (21, 40, 2500, 700, 2),
    (24, 40, 2500, 400, 2),    #   (was the last shape of the broadcast-extras kernel)
    (27, 40, 2500, 400, 2),    # 24 < K <= 32: two clusters per lane
    (32, 50, 3000, 2600, 2),   #   full ring, cells in several parts
    (64, 40, 4000, 500, 2),    # config-5 shape, few cells
    (70, 24, 5000, 1200, 2),   # K > 64: the general pair E-step and the (SNP, cluster)-parallel M-step; deep cells
    (100, 16, 5000, 1500, 2),
])
def test_em_trajectory_vs_oracle(eng, K, C, S, ment, iters):
    p = synth.make_pileup(C, S, K, seed=500 + K, mean_entries=ment, min_entries=30, with_gp=False)
    worst = run_em(eng, p, K, iters)
    assert worst < 1e-7


@pytest.mark.parametrize("K,used,C,S,ment", [(6, 3, 300, 400, 8), (16, 5, 400, 600, 6), (40, 6, 150, 500, 10), (4, 4, 500, 300, 3)])
def test_near_tie_calls_are_settled_as_the_reference_does(eng, K, used, C, S, ment):
    """Calls the kernels' numbers cannot decide: clusters WITHOUT cells (identical posteriors: their pairs tie exactly in the
    reference, which keeps the first in scan order), droplets of a handful of entries (hypotheses that differ in the last
    bits or not at all), duplicated droplets.  Every integer field, the three counters and the cluster pileups must equal
    the oracle's over the whole trajectory, and the exact path (fmx_exact.hip) must have been what decided."""
    p = synth.make_pileup(C, S, used, seed=900 + K, mean_entries=ment, min_entries=1, with_gp=False)
    rng = np.random.default_rng(K)
    clust = rng.integers(0, used, p.C).astype(np.int32)   # clusters used .. K-1 stay empty
    clust[rng.random(p.C) < 0.1] = -1
    worst = run_em(eng, p, K, 3, clust=clust)
    near, changed, unresolved = eng.fmx_exact_stats()
    print(f"K={K}: {near} near-tie cell-iterations settled by the exact path, {changed} of them decided differently from the kernels")
    assert worst < 1e-7 and near > 0 and unresolved == 0


@pytest.mark.parametrize("K,C,S,ment,lam", [(4, 60, 5000, 2500, 0.3), (16, 40, 6000, 3000, 0.3), (40, 24, 6000, 2500, 0.3),
                                          (5, 60, 800, 150, 50.0)])
def test_em_trajectory_deep(eng, K, C, S, ment, lam):
    """cells of several thousand entries (several chunks / staging passes of every E-step kernel) and entries of tens to
    hundreds of reads over the whole base-quality range (calculate_snp_droplet_pileup with its per-read normalisation)"""
    p = synth.make_pileup(C, S, min(K, 8), seed=700 + K, mean_entries=ment, min_entries=200 if lam < 1 else 30,
                          with_gp=False, reads_lambda=lam, min_bq=2, max_bq=93, cap_bq=127, other=0.02)
    worst = run_em(eng, p, K, 2)
    assert worst < 1e-7


@pytest.mark.parametrize("K,C,frac,thres", [
    (4, 150, 1.0, -1e300), (6, 100, 0.5, -1e300), (3, 100, 1.0, 5.0),
    (1, 40, 1.0, -1e300),      # a single cluster
    (20, 160, 1.0, -1e300),    # 32 lanes per entry stripe
    (64, 200, 1.0, -1e300),    # one stripe per wave
    (100, 260, 1.0, -1e300),   # clusters spread over two waves
    (4, 20, 1.0, -1e300),      # fewer cells than one batch
])
def test_greedy_init_vs_oracle(geng, K, C, frac, thres):
    eng = geng
    """b3+b4 of the product (muxgl_fmx_greedy_init: host sort + the persistent-workgroup kernel of fmx_greedy.hip) vs
    the oracle's restatement of cmd_cram_freemux2.cpp:217-261"""
    p = synth.make_pileup(C, 1500, min(K, 16), seed=900 + K, mean_entries=200, min_entries=30, with_gp=False)
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    llk0, llk2, _, _ = eng.fmx_prepare(p.af)
    e = ob.fmx_entry_pileup(p)
    o0, o2, _, _ = ob.fmx_cell_scores(p, e)
    scores = o2 - o0  # same scores on both sides: the sort must not depend on last-ulp differences of the GPU sums
    want = ob.fmx_greedy_init(p, e, K, scores, ob.fmx_sort(scores), frac, thres)
    got = eng.fmx_greedy_init(K, scores, frac, thres)
    assert np.array_equal(got, want)
    assert np.max(np.abs((llk2 - llk0) - scores)) < 1e-8


@pytest.mark.parametrize("K,C,S,me", [(8, 1500, 3000, 300), (48, 700, 1200, 200), (3, 400, 150, 100), (64, 96, 300, 250)])
def test_greedy_init_many_batches(geng, K, C, S, me):
    eng = geng
    """many batches of the batched kernel (32 cells each), dense SNP sharing between the cells of a batch (every cell
    of the third case covers two thirds of the markers, so most of its terms are corrected; in the fourth a cell has
    several thousand predecessors in its batch -- more than the kernel keeps in LDS -- and chains of more than eight)"""
    p = synth.make_pileup(C, S, min(K, 12), seed=31 + K, mean_entries=me, min_entries=20, with_gp=False)
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    eng.fmx_prepare(p.af)
    e = ob.fmx_entry_pileup(p)
    o0, o2, _, _ = ob.fmx_cell_scores(p, e)
    scores = o2 - o0
    want = ob.fmx_greedy_init(p, e, K, scores, ob.fmx_sort(scores))
    got = eng.fmx_greedy_init(K, scores)
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]


def test_greedy_init_deep_cells(geng):
    eng = geng
    """cells with more entries than one staging pass of the kernel (1024), next to empty and tiny ones"""
    K = 5
    p = synth.make_pileup(60, 6000, K, seed=77, mean_entries=1500, min_entries=0, with_gp=False)
    assert np.diff(p.cell_ptr).max() > 1024
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    eng.fmx_prepare(p.af)
    e = ob.fmx_entry_pileup(p)
    o0, o2, _, _ = ob.fmx_cell_scores(p, e)
    scores = o2 - o0
    want = ob.fmx_greedy_init(p, e, K, scores, ob.fmx_sort(scores))
    assert np.array_equal(eng.fmx_greedy_init(K, scores), want)


def test_init_cluster_with_unassigned_cells_and_params(eng):
    """--init-cluster style start with some cells unassigned (-1), non-default priors, geno_error = 0"""
    K = 3
    p = synth.make_pileup(90, 900, K, seed=42, mean_entries=200, min_entries=30, with_gp=False)
    clust = p.truth["s1"].astype(np.int32).copy()
    clust[::7] = -1
    run_em(eng, p, K, 3, clust=clust, doublet_prior=0.3, geno_error=0.0)
    run_em(eng, p, K, 2, clust=clust, doublet_prior=0.5, geno_error=0.25)


def test_ragged_cells(eng):
    """empty cells, single-entry cells, entries without reads / with only 'other' alleles, one deep SNP chain"""
    rng = np.random.default_rng(11)
    S, K = 300, 3
    lens = [0, 1, 2, 200, 0, 60, 150, 1, 90]
    cell_ptr = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=cell_ptr[1:])
    nnz = int(cell_ptr[-1])
    snps = []
    for n in lens:
        s = np.sort(rng.choice(S - 1, n, replace=False)) + 1 if n else np.zeros(0, dtype=np.int64)
        if n >= 60:
            s[0] = 0  # SNP 0 is covered by every long cell: a chain with several merges
        snps.append(s)
    entry_snp = np.concatenate(snps).astype(np.int32)
    nreads = rng.integers(0, 4, size=nnz)
    entry_rptr = np.zeros(nnz + 1, dtype=np.int64)
    np.cumsum(nreads, out=entry_rptr[1:])
    R = int(entry_rptr[-1])
    reads = ((rng.integers(0, 2, R) << 7) | rng.integers(13, 21, R)).astype(np.uint8)
    reads[rng.random(R) < 0.1] = 0xFF
    af = rng.uniform(0.05, 0.95, S)
    p = synth.Pileup(len(lens), S, cell_ptr, entry_snp, entry_rptr, reads, af)
    clust = np.array([0, 1, 2, 0, -1, 1, 2, 0, 1], dtype=np.int32)
    run_em(eng, p, K, 3, clust=clust)


def test_errors(eng):
    p = synth.make_pileup(10, 100, 2, seed=1, mean_entries=30, min_entries=5, with_gp=False)
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    with pytest.raises(muxgl.MuxglError):
        eng.fmx_set_clusters(2, np.zeros(p.C, dtype=np.int32))  # prepare not called
    eng.fmx_prepare(p.af)
    with pytest.raises(muxgl.MuxglError):
        eng.fmx_set_clusters(2, np.full(p.C, 2, dtype=np.int32))  # cluster id >= K
    with pytest.raises(muxgl.MuxglError):
        eng.fmx_iterate()  # no clusters


# ---- BASELINE.json configs[3] shape at reduced cell count: size-independent properties -------------------------

def test_em_cells_are_independent_given_clusters(eng):
    """E-step records of a cell depend only on the cell and the cluster pileups: running a subset of cells against
    the same initial assignment of ALL cells is not possible through the ABI, so check the weaker property that two
    identical runs are bit-identical (deterministic chunk and chain order) and that cluster relabelling permutes calls"""
    K = 4
    p = synth.make_pileup(400, 3000, K, seed=77, mean_entries=300, with_gp=False)
    e, scores, clust, cplp, cells = oracle_init(p, K)
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    eng.fmx_prepare(p.af)
    eng.fmx_set_clusters(K, clust)
    a1, s1 = eng.fmx_iterate()
    a2, s2 = eng.fmx_iterate()
    eng.fmx_set_clusters(K, clust)
    b1, t1 = eng.fmx_iterate()
    b2, t2 = eng.fmx_iterate()
    assert a1.tobytes() == b1.tobytes() and a2.tobytes() == b2.tobytes() and s1 == t1 and s2 == t2
    perm = np.array([2, 0, 3, 1], dtype=np.int32)
    eng.fmx_set_clusters(K, np.where(clust >= 0, perm[np.clip(clust, 0, K - 1)], -1).astype(np.int32))
    c1, u1 = eng.fmx_iterate()
    assert u1 == s1
    assert np.array_equal(c1["type"], a1["type"])
    sng = a1["type"] == 0
    assert np.array_equal(c1["clust"][sng], perm[a1["clust"][sng]])
    assert np.max(np.abs(c1["sngBestLLK"] - a1["sngBestLLK"])) < 1e-8


@pytest.mark.parametrize("C,S,K,me", [(3000, 400, 8, 12), (3000, 2000, 12, 25), (2000, 300, 6, 8), (4000, 800, 16, 15)])
def test_greedy_init_near_ties(geng, C, S, K, me):
    """Decisions that feed later state are taken on re-associated arithmetic (lk0 factorised as A * B, products kept as
    mantissa x exponent, reciprocal multiplies in merge(), ratios of replayed states in the batched kernels), the argmax
    is strict and every assignment changes what the next cell is scored against: a last-ulp flip on a near tie could in
    principle send the clustering down another path than the reference's.  Low-coverage cells (a dozen entries, scores
    near the exact 0 of an empty cluster) are where margins are smallest: count the assignments that differ from the
    oracle's sequential sum of logs.  Observed: none, on either kernel."""
    p = synth.make_pileup(C, S, K, seed=500 + K, mean_entries=me, min_entries=3, with_gp=False, sigma=0.8)
    e = ob.fmx_entry_pileup(p)
    o0, o2, _, _ = ob.fmx_cell_scores(p, e)
    scores = o2 - o0
    want = ob.fmx_greedy_init(p, e, K, scores, ob.fmx_sort(scores))
    geng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    geng.fmx_prepare(p.af)
    got = geng.fmx_greedy_init(K, scores)
    flips = np.flatnonzero(got != want)
    assert flips.size == 0, f"{flips.size} of {C} assignments differ from the oracle, first at cells {flips[:5]}"


def _read_score_dump(path, K):
    rec = np.dtype([("step", np.int64), ("sc", np.float64, (K,))])
    return np.fromfile(path, dtype=rec)


@pytest.mark.parametrize("flags", [0, muxgl.FLAG_FORCE_TILE_SWEEP])
@pytest.mark.parametrize("K,C,S,me", [(4, 150, 800, 150), (16, 120, 1500, 250), (7, 100, 300, 60)])
def test_greedy_exact_path_is_the_references_arithmetic(tmp_path, monkeypatch, flags, K, C, S, me):
    """Every step through the exact path (greedy_exact.hpp; MUXGL_GREEDY_TIE_EPS = 1e300 makes every margin a "near
    tie"): its K distances must be the oracle's -- i.e. the reference's, tests/test_oracle_ref.py -- BIT FOR BIT, since
    it claims to redo the reference's arithmetic (IEEE operations in the reference's order on the device, glibc log on the
    host), and the clustering must be the oracle's.  Both kernels (batched and serial) deliver the flags."""
    p = synth.make_pileup(C, S, min(K, 8), seed=300 + K, mean_entries=me, min_entries=3, with_gp=False, reads_lambda=1.0,
                          other=0.03)
    e = ob.fmx_entry_pileup(p)
    o0, o2, _, _ = ob.fmx_cell_scores(p, e)
    scores = o2 - o0
    want, want_sc = ob.fmx_greedy_init_scores(p, e, K, scores, ob.fmx_sort(scores))
    dump = str(tmp_path / "scores.bin")
    monkeypatch.setenv("MUXGL_TEST_HOOKS", "1")
    monkeypatch.setenv("MUXGL_GREEDY_TIE_EPS", "1e300")
    monkeypatch.setenv("MUXGL_GREEDY_DUMP_SCORES", dump)
    with muxgl.Engine(0, flags) as en:
        en.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        en.fmx_prepare(p.af)
        got = en.fmx_greedy_init(K, scores)
        near, over = en.fmx_greedy_stats()
    assert np.array_equal(got, want)
    d = _read_score_dump(dump, K)
    assert over == 0 and near == d.size and near > 0.8 * C   # (steps whose scores are all exactly 0 are not flagged)
    assert np.array_equal(d["sc"], want_sc[d["step"]]), "the exact path's distances are not the reference's bits"


@pytest.mark.parametrize("flags", [0, muxgl.FLAG_FORCE_TILE_SWEEP])
def test_greedy_exact_path_overrules_a_wrong_decision(monkeypatch, flags):
    """MUXGL_GREEDY_TEST_MISDECIDE makes the kernel take the wrong cluster at one step and flag it: the exact path must
    overrule it, force the reference's choice and repeat the run -- the clustering is the oracle's again"""
    K = 6
    p = synth.make_pileup(400, 1200, K, seed=333, mean_entries=150, min_entries=10, with_gp=False)
    e = ob.fmx_entry_pileup(p)
    o0, o2, _, _ = ob.fmx_cell_scores(p, e)
    scores = o2 - o0
    want = ob.fmx_greedy_init(p, e, K, scores, ob.fmx_sort(scores))
    monkeypatch.setenv("MUXGL_TEST_HOOKS", "1")
    monkeypatch.setenv("MUXGL_GREEDY_TEST_MISDECIDE", "137")
    with muxgl.Engine(0, flags) as en:
        en.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        en.fmx_prepare(p.af)
        got = en.fmx_greedy_init(K, scores)
        near, over = en.fmx_greedy_stats()
    assert np.array_equal(got, want)
    assert near >= 1 and over == 1


def test_greedy_default_run_reports_its_near_ties(geng):
    """an ordinary run: few or no near ties, none overruled, and the getter says so"""
    K = 8
    p = synth.make_pileup(1200, 2000, K, seed=444, mean_entries=200, min_entries=10, with_gp=False)
    e = ob.fmx_entry_pileup(p)
    o0, o2, _, _ = ob.fmx_cell_scores(p, e)
    scores = o2 - o0
    want = ob.fmx_greedy_init(p, e, K, scores, ob.fmx_sort(scores))
    geng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    geng.fmx_prepare(p.af)
    got = geng.fmx_greedy_init(K, scores)
    near, over = geng.fmx_greedy_stats()
    assert np.array_equal(got, want)
    assert near <= 3 and over == 0


@pytest.mark.parametrize("K,C,S,me", [(3, 2500, 60, 30), (16, 3000, 50, 25), (20, 2500, 40, 20), (32, 4000, 64, 30),
                                      (40, 3000, 48, 24), (64, 6000, 70, 40), (16, 64, 40, 30)])
def test_mstep_stream_vs_plain_chains_and_oracle(K, C, S, me):
    """Ordered clamped M-step (sc_drop_seq.h:77-101 driven by cmd_cram_freemux2.cpp:586-597) on markers covered by
    hundreds to thousands of cells -- lists of many batches per chain, clusters of very different sizes, unassigned
    cells: the stream kernel (lane = (marker, cluster) chain fed from a staged stream, fmx_mstep.hip) and the plain kernel
    (every chain walks the list on its own; MUXGL_FLAG_FORCE_TILE_SWEEP) agree to rounding with each other and with the
    oracle's cluster pileups."""
    p = synth.make_pileup(C, S, min(K, 8), seed=900 + K + C, mean_entries=me, min_entries=5, with_gp=False)
    rng = np.random.default_rng(K * 1000 + C)
    w = rng.dirichlet(np.full(K, 0.6))  # unbalanced clusters: some chains much longer than a batch, some empty
    clust = rng.choice(K, size=p.C, p=w).astype(np.int32)
    clust[rng.random(p.C) < 0.07] = -1
    e = ob.fmx_entry_pileup(p)
    want = ob.fmx_build_cluster_pileup(p, e, K, clust)
    got = []
    for flags in (0, muxgl.FLAG_FORCE_TILE_SWEEP):
        with muxgl.Engine(0, flags) as en:
            en.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
            en.fmx_prepare(p.af)
            en.fmx_set_clusters(K, clust)
            g, c = en.fmx_cluster_pileup()
            assert np.array_equal(c, np.stack([want["nreads"], want["nref"], want["nalt"]], axis=-1))
            assert np.allclose(g, want["gls"], rtol=1e-11, atol=1e-300)
            cells, _ = en.fmx_iterate(0.5, 0.1)  # the M-step behind a re-assignment
            g2, _ = en.fmx_cluster_pileup()
            got.append((g, g2, cells))
    assert np.array_equal(got[0][2]["clust"], got[1][2]["clust"])   # the same assignments went into the second merge
    assert np.allclose(got[0][0], got[1][0], rtol=1e-11, atol=1e-300) and np.allclose(got[0][1], got[1][1], rtol=1e-11, atol=1e-300)


def test_mstep_stream_without_the_lds_table(tmp_path):
    """Beyond ~60 k cells the assignments do not fit next to the staging areas in LDS and the stream M-step gathers them
    from global memory (one wave per workgroup): forced here on small inputs through MUXGL_MSTEP_NO_TABLE in a process
    of its own (the library reads the variable once), bit-identical to the same kernel with the table (run in a second
    process without the variable)."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from popscle_amd import muxgl, synth
for K, C, S, me in [(5, 2000, 40, 20), (16, 3000, 50, 25), (24, 2500, 40, 20), (48, 3000, 48, 24)]:
    p = synth.make_pileup(C, S, min(K, 8), seed=77 + K, mean_entries=me, min_entries=5, with_gp=False)
    rng = np.random.default_rng(K)
    clust = rng.choice(K, size=p.C, p=rng.dirichlet(np.full(K, 0.6))).astype(np.int32)
    clust[rng.random(p.C) < 0.07] = -1
    with muxgl.Engine(0) as en:
        en.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        en.fmx_prepare(p.af)
        en.fmx_set_clusters(K, clust)
        g, c = en.fmx_cluster_pileup()
        en.fmx_iterate(0.5, 0.1)
        g2, _ = en.fmx_cluster_pileup()
        import hashlib
        print("digest", K, hashlib.sha256(g.tobytes() + g2.tobytes()).hexdigest())
print("ok")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra in ({"MUXGL_MSTEP_NO_TABLE": "1"}, {}):
        env = {k: v for k, v in os.environ.items() if k != "MUXGL_MSTEP_NO_TABLE"}
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("digest")])
    assert len(outs[0]) == 4 and outs[0] == outs[1]


@pytest.mark.parametrize("K,C,S,me", [(1, 40, 3, 3), (2, 50, 2, 2), (33, 30, 700, 120), (16, 20, 1, 1)])
def test_em_edge_shapes(K, C, S, me):
    """one cluster, fewer markers than a wave holds, a ring with 31 idle lanes, a single marker; every second run starts with
    all cells unassigned (every chain of the M-step empty) -- default dispatch against the oracle"""
    p = synth.make_pileup(C, S, min(K, 4), seed=4000 + K + S, mean_entries=me, min_entries=1, with_gp=False)
    with muxgl.Engine(0) as en:
        run_em(en, p, K, 2)
        run_em(en, p, K, 2, clust=np.full(p.C, -1, dtype=np.int32))
