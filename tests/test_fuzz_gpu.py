"""Randomised differential test: libmuxgl (HIP, through the C-ABI) against the reference's own compiled loops
(oracle/_ref/libscdrop_ref.so) on small problems drawn to be UNFRIENDLY -- cells of one to a handful of entries (most
hypotheses tie exactly in the reference's arithmetic, so every call is a question of scan order and strict '<'),
duplicated samples, markers without genotypes, entries whose reads are all of another allele, deep entries, uncapped
qualities, genotype rows that do not sum to one, unusual alpha grids and priors, every kernel family by shape.

Bar: parity.compare_* -- every integer field equal, log-likelihoods within 1e-5 (the asserts below hold them to 1e-7).

pytest runs the seeds of FUZZ_SEEDS (a minute); a campaign is
    python tests/test_fuzz_gpu.py --seeds 1000:1400 [--kind demux|fmx] [--log gpurun_out/fuzz.jsonl]
which prints one JSON line per case and exits 1 at the first failing seed (the seed reproduces the case).
"""
import json
import os
import sys
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import parity  # noqa: E402
import ref_binding as rb  # noqa: E402
from popscle_amd import freemuxlet, muxgl, shard, synth  # noqa: E402

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not rb.available(), reason="oracle/_ref/libscdrop_ref.so not built")]

GRIDS = [(0.0, 0.5)] * 5 + [(0.0, 0.1, 0.2, 0.3, 0.4, 0.5)] * 2 + [(0.0, 0.25, 0.5), (0.0, 0.1, 0.3), (0.0, 0.5, 0.2),
                                                                   (0.0,), (0.0, 0.3)]
DEMUX_V = [1, 2, 3, 4, 5, 8, 12, 15, 16, 16, 16, 17, 24, 31, 32, 33, 40, 64, 65, 70, 96, 130, 255]
FMX_K = [2, 3, 4, 8, 15, 16, 16, 17, 24, 32, 33, 64, 65, 100, 130, 255]
FUZZ_SEEDS = list(range(12))
# kernel families by flag (include/muxgl.h), drawn per case: the default dispatch most of the time
DEMUX_FLAGS = [0] * 6 + [muxgl.FLAG_FORCE_ROW_KERNEL, muxgl.FLAG_FORCE_WAVE_KERNEL, muxgl.FLAG_FORCE_TILE_SWEEP,
                         muxgl.FLAG_NO_LINEAR_ENTRIES, muxgl.FLAG_SPLIT_GENERAL_SWEEP,
                         muxgl.FLAG_FORCE_ROW_KERNEL | muxgl.FLAG_NO_LINEAR_ENTRIES]
FMX_FLAGS = [0] * 6 + [muxgl.FLAG_FORCE_ROW_KERNEL, muxgl.FLAG_FORCE_WAVE_KERNEL, muxgl.FLAG_NO_LINEAR_ENTRIES,
                       muxgl.FLAG_NO_PIVOT_SUMS, muxgl.FLAG_FORCE_WAVE_KERNEL | muxgl.FLAG_NO_PIVOT_SUMS]


def _shape(r, width, per_hyp):
    """(C, S, mean entries, min entries, sigma): tiny cells more often than not; C bounded by the reference's run time"""
    ment = float(r.choice([1.5, 3, 8, 30, 120, 400], p=[0.2, 0.2, 0.2, 0.15, 0.15, 0.1]))
    if width > 80:   # (the widest shapes: the reference's time goes with the square of the width)
        ment = min(ment, 8.0)
    S = int(max(40, ment * float(r.choice([2, 8, 40]))))
    budget = 2.5e8
    C = int(np.clip(budget / (max(ment, 4) * per_hyp), max(8, 2 * width if width <= 80 else 60), 600))
    return C, S, ment, int(r.choice([0, 1, 2])), float(r.choice([0.3, 0.8, 1.3]))


def _pileup(r, seed, C, S, V, ment, mine, sigma, with_gp):
    cap = int(r.choice([20, 20, 40, 60, 93, 127]))
    return synth.make_pileup(C, S, V, seed=seed, mean_entries=ment, sigma=sigma, min_entries=mine,
                             reads_lambda=float(r.choice([0.0, 0.3, 0.3, 1.5, 6.0, 6.0, 60.0])),
                             other=float(r.choice([0.0, 0.02, 0.5])), doublet_frac=float(r.choice([0.0, 0.25, 0.6])),
                             flip=float(r.choice([0.0, 0.01, 0.2])), max_bq=cap, cap_bq=cap,
                             missing_gp_frac=float(r.choice([0.0, 0.0, 0.03, 0.5, 1.0])) if with_gp else 0.0, with_gp=with_gp)


def demux_case(seed):
    r = np.random.default_rng([seed, 77])
    V = int(r.choice(DEMUX_V))
    alphas = GRIDS[int(r.integers(len(GRIDS)))]
    C, S, ment, mine, sigma = _shape(r, V, V * V * len(alphas) * 9)
    p = _pileup(r, 3000 + seed, C, S, V, ment, mine, sigma, True)
    mode = str(r.choice(["gt", "gt", "dup", "all_same", "float_rows", "hard"]))
    gp = p.gp
    if mode == "dup" and V > 1:       # some samples are copies of others: exact ties between hypotheses
        for _ in range(max(1, V // 3)):
            a, b = r.integers(V, size=2)
            gp[:, a, :] = gp[:, b, :]
    elif mode == "all_same":          # every sample the same: every singlet ties, every pair ties
        gp[:, :, :] = gp[:, :1, :]
    elif mode == "hard":              # hard calls with almost no error mixed in: factors down to 1e-6 x 1e-10
        gp = synth.gt_to_gp(p.truth["G"].astype(np.int64), 1e-6)
    elif mode == "float_rows":        # rows as --field GP leaves them: normalised in float, sums off one by ~1e-8
        g = r.dirichlet([0.3, 0.3, 0.3], size=(p.S, V)).astype(np.float32) + np.float32(1e-4)
        g = g / g.sum(axis=2, keepdims=True, dtype=np.float32)
        gp = 0.9 * g.astype(np.float64) + 0.1 * g.astype(np.float64).mean(axis=1, keepdims=True)
    p.gp = np.ascontiguousarray(gp)
    dp = float(r.choice([0.5, 0.5, 0.1, 0.9]))
    flags = int(DEMUX_FLAGS[int(r.integers(len(DEMUX_FLAGS)))])
    how = str(r.choice(["one", "one", "one", "group"]))   # group: a device group of two members on this GPU
    return dict(kind="demux", seed=seed, V=V, alphas=alphas, C=C, S=S, ment=ment, mode=mode, dp=dp, flags=flags,
                how=how), p


def _engine(eng, info):
    """the shared engine for the default dispatch on one device, else one made for the case (closed by the caller)"""
    if info.get("how", "one") == "group":
        return muxgl.Engine([0, 0], flags=info["flags"]), True
    if info["flags"]:
        return muxgl.Engine(0, flags=info["flags"]), True
    return eng, False


def run_demux(eng, info, p):
    eng, own = _engine(eng, info)
    try:
        return _run_demux(eng, info, p)
    finally:
        if own:
            eng.close()


def _run_demux(eng, info, p):
    alphas, dp, V = info["alphas"], info["dp"], info["V"]
    want, _, want_ll = rb.RefScl.from_packed(p).demux(alphas, doublet_prior=dp, full_ll=True)
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    eng.demux_set_gp(p.gp, p.has_gp)
    got = eng.demux_run(alphas, dp)                      # the product path (calls made next to the sweep)
    rep = parity.compare_demux(got, want, alphas, p, doublet_prior=dp)
    if info.get("how", "one") == "group":
        assert rep["max_abs_ll_diff"] < 1e-7, rep
        st = rep["exact_pass"]
        return dict(ll=rep["max_abs_ll_diff"], looked_at=int(st["cells"]), near=int(st["near_ties"]),
                    deep=int(st["deep"]), changed=int(st["changed"]), raw_differing=rep["raw_records_differing"])
    got2, full = eng.demux_run(alphas, dp, want_full_ll=True)   # the tensor path
    rep2 = parity.compare_demux(got2, want, alphas, p, doublet_prior=dp)
    worst = parity.compare_full_ll(full, want_ll, V, alphas)
    assert rep["max_abs_ll_diff"] < 1e-7 and rep2["max_abs_ll_diff"] < 1e-7 and worst < 1e-7, (rep, rep2, worst)
    st = rep["exact_pass"]
    return dict(ll=max(rep["max_abs_ll_diff"], worst), looked_at=int(st["cells"]), near=int(st["near_ties"]),
                deep=int(st["deep"]), changed=int(st["changed"]), raw_differing=rep["raw_records_differing"])


def fmx_case(seed):
    r = np.random.default_rng([seed, 78])
    K = int(r.choice(FMX_K))
    C, S, ment, mine, sigma = _shape(r, 2 * K, K * K * 9 * 6)
    C = max(C, 3 * K) if K <= 70 else K + 20   # (beyond 70 clusters the reference takes seconds per hundred cells)
    p = _pileup(r, 5000 + seed, C, S, max(2, int(r.choice([K, max(2, K // 2), K + 1]))), ment, mine, sigma, False)
    dp = float(r.choice([0.5, 0.5, 0.1]))
    ge = float(r.choice([0.1, 0.1, 0.01]))
    flags = int(FMX_FLAGS[int(r.integers(len(FMX_FLAGS)))])
    # one: muxgl_fmx_iterate on one handle; group: a device group of two members on this GPU; shard: the phases of a
    # multi-rank run driven by hand, two or three handles as ranks, the exact path across them (freemuxlet.settle_near_ties)
    how = str(r.choice(["one", "one", "one", "group", "shard"]))
    # the start: the greedy pass over every cell; over a fraction of them / above a score threshold (the others start
    # without a cluster); or clusters handed in (--init-cluster), some cells without one
    start = str(r.choice(["greedy", "greedy", "greedy", "partial", "given"]))
    frac = float(r.choice([0.3, 0.8])) if start == "partial" else 1.0
    thres = float(r.choice([-1e300, -0.5, 0.0])) if start == "partial" else -1e300
    init = None
    if start == "given":
        init = r.integers(-1 if r.random() < 0.7 else 0, K, size=p.C).astype(np.int32)
    return dict(kind="fmx", seed=seed, K=K, C=C, S=S, ment=ment, dp=dp, ge=ge, flags=flags, how=how,
                world=int(r.choice([2, 3])), start=start, frac=frac, thres=thres, init=init), p


def _allgather(engs, which, ranges, row_bytes):
    for owner, (b, e) in enumerate(ranges):
        if e <= b:
            continue
        src, _ = engs[owner].fmx_buffer(which)
        for r, other in enumerate(engs):
            if r != owner:
                dst, _ = other.fmx_buffer(which)
                other.memcpy_dev(dst + b * row_bytes, src + b * row_bytes, (e - b) * row_bytes)


def _check_iteration(it, ref, cells, st, K):
    rep = parity.compare_fmx(cells, ref["cells"][it])
    assert tuple(int(x) for x in st) == tuple(ref["counters"][it]), (it, st, ref["counters"][it])
    return rep


def _check_cplp(g, c, w, s0=None, s1=None):
    sl = slice(s0, s1)
    assert np.array_equal(c[:, sl], np.stack([w["nreads"], w["nref"], w["nalt"]], axis=-1)[:, sl])
    assert np.allclose(g[:, sl], w["gls"][:, sl], rtol=1e-11, atol=1e-300)


def run_fmx(eng, info, p):
    K, dp, ge, how = info["K"], info["dp"], info["ge"], info.get("how", "one")
    init = info.get("init")
    ref = rb.RefScl.from_packed(p).freemux2(K, doublet_prior=dp, geno_error=ge, frac_init_clust=info.get("frac", 1.0),
                                            singlet_score_thres=info.get("thres", -1e300), init_clust=init, full_ll=True,
                                            cluster_pileups=True)
    # the start on one device with the whole pileup (the greedy pass is sequential over all cells)
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    llk0, llk2, ns, nr = eng.fmx_prepare(p.af)
    assert np.max(np.abs(llk0 - ref["llk0"])) < 1e-7 and np.max(np.abs(llk2 - ref["llk2"])) < 1e-7
    assert np.array_equal(ns, ref["nsnps"]) and np.array_equal(nr, ref["nreads"])
    if init is None:
        clust = eng.fmx_greedy_init(K, llk2 - llk0, info.get("frac", 1.0), info.get("thres", -1e300))
        assert np.array_equal(clust, ref["clust0"]), ("greedy start", np.flatnonzero(clust != ref["clust0"])[:5])
    else:
        clust = init
    out = dict(iters=int(ref["n_iter"]), exact_scores=int(eng.fmx_score_stats()),
               greedy_near=[int(x) for x in eng.fmx_greedy_stats()])
    worst, near = 0.0, 0
    if how == "shard":
        world = info["world"]
        engs = [muxgl.Engine(0, flags=info["flags"]) for _ in range(world)]
        try:
            c_ranges = shard.cell_shards(p.cell_ptr, world)
            s_ranges = shard.snp_shards(p.entry_snp, p.S, world)
            for r, e in enumerate(engs):
                e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
                e.fmx_prepare(p.af)
                e.fmx_set_shard(*c_ranges[r], *s_ranges[r])
                e.fmx_set_clusters(K, clust)
            for it in range(ref["n_iter"]):
                for e in engs:
                    e.fmx_iter_gp(dp, ge)
                _allgather(engs, muxgl.BUF_CGP, s_ranges, K * 3 * 8)
                for e in engs:
                    e.fmx_iter_estep(dp, ge)
                fetched = [e.fmx_iter_fetch() for e in engs]
                if sum(e.fmx_exact_pending() for e in engs) > 0:
                    freemuxlet.settle_near_ties(engs, lambda obj: [obj], dp, ge)
                    fetched = [e.fmx_iter_fetch() for e in engs]
                _allgather(engs, muxgl.BUF_CLUST, c_ranges, 4)
                for e in engs:
                    e.fmx_iter_mstep()
                cells = np.zeros(p.C, dtype=muxgl.FMX_CELL)
                stats = np.zeros(3, dtype=np.int64)
                for r, (cs, st) in enumerate(fetched):
                    b, en = c_ranges[r]
                    cells[b:en] = cs[b:en]
                    stats += np.array(st)
                rep = _check_iteration(it, ref, cells, stats, K)
                worst, near = max(worst, rep["max_abs_ll_diff"]), near + rep["near_tie_cells"]
                for r, e in enumerate(engs):
                    g, c = e.fmx_cluster_pileup()
                    _check_cplp(g, c, ref["cplp"][it], *s_ranges[r])
            out["exact"] = [int(sum(e.fmx_exact_stats()[k] for e in engs)) for k in range(3)]
        finally:
            for e in engs:
                e.close()
    else:
        run, own = _engine(eng, info)
        try:
            if own:
                run.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
                g0, g2, _, _ = run.fmx_prepare(p.af)
                # (a device group settles near-tied scores as one device does: the same bits)
                assert np.array_equal(g0, llk0) and np.array_equal(g2, llk2), "scores of the group / flagged engine differ"
            run.fmx_set_clusters(K, clust)
            for it in range(ref["n_iter"]):
                if how == "group":
                    cells, st = run.fmx_iterate(dp, ge)
                else:
                    cells, st, full = run.fmx_iterate(dp, ge, want_full_ll=True)
                    d = np.abs(full - ref["full_ll"][it])
                    d = d[np.isfinite(d)]
                    worst = max(worst, float(d.max()) if d.size else 0.0)
                rep = _check_iteration(it, ref, cells, st, K)
                worst, near = max(worst, rep["max_abs_ll_diff"]), near + rep["near_tie_cells"]
                g, c = run.fmx_cluster_pileup()
                _check_cplp(g, c, ref["cplp"][it])
            out["exact"] = [int(x) for x in run.fmx_exact_stats()]
        finally:
            if own:
                run.close()
    assert worst < 1e-7, worst
    out.update(ll=worst, near=near)
    return out


def run_case(eng, kind, seed):
    info, p = (demux_case if kind == "demux" else fmx_case)(seed)
    info["nnz"] = int(p.nnz)
    info.update((run_demux if kind == "demux" else run_fmx)(eng, info, p))
    info.pop("init", None)
    return info


@pytest.fixture(scope="module")
def eng():
    e = muxgl.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("seed", FUZZ_SEEDS)
def test_demuxlet_fuzz(eng, seed):
    run_case(eng, "demux", seed)


@pytest.mark.parametrize("seed", FUZZ_SEEDS)
def test_freemuxlet_fuzz(eng, seed):
    run_case(eng, "fmx", seed)


def test_unfiltered_droplets_among_cells():
    """a few hundred cells among thousands of droplets of one to a handful of reads (tests/stress_droplets.py): scores of
    0 +- rounding noise, noise-level ties in the greedy pass and in every iteration -- all of it the reference's"""
    import stress_droplets

    assert stress_droplets.main(["300", "3000", "8", "30000"]) == 0


def main(argv):
    import argparse
    import traceback

    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:50")
    ap.add_argument("--kind", default="both", choices=["both", "demux", "fmx"])
    ap.add_argument("--log", default=None)
    ap.add_argument("--keep-going", action="store_true")
    ap.add_argument("--budget-s", type=float, default=1e9, help="stop starting new cases after this many seconds")
    a = ap.parse_args(argv)
    lo, hi = (int(x) for x in a.seeds.split(":"))
    e = muxgl.Engine(0)
    log = open(a.log, "a") if a.log else None
    t0, fails, n = time.time(), 0, 0
    for seed in range(lo, hi):
        for kind in (("demux", "fmx") if a.kind == "both" else (a.kind,)):
            if time.time() - t0 > a.budget_s:
                break
            t = time.time()
            try:
                rec = run_case(e, kind, seed)
                rec["ok"] = True
            except Exception as ex:  # noqa: BLE001  (a campaign reports and goes on or stops, as asked)
                info, _ = (demux_case if kind == "demux" else fmx_case)(seed)
                info.pop("init", None)
                rec = dict(info, ok=False, error=f"{type(ex).__name__}: {str(ex)[:600]}",
                           where=traceback.format_exc().strip().splitlines()[-3:])
                fails += 1
            rec["s"] = round(time.time() - t, 2)
            n += 1
            line = json.dumps(rec, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
            print(line, flush=True)
            if log:
                log.write(line + "\n")
                log.flush()
            if fails and not a.keep_going:
                return 1
    print(json.dumps({"cases": n, "failed": fails, "seconds": round(time.time() - t0, 1)}), flush=True)
    if log:
        log.write(json.dumps({"cases": n, "failed": fails, "seconds": round(time.time() - t0, 1)}) + "\n")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
