"""Parity comparison helpers shared by the GPU tests, smoke() and bench.py's self-check.

Bar (BASELINE.json north_star): best-sample / doublet calls exact, log-likelihoods within 1e-5 absolute.

"Exact" means exact: every integer field of a record -- droplet type, best / next singlet, best / next doublet in the
ORDER the reference names its two samples, the derived guesses -- must equal the reference's, with no tie window and no
canonical pair order.  The kernels' log-likelihoods equal the reference's to ~1e-12, not to the last bit, so the records
are compared after the product's own exact-call pass (popscle_amd/host/exact_calls.hpp = muxgl_demux_exact_calls, what
`popscle-amd demuxlet` runs before it writes .best): it recomputes, in the reference's arithmetic, the hypotheses of every
cell where a comparison's margin is within rounding reach and leaves all other cells untouched -- for those the raw
device record is what is compared.
"""
from __future__ import annotations

import numpy as np

LL_TOL = 1e-5  # absolute, on log-likelihoods (north_star)

DEMUX_LL_FIELDS = ("sngBestLLK", "sngNextLLK", "dblBestLLK", "dblNextLLK", "sumLLK", "sngLLK", "bestLLK", "nextLLK")
DEMUX_PP_FIELDS = ("bestPP", "sngPP", "sngOnlyPP")
DEMUX_INT_FIELDS = ("valid", "nsnps", "type", "next_type", "sBest", "sNext", "dBest1", "dBest2", "dBestA", "dNext1",
                    "dNext2", "dNextA", "jBest", "kBest", "aBest", "jNext", "kNext", "aNext")


def _close(x, y, tol):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    same_inf = np.isinf(x) & np.isinf(y) & (np.sign(x) == np.sign(y))
    both_nan = np.isnan(x) & np.isnan(y)
    with np.errstate(invalid="ignore"):
        return same_inf | both_nan | (np.abs(x - y) <= tol)


def exact(got, alphas, p, doublet_prior=0.5, nthreads=0):
    """a copy of the raw device records `got` after the product's exact-call pass"""
    from popscle_amd import muxgl

    out = np.ascontiguousarray(got).copy()
    muxgl.demux_exact_calls(p, alphas, out, doublet_prior, nthreads=nthreads)
    return out


def compare_demux(got, want, alphas, p, tol=LL_TOL, doublet_prior=0.5, nthreads=0):
    """Compare [C] demux records of muxgl_demux_run over pileup `p` (numpy structured arrays with the muxgl_demux_cell
    fields; `got` is not modified) with the reference's / oracle's records `want`.

    Runs the product's exact-call pass on a copy of `got`, then requires every integer field to be EQUAL and every
    log-likelihood / posterior within `tol`.  Returns a report: the largest LL deviation, the pass's counters, and how
    many cells' raw device records differed from the reference before the pass (all of them cells the pass looked at).
    Raises AssertionError on any violation.
    """
    from popscle_amd import muxgl

    assert got.shape == want.shape == (p.C,)
    raw = got
    got = np.ascontiguousarray(got).copy()
    st = muxgl.demux_exact_calls(p, alphas, got, doublet_prior, nthreads=nthreads)
    v = want["valid"] == 1
    report = {"cells": int(v.sum()), "exact_pass": st}
    assert np.array_equal(raw["valid"] & 1, want["valid"]), "valid flags differ"
    for f in DEMUX_INT_FIELDS:
        bad = np.flatnonzero(got[f] != want[f])
        assert bad.size == 0, (f"{f} differs in {bad.size} cells: {bad[:5].tolist()}: got {got[f][bad[:5]].tolist()}, "
                               f"reference {want[f][bad[:5]].tolist()}")
    g, w = got[v], want[v]
    worst = 0.0
    for f in DEMUX_LL_FIELDS + DEMUX_PP_FIELDS:
        ok = _close(g[f], w[f], tol)
        with np.errstate(invalid="ignore"):
            d = np.abs(g[f] - w[f])
        d = d[np.isfinite(d)]
        if d.size:
            worst = max(worst, float(d.max()))
        assert ok.all(), f"{f}: {int((~ok).sum())} cells beyond {tol}; worst {d.max() if d.size else 'nan'}"
    report["max_abs_ll_diff"] = worst
    differs = np.zeros(got.shape, dtype=bool)
    for f in DEMUX_INT_FIELDS:
        differs |= (raw[f] & 1 if f == "valid" else raw[f]) != want[f]
    report["raw_records_differing"] = int(differs.sum())   # the pass's work, seen from outside
    assert report["raw_records_differing"] <= st["cells"]
    # kept for the readers of bench lines of earlier rounds: no relaxation exists any more
    report["excuses_used"] = {"singlet_tie": 0, "doublet_tie": 0, "mirrored_pair_order": 0}
    return report


def needed_ll_mask(V, alphas):
    """[V][V][A] mask of the llksAB slots the reference ever reads: (j,0,0) and (j,k!=j,n>=1)"""
    A = len(alphas)
    m = np.zeros((V, V, A), dtype=bool)
    m[:, 0, 0] = True
    off = ~np.eye(V, dtype=bool)
    for n in range(1, A):
        m[:, :, n] = off
    return m


def compare_full_ll(got, want, V, alphas, tol=LL_TOL):
    m = needed_ll_mask(V, alphas)
    d = np.abs(got[:, m] - want[:, m])
    ok = _close(got[:, m], want[:, m], tol)
    assert ok.all(), f"full LL tensor: {int((~ok).sum())} slots beyond {tol}"
    d = d[np.isfinite(d)]
    return float(d.max()) if d.size else 0.0


FMX_LL_FIELDS = ("bestLLK", "nextLLK", "sngBestLLK", "sngNextLLK", "dblBestLLK", "dblNextLLK", "bestPP", "sngPP",
                 "sngOnlyPP", "sumLLK")
FMX_INT_FIELDS = ("type", "clust", "jBest", "kBest", "jNext", "kNext", "sBest", "sNext", "dBest1", "dBest2", "dNext1",
                  "dNext2")


def fmx_near_tie_mask(rec):
    """cells of [C] muxgl_fmx_cell records whose call is within rounding reach of the kernels' numbers: the rule of
    fmx_call_kernel (popscle_amd/csrc/fmx_kernels.hip), restated on the record's own fields"""
    def some(v):
        return v > -1e299

    mag = np.ones(rec.shape)
    for f in ("sngBestLLK", "sngNextLLK", "dblBestLLK", "dblNextLLK"):
        mag = np.where(some(rec[f]), np.maximum(mag, np.abs(rec[f])), mag)
    eps = 1e-9 * mag

    def near(a, b):
        with np.errstate(invalid="ignore", over="ignore"):
            return some(a) & some(b) & (np.abs(a - b) <= eps)

    sB, sN, s3 = rec["sngBestLLK"], rec["sngNextLLK"], rec["sngThirdLLK"]
    dB, dN, d3 = rec["dblBestLLK"], rec["dblNextLLK"], rec["dblThirdLLK"]
    return (near(sB, sN) | near(sN, s3) | near(dB, dN) | near(dN, d3) | near(dB, sB + 2) | near(dN, sB + 2) |
            near(sB, sN + 2) | near(dB, sN + 2))


def compare_fmx(got, want, tol=LL_TOL, resolved=True):
    """[C] freemuxlet records of the library against the reference's / oracle's.  Every log-likelihood and posterior
    within `tol`; every integer field EQUAL -- no tie window.

    resolved=True (muxgl_fmx_iterate on one device): the library settled its near-tie calls itself in the reference's
    arithmetic (fmx_exact.hip), so there is nothing to excuse.  resolved=False (the sharded phases and device groups,
    which only COUNT such cells): a cell the record itself shows to be within rounding reach (fmx_near_tie_mask) may
    differ; the report says how many did.
    """
    assert got.shape == want.shape
    worst = 0.0
    for f in FMX_LL_FIELDS:
        ok = _close(got[f], want[f], tol)
        with np.errstate(invalid="ignore"):
            d = np.abs(got[f] - want[f])
        d = d[np.isfinite(d)]
        if d.size:
            worst = max(worst, float(d.max()))
        assert ok.all(), f"{f}: {int((~ok).sum())} cells beyond {tol}"
    near = fmx_near_tie_mask(got) if "sngThirdLLK" in (got.dtype.names or ()) else np.zeros(got.shape, dtype=bool)
    differs = np.zeros(got.shape, dtype=bool)
    for f in FMX_INT_FIELDS:
        differs |= got[f] != want[f]
    bad = differs if resolved else differs & ~near
    if bad.any():
        i = int(np.flatnonzero(bad)[0])
        assert False, (f"{int(bad.sum())} cells differ in an integer field; first: cell {i}: "
                       f"got {[(f, int(got[f][i])) for f in FMX_INT_FIELDS]}, reference {[(f, int(want[f][i])) for f in FMX_INT_FIELDS]}")
    return {"cells": int(got.size), "max_abs_ll_diff": worst, "near_tie_cells": int(near.sum()),
            "unresolved_near_ties_differing": int(differs.sum()) if not resolved else 0,
            "cells_needing_an_excuse": 0 if resolved else int(differs.sum())}


SUM_FIELDS = ("sumLLK", "sngLLK", "bestPP", "sngPP", "sngOnlyPP")


def same_records(a, b, sum_rtol=1e-12):
    """Two sets of demuxlet records made by different call paths of the library (in the sweep's workgroup / by the call
    kernel from the result slab): every integer field and every hypothesis log-likelihood identical bit for bit; the
    evidence sums and the posteriors derived from them are added up in another association and may differ in the last
    bits (relative to 1, or to the value where that is larger)."""
    assert a.shape == b.shape
    for name in a.dtype.names:
        x, y = a[name], b[name]
        if name in SUM_FIELDS:
            ok = (x == y) | (np.abs(x - y) <= sum_rtol * np.maximum(1.0, np.maximum(np.abs(x), np.abs(y))))
            assert ok.all(), (name, x[~ok][:3], y[~ok][:3])
        else:
            assert np.array_equal(x, y), (name, np.flatnonzero(x != y)[:5])
    return True
