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
        differs |= raw[f] != want[f]
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


def compare_fmx(got, want, tol=LL_TOL, tie_eps=1e-7, want_full=None):
    """want_full: the oracle's packed LL triangle [C][K(K+1)/2] (optional).  With it, a different runner-up doublet is
    accepted where the pair the GPU names is tied with the oracle's runner-up IN THE ORACLE'S OWN NUMBERS (clusters
    without cells have identical posteriors, so their pairs tie exactly in the reference, which then keeps the first
    in scan order; the kernels evaluate the two tied pairs in different associations)."""
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
    tie = (np.abs(want["sngBestLLK"] - want["sngNextLLK"]) < tie_eps) | \
          (np.abs(want["dblBestLLK"] - want["dblNextLLK"]) < tie_eps)
    next_tie = np.zeros(got.shape, dtype=bool)
    if want_full is not None:
        hi = np.maximum(got["dNext1"], got["dNext2"]).astype(np.int64)
        lo = np.minimum(got["dNext1"], got["dNext2"]).astype(np.int64)
        ok_idx = (lo >= 0) & (hi * (hi + 1) // 2 + lo < want_full.shape[1])
        named = want_full[np.arange(got.size), np.where(ok_idx, hi * (hi + 1) // 2 + lo, 0)]
        next_tie = ok_idx & (np.abs(named - want["dblNextLLK"]) < tie_eps)
    differs = np.zeros(got.shape, dtype=bool)
    for f in FMX_INT_FIELDS:
        excused = tie | (next_tie if f in ("dNext1", "dNext2") else False)
        bad = (got[f] != want[f]) & ~excused
        assert not bad.any(), f"{f} differs in {int(bad.sum())} cells"
        differs |= got[f] != want[f]
    # "ties": cells whose oracle numbers tie (an excuse was AVAILABLE); "cells_needing_an_excuse": where one was USED
    return {"cells": int(got.size), "max_abs_ll_diff": worst, "ties": int(tie.sum()),
            "cells_needing_an_excuse": int(differs.sum())}


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
