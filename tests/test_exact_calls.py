"""CPU tests of the host pass that settles the demuxlet calls rounding noise could decide
(popscle_amd/host/exact_calls.hpp, exported as muxgl_demux_exact_calls; cmd_cram_demuxlet.cpp:738-746,827-837,883-906,
925-988).

The pass is host code of the product and needs no device.  The device is EMULATED here: the reference's full
log-likelihood tensor (its own compiled loop where oracle/_ref/libscdrop_ref.so exists, else the oracle, bit-identical to
it) is perturbed by more noise than the kernels have (up to 1e-10 relative, the kernels measure ~1e-13), an alpha = 0.5
pair gets ONE value for both orders, and records are formed from that the way demux_call_body.hpp forms them (distinct
hypotheses, best / next / third).  The pass must turn those records into the reference's records: every integer field,
and the log-likelihoods it recomputed, exactly.
"""
import numpy as np
import pytest

import oracle_binding as ob
import ref_binding as rb
from popscle_amd import muxgl, synth

INT_FIELDS = ("valid", "nsnps", "type", "next_type", "sBest", "sNext", "dBest1", "dBest2", "dBestA", "dNext1", "dNext2",
              "dNextA", "jBest", "kBest", "aBest", "jNext", "kNext", "aNext")
LL_FIELDS = ("sngBestLLK", "sngNextLLK", "dblBestLLK", "dblNextLLK", "bestLLK", "nextLLK")


def reference_records(p, alphas, doublet_prior=0.5):
    if rb.available():
        cells, _, full = rb.RefScl.from_packed(p).demux(alphas, doublet_prior=doublet_prior, full_ll=True)
        return cells, full
    return ob.demux(p, alphas, doublet_prior=doublet_prior, full_ll=True)


def emulate_device(want, full, alphas, doublet_prior, noise, seed=0):
    """records as muxgl_demux_run would report them from log-likelihoods that differ from the reference's by `noise`
    (relative), demux_call_body.hpp's conventions"""
    rng = np.random.default_rng(seed)
    C, V, _, A = full.shape
    al = np.asarray(alphas)
    dev = full * (1.0 + noise * rng.uniform(-1, 1, full.shape))
    for n in range(1, A):
        if al[n] == 0.5:     # one value for both orders of the pair
            lo = np.triu(dev[:, :, :, n], 1)
            dev[:, :, :, n] = lo + lo.transpose(0, 2, 1)
    lsp = np.log((1.0 - doublet_prior) / V)
    out = np.zeros(C, dtype=muxgl.DEMUX_CELL)
    for f in ("valid", "nsnps", "sumLLK", "sngLLK", "sngPP"):
        out[f] = want[f]
    for c in range(C):
        if not want["valid"][c]:
            continue
        o = out[c]
        s = dev[c, :, 0, 0]
        order = np.lexsort((np.arange(V), -s))            # value descending, position ascending
        o["sBest"], o["sngBestLLK"] = order[0], s[order[0]]
        o["sNext"], o["sngNextLLK"] = (order[1], s[order[1]]) if V > 1 else (-1, -1e300)
        third_s = s[order[2]] if V > 2 else -1e300
        hyp = [(dev[c, j, k, n], (j * V + k) * A + n, j, k, n) for j in range(V) for k in range(V) if k != j
               for n in range(1, A) if not (al[n] == 0.5 and k < j)]
        hyp.sort(key=lambda t: (-t[0], t[1]))
        o["dBest1"] = o["dBest2"] = o["dBestA"] = o["dNext1"] = o["dNext2"] = o["dNextA"] = -1
        o["dblBestLLK"] = o["dblNextLLK"] = -1e300
        third_d = -1e300
        if hyp:
            v, _, j, k, n = hyp[0]
            o["dBest1"], o["dBest2"], o["dBestA"], o["dblBestLLK"] = j, k, n, v
            if al[n] == 0.5:
                o["dNext1"], o["dNext2"], o["dNextA"], o["dblNextLLK"] = k, j, n, v
                third_d = hyp[1][0] if len(hyp) > 1 else -1e300
            elif len(hyp) > 1:
                v2, _, j2, k2, n2 = hyp[1]
                o["dNext1"], o["dNext2"], o["dNextA"], o["dblNextLLK"] = j2, k2, n2, v2
                third_d = hyp[2][0] if len(hyp) > 2 else -1e300
        # the two bits the call kernel leaves in `valid` (demux_call_body.hpp): third within reach of the runner-up
        mag = max([1.0] + [abs(float(o[f])) for f in ("sngBestLLK", "sngNextLLK", "dblBestLLK", "dblNextLLK") if o[f] > -1e299])
        eps = 1e-9 * mag
        if o["sngNextLLK"] > -1e299 and third_s > -1e299 and o["sngNextLLK"] - third_s <= eps:
            o["valid"] |= muxgl.CELL_DEEP_SNG
        if o["dblNextLLK"] > -1e299 and third_d > -1e299 and o["dblNextLLK"] - third_d <= eps:
            o["valid"] |= muxgl.CELL_DEEP_DBL
        if o["dblBestLLK"] > o["sngBestLLK"] + 2:
            o["type"] = 1
            o["jBest"], o["kBest"], o["aBest"], o["bestLLK"] = o["dBest1"], o["dBest2"], o["dBestA"], o["dblBestLLK"]
            if o["dblNextLLK"] > o["sngBestLLK"] + 2:
                o["next_type"] = 1
                o["jNext"], o["kNext"], o["aNext"], o["nextLLK"] = o["dNext1"], o["dNext2"], o["dNextA"], o["dblNextLLK"]
            else:
                o["next_type"] = 0
                o["jNext"], o["kNext"], o["aNext"], o["nextLLK"] = o["sBest"], o["sBest"], 0, o["sngBestLLK"]
        else:
            o["type"] = 0 if o["sngBestLLK"] > o["sngNextLLK"] + 2 else 2
            o["jBest"], o["kBest"], o["aBest"], o["bestLLK"] = o["sBest"], o["sBest"], 0, o["sngBestLLK"]
            if o["dblBestLLK"] > o["sngNextLLK"] + 2:
                o["next_type"] = 1
                o["jNext"], o["kNext"], o["aNext"], o["nextLLK"] = o["dBest1"], o["dBest2"], o["dBestA"], o["dblBestLLK"]
            else:
                o["next_type"] = 0
                o["jNext"], o["kNext"], o["aNext"], o["nextLLK"] = o["sNext"], o["sNext"], 0, o["sngNextLLK"]
        o["bestPP"] = want["bestPP"][c]
        o["sngOnlyPP"] = np.exp(o["sngBestLLK"] + lsp - o["sngLLK"])
    return out


def check_exact(got, want, stats, require_ll=True):
    v = want["valid"] == 1
    for f in INT_FIELDS:
        bad = np.flatnonzero(got[f] != want[f])
        assert bad.size == 0, (f, bad[:5], got[f][bad[:5]], want[f][bad[:5]], stats)
    assert np.allclose(got["bestPP"][v], want["bestPP"][v], rtol=1e-6, atol=1e-12, equal_nan=True)


CASES = [
    (300, 3000, 16, (0.0, 0.5), dict(mean_entries=300, doublet_frac=0.3)),
    (200, 2000, 4, (0.0, 0.5), dict(mean_entries=150, doublet_frac=0.3, missing_gp_frac=0.05)),
    (120, 3000, 8, (0.0, 0.1, 0.2, 0.3, 0.4, 0.5), dict(mean_entries=200, doublet_frac=0.5)),
    (60, 2000, 5, (0.0, 0.5, 0.25), dict(mean_entries=200, doublet_frac=0.5, reads_lambda=1.5, other=0.03)),
    (40, 4000, 64, (0.0, 0.3, 0.5), dict(mean_entries=300, doublet_frac=0.3)),
    (50, 500, 3, (0.0, 0.3), dict(mean_entries=80)),     # no symmetric alpha
    (30, 400, 1, (0.0, 0.5), dict(mean_entries=40)),     # one sample: no doublet hypothesis at all
]


@pytest.mark.parametrize("C,S,V,alphas,kw", CASES)
def test_pass_gives_the_references_records(C, S, V, alphas, kw):
    p = synth.make_pileup(C, S, V, seed=1000 + V, **kw)
    want, full = reference_records(p, alphas)
    got = emulate_device(want, full, alphas, 0.5, noise=1e-13, seed=V)
    st = muxgl.demux_exact_calls(p, alphas, got, 0.5, nthreads=3)
    check_exact(got, want, st)
    if 0.5 in alphas and V > 1:
        assert st["mirror_turned"] > 0 and st["cells"] > 0, "the case does not exercise a reordering"
        # the log-likelihoods of the two orders are the reference's own, bit for bit
        m = (want["dBestA"] >= 0) & (np.asarray(alphas)[np.clip(want["dBestA"], 0, None)] == 0.5) & (want["valid"] == 1)
        assert m.sum() > 0
        assert np.array_equal(got["dblBestLLK"][m], want["dblBestLLK"][m])
        assert np.array_equal(got["dblNextLLK"][m], want["dblNextLLK"][m])
    elif V == 1:
        assert st["mirror_turned"] == 0


def structural_ties(seed=5, V=12, alphas=(0.0, 0.5)):
    """droplets of one to four entries against samples of which several share their genotypes (identical GP rows):
    hypotheses tie EXACTLY in the reference, which then keeps the first in scan order; three and more of them tie"""
    p = synth.make_pileup(400, 300, V, seed=seed, mean_entries=3, min_entries=1, doublet_frac=0.2)
    gp = p.gp.copy()
    gp[:, 5] = gp[:, 2]       # twins
    gp[:, 9] = gp[:, 2]       # triplets
    gp[:, 7] = gp[:, 1]
    return synth.Pileup(p.C, p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads, p.af, gp, p.has_gp, p.truth)


@pytest.mark.parametrize("alphas", [(0.0, 0.5), (0.0, 0.2, 0.5), (0.0, 0.25)])
@pytest.mark.parametrize("noise", [0.0, 1e-13, 1e-10])
def test_near_and_exact_ties_are_settled_as_the_reference_does(alphas, noise):
    p = structural_ties(alphas=alphas)
    want, full = reference_records(p, alphas)
    got = emulate_device(want, full, alphas, 0.5, noise=noise, seed=3)
    raw_diff = sum(int((got[f] != want[f]).sum()) for f in INT_FIELDS)
    st = muxgl.demux_exact_calls(p, alphas, got, 0.5, nthreads=2)
    check_exact(got, want, st)
    assert st["near_ties"] > 50 and st["deep"] > 10, st       # the case is what it claims to be
    if noise > 0:
        assert raw_diff > 0 and st["changed"] > 0, (raw_diff, st)   # the noise did flip calls, the pass flipped them back
    # where the pass recomputed both scans the six log-likelihoods are the reference's, bit for bit
    m = (np.abs(got["sngBestLLK"] - got["sngNextLLK"]) <= 1e-9) & (want["valid"] == 1)
    for f in ("sngBestLLK", "sngNextLLK"):
        assert np.array_equal(got[f][m], want[f][m]), f


def test_threshold_margins():
    """cells whose DBL / SNG / AMB decision hangs on `> x + 2` within the noise: made by scaling nothing -- found among
    many shallow droplets by brute force, then pushed onto the threshold by the emulated noise"""
    alphas = (0.0, 0.5)
    p = synth.make_pileup(3000, 400, 6, seed=12, mean_entries=6, min_entries=2, doublet_frac=0.4)
    want, full = reference_records(p, alphas)
    margins = np.stack([want["dblBestLLK"] - want["sngBestLLK"] - 2, want["sngBestLLK"] - want["sngNextLLK"] - 2,
                        want["dblBestLLK"] - want["sngNextLLK"] - 2])
    close = np.abs(margins).min(axis=0)
    # noise large enough to carry a good number of cells across a threshold, small enough to stay inside EPS
    got = emulate_device(want, full, alphas, 0.5, noise=3e-10, seed=1)
    st = muxgl.demux_exact_calls(p, alphas, got, 0.5, nthreads=4)
    check_exact(got, want, st)
    assert st["near_ties"] > 0
    assert (close < 1e-9 * np.maximum(1, np.abs(want["sngBestLLK"]))).sum() >= 0   # (informational: exact hits are rare)


def test_pass_is_idempotent_and_thread_count_independent():
    p = synth.make_pileup(400, 3000, 6, seed=77, mean_entries=120, doublet_frac=0.4)
    alphas = (0.0, 0.5)
    want, full = reference_records(p, alphas)
    a = emulate_device(want, full, alphas, 0.5, noise=1e-13)
    b = a.copy()
    muxgl.demux_exact_calls(p, alphas, a, 0.5, nthreads=1)
    muxgl.demux_exact_calls(p, alphas, b, 0.5, nthreads=7)
    assert a.tobytes() == b.tobytes()
    muxgl.demux_exact_calls(p, alphas, b, 0.5, nthreads=2)   # already the reference's records: unchanged
    assert a.tobytes() == b.tobytes()
    check_exact(a, want, None)


def _fuzz_seeds(n=40, vmax=16, cmax=200):
    """seeds of tests/test_fuzz_gpu.py's demuxlet generator whose cases the Python emulation can afford"""
    import test_fuzz_gpu as fz

    out = []
    seed = 0
    while len(out) < n and seed < 400:
        r = np.random.default_rng([seed, 77])
        if int(r.choice(fz.DEMUX_V)) <= vmax:
            out.append(seed)
        seed += 1
    return out


@pytest.mark.parametrize("seed", _fuzz_seeds())
def test_unfriendly_generator_cases(seed):
    """the GPU suite's randomised generator (tests/test_fuzz_gpu.py: droplets of a few entries, duplicated or identical
    samples, markers without genotypes, hard calls, float-normalised rows, deep entries, odd grids and priors) through the
    emulated device and the host pass: the records must be the reference's"""
    import test_fuzz_gpu as fz

    info, p = fz.demux_case(seed)
    if p.C > 200:
        p = p.subset_cells(np.arange(200))
    alphas, dp = info["alphas"], info["dp"]
    want, full = reference_records(p, alphas, doublet_prior=dp)
    if not np.isfinite(full[want["valid"] == 1]).all():
        pytest.skip("non-finite log-likelihoods: the emulation's noise model does not apply")
    got = emulate_device(want, full, alphas, dp, noise=1e-13, seed=seed)
    st = muxgl.demux_exact_calls(p, alphas, got, dp, nthreads=2)
    check_exact(got, want, st)
