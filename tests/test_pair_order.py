"""CPU tests of the host pass that orders mirrored alpha = 0.5 doublet pairs as the reference's scan does
(popscle_amd/host/pair_order.hpp, exported as muxgl_demux_reference_pair_order; cmd_cram_demuxlet.cpp:738-746,883-906).

The pass is host code of the product and needs no device: records are put into the DEVICE'S convention here (a symmetric
pair named (lo, hi), its mirror as the runner-up) and the pass must turn them into the reference's records exactly --
the reference being its own compiled loop where oracle/_ref/libscdrop_ref.so exists, the oracle (bit-identical to it,
tests/test_oracle_ref.py) otherwise.
"""
import numpy as np
import pytest

import oracle_binding as ob
import ref_binding as rb
from popscle_amd import muxgl, synth

INT_FIELDS = ("dBest1", "dBest2", "dBestA", "dNext1", "dNext2", "dNextA", "jBest", "kBest", "aBest", "jNext", "kNext",
              "aNext", "type", "next_type", "sBest", "sNext")


def reference_records(p, alphas):
    if rb.available():
        return rb.RefScl.from_packed(p).demux(alphas, doublet_prior=0.5)[0]
    return ob.demux(p, alphas, doublet_prior=0.5)


def device_convention(want, alphas):
    """what muxgl_demux_run reports for the same cells: a pair at alpha 0.5 is computed once and mirrored, so the scan
    meets (lo, hi) first and its mirror becomes the runner-up whenever the pair is the best one"""
    got = np.zeros(want.shape, dtype=muxgl.DEMUX_CELL)
    for n in want.dtype.names:
        got[n] = want[n]
    al = np.asarray(alphas)

    def sym(a, b, n):
        return (a >= 0) & (b >= 0) & (n >= 1) & (al[np.clip(n, 0, al.size - 1)] == 0.5)

    sb = sym(got["dBest1"], got["dBest2"], got["dBestA"])
    lo, hi = np.minimum(got["dBest1"], got["dBest2"]), np.maximum(got["dBest1"], got["dBest2"])
    mirror = sb & (got["dBestA"] == got["dNextA"]) & (np.minimum(got["dNext1"], got["dNext2"]) == lo) & \
        (np.maximum(got["dNext1"], got["dNext2"]) == hi)
    got["dBest1"][sb], got["dBest2"][sb] = lo[sb], hi[sb]
    got["dNext1"][mirror], got["dNext2"][mirror] = hi[mirror], lo[mirror]
    sn = sym(got["dNext1"], got["dNext2"], got["dNextA"]) & ~mirror
    lo, hi = np.minimum(got["dNext1"], got["dNext2"]), np.maximum(got["dNext1"], got["dNext2"])
    got["dNext1"][sn], got["dNext2"][sn] = lo[sn], hi[sn]
    dbl, ndbl = got["type"] == 1, got["next_type"] == 1
    got["jBest"][dbl], got["kBest"][dbl] = got["dBest1"][dbl], got["dBest2"][dbl]
    m = dbl & ndbl
    got["jNext"][m], got["kNext"][m] = got["dNext1"][m], got["dNext2"][m]
    m = ~dbl & ndbl
    got["jNext"][m], got["kNext"][m] = got["dBest1"][m], got["dBest2"][m]
    return got


@pytest.mark.parametrize("C,S,V,alphas,kw", [
    (300, 3000, 16, (0.0, 0.5), dict(mean_entries=300, doublet_frac=0.3)),
    (200, 2000, 4, (0.0, 0.5), dict(mean_entries=150, doublet_frac=0.3, missing_gp_frac=0.05)),
    (120, 3000, 8, (0.0, 0.1, 0.2, 0.3, 0.4, 0.5), dict(mean_entries=200, doublet_frac=0.5)),
    (60, 2000, 5, (0.0, 0.5, 0.25), dict(mean_entries=200, doublet_frac=0.5, reads_lambda=1.5, other=0.03)),
    (40, 4000, 64, (0.0, 0.3, 0.5), dict(mean_entries=300, doublet_frac=0.3)),
    (50, 500, 3, (0.0, 0.3), dict(mean_entries=80)),     # no symmetric alpha: nothing to do
])
def test_pass_restores_the_references_order(C, S, V, alphas, kw):
    p = synth.make_pileup(C, S, V, seed=1000 + V, **kw)
    want = reference_records(p, alphas)
    got = device_convention(want, alphas)
    before = sum(int((got[f] != want[f]).sum()) for f in INT_FIELDS)
    looked, turned, ties = muxgl.demux_reference_pair_order(p, alphas, got, nthreads=3)
    for f in INT_FIELDS:
        assert np.array_equal(got[f], want[f]), f
    if 0.5 in alphas:
        assert looked == int(want["valid"].sum()) or len(alphas) > 2
        assert before > 0 and turned > 0, "the case does not exercise a reordering"
    else:
        assert looked == 0 and before == 0


def test_pass_is_idempotent_and_thread_count_independent():
    p = synth.make_pileup(400, 3000, 6, seed=77, mean_entries=120, doublet_frac=0.4)
    alphas = (0.0, 0.5)
    want = reference_records(p, alphas)
    a = device_convention(want, alphas)
    b = a.copy()
    muxgl.demux_reference_pair_order(p, alphas, a, nthreads=1)
    muxgl.demux_reference_pair_order(p, alphas, b, nthreads=7)
    assert a.tobytes() == b.tobytes()
    muxgl.demux_reference_pair_order(p, alphas, b, nthreads=2)   # already in the reference's order: unchanged
    assert a.tobytes() == b.tobytes()
