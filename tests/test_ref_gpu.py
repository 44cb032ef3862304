"""GPU parity tests of libmuxgl (HIP, through the C-ABI) DIRECTLY against the reference's own code -- the hot loops of
cmd_cram_demuxlet.cpp / cmd_cram_freemux2.cpp compiled from /root/reference into oracle/_ref/libscdrop_ref.so
(oracle/ref_hot.cpp.in; the built .so travels to the GPU box, the reference's sources do not).  The other GPU tests
compare with the oracle, which tests/test_oracle_ref.py holds to this library bit for bit; these close the chain
without the oracle in between.  Skipped where the library was not built.

Bar: calls exact (parity.compare_*), log-likelihoods within 1e-5 absolute (observed ~1e-11).
"""
import numpy as np
import pytest

import parity
import ref_binding as rb
from popscle_amd import muxgl, synth

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not rb.available(), reason="oracle/_ref/libscdrop_ref.so not built")]

GRID6 = (0.0, 0.1, 0.2, 0.3, 0.4, 0.5)


@pytest.fixture(scope="module")
def eng():
    e = muxgl.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("V,alphas,C,S,ment", [
    (16, (0.0, 0.5), 400, 6000, 500),      # the headline shape's kernel (oct)
    (4, (0.0, 0.5), 300, 2000, 250),
    (16, GRID6, 100, 5000, 500),           # row kernel
    (24, (0.0, 0.5), 60, 4000, 500),       # sixteen lanes per entry
    (64, GRID6, 24, 8000, 900),            # configs[2]'s kernel (ring)
    (70, (0.0, 0.3, 0.5), 10, 5000, 600),  # 64 x 64 blocks
])
def test_demuxlet_vs_reference_library(eng, V, alphas, C, S, ment):
    p = synth.make_pileup(C, S, V, seed=900 + V, mean_entries=ment, reads_lambda=0.6, other=0.02, doublet_frac=0.25,
                          missing_gp_frac=0.02)
    want, _, want_ll = rb.RefScl.from_packed(p).demux(alphas, doublet_prior=0.5, full_ll=True)
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    eng.demux_set_gp(p.gp, p.has_gp)
    got, full = eng.demux_run(alphas, 0.5, want_full_ll=True)
    rep = parity.compare_demux(got, want, alphas, p)   # the reference library itself: every integer field equal
    worst = parity.compare_full_ll(full, want_ll, V, alphas)
    assert rep["max_abs_ll_diff"] < 1e-7 and worst < 1e-7


@pytest.mark.parametrize("K,C,S,ment", [(16, 500, 4000, 300), (4, 300, 600, 70), (64, 120, 6000, 500)])
def test_freemuxlet_vs_reference_library(eng, K, C, S, ment):
    """the reference's own run: greedy start, EM until its early stop; the device replays it phase by phase"""
    p = synth.make_pileup(C, S, K, seed=950 + K, mean_entries=ment, min_entries=5, reads_lambda=0.8, other=0.02,
                          doublet_frac=0.25, with_gp=False)
    ref = rb.RefScl.from_packed(p).freemux2(K, full_ll=True, cluster_pileups=True)
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    llk0, llk2, ns, nr = eng.fmx_prepare(p.af)
    assert np.max(np.abs(llk0 - ref["llk0"])) < 1e-7 and np.max(np.abs(llk2 - ref["llk2"])) < 1e-7
    assert np.array_equal(ns, ref["nsnps"]) and np.array_equal(nr, ref["nreads"])
    clust = eng.fmx_greedy_init(K, llk2 - llk0)
    assert np.array_equal(clust, ref["clust0"]), "greedy initial clusters differ from the reference's"
    eng.fmx_set_clusters(K, clust)
    for it in range(ref["n_iter"]):
        cells, st, full = eng.fmx_iterate(0.5, 0.1, want_full_ll=True)
        assert tuple(st) == tuple(ref["counters"][it]), (it, st, ref["counters"][it])
        assert np.max(np.abs(full - ref["full_ll"][it])) < 1e-7
        parity.compare_fmx(cells, ref["cells"][it])   # the reference library itself: every integer field equal
        g, c = eng.fmx_cluster_pileup()
        w = ref["cplp"][it]
        assert np.array_equal(c, np.stack([w["nreads"], w["nref"], w["nalt"]], axis=-1))
        assert np.allclose(g, w["gls"], rtol=1e-11, atol=1e-300)
    assert st[2] == 0 or ref["n_iter"] == 10
