"""Regenerates the golden vectors under tests/golden/.

demux_*.npz and fmx_k4.npz hold OUTPUTS OF THE REFERENCE'S OWN CODE run in the build container: the hot loops of
cmd_cram_demuxlet.cpp / cmd_cram_freemux2.cpp and sc_drop_seq.cpp's per-entry arithmetic, compiled from /root/reference
into oracle/_ref/libscdrop_ref.so (oracle/Makefile, oracle/ref_hot.cpp.in: verbatim line ranges, no stand-ins), fed the
packed inputs in packed order (tests/ref_binding.py RefScl.from_packed).  The generator also requires the CPU oracle
(oracle/muxgl_oracle.c) to reproduce every array bit for bit before it writes a file, and records `source` in it.
fmxold_k4.npz: its initial clustering part (cmd_cram_freemuxlet.cpp:176-346, out of scope per SURVEY section 2 row 2b)
remains a regression vector of the oracle; its EM part (em_* arrays) is the reference's own cmd_cram_freemuxlet.cpp:456-653
run from those clusters (scref_freemuxlet_old).  Inputs come from popscle_amd.synth with fixed seeds; each .npz holds the packed inputs and the
expected outputs, so the GPU box can check the HIP path against the reference's numbers without /root/reference.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_binding as ob  # noqa: E402
import ref_binding as rb  # noqa: E402
from popscle_amd import synth  # noqa: E402

SOURCE = ("reference: oracle/_ref/libscdrop_ref.so (cmd_cram_demuxlet.cpp:428-440,590-622,634-991; "
          "cmd_cram_freemux2.cpp:108-109,114-159,184-189,192-262,277-288,350-370,373-605; sc_drop_seq.cpp:1-92,386-578); "
          "oracle/muxgl_oracle.c reproduces every array bit for bit")


def demux_case(name, C, S, V, alphas, seed, **kw):
    p = synth.make_pileup(C, S, V, seed=seed, **kw)
    cells, _, full = rb.RefScl.from_packed(p).demux(alphas, doublet_prior=0.5, full_ll=True)   # the reference
    ocells, ofull = ob.demux(p, alphas=alphas, doublet_prior=0.5, full_ll=True)
    assert ocells.tobytes() == cells.tobytes() and np.array_equal(ofull, full), "oracle != reference"
    np.savez_compressed(os.path.join(HERE, name + ".npz"), C=p.C, S=p.S, cell_ptr=p.cell_ptr, entry_snp=p.entry_snp,
                        entry_rptr=p.entry_rptr, reads=p.reads, af=p.af, gp=p.gp, has_gp=p.has_gp,
                        alphas=np.array(alphas), doublet_prior=0.5, cells=cells, full_ll=full, source=SOURCE)
    print(name, "cells", p.C, "entries", p.nnz, "types", np.bincount(cells["type"], minlength=3))


def fmx_case(name, C, S, K, seed, n_iter, **kw):
    p = synth.make_pileup(C, S, K, seed=seed, with_gp=False, **kw)
    e = ob.fmx_entry_pileup(p)
    llk0, llk2, ns, nr = ob.fmx_cell_scores(p, e)
    clust0 = ob.fmx_greedy_init(p, e, K, llk2 - llk0, ob.fmx_sort(llk2 - llk0))
    cplp = ob.fmx_build_cluster_pileup(p, e, K, clust0)
    cells = ob.fmx_init_cells(clust0)
    stats = []
    ll1 = None
    for it in range(n_iter):
        r = ob.fmx_iterate(p, e, K, cplp, cells, full_ll=(it == 0))
        if it == 0:
            ll1 = r[3]
        stats.append(r[:3])
    # the reference's own run of the same job: ten iterations at most with its early stop; the first n_iter of them
    # must be the oracle's, array for array
    r = rb.RefScl.from_packed(p)
    ref = r.freemux2(K, full_ll=True, cluster_pileups=True)
    m = min(n_iter, ref["n_iter"])   # past its early stop (nchanged == 0) an iteration repeats the previous one
    assert r.entry_pileup(p.nnz).tobytes() == e.tobytes()
    assert np.array_equal(ref["llk0"], llk0) and np.array_equal(ref["llk2"], llk2)
    assert np.array_equal(ref["nsnps"], ns) and np.array_equal(ref["nreads"], nr)
    assert np.array_equal(ref["clust0"], clust0)
    assert np.array_equal(ref["counters"][:m], np.array(stats)[:m])
    assert all(st == stats[m - 1] for st in stats[m:]) and stats[m - 1][2] == 0 or m == n_iter
    assert ref["cells"][m - 1].tobytes() == cells.tobytes()
    assert ref["cplp"][m - 1].tobytes() == cplp.tobytes()
    assert np.array_equal(ref["full_ll"][0], ll1)
    cnt = np.stack([cplp["nreads"], cplp["nref"], cplp["nalt"]], axis=-1)
    ecnt = np.stack([e["nreads"], e["nref"], e["nalt"]], axis=-1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), C=p.C, S=p.S, K=K, cell_ptr=p.cell_ptr,
                        entry_snp=p.entry_snp, entry_rptr=p.entry_rptr, reads=p.reads, af=p.af, entry_gls=e["gls"],
                        entry_cnt=ecnt, llk0=llk0, llk2=llk2, nsnps=ns, nreads=nr, clust0=clust0, n_iter=n_iter,
                        stats=np.array(stats), cells=cells, cluster_gls=cplp["gls"], cluster_cnt=cnt, full_ll_iter1=ll1,
                        source=SOURCE)
    print(name, "cells", p.C, "entries", p.nnz, "stats", stats)


def fmxold_case(name, C, S, K, seed, **kw):
    """freemuxlet-old's initial clustering (cmd_cram_freemuxlet.cpp:176-343): pair records, first pass, three refinement
    passes; jitters and orders are part of the fixture (the RNG belongs to the caller)"""
    p = synth.make_pileup(C, S, K, seed=seed, with_gp=False, cap_bq=60, min_bq=2, **kw)
    e = ob.fmx_entry_pileup(p)
    llk0, llk2, _, _ = ob.fmx_cell_scores(p, e)
    order = ob.fmx_sort(llk2 - llk0)
    dd = ob.fmxold_pair_dist(p, e)
    rng = np.random.default_rng(seed)
    thres, frac = 2.0, 0.8
    nvis = sum(1 for i in range(C) if not i > C * frac)
    jit0 = rng.integers(0, 2**31, (nvis, K)) / 2.0**31 / 1000.0
    clust0, cc0 = ob.fmxold_vote_init(C, K, dd, order, jit0, thres, frac)
    orands, jits, clusts, changed = [], [], [], []
    cl = clust0
    for it in range(3):
        orand = rng.permutation(C).astype(np.int32)
        jit = rng.integers(0, 2**31, (C, K)) / 2.0**31 / 1000.0
        cl, ch, _ = ob.fmxold_vote_refine(C, K, dd, orand, jit, cl, thres, keep_init_missing=(it == 0))
        orands.append(orand), jits.append(jit), clusts.append(cl), changed.append(ch)
    # the EM loop from those clusters: THE REFERENCE'S OWN cmd_cram_freemuxlet.cpp:107-161,359-370,432-653 (compiled as
    # verbatim ranges, oracle/ref_hot.cpp.in scref_freemuxlet_old); ten iterations, geno_error in the last one only
    em_ge, em_dp = 0.05, 0.5
    ref = rb.RefScl.from_packed(p).freemuxlet_old(K, cl, em_dp, em_ge, full_ll=True, cluster_pileups=True)
    assert ref["n_iter"] == 10
    cplp = ob.fmx_build_cluster_pileup(p, e, K, cl)
    cells = ob.fmx_init_cells(cl)
    for it in range(10):
        nsng, namb, _, full = ob.fmx_iterate(p, e, K, cplp, cells, em_dp, em_ge if it == 9 else 0.0, full_ll=True)
        assert (nsng, namb) == tuple(ref["counters"][it]), "oracle != reference"
        for f in cells.dtype.names:
            if f not in ("clust", "_pad"):
                assert cells[f].tobytes() == ref["cells"][it][f].tobytes(), ("oracle != reference", it, f)
        assert np.array_equal(full, ref["full_ll"][it]) and cplp.tobytes() == ref["cplp"][it].tobytes()
    fin = ref["cplp"][9]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), C=p.C, S=p.S, K=K, cell_ptr=p.cell_ptr,
                        entry_snp=p.entry_snp, entry_rptr=p.entry_rptr, reads=p.reads, af=p.af, bf_thres=thres,
                        frac_init_clust=frac, order=order, dropd=dd, jitter0=jit0, clust0=clust0, ccounts0=cc0,
                        orands=np.array(orands), jitters=np.array(jits), clusts=np.array(clusts),
                        changed=np.array(changed),
                        em_geno_error=em_ge, em_doublet_prior=em_dp, em_counters=ref["counters"], em_cells=ref["cells"],
                        em_full_ll_first=ref["full_ll"][0], em_full_ll_last=ref["full_ll"][9],
                        em_cluster_gls=fin["gls"], em_cluster_cnt=np.stack([fin["nreads"], fin["nref"], fin["nalt"]], axis=-1),
                        em_source="reference: oracle/_ref/libscdrop_ref.so scref_freemuxlet_old "
                                  "(cmd_cram_freemuxlet.cpp:107-108,113-161,359-370,432-453,456-653)")
    print(name, "cells", p.C, "pairs", dd.size, "first pass", np.bincount(clust0[clust0 >= 0], minlength=K), "changed",
          changed)


if __name__ == "__main__":
    demux_case("demux_v4_a2", 60, 1500, 4, (0.0, 0.5), seed=101, mean_entries=200, missing_gp_frac=0.05)
    demux_case("demux_v4_a6", 40, 1500, 4, (0.0, 0.1, 0.2, 0.3, 0.4, 0.5), seed=102, mean_entries=200)
    demux_case("demux_v16_a2", 40, 3000, 16, (0.0, 0.5), seed=103, mean_entries=300)
    fmx_case("fmx_k4", 120, 1500, 4, seed=104, n_iter=4, mean_entries=200)
    # round 5: richer cases (deep entries, allele "2", a third doublets, shallow droplets that come out AMB)
    demux_case("demux_v8_a3_deep", 48, 1200, 8, (0.0, 0.25, 0.5), seed=106, mean_entries=120, reads_lambda=2.5,
               other=0.03, doublet_frac=0.35, missing_gp_frac=0.03)
    demux_case("demux_v64_a6", 6, 2500, 64, (0.0, 0.1, 0.2, 0.3, 0.4, 0.5), seed=107, mean_entries=350)
    fmx_case("fmx_k4_mixed", 200, 400, 4, seed=108, n_iter=4, mean_entries=60, min_entries=4, reads_lambda=1.0,
             other=0.03, doublet_frac=0.3)   # 154 SNG / 45 DBL / 1 AMB, assignments moving for three iterations
    fmxold_case("fmxold_k4", 90, 600, 4, seed=105, mean_entries=150, min_entries=30)
