"""Regenerates the golden vectors under tests/golden/ from the CPU oracle (oracle/muxgl_oracle.c).

These are REGRESSION vectors of the restatement, not outputs of the reference binary: popscle's demuxlet/freemuxlet
translation units need htslib, which this image lacks, so the reference cannot be run here and ships no fixtures of its
own (oracle/muxgl_oracle.h, "parity unpinned").  Inputs come from popscle_amd.synth with fixed seeds; each .npz holds
the packed inputs and the oracle's outputs, so the GPU box can check the HIP path without the oracle source of truth
changing under it.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_binding as ob  # noqa: E402
from popscle_amd import synth  # noqa: E402


def demux_case(name, C, S, V, alphas, seed, **kw):
    p = synth.make_pileup(C, S, V, seed=seed, **kw)
    cells, full = ob.demux(p, alphas=alphas, doublet_prior=0.5, full_ll=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), C=p.C, S=p.S, cell_ptr=p.cell_ptr, entry_snp=p.entry_snp,
                        entry_rptr=p.entry_rptr, reads=p.reads, af=p.af, gp=p.gp, has_gp=p.has_gp,
                        alphas=np.array(alphas), doublet_prior=0.5, cells=cells, full_ll=full)
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
    cnt = np.stack([cplp["nreads"], cplp["nref"], cplp["nalt"]], axis=-1)
    ecnt = np.stack([e["nreads"], e["nref"], e["nalt"]], axis=-1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), C=p.C, S=p.S, K=K, cell_ptr=p.cell_ptr,
                        entry_snp=p.entry_snp, entry_rptr=p.entry_rptr, reads=p.reads, af=p.af, entry_gls=e["gls"],
                        entry_cnt=ecnt, llk0=llk0, llk2=llk2, nsnps=ns, nreads=nr, clust0=clust0, n_iter=n_iter,
                        stats=np.array(stats), cells=cells, cluster_gls=cplp["gls"], cluster_cnt=cnt, full_ll_iter1=ll1)
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
    np.savez_compressed(os.path.join(HERE, name + ".npz"), C=p.C, S=p.S, K=K, cell_ptr=p.cell_ptr,
                        entry_snp=p.entry_snp, entry_rptr=p.entry_rptr, reads=p.reads, af=p.af, bf_thres=thres,
                        frac_init_clust=frac, order=order, dropd=dd, jitter0=jit0, clust0=clust0, ccounts0=cc0,
                        orands=np.array(orands), jitters=np.array(jits), clusts=np.array(clusts),
                        changed=np.array(changed))
    print(name, "cells", p.C, "pairs", dd.size, "first pass", np.bincount(clust0[clust0 >= 0], minlength=K), "changed",
          changed)


if __name__ == "__main__":
    demux_case("demux_v4_a2", 60, 1500, 4, (0.0, 0.5), seed=101, mean_entries=200, missing_gp_frac=0.05)
    demux_case("demux_v4_a6", 40, 1500, 4, (0.0, 0.1, 0.2, 0.3, 0.4, 0.5), seed=102, mean_entries=200)
    demux_case("demux_v16_a2", 40, 3000, 16, (0.0, 0.5), seed=103, mean_entries=300)
    fmx_case("fmx_k4", 120, 1500, 4, seed=104, n_iter=4, mean_entries=200)
    fmxold_case("fmxold_k4", 90, 600, 4, seed=105, mean_entries=150, min_entries=30)
