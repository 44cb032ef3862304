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


if __name__ == "__main__":
    demux_case("demux_v4_a2", 60, 1500, 4, (0.0, 0.5), seed=101, mean_entries=200, missing_gp_frac=0.05)
    demux_case("demux_v4_a6", 40, 1500, 4, (0.0, 0.1, 0.2, 0.3, 0.4, 0.5), seed=102, mean_entries=200)
    demux_case("demux_v16_a2", 40, 3000, 16, (0.0, 0.5), seed=103, mean_entries=300)
    fmx_case("fmx_k4", 120, 1500, 4, seed=104, n_iter=4, mean_entries=200)
