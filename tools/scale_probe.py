#!/usr/bin/env python
"""Timing probe of the larger BASELINE.json configs (not the headline bench; numbers go to DESIGN.md).

    python tools/scale_probe.py demux 2 0.1      # configs[2] (100k x 64 x 200k, 6 alphas) at 10 % of the cells
    python tools/scale_probe.py fmx 3 0.2 5      # configs[3] (50k cells, K=16, 100k SNPs) at 20 %, 5 EM iterations
    python tools/scale_probe.py fmxold 1 1.0     # freemuxlet-old's pair matrix + votes on the cells/SNPs of configs[1]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from popscle_amd import muxgl, synth  # noqa: E402


def main():
    kind, idx, scale = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    cfg = synth.CONFIGS[idx]
    C = int(cfg["C"] * scale)
    t0 = time.time()
    p = synth.make_pileup(C, cfg["S"], cfg["V"], seed=synth.BASE_SEED + idx, with_gp=(kind == "demux"),
                          **({"cap_bq": 60} if kind == "fmxold" else {}))
    out = {"kind": kind, "config": idx, "cells": C, "V": cfg["V"], "S": cfg["S"], "entries": p.nnz, "reads": p.R,
           "gen_s": round(time.time() - t0, 1)}
    eng = muxgl.Engine(0, int(os.environ.get("MUXGL_PROBE_FLAGS", "0")))
    t0 = time.time()
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    out["upload_s"] = round(time.time() - t0, 2)
    if kind == "demux":
        alphas = cfg["alphas"]
        eng.demux_set_gp(p.gp, p.has_gp)
        eng.demux_run(alphas, 0.5, want_cells=False)
        ts = []
        for _ in range(3):
            t0 = time.time()
            eng.demux_run(alphas, 0.5, want_cells=False)
            ts.append(time.time() - t0)
        ms = eng.timing()
        cells = eng.demux_results_view()
        V, A = cfg["V"], len(alphas)
        out.update(step_s=min(ts), sweep_ms=float(ms[muxgl.T_DEMUX_SWEEP]), reduce_ms=float(ms[muxgl.T_DEMUX_REDUCE]),
                   call_ms=float(ms[muxgl.T_DEMUX_CALL]), lls_per_s=C * (V + V * (V - 1) * (A - 1)) / min(ts),
                   entries_per_s=p.nnz / min(ts), types=np.bincount(cells["type"], minlength=3).tolist(),
                   singlet_acc=float((cells["sBest"][~p.truth["is_doublet"]] == p.truth["s1"][~p.truth["is_doublet"]]).mean()))
    elif kind == "fmxold":
        K = cfg["V"]
        llk0, llk2, _, _ = eng.fmx_prepare(p.af)
        snp_n = np.bincount(p.entry_snp, minlength=p.S).astype(np.int64)
        out["pair_terms"] = int((snp_n * (snp_n - 1) // 2).sum())
        t0 = time.time()
        eng.fmxold_pair_dist(5.41)
        out["pair_dist_s"] = round(time.time() - t0, 3)
        out["pair_kernel_ms"] = float(eng.timing()[muxgl.T_FMXOLD_PAIR])
        t0 = time.time()
        eng.fmxold_pair_dist(5.41)
        out["pair_dist_s_2nd"] = round(time.time() - t0, 3)
        out["pair_kernel_ms_2nd"] = float(eng.timing()[muxgl.T_FMXOLD_PAIR])
        out["pair_terms_per_s"] = out["pair_terms"] / (out["pair_kernel_ms_2nd"] * 1e-3)
        rng = np.random.default_rng(1)
        order = np.argsort(-(llk2 - llk0), kind="stable").astype(np.int32)
        jit = rng.integers(0, 2**31, (C, K)) / 2.0**31 / 1000.0
        t0 = time.time()
        clust, cc = eng.fmxold_vote_init(K, order, jit)
        out["vote_init_s"] = round(time.time() - t0, 3)
        out["vote_init_kernel_ms"] = float(eng.timing()[muxgl.T_FMXOLD_VOTE])
        passes = []
        for it in range(3):
            orand = rng.permutation(C).astype(np.int32)
            t0 = time.time()
            clust, ch, cc = eng.fmxold_vote_refine(K, orand, jit, clust)
            passes.append(dict(wall_s=round(time.time() - t0, 3), kernel_ms=float(eng.timing()[muxgl.T_FMXOLD_VOTE]),
                               changed=int(ch)))
        out["refine_passes"] = passes
        out["cluster_sizes"] = cc.tolist()
        sng = ~p.truth["is_doublet"]
        agree = sum(int(np.bincount(p.truth["s1"][sng & (clust == k)], minlength=K).max()) for k in range(K))
        out["singlets_with_cluster_majority"] = agree / int(sng.sum())
    else:
        K = cfg["V"]
        iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
        t0 = time.time()
        llk0, llk2, _, _ = eng.fmx_prepare(p.af)
        out["prepare_s"] = round(time.time() - t0, 2)
        out["prepare_kernel_ms"] = float(eng.timing()[muxgl.T_FMX_ENTRY])
        if not os.environ.get("MUXGL_PROBE_NO_GREEDY"):  # (thousands of small launches: skipped under counter collection)
            t0 = time.time()
            g = eng.fmx_greedy_init(K, llk2 - llk0)
            out["greedy_init_s"] = round(time.time() - t0, 3)
            out["greedy_clusters_used"] = int(len(np.unique(g[g >= 0])))
        clust0 = np.where(np.random.default_rng(0).random(C) < 0.9, p.truth["s1"], -1).astype(np.int32)
        t0 = time.time()
        eng.fmx_set_clusters(K, clust0)
        out["set_clusters_s"] = round(time.time() - t0, 3)
        its = []
        for _ in range(iters):
            t0 = time.time()
            cells, st = eng.fmx_iterate(0.5, 0.1)
            ms = eng.timing()
            its.append(dict(wall_s=round(time.time() - t0, 4), gp=float(ms[muxgl.T_FMX_GP]),
                            estep=float(ms[muxgl.T_FMX_ESTEP]), call=float(ms[muxgl.T_FMX_CALL]),
                            mstep=float(ms[muxgl.T_FMX_MSTEP]), stats=st))
        out["iterations"] = its
        ok = (cells["type"] == 0) & ~p.truth["is_doublet"]
        out["singlets_in_true_cluster"] = float((cells["clust"][ok] == p.truth["s1"][ok]).mean())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
