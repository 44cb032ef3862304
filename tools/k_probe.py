#!/usr/bin/env python
"""freemuxlet EM iteration time for a given number of clusters (which E-step kernel a shape gets: quad <= 16, two
clusters per lane <= 32, wave kernels above):  python tools/k_probe.py K cells [snps]   (MUXGL_FLAGS=<int> to force)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from popscle_amd import muxgl, synth  # noqa: E402

K, C = int(sys.argv[1]), int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
p = synth.make_pileup(C, S, K, seed=7, with_gp=False)
eng = muxgl.Engine(0, int(os.environ.get("MUXGL_FLAGS", "0")))
eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
eng.fmx_prepare(p.af)
eng.fmx_set_clusters(K, (np.arange(C) % K).astype(np.int32))
best, slots = 1e9, None
for it in range(6):
    t0 = time.time()
    cells, st = eng.fmx_iterate(0.5, 0.1, want_cells=False)[:2]
    dt = time.time() - t0
    if dt < best:
        best, slots = dt, eng.timing()
print(json.dumps({"K": K, "cells": C, "entries": p.nnz, "iter_s": best, "gp_ms": float(slots[muxgl.T_FMX_GP]),
                  "estep_ms": float(slots[muxgl.T_FMX_ESTEP]), "call_ms": float(slots[muxgl.T_FMX_CALL]),
                  "mstep_ms": float(slots[muxgl.T_FMX_MSTEP]), "stats": [int(x) for x in st]}))
