#!/usr/bin/env python
"""Sustained kernel times of the demuxlet pass on BASELINE configs[1] (or a config given as argv[1]):
mean hipEvent times over N passes after a warm-up.  MUXGL_LIB selects a variant build (tools/build_variant.sh)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from popscle_amd import muxgl, synth  # noqa: E402

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cfg = synth.CONFIGS[idx]
p = synth.make_pileup(cfg["C"], cfg["S"], cfg["V"], seed=synth.BASE_SEED + idx)
eng = muxgl.Engine(0, int(os.environ.get("MUXGL_PROBE_FLAGS", "0")))
eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
eng.demux_set_gp(p.gp, p.has_gp)
for _ in range(n // 2):
    eng.demux_run(cfg["alphas"], 0.5, want_cells=False)
import time
ms = np.zeros(muxgl.T_COUNT)
t0 = time.perf_counter()
for _ in range(n):
    eng.demux_run(cfg["alphas"], 0.5, want_cells=False)
    ms += eng.timing()
wall = (time.perf_counter() - t0) / n * 1e3
ms /= n
print(f"wall {wall:.4f} ms/step  {os.environ.get('MUXGL_LIB', 'default')[-28:]:28s} sweep {ms[muxgl.T_DEMUX_SWEEP]:.4f} ms  finish {ms[muxgl.T_DEMUX_REDUCE]:.4f} ms")
