#!/usr/bin/env python
"""Where a demuxlet step's time goes besides its kernels (configs[1]): wall time per muxgl_demux_run call against the
hipEvent kernel times, with and without the Python-side per-step calls.  MUXGL_NO_EVENTS=1 drops the event records."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from popscle_amd import muxgl, synth  # noqa: E402

cfg = synth.CONFIGS[1]
p = synth.make_pileup(cfg["C"], cfg["S"], cfg["V"], seed=synth.BASE_SEED + 1)
eng = muxgl.Engine(0)
eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
eng.demux_set_gp(p.gp, p.has_gp)
al = cfg["alphas"]
for _ in range(3000):
    eng.demux_run(al, 0.5, want_cells=False)
n = 2000
tb = np.zeros(muxgl.T_COUNT, dtype=np.float32)
ks = np.zeros(muxgl.T_COUNT)
t0 = time.perf_counter()
for _ in range(n):
    eng.demux_run(al, 0.5, want_cells=False)
    ks += eng.timing(tb)
w1 = (time.perf_counter() - t0) / n * 1e3
t0 = time.perf_counter()
for _ in range(n):
    eng.demux_run(al, 0.5, want_cells=False)
w2 = (time.perf_counter() - t0) / n * 1e3
import ctypes as C
lib, h, dp = eng.lib, eng.h, eng._dp
f = lib.muxgl_demux_run
t0 = time.perf_counter()
for _ in range(n):
    f(h, C.byref(dp), None, None)
w3 = (time.perf_counter() - t0) / n * 1e3
ks /= n
print(f"events {'off' if os.environ.get('MUXGL_NO_EVENTS') else 'on'}: run+timing {w1:.4f} ms, run only {w2:.4f} ms, bare ctypes call {w3:.4f} ms; "
      f"kernels sweep {ks[muxgl.T_DEMUX_SWEEP]:.4f} + finish {ks[muxgl.T_DEMUX_REDUCE]:.4f} = {ks[muxgl.T_DEMUX_SWEEP] + ks[muxgl.T_DEMUX_REDUCE]:.4f} ms")
