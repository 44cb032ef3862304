#!/usr/bin/env python
"""Kernel times of the configs[1] step with the library named by MUXGL_LIB (also one of an earlier round, whose ABI lacks
the newer symbols): python tools/ab_headline.py [passes]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popscle_amd import muxgl, synth

lib = ctypes.CDLL(muxgl.LIB_PATH)
for name in list(muxgl.SYMBOLS):
    if not hasattr(lib, name):
        del muxgl.SYMBOLS[name]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
p = synth.make_config(1)
al = synth.CONFIGS[1]["alphas"]
with muxgl.Engine(0) as e:
    e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    e.demux_set_gp(p.gp, p.has_gp)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        e.demux_run(al, 0.5, want_cells=False)
    e.timing_sum(reset=True)
    t0 = time.perf_counter()
    for _ in range(n):
        e.demux_run(al, 0.5, want_cells=False)
    dt = time.perf_counter() - t0
    ms, k = e.timing_sum()
    print(os.path.basename(muxgl.LIB_PATH), f"step {dt / n * 1e3:.4f} ms  sweep {ms[muxgl.T_DEMUX_SWEEP] / k:.4f}  finish {ms[muxgl.T_DEMUX_REDUCE] / k:.4f}  d2h {ms[muxgl.T_DEMUX_D2H] / k:.4f}")
