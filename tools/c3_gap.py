"""configs[3]: wall time of an EM iteration through the C-ABI (ctypes, no bench.py around it), with and without the event
records around the kernels (MUXGL_NO_EVENTS=1 in the environment).  usage: python tools/c3_gap.py [iterations]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from popscle_amd import muxgl, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = synth.CONFIGS[3]
C, S, K = cfg["C"], cfg["S"], cfg["V"]
p = synth.make_pileup(C, S, K, seed=synth.BASE_SEED + 3, with_gp=False, mean_entries=960)
with muxgl.Engine(0) as eng:
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    llk0, llk2, _, _ = eng.fmx_prepare(p.af)
    clust = eng.fmx_greedy_init(K, llk2 - llk0)
    eng.fmx_set_clusters(K, clust)
    for _ in range(30):
        eng.fmx_iterate(0.5, 0.1, want_cells=False)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.fmx_iterate(0.5, 0.1, want_cells=False)
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f"events {'off' if os.environ.get('MUXGL_NO_EVENTS') else 'on'}: {dt:.4f} ms per iteration")
