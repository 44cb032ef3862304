#!/usr/bin/env python
"""How often the exact path runs on a BASELINE-shaped freemuxlet job and what an iteration costs with it
(usage: python tools/fmx_exact_probe.py [config] [iterations])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popscle_amd import muxgl, synth

config = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = synth.CONFIGS[config]
K = cfg["V"]
if os.environ.get("PROBE_NUMPY"):
    p = synth.make_pileup(cfg["C"], cfg["S"], K, seed=synth.BASE_SEED + config, with_gp=False)
else:   # what bench.py feeds the legs of this size: the torch generator on the device
    d = synth.make_pileup_device(cfg["C"], cfg["S"], K, device="cuda:0", seed=synth.BASE_SEED + config, with_gp=False,
                                 mean_entries=960.0, min_entries=50)
    p = d.host()
    del d
clust0 = np.where(np.random.default_rng(0).random(p.C) < 0.9, p.truth["s1"], -1).astype(np.int32)
with muxgl.Engine(0) as e:
    e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    e.fmx_prepare(p.af)
    for rep in range(3):
        t0 = time.perf_counter()
        e.fmx_set_clusters(K, clust0)
        print(f"set_clusters: {(time.perf_counter() - t0) * 1e3:.2f} ms")
        for it in range(iters):
            t0 = time.perf_counter()
            _, st = e.fmx_iterate(0.5, 0.1, want_cells=False)
            dt = time.perf_counter() - t0
            print(f"iter {it}: {dt*1e3:.3f} ms host, stats {st}, exact {e.fmx_exact_stats()}, kernels {np.round(e.timing()[4:9], 3).tolist()}")
