"""stage times of muxgl_fmx_prepare / set_clusters at a BASELINE config size (GPU box): python tools/prep_probe.py [4]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from popscle_amd import muxgl, synth
cfgi = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = synth.CONFIGS[cfgi]
d = synth.make_pileup_device(cfg["C"], cfg["S"], cfg["V"], seed=synth.BASE_SEED + cfgi, with_gp=False)
p = d.host()
del d
with muxgl.Engine(0) as e:
    t0 = time.perf_counter(); e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads); print(f"set_pileup {time.perf_counter()-t0:.3f} s")
    for rep in range(2):
        t0 = time.perf_counter(); llk0, llk2, _, _ = e.fmx_prepare(p.af); print(f"fmx_prepare {time.perf_counter()-t0:.3f} s")
    clust = (np.arange(cfg["C"]) % cfg["V"]).astype(np.int32)
    for rep in range(2):
        t0 = time.perf_counter(); e.fmx_set_clusters(cfg["V"], clust); print(f"fmx_set_clusters {time.perf_counter()-t0:.3f} s")
    t0 = time.perf_counter(); r = e.fmx_iterate(); print(f"fmx_iterate {time.perf_counter()-t0:.3f} s")
