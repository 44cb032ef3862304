"""greedy initial clustering: time and clusters at a BASELINE config size (GPU box).  usage: python tools/greedy_time.py [3|4] [scale]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from popscle_amd import muxgl, synth

cfgi = int(sys.argv[1]) if len(sys.argv) > 1 else 3
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cfg = synth.CONFIGS[cfgi]
C = int(cfg["C"] * scale)
d = synth.make_pileup_device(C, cfg["S"], cfg["V"], seed=synth.BASE_SEED + cfgi, with_gp=False)
p = d.host()
del d
with muxgl.Engine(0) as e:
    e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    llk0, llk2, _, _ = e.fmx_prepare(p.af)
    for rep in range(3):
        t0 = time.perf_counter()
        clust = e.fmx_greedy_init(cfg["V"], llk2 - llk0)
        dt = time.perf_counter() - t0
        print(f"config {cfgi}: {C} cells, K = {cfg['V']}: greedy init {dt:.3f} s, clusters used {len(np.unique(clust))}, checksum {int(np.sum(clust.astype(np.int64) * (np.arange(C) % 997)))}")
