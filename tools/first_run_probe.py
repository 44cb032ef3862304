"""first and later muxgl_demux_run calls at a BASELINE config (GPU box): python tools/first_run_probe.py [config] [scale]"""
import sys, time, os
sys.path.insert(0, ".")
from popscle_amd import muxgl, synth
cfgi = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cfg = synth.CONFIGS[cfgi]
p = synth.make_pileup_device(int(cfg["C"] * scale), cfg["S"], cfg["V"], seed=synth.BASE_SEED + cfgi).host()
with muxgl.Engine(0, int(os.environ.get('PROBE_FLAGS', '0'))) as e:
    t0 = time.perf_counter(); e.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads); e.demux_set_gp(p.gp, p.has_gp)
    print(f"hand-over {time.perf_counter() - t0:.3f} s")
    for i in range(3):
        t0 = time.perf_counter(); e.demux_run(tuple(cfg["alphas"]), 0.5, want_cells=False)
        print(f"demux_run #{i}: {time.perf_counter() - t0:.3f} s")
