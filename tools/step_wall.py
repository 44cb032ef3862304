"""wall time of the configs[1] step (muxgl_demux_run, records on the host at return) over N calls; environment knobs of
the library (MUXGL_NO_EVENTS, ...) are inherited.  usage: python tools/step_wall.py [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popscle_amd import muxgl, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
p = synth.make_config(1)
eng = muxgl.Engine(0)
eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
eng.demux_set_gp(p.gp, p.has_gp)
t = time.perf_counter()
while time.perf_counter() - t < 1.0:
    eng.demux_run((0.0, 0.5), 0.5, want_cells=False)
best = 1e9
for rep in range(3):
    t = time.perf_counter()
    for _ in range(steps):
        eng.demux_run((0.0, 0.5), 0.5, want_cells=False)
    best = min(best, (time.perf_counter() - t) / steps)
print(f"step wall {best * 1e3:.4f} ms  (MUXGL_NO_EVENTS={os.environ.get('MUXGL_NO_EVENTS')})")
