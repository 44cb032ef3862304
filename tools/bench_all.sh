#!/bin/bash
# Runs on the GPU box: the bench line of every BASELINE config (1, 2 demuxlet; 3, 4 freemuxlet), one JSON line each,
# into gpurun_out/bench_lines.jsonl
mkdir -p gpurun_out; : > gpurun_out/bench_lines.jsonl
python bench.py --steps 2000 --warmup 200 2> gpurun_out/bench_c1.err | grep "^{" >> gpurun_out/bench_lines.jsonl; echo "c1 rc=$?"
python bench.py --config 2 2> gpurun_out/bench_c2.err | grep "^{" >> gpurun_out/bench_lines.jsonl; echo "c2 rc=$?"
python bench.py --config 3 --steps 200 --warmup 20 2> gpurun_out/bench_c3.err | grep "^{" >> gpurun_out/bench_lines.jsonl; echo "c3 rc=$?"
python bench.py --config 4 2> gpurun_out/bench_c4.err | grep "^{" >> gpurun_out/bench_lines.jsonl; echo "c4 rc=$?"
python - <<'P'
import json
for ln in open("gpurun_out/bench_lines.jsonl"):
    d = json.loads(ln); r = d["roofline"]
    print(d["config"]["workload"][:60], "| %.4g %s | %.3f ms/step | bound %s frac %.3f | hbm %s | fp64 %.3f valu %s" % (
        d["value"], d["unit"], d["ms_per_step"], r["bound"], r["frac"] or 0, r["hbm"]["frac"], r["fp64"]["frac"], r["fp64"]["valu_issue_frac"]))
    if d.get("freemuxlet_em"): print("   fmx leg:", d["freemuxlet_em"].get("ms_per_step"), d["freemuxlet_em"].get("error"))
P
