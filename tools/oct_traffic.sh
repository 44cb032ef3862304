#!/bin/bash
# Runs on the GPU box: step time and fabric traffic (FETCH_SIZE / WRITE_SIZE passes, raw counter units / 1e6) of the oct
# kernels of configs[3] and configs[1] under the current environment (MUXGL_FMX_CH, MUXGL_OCT_CH, MUXGL_LIB ...), once per
# label given.  usage: bash tools/oct_traffic.sh "label"
cd /root/repo; export TMPDIR=/tmp
GS=${1:-"run"}
O=/root/repo/gpurun_out/octg; rm -rf $O; mkdir -p $O
for cfg in 3 1; do
  pat=$([ $cfg = 3 ] && echo fmx_estep_oct_kernel || echo demux_oct_kernel)
  for g in $GS; do
    python bench.py --config $cfg --steps $([ $cfg = 3 ] && echo 60 || echo 600) --warmup 20 --no-cpu-baseline --no-fmx-leg 2>/dev/null | tail -1 > $O/c${cfg}_g$g.json
    for ctr in FETCH_SIZE WRITE_SIZE; do
      (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $O/c${cfg}_g${g}_$ctr -- python /root/repo/bench.py --config $cfg --no-cpu-baseline --no-fmx-leg --steps 3 --warmup 1 --ramp-seconds 0 > $O/c${cfg}_g${g}_$ctr.log 2>&1)
    done
    python - $O $cfg $g $pat <<'P'
import csv, glob, json, sys
O, cfg, g, pat = sys.argv[1:5]
d = json.loads(open(f"{O}/c{cfg}_g{g}.json").read())
k = d.get("kernel_ms_rank0_last_iteration") or d.get("kernel_ms") or {}
out = {"cfg": cfg, "G": g, "ms_per_step": round(d["ms_per_step"], 4), "kernel_ms": d.get("roofline", {}).get("kernel_ms")}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(f"{O}/c{cfg}_g{g}_{ctr}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                v.append(float(r["Counter_Value"]))
    out[ctr] = round(sum(v) / len(v) / 1e6, 1) if v else None
print(json.dumps(out))
P
  done
done
