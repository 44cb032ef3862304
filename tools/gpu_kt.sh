#!/bin/bash
# kernel trace of a command: top of the per-kernel statistics.  usage: bash tools/gpu_kt.sh <tag> <command...>
tag=$1; shift
mkdir -p /root/repo/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/kt_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/kt_$tag -- "$@" > /root/repo/gpurun_out/kt_$tag.log 2>&1
cd /root/repo; f=$(find gpurun_out/kt_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv, sys, re
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n)
    print(f"{n[:70]:70s} calls {r['Calls']:>5} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
P
find gpurun_out/kt_$tag -type f ! -name "*kernel_stats.csv" -delete
