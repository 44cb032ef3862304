#!/bin/bash
# clocks and power while the freemuxlet E-step runs back to back (configs[4] at 10 %, 150 iterations), then a kernel trace
mkdir -p gpurun_out; O=gpurun_out
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > $O/smi_fmx.jsonl &
SMI=$!
MUXGL_PROBE_NO_GREEDY=1 python tools/scale_probe.py fmx 4 0.1 ${CLK_ITERS:-150} | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=[i['estep'] for i in d['iterations']]; print('estep first/median/last', e[0], sorted(e)[len(e)//2], e[-1])"
wait $SMI
python - <<'P'
import json
for l in open('gpurun_out/smi_fmx.jsonl'):
    try: d=json.loads(l)['card0']
    except Exception: continue
    print({k:v for k,v in d.items() if 'sclk' in k or 'Power' in k or 'mclk' in k})
P
cd /tmp; export TMPDIR=/tmp
MUXGL_PROBE_NO_GREEDY=1 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/kt_fmx -- python /root/repo/tools/scale_probe.py fmx 4 0.1 3 > /root/repo/$O/kt_fmx.log 2>&1
cd /root/repo; f=$(find $O/kt_fmx -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-160
