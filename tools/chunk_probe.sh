#!/bin/bash
# Runs on the GPU box: chunk length sweep of the oct kernels (configs[3]: MUXGL_FMX_CH, configs[1]: MUXGL_OCT_CH) with
# step time, kernel time and fabric traffic.  usage: bash tools/chunk_probe.sh "48 64 96 128 192"
cd /root/repo
for ch in ${1:-"48 64 96 128 192"}; do
  echo "== chunk $ch"
  MUXGL_FMX_CH=$ch MUXGL_OCT_CH=$ch bash tools/oct_traffic.sh "ch$ch" 2>&1 | grep '^{'
  MUXGL_FMX_CH=$ch python bench.py --config 3 --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   c3 kernels', {k: round(v,3) for k,v in d['kernel_ms_rank0_last_iteration'].items()})"
  MUXGL_OCT_CH=$ch python bench.py --config 1 --steps 600 --warmup 50 --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   c1 kernels', {k: round(v,4) for k,v in d['kernel_ms'].items()})"
done
