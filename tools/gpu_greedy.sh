#!/bin/bash
# batch size of the greedy initial clustering: time at configs[3] (and the greedy tests) for several builds
mkdir -p gpurun_out
run() { python tools/scale_probe.py fmx 3 1.0 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d.get('greedy_init_s'), d.get('greedy_clusters_used'))"; }
run "GB=32"
for gb in 64 128; do
  touch popscle_amd/csrc/fmx_greedy.hip
  make -C popscle_amd/csrc EXTRA="-DFMX_GREEDY_GB=$gb" > gpurun_out/greedy_build.log 2>&1 || { echo build failed; tail -3 gpurun_out/greedy_build.log; continue; }
  run "GB=$gb"; run "GB=$gb again"
  timeout 600 python -m pytest tests/test_fmx_gpu.py -x -q -k greedy 2>&1 | tail -1
done
