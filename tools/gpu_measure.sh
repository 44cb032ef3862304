#!/bin/bash
# measurement pass: GPU tests, then kernel trace + PMC passes of every BASELINE config (summaries: tools/prof_summary.py,
# tools/traffic_update.py; bench lines: tools/bench_all.sh)
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu_full.log
rm -rf gpurun_out/prof_c[1-4]_*
bash tools/profile_bench.sh 1 200 2>&1 | tail -1 | cut -c1-150
bash tools/profile_bench.sh 3 20 2>&1 | tail -1 | cut -c1-150
bash tools/profile_bench.sh 2 3 2>&1 | tail -1 | cut -c1-150
bash tools/profile_bench.sh 4 3 --scale 0.1 2>&1 | tail -1 | cut -c1-150
