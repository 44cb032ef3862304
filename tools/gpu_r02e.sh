#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_fmx_gpu.py tests/test_bench_gpu.py -x -q > gpurun_out/pytest_fq.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_fq.log
rm -rf gpurun_out/prof_c[1-4]_*
bash tools/profile_bench.sh 1 200 2>&1 | tail -1 | cut -c1-100
bash tools/profile_bench.sh 3 20 2>&1 | tail -1 | cut -c1-100
bash tools/profile_bench.sh 2 3 2>&1 | tail -1 | cut -c1-100
bash tools/profile_bench.sh 4 3 --scale 0.1 2>&1 | tail -1 | cut -c1-100
