#!/bin/bash
# round 2, third measurement pass: bench lines of all configs, then kernel trace + PMC passes of each
cd /root/repo
bash tools/bench_all.sh 2>&1 | tail -12
bash tools/profile_bench.sh 1 200 2>&1 | tail -3
bash tools/profile_bench.sh 3 20 2>&1 | tail -3
bash tools/profile_bench.sh 2 3 2>&1 | tail -3
bash tools/profile_bench.sh 4 3 --scale 0.1 2>&1 | tail -3
du -sh gpurun_out | tail -1
