#!/bin/bash
# measurement pass: kernel trace + PMC passes of every BASELINE config at FULL size, then the bench lines
# (summaries: tools/prof_summary.py, tools/traffic_update.py)
# NB: gpurun MERGES what this writes into the container's gpurun_out/: delete gpurun_out/prof_c* THERE before the call, or
# the summaries mix this pass with the files of an earlier one (their names carry process ids).
cd /root/repo
rm -rf gpurun_out/prof_c[1-4]_*
bash tools/profile_bench.sh 1 200 2>&1 | tail -1 | cut -c1-150
bash tools/profile_bench.sh 3 20 2>&1 | tail -1 | cut -c1-150
bash tools/profile_bench.sh 2 3 2>&1 | tail -1 | cut -c1-150
bash tools/profile_bench.sh 4 3 2>&1 | tail -1 | cut -c1-150
bash tools/bench_all.sh 2>&1 | tail -8
