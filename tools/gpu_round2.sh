#!/bin/bash
# GPU check of the multi-device layers: new group/slab tests, the distributed drivers, a short bench of both legs.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_group_gpu.py tests/test_fmx_shard_gpu.py tests/test_fmx_dist_gpu.py -x -q > gpurun_out/pytest_group.log 2>&1; echo "pytest group rc=$?"; tail -15 gpurun_out/pytest_group.log
timeout 900 python bench.py --steps 200 --warmup 20 > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"; tail -c 3000 gpurun_out/bench_default.log
