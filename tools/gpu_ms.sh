#!/bin/bash
# M-step on six-value states: freemuxlet parity tests, then the iteration times of configs[3] and 10 % of configs[4]
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_fmx_gpu.py tests/test_fmx_shard_gpu.py tests/test_group_gpu.py tests/test_fmxold_gpu.py tests/test_fmx_dist_gpu.py -x -q > gpurun_out/pytest_ms.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_ms.log
for a in "3 1.0" "4 0.1"; do
MUXGL_PROBE_NO_GREEDY=1 python tools/scale_probe.py fmx $a 4 | python -c "import sys,json; d=json.loads(sys.stdin.read()); i=d['iterations'][-1]; print('$a', {k:round(i[k],3) for k in ('gp','estep','call','mstep')})"
done
