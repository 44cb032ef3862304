#!/bin/bash
# fmx quad E-step (K <= 16): parity tests, then the iteration times of configs[3]
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_fmx_gpu.py tests/test_fmx_shard_gpu.py -x -q > gpurun_out/pytest_fq.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_fq.log
for i in 1 2; do
MUXGL_PROBE_NO_GREEDY=1 python tools/scale_probe.py fmx 3 1.0 4 | python -c "import sys,json; d=json.loads(sys.stdin.read()); i=d['iterations'][-1]; print({k:round(i[k],3) for k in ('gp','estep','call','mstep')})"
done
