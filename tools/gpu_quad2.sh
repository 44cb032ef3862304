#!/bin/bash
# bucket size of the quad kernel's launch order: sweep time of config 1
for b in 64 256 1024 4096 16384 1000000; do
MUXGL_QUAD_BUCKET=$b python tools/scale_probe.py demux 1 1.0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bucket=$b', {k:d[k] for k in ('step_s','sweep_ms','reduce_ms')})"
done
for b in 256 1024 4096 1000000; do
MUXGL_QUAD_BUCKET=$b MUXGL_PROBE_NO_GREEDY=1 python tools/scale_probe.py fmx 3 1.0 4 | python -c "import sys,json; d=json.loads(sys.stdin.read()); i=d['iterations'][-1]; print('fmx bucket=$b', {k:round(i[k],3) for k in ('estep','mstep')})"
done
timeout 600 python -m pytest tests/test_fmx_gpu.py tests/test_fmx_shard_gpu.py tests/test_demux_gpu.py -x -q 2>&1 | tail -2
