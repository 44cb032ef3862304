#!/bin/bash
# linear-entry path: parity of the K > 32 E-step tests + timing of configs[4] at 10 % with and without it
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fmx_gpu.py tests/test_group_gpu.py -x -q -k "64 or 70 or 100 or 33" > gpurun_out/pytest_lin.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_lin.log
python tools/scale_probe.py fmx 4 0.1 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lin   ', [round(i['estep'],2) for i in d['iterations']], d['entries'])"
MUXGL_PROBE_FLAGS=64 python tools/scale_probe.py fmx 4 0.1 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-lin', [round(i['estep'],2) for i in d['iterations']], d['entries'])"
