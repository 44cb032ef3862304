#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_demux_gpu.py tests/test_large_gpu.py -x -q -k "not config4 and not config3 and not pair_matrix" > gpurun_out/pytest_lin2.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_lin2.log
python tools/scale_probe.py demux 2 0.25 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lin   ', d['sweep_ms'], d['step_s'], d['entries'])"
MUXGL_PROBE_FLAGS=64 python tools/scale_probe.py demux 2 0.25 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-lin', d['sweep_ms'], d['step_s'], d['entries'])"
