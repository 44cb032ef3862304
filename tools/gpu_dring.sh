#!/bin/bash
# demux_wave.hip with the ring in LDS: demuxlet parity tests, then the sweep time of configs[2] at 20 %
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_demux_gpu.py tests/test_group_gpu.py tests/test_large_gpu.py -x -q -k "not freemuxlet and not fmx" > gpurun_out/pytest_dring.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_dring.log
probe() { python tools/scale_probe.py demux 2 ${DRING_SCALE:-0.2} | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', {k:d[k] for k in ('step_s','sweep_ms','call_ms','entries_per_s','singlet_acc')})"; }
probe lin
MUXGL_PROBE_FLAGS=64 probe no-lin
