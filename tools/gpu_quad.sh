#!/bin/bash
# quad kernel with the linear-entry loop: demuxlet parity tests, the full-size config-1 checks, then bench config 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_demux_gpu.py tests/test_large_gpu.py tests/test_group_gpu.py tests/test_cli_gpu.py -x -q -k "not freemuxlet and not fmx" > gpurun_out/pytest_quad.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_quad.log
python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-fmx-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench ', d['ms_per_step'], d['kernel_ms'])"
for f in 0 64; do MUXGL_PROBE_FLAGS=$f python tools/scale_probe.py demux 1 1.0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags=$f', {k:d[k] for k in ('step_s','sweep_ms','reduce_ms','singlet_acc')})"; done
