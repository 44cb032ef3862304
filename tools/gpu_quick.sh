#!/bin/bash
# quick GPU check used during kernel iteration: demuxlet parity tests + short bench (prints the kernel times)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_demux_gpu.py -x -q > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_quick.log
timeout 600 python bench.py --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench_quick.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LLs/s %.3e  step %.3f ms  kernels %s  frac %.3f' % (d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac']))"
