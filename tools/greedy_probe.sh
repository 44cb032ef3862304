#!/bin/bash
# Runs on the GPU box: the greedy tests and the greedy timing of configs[3]
cd /root/repo
python -m pytest tests/test_fmx_gpu.py -m gpu -x -q -k "greedy" -p no:cacheprovider 2>&1 | tail -15
python -m pytest tests/test_ref_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
MUXGL_TIMING=1 python tools/greedy_time.py 2>&1 | grep -v "^\[muxgl\] greedy_init:   " | tail -25
