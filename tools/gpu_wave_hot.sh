#!/bin/bash
# Timing experiment (wrong results, nothing kept): the demuxlet wave kernels with every marker row redirected to 256
# cache-resident rows -- what the memory side still costs at 20 % of configs[2].
probe() { python tools/scale_probe.py demux 2 0.2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', {k:d[k] for k in ('sweep_ms','call_ms')})"; }
probe "as shipped"
cp popscle_amd/csrc/demux_wave.hip /tmp/dw.orig
sed -i 's/const double\* row = gp + (size_t)ids\[s\] \* V3 + jo;/const double* row = gp + (size_t)(ids[s] \& 255) * V3 + jo;/' popscle_amd/csrc/demux_wave.hip
make -C popscle_amd/csrc > gpurun_out/hot_build.log 2>&1 || tail -3 gpurun_out/hot_build.log
probe "hot marker rows"
