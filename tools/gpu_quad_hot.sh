#!/bin/bash
# Timing experiment (wrong results, nothing kept): the quad kernels with every row gather hitting the same 256 rows --
# an upper bound on what the row path (L2 -> L1, address processing) costs at config 1 and config 3.
probe() { python tools/scale_probe.py demux 1 1.0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 demux', {k:d[k] for k in ('sweep_ms','reduce_ms')})"
MUXGL_PROBE_NO_GREEDY=1 python tools/scale_probe.py fmx 3 1.0 4 | python -c "import sys,json; d=json.loads(sys.stdin.read()); i=d['iterations'][-1]; print('$1 fmx', {k:round(i[k],3) for k in ('estep','mstep')})"; }
probe "as shipped"
sed -i 's/gmq + (size_t)sidx \* 32/gmq + (size_t)(sidx \& 255) * 32/; s/gpq + (size_t)s \* 48/gpq + (size_t)(s \& 255) * 48/' popscle_amd/csrc/demux_quad.hip
sed -i 's/ceq + (size_t)sidx \* 16/ceq + (size_t)(sidx \& 255) * 16/; s/cgpq + (size_t)s \* 48/cgpq + (size_t)(s \& 255) * 48/' popscle_amd/csrc/fmx_quad.hip
grep -c "& 255" popscle_amd/csrc/demux_quad.hip popscle_amd/csrc/fmx_quad.hip
make -C popscle_amd/csrc > gpurun_out/hot_build.log 2>&1 || tail -3 gpurun_out/hot_build.log
probe "hot rows"
