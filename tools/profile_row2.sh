#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel traces of the two-per-lane row kernels at 32 samples / clusters
# (tools/v_probe.py, tools/k_probe.py); tools/prof_summary.py --kt condenses each for profiles/.
mkdir -p /root/repo/gpurun_out; cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out
rm -rf $O/prof_demux_v32 $O/prof_fmx_k32
VPROBE_DEFAULT_GRID=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_demux_v32 -- python /root/repo/tools/v_probe.py 32 10000 50000 > $O/prof_demux_v32.log 2>&1
echo "demux_v32 rc=$?"; grep "^{" $O/prof_demux_v32.log | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fmx_k32 -- python /root/repo/tools/k_probe.py 32 20000 > $O/prof_fmx_k32.log 2>&1
echo "fmx_k32 rc=$?"; grep "^{" $O/prof_fmx_k32.log | cut -c1-300
