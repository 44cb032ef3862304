#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel traces of the larger shapes (tools/scale_probe.py), one directory
# per shape under gpurun_out/prof_<tag>; tools/prof_summary.py --kt condenses each for profiles/.
mkdir -p /root/repo/gpurun_out; cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out
run() {  # tag, probe args...
  tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python /root/repo/tools/scale_probe.py "$@" > $O/prof_$tag.log 2>&1
  echo "$tag rc=$?"; grep "^{" $O/prof_$tag.log | cut -c1-400
}
run demux2 demux 2 ${DEMUX2_SCALE:-0.25}
run fmx3 fmx 3 1.0 5
run fmx4 fmx 4 ${FMX4_SCALE:-0.1} 3
run fmxold1 fmxold 1 1.0
