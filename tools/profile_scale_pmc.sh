#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes (each in its own run, no trace domains besides the kernel trace that
# --pmc implies) of the larger shapes; tools/prof_summary.py merges them with the kernel trace of profile_scale.sh.
mkdir -p /root/repo/gpurun_out; cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out
pass() {  # tag, counter-set name, counters, probe args...
  tag=$1; set_=$2; ctr=$3; shift 3
  timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $O/prof_${tag}_$set_ -- python /root/repo/tools/scale_probe.py "$@" > $O/prof_${tag}_$set_.log 2>&1
  echo "$tag $set_ rc=$?"
}
shape() {
  tag=$1; shift
  pass $tag fetch "FETCH_SIZE" "$@"
  pass $tag write "WRITE_SIZE" "$@"
  pass $tag sq "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "$@"
}
shape demux2 demux 2 ${DEMUX2_SCALE:-0.1}
shape fmx3 fmx 3 0.5 2
shape fmx4 fmx 4 ${FMX4_SCALE:-0.05} 2
shape fmxold1 fmxold 1 0.5
