#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel traces of the row kernels for 17..32 samples / clusters -- two per
# lane at 32, broadcast extras at 20 -- (tools/v_probe.py, tools/k_probe.py); tools/prof_summary.py --kt condenses each
# for profiles/.
mkdir -p /root/repo/gpurun_out; cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out
for n in 32 20; do
  rm -rf $O/prof_demux_v$n $O/prof_fmx_k$n
  VPROBE_DEFAULT_GRID=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_demux_v$n -- python /root/repo/tools/v_probe.py $n 10000 50000 > $O/prof_demux_v$n.log 2>&1
  echo "demux_v$n rc=$?"; grep "^{" $O/prof_demux_v$n.log | cut -c1-300
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fmx_k$n -- python /root/repo/tools/k_probe.py $n 20000 > $O/prof_fmx_k$n.log 2>&1
  echo "fmx_k$n rc=$?"; grep "^{" $O/prof_fmx_k$n.log | cut -c1-300
done
# PMC passes of the same shapes, each in its own run (no trace domains besides the kernel trace --pmc implies)
for n in 32 20; do
  for set_ in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"; do
    name=${set_%%:*}; ctr=${set_#*:}
    rm -rf $O/prof_demux_v${n}_$name $O/prof_fmx_k${n}_$name
    VPROBE_DEFAULT_GRID=1 timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $O/prof_demux_v${n}_$name -- python /root/repo/tools/v_probe.py $n 10000 50000 > $O/prof_demux_v${n}_$name.log 2>&1
    echo "demux_v$n $name rc=$?"
    timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $O/prof_fmx_k${n}_$name -- python /root/repo/tools/k_probe.py $n 20000 > $O/prof_fmx_k${n}_$name.log 2>&1
    echo "fmx_k$n $name rc=$?"
  done
done
