#!/bin/bash
# Runs on the GPU box (via gpurun): FP64 instruction-mix counters (own --pmc passes, no trace domains) of the bench
# command and of the larger shapes, plus a clock/power log during a sustained bench run.
mkdir -p /root/repo/gpurun_out; cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out
rocprofv3 -L > $O/counters_avail.txt 2>&1
pass() {  # tag, counters, command...
  tag=$1; ctr=$2; shift 2
  timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $O/prof_$tag -- "$@" > $O/prof_$tag.log 2>&1
  echo "$tag rc=$?"
}
B="python /root/repo/bench.py --no-cpu-baseline --steps 3 --warmup 1 --ramp-seconds 0"
F64A="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
MIX="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES"
pass c1_f64 "$F64A" $B
pass c1_mix "$MIX" $B
pass d2_f64 "$F64A" python /root/repo/tools/scale_probe.py demux 2 ${DEMUX2_SCALE:-0.05}
pass d2_mix "$MIX" python /root/repo/tools/scale_probe.py demux 2 ${DEMUX2_SCALE:-0.05}
pass f4_f64 "$F64A" python /root/repo/tools/scale_probe.py fmx 4 ${FMX4_SCALE:-0.03} 2
pass f4_mix "$MIX" python /root/repo/tools/scale_probe.py fmx 4 ${FMX4_SCALE:-0.03} 2
pass f3_f64 "$F64A" python /root/repo/tools/scale_probe.py fmx 3 0.3 2
# clocks and power while the bench runs sustained (no profiler attached)
cd /root/repo
( for i in $(seq 1 24); do rocm-smi --showclocks --showpower --showtemp --json 2>/dev/null | tr -d '\n'; echo; sleep 0.5; done ) > $O/smi_during_bench.jsonl &
SMI=$!
python bench.py --no-cpu-baseline --steps 12000 --warmup 200 > $O/bench_sustained.log 2>&1; echo "bench rc=$?"
wait $SMI
tail -1 $O/bench_sustained.log | cut -c1-400
