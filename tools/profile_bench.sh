#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate PMC passes (one counter set per run, no other trace domains)
# of a bench.py configuration, into gpurun_out/prof_c<config>_*.  Afterwards, in the container:
#   python tools/prof_summary.py --kt gpurun_out/prof_c1_kt --fetch gpurun_out/prof_c1_fetch --write gpurun_out/prof_c1_write \
#       --sq gpurun_out/prof_c1_sq --out profiles/r02x_demux_config1
#   python tools/traffic_update.py --config 1 --prefix gpurun_out/prof_c1
# usage: bash tools/profile_bench.sh <config> [steps-for-the-trace] [extra bench.py args]
CFG=${1:-1}; KT_STEPS=${2:-200}; shift 2 2>/dev/null
mkdir -p /root/repo/gpurun_out; cd /tmp; export TMPDIR=/tmp
B="python /root/repo/bench.py --config $CFG --no-cpu-baseline --no-fmx-leg $*"
O=/root/repo/gpurun_out
P=$O/prof_c${CFG}
pass() {  # tag, counters, bench args
  tag=$1; ctr=$2; shift 2
  if [ "$ctr" = "KT" ]; then
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_$tag -- $B "$@" > ${P}_$tag.log 2>&1
  else
    timeout 600 rocprofv3 --pmc $ctr --output-format csv -d ${P}_$tag -- $B "$@" > ${P}_$tag.log 2>&1
  fi
  echo "c$CFG $tag rc=$?"
}
SHORT="--steps 3 --warmup 1 --ramp-seconds 0"
# shader clock and package power while the trace pass runs (rocm-smi every 0.25 s; the summary quotes the busy samples)
( while true; do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > ${P}_smi.jsonl &
SMI=$!
pass kt KT --steps $KT_STEPS --warmup $((KT_STEPS / 10 + 1)) --ramp-seconds 0.3
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
pass fetch "FETCH_SIZE" $SHORT
pass write "WRITE_SIZE" $SHORT
pass f64 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" $SHORT
pass mix "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES" $SHORT
pass sq "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" $SHORT
pass cache "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" $SHORT
grep "^{" ${P}_kt.log | cut -c1-200
