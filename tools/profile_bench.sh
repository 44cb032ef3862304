#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate PMC passes of the bench command, into gpurun_out/prof_*.
# usage: bash tools/profile_bench.sh [extra bench.py args]      (clean local gpurun_out/prof_* first)
mkdir -p /root/repo/gpurun_out; cd /tmp; export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline $*"
O=/root/repo/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- $B --steps 200 --warmup 20 --ramp-seconds 0.3 > $O/prof_kt.log 2>&1; echo "kt rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $B --steps 3 --warmup 1 --ramp-seconds 0 > $O/prof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $B --steps 3 --warmup 1 --ramp-seconds 0 > $O/prof_write.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/prof_sq -- $B --steps 3 --warmup 1 --ramp-seconds 0 > $O/prof_sq.log 2>&1; echo "sq rc=$?"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $O/prof_cache -- $B --steps 3 --warmup 1 --ramp-seconds 0 > $O/prof_cache.log 2>&1; echo "cache rc=$?"
grep "^{" $O/prof_kt.log | cut -c1-160
