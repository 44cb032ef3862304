#!/bin/bash
# wave-cycle, instruction and LDS counters of the kernels matching a regex (own --pmc passes, no trace domains).
# usage: bash tools/gpu_pmc.sh <tag> <kernel-regex> <command...>   (absolute paths in the command: it runs from /tmp)
tag=$1; re=$2; shift 2
mkdir -p /root/repo/gpurun_out; cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out
pass() { t=$1; ctr=$2; shift 2; rm -rf $O/pmc_${tag}_$t; timeout 300 rocprofv3 --pmc $ctr --kernel-include-regex "$re" --output-format csv -d $O/pmc_${tag}_$t -- "$@" > $O/pmc_${tag}_$t.log 2>&1; echo "$t rc=$?"; }
pass cyc "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "$@"
pass ins "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES" "$@"
cd /root/repo
for t in cyc ins; do python tools/pmc_kernel.py "" $O/pmc_${tag}_$t; done
find $O/pmc_${tag}_cyc $O/pmc_${tag}_ins -type f ! -name "*counter_collection.csv" -delete
