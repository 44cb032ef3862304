#!/bin/bash
# wave-cycle and LDS counters of the freemuxlet wave E-step at 10 % of configs[4] (own --pmc passes, no trace domains)
mkdir -p /root/repo/gpurun_out; cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out
pass() { tag=$1; ctr=$2; shift 2; rm -rf $O/pmc_$tag; MUXGL_PROBE_NO_GREEDY=1 timeout 240 rocprofv3 --pmc $ctr --kernel-include-regex 'fmx_estep_wave' --output-format csv -d $O/pmc_$tag -- python /root/repo/tools/scale_probe.py fmx 4 0.1 2 > $O/pmc_$tag.log 2>&1; echo "$tag rc=$?"; }
pass cyc "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
pass lds "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SALU"
pass misc "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_LDS_UNALIGNED_STALL"
cd /root/repo
for t in cyc lds misc; do python tools/pmc_kernel.py fmx_estep_wave $O/pmc_$t; done
# keep only the csv files small: drop everything but the counter collections
find $O/pmc_cyc $O/pmc_lds $O/pmc_misc -type f ! -name "*counter_collection.csv" -delete
