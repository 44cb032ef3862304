#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass per counter group of `bench.py --config $1` (short), printing the
# per-launch averages of the kernels matching $2.  usage: bash tools/gpu_pmc.sh <config> <kernel regex> [bench args]
CFG=$1; PAT=$2; shift 2
cd /tmp; export TMPDIR=/tmp
B="python /root/repo/bench.py --config $CFG --no-cpu-baseline --no-fmx-leg --steps 3 --warmup 1 --ramp-seconds 0 $*"
O=/root/repo/gpurun_out/pmc_c$CFG; rm -rf $O; mkdir -p $O
i=0
for ctr in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_LEVEL_WAVES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $O/p$i -- $B > $O/p$i.log 2>&1 || echo "pass $i rc=$?"
done
python - "$O" "$PAT" <<'P'
import csv, glob, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(sys.argv[2], r["Kernel_Name"]):
            k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])[:60]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v)/len(v):14.4g}  (n={len(v)})")
P
