#!/bin/bash
# M-step timing with / without the LDS table of assignments (MUXGL_MSTEP_NO_TABLE), configs[3] and configs[4] at 10 %
cd /root/repo
run() { python bench.py $2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0_last_iteration']; print('$1', round(d['ms_per_step'],3), {x: round(k[x],3) for x in k})"; }
run c3_table "--config 3 --steps 100 --warmup 10"
MUXGL_MSTEP_NO_TABLE=1 run c3_notable "--config 3 --steps 100 --warmup 10"
run c4s_table "--config 4 --scale 0.1 --steps 10 --warmup 2"
MUXGL_MSTEP_NO_TABLE=1 run c4s_notable "--config 4 --scale 0.1 --steps 10 --warmup 2"
