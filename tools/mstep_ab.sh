#!/bin/bash
# Runs on the GPU box: M-step with and without the coded stream (MUXGL_MSTEP_NO_CODES), configs[3] and configs[4]
cd /root/repo
run() { python bench.py --config $1 --steps $2 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0_last_iteration']; print('$3 c$1', round(d['ms_per_step'],3), {x: round(k[x],3) for x in k})"; }
for i in 1 2; do run 3 100 codes; MUXGL_MSTEP_NO_CODES=1 run 3 100 plain; done
run 4 4 codes; MUXGL_MSTEP_NO_CODES=1 run 4 4 plain
