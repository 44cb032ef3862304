#!/bin/bash
# E-step timing of configs[3] for chunk lengths / variant libraries (tools/build_variant.sh)
cd /root/repo
run() { python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_rank0_last_iteration']; print('$1', round(d['ms_per_step'],3), {x: round(k[x],3) for x in k})"; }
run default
for c in $FMX_CHS; do MUXGL_FMX_CH=$c run chunk=$c; done
for f in popscle_amd/lib/var/libmuxgl_*.so; do [ -e $f ] && MUXGL_LIB=$PWD/$f run $(basename $f); done
