#!/bin/bash
# configs[2] pass times for variant libraries (tools/build_variant.sh)
cd /root/repo
run() { python bench.py --config 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$1', round(d['ms_per_step'],2), {x: round(k[x],2) for x in k})"; }
run default
for f in popscle_amd/lib/var/libmuxgl_*.so; do [ -e $f ] && MUXGL_LIB=$PWD/$f run $(basename $f); done
