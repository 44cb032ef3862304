#!/bin/bash
# timing experiments of the configs[1] sweep: the default build, env-var variants, and every variant library under
# popscle_amd/lib/var (tools/build_variant.sh)
cd /root/repo
python tools/quad_time.py 1 400
for w in $OCT_CHS; do MUXGL_OCT_CH=$w python tools/quad_time.py 1 400 | sed "s/^/chunk=$w /"; done
for f in popscle_amd/lib/var/libmuxgl_*.so; do [ -e $f ] && MUXGL_LIB=$PWD/$f python tools/quad_time.py 1 400; done
