#!/bin/bash
# timing experiments of the configs[1] sweep: the default build, env-var variants, and every variant library under
# popscle_amd/lib/var (tools/build_variant.sh)
cd /root/repo
python tools/quad_time.py 1 400
for w in $FINISH_GRIDS; do MUXGL_FINISH_GRID=$w python tools/quad_time.py 1 400 | sed "s/^/finish grid=$w /"; done
for f in popscle_amd/lib/var/libmuxgl_*.so; do [ -e $f ] && MUXGL_LIB=$PWD/$f python tools/quad_time.py 1 400; done
