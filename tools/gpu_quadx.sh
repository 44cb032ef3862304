#!/bin/bash
# timing experiments of the quad kernel: the default build and every variant library under popscle_amd/lib/var
cd /root/repo
python tools/quad_time.py 1 400
for f in popscle_amd/lib/var/libmuxgl_*.so; do
  case $f in *x17*) MUXGL_LIB=$PWD/$f python tools/quad_time.py 1 2 | sort | uniq | head -40;; *) MUXGL_LIB=$PWD/$f python tools/quad_time.py 1 400;; esac
done
