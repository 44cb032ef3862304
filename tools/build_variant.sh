#!/bin/bash
# builds popscle_amd/lib/var/libmuxgl_<tag>.so with one translation unit recompiled under extra flags (kernel timing
# experiments; select with MUXGL_LIB=...):  bash tools/build_variant.sh <tag> <unit[.hip]> <flags...>
#   e.g.  tools/build_variant.sh w4 demux_oct -DOCT_WAVES=4   (tuning constants; timing-only arms live as patches under tools/patches)
set -e
tag=$1; src=$2; shift 2; flags="$*"
cd "$(dirname "$0")/../popscle_amd/csrc"
mkdir -p ../lib/var
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -Wall -Wno-unused-function -I../../include $flags -c $base.hip -o ../lib/var/${base}_$tag.o
# link exactly the Makefile's SRCS (stale objects of removed units may sit in ../lib), all of them up to date
make -s
objs=$(make -s -pn | sed -n 's/^SRCS = //p' | head -1 | tr ' ' '\n' | sed 's/\.hip$//' | grep -vx "${base}" | sed 's#^#../lib/#; s#$#.o#')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var/libmuxgl_$tag.so $objs ../lib/var/${base}_$tag.o
echo built ../lib/var/libmuxgl_$tag.so
