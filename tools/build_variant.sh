#!/bin/bash
# builds popscle_amd/lib/var/libmuxgl_<name>.so with one translation unit recompiled with extra flags
# usage: tools/build_variant.sh <name> <unit> <flags...>     e.g.  tools/build_variant.sh skipgen demux_ring -DRING_TIMING_SKIP_GEN
set -e
name=$1; unit=$2; shift 2
cd /root/repo/popscle_amd/csrc
mkdir -p ../lib/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -I../../include "$@" -c $unit.hip -o ../lib/var/${unit}_$name.o
objs=$(ls ../lib/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var/libmuxgl_$name.so $objs ../lib/var/${unit}_$name.o
echo built ../lib/var/libmuxgl_$name.so
