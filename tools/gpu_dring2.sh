#!/bin/bash
# sweep time of configs[2] at 20 % for several builds of demux_wave.hip (DRING_CFGS: ';'-separated -D lists)
mkdir -p gpurun_out
probe() { python tools/scale_probe.py demux 2 ${DRING_SCALE:-0.2} | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', {k:d[k] for k in ('step_s','sweep_ms','call_ms','singlet_acc')})"; }
probe "shipped lib, lin"
MUXGL_PROBE_FLAGS=64 probe "shipped lib, no-lin"
IFS=';' read -ra CFGS <<< "${DRING_CFGS:-}"
for cfg in "${CFGS[@]}"; do
  touch popscle_amd/csrc/demux_wave.hip
  make -C popscle_amd/csrc EXTRA="$cfg" > gpurun_out/dring_build.log 2>&1 || { echo "build failed $cfg"; tail -5 gpurun_out/dring_build.log; continue; }
  probe "$cfg lin"
done
