#!/bin/bash
# fmx_wave.hip: parity of the K > 16 E-step tests, then the E-step time of configs[4] at 10 % for several builds
# (rebuilt on the box: hipcc is in the image).  RING_CFGS: ';'-separated lists of -D flags.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fmx_gpu.py tests/test_group_gpu.py tests/test_fmx_shard_gpu.py -x -q -k "${RING_K:-64 or 70 or 100 or 33 or 130 or 20 or 32}" > gpurun_out/pytest_ring.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_ring.log
probe() {
  python tools/scale_probe.py fmx 4 ${RING_SCALE:-0.1} 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', [round(i['estep'],2) for i in d['iterations']], d['entries'])"
}
probe "default"
MUXGL_PROBE_FLAGS=64 python tools/scale_probe.py fmx 4 ${RING_SCALE:-0.1} 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-lin', [round(i['estep'],2) for i in d['iterations']], d['entries'])"
IFS=';' read -ra CFGS <<< "${RING_CFGS:-}"
for cfg in "${CFGS[@]}"; do
  touch popscle_amd/csrc/fmx_wave.hip
  make -C popscle_amd/csrc EXTRA="$cfg" > gpurun_out/ring_build.log 2>&1 || { echo "build failed $cfg"; tail -5 gpurun_out/ring_build.log; continue; }
  probe "$cfg"
done
