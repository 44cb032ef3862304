cd /root/repo
for r in 1 2 3; do for ch in ${CHS:-96 128 160 192 256}; do
  MUXGL_FMX_CH=$ch python bench.py --config 3 --steps 200 --warmup 20 --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ch$ch', round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms_rank0_last_iteration'].items()})"
done; done
