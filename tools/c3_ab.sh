# A/B of builds of the library on one box: configs[3] bench leg (popscle_amd/lib/var/libmuxgl_<name>.so, tools/build_variant.sh)
cd /root/repo
run() { python bench.py --config $1 ${BARGS:---steps 200 --warmup 20} --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_ms') or d.get('kernel_ms_rank0_last_iteration'); print('$2 c$1', round(d['ms_per_step'],4), {x: round(v,4) for x,v in k.items()})"; }
for i in ${ROUNDS:-1 2 3}; do
  run ${CFG:-3} new
  for v in "$@"; do MUXGL_LIB=$PWD/popscle_amd/lib/var/libmuxgl_$v.so run ${CFG:-3} $v; done
done
