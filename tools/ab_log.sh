cd /root/repo
run() { python bench.py --config $1 --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_ms') or d.get('kernel_ms_rank0_last_iteration'); print('$2 c$1', round(d['ms_per_step'],3), {x: round(v,3) for x,v in k.items()})"; }
for i in 1 2; do
  run 2 poslog; MUXGL_LIB=$PWD/popscle_amd/lib/var/libmuxgl_liblog.so run 2 liblog
  run 4 poslog; MUXGL_LIB=$PWD/popscle_amd/lib/var/libmuxgl_liblog.so run 4 liblog
done
python tools/quad_time.py 1 600; python tools/quad_time.py 1 600
