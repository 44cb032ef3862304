cd /root/repo
run() { python bench.py --no-fmx-leg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$1', round(d['ms_per_step'],4), {x: round(k[x],4) for x in k})"; }
run default
for f in popscle_amd/lib/var/libmuxgl_*.so; do MUXGL_LIB=$PWD/$f run $(basename $f); done
