# usage: bash tools/_bench_variants.sh <outdir> [variant names...]; "default" = the in-tree libmcpt.so
out=$1; shift
mkdir -p $out
for v in "$@"; do
  if [ "$v" = default ]; then unset MCPT_LIB; else export MCPT_LIB=$PWD/montecarlopathtracing_amd/csrc/variants/libmcpt_$v.so; fi
  MCPT_PRINT_DIAG=1 timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err || { echo "$v FAILED"; tail -5 $out/bench_$v.err; continue; }
  python - $v $out <<'PY'
import json,sys
v,out=sys.argv[1],sys.argv[2]
d=json.load(open('%s/bench_%s.json'%(out,v))); r=d['roofline']
print('%-12s ms/frame %.2f  trace avg ms %.3f  nodes/ray %.2f tris/ray %.2f'%(v, d['ms_per_step'], r['avg_launch_ms'], d['nodes_per_ray'], d['tris_per_ray']))
PY
  grep -h "deferred" $out/bench_$v.err | tail -1
done
