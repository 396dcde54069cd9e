mkdir -p gpurun_out/r3e
for v in default sh7 sh12 sh14 sh16; do
  if [ "$v" = default ]; then unset MCPT_LIB; else export MCPT_LIB=$PWD/montecarlopathtracing_amd/csrc/variants/libmcpt_$v.so; fi
  for n in 8 1; do
  if [ $n = 1 ]; then a=""; else a="--sim-world $n"; fi
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline $a > gpurun_out/r3e/s.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/r3e/s.json')); print('$v N=$n ms/frame %.2f trace avg %.3f'%(d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  done
done
