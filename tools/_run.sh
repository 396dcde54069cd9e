mkdir -p gpurun_out/r2z
for v in default lw4; do
  if [ "$v" = default ]; then unset MCPT_LIB; else export MCPT_LIB=$PWD/montecarlopathtracing_amd/csrc/variants/libmcpt_$v.so; fi
  for sc in "interior" "veach-mis --spp 100" "synthetic --spp 16"; do
    timeout -k 10 300 python bench.py --scene $sc --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2z/b.json 2> gpurun_out/r2z/b.err || { tail -3 gpurun_out/r2z/b.err; continue; }
    python -c "
import json
d=json.load(open('gpurun_out/r2z/b.json')); print('$v', '$sc', 'ms/frame %.2f'%(d['ms_per_step']))"
  done
done
