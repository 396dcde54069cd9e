mkdir -p gpurun_out/r3g
MCPT_LIB=$PWD/montecarlopathtracing_amd/csrc/variants/libmcpt_ib3.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "closest_hit or bulk or overflow or edge or pipelines or wavefront_iterations or image_matches" > gpurun_out/r3g/tests.log 2>&1 || { tail -30 gpurun_out/r3g/tests.log | cut -c1-250; exit 1; }
tail -2 gpurun_out/r3g/tests.log
bash tools/_bench_variants.sh gpurun_out/r3g default ib3
for v in default ib3; do
  if [ "$v" = default ]; then unset MCPT_LIB; else export MCPT_LIB=$PWD/montecarlopathtracing_amd/csrc/variants/libmcpt_$v.so; fi
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --sim-world 8 > gpurun_out/r3g/s.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/r3g/s.json')); print('$v sim8 ms/frame %.2f trace avg %.3f'%(d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
