mkdir -p gpurun_out/r3i
python -m pytest tests -q -m gpu > gpurun_out/r3i/tests.log 2>&1; rc=$?
tail -4 gpurun_out/r3i/tests.log | cut -c1-220
bash tools/_bench_variants.sh gpurun_out/r3i default
for sc in "synthetic --spp 16" "interior" "veach-mis --spp 100"; do
    timeout -k 10 300 python bench.py --scene $sc --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3i/b.json 2> gpurun_out/r3i/b.err || { tail -3 gpurun_out/r3i/b.err; continue; }
    python -c "
import json
d=json.load(open('gpurun_out/r3i/b.json')); print('$sc', 'ms/frame %.2f Mrays/s %.0f'%(d['ms_per_step'], d['value']))"
done
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --sim-world 8 > gpurun_out/r3i/s.json 2>/dev/null; python -c "
import json
d=json.load(open('gpurun_out/r3i/s.json')); print('sim8 ms/frame %.2f'%d['ms_per_step'])"
exit $rc
